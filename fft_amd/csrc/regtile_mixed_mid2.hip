// regtile_mixed_mid2.hip — n_fft 1200 = 40 x 30, 1920 = 48 x 40: mixed-radix register-resident kernels (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(40, 30) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(40, 30) SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(48, 40) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(48, 40) }
