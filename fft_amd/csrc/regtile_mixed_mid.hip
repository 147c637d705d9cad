// regtile_mixed_mid.hip — n_fft 640 = 32 x 20, 960 = 32 x 30: mixed-radix register-resident kernels (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(32, 20) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(32, 20) SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(32, 30) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(32, 30) }
