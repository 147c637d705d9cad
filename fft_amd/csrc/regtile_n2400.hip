// regtile_n2400.hip — n_fft 2400 = 60 x 40: mixed-radix register-resident kernels (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(60, 40) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(60, 40) }
