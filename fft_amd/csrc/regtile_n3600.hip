// regtile_n3600.hip — n_fft 3600 = 60 x 60: mixed-radix register-resident kernels (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(60, 60) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(60, 60) }
