// regtile_n1280.hip — n_fft = 1280 (= 40 x 32) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(40, 32) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(40, 32) }
