// regtile_n2560.hip — n_fft = 2560 (= 64 x 40) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(64, 40) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(64, 40) }
