// regtile_n3000.hip — n_fft = 3000 (= 60 x 50) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER(60, 50) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(60, 50) }
