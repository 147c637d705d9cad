// regtile_n768.hip — n_fft = 768 (= 32 x 24) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(32, 24) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(32, 24) }
