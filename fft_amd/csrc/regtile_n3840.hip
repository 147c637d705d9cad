// regtile_n3840.hip — n_fft = 3840 (= 64 x 60) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(64, 60) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(64, 60) }
