// regtile_n1024.hip — n_fft = 1024 (= 32 x 32) instantiations of the register-resident kernel (own TU: parallel builds)
#include "kernel_regtile_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_LAUNCHER(32, 32) SFFT_DEFINE_GATE_GRAD_LAUNCHER(32, 32) }
