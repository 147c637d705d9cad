// regtile_n512.hip — n_fft = 512 (= 32 x 16) instantiations of the register-resident kernel (own TU: parallel builds)
#include "kernel_regtile_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_LAUNCHER(32, 16) SFFT_DEFINE_GATE_GRAD_LAUNCHER(32, 16) }
