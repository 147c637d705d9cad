// regtile_n2048.hip — n_fft = 2048 (= 64 x 32) instantiations of the register-resident kernel (own TU: parallel builds)
#include "kernel_regtile_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_LAUNCHER(64, 32) SFFT_DEFINE_GATE_GRAD_LAUNCHER(64, 32) }
