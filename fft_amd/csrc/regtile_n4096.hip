// regtile_n4096.hip — n_fft = 4096 (= 64 x 64) instantiations of the register-resident kernel (own TU: parallel builds)
#include "kernel_regtile_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_LAUNCHER(64, 64) SFFT_DEFINE_GATE_GRAD_LAUNCHER(64, 64) }
