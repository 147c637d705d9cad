// regtile_r16.hip — n_fft = 256 instantiations of the register-resident kernel (own TU: parallel builds)
#include "kernel_regtile.h"
namespace sfft { SFFT_DEFINE_REGTILE_LAUNCHER(16) }
