// regtile_r64.hip — n_fft = 4096 instantiations of the register-resident kernel (own TU: parallel builds)
#include "kernel_regtile.h"
namespace sfft { SFFT_DEFINE_REGTILE_LAUNCHER(64) }
