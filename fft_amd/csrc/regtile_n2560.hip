// regtile_n2560.hip — n_fft = 2560 (= 64 x 40) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(64, 40) }
