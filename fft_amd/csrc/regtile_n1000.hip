// regtile_n1000.hip — n_fft = 1000 (= 40 x 25) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(40, 25) }
