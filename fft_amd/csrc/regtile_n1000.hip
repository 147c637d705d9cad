// regtile_n1000.hip — n_fft = 1000 (= 40 x 25) instantiations of the mixed-radix register-resident kernel (own TU)
#include "kernel_regtile_mixed.h"
#include "kernel_regtile_mixed_grad.h"
// forward: 40 x 25; the gate gradient's half exchange needs an even RS, so it runs the transposed factorisation 25 x 40
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(40, 25) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(25, 40) }
