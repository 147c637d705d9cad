// regtile_n2000.hip — n_fft = 2000 (= 50 x 40) instantiations of the mixed-radix register-resident kernel (own TU)
// This unit keeps the canonical-NaN patch of the bf16 stores (kernel_regtile.h, round 6): the 50 x 40 kernels run at 128 registers with 100 of
// them holding data, and WITHOUT the patch's instructions hipcc's allocation of the bf16 -> bf16 fast mode spills the lane's row index — every
// twiddle-base load then waits for a scratch reload (26 serialised loads: fft_amd/isa_lint.py refused the build; 1.375 -> 1.48 ms at
// (384, 2000, 768), tools/gpu_jobs/r06_n2000.sh).  Same bits for every non-NaN value either way.
#define SPECTRE_BF16_CANONICAL_NAN 1
#include "kernel_regtile_mixed_grad.h"
namespace sfft { SFFT_DEFINE_REGTILE_MIXED_LAUNCHER_SAME_DTYPE(50, 40) SFFT_DEFINE_GATE_GRAD_MIXED_LAUNCHER(50, 40) }
