// regtile_n8192.hip — n_fft = 8192 (= 64 x 128, lane-pair split of the 128-point transform): own TU
#include "kernel_regtile_long.h"
namespace sfft { hipError_t launch_regtile_long_8192(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) { return launch_regtile_long<64>(a, in_bf16, out_bf16, mode, stream); } }
