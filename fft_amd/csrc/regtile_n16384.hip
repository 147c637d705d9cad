// regtile_n16384.hip — n_fft = 16384 (= 64 x 256, lane-quad split of the 256-point transform): own TU
#include "kernel_regtile_quad.h"
namespace sfft { hipError_t launch_regtile_quad_16384(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) { return launch_regtile_quad<64>(a, in_bf16, out_bf16, mode, stream); } }
