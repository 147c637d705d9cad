// regtile_n8192.hip — n_fft = 8192 (= 64 x 128, lane-pair split of the 128-point transform): own TU
#include "kernel_regtile_long_grad.h"
namespace sfft {
hipError_t launch_regtile_long_8192(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) { return launch_regtile_long<64>(a, in_bf16, out_bf16, mode, stream); }
hipError_t launch_gate_grad_long_8192(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream) { return launch_gate_grad_long<64>(a, io_bf16, general, stream); }
}
