// regtile_n5120.hip — n_fft = 5120 (= 40 x 128, lane-pair split of the 128-point transform): own TU
#include "kernel_regtile_long_grad.h"
namespace sfft {
hipError_t launch_regtile_long_5120(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) { return launch_regtile_long<40>(a, in_bf16, out_bf16, mode, stream); }
hipError_t launch_gate_grad_long_5120(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream) { return launch_gate_grad_long<40>(a, io_bf16, general, stream); }
}
