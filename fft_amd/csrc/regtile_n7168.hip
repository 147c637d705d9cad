// regtile_n7168.hip — n_fft = 7168 (= 56 x 128, lane-pair split of the 128-point transform): own TU
#include "kernel_regtile_long_grad.h"
namespace sfft {
hipError_t launch_regtile_long_7168(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) { return launch_regtile_long<56>(a, in_bf16, out_bf16, mode, stream); }
hipError_t launch_gate_grad_long_7168(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream) { return launch_gate_grad_long<56>(a, io_bf16, general, stream); }
}
