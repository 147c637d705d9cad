// regtile_n6144.hip — n_fft = 6144 (= 48 x 128, lane-pair split of the 128-point transform): own TU
#include "kernel_regtile_long_grad.h"
namespace sfft {
hipError_t launch_regtile_long_6144(const RegtileArgs& a, bool in_bf16, bool out_bf16, int mode, hipStream_t stream) { return launch_regtile_long<48>(a, in_bf16, out_bf16, mode, stream); }
hipError_t launch_gate_grad_long_6144(const GateGradArgs& a, bool io_bf16, bool general, hipStream_t stream) { return launch_gate_grad_long<48>(a, io_bf16, general, stream); }
}
