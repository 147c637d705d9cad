// regtile_wide.hip — 32-channel (whole-line) tiles for n_fft = 256, 512, 1024 (kernel_regtile_wide.h); own TU: parallel builds
#include "kernel_regtile_wide.h"
#include <atomic>
namespace sfft { SFFT_DEFINE_REGTILE_WIDE_LAUNCHER(16, 16) SFFT_DEFINE_REGTILE_WIDE_LAUNCHER(32, 16) SFFT_DEFINE_REGTILE_WIDE_LAUNCHER(32, 32) SFFT_DEFINE_REGTILE_WIDE_LAUNCHER(64, 32) }
