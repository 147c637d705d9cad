// regtile_widep.hip — persistent whole-line tiles (kernel_regtile_widep.h) for n_fft = 256, 512, 1024; own TU: parallel builds
#include "kernel_regtile_widep.h"
#include <atomic>
namespace sfft { SFFT_DEFINE_REGTILE_WIDEP_LAUNCHER(16, 16) SFFT_DEFINE_REGTILE_WIDEP_LAUNCHER(32, 16) SFFT_DEFINE_REGTILE_WIDEP_LAUNCHER(32, 32) }
