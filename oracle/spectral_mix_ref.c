/* spectral_mix_ref.c — plain-C CPU oracle for the SPECTRE spectral-mix hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by or called from the product library
 * (fft_amd/csrc); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Restates /root/reference/spectre.py:506 (rfft along the sequence axis, zero-pad/truncate to n_fft),
 * :542-545 (gate[b, c / d_g, k] broadcast over the d_g channels of a group, complex multiply),
 * :548-549 (optional batch-invariant memory_fft add), :551-553 (irfft with Hermitian symmetry implied:
 * Im of bin 0 and, for even n_fft, of bin n_fft/2 is ignored; scale 1/n_fft; keep the first
 * min(N, n_fft) rows).  The reference delegates the transforms to torch.fft (MKL, not in the reference
 * tree); here they are the textbook O(N^2) DFT sums in double precision with an exact-index
 * cos/sin table (angle = 2*pi*((k*n) mod n_fft)/n_fft), so the result does not depend on any FFT library.
 *
 * Layouts (all contiguous unless a stride is given):
 *   v    [B][N_in][D]  float   (row stride v_sn, batch stride v_sb, in elements)
 *   gate [B][G][F][2]  float   (re, im), F = n_fft/2 + 1
 *   mem  [F][D][2]     float   or NULL
 *   out  [B][N_out][D] float   N_out = min(N_in, n_fft)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int spectral_mix_ref(const float *v, const float *gate, const float *mem, float *out,
                     int64_t B, int64_t N_in, int64_t n_fft, int64_t D, int64_t G,
                     int64_t v_sb, int64_t v_sn, int64_t out_sb, int64_t out_sn) {
  if (B < 0 || N_in < 1 || n_fft < 1 || D < 1 || G < 1 || D % G) return 1;
  const int64_t F = n_fft / 2 + 1, d_g = D / G;
  const int64_t N_use = N_in < n_fft ? N_in : n_fft; /* rows that enter the transform == rows kept */
  double *ct = (double *)malloc(sizeof(double) * (size_t)n_fft);
  double *st = (double *)malloc(sizeof(double) * (size_t)n_fft);
  double *xr = (double *)malloc(sizeof(double) * (size_t)F);
  double *xi = (double *)malloc(sizeof(double) * (size_t)F);
  if (!ct || !st || !xr || !xi) { free(ct); free(st); free(xr); free(xi); return 2; }
  for (int64_t m = 0; m < n_fft; ++m) {
    ct[m] = cos(2.0 * M_PI * (double)m / (double)n_fft);
    st[m] = sin(2.0 * M_PI * (double)m / (double)n_fft);
  }
  for (int64_t b = 0; b < B; ++b) {
    for (int64_t c = 0; c < D; ++c) {
      const int64_t g = c / d_g;
      /* rfft, spectre.py:506 */
      for (int64_t k = 0; k < F; ++k) {
        double sr = 0.0, si = 0.0;
        for (int64_t n = 0; n < N_use; ++n) {
          const double x = (double)v[b * v_sb + n * v_sn + c];
          const int64_t m = (k * n) % n_fft;
          sr += x * ct[m];
          si -= x * st[m];
        }
        /* gate multiply, :542-545 */
        const double gr = (double)gate[((b * G + g) * F + k) * 2 + 0];
        const double gi = (double)gate[((b * G + g) * F + k) * 2 + 1];
        double yr = gr * sr - gi * si, yi = gr * si + gi * sr;
        /* memory add, :548-549 */
        if (mem) { yr += (double)mem[(k * D + c) * 2 + 0]; yi += (double)mem[(k * D + c) * 2 + 1]; }
        xr[k] = yr; xi[k] = yi;
      }
      /* irfft, :551 — Hermitian extension implied, Im(DC)/Im(Nyquist) ignored */
      for (int64_t n = 0; n < N_use; ++n) {
        double acc = xr[0];
        const int64_t last = (n_fft % 2 == 0) ? F - 1 : F;  /* bins 1 .. last-1 are doubled */
        for (int64_t k = 1; k < last; ++k) {
          const int64_t m = (k * n) % n_fft;
          acc += 2.0 * (xr[k] * ct[m] - xi[k] * st[m]);
        }
        if (n_fft % 2 == 0 && F > 1) acc += xr[F - 1] * ((n & 1) ? -1.0 : 1.0);
        out[b * out_sb + n * out_sn + c] = (float)(acc / (double)n_fft); /* :553 keeps rows < N */
      }
    }
  }
  free(ct); free(st); free(xr); free(xi);
  return 0;
}
