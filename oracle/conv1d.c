/* Plain-C restatement of torch.nn.Conv1d (stride 1, zero padding, dilation) for the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/ref_numpy.py): used by the numpy oracle as a faster conv1d and by bench.py's
 * cpu_baseline leg.  Semantics restated: out[b][co][t] = bias[co] + sum_ci sum_k w[co][ci][k] * xpad[b][ci][t + k*dil]
 * with xpad = x zero-padded by `pad` on both sides (reference call sites: V/layers.py:14-29, V/generator.py:18,33,
 * U/blocks/transformer.py:271-284, U/layers.py:393-400,475-482,518-553).
 *
 * Build (done by __graft_entry__.build()):  gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC oracle/conv1d.c -o oracle/lib/libref_conv1d.so
 */
#include <stdlib.h>
#include <string.h>

#define TB 256 /* output positions per block */
#define CB 8   /* output channels per block  */

int ref_conv1d_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int T, int Cout, int K,
                   int pad, int dil) {
  const long Tp = (long)T + 2L * pad;
  const long Tout = Tp - (long)dil * (K - 1);
  if (Tout <= 0) return -1;
  const long Tpa = Tp + TB; /* slack so a block may read past the row end */
  float* xp = (float*)calloc((size_t)B * Cin * Tpa, sizeof(float));
  if (!xp) return -2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int ci = 0; ci < Cin; ++ci)
      memcpy(xp + ((long)b * Cin + ci) * Tpa + pad, x + ((long)b * Cin + ci) * T, (size_t)T * sizeof(float));
  const long ntb = (Tout + TB - 1) / TB;
  const int ncb = (Cout + CB - 1) / CB;
#pragma omp parallel for collapse(3) schedule(dynamic, 4)
  for (int b = 0; b < B; ++b)
    for (int cb = 0; cb < ncb; ++cb)
      for (long tb = 0; tb < ntb; ++tb) {
        float acc[CB][TB];
        const int co0 = cb * CB;
        const int nco = Cout - co0 < CB ? Cout - co0 : CB;
        const long t0 = tb * TB;
        for (int c = 0; c < CB; ++c) {
          const float bv = (bias && c < nco) ? bias[co0 + c] : 0.0f;
          for (int t = 0; t < TB; ++t) acc[c][t] = bv;
        }
        for (int ci = 0; ci < Cin; ++ci) {
          const float* xr0 = xp + ((long)b * Cin + ci) * Tpa + t0;
          for (int k = 0; k < K; ++k) {
            const float* xr = xr0 + (long)k * dil;
            for (int c = 0; c < nco; ++c) {
              const float wv = w[((long)(co0 + c) * Cin + ci) * K + k];
              float* a = acc[c];
#pragma omp simd
              for (int t = 0; t < TB; ++t) a[t] += wv * xr[t];
            }
          }
        }
        const long n = Tout - t0 < TB ? Tout - t0 : TB;
        for (int c = 0; c < nco; ++c) memcpy(out + ((long)b * Cout + co0 + c) * Tout + t0, acc[c], (size_t)n * sizeof(float));
      }
  free(xp);
  return 0;
}
