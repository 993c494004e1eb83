// Micro-benchmark + self-check of conv_gemm on the shapes of the B = 32, T = 768 workload.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I e2e_tts_amd/csrc tools/conv_bench.hip e2e_tts_amd/csrc/conv_gemm.hip e2e_tts_amd/csrc/small_kernels.hip -o tools/bin/conv_bench
// Usage: [CONV_BENCH_B=1] [CONV_BENCH_KERNEL=rows|ksplit] conv_bench [reps] [name-filter | -] [f32 | f32f | x3 | x3f]
//   (f32f / x3f: weights in MFMA-fragment order; CONV_BENCH_KERNEL: check and time conv_ksplit.hip's kernels -- they need f32f / x3f --
//   instead of conv_gemm where they support the shape)
// Build adds e2e_tts_amd/csrc/conv_ksplit.hip to the line below.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.h"

using namespace e2etts;
#ifdef E2ETTS_DIAG
namespace e2etts { void conv_gemm_read_diag(unsigned long long* out); }
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void naive_conv(ConvParams p) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long total = (long long)p.B * p.T * p.Cout;
  if (idx >= total) return;
  int n = idx % p.Cout;
  long long bt = idx / p.Cout;
  int t = bt % p.T, b = bt / p.T;
  double acc = 0;
  for (int j = 0; j < p.KW; ++j) {
    int tt = t - p.pad + j * p.dil;
    if (tt < 0 || tt >= p.T) continue;
    for (int c = 0; c < p.Cin; ++c) {
      float v = p.in[(long long)b * p.in_bs + (long long)tt * p.in_ld + c];
      if (p.in_slope != 1.f && v < 0) v *= p.in_slope;
      acc += (double)v * p.w[(long long)n * p.KW * p.Cin + j * p.Cin + c];
    }
  }
  float v = (float)acc + (p.bias ? p.bias[n] : 0.f);
  if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
  else if (p.act == ACT_TANH) v = tanhf(v);
  else if (p.act == ACT_LRELU) v = v >= 0 ? v : v * p.act_slope;
  if (p.res) v += p.res[(long long)b * p.res_bs + (long long)t * p.res_ld + n];
  if (p.lens && t >= p.lens[b]) v = 0;
  float* o = p.out + (long long)b * p.out_bs + (long long)t * p.out_ld + n;
  if (p.accumulate) v += *o;
  if (p.out_div != 1.f) v /= p.out_div;
  *o = v;
}

static unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
  return (unsigned short)u;
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
// fp32 [Cout][KW*Cin] -> x3 layout [Cout][KW][nchunk][32 bf16 hi | 32 bf16 lo] (as 32-bit words)
static std::vector<float> pack_x3(const std::vector<float>& w, int Cout, int KW, int Cin) {
  int nchunk = (Cin + 31) / 32;
  std::vector<unsigned short> out((size_t)Cout * KW * nchunk * 64, 0);
  for (int n = 0; n < Cout; ++n) for (int j = 0; j < KW; ++j) for (int c = 0; c < Cin; ++c) {
    float v = w[((size_t)n * KW + j) * Cin + c];
    unsigned short hi = f2bf(v), lo = f2bf(v - bf2f(hi));
    size_t base = (((size_t)n * KW + j) * nchunk + c / 32) * 64;
    out[base + (c % 32)] = hi; out[base + 32 + (c % 32)] = lo;
  }
  std::vector<float> r(out.size() / 2); memcpy(r.data(), out.data(), out.size() * 2); return r;
}

// fp32 [Cout][KW*Cin] -> fragment order [ntile32][KW][nchunk][ks(2)][hi|lo][lane(64)][8 bf16] (as 32-bit words)
static std::vector<float> pack_x3_frag(const std::vector<float>& w, int Cout, int KW, int Cin) {
  int nchunk = (Cin + 31) / 32, nt = (Cout + 31) / 32;
  std::vector<unsigned short> out((size_t)nt * KW * nchunk * 2 * 2 * 64 * 8, 0);
  for (int t = 0; t < nt; ++t) for (int j = 0; j < KW; ++j) for (int c = 0; c < nchunk; ++c) for (int ks = 0; ks < 2; ++ks)
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
      int n = t * 32 + (lane & 31), ch = c * 32 + ks * 16 + (lane >> 5) * 8 + e;
      float v = (n < Cout && ch < Cin) ? w[((size_t)n * KW + j) * Cin + ch] : 0.f;
      unsigned short hi = f2bf(v), lo = f2bf(v - bf2f(hi));
      size_t base = ((((size_t)(t * KW + j) * nchunk + c) * 2 + ks) * 2) * 512 + (size_t)lane * 8 + e;
      out[base] = hi; out[base + 512] = lo;
    }
  std::vector<float> r(out.size() / 2); memcpy(r.data(), out.data(), out.size() * 2); return r;
}

static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
}

struct Shape { const char* name; int B, T, Cin, Cout, KW, dil; bool res, acc; float slope; };

int main(int argc, char** argv) {
  int reps = argc > 1 ? atoi(argv[1]) : 10;
  const char* filter = argc > 2 && strcmp(argv[2], "-") ? argv[2] : nullptr;
  const int x3 = argc > 3 && (!strcmp(argv[3], "x3") || !strcmp(argv[3], "x3f"));
  const int frag = argc > 3 && !strcmp(argv[3], "x3f");
  const int frag32 = argc > 3 && !strcmp(argv[3], "f32f");
  hipStream_t s;
  CK(hipStreamCreate(&s));
  // ---- correctness on small awkward shapes
  if (!filter || x3 || frag32) {
    Shape checks[] = {{"chk1", 2, 300, 80, 80, 5, 1, true, false, 0.1f}, {"chk2", 3, 777, 32, 32, 11, 5, true, true, 0.1f},
                      {"chk3", 1, 129, 128, 200, 3, 3, false, false, 1.0f}, {"chk4", 2, 64, 384, 1152, 1, 1, false, false, 1.0f},
                      {"chk5", 2, 1000, 64, 64, 7, 3, true, false, 0.1f}, {"chk6", 1, 50, 12, 20, 9, 1, false, false, 1.0f},
                      {"chk7 odd", 2, 333, 32, 18, 7, 1, false, false, 1.0f}, {"chk8 odd", 1, 100, 64, 131, 3, 2, true, false, 0.1f},  // Cout % 4 != 0: scalar epilogue
                      // enough 128 x 128 tiles for the persistent three-wave kernel (f32f): several tiles and chunks per workgroup, ragged last tile
                      {"chk9 big", 8, 8200, 128, 128, 11, 5, true, false, 0.1f}, {"chk10 big", 8, 4100, 256, 256, 3, 1, true, true, 0.1f}};
    for (auto& c : checks) {
      size_t nin = (size_t)c.B * c.T * c.Cin, nout = (size_t)c.B * c.T * c.Cout, nw = (size_t)c.Cout * c.KW * c.Cin;
      std::vector<float> hin(nin), hw(nw), hb(c.Cout), hres(nout), hout0(nout);
      fill(hin, 1); fill(hw, 2); fill(hb, 3); fill(hres, 4); fill(hout0, 5);
      for (auto& x : hw) x *= 0.05f;
      float *din, *dw, *db, *dres, *dout, *dref; int* dlens;
      CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, c.Cout * 4)); CK(hipMalloc(&dres, nout * 4));
      CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dref, nout * 4)); CK(hipMalloc(&dlens, c.B * 4));
      std::vector<int> lens(c.B); for (int b = 0; b < c.B; ++b) lens[b] = c.T - 7 * b;
      float* dwx = nullptr;
      float* dwf = nullptr;
      if (x3) { auto px = pack_x3(hw, c.Cout, c.KW, c.Cin); CK(hipMalloc(&dwx, px.size() * 4)); CK(hipMemcpy(dwx, px.data(), px.size() * 4, hipMemcpyHostToDevice)); }
      if (frag) { auto pf = pack_x3_frag(hw, c.Cout, c.KW, c.Cin); CK(hipMalloc(&dwf, pf.size() * 4)); CK(hipMemcpy(dwf, pf.data(), pf.size() * 4, hipMemcpyHostToDevice)); }
      CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(db, hb.data(), c.Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dres, hres.data(), nout * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dout, hout0.data(), nout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dref, hout0.data(), nout * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dlens, lens.data(), c.B * 4, hipMemcpyHostToDevice));
      ConvParams p; p.in = din; p.w = dw; p.bias = db; p.res = c.res ? dres : nullptr; p.out = dout; p.lens = dlens;
      p.B = c.B; p.T = c.T; p.Cin = c.Cin; p.Cout = c.Cout; p.KW = c.KW; p.dil = c.dil; p.pad = c.dil * (c.KW - 1) / 2;
      p.in_ld = c.Cin; p.out_ld = c.Cout; p.res_ld = c.Cout; p.in_bs = (long long)c.T * c.Cin; p.out_bs = (long long)c.T * c.Cout; p.res_bs = p.out_bs;
      p.in_slope = c.slope; p.act = ACT_LRELU; p.act_slope = 0.1f; p.accumulate = c.acc; p.out_div = c.acc ? 3.f : 1.f;
      ConvParams q = p; q.out = dref;
      if (x3) { p.w = dwx; p.x3 = 1; }
      if (frag) p.wfrag = dwf;
      if (frag32) {
        CK(hipMalloc(&dwf, x3_frag_bytes(c.Cout, c.KW, c.Cin)));
        const char* fm = launch_f32_to_frag(dw, dwf, c.Cout, c.KW, c.Cin, s);
        if (fm) { printf("%s\n", fm); return 1; }
        p.wfrag = dwf;
      }
      const char* kern = getenv("CONV_BENCH_KERNEL");
      const bool use_rows = kern && !strcmp(kern, "rows") && conv_rows_supported(p);
      const bool use_ks = kern && !strcmp(kern, "ksplit") && conv_ksplit_supported(p);
      const char* m = use_rows ? launch_conv_rows(p, s) : (use_ks ? launch_conv_ksplit(p, s) : launch_conv_gemm(p, s));
      if (m) { printf("%s: %s\n", c.name, m); return 1; }
      if (kern) printf("  [%s] ", use_rows ? "conv_rows" : (use_ks ? "conv_ksplit" : "conv_gemm (shape not supported by the asked kernel)"));
      hipLaunchKernelGGL(naive_conv, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, s, q);
      CK(hipStreamSynchronize(s));
      std::vector<float> a(nout), r(nout);
      CK(hipMemcpy(a.data(), dout, nout * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), dref, nout * 4, hipMemcpyDeviceToHost));
      double maxd = 0; for (size_t i = 0; i < nout; ++i) maxd = fmax(maxd, fabs((double)a[i] - r[i]));
      printf("check %-5s Cin=%d Cout=%d KW=%d dil=%d: max |diff| = %.3g %s\n", c.name, c.Cin, c.Cout, c.KW, c.dil, maxd, maxd < (x3 ? 5e-4 : 2e-4) ? "ok" : "FAIL");
      if (!(maxd < (x3 ? 5e-4 : 2e-4))) return 1;
      hipFree(din); hipFree(dw); hipFree(db); hipFree(dres); hipFree(dout); hipFree(dref); hipFree(dlens);
    }
  }
  // ---- timing
  const int B = getenv("CONV_BENCH_B") ? atoi(getenv("CONV_BENCH_B")) : 32;  // 1: the latency path's launches
  Shape shapes[] = {
      {"s1 k3d1", B, 6144, 256, 256, 3, 1, false, false, 0.1f}, {"s1 k11d5", B, 6144, 256, 256, 11, 5, false, false, 0.1f},
      {"s1 k11 c2+res", B, 6144, 256, 256, 11, 1, true, false, 1.0f},
      {"s2 k3d1", B, 49152, 128, 128, 3, 1, false, false, 0.1f}, {"s2 k3 c2+res", B, 49152, 128, 128, 3, 1, true, false, 1.0f},
      {"s2 k7d3", B, 49152, 128, 128, 7, 3, false, false, 0.1f}, {"s2 k11d5", B, 49152, 128, 128, 11, 5, false, false, 0.1f},
      {"s2 k11 c2+res+acc", B, 49152, 128, 128, 11, 1, true, true, 1.0f},
      {"s3 k3d1", B, 98304, 64, 64, 3, 1, false, false, 0.1f}, {"s3 k7d3", B, 98304, 64, 64, 7, 3, false, false, 0.1f},
      {"s3 k11 c2+res", B, 98304, 64, 64, 11, 1, true, false, 1.0f},
      {"s4 k3d1", B, 196608, 32, 32, 3, 1, false, false, 0.1f}, {"s4 k7 c2+res", B, 196608, 32, 32, 7, 1, true, false, 1.0f},
      {"s4 k11d5", B, 196608, 32, 32, 11, 5, false, false, 0.1f},
      {"ffn k9", B, 768, 384, 1024, 9, 1, false, false, 1.0f}, {"ffn w2+res", B, 768, 1024, 384, 1, 1, true, false, 1.0f},
      {"qkv", B, 768, 384, 1152, 1, 1, false, false, 1.0f}, {"postnet", B, 768, 512, 512, 5, 1, false, false, 1.0f},
      {"up0", B, 768, 512, 2048, 3, 1, false, false, 0.1f}, {"up1", B, 6144, 256, 1024, 3, 1, false, false, 0.1f},
      {"up2", B, 49152, 128, 128, 3, 1, false, false, 0.1f}, {"up3", B, 98304, 64, 64, 3, 1, false, false, 0.1f},
  };
  size_t maxel = (size_t)B * 196608 * 32;
  maxel = std::max(maxel, (size_t)B * 768 * 2048);
  float *din, *dout, *dres, *dw, *db;
  CK(hipMalloc(&din, maxel * 4)); CK(hipMalloc(&dout, maxel * 4)); CK(hipMalloc(&dres, maxel * 4));
  CK(hipMalloc(&dw, (size_t)2048 * 3 * 512 * 4 + (1 << 20))); CK(hipMalloc(&db, 4096 * 4));
  {
    std::vector<float> h(maxel); fill(h, 7);
    CK(hipMemcpy(din, h.data(), maxel * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dres, h.data(), maxel * 4, hipMemcpyHostToDevice));
    std::vector<float> w((size_t)2048 * 3 * 512 + (1 << 18)); fill(w, 8); for (auto& x : w) x *= 0.03f;
    CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, w.data(), 4096 * 4, hipMemcpyHostToDevice));
  }
  float* dwx3 = nullptr;
  {  // any finite bf16 pattern will do for timing: pack a random [2048][3*512] matrix once (covers every shape's footprint)
    std::vector<float> w((size_t)2048 * 3 * 512); fill(w, 9); for (auto& x : w) x *= 0.03f;
    auto px = pack_x3(w, 2048, 3, 512);
    CK(hipMalloc(&dwx3, px.size() * 4 + (4 << 20)));  // the largest fragment image (ffn k9: 14.2 MB) must fit: timing reads it as wfrag
    CK(hipMemset(dwx3, 0, px.size() * 4 + (4 << 20)));
    CK(hipMemcpy(dwx3, px.data(), px.size() * 4, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double tot_ms = 0, tot_fl = 0;
  for (auto& c : shapes) {
    if (filter && !strstr(c.name, filter)) continue;
    ConvParams p; p.in = din; p.w = dw; p.bias = db; p.res = c.res ? dres : nullptr; p.out = dout;
    p.B = c.B; p.T = c.T; p.Cin = c.Cin; p.Cout = c.Cout; p.KW = c.KW; p.dil = c.dil; p.pad = c.dil * (c.KW - 1) / 2;
    p.in_ld = c.Cin; p.out_ld = c.Cout; p.res_ld = c.Cout; p.in_bs = (long long)c.T * c.Cin; p.out_bs = (long long)c.T * c.Cout; p.res_bs = p.out_bs;
    p.in_slope = c.slope; p.act = c.res ? ACT_NONE : ACT_LRELU; p.act_slope = 0.1f; p.accumulate = c.acc; p.out_div = 1.f;
    p.x3 = x3; if (x3) p.w = dwx3;
    if (frag || frag32) p.wfrag = dwx3;  // timing only: any finite pattern, same footprint
    const char* kern = getenv("CONV_BENCH_KERNEL");
    const bool use_rows = kern && !strcmp(kern, "rows") && conv_rows_supported(p);
    const bool use_ks = kern && !strcmp(kern, "ksplit") && conv_ksplit_supported(p);
    auto launch = [&]() { return use_rows ? launch_conv_rows(p, s) : (use_ks ? launch_conv_ksplit(p, s) : launch_conv_gemm(p, s)); };
    for (int i = 0; i < 2; ++i) { const char* m = launch(); if (m) { printf("%s: %s\n", c.name, m); return 1; } }
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double fl = conv_gemm_flops(p);
#ifdef E2ETTS_DIAG
    { unsigned long long d[8]; conv_gemm_read_diag(d);
      const char* nm[7] = {"issue loads", "barrier", "frags+mfma", "store_b", "item barrier", "epilogue", "store_a"};
      printf("   diag (cycles per iteration, %llu iterations):", d[7]);
      for (int i = 0; i < 7; ++i) printf(" %s=%.0f", nm[i], (double)d[i] / (double)d[7]);
      printf("\n"); }
#endif
    printf("%-20s Cin=%4d Cout=%4d KW=%2d dil=%d T=%6d : %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s\n", c.name, c.Cin, c.Cout, c.KW, c.dil, c.T, ms,
           fl / ms / 1e9, conv_gemm_bytes(p) / ms / 1e6);
    tot_ms += ms; tot_fl += fl;
  }
  printf("TOTAL %.3f ms, %.2f TFLOP/s (unweighted list)\n", tot_ms, tot_fl / tot_ms / 1e9);
  return 0;
}
