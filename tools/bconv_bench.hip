// Self-check + micro-benchmark of conv_bf16.hip on the shapes of BASELINE config 5 (48 kHz HiFi-GAN, 8 x 8 x 4 x 2, width 512, B = 1):
// every shape is run through conv_gemm in mode 2 (plain bf16: the path the engine used through round 3) and through conv_bf16, the outputs
// are compared BIT FOR BIT (the two must be interchangeable per launch), and both are timed.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I e2e_tts_amd/csrc tools/bconv_bench.hip e2e_tts_amd/csrc/conv_bf16.hip \
//          e2e_tts_amd/csrc/conv_gemm.hip e2e_tts_amd/csrc/resblock_pair.hip e2e_tts_amd/csrc/small_kernels.hip -o tools/bin/bconv_bench
// Usage: bconv_bench [reps] [name-filter | -] [frames: 542 = one streaming window (default), 5632 = the whole utterance]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kernels.h"

using namespace e2etts;
#ifdef E2ETTS_BC_DIAG
namespace e2etts { void conv_bf16_read_diag(unsigned long long* out); }
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
  return (unsigned short)u;
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
// fp32 [Cout][KW*Cin] -> x3 layout [Cout][KW][nchunk][32 bf16 hi | 32 bf16 lo] (as 32-bit words): packer.pack_x3
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
static void fill(std::vector<float>& v, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
}

struct Shape { const char* name; int rows_per_frame, Cin, Cout, KW, dil; bool c2; int poly; };  // c2: conv2 of a pair (no input slope, residual); poly: upsampler rate

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 20;
  const char* filter = argc > 2 && strcmp(argv[2], "-") ? argv[2] : nullptr;
  const int frames = argc > 3 ? atoi(argv[3]) : 542;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  Shape shapes[] = {
      {"pre 80>512 k7", 1, 80, 512, 7, 1, false, 0},
      {"up1 512>8x256", 1, 512, 2048, 3, 1, false, 8},
      {"s1 c1 k3d1", 8, 256, 256, 3, 1, false, 0}, {"s1 c1 k3d5", 8, 256, 256, 3, 5, false, 0}, {"s1 c2 k3", 8, 256, 256, 3, 1, true, 0},
      {"s1 c1 k7d3", 8, 256, 256, 7, 3, false, 0}, {"s1 c2 k7", 8, 256, 256, 7, 1, true, 0},
      {"s1 c1 k11d5", 8, 256, 256, 11, 5, false, 0}, {"s1 c2 k11", 8, 256, 256, 11, 1, true, 0},
      {"up2 256>8x128", 8, 256, 1024, 3, 1, false, 8},
      {"s2 c1 k3d3", 64, 128, 128, 3, 3, false, 0}, {"s2 c1 k11d5", 64, 128, 128, 11, 5, false, 0}, {"s2 c2 k11", 64, 128, 128, 11, 1, true, 0},
      {"s2 c1 k7d1", 64, 128, 128, 7, 1, false, 0},
      {"up3 128>4x64", 64, 128, 256, 3, 1, false, 4},
      {"s3 c1 k7d3", 256, 64, 64, 7, 3, false, 0}, {"s3 c2 k11", 256, 64, 64, 11, 1, true, 0},
      {"up4 64>2x32", 256, 64, 64, 3, 1, false, 2},
      {"s4 c1 k11d5", 512, 32, 32, 11, 5, false, 0}, {"s4 c2 k7", 512, 32, 32, 7, 1, true, 0},
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int bad = 0;
  for (auto& c : shapes) {
    if (filter && !strstr(c.name, filter)) continue;
    const int T = frames * c.rows_per_frame;
    const size_t nin = (size_t)T * c.Cin, nout = (size_t)T * c.Cout, nw = (size_t)c.Cout * c.KW * c.Cin;
    std::vector<float> hin(nin), hw(nw), hb(c.Cout), hres(nout);
    fill(hin, 1); fill(hw, 2); fill(hb, 3); fill(hres, 4);
    for (auto& x : hw) x *= 0.05f;
    const int split = c.poly ? c.Cout / 2 : 0;
    if (c.poly)   // the polyphase image's structural zeros (packer.polyphase_upsampler): columns < split: tap 2, columns >= split: tap 0
      for (int n = 0; n < c.Cout; ++n) for (int ch = 0; ch < c.Cin; ++ch) hw[((size_t)n * 3 + (n < split ? 2 : 0)) * c.Cin + ch] = 0.f;
    float *din, *db, *dres, *dold, *dnew, *dwx, *dwf;
    void* dimg;
    CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&db, c.Cout * 4)); CK(hipMalloc(&dres, nout * 4));
    CK(hipMalloc(&dold, nout * 4)); CK(hipMalloc(&dnew, nout * 4));
    auto px = pack_x3(hw, c.Cout, c.KW, c.Cin);
    CK(hipMalloc(&dwx, px.size() * 4)); CK(hipMemcpy(dwx, px.data(), px.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dwf, x3_frag_bytes(c.Cout, c.KW, c.Cin)));
    CK(hipMalloc(&dimg, bf16_image_bytes(c.Cout, c.KW, c.Cin, split)));
    CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), c.Cout * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dres, hres.data(), nout * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dold, 0, nout * 4)); CK(hipMemset(dnew, 0xff, nout * 4));
    const char* m = launch_x3_to_frag(dwx, dwf, c.Cout, c.KW, c.Cin, s);
    if (!m) m = launch_bf16_image(dwx, dimg, c.Cout, c.KW, c.Cin, split, s);
    if (m) { printf("%s: %s\n", c.name, m); return 1; }
    ConvParams p; p.in = din; p.w = dwx; p.x3 = 2; p.wfrag = dwf; p.bias = db; p.res = c.c2 ? dres : nullptr; p.out = dold;
    p.B = 1; p.T = T; p.Cin = c.Cin; p.Cout = c.Cout; p.KW = c.KW; p.dil = c.dil; p.pad = c.dil * (c.KW - 1) / 2;
    p.in_ld = c.Cin; p.out_ld = c.Cout; p.res_ld = c.Cout; p.in_bs = (long long)T * c.Cin; p.out_bs = (long long)T * c.Cout; p.res_bs = p.out_bs;
    p.in_slope = c.c2 ? 1.0f : 0.1f; p.act = (c.c2 || c.poly || c.Cin == 80) ? ACT_NONE : ACT_LRELU; p.act_slope = 0.1f;
    p.zero_tap_split = split;
    BConvParams q; q.in = din; q.in_slope = p.in_slope; q.wimg = dimg; q.KWe = c.poly ? 2 : c.KW; q.tap_split = split; q.bias = db;
    q.act_slope = p.act == ACT_LRELU ? 0.1f : 1.0f; q.res = p.res; q.out = dnew;
    q.B = 1; q.T = T; q.Cin = c.Cin; q.Cout = c.Cout; q.KW = c.KW; q.dil = c.dil; q.pad = p.pad;
    if (!conv_bf16_supported(q)) { printf("%-16s not supported by conv_bf16\n", c.name); continue; }
    m = launch_conv_gemm(p, s);
    if (!m) m = launch_conv_bf16(q, s);
    if (m) { printf("%s: %s\n", c.name, m); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> a(nout), r(nout);
    CK(hipMemcpy(a.data(), dnew, nout * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), dold, nout * 4, hipMemcpyDeviceToHost));
    size_t ndiff = 0; double maxd = 0, maxv = 0;
    for (size_t i = 0; i < nout; ++i) {
      if (memcmp(&a[i], &r[i], 4)) { ++ndiff; maxd = fmax(maxd, fabs((double)a[i] - r[i])); }
      maxv = fmax(maxv, fabs((double)r[i]));
    }
    // bf16 hand-over: conv_bf16 writing its result as a bf16 image, then a second conv_bf16 reading it, against fp32 in between
    // (same values: the image is what the reader's staging would have formed)
    size_t ndiff_b = 0;
    if (!c.c2 && !c.poly && c.Cin == c.Cout) {
      void* dmid; float *d2a, *d2b;
      CK(hipMalloc(&dmid, nout * 2)); CK(hipMalloc(&d2a, nout * 4)); CK(hipMalloc(&d2b, nout * 4));
      BConvParams q1 = q; q1.out = nullptr; q1.out_b = dmid; q1.outb_slope = 1.0f;
      BConvParams q2 = q; q2.in = dnew; q2.in_slope = 1.0f; q2.res = dres; q2.act_slope = 1.0f; q2.out = d2a;     // fp32 hand-over (dnew holds conv1's output)
      BConvParams q3 = q2; q3.in = dmid; q3.in_bf16 = 1; q3.out = d2b;
      m = launch_conv_bf16(q1, s);
      if (!m) m = launch_conv_bf16(q2, s);
      if (!m) m = launch_conv_bf16(q3, s);
      if (m) { printf("%s: %s\n", c.name, m); return 1; }
      CK(hipStreamSynchronize(s));
      std::vector<float> x(nout), y(nout);
      CK(hipMemcpy(x.data(), d2a, nout * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), d2b, nout * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < nout; ++i) ndiff_b += memcmp(&x[i], &y[i], 4) ? 1 : 0;
      (void)hipFree(dmid); (void)hipFree(d2a); (void)hipFree(d2b);
    }
    float ms_old, ms_new;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_conv_gemm(p, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_old, e0, e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_conv_bf16(q, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
#ifdef E2ETTS_BC_DIAG
    { unsigned long long d[8]; conv_bf16_read_diag(d);
      const char* nm[6] = {"ring requests", "staging", "barrier", "K loop", "epilogue", "whole"};
      printf("   diag (cycles per workgroup, wave 0; %llu workgroups):", d[7]);
      for (int i = 0; i < 6; ++i) printf(" %s=%.0f", nm[i], (double)d[i] / (double)(d[7] ? d[7] : 1));
      printf("\n"); }
#endif
    const double fl = 2.0 * T * c.Cout * (double)c.Cin * (c.poly ? 2 : c.KW);
    printf("%-16s T=%7d %s | %-18s %7.1f us %7.1f TFLOP/s | %-18s %7.1f us %7.1f TFLOP/s | %zu differing of %zu (max |d| %.3g, max |v| %.3g)%s%s\n", c.name, T,
           conv_gemm_class(p), "conv_gemm mode 2", ms_old / reps * 1e3, fl / (ms_old / reps) / 1e9, conv_bf16_class(q), ms_new / reps * 1e3,
           fl / (ms_new / reps) / 1e9, ndiff, nout, maxd, maxv, ndiff ? "  MISMATCH" : "", ndiff_b ? "  BF16-HANDOVER MISMATCH" : "");
    bad += (ndiff || ndiff_b) ? 1 : 0;
    (void)hipFree(din); (void)hipFree(db); (void)hipFree(dres); (void)hipFree(dold); (void)hipFree(dnew); (void)hipFree(dwx); (void)hipFree(dwf); (void)hipFree(dimg);
  }
  // ---- fused pairs: resblock_pair (mode 2) against pair_bf16, bit for bit, and both timed
  struct PShape { const char* name; int rows_per_frame, C, KW, dil; bool acc; };
  PShape pshapes[] = {{"pair s1 k3d1", 8, 256, 3, 1, false}, {"pair s1 k7d3+a", 8, 256, 7, 3, true}, {"pair s1 k11d5", 8, 256, 11, 5, false},
                      {"pair s2 k3d1", 64, 128, 3, 1, false}, {"pair s2 k7d3", 64, 128, 7, 3, false}, {"pair s2 k11d5+a", 64, 128, 11, 5, true},
                      {"pair s3 k7d1", 256, 64, 7, 1, false}, {"pair s3 k11d5+a", 256, 64, 11, 5, true}, {"pair s3 k3d3", 256, 64, 3, 3, false},
                      {"pair s4 k7d3", 512, 32, 7, 3, false}, {"pair s4 k11d1", 512, 32, 11, 1, false}, {"pair s4 k11d5+a", 512, 32, 11, 5, true}};
  for (auto& c : pshapes) {
    if (filter && !strstr(c.name, filter)) continue;
    const int T = frames * c.rows_per_frame;
    const size_t nx = (size_t)T * c.C, nw = (size_t)c.C * c.KW * c.C;
    std::vector<float> hx(nx), hw1(nw), hw2(nw), hb1(c.C), hb2(c.C), hs(nx);
    fill(hx, 11); fill(hw1, 12); fill(hw2, 13); fill(hb1, 14); fill(hb2, 15); fill(hs, 16);
    for (auto& x : hw1) x *= 0.05f;
    for (auto& x : hw2) x *= 0.05f;
    float *dx, *db1, *db2, *dold, *dnew, *dwx1, *dwx2, *dwf;
    void *dimg1, *dimg2;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&db1, c.C * 4)); CK(hipMalloc(&db2, c.C * 4)); CK(hipMalloc(&dold, nx * 4)); CK(hipMalloc(&dnew, nx * 4));
    auto p1 = pack_x3(hw1, c.C, c.KW, c.C), p2 = pack_x3(hw2, c.C, c.KW, c.C);
    CK(hipMalloc(&dwx1, p1.size() * 4)); CK(hipMalloc(&dwx2, p2.size() * 4));
    CK(hipMemcpy(dwx1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwx2, p2.data(), p2.size() * 4, hipMemcpyHostToDevice));
    const size_t one = x3_frag_bytes(c.C, c.KW, c.C);
    CK(hipMalloc(&dwf, 2 * one)); CK(hipMalloc(&dimg1, bf16_image_bytes(c.C, c.KW, c.C, 0))); CK(hipMalloc(&dimg2, bf16_image_bytes(c.C, c.KW, c.C, 0)));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db1, hb1.data(), c.C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db2, hb2.data(), c.C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dold, hs.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dnew, hs.data(), nx * 4, hipMemcpyHostToDevice));
    const char* m = launch_x3_to_frag(dwx1, dwf, c.C, c.KW, c.C, s);
    if (!m) m = launch_x3_to_frag(dwx2, dwf + one / 4, c.C, c.KW, c.C, s);
    if (!m) m = launch_bf16_image(dwx1, dimg1, c.C, c.KW, c.C, 0, s);
    if (!m) m = launch_bf16_image(dwx2, dimg2, c.C, c.KW, c.C, 0, s);
    if (m) { printf("%s: %s\n", c.name, m); return 1; }
    PairParams q; q.x = dx; q.wfrag = dwf; q.b1 = db1; q.b2 = db2; q.out = dold; q.B = 1; q.T = T; q.C = c.C; q.KW = c.KW; q.dil = c.dil;
    q.x_bs = q.out_bs = (long long)T * c.C; q.slope = 0.1f; q.mode = 2; q.accumulate = c.acc; q.out_div = c.acc ? 3.0f : 1.0f;
    PairParams r = q; r.out = dnew; r.bimg1 = dimg1; r.bimg2 = dimg2;
    if (!pair_bf16_supported(r)) { printf("%-16s not supported by pair_bf16\n", c.name); continue; }
    m = launch_resblock_pair(q, s);
    if (!m) m = launch_pair_bf16(r, s);
    if (m) { printf("%s: %s\n", c.name, m); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> a(nx), o(nx);
    CK(hipMemcpy(a.data(), dnew, nx * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), dold, nx * 4, hipMemcpyDeviceToHost));
    size_t ndiff = 0, first = 0; double maxd = 0, maxv = 0;
    for (size_t i = 0; i < nx; ++i) {
      if (memcmp(&a[i], &o[i], 4)) { if (!ndiff) first = i; ++ndiff; maxd = fmax(maxd, fabs((double)a[i] - o[i])); }
      maxv = fmax(maxv, fabs((double)o[i]));
    }
    q.accumulate = r.accumulate = 0; q.out_div = r.out_div = 1.0f;   // timing: no read-modify-write drift over the repetitions
    float ms_old, ms_new;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_resblock_pair(q, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_old, e0, e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_pair_bf16(r, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
#ifdef E2ETTS_BC_DIAG
    { unsigned long long d[8]; conv_bf16_read_diag(d);
      const char* nm[7] = {"ring + staging", "barrier", "conv1", "epilogue 1 + barriers", "conv2", "residual loads + epilogue 2", "whole"};
      printf("   diag (cycles per workgroup, wave 0; %llu workgroups):", d[7]);
      for (int i = 0; i < 7; ++i) printf(" %s=%.0f", nm[i], (double)d[i] / (double)(d[7] ? d[7] : 1));
      printf("\n"); }
#endif
    const double fl = 2.0 * 2.0 * T * c.C * (double)c.C * c.KW;
    printf("%-16s T=%7d | resblock_pair mode 2 %7.1f us %7.1f TFLOP/s | pair_bf16 %7.1f us %7.1f TFLOP/s | %zu differing of %zu (first at row %zu, max |d| %.3g, max |v| %.3g)%s\n",
           c.name, T, ms_old / reps * 1e3, fl / (ms_old / reps) / 1e9, ms_new / reps * 1e3, fl / (ms_new / reps) / 1e9, ndiff, nx, first / c.C, maxd, maxv,
           ndiff ? "  MISMATCH" : "");
    bad += ndiff ? 1 : 0;
    (void)hipFree(dx); (void)hipFree(db1); (void)hipFree(db2); (void)hipFree(dold); (void)hipFree(dnew); (void)hipFree(dwx1); (void)hipFree(dwx2);
    (void)hipFree(dwf); (void)hipFree(dimg1); (void)hipFree(dimg2);
  }
  // ---- whole ResBlocks: three pair_bf16 launches (dilations 1, 3, 5) against one rb_bf16 launch, bit for bit, both timed
  struct RShape { const char* name; int rows_per_frame, C, KW; bool acc; };
  RShape rshapes[] = {{"rb s3 k3", 256, 64, 3, false}, {"rb s3 k7", 256, 64, 7, false}, {"rb s3 k11+a", 256, 64, 11, true},
                      {"rb s4 k3", 512, 32, 3, false}, {"rb s4 k7+a", 512, 32, 7, true}, {"rb s4 k11", 512, 32, 11, false}};
  for (auto& c : rshapes) {
    if (filter && !strstr(c.name, filter)) continue;
    const int T = frames * c.rows_per_frame;
    const int dils[3] = {1, 3, 5};
    const size_t nx = (size_t)T * c.C, nw = (size_t)c.C * c.KW * c.C;
    std::vector<float> hx(nx), hs(nx);
    fill(hx, 21); fill(hs, 22);
    float *dx, *da, *dbb, *dold, *dnew, *dbias;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&da, nx * 4)); CK(hipMalloc(&dbb, nx * 4)); CK(hipMalloc(&dold, nx * 4)); CK(hipMalloc(&dnew, nx * 4));
    CK(hipMalloc(&dbias, 6 * c.C * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dold, hs.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dnew, hs.data(), nx * 4, hipMemcpyHostToDevice));
    void* img[3][2]; float* wx[3][2];
    for (int m = 0; m < 3; ++m) for (int h = 0; h < 2; ++h) {
      std::vector<float> hw(nw), hb(c.C);
      fill(hw, 30 + 2 * m + h); fill(hb, 40 + 2 * m + h);
      for (auto& x : hw) x *= 0.05f;
      auto px = pack_x3(hw, c.C, c.KW, c.C);
      CK(hipMalloc(&wx[m][h], px.size() * 4)); CK(hipMemcpy(wx[m][h], px.data(), px.size() * 4, hipMemcpyHostToDevice));
      CK(hipMalloc(&img[m][h], bf16_image_bytes(c.C, c.KW, c.C, 0)));
      const char* m0 = launch_bf16_image(wx[m][h], img[m][h], c.C, c.KW, c.C, 0, s);
      if (m0) { printf("%s: %s\n", c.name, m0); return 1; }
      CK(hipMemcpy(dbias + (2 * m + h) * c.C, hb.data(), c.C * 4, hipMemcpyHostToDevice));
    }
    PairParams q[3];
    for (int m = 0; m < 3; ++m) {
      q[m] = PairParams();
      q[m].x = m == 0 ? dx : (m == 1 ? da : dbb); q[m].out = m == 0 ? da : (m == 1 ? dbb : dold);
      q[m].b1 = dbias + (2 * m) * c.C; q[m].b2 = dbias + (2 * m + 1) * c.C; q[m].bimg1 = img[m][0]; q[m].bimg2 = img[m][1];
      q[m].B = 1; q[m].T = T; q[m].C = c.C; q[m].KW = c.KW; q[m].dil = dils[m]; q[m].x_bs = q[m].out_bs = (long long)T * c.C; q[m].slope = 0.1f; q[m].mode = 2;
    }
    q[2].accumulate = c.acc; q[2].out_div = c.acc ? 3.0f : 1.0f;
    RbParams r; r.x = dx; r.out = dnew; r.n_pairs = 3; r.B = 1; r.T = T; r.C = c.C; r.KW = c.KW; r.x_bs = r.out_bs = (long long)T * c.C; r.slope = 0.1f;
    r.accumulate = c.acc; r.out_div = c.acc ? 3.0f : 1.0f;
    for (int m = 0; m < 3; ++m) { r.bimg[m][0] = img[m][0]; r.bimg[m][1] = img[m][1]; r.b1[m] = q[m].b1; r.b2[m] = q[m].b2; r.dil[m] = dils[m]; }
    if (!rb_bf16_supported(r)) { printf("%-16s not supported by rb_bf16\n", c.name); continue; }
    const char* m = nullptr;
    for (int k = 0; k < 3 && !m; ++k) m = launch_pair_bf16(q[k], s);
    if (!m) m = launch_rb_bf16(r, s);
    if (m) { printf("%s: %s\n", c.name, m); return 1; }
    CK(hipStreamSynchronize(s));
    std::vector<float> a(nx), o(nx);
    CK(hipMemcpy(a.data(), dnew, nx * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o.data(), dold, nx * 4, hipMemcpyDeviceToHost));
    size_t ndiff = 0, first = 0; double maxd = 0, maxv = 0;
    for (size_t i = 0; i < nx; ++i) {
      if (memcmp(&a[i], &o[i], 4)) { if (!ndiff) first = i; ++ndiff; maxd = fmax(maxd, fabs((double)a[i] - o[i])); }
      maxv = fmax(maxv, fabs((double)o[i]));
    }
    q[2].accumulate = r.accumulate = 0; q[2].out_div = r.out_div = 1.0f;
    float ms_old, ms_new;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) for (int k = 0; k < 3; ++k) launch_pair_bf16(q[k], s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_old, e0, e1));
#ifdef E2ETTS_BC_DIAG
    { unsigned long long d[8]; conv_bf16_read_diag(d); }   // the pair launches' stamps: not this table's
#endif
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch_rb_bf16(r, s);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_new, e0, e1));
#ifdef E2ETTS_BC_DIAG
    { unsigned long long d[8]; conv_bf16_read_diag(d);
      const char* nm[7] = {"guards + x0 + image", "conv1 (3 pairs)", "image I + barriers (3)", "conv2 (3)", "x update + image X (3)", "store", "whole"};
      printf("   diag (cycles per workgroup, wave 0; %llu workgroups):", d[7]);
      for (int i = 0; i < 7; ++i) printf(" %s=%.0f", nm[i], (double)d[i] / (double)(d[7] ? d[7] : 1));
      printf("\n"); }
#endif
    const double fl = 3 * 2.0 * 2.0 * T * c.C * (double)c.C * c.KW;
    printf("%-16s T=%7d | 3 x pair_bf16 %7.1f us %7.1f TFLOP/s | rb_bf16 %7.1f us %7.1f TFLOP/s | %zu differing of %zu (first at row %zu, max |d| %.3g, max |v| %.3g)%s\n",
           c.name, T, ms_old / reps * 1e3, fl / (ms_old / reps) / 1e9, ms_new / reps * 1e3, fl / (ms_new / reps) / 1e9, ndiff, nx, first / c.C, maxd, maxv,
           ndiff ? "  MISMATCH" : "");
    bad += ndiff ? 1 : 0;
    (void)hipFree(dx); (void)hipFree(da); (void)hipFree(dbb); (void)hipFree(dold); (void)hipFree(dnew); (void)hipFree(dbias);
    for (int mm = 0; mm < 3; ++mm) for (int h = 0; h < 2; ++h) { (void)hipFree(wx[mm][h]); (void)hipFree(img[mm][h]); }
  }
  printf(bad ? "FAILED: %d shapes differ\n" : "all shapes bit-identical\n", bad);
  return bad ? 1 : 0;
}
