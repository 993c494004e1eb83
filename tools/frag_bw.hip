// How fast can waves pull 1 KiB MFMA weight fragments (one buffer_load_dwordx4 per wave, scalar offset) out of an L2-resident image?
// The BFRAG conv kernel and the fused ResBlock kernel feed their B operand this way: per split-precision k-step and 32-column tile a
// wave loads 2 KiB (hi | lo) for 3 x MT MFMAs, i.e. at MT = 2 about 10.7 B per MFMA-cycle per wave, 43 B / clk per CU with 4 waves
// issuing -- this tool measures what the L1 / L2 path delivers with nothing else going on, for images of the sizes those kernels walk
// (one tap of 128 columns x 128 channels = 64 KiB; a whole k = 11 pair = 1408 KiB), with 1, 2 or 4 waves of a workgroup reading the
// same stream (2 = the 2 x 2 wave layout of the 128-column kernels, 4 = the 64- and 32-column fused kernels).
//   hipcc --offload-arch=gfx950 -O3 tools/frag_bw.hip -o tools/bin/frag_bw && tools/bin/frag_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int SHARE>
__global__ __launch_bounds__(256) void frag_loop(const float* __restrict__ img, int img_kib, int iters, float* __restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, img_kib * 1024, 0x00020000);
  const int nstreams = 4 / SHARE, region = img_kib / nstreams;  // KiB per stream; waves of one stream read the same addresses
  const int base = (wave / SHARE) * region;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
    for (int f = 0; f < region; f += 8) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, (base + f + i) * 1024, 0));
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
    }
  }
  out[blockIdx.x * 256 + tid] = acc.x + acc.y + acc.z + acc.w;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SHARE>
static int run(const float* img, float* out, int img_kib, int blocks, hipEvent_t e0, hipEvent_t e1) {
  const int region = img_kib / (4 / SHARE);
  const int iters = (int)(262144LL * 8 / region);  // ~2 GiB per wave-stream... scaled so every case moves the same bytes per wave
  hipLaunchKernelGGL(frag_loop<SHARE>, dim3(blocks), dim3(256), 0, 0, img, img_kib, 2, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(frag_loop<SHARE>, dim3(blocks), dim3(256), 0, 0, img, img_kib, iters, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)blocks * 4 * iters * region * 1024.0;
  const double tbs = bytes / ms / 1e9;
  printf("image %5d KiB, %d wave(s) per stream, %d workgroups/CU: %7.2f TB/s into registers = %6.1f GB/s per CU = %5.1f B/clk/CU at 2.1 GHz\n",
         img_kib, SHARE, blocks / 256, tbs, tbs * 1e3 / 256, tbs * 1e12 / 256 / 2.1e9);
  return 0;
}

int main() {
  const int max_kib = 2048;
  std::vector<float> h(max_kib * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 251) * 1e-3f;
  float *img, *out;
  CK(hipMalloc(&img, h.size() * 4));
  CK(hipMalloc(&out, 1024 * 256 * 4));
  CK(hipMemcpy(img, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int kib : {32, 64, 128, 1408}) {
    for (int wg = 1; wg <= 2; ++wg) {
      if (run<1>(img, out, kib, 256 * wg, e0, e1)) return 1;
      if (run<2>(img, out, kib, 256 * wg, e0, e1)) return 1;
      if (run<4>(img, out, kib, 256 * wg, e0, e1)) return 1;
    }
  }
  return 0;
}
