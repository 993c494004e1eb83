// Bare bf16 MFMA loop: what the chip sustains with NOTHING but v_mfma_f32_32x32x16_bf16 on random operands held in registers,
// every SIMD busy (and on zeros, where it holds a higher clock).  The practical ceiling the split-precision kernels are measured
// against in DESIGN.md section 5 -- the nominal 2.5 PFLOP/s assumes 2.4 GHz under load, which this part does not hold.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD / 1) void mfma_loop(const uint4* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, in[(tid * 8 + i) & 0xffff]);
    b[i] = __builtin_bit_cast(bf16x8, in[(tid * 8 + 4 + i) & 0xffff]);
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[tid] = s;
}

// The same for the exact-fp32 path: nothing but v_mfma_f32_32x32x2_f32 (64 cycles each, 4096 FLOP), 16 independent-enough MFMAs per iteration.
__global__ __launch_bounds__(256) void mfma_loop_f32(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(tid * 8 + i) & 0xffff];
    b[i] = in[(tid * 8 + 4 + i) & 0xffff];
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(i + j) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int CUS = 256, iters = 200000;
  std::vector<unsigned> h(65536 * 4);
  uint4* din; float* dout;
  CK(hipMalloc(&din, h.size() * 4));
  CK(hipMalloc(&dout, CUS * 12 * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int zeros = 0; zeros < 2; ++zeros) {
    srand(1);
    for (auto& w : h) {  // two bf16 per word, N(0,1)-ish finite values
      if (zeros) { w = 0; continue; }
      auto rb = []() { float f = (float)(rand() % 2001 - 1000) / 500.f; unsigned u; memcpy(&u, &f, 4); return u >> 16; };
      w = rb() | (rb() << 16);
    }
    CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int wps = 1; wps <= 2; ++wps) {
      const int blocks = CUS * wps;  // 256-thread blocks: 4 waves = one per SIMD; wps blocks per CU
      hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, din, dout, 1000);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double mfmas = (double)blocks * 4 * iters * 16;          // per wave 16 MFMAs per iteration
      const double flops = mfmas * 2.0 * 32 * 32 * 16;
      const double cyc_per_simd = (double)iters * 16 * 32 * wps;     // 32 cycles per MFMA, wps waves share a SIMD
      printf("%s operands, %d wave(s) per SIMD: %.1f ms  %.0f TFLOP/s bf16 dense (= %.0f TFLOP/s of split-precision products)  implied clock %.2f GHz\n",
             zeros ? "zero  " : "random", wps, ms, flops / ms / 1e9, flops / ms / 1e9 / 3, cyc_per_simd / (ms * 1e6));
    }
  }
  // exact fp32: operands small enough that 3.2 M accumulations stay finite
  std::vector<float> hf(65536 * 4);
  for (int zeros = 0; zeros < 2; ++zeros) {
    srand(2);
    for (auto& w : hf) w = zeros ? 0.f : (float)(rand() % 2001 - 1000) / 4.0e6f;
    CK(hipMemcpy(din, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    const int it32 = 50000;
    for (int wps = 1; wps <= 3; ++wps) {
      const int blocks = CUS * wps;
      hipLaunchKernelGGL(mfma_loop_f32, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float*>(din), dout, 1000);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(mfma_loop_f32, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const float*>(din), dout, it32);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double mfmas = (double)blocks * 4 * it32 * 16;
      const double flops = mfmas * 2.0 * 32 * 32 * 2;
      const double cyc_per_simd = (double)it32 * 16 * 64 * wps;      // 64 cycles per MFMA
      printf("fp32 %s operands, %d wave(s) per SIMD: %.1f ms  %.1f TFLOP/s fp32 (v_mfma_f32_32x32x2_f32)  implied clock %.2f GHz\n",
             zeros ? "zero  " : "random", wps, ms, flops / ms / 1e9, cyc_per_simd / (ms * 1e6));
    }
  }
  return 0;
}
