// Does v_mfma_f32_16x16x4_f32 sum its four k-products in the order two chained v_mfma_f32_32x32x2_f32 would (k ascending, one fp32
// FMA per product)?  If so, a 16 x 16-tile kernel can serve launches with few rows -- four times the wavefronts, a quarter of the MFMA
// time each -- with the SAME bits as the 32 x 32-tile kernels.  Compares, bit for bit, on random data with a wide dynamic range:
// (a) a 32x32x2 chain, (b) a 16x16x4 chain, (c) a scalar fmaf chain in k order.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_order.hip -o tools/bin/mfma_order && tools/bin/mfma_order
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 256;

// A [32][K] row-major, B [K][32] row-major (B[k][n])
__global__ void k32(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < K; k += 2) {
    const float a = A[li * K + k + lh];      // A operand: row li, k = lh
    const float b = B[(k + lh) * 32 + li];   // B operand: col li, k = lh
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    C[row * 32 + li] = acc[r];
  }
}

// same product on 16 x 16 tiles: 4 tiles (tm, tn), each a 16x16x4 chain
__global__ void k16(const float* A, const float* B, float* C) {
  const int lane = threadIdx.x, l16 = lane & 15, kg = lane >> 4;
  for (int tm = 0; tm < 2; ++tm)
    for (int tn = 0; tn < 2; ++tn) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < K; k += 4) {
        const float a = A[(tm * 16 + l16) * K + k + kg];
        const float b = B[(k + kg) * 32 + tn * 16 + l16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      }
      // C/D layout of 16x16: col = lane & 15, row = 4 * (lane >> 4) + r
      for (int r = 0; r < 4; ++r) C[(tm * 16 + 4 * kg + r) * 32 + tn * 16 + l16] = acc[r];
    }
}

__global__ void kref(const float* A, const float* B, float* C) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= 32 * 32) return;
  const int row = i / 32, col = i % 32;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = __builtin_fmaf(A[row * K + k], B[k * 32 + col], acc);
  C[i] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  std::vector<float> A(32 * K), B(K * 32);
  srand(7);
  auto rnd = []() {  // wide dynamic range so that rounding differences show
    const float m = (float)(rand() % 20001 - 10000) / 10000.f;
    return ldexpf(m, rand() % 24 - 12);
  };
  for (auto& v : A) v = rnd();
  for (auto& v : B) v = rnd();
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 3 * 1024 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC + 1024);
  hipLaunchKernelGGL(kref, dim3(4), dim3(256), 0, 0, dA, dB, dC + 2048);
  std::vector<float> C(3 * 1024);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  int d16 = 0, dref32 = 0, dref16 = 0;
  double maxrel = 0;
  for (int i = 0; i < 1024; ++i) {
    d16 += memcmp(&C[i], &C[1024 + i], 4) != 0;
    dref32 += memcmp(&C[i], &C[2048 + i], 4) != 0;
    dref16 += memcmp(&C[1024 + i], &C[2048 + i], 4) != 0;
    maxrel = fmax(maxrel, fabs((double)C[i] - C[1024 + i]) / (fabs((double)C[i]) + 1e-30));
  }
  printf("elements that differ: 32x32x2 vs 16x16x4: %d / 1024 (max rel %.3e); 32x32x2 vs fmaf chain: %d; 16x16x4 vs fmaf chain: %d\n", d16, maxrel, dref32, dref16);
  printf("sample: %.9g %.9g %.9g\n", C[5], C[1024 + 5], C[2048 + 5]);
  return 0;
}
