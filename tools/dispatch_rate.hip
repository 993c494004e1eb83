// How fast does the chip hand out workgroups?  conv_gemm's ragged launches contain whole runs of workgroups that find nothing to do
// (rows beyond an utterance's act_rows) and every one-tile workgroup is launched, runs and retires: what a launch costs per workgroup
// decides between that form and a persistent grid that pulls tiles from a queue.
//   (a) grids of N workgroups that exit at once, with conv_gemm's footprint (256 threads, 168 VGPRs, 25 KB of LDS);
//   (b) the same grid where every workgroup spins for `busy` microseconds first (slots stay full: the dispatcher works behind retiring
//       workgroups, as in a real launch), with a fraction of empties in runs, as a ragged batch produces them.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_rate.hip -o tools/bin/dispatch_rate && tools/bin/dispatch_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 3) void wg_kernel(const int* __restrict__ flags, float* out, int run, int period, long long busy_ticks) {
  extern __shared__ float sm[];
  asm volatile("" ::: "v160");  // hold conv_gemm's register footprint
  const int g = blockIdx.x >> 3;
  const bool empty = period > 0 && (g % period) >= run;   // runs of (period - run) empties after `run` real groups, per XCD share
  if (flags[0] == 12345) sm[threadIdx.x] = 1.f;           // keep the LDS allocation alive
  if (empty || busy_ticks == 0) return;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < busy_ticks) __builtin_amdgcn_s_sleep(8);
  if (flags[0] == 54321) out[blockIdx.x] = sm[0];
}

int main() {
  int* flags; float* out;
  CK(hipMalloc(&flags, 64)); CK(hipMemset(flags, 0, 64)); CK(hipMalloc(&out, 1 << 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 25 * 1024;
  auto time_it = [&](int n, int run, int period, long long busy) -> float {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(wg_kernel, dim3(n), dim3(256), lds, 0, flags, out, run, period, busy);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    return best;
  };
  printf("(a) empty workgroups (256 threads, 168 VGPRs, 25 KB LDS)\n");
  for (int n : {768, 7680, 76800, 768000}) {
    const float ms = time_it(n, 0, 0, 0);
    printf("  %7d workgroups: %8.1f us  -> %6.1f ns per workgroup (%5.1f ns per workgroup and XCD)\n", n, ms * 1e3, ms * 1e6 / n, ms * 1e6 / n * 8);
  }
  // wall_clock64 ticks at 100 MHz
  printf("(b) busy workgroups of 100 / 500 us, 3072 real ones (4 rounds of 768), with and without runs of empties\n");
  for (long long us : {100LL, 500LL}) {
    const long long ticks = us * 100;
    const float ideal = 4.0f * us;
    const float t0 = time_it(3072, 0, 0, ticks);
    // 60 % more workgroups as empties: per XCD share, runs of 5 real then 3 empty
    const float t1 = time_it(3072 * 8 / 5, 5, 8, ticks);
    const float t2 = time_it(3072 * 8 / 5, 50, 80, ticks);
    printf("  busy %3lld us: no empties %8.1f us (ideal %6.1f), 5 real + 3 empty %8.1f us, 50 real + 30 empty %8.1f us\n", us, t0 * 1e3, ideal, t1 * 1e3, t2 * 1e3);
  }
  printf("(c) one persistent round: 768 workgroups x 4 x busy\n");
  for (long long us : {100LL, 500LL}) {
    const float t0 = time_it(768, 0, 0, us * 100 * 4);
    printf("  busy 4 x %3lld us: %8.1f us (ideal %6.1f)\n", us, t0 * 1e3, 4.0f * us);
  }
  return 0;
}
