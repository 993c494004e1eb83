// conv_ksplit: "same" 1-D convolution / Linear in exact fp32 for the PHONEME-LEVEL layers -- the encoder's FFT blocks (reference
// U/blocks/transformer.py:213-240, 289-297) and the duration / pitch / energy predictors (U/layers.py:410-420, 491-505) -- with the K
// dimension split four ways INSIDE the workgroup.
//
// Why a second convolution kernel.  These layers see B x L phonemes, not B x T frames: 128 rows at B = 1, 4 096 at B = 32 -- 1.3 % of the
// step's FLOPs -- but each output tile sums over K = KW x Cin up to 3 456 (the FFN's k = 9 convolution), and in conv_gemm.hip one wave
// walks that whole K as ONE chain of dependent v_mfma_f32_32x32x2_f32: 1 728 MFMAs x 64 cycles = 46 us per launch however few rows there
// are, 30 such launches per step.  Here the four waves of a workgroup share ONE 32 x 32 (or 32 x 64) output tile and take the 32-channel
// chunks of Cin round-robin (wave w: chunks w, w + 4, ...), each with its own slab rows in a private piece of LDS and its own weight
// fragments from L2 (launch_f32_to_frag's order): no workgroup barrier until the end, a chain a quarter as long, four times the waves.
// The partial sums meet in LDS and are added in a FIXED order, ((P0 + P1) + (P2 + P3)), before bias / activation / residual.
//
// That order differs from conv_gemm's single chain, so this kernel serves these layers at EVERY batch size: an utterance's durations,
// pitch and energy buckets and encoder output do not depend on what it is batched with (tests/test_gpu_parity.py: B = 1 against the
// same utterance inside B = 32, bit for bit), and the discrete outputs keep matching the reference on every fixture.
#include <type_traits>

#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int LDK = 36;        // LDS row stride (floats): 32 channels + 4 pad
constexpr int KS_MAX_HALO = 16;  // dil * (KW - 1) supported (k = 9: 8)

template <int NT>
__global__ __launch_bounds__(256) void conv_ksplit_kernel(const ConvParams p) {
  constexpr int SROWS = 32 + KS_MAX_HALO;
  __shared__ __attribute__((aligned(16))) float slab[4][SROWS * LDK];   // one private slab per wave
  __shared__ __attribute__((aligned(16))) float red[3][NT][16 * 64];    // partial sums of waves 1..3

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z;
  const int t0 = blockIdx.x * 32;
  const int n0 = blockIdx.y * (32 * NT);
  const int halo = p.dil * (p.KW - 1);
  const int srows = 32 + halo;
  const int nchunk = (p.Cin + 31) / 32;
  const int ntile32 = (p.Cout + 31) / 32;
  const float* in_b = p.in + (long long)b * p.in_bs;
  const __amdgpu_buffer_rsrc_t in_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, (int)((long long)p.T * p.in_ld * 4), 0x00020000);
  // fragment order [32-column tile][tap][chunk][q 0..3][lane][4 floats]: 1 KiB per (tile, tap, chunk, q)
  const __amdgpu_buffer_rsrc_t wf_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.wfrag), 0, (int)((long long)ntile32 * p.KW * nchunk * 4096), 0x00020000);
  int fnt[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) fnt[n] = min(n0 / 32 + n, ntile32 - 1);

  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;

  float* my = slab[wave];
  const int srow = lane >> 3, sc4 = (lane & 7) * 4;   // staging: 8 lanes per row, 8 rows per pass
  float4 bfr[2][4][NT];                               // weight fragments of the current and the next tap
  auto load_frag = [&](int buf, int chunk, int j) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int soff = (((fnt[n] * p.KW + j) * nchunk + chunk) * 4 + q) * 1024;
        bfr[buf][q][n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wf_rsrc, lane * 16, soff, 0));
      }
  };

  for (int c = wave; c < nchunk; c += 4) {
    load_frag(0, c, 0);
    // this wave's slab: rows t0 - pad .. + srows of channels 32 c .. + 31 (zero outside [0, T) and beyond Cin)
    for (int r0 = 0; r0 < srows; r0 += 8) {
      const int r = r0 + srow;
      const int t = t0 - p.pad + r;
      const int ch = c * 32 + sc4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < srows && t >= 0 && t < p.T && ch < p.Cin)
        v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (t * p.in_ld + ch) * 4, 0, 0));
      if (r < srows) *reinterpret_cast<float4*>(my + r * LDK + sc4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the slab was written by other lanes of this wave: keep the reads below behind the writes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // taps in pairs so that the fragment buffer of a tap is a compile-time index; the next tap's fragments are requested a tap ahead
    auto tap = [&](auto buf, int j) {
      constexpr int P = decltype(buf)::value;
      const float* a_base = my + (li + j * p.dil) * LDK + lh * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(a_base + q * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 w = bfr[P][q][n];
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc[n], 0, 0, 0);
        }
      }
    };
    int j = 0;
    for (; j + 1 < p.KW; j += 2) {
      load_frag(1, c, j + 1);
      tap(std::integral_constant<int, 0>{}, j);
      if (j + 2 < p.KW) load_frag(0, c, j + 2);
      tap(std::integral_constant<int, 1>{}, j + 1);
    }
    if (j < p.KW) tap(std::integral_constant<int, 0>{}, j);   // odd KW: the last tap's fragments are in buffer 0
    __builtin_amdgcn_wave_barrier();   // every lane is done reading the slab before the next chunk overwrites it
  }

  // ---- ((P0 + P1) + (P2 + P3)): waves 1..3 hand their partial sums over, wave 0 adds them in that order and runs the epilogue
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][n][r * 64 + lane] = acc[n][r];
  }
  __syncthreads();
  if (wave != 0) return;
  const int len = p.lens ? p.lens[b] : p.T;
  float* out_b = p.out + (long long)b * p.out_bs;
  const float* res_b = p.res ? p.res + (long long)b * p.res_bs : nullptr;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + n * 32 + li;
    const bool col_ok = col < p.Cout;
    const float bias = (p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float p01 = acc[n][r] + red[0][n][r * 64 + lane];
      const float p23 = red[1][n][r * 64 + lane] + red[2][n][r * 64 + lane];
      float v = (p01 + p23) + bias;
      if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (p.act == ACT_TANH) v = tanhf(v);
      else if (p.act == ACT_LRELU) v = v >= 0.f ? v : v * p.act_slope;
      else if (p.act == ACT_SWISH) v = v * (1.0f / (1.0f + expf(-v)));
      if (t < p.T && col_ok) {
        if (res_b) v += res_b[(long long)t * p.res_ld + col];
        if (t >= len) v = 0.f;
        out_b[(long long)t * p.out_ld + col] = v;
      }
    }
  }
}

}  // namespace

bool conv_ksplit_supported(const ConvParams& p) {
  return p.wfrag && p.x3 == 0 && !p.accumulate && p.in_slope == 1.0f && p.zero_tap_split == 0 && !p.act_rows && p.KW >= 1 && p.dil >= 1 &&
         p.dil * (p.KW - 1) <= KS_MAX_HALO && p.pad >= 0 && p.pad <= p.dil * (p.KW - 1) && (p.Cin % 4) == 0 && (p.in_ld % 4) == 0;
}

const char* launch_conv_ksplit(const ConvParams& p, hipStream_t s) {
  if (!p.in || !p.wfrag || !p.out) return "conv_ksplit: null pointer";
  if (p.B <= 0 || p.T <= 0 || p.Cin <= 0 || p.Cout <= 0) return "conv_ksplit: bad dims";
  if (!conv_ksplit_supported(p)) return "conv_ksplit: unsupported launch (fp32 fragment-order weights, dil (KW - 1) <= 16, no input activation)";
  if (((uintptr_t)p.in & 15) || (p.in_bs & 3)) return "conv_ksplit: input must be 16-byte aligned";
  if (p.in_ld < p.Cin || p.out_ld < p.Cout || (p.res && p.res_ld < p.Cout)) return "conv_ksplit: row stride < channels";
  if ((long long)p.T * p.in_ld * 4 >= (1LL << 31)) return "conv_ksplit: one utterance must stay below 2 GiB (32-bit buffer offsets)";
  const int rt = (p.T + 31) / 32;
  // 64-column tiles halve the slab traffic per MFMA; 32-column tiles when that would leave most CUs without a workgroup
  const bool wide = p.Cout > 32 && (long long)rt * ((p.Cout + 63) / 64) * p.B >= 256;
  if (wide) {
    dim3 grid(rt, (p.Cout + 63) / 64, p.B);
    hipLaunchKernelGGL(conv_ksplit_kernel<2>, grid, dim3(256), 0, s, p);
  } else {
    dim3 grid(rt, (p.Cout + 31) / 32, p.B);
    hipLaunchKernelGGL(conv_ksplit_kernel<1>, grid, dim3(256), 0, s, p);
  }
  return hipGetLastError() == hipSuccess ? nullptr : "conv_ksplit: launch failed";
}

}  // namespace e2etts
