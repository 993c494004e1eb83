// resblock_chain: a WHOLE HiFi-GAN ResBlock1 with kernel size 3 -- three (dilated conv -> leaky ReLU -> conv -> + x) pairs -- in ONE kernel.
//
// Reference V/layers.py:33-40:   for c1, c2 in zip(convs1, convs2): xt = c2(lrelu(c1(lrelu(x)))); x = xt + x      (dilations 1, 3, 5)
// and, as in resblock_pair.hip, the running sum over the parallel ResBlocks / num_kernels (V/generator.py:44-48).
//
// Why: the k = 3 pairs at 32 / 64 channels sit below the MFMA / HBM ridge even as fused pairs (48 FLOP / B against ~132: profiles/r1
// measured 3.9 TB/s of HBM traffic for resblock_pair_32).  Chaining the three pairs of the ResBlock leaves x in, x3 out (+ the running
// sum in) as the only HBM traffic: a third of the pair form's.
//
// How.  A workgroup owns R consecutive positions of one utterance; ALL three pairs are computed on all R positions, and the positions
// whose receptive field leaves the tile simply come out wrong and are discarded: pair m consumes d_m + 1 positions at either edge, so of
// R positions R - 2 H are valid at the end, H = sum_m (d_m + 1) = 12 (R = 256: 9 % recompute; R = 128: 23 %).
//   * the residual stream x_m lives in REGISTERS, in the MFMA accumulator layout of the wave that owns those (positions, channels) --
//     x_{m+1} = (acc + b2) + x_m is element-wise there, no LDS or HBM traffic at all;
//   * LDS holds two A-operand images (bf16 hi | lo rows of 144 B per 32 channels, as in conv_gemm.hip): X = lrelu(x_m) with d_max guard
//     rows of zeros at either end (conv1's dilated taps read row-shifted views of it) and I = lrelu(c1 + b1) with one guard row (conv2);
//     positions outside [0, T) are written as zeros into both (the zero padding of the two convolutions);
//   * weights come as MFMA fragments from L2, conv1 | conv2 of pair 0, 1, 2 contiguous (engine.hip lays the ResBlock's images out so);
//   * the MFMAs compute the TRANSPOSED product D^T = W . X^T (weights as the A operand, activations as B; the operand images are the
//     same either way): an accumulator lane then holds ONE position and, per register quad, four CONSECUTIVE channels -- so x_0 comes
//     in and x_3 goes out as float4 per lane, and an A-operand image row is written with two ds_write_b64 per quad (packed
//     v_cvt_pk_bf16_f32) instead of eight 2-byte stores.  (With positions on the registers -- conv_gemm's orientation -- these image
//     writes, not the MFMAs, set the kernel's time: 2.25 ms per ResBlock at 32 channels against 1.2 ms for three pair launches.)
// Arithmetic per output element is that of three resblock_pair launches in the same order (chunk-major, tap, k-step; lo*hi, hi*lo, hi*hi;
// (acc + b2) + x; lrelu then split), so valid samples are bit-identical to them (tests/test_gpu_parity.py).
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int LDK = 36;  // LDS row: 32 bf16 hi | 32 bf16 lo | 16 B pad = 36 words
constexpr int NP = 3;    // pairs per ResBlock1

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 r = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, r);
}

template <int R, int C, int WM, int WN>
constexpr int chain_threads() { return 64 * (R / WM) * (C / WN); }

template <int R, int C, int WM, int WN, bool SPLIT, bool ACCUM>
__global__ __launch_bounds__((chain_threads<R, C, WM, WN>()), 2) void resblock_chain_kernel(const ChainParams p, const RowMap rm) {
  constexpr int NCH = C / 32;
  constexpr int NWN = C / WN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int NWAVE = (R / WM) * NWN;
  constexpr int NTHR = 64 * NWAVE;
  static_assert(NWAVE == 4 || NWAVE == 8, "4 or 8 wavefronts per workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KW = p.KW;
  const int hk = (KW - 1) / 2;
  int H = 0, gx = 0;
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    H += hk * (p.dil[m] + 1);
    gx = max(gx, hk * p.dil[m]);
  }
  const int RO = R - 2 * H;             // valid output positions per tile
  const int xrows = R + 2 * gx;         // X image rows per chunk (guard rows of zeros at either end)
  const int irows = R + 2 * hk;         // I image rows per chunk
  float* Xs = smem;
  float* Is = smem + NCH * xrows * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;

  int b, xb;  // compact 1-D grid of a ragged batch, or (x, y) of the padded grid: as in resblock_pair.hip
  if (rm.n > 0) {
    if (!rowmap_find(rm, (int)blockIdx.x, b, xb)) return;
  } else {
    b = blockIdx.y;
    xb = blockIdx.x;
  }
  const int t_act = p.act_rows ? min(p.act_rows[b], p.T) : p.T;
  const int mtiles = (t_act + RO - 1) / RO;
  // XCD-aware tile order, as in resblock_pair.hip: every XCD walks a contiguous eighth of this utterance's tiles
  const int eighth = (mtiles + 7) >> 3;
  const int tile = (xb & 7) * eighth + (xb >> 3);
  if ((xb >> 3) >= eighth || tile >= mtiles) return;
  const int origin = tile * RO - H;     // global position of tile row 0

  const float* x_b = p.x + (long long)b * p.x_bs;
  float* out_b = p.out + (long long)b * p.out_bs;
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_b), 0, (int)((long long)p.T * C * 4), 0x00020000);
  // fragment order [32-column tile][tap][chunk][k-step][hi|lo][lane][8 bf16] per convolution; the six images of the ResBlock are contiguous
  const int frag_words = NCH * KW * NCH * 1024;  // 32-bit words of one convolution's image
  const __amdgpu_buffer_rsrc_t wf_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wfrag), 0, 2 * NP * frag_words * 4, 0x00020000);

  f32x16 acc[MT][NT], xres[MT][NT];

  // ---- zero the guard rows of both images (never written again)
  for (int i = tid; i < NCH * 2 * gx * LDK; i += NTHR) {
    const int c = i / (2 * gx * LDK), w = i % (2 * gx * LDK);
    const int row = w / LDK < gx ? w / LDK : R + w / LDK;   // rows [0, gx) and [R + gx, R + 2 gx)
    Xs[(c * xrows + row) * LDK + w % LDK] = 0.f;
  }
  for (int i = tid; i < NCH * 2 * hk * LDK; i += NTHR) {
    const int c = i / (2 * hk * LDK), w = i % (2 * hk * LDK);
    const int row = w / LDK < hk ? w / LDK : R + w / LDK;
    Is[(c * irows + row) * LDK + w % LDK] = 0.f;
  }

  // ---- x_0 into registers, accumulator layout of the transposed product: lane (li, lh) = position li of the 32-position tile,
  // register 4 q + i = channel 8 q + 4 lh + i of the 32-channel tile -> one float4 per (tile, q)
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int g = origin + wm * WM + m * 32 + li;
    const int gc = min(max(g, 0), p.T - 1);
    const bool ok = g >= 0 && g < p.T;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (gc * C + wn * WN + n * 32 + 8 * q + 4 * lh) * 4, 0, 0));
        xres[m][n][4 * q + 0] = ok ? v.x : 0.f;
        xres[m][n][4 * q + 1] = ok ? v.y : 0.f;
        xres[m][n][4 * q + 2] = ok ? v.z : 0.f;
        xres[m][n][4 * q + 3] = ok ? v.w : 0.f;
      }
  }

  // an accumulator-layout tile -> operand image: (+ bias), lrelu, zero outside [0, T), split to bf16 hi | lo; four consecutive channels
  // of one position per register quad = 8 bytes of the hi half and 8 of the lo half of that position's row
  auto write_image = [&](const f32x16 (&src)[MT][NT], float* img, int rows_per_chunk, int guard, const float* bias /* [C] or null */) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int chunk = wn * NT + n;
      float4 bq[4];
      if (bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + wn * WN + n * 32 + 8 * q + 4 * lh);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = wm * WM + m * 32 + li;
        const int g = origin + row;
        const bool ok = g >= 0 && g < p.T;
        float* dst = img + (chunk * rows_per_chunk + row + guard) * LDK + 2 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4] = {src[m][n][4 * q], src[m][n][4 * q + 1], src[m][n][4 * q + 2], src[m][n][4 * q + 3]};
          if (bias) { v[0] += bq[q].x; v[1] += bq[q].y; v[2] += bq[q].z; v[3] += bq[q].w; }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(v[i], v[i] * p.slope);
            v[i] = ok ? v[i] : 0.f;
          }
          uint2 hi, lo;
          hi.x = pack_bf16(v[0], v[1]);
          hi.y = pack_bf16(v[2], v[3]);
          *reinterpret_cast<uint2*>(dst + 4 * q) = hi;          // bf16 channels 8 q + 4 lh .. + 3 of the hi half
          if constexpr (SPLIT) {
            const float h0 = __builtin_bit_cast(float, hi.x << 16), h1 = __builtin_bit_cast(float, hi.x & 0xffff0000u);
            const float h2 = __builtin_bit_cast(float, hi.y << 16), h3 = __builtin_bit_cast(float, hi.y & 0xffff0000u);
            lo.x = pack_bf16(v[0] - h0, v[1] - h1);
            lo.y = pack_bf16(v[2] - h2, v[3] - h3);
            *reinterpret_cast<uint2*>(dst + 16 + 4 * q) = lo;   // same channels of the lo half
          }
        }
      }
    }
  };

  // ---- weight fragments (see resblock_pair.hip: DEEP = one whole tap ahead in two buffers at 32 channels)
  constexpr bool DEEP = C == 32;
  float4 bfr[DEEP ? 2 : 1][2][NT][2];
  auto load_frag_n = [&](auto par, int conv, int chunk, int j, int ks, int n) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int hl = 0; hl < (SPLIT ? 2 : 1); ++hl) {
      const int nt = wn * NT + n;
      const int soff = (conv * frag_words + (((((nt * KW + j) * NCH + chunk) * 2 + ks) * 2 + hl) << 8)) * 4;
      bfr[P][ks][n][hl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wf_rsrc, lane * 16, soff, 0));
    }
  };
  auto mma_tap = [&](auto par, const float* a_base, int nconv, int nchunk, int nj) {
    constexpr int P = decltype(par)::value;
    if constexpr (DEEP) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int n = 0; n < NT; ++n) load_frag_n(std::integral_constant<int, 1 - P>{}, nconv, nchunk, nj, ks, n);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[MT], al[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        ah[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + ks * 8));
        if constexpr (SPLIT) al[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + 16 + ks * 8));
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bfr[P][ks][n][0]);
          // transposed product: the weight fragment is the A operand (rows = output channels), the activation rows are B
          // (columns = positions); the three terms in conv_gemm's order: x_lo w_hi, x_hi w_lo, x_hi w_hi
          if constexpr (SPLIT) {
            const bf16x8 bl = __builtin_bit_cast(bf16x8, bfr[P][ks][n][1]);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[m], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[m], acc[m][n], 0, 0, 0);
          }
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[m], acc[m][n], 0, 0, 0);
        }
        if constexpr (!DEEP) {
          __builtin_amdgcn_sched_barrier(0);
          load_frag_n(std::integral_constant<int, 0>{}, nconv, nchunk, nj, ks, n);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  // the KW taps of one (conv, chunk) run, KW odd; after the last tap the fragment order continues with (nconv_end, nchunk_end, tap 0)
  auto run_taps = [&](auto par0, const float* a0, int row_step, int conv, int c, int nconv_end, int nchunk_end) {
    constexpr int P0 = DEEP ? decltype(par0)::value : 0;
    constexpr int P1 = DEEP ? 1 - P0 : 0;
    int j = 0;
    for (; j + 2 < KW; j += 2) {
      mma_tap(std::integral_constant<int, P0>{}, a0 + j * row_step, conv, c, j + 1);
      mma_tap(std::integral_constant<int, P1>{}, a0 + (j + 1) * row_step, conv, c, j + 2);
    }
    mma_tap(std::integral_constant<int, P0>{}, a0 + j * row_step, nconv_end, nchunk_end, 0);
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  // ---- main
  write_image(xres, Xs, xrows, gx, nullptr);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int n = 0; n < NT; ++n) load_frag_n(std::integral_constant<int, 0>{}, 0, 0, 0, ks, n);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing in flight at the loop heads: their waits stay counted (see conv_gemm.hip)
  zero_acc();

#pragma unroll
  for (int pm = 0; pm < NP; ++pm) {
    const int d = p.dil[pm];
    __syncthreads();  // X = lrelu(x_pm) visible (and every wave is done with I of the previous pair)
    // conv1: output position r reads X rows r + (j - hk) d; the whole image is resident, so no barrier between chunks.
    // A run of KW (odd) taps flips the DEEP buffer parity: with one chunk (C = 32) conv1 starts at parity 0 and conv2 at parity 1.
    for (int c = 0; c < NCH; ++c) {
      const bool lastc = c + 1 == NCH;
      run_taps(std::integral_constant<int, 0>{}, Xs + (c * xrows + gx - hk * d + wm * WM + li) * LDK + lh * 4, d * LDK, 2 * pm, c,
               lastc ? 2 * pm + 1 : 2 * pm, lastc ? 0 : c + 1);
    }
    write_image(acc, Is, irows, hk, p.b1[pm]);   // I = lrelu(c1 + b1), zero outside [0, T)
    zero_acc();
    __syncthreads();  // I visible; every wave is done reading X
    const int nxt = pm + 1 < NP ? 2 * pm + 2 : 0;  // the requests behind the very last tap re-read the first image: unused
    for (int c = 0; c < NCH; ++c) {
      const bool lastc = c + 1 == NCH;
      run_taps(std::integral_constant<int, 1>{}, Is + (c * irows + wm * WM + li) * LDK + lh * 4, LDK, 2 * pm + 1, c, lastc ? nxt : 2 * pm + 1,
               lastc ? 0 : c + 1);
    }
    // x_{pm+1} = (c2 + b2) + x_pm, in registers
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = *reinterpret_cast<const float4*>(p.b2[pm] + wn * WN + n * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          xres[m][n][4 * q + 0] = (acc[m][n][4 * q + 0] + bv.x) + xres[m][n][4 * q + 0];
          xres[m][n][4 * q + 1] = (acc[m][n][4 * q + 1] + bv.y) + xres[m][n][4 * q + 1];
          xres[m][n][4 * q + 2] = (acc[m][n][4 * q + 2] + bv.z) + xres[m][n][4 * q + 2];
          xres[m][n][4 * q + 3] = (acc[m][n][4 * q + 3] + bv.w) + xres[m][n][4 * q + 3];
          acc[m][n][4 * q + 0] = 0.f; acc[m][n][4 * q + 1] = 0.f; acc[m][n][4 * q + 2] = 0.f; acc[m][n][4 * q + 3] = 0.f;
        }
      }
    if (pm + 1 < NP) write_image(xres, Xs, xrows, gx, nullptr);  // X is dead since the barrier above
  }

  // ---- out = x_3 (+ out, / div) on the positions this tile owns: tile rows [H, R - H), global rows < T; float4 per lane and quad
  const int g_end = min((tile + 1) * RO, p.T);
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int row = wm * WM + m * 32 + li;
    const int g = origin + row;
    const bool mine = row >= H && row < R - H && g < g_end;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float* o = out_b + (long long)min(max(g, 0), p.T - 1) * C + wn * WN + n * 32 + 8 * q + 4 * lh;
        f32x4_t v = {xres[m][n][4 * q], xres[m][n][4 * q + 1], xres[m][n][4 * q + 2], xres[m][n][4 * q + 3]};
        if constexpr (ACCUM) {
          const f32x4_t ov = *reinterpret_cast<const f32x4_t*>(o);
          v += ov;
          if (p.out_div != 1.0f) v = v / p.out_div;
        }
        if (mine) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(o));
      }
  }
}

template <int R, int C, int WM, int WN, bool SPLIT>
const char* launch_chain_cfg(const ChainParams& p, hipStream_t s) {
  const int hk = (p.KW - 1) / 2;
  int H = 0, gx = 0;
  for (int m = 0; m < NP; ++m) {
    H += hk * (p.dil[m] + 1);
    gx = std::max(gx, hk * p.dil[m]);
  }
  const int RO = R - 2 * H;
  if (RO < R / 2) return "resblock_chain: receptive field too wide for the tile";
  const size_t lds = (size_t)(C / 32) * ((R + 2 * gx) + (R + 2 * hk)) * LDK * sizeof(float);
  if (lds > 160 * 1024) return "resblock_chain: LDS images exceed the CU's 160 KiB";
  static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the opt-in, once per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_chain_kernel<R, C, WM, WN, SPLIT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_chain_kernel<R, C, WM, WN, SPLIT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  const int mtiles = (p.T + RO - 1) / RO;
  dim3 grid((mtiles + 7) / 8 * 8, p.B);
  RowMap rm;
  if (p.act_rows && p.act_rows_host && p.B <= ROWMAP_MAX) {  // ragged: only the blocks of tiles that exist (resblock_pair.hip)
    rm.n = p.B;
    rm.identity();
    rm.cum[0] = 0;
    for (int b = 0; b < p.B; ++b) {
      const int mt = (std::min(std::max(p.act_rows_host[b], 0), p.T) + RO - 1) / RO;
      rm.cum[b + 1] = rm.cum[b] + (mt + 7) / 8 * 8;
    }
    if (rm.cum[p.B] == 0) return nullptr;
    grid = dim3(rm.cum[p.B]);
  }
  constexpr int NTHR = chain_threads<R, C, WM, WN>();
  if (p.accumulate)
    hipLaunchKernelGGL((resblock_chain_kernel<R, C, WM, WN, SPLIT, true>), grid, dim3(NTHR), lds, s, p, rm);
  else
    hipLaunchKernelGGL((resblock_chain_kernel<R, C, WM, WN, SPLIT, false>), grid, dim3(NTHR), lds, s, p, rm);
  return hipGetLastError() == hipSuccess ? nullptr : "resblock_chain: launch failed";
}

}  // namespace

bool resblock_chain_supported(int C, int KW, const int* dil, int n_dil) {
  if (!((C == 32 || C == 64) && KW == 3 && n_dil == NP)) return false;
  for (int m = 0; m < NP; ++m)
    if (dil[m] < 1 || dil[m] > 8) return false;
  return true;
}

double resblock_chain_flops(const ChainParams& p) { return NP * 2.0 * 2.0 * p.B * (double)p.T * p.act_frac * p.C * p.KW * p.C; }

double resblock_chain_bytes(const ChainParams& p) {
  return 4.0 * ((double)p.B * p.T * p.act_frac * p.C * (2.0 + (p.accumulate ? 1 : 0)) + NP * 2.0 * p.C * p.KW * p.C);
}

const char* launch_resblock_chain(const ChainParams& p, hipStream_t s) {
  if (!p.x || !p.wfrag || !p.out) return "resblock_chain: null pointer";
  for (int m = 0; m < NP; ++m) {
    if (!p.b1[m] || !p.b2[m]) return "resblock_chain: null bias";
    if (((uintptr_t)p.b1[m] | (uintptr_t)p.b2[m]) & 15) return "resblock_chain: bias vectors must be 16-byte aligned (read as float4)";
  }
  if (p.B <= 0 || p.T <= 0) return "resblock_chain: bad dims";
  if (!resblock_chain_supported(p.C, p.KW, p.dil, NP)) return "resblock_chain: unsupported channels / kernel / dilations";
  if (p.mode != 1 && p.mode != 2) return "resblock_chain: mode must be 1 (bf16x3) or 2 (bf16)";
  if (p.slope < 0.f || p.slope > 1.f) return "resblock_chain: slope must lie in [0, 1]";
  if (p.out_div != 1.0f && !p.accumulate) return "resblock_chain: out_div needs accumulate";
  if ((((uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.wfrag) & 15) || (p.x_bs & 3) || (p.out_bs & 3)) return "resblock_chain: pointers must be 16-byte aligned";
  if ((long long)p.T * p.C * 4 >= (1LL << 31)) return "resblock_chain: one utterance must stay below 2 GiB (32-bit buffer offsets)";
  if (p.x == p.out) return "resblock_chain: in-place is not possible (tiles read their neighbours' rows)";
  if (p.C == 32) return p.mode == 1 ? launch_chain_cfg<256, 32, 64, 32, true>(p, s) : launch_chain_cfg<256, 32, 64, 32, false>(p, s);
  return p.mode == 1 ? launch_chain_cfg<128, 64, 64, 32, true>(p, s) : launch_chain_cfg<128, 64, 64, 32, false>(p, s);
}

}  // namespace e2etts
