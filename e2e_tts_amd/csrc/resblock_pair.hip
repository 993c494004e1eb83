// resblock_pair: one (dilated conv -> leaky ReLU -> conv -> + x) pair of HiFi-GAN's ResBlock1 in ONE kernel.
//
// Reference V/layers.py:33-40:   xt = lrelu(x); xt = c1(xt); xt = lrelu(xt); xt = c2(xt); x = xt + x
// (c1: kernel k, dilation d; c2: kernel k, dilation 1; both C -> C channels, "same" zero padding) and, for the last
// pair of a ResBlock, the running sum over the parallel ResBlocks and its division by num_kernels
// (V/generator.py:44-48).
//
// Why: as two conv_gemm launches the intermediate [B, T, C] tensor (805 MB per stage at B = 32) is written to and read
// back from HBM, and the C = 32 / 64 stages and every k = 3 layer sit on the HBM roofline (profiles/r1: measured HBM
// bytes = algorithmic bytes, 4.4-5.0 TB/s).  Here the intermediate never leaves the CU: conv1's accumulators are
// split to bf16 hi | lo and written straight into LDS in A-operand layout, conv2 reads them as row-shifted views.
// HBM traffic per pair: x in, out out (+ out in when accumulating) instead of 5-6 tensor passes.
//
// Arithmetic is the split-precision ("bf16x3") or plain-bf16 MFMA path of conv_gemm.hip, with the same operation order
// per output element (chunk-major, tap, k-step; lo*hi, hi*lo, hi*hi), so the result is bit-identical to the two-launch
// form (tests/test_gpu_parity.py compares them).  Weights come as MFMA fragments straight from L2 (launch_x3_to_frag).
//
// Geometry.  A workgroup (4 waves) owns BMI intermediate rows = BMI - (KW-1) output rows:
//   x rows [o0 - pad2 - pad1, ... + BMI + (KW-1) d)  --conv1-->  intermediate rows [o0 - pad2, o0 - pad2 + BMI)
//   --conv2-->  output rows [o0, o0 + BMI - (KW-1)).   The (KW-1)/BMI recompute is the price of the fusion (4 % at k = 11,
//   BMI = 256).  Intermediate rows outside [0, T) are ZERO (conv2's zero padding), not conv1 of padding.
// LDS: one region, first the x slab of one 32-channel chunk [(BMI + (KW-1) d) x 36 words], then -- conv1's accumulators
// being complete -- overwritten by the intermediate [C/32][(BMI + KW - 1) x 36 words].
// MFMA orientation (round 2): the TRANSPOSED product D^T = W . X^T (weight fragment = A operand, activation rows = B), which puts a
// POSITION on each accumulator lane and four consecutive channels on each register quad: the intermediate is written with packed
// converts and 8-byte LDS stores (instead of a 2-byte store and a scalar convert per element), and the output epilogue fills its
// row-major LDS patches with one ds_write_b128 per quad.  Bits are unchanged by the swap.
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int LDK = 36;        // LDS row: 32 bf16 hi | 32 bf16 lo | 16 B pad = 36 words
constexpr int MAX_HALO1 = 64;  // (KW - 1) * dil of conv1
constexpr size_t PAIR_MAX_LDS = 160 * 1024;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 r = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  hi.x = pack_bf16(v.x, v.y);
  hi.y = pack_bf16(v.z, v.w);
  const float hx = __builtin_bit_cast(float, hi.x << 16), hy = __builtin_bit_cast(float, hi.x & 0xffff0000u);
  const float hz = __builtin_bit_cast(float, hi.y << 16), hw = __builtin_bit_cast(float, hi.y & 0xffff0000u);
  lo.x = pack_bf16(v.x - hx, v.y - hy);
  lo.y = pack_bf16(v.z - hz, v.w - hw);
}

// Workgroup = (BMI / WM) x (C / WN) wavefronts: 4 (256 threads; two or three workgroups per CU) at 32 / 64 / 128 channels, 8 (512 threads,
// 2 x 4 waves of 64 x 64, ONE workgroup per CU) at 256 channels, whose intermediate [8 chunks][BMI + KW - 1 rows] x 144 B = 155-159 KB
// fills the CU's LDS.  Same MFMA : weight-fragment ratio per wave as the 128-channel form (a 64-row tile would double the fragment
// traffic per MFMA to ~43 B / clk / CU, against ~64 available from L2).
template <int BMI, int C, int WM, int WN>
constexpr int pair_threads() { return 64 * (BMI / WM) * (C / WN); }

// MODE 0: exact fp32 (v_mfma_f32_32x32x2_f32; LDS rows hold 32 fp32 channels; fp32 fragment order of launch_f32_to_frag) -- round 2;
// MODE 1: bf16x3 split precision; MODE 2: plain bf16 (rows hold 32 bf16 hi | 32 bf16 lo; fragment order of launch_x3_to_frag).
template <int BMI, int C, int WM, int WN, int MODE, bool ACCUM>
__global__ __launch_bounds__((pair_threads<BMI, C, WM, WN>()), C == 32 ? 3 : 2) void resblock_pair_kernel(const PairParams p, const RowMap rm) {
  constexpr bool X3 = MODE != 0;
  constexpr bool SPLIT = MODE == 1;
  constexpr int KS = X3 ? 2 : 4;               // k-steps per 32-channel chunk: 2 x 16 (bf16 MFMA) or 4 x (4 MFMAs of k = 2) (fp32)
  constexpr int NCH = C / 32;                  // 32-channel chunks (K of both convs, and N tiles of the intermediate)
  constexpr int NWN = C / WN;
  constexpr int MT = WM / 32, NT = WN / 32;
  constexpr int NWAVE = (BMI / WM) * NWN;
  static_assert(NWAVE == 4 || NWAVE == 8, "4 or 8 wavefronts per workgroup");
  constexpr int SROWS = NWAVE * 8;             // slab rows one staging pass of the workgroup covers (8 threads per 32-channel row)
  constexpr int AROWS = (BMI + MAX_HALO1 + SROWS - 1) / SROWS;  // slab rows staged per thread (upper bound)

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int KW = p.KW, dil = p.dil;
  const int halo1 = dil * (KW - 1), pad1 = halo1 / 2, pad2 = (KW - 1) / 2;
  const int BMO = BMI - (KW - 1);
  const int irows = BMI + KW - 1;              // intermediate rows kept per chunk (the last KW-1 feed discarded outputs only)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

  // ragged batch whose lengths the host knows: a 1-D grid that holds every utterance's blocks back to back (kernels.h: RowMap; each
  // utterance's share is a multiple of 8 blocks, so `xb & 7` below is still the XCD the block runs on); else (xb, b) = (x, y) of the grid
  int b, xb;
  if (rm.n > 0) {
    if (!rowmap_find(rm, (int)blockIdx.x, b, xb)) return;
  } else {
    b = blockIdx.y;
    xb = blockIdx.x;
  }
  const int t_act = p.act_rows ? min(p.act_rows[b], p.T) : p.T;
  const int mtiles = (t_act + BMO - 1) / BMO;
  // One tile per workgroup: many short workgroups balance better than a few persistent ones (measured: 8 workgroups per CU walking 6
  // tiles each 53.4 ms/step, one tile each 51.9), and the partner workgroup on the CU covers this one's staging and epilogues.
  // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive block ids would put
  // neighbouring tiles -- which share (KW-1)(d+1) halo rows -- on different L2s.  Block x of an utterance's row of the grid (a
  // multiple of 8 wide) works on tile (x % 8) * ceil(mtiles / 8) + x / 8: every XCD walks a contiguous eighth of the tiles THIS
  // utterance really has (ragged batches), so all XCDs get the same share of every utterance.
  const int eighth = (mtiles + 7) >> 3;
  const int tile = (xb & 7) * eighth + (xb >> 3);
  if ((xb >> 3) >= eighth || tile >= mtiles) return;

  const float* x_b = p.x + (long long)b * p.x_bs;
  float* out_b = p.out + (long long)b * p.out_bs;
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_b), 0, (int)((long long)p.T * C * 4), 0x00020000);
  // fragment order [32-column tile][tap][chunk][k-step][hi|lo][lane][8 bf16]; conv2's image follows conv1's
  const int frag_words = NCH * KW * NCH * 1024;  // 32-bit words of one conv's image
  const __amdgpu_buffer_rsrc_t wf_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wfrag), 0, 2 * frag_words * 4, 0x00020000);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // ---- x slab staging (one 32-channel chunk): registers now, LDS later
  float4 areg[AROWS];
  bool aok[AROWS];
  const int srows = BMI + halo1;
  auto load_a = [&](int tile, int chunk) {
    const int x0 = tile * BMO - pad2 - pad1;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = lrow + i * SROWS;
      const int g = x0 + r;
      aok[i] = r < srows && g >= 0 && g < p.T;
      const int gc = min(max(g, 0), p.T - 1);
      areg[i] = __builtin_bit_cast(
          float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (gc * C + lc4) * 4, chunk * 128, 0));
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = lrow + i * SROWS;
      if (r < srows) {
        float4 v = aok[i] ? areg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = fmaxf(v.x, v.x * p.slope); v.y = fmaxf(v.y, v.y * p.slope);
        v.z = fmaxf(v.z, v.z * p.slope); v.w = fmaxf(v.w, v.w * p.slope);
        if constexpr (X3) {
          uint2 hi, lo;
          split4(v, hi, lo);
          uint2* row = reinterpret_cast<uint2*>(smem + r * LDK);
          row[lc4 >> 2] = hi;        // 4 bf16 = 8 bytes at bf16 index lc4
          row[8 + (lc4 >> 2)] = lo;  // lo half starts at byte 64
        } else {
          *reinterpret_cast<float4*>(smem + r * LDK + lc4) = v;
        }
      }
    }
  };

  // ---- weight fragments: B operand of one k-step, one 1 KiB load per (32-column tile, hi | lo)
  // DEEP (32 channels: 12 MFMAs per tap are far less than an L2 round trip, and registers are plentiful): the fragments of the next tap
  // are requested one whole tap ahead into the other of two buffers.  Which buffer a tap reads is a compile-time parity; taps
  // alternate, a run has an odd number of taps, so conv1 starts at parity 0 and conv2 at parity 1.
  constexpr bool DEEP = C == 32;
  float4 bfr[DEEP ? 2 : 1][KS][NT][SPLIT ? 2 : 1];
  auto load_frag_n = [&](auto par, int conv, int chunk, int j, int ks, int n) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int hl = 0; hl < (SPLIT ? 2 : 1); ++hl) {
      const int nt = wn * NT + n;
      // 4 KiB per (32-column tile, tap, chunk) in either order: [k-step 0..1][hi | lo] or [q 0..3], 1 KiB each
      const int piece = X3 ? ks * 2 + hl : ks;
      const int soff = (conv * frag_words + (((((nt * KW + j) * NCH + chunk) * 4) + piece) << 8)) * 4;
      bfr[P][ks][n][hl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wf_rsrc, lane * 16, soff, 0));
    }
  };
  auto load_frag = [&](int conv, int chunk, int j, int ks) {  // (prologue only)
#pragma unroll
    for (int n = 0; n < NT; ++n) load_frag_n(std::integral_constant<int, 0>{}, conv, chunk, j, ks, n);
  };
  // one tap of one chunk: 2 k-steps of 16.  par: the buffer this tap reads (DEEP), else 0.  Column tiles outermost: without DEEP the
  // fragments of a column tile are re-requested for the next iteration (nconv, nchunk, nj) as soon as ITS MFMAs have been issued
  auto mma_tap = [&](auto par, const float* a_base, int nconv, int nchunk, int nj) {
    constexpr int P = decltype(par)::value;
    if constexpr (DEEP) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int n = 0; n < NT; ++n) load_frag_n(std::integral_constant<int, 1 - P>{}, nconv, nchunk, nj, ks, n);
      __builtin_amdgcn_sched_barrier(0);
    }
    // TRANSPOSED product D^T = W . X^T: the weight fragment is the A operand (rows = output channels), the activation rows
    // are B (columns = positions) -- the operand images are the same either way and so are the bits (conv_gemm's order of terms
    // and k-steps is kept).  An accumulator lane then holds ONE position and, per register quad, four consecutive channels.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (X3) {
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
            if constexpr (SPLIT) {  // x_lo w_hi, x_hi w_lo, x_hi w_hi
              const bf16x8 bl = __builtin_bit_cast(bf16x8, bfr[P][ks][n][1]);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al[m], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah[m], acc[m][n], 0, 0, 0);
            }
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah[m], acc[m][n], 0, 0, 0);
          }
          if constexpr (!DEEP) {
            // keep the request right behind the MFMAs that free its registers (left alone, hipcc sinks all requests to the end
            // of the iteration, a few cycles before the next one waits for them)
            __builtin_amdgcn_sched_barrier(0);
            load_frag_n(std::integral_constant<int, 0>{}, nconv, nchunk, nj, ks, n);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
        // exact fp32: lane (li, lh) holds k = 8 ks + 4 lh + r of its row for both operands; MFMA r consumes component r
        float4 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + ks * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 bf = bfr[P][ks][n][0];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.x, af[m].x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.y, af[m].y, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.z, af[m].z, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf.w, af[m].w, acc[m][n], 0, 0, 0);
          }
          if constexpr (!DEEP) {
            __builtin_amdgcn_sched_barrier(0);
            load_frag_n(std::integral_constant<int, 0>{}, nconv, nchunk, nj, ks, n);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  };
  // the KW taps of one (conv, chunk) run, KW odd.  par0: parity of its first tap (DEEP).  After the last tap the fragment order continues
  // with (nconv_end, nchunk_end, tap 0)
  auto run_taps = [&](auto par0, const float* a0, int row_step, int conv, int c, int nconv_end, int nchunk_end) {
    constexpr int P0 = DEEP ? decltype(par0)::value : 0;
    constexpr int P1 = DEEP ? 1 - P0 : 0;
    int j = 0;
    for (; j + 2 < KW; j += 2) {
      mma_tap(std::integral_constant<int, P0>{}, a0 + j * row_step, conv, c, j + 1);
      mma_tap(std::integral_constant<int, P1>{}, a0 + (j + 1) * row_step, conv, c, j + 2);
    }
    mma_tap(std::integral_constant<int, P0>{}, a0 + j * row_step, nconv_end, nchunk_end, 0);  // j == KW - 1
  };

  // ---- epilogue 1: intermediate = lrelu(acc + b1) (zero outside [0, T)), split, into LDS as conv2's operand image.
  // Accumulator layout of the transposed product: lane (li, lh) = position li of the 32-position tile, register 4 q + i = channel
  // 8 q + 4 lh + i of the 32-channel tile: four consecutive channels of one position = 8 bytes of the hi half and 8 of the lo half
  // of that position's image row (two ds_write_b64, packed v_cvt_pk_bf16_f32).
  auto epilogue1 = [&](int tile) {
    const int i0 = tile * BMO - pad2;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int chunk = wn * NT + n;
      float4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(p.b1 + wn * WN + n * 32 + 8 * q + 4 * lh);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = wm * WM + m * 32 + li;
        const int g = i0 + row;
        const bool ok = g >= 0 && g < p.T;
        float* dst = smem + (chunk * irows + row) * LDK + 2 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4] = {acc[m][n][4 * q] + bq[q].x, acc[m][n][4 * q + 1] + bq[q].y, acc[m][n][4 * q + 2] + bq[q].z, acc[m][n][4 * q + 3] + bq[q].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(v[i], v[i] * p.slope);
            v[i] = ok ? v[i] : 0.f;
            acc[m][n][4 * q + i] = 0.f;
          }
          if constexpr (!X3) {  // fp32 image row: the four channels as they are
            *reinterpret_cast<float4*>(smem + (chunk * irows + row) * LDK + 8 * q + 4 * lh) = make_float4(v[0], v[1], v[2], v[3]);
            continue;
          }
          uint2 hi;
          hi.x = pack_bf16(v[0], v[1]);
          hi.y = pack_bf16(v[2], v[3]);
          *reinterpret_cast<uint2*>(dst + 4 * q) = hi;
          if constexpr (SPLIT) {
            const float h0 = __builtin_bit_cast(float, hi.x << 16), h1 = __builtin_bit_cast(float, hi.x & 0xffff0000u);
            const float h2 = __builtin_bit_cast(float, hi.y << 16), h3 = __builtin_bit_cast(float, hi.y & 0xffff0000u);
            uint2 lo;
            lo.x = pack_bf16(v[0] - h0, v[1] - h1);
            lo.y = pack_bf16(v[2] - h2, v[3] - h3);
            *reinterpret_cast<uint2*>(dst + 16 + 4 * q) = lo;
          }
        }
      }
    }
  };

  // ---- epilogue 2: out = acc + b2 + x (+ out, / div).  Global memory wants whole rows per wave instruction (16 lanes x 16 B = one
  // 256-byte row segment), the accumulators hold a position per lane: each 32-position block goes through a wave-private LDS patch
  // [32 positions][WN channels] -- written as one ds_write_b128 per register quad, read back with the lanes along the channels.
  // (Straight float4 I/O from the accumulator layout -- 32 B per row and instruction -- was measured 8-27 % slower per pair.)
  constexpr int ELD = WN + 4;                          // patch row stride (floats): a 16-lane group of b128 stores covers all 64 banks once
  static_assert(NWAVE * 32 * ELD <= NCH * BMI * LDK, "epilogue patches must fit in the region (>= the intermediate: NCH x (BMI + KW - 1) rows)");
  constexpr int LPR = WN / 4, RPP = 64 / LPR, PASSES = 32 / RPP;
  const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
  const int ecol = wn * WN + pc4;
  const float4 bias2 = *reinterpret_cast<const float4*>(p.b2 + ecol);
  auto epilogue2 = [&](int tile) {
    const int o0 = tile * BMO;
    const int t_end = min(o0 + BMO, p.T);  // rows this tile owns (tile rows >= BMO are the recomputed overlap)
    float* patch = smem + wave * (32 * ELD);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int tb = o0 + wm * WM + m * 32;   // first output row of this 32-position block
      float4 resv[PASSES];  // residual rows of the block, requested before its transposes
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int t = min(tb + ps * RPP + prow, p.T - 1);
        resv[ps] = *reinterpret_cast<const float4*>(x_b + (long long)t * C + ecol);
      }
      float4 accv[PASSES];  // accumulate mode: what the output rows hold now
      if constexpr (ACCUM) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int t = min(tb + ps * RPP + prow, p.T - 1);
          accv[ps] = *reinterpret_cast<const float4*>(out_b + (long long)t * C + ecol);
        }
      }
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(patch + li * ELD + n * 32 + 8 * q + 4 * lh) =
              make_float4(acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]);
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int row = ps * RPP + prow;
        const int t = tb + row;
        float4 v = *reinterpret_cast<const float4*>(patch + row * ELD + pc4);
        v.x += bias2.x; v.y += bias2.y; v.z += bias2.z; v.w += bias2.w;
        const float4 rv = resv[ps];
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        if constexpr (ACCUM) {
          const float4 ov = accv[ps];
          v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
          if (p.out_div != 1.0f) {
            v.x = v.x / p.out_div; v.y = v.y / p.out_div; v.z = v.z / p.out_div; v.w = v.w / p.out_div;
          }
        }
        // streaming store: this kernel never reads the tile back, and keeping it out of the way leaves the x rows (slab halo of the
        // neighbour tile, residual re-read) in L2
        if (t < t_end && wm * WM + m * 32 + row < BMO) {
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          const f32x4_t nv = {v.x, v.y, v.z, v.w};
          __builtin_nontemporal_store(nv, reinterpret_cast<f32x4_t*>(out_b + (long long)t * C + ecol));
        }
      }
    }
  };

  // ---- main
  load_a(tile, 0);
  store_a();
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) load_frag(0, 0, 0, ks);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing in flight at the loop heads: their waits stay counted (see conv_gemm.hip)

  // conv1: K = chunks of x, slab re-staged per chunk
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) load_a(tile, c + 1);
    __syncthreads();  // slab of chunk c visible
    const bool lastc = c + 1 == NCH;
    // DEEP implies NCH == 1, so the run's parity is known at compile time
    run_taps(std::integral_constant<int, 0>{}, smem + (wm * WM + li) * LDK + lh * 4, dil * LDK, 0, c, lastc ? 1 : 0, lastc ? 0 : c + 1);
    if (c + 1 < NCH) {
      __syncthreads();  // every wave is done with the slab of chunk c
      store_a();
    }
  }
  __syncthreads();  // x slab dead
  epilogue1(tile);
  __syncthreads();  // intermediate visible
  // conv2: K = chunks of the intermediate, dilation 1 (the requests behind its very last tap re-read its first fragments: unused)
  for (int c = 0; c < NCH; ++c) {
    const bool lastc = c + 1 == NCH;
    run_taps(std::integral_constant<int, 1>{}, smem + (c * irows + wm * WM + li) * LDK + lh * 4, LDK, 1, c, 1, lastc ? 0 : c + 1);
  }
  __syncthreads();  // intermediate dead: the region now carries the epilogue's patches
  epilogue2(tile);
}

template <int BMI, int C, int WM, int WN, int MODE>
const char* launch_pair_cfg(const PairParams& p, hipStream_t s) {
  const int halo1 = p.dil * (p.KW - 1);
  const int BMO = BMI - (p.KW - 1);
  const size_t words = (size_t)std::max((BMI + halo1) * LDK, (C / 32) * (BMI + p.KW - 1) * LDK);
  const size_t lds = words * sizeof(float);
  if (lds > PAIR_MAX_LDS) return "resblock_pair: LDS region exceeds the CU's 160 KiB";
  const int mtiles = (p.T + BMO - 1) / BMO;
  dim3 grid((mtiles + 7) / 8 * 8, p.B);  // a multiple of 8 (see the XCD-aware order in the kernel); extra blocks exit
  RowMap rm;
  if (p.act_rows && p.act_rows_host && p.B <= ROWMAP_MAX) {  // ragged: only the blocks of tiles that exist (at most 7 spare per utterance)
    rm.n = p.B;
    rm.identity();
    rm.cum[0] = 0;
    for (int b = 0; b < p.B; ++b) {
      const int mt = (std::min(std::max(p.act_rows_host[b], 0), p.T) + BMO - 1) / BMO;
      rm.cum[b + 1] = rm.cum[b] + (mt + 7) / 8 * 8;
    }
    if (rm.cum[p.B] == 0) return nullptr;
    grid = dim3(rm.cum[p.B]);
  }
  constexpr int NTHR = pair_threads<BMI, C, WM, WN>();
  if (p.accumulate)
    hipLaunchKernelGGL((resblock_pair_kernel<BMI, C, WM, WN, MODE, true>), grid, dim3(NTHR), lds, s, p, rm);
  else
    hipLaunchKernelGGL((resblock_pair_kernel<BMI, C, WM, WN, MODE, false>), grid, dim3(NTHR), lds, s, p, rm);
  return hipGetLastError() == hipSuccess ? nullptr : "resblock_pair: launch failed";
}

template <int BMI, int C, int WM, int WN>
const char* launch_pair_mode(const PairParams& p, hipStream_t s) {
  static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the opt-in, once per instantiation
  if (!attr_done) {
    const int cap = C == 256 ? (int)PAIR_MAX_LDS : 80 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<BMI, C, WM, WN, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<BMI, C, WM, WN, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<BMI, C, WM, WN, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<BMI, C, WM, WN, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<BMI, C, WM, WN, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<BMI, C, WM, WN, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    attr_done = true;
  }
  if (p.mode == 0) return launch_pair_cfg<BMI, C, WM, WN, 0>(p, s);
  return p.mode == 1 ? launch_pair_cfg<BMI, C, WM, WN, 1>(p, s) : launch_pair_cfg<BMI, C, WM, WN, 2>(p, s);
}

}  // namespace

bool resblock_pair_supported(int C, int KW, int dil) {
  if (!((C == 32 || C == 64 || C == 128 || C == 256) && (KW & 1) && KW >= 3 && KW <= 15 && dil >= 1 && dil * (KW - 1) <= MAX_HALO1)) return false;
  // 256 channels: the intermediate [8][128 + KW - 1] x 144 B must fit the CU's LDS
  return C != 256 || (size_t)(C / 32) * (128 + KW - 1) * LDK * 4 <= PAIR_MAX_LDS;
}

double resblock_pair_flops(const PairParams& p) { return 2.0 * 2.0 * p.B * (double)p.T * p.act_frac * p.C * p.KW * p.C; }

double resblock_pair_bytes(const PairParams& p) {
  return 4.0 * ((double)p.B * p.T * p.act_frac * p.C * (2.0 + (p.accumulate ? 1 : 0)) + 2.0 * p.C * p.KW * p.C);
}

const char* launch_resblock_pair(const PairParams& p, hipStream_t s) {
  if (!p.x || !p.wfrag || !p.b1 || !p.b2 || !p.out) return "resblock_pair: null pointer";
  if (p.B <= 0 || p.T <= 0) return "resblock_pair: bad dims";
  if (!resblock_pair_supported(p.C, p.KW, p.dil)) return "resblock_pair: unsupported channels / kernel / dilation";
  if (p.mode < 0 || p.mode > 2) return "resblock_pair: mode must be 0 (fp32), 1 (bf16x3) or 2 (bf16)";
  if (p.slope < 0.f || p.slope > 1.f) return "resblock_pair: slope must lie in [0, 1]";
  if (p.out_div != 1.0f && !p.accumulate) return "resblock_pair: out_div needs accumulate";
  if ((((uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.wfrag | (uintptr_t)p.b1 | (uintptr_t)p.b2) & 15) || (p.x_bs & 3) || (p.out_bs & 3))
    return "resblock_pair: pointers must be 16-byte aligned";
  if ((long long)p.T * p.C * 4 >= (1LL << 31)) return "resblock_pair: one utterance must stay below 2 GiB (32-bit buffer offsets)";
  if (p.x == p.out) return "resblock_pair: in-place is not possible (tiles read their neighbours' rows)";
  if (p.C == 32) return launch_pair_mode<256, 32, 64, 32>(p, s);
  if (p.C == 64) return launch_pair_mode<256, 64, 64, 64>(p, s);
  if (p.C == 256) return launch_pair_mode<128, 256, 64, 64>(p, s);
  return launch_pair_mode<128, 128, 64, 64>(p, s);
}

}  // namespace e2etts
