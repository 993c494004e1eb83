// conv_ksplit.hip -- the convolutions of the LATENCY path: one kernel, conv_rows_kernel, in which every wavefront owns its output tile,
// its slab rows (private, double-buffered LDS) and a ring of weight fragments, and no workgroup barrier exists.  Two uses:
//
// (1) launch_conv_ksplit (SPLITK): "same" 1-D convolution / Linear in exact fp32 for the PHONEME-LEVEL layers -- the encoder's FFN
// convolutions (reference U/blocks/transformer.py:289-297) and the duration / pitch / energy predictors (U/layers.py:410-420, 491-505)
// -- with the K dimension split four ways INSIDE the workgroup.
//   These layers see B x L phonemes, not B x T frames: 128 rows at B = 1, 4 096 at B = 32 -- 1.3 % of the step's FLOPs -- but each
// output tile sums over K = KW x Cin up to 3 456 (the FFN's k = 9 convolution), and in conv_gemm.hip one wave walks that whole K as ONE
// chain of dependent v_mfma_f32_32x32x2_f32: 1 728 MFMAs x 64 cycles = 46 us per launch however few rows there are.  Here the four waves
// of a workgroup share ONE 32 x 32 (or 32 x 64) output tile and take the 32-channel chunks of Cin round-robin (wave w: chunks w, w + 4,
// ...): a chain a quarter as long, four times the waves.  The partial sums meet in LDS and are added in a FIXED order,
// ((P0 + P1) + (P2 + P3)), before bias / activation / residual.
//   That order differs from conv_gemm's single chain, so this form serves these layers at EVERY batch size: an utterance's durations,
// pitch and energy buckets and encoder output do not depend on what it is batched with (tests/test_gpu_parity.py: B = 1 against the
// same utterance inside B = 32, bit for bit), and the discrete outputs keep matching the reference on every fixture.
//
// (2) launch_conv_rows: conv_gemm's OWN arithmetic, MFMA for MFMA, for launches with few rows (the decoder and postnet convolutions of
// the B = 1 path) -- see the comment above the kernel.  Same bits as conv_gemm, so the engine picks per launch.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int LDK = 36;        // LDS row stride (floats): 32 channels + 4 pad
constexpr int KS_MAX_HALO = 16;  // dil * (KW - 1) supported (k = 9: 8)

// ---- conv_rows: the few-rows launches of the FRAME-level layers (the decoder's FFT blocks, mel_linear, the postnet at small batches:
// the B = 1 latency path) -- conv_gemm's 64 x 64 tile otherwise.  Same arithmetic as conv_gemm, MFMA for MFMA (chunk-major, tap-minor,
// one accumulator chain per 32 x 32 tile; bf16x3: lo x hi, hi x lo, hi x hi), so the two give the same bits and the choice is free per
// launch.  What differs is who waits for whom.  conv_gemm's workgroup shares a slab and meets at a barrier per chunk, and its waves
// request weight fragments one iteration ahead: right when several workgroups per CU overlap, but at B = 1 there is ONE wave per SIMD
// and every weight is read once, from HBM -- each of the 108 iterations of the decoder's k = 9 convolution waited out most of a DRAM
// round trip (63 us for 5.4 GFLOP).  Here a wavefront owns its 32 x 32 (x NT) tile outright: private slab rows in LDS (double buffered,
// the next chunk's rows requested a chunk ahead), no workgroup barrier anywhere, and a ring of D units of weight fragments in flight.
// The four wavefronts of a workgroup are stacked on the rows, so they ask for the same fragments at about the same time.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {  // conv_gemm.hip's: x = hi + lo, two bf16
  hi.x = pack_bf16(v.x, v.y);
  hi.y = pack_bf16(v.z, v.w);
  const float hx = __builtin_bit_cast(float, hi.x << 16), hy = __builtin_bit_cast(float, hi.x & 0xffff0000u);
  const float hz = __builtin_bit_cast(float, hi.y << 16), hw = __builtin_bit_cast(float, hi.y & 0xffff0000u);
  lo.x = pack_bf16(v.x - hx, v.y - hy);
  lo.y = pack_bf16(v.z - hz, v.w - hw);
}

// SPLITK (conv_ksplit, fp32 only): the four wavefronts share ONE 32-row tile and take the 32-channel chunks round-robin (wave w: chunks
// w, w + 4, ...); their partial sums meet in LDS and wave 0 adds them as ((P0 + P1) + (P2 + P3)) -- see the head of this file.
template <int NT, int MODE, int D, bool SPLITK>
__global__ __launch_bounds__(256) void conv_rows_kernel(const ConvParams p, const int rg, const int ct, const RowMap rm) {
  static_assert(!SPLITK || MODE == 0, "the K-split form serves the exact-fp32 phoneme-level layers");
  constexpr bool X3 = MODE != 0;
  constexpr bool SPLIT = MODE == 1;
  constexpr int KSN = X3 ? 2 : 4;  // fragment groups per 32-channel chunk: 2 k-steps of 16 (bf16 MFMA) or 4 x (4 MFMAs of k = 2)
  constexpr int HL = X3 ? 2 : 1;
  constexpr int SROWS = 32 + KS_MAX_HALO;
  constexpr int NPASS = SROWS / 8;
  __shared__ __attribute__((aligned(16))) float slab[4][2][SROWS * LDK];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  // column tile fastest: with ct a multiple of 8 the workgroups of one XCD (id % 8) share an eighth of the weights
  const int cy = blockIdx.x % ct, g = blockIdx.x / ct;
  int b, gl;  // utterance, row unit inside it: compact grid of a ragged batch (kernels.h: RowMap) or rg units per utterance
  if (rm.n > 0) {
    if (!rowmap_find(rm, g, b, gl)) return;
  } else {
    b = g / rg;
    gl = g - b * rg;
  }
  const int t0 = SPLITK ? gl * 32 : gl * 128 + wave * 32;   // units: row tiles of 32 (SPLITK) or groups of 128
  const int t_act = p.act_rows ? min(p.act_rows[b], p.T) : p.T;
  if (t0 >= t_act) return;   // !SPLITK: no workgroup barrier, a wavefront without rows simply leaves; SPLITK: uniform for the workgroup
  const int n0 = cy * (32 * NT);
  const int halo = p.dil * (p.KW - 1);
  const int srows = 32 + halo;
  const int nchunk = (p.Cin + 31) / 32;
  const int ntile32 = (p.Cout + 31) / 32;
  const int nmy = SPLITK ? (nchunk - wave + 3) / 4 : nchunk;   // this wavefront's chunks: cc(0), cc(1), ...
  auto cc = [&](int c) __attribute__((always_inline)) { return SPLITK ? wave + 4 * c : c; };
  const int nu = nmy * p.KW;
  const float* in_b = p.in + (long long)b * p.in_bs;
  const __amdgpu_buffer_rsrc_t in_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, (int)((long long)p.T * p.in_ld * 4), 0x00020000);
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

  // ---- slab staging: 8 lanes per row (4 channels each), 8 rows per pass; loads unconditional and clamped, zeros applied on the way to LDS
  const int srow = lane >> 3, sc4 = (lane & 7) * 4;
  float4 sreg[NPASS];
  auto stage_load = [&](int c) __attribute__((always_inline)) {
    const int ch = min(c * 32 + sc4, p.Cin - 4);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int t = min(max(t0 - p.pad + ps * 8 + srow, 0), p.T - 1);
      sreg[ps] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (t * p.in_ld + ch) * 4, 0, 0));
    }
  };
  auto stage_store = [&](int c, float* dst) __attribute__((always_inline)) {
    const bool ch_ok = c * 32 + sc4 < p.Cin;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int r = ps * 8 + srow;
      const int t = t0 - p.pad + r;
      const bool ok = ch_ok && t >= 0 && t < p.T;
      const float4 v = make_float4(ok ? sreg[ps].x : 0.f, ok ? sreg[ps].y : 0.f, ok ? sreg[ps].z : 0.f, ok ? sreg[ps].w : 0.f);
      if (r < srows) {
        if constexpr (X3) {
          uint2 hi, lo;
          split4(v, hi, lo);
          *reinterpret_cast<uint2*>(dst + r * LDK + (sc4 >> 1)) = hi;
          *reinterpret_cast<uint2*>(dst + r * LDK + 16 + (sc4 >> 1)) = lo;
        } else {
          *reinterpret_cast<float4*>(dst + r * LDK + sc4) = v;
        }
      }
    }
  };

  // ---- weight fragments: ring of D units (unit u = chunk * KW + tap lives in slot u % D)
  float4 bfr[D][KSN][NT][HL];
  auto load_frag = [&](auto slot, int c, int j) __attribute__((always_inline)) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int hl = 0; hl < HL; ++hl) {
          const int soff = ((((fnt[n] * p.KW + j) * nchunk + c) * KSN + ks) * HL + hl) * 1024;
          bfr[S][ks][n][hl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wf_rsrc, lane * 16, soff, 0));
        }
  };

  float* const slab0 = slab[wave][0];
  int buf = 0;   // which half of this wave's double buffer holds the current chunk (an index, not a swapped pointer: that went via scratch)
  stage_load(cc(0));
  int pc = 0, pj = 0;  // (chunk, tap) of the unit whose fragments are requested next; wraps (requests past the end are never used)
  auto advance_p = [&]() __attribute__((always_inline)) {
    if (++pj == p.KW) {
      pj = 0;
      if (++pc == nmy) pc = 0;
    }
  };
  static_for<D>([&](auto slot) __attribute__((always_inline)) {
    load_frag(slot, cc(pc), pj);
    advance_p();
  });
  stage_store(cc(0), slab0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS written by other lanes of THIS wave: order the reads behind the writes
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  int c = 0, j = 0;
  auto unit = [&](auto slot, const int c, const int j, const int buf) __attribute__((always_inline)) {
    constexpr int S = decltype(slot)::value;
    if (j == 0 && c + 1 < nmy) stage_load(cc(c + 1));   // the next chunk's rows: in registers until this chunk's last tap
    const float* a_base = slab0 + buf * (SROWS * LDK) + (li + j * p.dil) * LDK + lh * 4;
    if constexpr (X3) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + ks * 8));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + 16 + ks * 8));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const bf16x8 bh = __builtin_bit_cast(bf16x8, bfr[S][ks][n][0]);
          const bf16x8 bl = __builtin_bit_cast(bf16x8, bfr[S][ks][n][1]);
          if constexpr (SPLIT) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[n], 0, 0, 0);
          }
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(a_base + q * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const float4 w = bfr[S][q][n][0];
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc[n], 0, 0, 0);
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc[n], 0, 0, 0);
        }
      }
    }
    load_frag(slot, cc(pc), pj);   // this slot is consumed: the unit D ahead
    advance_p();
  };
  // (spelled out, with the counters updated and the wavefront fences issued HERE rather than inside a lambda: with both inside one,
  // hipcc kept c and j in scratch memory, and every scratch load waits for vmcnt(0) -- the whole fragment ring)
#define E2ETTS_UNIT(S)                                                    \
  if constexpr (S < D) {                                                  \
    if (u0 + S < nu) {                                                    \
      unit(std::integral_constant<int, S>{}, c, j, buf);                  \
      if (j == p.KW - 1) {                                                \
        if (c + 1 < nmy) {                                                \
          stage_store(cc(c + 1), slab0 + (buf ^ 1) * (SROWS * LDK));      \
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          \
          __builtin_amdgcn_wave_barrier();                                \
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          \
          buf ^= 1;                                                       \
        }                                                                 \
        j = 0;                                                            \
        ++c;                                                              \
      } else {                                                            \
        ++j;                                                              \
      }                                                                   \
    }                                                                     \
  }
  for (int u0 = 0; u0 < nu; u0 += D) {
    E2ETTS_UNIT(0) E2ETTS_UNIT(1) E2ETTS_UNIT(2) E2ETTS_UNIT(3) E2ETTS_UNIT(4) E2ETTS_UNIT(5) E2ETTS_UNIT(6) E2ETTS_UNIT(7)
  }
#undef E2ETTS_UNIT
  static_assert(D <= 8, "ring depth");

  const int len = p.lens ? p.lens[b] : p.T;
  float* out_b = p.out + (long long)b * p.out_bs;
  const float* res_b = p.res ? p.res + (long long)b * p.res_bs : nullptr;
  if constexpr (SPLITK) {
    // ---- ((P0 + P1) + (P2 + P3)): waves 1..3 leave their partial sums in their own (now idle) slab, wave 0 adds them in that order
    static_assert(2 * SROWS * LDK >= NT * 16 * 64, "a wave's slab must hold its partial tile");
    if (wave > 0) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab0[(n * 16 + r) * 64 + lane] = acc[n][r];
    }
    __syncthreads();
    if (wave != 0) return;
    const float* r1 = slab[1][0];
    const float* r2 = slab[2][0];
    const float* r3 = slab[3][0];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = n0 + n * 32 + li;
      const bool col_ok = col < p.Cout;
      const float bias = (p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float p01 = acc[n][r] + r1[(n * 16 + r) * 64 + lane];
        const float p23 = r2[(n * 16 + r) * 64 + lane] + r3[(n * 16 + r) * 64 + lane];
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
    return;
  }
  // ---- epilogue: conv_gemm's order and formulas (bias, max(v, v * slope) for none / ReLU / leaky ReLU, tanh, swish, residual, mask)
  const float eslope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LRELU ? p.act_slope : 1.f);
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int col = n0 + n * 32 + li;
    const bool col_ok = col < p.Cout;
    const float bias = (p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = acc[n][r] + bias;
      v = fmaxf(v, v * eslope);
      if (p.act == ACT_TANH) v = tanhf(v);
      if (p.act == ACT_SWISH) v *= 1.0f / (1.0f + expf(-v));
      if (t < p.T && col_ok) {
        if (res_b) v += res_b[(long long)t * p.res_ld + col];
        if (t >= len) v = 0.f;
        out_b[(long long)t * p.out_ld + col] = v;
      }
    }
  }
}

}  // namespace

// Row units (of `rows` output rows each) of the launch: the padded count per utterance, or -- ragged batch whose lengths the host knows --
// only the units that have rows to compute, with the table the kernel finds its utterance in (kernels.h: RowMap).
static long long ragged_units(const ConvParams& p, int rows, int padded_per_utt, RowMap& rm) {
  rm.n = 0;
  if (!(p.act_rows && p.act_rows_host && p.B <= ROWMAP_MAX)) return (long long)padded_per_utt * p.B;
  rm.n = p.B;
  rm.identity();
  rm.cum[0] = 0;
  for (int b = 0; b < p.B; ++b) rm.cum[b + 1] = rm.cum[b] + (std::min(std::max(p.act_rows_host[b], 0), p.T) + rows - 1) / rows;
  return rm.cum[p.B];
}

bool conv_ksplit_supported(const ConvParams& p) {
  return p.wfrag && p.x3 == 0 && !p.accumulate && p.in_slope == 1.0f && p.zero_tap_split == 0 && p.KW >= 1 && p.dil >= 1 &&
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
  const int ct = wide ? (p.Cout + 63) / 64 : (p.Cout + 31) / 32;
  RowMap rm;
  const long long units = ragged_units(p, 32, rt, rm);
  if (units == 0) return nullptr;
  const long long nwg = (long long)ct * units;
  if (nwg >= (1LL << 31)) return "conv_ksplit: grid too large";
  if (wide) hipLaunchKernelGGL((conv_rows_kernel<2, 0, 3, true>), dim3((unsigned)nwg), dim3(256), 0, s, p, rt, ct, rm);
  else hipLaunchKernelGGL((conv_rows_kernel<1, 0, 3, true>), dim3((unsigned)nwg), dim3(256), 0, s, p, rt, ct, rm);
  return hipGetLastError() == hipSuccess ? nullptr : "conv_ksplit: launch failed";
}

bool conv_rows_supported(const ConvParams& p) {
  return p.wfrag && p.x3 >= 0 && p.x3 <= 2 && !p.accumulate && p.out_div == 1.0f && p.in_slope == 1.0f && p.zero_tap_split == 0 && p.KW >= 1 &&
         p.dil >= 1 && p.dil * (p.KW - 1) <= KS_MAX_HALO && p.pad >= 0 && p.pad <= p.dil * (p.KW - 1) && (p.Cin % 4) == 0 && p.Cin >= 4 &&
         (p.in_ld % 4) == 0;
}

const char* launch_conv_rows(const ConvParams& p, hipStream_t s) {
  if (!p.in || !p.wfrag || !p.out) return "conv_rows: null pointer";
  if (p.B <= 0 || p.T <= 0 || p.Cin <= 0 || p.Cout <= 0) return "conv_rows: bad dims";
  if (!conv_rows_supported(p)) return "conv_rows: unsupported launch (fragment-order weights, dil (KW - 1) <= 16, no input activation, no accumulate)";
  if (((uintptr_t)p.in & 15) || (p.in_bs & 3)) return "conv_rows: input must be 16-byte aligned";
  if (p.in_ld < p.Cin || p.out_ld < p.Cout || (p.res && p.res_ld < p.Cout)) return "conv_rows: row stride < channels";
  if ((long long)p.T * p.in_ld * 4 >= (1LL << 31)) return "conv_rows: one utterance must stay below 2 GiB (32-bit buffer offsets)";
  const int rg = (p.T + 127) / 128;
  // 64-column tiles halve the slab traffic and LDS reads per MFMA; 32-column tiles when that would leave CUs without a workgroup
  const bool wide = p.Cout > 32 && (long long)rg * ((p.Cout + 63) / 64) * p.B >= 256;
  const int ct = wide ? (p.Cout + 63) / 64 : (p.Cout + 31) / 32;
  RowMap rm;
  const long long units = ragged_units(p, 128, rg, rm);
  if (units == 0) return nullptr;
  const long long nwg = (long long)ct * units;
  if (nwg >= (1LL << 31)) return "conv_rows: grid too large";
  const dim3 grid((unsigned)nwg), block(256);
  // ring depth: a unit is 16 MFMAs of 64 cycles in fp32, 6 (bf16x3) or 2 (bf16) of 32 in the bf16 modes
  if (p.x3 == 0) {
    if (wide) hipLaunchKernelGGL((conv_rows_kernel<2, 0, 3, false>), grid, block, 0, s, p, rg, ct, rm);
    else hipLaunchKernelGGL((conv_rows_kernel<1, 0, 3, false>), grid, block, 0, s, p, rg, ct, rm);
  } else if (p.x3 == 1) {
    if (wide) hipLaunchKernelGGL((conv_rows_kernel<2, 1, 4, false>), grid, block, 0, s, p, rg, ct, rm);
    else hipLaunchKernelGGL((conv_rows_kernel<1, 1, 6, false>), grid, block, 0, s, p, rg, ct, rm);
  } else {
    if (wide) hipLaunchKernelGGL((conv_rows_kernel<2, 2, 4, false>), grid, block, 0, s, p, rg, ct, rm);
    else hipLaunchKernelGGL((conv_rows_kernel<1, 2, 6, false>), grid, block, 0, s, p, rg, ct, rm);
  }
  return hipGetLastError() == hipSuccess ? nullptr : "conv_rows: launch failed";
}

}  // namespace e2etts
