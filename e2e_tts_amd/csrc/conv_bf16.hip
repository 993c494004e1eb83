// conv_bf16.hip -- the vocoder's convolutions in PLAIN bf16 (engine precision "bf16": BASELINE config 5, the 48 kHz long-form stream).
//
// Replaces, for that arithmetic mode: every Conv1d / ConvTranspose1d of HifiGan.forward (reference V/generator.py:37-53) and the two
// convolutions of a ResBlock1 pair (V/layers.py:33-40) where they run as separate launches.
//
// Why a kernel of its own.  conv_gemm.hip / resblock_pair.hip were shaped around the split-precision mode: three MFMAs per pair of
// operand fragments, a 32-channel slab re-staged per chunk behind two workgroup barriers.  In plain bf16 the same loop has a THIRD of the
// MFMA work between the same barriers, and each weight fragment (1 KiB through the CU's 64 B / clk L1) feeds a third of the MFMAs: the
// PMC pass of the 48 kHz stream (profiles/r4/pmc_c5_*.md) shows waves parked 50-66 % of their cycles and the matrix pipe busy 0.04-0.16
// at 542-frame windows (0.30-0.38 in one call).  Worst were the few-row launches of the 256-channel stage (4 336 rows): 272 workgroups
// walking an 88-link chain (8 chunks x 11 taps) of "barrier, wait for fragments, 2 MFMAs" -- 36 us for 2.4 us of matrix work.
//
// This kernel: a workgroup (4 wavefronts) owns a tile of BM rows x BN columns and stages the WHOLE slab -- (BM + halo) rows x all Cin
// channels -- into LDS once, as bf16 (leaky ReLU + round-to-nearest-even while staging an fp32 input: the same values conv_gemm's mode 2
// forms per chunk; a bf16 input, which the producing launch already rounded, is copied): ONE workgroup barrier, then every wavefront
// walks the K dimension on its own, (chunk, tap) unit by unit, with the weight fragments of the next D units in flight from L2 (ring in
// registers, requested right behind the MFMAs that free them).  Order of terms per output element: chunk-major, then tap, then k-step,
// one fp32 accumulator chain -- conv_gemm's order, so the results are bit-identical to it (tools/bconv_bench checks that, shape by
// shape) and the engine may choose per launch.  MFMA orientation: D^T = W . X^T (weight fragment = A operand, activation rows = B), as in
// resblock_pair.hip: an accumulator lane holds one position and, per register quad, four consecutive channels.
//
// LDS row = Cin (padded to 32) bf16 + 16 bytes: the row stride in words is 4 mod 16, so the 16 rows a ds_read_b128 lane group touches
// cover all 64 banks exactly once, for every tap and chunk.
//
// Weights: launch_bf16_image's order [Cout / 32][chunk][tap slot][k-step][lane][8 bf16] -- the hi halves of the split-precision image,
// with the (chunk, tap) units of a 32-column tile contiguous: 2 KiB per unit, one scalar offset step per unit.  A polyphase upsampler
// (ConvTranspose1d as a 3-tap convolution whose columns < tap_split never use tap 2 and the others never tap 0: packer.polyphase_upsampler)
// keeps only the two live taps of each tile.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifdef E2ETTS_BC_DIAG   // tools/bconv_bench diagnostic build only: s_memtime stamps per phase, summed over the workgroups' wave 0
__device__ unsigned long long g_bc_diag[8];
#define BC_STAMP(var)                                                                          \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
void conv_bf16_read_diag(unsigned long long* out) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc_diag), sizeof(g_bc_diag));
  unsigned long long z[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bc_diag), z, sizeof(z));
}
#else
#define BC_STAMP(var) do {} while (0)
#endif

namespace {

constexpr int BC_MAX_HALO = 64;   // dil * (KW - 1): 50 for k = 11, dilation 5

__device__ __forceinline__ unsigned bc_pack(float a, float b) {
  const bf16x2 r = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round to nearest even: conv_gemm's split4 hi half
  return __builtin_bit_cast(unsigned, r);
}

// The instruction order of one (chunk, tap) unit, handed to the scheduler as a pipeline (sched_group_barrier): M MFMAs, each followed by
// its share of the unit's R LDS reads (the NEXT unit's activation fragments) and, behind the last MFMA of each k-step, that k-step's
// share of the L weight requests (the slot those MFMAs just freed).  An MFMA occupies the matrix pipe for 32 cycles and the wavefront can
// issue ~6 other instructions in its shadow; left in blocks between the units (reads | MFMAs | requests, as first written), the ~30
// non-MFMA instructions of a unit cost 100-130 cycles per unit on top of the MFMAs (s_memtime stamps: 368 cycles per 8-MFMA unit,
// 296 with the pipeline).
template <int I, int M, int R, int L>
__device__ __forceinline__ void bc_pipeline() {
  if constexpr (I < M) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        // one MFMA
    if constexpr (R >= M) __builtin_amdgcn_sched_group_barrier(0x100, R / M, 0);              // LDS reads
    else if constexpr ((I % (M / R)) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if constexpr (I == M / 2 - 1 || I == M - 1) __builtin_amdgcn_sched_group_barrier(0x020, L / 2, 0);   // weight requests of the k-step just done
    bc_pipeline<I + 1, M, R, L>();
  }
}

// Slab staging, by the whole workgroup (256 threads): rows [t_first, t_first + srows) of one utterance's [T, Cin] tensor -> LDS rows of
// RS bytes holding NCH x 32 bf16 channels; rows outside [0, T) and channels >= Cin are zero.  fp32 input: leaky ReLU (slope) and
// round-to-nearest-even on the way (conv_gemm's mode-2 staging, value for value); bf16 input: copied.  SB 16-byte LDS pieces per thread
// and batch -- all of a batch's loads are issued before the first is used, and SB is sized so that a slab is one or two batches (each
// batch is a round trip to L2 / HBM with the CU otherwise idle: 6-piece batches took 8 600 cycles for a 178-row x 256-channel slab).
// in_add / in_div (fp32 input only): further tensors summed into the input while staging -- x = (((in + a0) + a1) + a2) / div, the order and
// operations of accum_div_kernel (small_kernels.hip), whose launch and whose pass over the tensors this replaces.
template <bool IN_BF16, int SB, int NTHR = 256>
__device__ __forceinline__ void bc_stage(unsigned char* smem, const void* in_utt, const int T, const int Cin, const int NCH, const int RS,
                                         const int t_first, const int srows, const float slope, const int tid,
                                         const float* a0 = nullptr, const float* a1 = nullptr, const float* a2 = nullptr, const float div = 1.0f) {
  const int ppr = NCH * 4;             // 16-byte pieces (8 channels) per LDS row
  const int npieces = srows * ppr;
  constexpr int esz = IN_BF16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t in_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(in_utt), 0, (int)((long long)T * Cin * esz), 0x00020000);
  if constexpr (!IN_BF16) {
    if (a0) {   // the joined form: four pieces per thread and batch, every input's loads issued before the first is used
      constexpr int SJ = 4;
      const int bytes = (int)((long long)T * Cin * 4);
      const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a0), 0, bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a1 ? a1 : a0), 0, bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a2 ? a2 : a0), 0, bytes, 0x00020000);
      for (int base = tid; base < npieces; base += NTHR * SJ) {
        float4 v[SJ][2], w0[SJ][2], w1[SJ][2], w2[SJ][2];
        bool ok[SJ];
        int dst[SJ];
#pragma unroll
        for (int i = 0; i < SJ; ++i) {
          const int idx = base + i * NTHR;
          const int row = idx / ppr, pc = idx - row * ppr;
          const int t = t_first + row;
          ok[i] = idx < npieces && t >= 0 && t < T && pc * 8 < Cin;
          dst[i] = idx < npieces ? row * RS + pc * 16 : -1;
          const int off = (min(max(t, 0), T - 1) * Cin + min(pc * 8, Cin - 8)) * 4;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            v[i][h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 16 * h, 0, 0));
            w0[i][h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r0, off + 16 * h, 0, 0));
            if (a1) w1[i][h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r1, off + 16 * h, 0, 0));
            if (a2) w2[i][h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r2, off + 16 * h, 0, 0));
          }
        }
#pragma unroll
        for (int i = 0; i < SJ; ++i) {
          float4 x[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float4 a = v[i][h];
            a.x += w0[i][h].x; a.y += w0[i][h].y; a.z += w0[i][h].z; a.w += w0[i][h].w;
            if (a1) { a.x += w1[i][h].x; a.y += w1[i][h].y; a.z += w1[i][h].z; a.w += w1[i][h].w; }
            if (a2) { a.x += w2[i][h].x; a.y += w2[i][h].y; a.z += w2[i][h].z; a.w += w2[i][h].w; }
            if (div != 1.0f) { a.x = a.x / div; a.y = a.y / div; a.z = a.z / div; a.w = a.w / div; }
            a.x = fmaxf(a.x, a.x * slope); a.y = fmaxf(a.y, a.y * slope); a.z = fmaxf(a.z, a.z * slope); a.w = fmaxf(a.w, a.w * slope);
            x[h] = a;
          }
          uint4 o;
          o.x = bc_pack(x[0].x, x[0].y); o.y = bc_pack(x[0].z, x[0].w); o.z = bc_pack(x[1].x, x[1].y); o.w = bc_pack(x[1].z, x[1].w);
          if (!ok[i]) o = make_uint4(0, 0, 0, 0);
          if (dst[i] >= 0) *reinterpret_cast<uint4*>(smem + dst[i]) = o;
        }
      }
      return;
    }
  }
  for (int base = tid; base < npieces; base += NTHR * SB) {
    float4 ra[SB], rb[SB];
    bool ok[SB];
    int dst[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int idx = base + i * NTHR;
      const int row = idx / ppr, pc = idx - row * ppr;
      const int t = t_first + row;
      ok[i] = idx < npieces && t >= 0 && t < T && pc * 8 < Cin;
      dst[i] = idx < npieces ? row * RS + pc * 16 : -1;
      const int tc = min(max(t, 0), T - 1), cc = min(pc * 8, Cin - 8);
      if constexpr (IN_BF16) {
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (tc * Cin + cc) * 2, 0, 0));
      } else {
        ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (tc * Cin + cc) * 4, 0, 0));
        rb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (tc * Cin + cc) * 4 + 16, 0, 0));
      }
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      uint4 v;
      if constexpr (IN_BF16) {
        v = __builtin_bit_cast(uint4, ra[i]);
      } else {
        float4 a = ra[i], c = rb[i];
        a.x = fmaxf(a.x, a.x * slope); a.y = fmaxf(a.y, a.y * slope); a.z = fmaxf(a.z, a.z * slope); a.w = fmaxf(a.w, a.w * slope);
        c.x = fmaxf(c.x, c.x * slope); c.y = fmaxf(c.y, c.y * slope); c.z = fmaxf(c.z, c.z * slope); c.w = fmaxf(c.w, c.w * slope);
        v.x = bc_pack(a.x, a.y); v.y = bc_pack(a.z, a.w); v.z = bc_pack(c.x, c.y); v.w = bc_pack(c.z, c.w);
      }
      if (!ok[i]) v = make_uint4(0, 0, 0, 0);
      if (dst[i] >= 0) *reinterpret_cast<uint4*>(smem + dst[i]) = v;
    }
  }
}

// ---- the K loop, shared by both kernels (macros: the ring slots and register sets must be compile-time constants, and counters updated
// inside lambdas ended up in scratch memory -- conv_ksplit.hip).  Expects in scope: D, MT, NT, lane, wr[D][2][NT], xb[2][2][MT],
// acc[MT][NT], RS, and the lambdas request(slot, rsrc, nt0, NU, u), readx(par, a_lane, off), compute(slot).
//
// The main loop walks whole groups of D units with NO condition around a unit: hipcc's s_waitcnt insertion is path-insensitive, and with
// an `if (unit exists)` around each slot it assumed the path on which only slot 0 ran and waited for vmcnt(3) at the head of every group
// -- the whole ring drained once per D units.  Unconditional, slot S waits for exactly its own fragments.  The NU % D units left over run
// after the loop from the fragments already requested.  The activation fragments of unit u + 1 are read from LDS during the MFMAs of
// unit u (the read behind the last unit takes the offset one past the last chunk: inside the allocation, which has 64 spare bytes).
#define E2ETTS_BC_RING_FILL(RSRC, NT0, NU_)                                                                     \
  request(std::integral_constant<int, 0>{}, RSRC, NT0, NU_, 0);                                                 \
  request(std::integral_constant<int, 1>{}, RSRC, NT0, NU_, min(1, NU_ - 1));                                   \
  if constexpr (D > 2) request(std::integral_constant<int, 2>{}, RSRC, NT0, NU_, min(2, NU_ - 1));              \
  if constexpr (D > 3) request(std::integral_constant<int, 3>{}, RSRC, NT0, NU_, min(3, NU_ - 1));              \
  if constexpr (D > 4) request(std::integral_constant<int, 4>{}, RSRC, NT0, NU_, min(4, NU_ - 1));              \
  if constexpr (D > 5) request(std::integral_constant<int, 5>{}, RSRC, NT0, NU_, min(5, NU_ - 1));              \
  if constexpr (D > 6) request(std::integral_constant<int, 6>{}, RSRC, NT0, NU_, min(6, NU_ - 1));              \
  if constexpr (D > 7) request(std::integral_constant<int, 7>{}, RSRC, NT0, NU_, min(7, NU_ - 1));
#define E2ETTS_BC_STEP(KWE_, TAP_STEP_)                                     \
  if (++ju == KWE_) {                                                       \
    ju = 0;                                                                 \
    aoff += 64 - (KWE_ - 1) * TAP_STEP_;                                    \
  } else {                                                                  \
    aoff += TAP_STEP_;                                                      \
  }
#define E2ETTS_BC_UNIT(S, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_)          \
  if constexpr (S < D) {                                                    \
    readx(std::integral_constant<int, (S + 1) & 1>{}, A_LANE, aoff);        \
    compute(std::integral_constant<int, S>{});                              \
    request(std::integral_constant<int, S>{}, RSRC, NT0, NU_, min(ul, NU_ - 1)); \
    bc_pipeline<0, 2 * MT * NT, 2 * MT, 2 * NT>();                          \
    __builtin_amdgcn_sched_barrier(0); /* units do not mix */               \
    ++ul;                                                                   \
    E2ETTS_BC_STEP(KWE_, TAP_STEP_)                                         \
  }
#define E2ETTS_BC_TAIL(S, A_LANE, KWE_, TAP_STEP_)                          \
  if constexpr (S < D - 1) {                                                \
    if (S < ntail) {                                                        \
      readx(std::integral_constant<int, (S + 1) & 1>{}, A_LANE, aoff);      \
      __builtin_amdgcn_sched_barrier(0);                                    \
      compute(std::integral_constant<int, S>{});                            \
      E2ETTS_BC_STEP(KWE_, TAP_STEP_)                                       \
    }                                                                       \
  }
#define E2ETTS_BC_KLOOP(RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_)                                                \
  {                                                                                                             \
    static_assert((D & 1) == 0, "the parity of a unit's register set is its slot's");                           \
    int aoff = 0; /* byte offset of the unit whose fragments are read next: j * tap_step + c * 64 */            \
    int ju = 0;   /* its tap slot */                                                                            \
    int ul = D;   /* next unit whose weights are requested */                                                   \
    const int ngroups = (NU_) / D, ntail = (NU_) - ngroups * D;                                                 \
    readx(std::integral_constant<int, 0>{}, A_LANE, 0);                                                         \
    E2ETTS_BC_STEP(KWE_, TAP_STEP_)                                                                             \
    for (int gi = 0; gi < ngroups; ++gi) {                                                                      \
      E2ETTS_BC_UNIT(0, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_UNIT(1, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) \
      E2ETTS_BC_UNIT(2, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_UNIT(3, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) \
      E2ETTS_BC_UNIT(4, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_UNIT(5, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) \
      E2ETTS_BC_UNIT(6, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_UNIT(7, RSRC, NT0, NU_, A_LANE, KWE_, TAP_STEP_) \
    }                                                                                                           \
    E2ETTS_BC_TAIL(0, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_TAIL(1, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_TAIL(2, A_LANE, KWE_, TAP_STEP_)  \
    E2ETTS_BC_TAIL(3, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_TAIL(4, A_LANE, KWE_, TAP_STEP_) E2ETTS_BC_TAIL(5, A_LANE, KWE_, TAP_STEP_)  \
    E2ETTS_BC_TAIL(6, A_LANE, KWE_, TAP_STEP_)                                                                  \
  }
// the lambdas the macros use, defined in each kernel body by this macro (wr, xb, acc in scope)
#define E2ETTS_BC_LAMBDAS                                                                                                              \
  auto request = [&](auto slot, const __amdgpu_buffer_rsrc_t rsrc, const int nt0_, const int nu_, const int u) __attribute__((always_inline)) { \
    constexpr int S = decltype(slot)::value;                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                                   \
      _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                                                   \
        wr[S][ks][n] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, ((nt0_ + n) * nu_ + u) * 2048 + ks * 1024, 0)); \
  };                                                                                                                                   \
  auto readx = [&](auto par, const unsigned char* a_lane_, const int off) __attribute__((always_inline)) {                             \
    constexpr int P = decltype(par)::value;                                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                                   \
      _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                                   \
        xb[P][ks][m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_lane_ + off + m * 32 * RS + ks * 32));             \
  };                                                                                                                                   \
  auto compute = [&](auto slot) __attribute__((always_inline)) {                                                                       \
    constexpr int S = decltype(slot)::value;                                                                                           \
    constexpr int P = S & 1;                                                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                                   \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) {                                                                                 \
        const bf16x8 w = __builtin_bit_cast(bf16x8, wr[S][ks][n]);                                                                     \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                                 \
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, xb[P][ks][m], acc[m][n], 0, 0, 0);                                    \
      }                                                                                                                                \
  };

// MT x NT blocks of 32 x 32 per wavefront; WGM x WGN wavefronts per workgroup (4); D units of weight fragments in flight.
struct BConvGroup {
  int n;
  int wg_end[BC_GROUP_MAX];   // workgroups of the members < k + 1
  int rtiles, ncg;            // row tiles per utterance, column groups (the same for every member)
  BConvParams p[BC_GROUP_MAX];
};
struct BPairGroup {
  int n;
  int wg_end[BC_GROUP_MAX];
  int rtiles[BC_GROUP_MAX];   // a member's tile height depends on its kernel size
  PairParams p[BC_GROUP_MAX];
};

template <int MT, int NT, int WGM, int WGN, int D, bool IN_BF16>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const BConvGroup grp) {
  static_assert(WGM * WGN == 4, "four wavefronts per workgroup");
  // the member this workgroup serves (uniform): its parameters stay in the kernel-argument segment, read through scalar loads
  int member = 0;
  for (int k = 0; k + 1 < grp.n; ++k) member += (int)blockIdx.x >= grp.wg_end[k] ? 1 : 0;
  const BConvParams& p = grp.p[member];
  const int bid = (int)blockIdx.x - (member ? grp.wg_end[member - 1] : 0);
  const int rtiles = grp.rtiles, ncg = grp.ncg;
  static_assert(D >= 2 && D <= 8, "ring depth");
  constexpr int BM = 32 * MT * WGM;
  constexpr int WN = 32 * NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  // column group fastest: the workgroups that share a slab run next to each other, and with ncg a multiple of 8 each XCD (id % 8) keeps
  // an eighth of the weights in its L2
  const int cg = bid % ncg;
  const int g = bid / ncg;
  const int b = g / rtiles, rt = g - b * rtiles;
  const int t0 = rt * BM;
  const int n0 = (cg * WGN + wn) * WN;   // first column of this wavefront
  const int nt0 = n0 >> 5;
  const int KWe = p.KWe;
  const int NCH = (p.Cin + 31) >> 5;
  const int NU = NCH * KWe;
  const int j0 = (p.tap_split > 0 && n0 >= p.tap_split) ? 1 : 0;   // first live tap of this wavefront's columns
  const int RS = NCH * 64 + 16;          // LDS row stride (bytes)
  const int srows = BM + p.dil * (p.KW - 1);

  unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0;
  BC_STAMP(st0);
  uint4 wr[D][2][NT];
  bf16x8 xb[2][2][MT];
  f32x16 acc[MT][NT];
  E2ETTS_BC_LAMBDAS
  // ---- weight fragments: ring of D units (unit u = chunk * KWe + tap slot), requested before anything else
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(p.wimg), 0, (int)((long long)(p.Cout >> 5) * NU * 2048), 0x00020000);
  E2ETTS_BC_RING_FILL(w_rsrc, nt0, NU)
  BC_STAMP(st1);
  // ---- slab: rows [t0 - pad, t0 - pad + srows) x all channels, as bf16
  constexpr int SB = (2 * NT * D >= 24) ? (IN_BF16 ? 12 : 8) : (IN_BF16 ? 16 : 12);   // the ring's registers are live while the slab is staged
  bc_stage<IN_BF16, SB>(smem, reinterpret_cast<const char*>(p.in) + (long long)b * p.T * p.Cin * (IN_BF16 ? 2 : 4), p.T, p.Cin, NCH, RS,
                                       t0 - p.pad, srows, p.in_slope, tid,
                                       p.in_add[0] ? p.in_add[0] + (long long)b * p.T * p.Cin : nullptr, p.in_add[1] ? p.in_add[1] + (long long)b * p.T * p.Cin : nullptr,
                                       p.in_add[2] ? p.in_add[2] + (long long)b * p.T * p.Cin : nullptr, p.in_div);
  BC_STAMP(st2);
  __syncthreads();   // the one barrier of the tile: slab visible
  BC_STAMP(st3);

#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // ---- K loop: unit (c, j) reads slab rows (output row + (j0 + j) dil), channels [32 c, 32 c + 32)
  const unsigned char* a_lane = smem + (wm * (32 * MT) + li + j0 * p.dil) * RS + lh * 16;
  const int tap_step = p.dil * RS;
  E2ETTS_BC_KLOOP(w_rsrc, nt0, NU, a_lane, KWe, tap_step)

  BC_STAMP(st4);
  // ---- epilogue: v = acc + bias; activation; + residual; + out_old; / div -- conv_gemm's order.  Each 32-position block goes through a
  // wave-private LDS patch [32 positions][WN channels] (written as one ds_write_b128 per register quad, read back with the lanes along
  // the channels) so that global memory sees whole row segments.  The residual / accumulate rows of ALL the wavefront's blocks are
  // requested before the barrier (requested per block, each block waited out a round trip: 7 500 cycles for four blocks).
  constexpr int ELD = WN + 4;                         // patch row stride (floats)
  constexpr int LPR = WN / 4, RPP = 64 / LPR, PASSES = 32 / RPP;
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * ELD);
  const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
  const int ecol = n0 + pc4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + ecol);
  const long long ob = (long long)b * p.T * p.Cout;
  float4 resv[MT][PASSES], accv[MT][PASSES];
  if (p.res) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int t = min(t0 + (wm * MT + m) * 32 + ps * RPP + prow, p.T - 1);
        resv[m][ps] = *reinterpret_cast<const float4*>(p.res + ob + (long long)t * p.Cout + ecol);
      }
  }
  if (p.accumulate) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int t = min(t0 + (wm * MT + m) * 32 + ps * RPP + prow, p.T - 1);
        accv[m][ps] = *reinterpret_cast<const float4*>(p.out + ob + (long long)t * p.Cout + ecol);
      }
  }
  __syncthreads();   // the patches lie over the slab: every wavefront must be done with it
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int tb = t0 + (wm * MT + m) * 32;
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
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      v.x = fmaxf(v.x, v.x * p.act_slope); v.y = fmaxf(v.y, v.y * p.act_slope);
      v.z = fmaxf(v.z, v.z * p.act_slope); v.w = fmaxf(v.w, v.w * p.act_slope);
      if (p.res) {
        const float4 rv = resv[m][ps];
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      if (p.accumulate) {
        const float4 ov = accv[m][ps];
        v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
        if (p.out_div != 1.0f) {
          v.x = v.x / p.out_div; v.y = v.y / p.out_div; v.z = v.z / p.out_div; v.w = v.w / p.out_div;
        }
      }
      if (t < p.T) {
        if (p.out) *reinterpret_cast<float4*>(p.out + ob + (long long)t * p.Cout + ecol) = v;
        if (p.out_b) {
          uint2 h;
          h.x = bc_pack(fmaxf(v.x, v.x * p.outb_slope), fmaxf(v.y, v.y * p.outb_slope));
          h.y = bc_pack(fmaxf(v.z, v.z * p.outb_slope), fmaxf(v.w, v.w * p.outb_slope));
          *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(p.out_b) + ob + (long long)t * p.Cout + ecol) = h;
        }
      }
    }
  }
#ifdef E2ETTS_BC_DIAG
  BC_STAMP(st5);
  if (tid == 0) {
    atomicAdd(&g_bc_diag[0], st1 - st0); atomicAdd(&g_bc_diag[1], st2 - st1); atomicAdd(&g_bc_diag[2], st3 - st2);
    atomicAdd(&g_bc_diag[3], st4 - st3); atomicAdd(&g_bc_diag[4], st5 - st4); atomicAdd(&g_bc_diag[5], st5 - st0); atomicAdd(&g_bc_diag[7], 1ull);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// pair_bf16: one (conv k, dilation d -> leaky ReLU -> conv k -> + x) pair of ResBlock1 (reference V/layers.py:33-40) in one launch, plain
// bf16 -- resblock_pair.hip's arithmetic in mode 2, term for term (same bits), on this file's machinery.  A workgroup owns BMI
// intermediate rows = BMI - (KW - 1) output rows and ALL C channels: the x slab (BMI + (KW - 1) d rows x C, bf16, staged once), conv1 into
// registers, the intermediate = bf16(lrelu(conv1 + b1)) (zero outside [0, T): conv2's padding) written over the slab in the same row
// layout, conv2 from it, the epilogue out = conv2 + b2 + x (+ out_old, / div).  Four workgroup barriers per tile in all (resblock_pair:
// two per 32-channel chunk of conv1 alone), 128-position x 32-channel wavefront tiles (MT = 4: one 1-KiB weight fragment per four MFMAs,
// half of what the L1 can deliver), the weights of conv2's first D units requested before conv1's epilogue.
template <int MT, int WGM, int WGN, int D, bool ACCUM>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void pair_bf16_kernel(const BPairGroup grp) {
  static_assert(WGM * WGN == 4 || WGM * WGN == 8, "four or eight wavefronts per workgroup (eight: 256 channels, one workgroup per CU)");
  constexpr int NTHR = 64 * WGM * WGN;
  int member = 0;
  for (int k = 0; k + 1 < grp.n; ++k) member += (int)blockIdx.x >= grp.wg_end[k] ? 1 : 0;
  const PairParams& p = grp.p[member];
  const int bid = (int)blockIdx.x - (member ? grp.wg_end[member - 1] : 0);
  const int rtiles = grp.rtiles[member];
  constexpr int NT = 1;
  constexpr int C = 32 * WGN;
  constexpr int NCH = WGN;
  constexpr int BMI = 32 * MT * WGM;
  constexpr int RS = NCH * 64 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int b = bid / rtiles, tile = bid - b * rtiles;
  const int KW = p.KW, dil = p.dil;
  const int halo1 = dil * (KW - 1), pad1 = halo1 / 2, pad2 = (KW - 1) / 2;
  const int BMO = BMI - (KW - 1);
  const int o0 = tile * BMO;            // first output row of the tile
  const int i0 = o0 - pad2;             // first intermediate row
  const int NU = NCH * KW;
  const int nt0 = wn;

  unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0;
  BC_STAMP(st0);
  uint4 wr[D][2][NT];
  bf16x8 xb[2][2][MT];
  f32x16 acc[MT][NT];
  E2ETTS_BC_LAMBDAS
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bimg1), 0, NCH * NU * 2048, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bimg2), 0, NCH * NU * 2048, 0x00020000);
  E2ETTS_BC_RING_FILL(w1_rsrc, nt0, NU)
  const float* x_b = p.x + (long long)b * p.x_bs;
  float* out_b = p.out + (long long)b * p.out_bs;
  bc_stage<false, (NTHR == 512 ? 8 : 12), NTHR>(smem, x_b, p.T, C, NCH, RS, i0 - pad1, BMI + halo1, p.slope, tid);
  BC_STAMP(st1);
  __syncthreads();   // slab visible
  BC_STAMP(st2);

#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;

  // ---- conv1: intermediate row r = sum over taps of slab row r + j dil
  const unsigned char* a_lane = smem + (wm * (32 * MT) + li) * RS + lh * 16;
  {
    const int tap_step = dil * RS;
    E2ETTS_BC_KLOOP(w1_rsrc, nt0, NU, a_lane, KW, tap_step)
  }
  BC_STAMP(st3);
  // conv2's first D units: every slot is free now, and the requests fly during the epilogue below
  E2ETTS_BC_RING_FILL(w2_rsrc, nt0, NU)
  __syncthreads();   // every wavefront is done with the slab: the intermediate goes over it

  // ---- epilogue 1: intermediate = lrelu(acc + b1), zero outside [0, T), as bf16 in the slab's row layout.  Accumulator layout of the
  // transposed product: lane (li, lh) = position li of the block, register 4 q + i = channel 8 q + 4 lh + i of the wavefront's 32
  {
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(p.b1 + wn * 32 + 8 * q + 4 * lh);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = wm * (32 * MT) + m * 32 + li;
      const int gr = i0 + row;
      const bool ok = gr >= 0 && gr < p.T;
      unsigned char* dst = smem + row * RS + wn * 64 + lh * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4] = {acc[m][0][4 * q] + bq[q].x, acc[m][0][4 * q + 1] + bq[q].y, acc[m][0][4 * q + 2] + bq[q].z, acc[m][0][4 * q + 3] + bq[q].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = fmaxf(v[i], v[i] * p.slope);
          v[i] = ok ? v[i] : 0.f;
          acc[m][0][4 * q + i] = 0.f;
        }
        uint2 h;
        h.x = bc_pack(v[0], v[1]);
        h.y = bc_pack(v[2], v[3]);
        *reinterpret_cast<uint2*>(dst + q * 16) = h;
      }
    }
  }
  __syncthreads();   // intermediate visible
  BC_STAMP(st4);

  // ---- conv2: output row q = sum over taps of intermediate row q + j
  {
    const int tap_step = RS;
    E2ETTS_BC_KLOOP(w2_rsrc, nt0, NU, a_lane, KW, tap_step)
  }
  BC_STAMP(st5);
  // residual (and running-sum) rows of the wavefront's output blocks, all requested before the barrier (held across conv2 they cost 64-128
  // registers and spilled)
  constexpr int ELD = 32 + 4;
  constexpr int LPR = 8, RPP = 8, PASSES = 4;
  const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
  const int ecol = wn * 32 + pc4;
  float4 resv[MT][PASSES], accv[MT][PASSES];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int t = min(max(o0 + wm * (32 * MT) + m * 32 + ps * RPP + prow, 0), p.T - 1);
      resv[m][ps] = *reinterpret_cast<const float4*>(x_b + (long long)t * C + ecol);
      if constexpr (ACCUM) { if (p.accumulate) accv[m][ps] = *reinterpret_cast<const float4*>(out_b + (long long)t * C + ecol); }
    }
  const float4 bias2 = *reinterpret_cast<const float4*>(p.b2 + ecol);
  __syncthreads();   // intermediate dead: the region now carries the epilogue's patches

  // ---- epilogue 2: out = acc + b2 + x (+ out_old, / div); resblock_pair's order
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * ELD);
  const int t_end = min(o0 + BMO, p.T);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int rb = wm * (32 * MT) + m * 32;   // first tile row of this block
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(patch + li * ELD + 8 * q + 4 * lh) =
          make_float4(acc[m][0][4 * q], acc[m][0][4 * q + 1], acc[m][0][4 * q + 2], acc[m][0][4 * q + 3]);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int row = ps * RPP + prow;
      const int t = o0 + rb + row;
      float4 v = *reinterpret_cast<const float4*>(patch + row * ELD + pc4);
      v.x += bias2.x; v.y += bias2.y; v.z += bias2.z; v.w += bias2.w;
      const float4 rv = resv[m][ps];
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      if (ACCUM && p.accumulate) {   // (a grouped launch may mix accumulating and plain members)
        const float4 ov = accv[m][ps];
        v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
        if (p.out_div != 1.0f) {
          v.x = v.x / p.out_div; v.y = v.y / p.out_div; v.z = v.z / p.out_div; v.w = v.w / p.out_div;
        }
      }
      if (t < t_end && rb + row < BMO) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t nv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(nv, reinterpret_cast<f32x4_t*>(out_b + (long long)t * C + ecol));
      }
    }
  }
#ifdef E2ETTS_BC_DIAG
  BC_STAMP(st6);
  if (tid == 0 && member == 0) {   // the first member's tiles only (largest kernel size in the engine's order)
    atomicAdd(&g_bc_diag[0], st1 - st0); atomicAdd(&g_bc_diag[1], st2 - st1); atomicAdd(&g_bc_diag[2], st3 - st2);
    atomicAdd(&g_bc_diag[3], st4 - st3); atomicAdd(&g_bc_diag[4], st5 - st4); atomicAdd(&g_bc_diag[5], st6 - st5); atomicAdd(&g_bc_diag[6], st6 - st0);
    atomicAdd(&g_bc_diag[7], 1ull);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// rb_bf16: a whole ResBlock1 in one launch (kernels.h: RbParams).  A workgroup owns R = 512 consecutive positions (a tile) of one
// utterance and ALL C channels: 4 wavefronts of 128 positions x 32 channels at C = 32, 8 wavefronts (4 x 2) at C = 64.  Every pair is
// computed on all R positions; positions whose receptive field left the tile come out wrong and are never stored (pair m consumes
// hk (d_m + 1) positions per edge, hk = (KW - 1) / 2: H = 12 / 36 / 60 for the kernel sizes 3 / 7 / 11 at dilations 1, 3, 5).
//   * residual stream x_m: registers, accumulator layout (lane = position, register quad = four consecutive channels);
//   * LDS: X = bf16(lrelu(x_m)) with gx = hk d_max guard rows of zeros at either end, I = bf16(lrelu(c1 + b1)) with hk guard rows; rows
//     outside [0, T) are zero in both (the convolutions' zero padding);
//   * two workgroup barriers per pair; the weights of the next convolution's first D units are requested before each epilogue.
struct BRbGroup {
  int n;
  int wg_end[BC_GROUP_MAX];
  int rtiles[BC_GROUP_MAX];
  int st_H, st_gx, st_hk;   // STAGE form: the tile geometry all members share (the maxima over them)
  RbParams p[BC_GROUP_MAX];
};

// STAGE: ONE workgroup computes ALL the group's ResBlocks on its tile, one after the other, from the x_0 it loaded once, and writes their
// sum / n: the whole `xs / num_kernels` of reference V/generator.py:44-48 in one launch, with the additions in accum_div's order
// ((S_0 + S_1) + S_2) / n, so the bits are those of the separate launches + join.  Per stage the activations then cross the memory system
// twice (x in, sum out) instead of nine times (x in three times, three partial sums out, three in again at the join or in the next layer's
// staging) -- which is what the 32-channel stage of a 48 kHz window (35 MB per tensor) spends its time on.
template <int MT, int WGM, int WGN, int D, bool ACCUM, bool STAGE>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN) / 4) void rb_bf16_kernel(const BRbGroup grp) {
  constexpr int NT = 1;
  constexpr int NWAVE = WGM * WGN, NTHR = 64 * NWAVE;
  constexpr int C = 32 * WGN;
  constexpr int NCH = WGN;
  constexpr int R = 32 * MT * WGM;
  constexpr int RS = NCH * 64 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int member = 0;
  if constexpr (!STAGE)
    for (int k = 0; k + 1 < grp.n; ++k) member += (int)blockIdx.x >= grp.wg_end[k] ? 1 : 0;
  const int bid = (int)blockIdx.x - (member ? grp.wg_end[member - 1] : 0);
  const int rtiles = grp.rtiles[member];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int li = lane & 31, lh = lane >> 5;
  const int b = bid / rtiles, tile = bid - b * rtiles;
  // tile geometry: H positions per edge are lost to the receptive field, X has gx guard rows of zeros at either end, I has hkI
  int H = 0, gx = 0, hkI = 0;
  if constexpr (STAGE) {
    H = grp.st_H; gx = grp.st_gx; hkI = grp.st_hk;
  } else {
    const RbParams& p0 = grp.p[member];
    hkI = (p0.KW - 1) / 2;
    for (int m = 0; m < p0.n_pairs; ++m) {
      H += hkI * (p0.dil[m] + 1);
      gx = max(gx, hkI * p0.dil[m]);
    }
  }
  const int T = grp.p[member].T;
  const int RO = R - 2 * H;             // valid output positions per tile
  const int xrows = R + 2 * gx;
  unsigned char* Xs = smem;
  unsigned char* Is = smem + xrows * RS;
  const int origin = tile * RO - H;     // global position of tile row 0
  const int nt0 = wn;
  const float* x_b = grp.p[member].x + (long long)b * grp.p[member].x_bs;
  float* out_b = grp.p[member].out + (long long)b * grp.p[member].out_bs;

  unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, sc1 = 0, se1 = 0, sc2 = 0, se2 = 0;
  BC_STAMP(d0);
  uint4 wr[D][2][NT];
  bf16x8 xb[2][2][MT];
  f32x16 acc[MT][NT], xres[MT];
  f32x16 x0[STAGE ? MT : 1], sum[STAGE ? MT : 1];
  E2ETTS_BC_LAMBDAS
  // the first convolution's first D units
  {
    const RbParams& pf = grp.p[member];
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pf.bimg[0][0]), 0, NCH * (NCH * pf.KW) * 2048, 0x00020000);
    E2ETTS_BC_RING_FILL(w_rsrc, nt0, NCH * pf.KW)
  }
  // ---- zero the guard rows of both images (never written again)
  for (int i = tid; i < 2 * gx * (RS / 16); i += NTHR) {
    const int r = i / (RS / 16), c16 = i - r * (RS / 16);
    const int row = r < gx ? r : R + r;   // rows [0, gx) and [R + gx, R + 2 gx)
    *reinterpret_cast<uint4*>(Xs + row * RS + c16 * 16) = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < 2 * hkI * (RS / 16); i += NTHR) {
    const int r = i / (RS / 16), c16 = i - r * (RS / 16);
    const int row = r < hkI ? r : R + r;
    *reinterpret_cast<uint4*>(Is + row * RS + c16 * 16) = make_uint4(0, 0, 0, 0);
  }
  // ---- x_0 into registers: lane (li, lh) = position li of the block, register 4 q + i = channel 8 q + 4 lh + i of the wavefront's 32.
  // Global memory is read in whole row segments (8 lanes x 16 B = the wavefront's 128 B of a row, 8 rows per instruction) into a
  // wave-private LDS patch and picked up from there in accumulator layout: read straight in that layout -- 32 bytes per row and
  // instruction -- the 16 loads of a 512-position tile and the 16 stores at the end took 17 000 + 30 000 of the tile's 65 000 cycles.
  // The patches lie in I's interior (written by nobody before pair 0's conv1 is done).
  constexpr int ELD = 36;                 // patch row stride (floats): a 16-lane group of b128 accesses covers all 64 banks once
  constexpr int LPR = 8, RPP = 8, PASSES = 4;
  const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
  {
    float* patch = reinterpret_cast<float*>(Is + hkI * RS) + wave * (32 * ELD);
    float4 rowv[MT][PASSES];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int g = origin + wm * (32 * MT) + m * 32 + ps * RPP + prow;
        const float4 v = *reinterpret_cast<const float4*>(x_b + (long long)min(max(g, 0), T - 1) * C + wn * 32 + pc4);
        rowv[m][ps] = (g >= 0 && g < T) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) *reinterpret_cast<float4*>(patch + (ps * RPP + prow) * ELD + pc4) = rowv[m][ps];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(patch + li * ELD + 8 * q + 4 * lh);
        xres[m][4 * q + 0] = v.x; xres[m][4 * q + 1] = v.y; xres[m][4 * q + 2] = v.z; xres[m][4 * q + 3] = v.w;
      }
      if constexpr (STAGE) x0[m] = xres[m];
    }
  }
  BC_STAMP(d1);

  const int nmem = STAGE ? grp.n : 1;
  for (int mem = 0; mem < nmem; ++mem) {
    const RbParams& p = grp.p[STAGE ? mem : member];
    const int KW = p.KW, hk = (KW - 1) / 2, NP = p.n_pairs;
    const int NU = NCH * KW;
    // an accumulator-layout tile -> operand image: (+ bias), lrelu, zero outside [0, T), round to bf16; four consecutive channels of one
    // position per register quad = 8 bytes of that position's row
    auto write_image = [&](const f32x16 (&src)[MT], unsigned char* img, const int guard, const float* bias /* [C] or null */) __attribute__((always_inline)) {
      float4 bq[4];
      if (bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + wn * 32 + 8 * q + 4 * lh);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int row = wm * (32 * MT) + m * 32 + li;
        const int g = origin + row;
        const bool ok = g >= 0 && g < T;
        unsigned char* dst = img + (row + guard) * RS + wn * 64 + lh * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4] = {src[m][4 * q], src[m][4 * q + 1], src[m][4 * q + 2], src[m][4 * q + 3]};
          if (bias) { v[0] += bq[q].x; v[1] += bq[q].y; v[2] += bq[q].z; v[3] += bq[q].w; }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = fmaxf(v[i], v[i] * p.slope);
            v[i] = ok ? v[i] : 0.f;
          }
          uint2 h;
          h.x = bc_pack(v[0], v[1]);
          h.y = bc_pack(v[2], v[3]);
          *reinterpret_cast<uint2*>(dst + q * 16) = h;
        }
      }
    };
    if constexpr (STAGE) {
      if (mem > 0) {   // the next ResBlock starts from x_0 again; its first convolution's weights
#pragma unroll
        for (int m = 0; m < MT; ++m) xres[m] = x0[m];
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bimg[0][0]), 0, NCH * NU * 2048, 0x00020000);
        E2ETTS_BC_RING_FILL(w_rsrc, nt0, NU)
      }
    }
    write_image(xres, Xs, gx, nullptr);   // (X was last read before the previous ResBlock's last barrier)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;

    for (int pm = 0; pm < NP; ++pm) {
      const int d = p.dil[pm];
      __syncthreads();  // X = lrelu(x_pm) visible (and every wavefront is done with I of the previous pair)
      BC_STAMP(d2);
      // conv1: position r reads X rows r + gx + (j - hk) d
      {
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bimg[pm][0]), 0, NCH * NU * 2048, 0x00020000);
        const unsigned char* a_lane = Xs + (wm * (32 * MT) + li + gx - hk * d) * RS + lh * 16;
        const int tap_step = d * RS;
        E2ETTS_BC_KLOOP(w_rsrc, nt0, NU, a_lane, KW, tap_step)
      }
      BC_STAMP(d3);
      sc1 += d3 - d2;
      const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bimg[pm][1]), 0, NCH * NU * 2048, 0x00020000);
      E2ETTS_BC_RING_FILL(w2_rsrc, nt0, NU)
      {
        f32x16 t[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) t[m] = acc[m][0];
        write_image(t, Is, hkI, p.b1[pm]);   // I = lrelu(c1 + b1), zero outside [0, T)
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.f;
      __syncthreads();  // I visible; every wavefront is done reading X
      BC_STAMP(d4);
      se1 += d4 - d3;
      // conv2: position r reads I rows r + hkI + (j - hk)
      {
        const unsigned char* a_lane = Is + (wm * (32 * MT) + li + hkI - hk) * RS + lh * 16;
        const int tap_step = RS;
        E2ETTS_BC_KLOOP(w2_rsrc, nt0, NU, a_lane, KW, tap_step)
      }
      BC_STAMP(d5);
      sc2 += d5 - d4;
      if (pm + 1 < NP) {
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.bimg[pm + 1][0]), 0, NCH * NU * 2048, 0x00020000);
        E2ETTS_BC_RING_FILL(w_rsrc, nt0, NU)
      }
      // x_{pm+1} = (c2 + b2) + x_pm, in registers
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = *reinterpret_cast<const float4*>(p.b2[pm] + wn * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          xres[m][4 * q + 0] = (acc[m][0][4 * q + 0] + bv.x) + xres[m][4 * q + 0];
          xres[m][4 * q + 1] = (acc[m][0][4 * q + 1] + bv.y) + xres[m][4 * q + 1];
          xres[m][4 * q + 2] = (acc[m][0][4 * q + 2] + bv.z) + xres[m][4 * q + 2];
          xres[m][4 * q + 3] = (acc[m][0][4 * q + 3] + bv.w) + xres[m][4 * q + 3];
          acc[m][0][4 * q + 0] = 0.f; acc[m][0][4 * q + 1] = 0.f; acc[m][0][4 * q + 2] = 0.f; acc[m][0][4 * q + 3] = 0.f;
        }
      }
      if (pm + 1 < NP) write_image(xres, Xs, gx, nullptr);  // X is dead since the barrier above
#ifdef E2ETTS_BC_DIAG
      { unsigned long long dz; BC_STAMP(dz); se2 += dz - d5; }
#endif
    }
    if constexpr (STAGE) {   // the running sum over the ResBlocks, in the join's order: (S_0 + S_1) + S_2 ...
#pragma unroll
      for (int m = 0; m < MT; ++m) sum[m] = mem == 0 ? xres[m] : sum[m] + xres[m];
    }
  }
  BC_STAMP(d2);

  // ---- out = x_n (+ out_old, / div) -- STAGE: the sum over the ResBlocks / n -- on the positions this tile owns: tile rows [H, R - H),
  // global rows < T.  Through wave-private patches again (over X, dead since the last conv1), so that the stores are whole row segments.
  const int g_end = min((tile + 1) * RO, T);
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  {
    const RbParams& p = grp.p[member];
    const float sdiv = (float)grp.n;
    float* patch = reinterpret_cast<float*>(Xs) + wave * (32 * ELD);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4_t ov[PASSES];
      if (ACCUM && p.accumulate) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int g = origin + wm * (32 * MT) + m * 32 + ps * RPP + prow;
          ov[ps] = *reinterpret_cast<const f32x4_t*>(out_b + (long long)min(max(g, 0), T - 1) * C + wn * 32 + pc4);
        }
      }
      const f32x16& src = STAGE ? sum[STAGE ? m : 0] : xres[m];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(patch + li * ELD + 8 * q + 4 * lh) = make_float4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int row = wm * (32 * MT) + m * 32 + ps * RPP + prow;
        const int g = origin + row;
        const float4 pv = *reinterpret_cast<const float4*>(patch + (ps * RPP + prow) * ELD + pc4);
        f32x4_t v = {pv.x, pv.y, pv.z, pv.w};
        if (STAGE) {
          if (grp.n > 1) v = v / sdiv;
        } else if (ACCUM && p.accumulate) {
          v += ov[ps];
          if (p.out_div != 1.0f) v = v / p.out_div;
        }
        if (row >= H && row < R - H && g < g_end) __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(out_b + (long long)g * C + wn * 32 + pc4));
      }
    }
  }
#ifdef E2ETTS_BC_DIAG
  BC_STAMP(d3);
  if (tid == 0 && member == 0) {
    atomicAdd(&g_bc_diag[0], d1 - d0); atomicAdd(&g_bc_diag[1], sc1); atomicAdd(&g_bc_diag[2], se1); atomicAdd(&g_bc_diag[3], sc2);
    atomicAdd(&g_bc_diag[4], se2); atomicAdd(&g_bc_diag[5], d3 - d2); atomicAdd(&g_bc_diag[6], d3 - d0); atomicAdd(&g_bc_diag[7], 1ull);
  }
#endif
}

// hi halves of the split-precision image in conv_bf16's order; one thread moves one lane's 16 bytes
__global__ void bf16_image_kernel(const uint4* __restrict__ x3, uint4* __restrict__ img, int Cout, int KW, int KWe, int nchunk, int tap_split,
                                  long long groups) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const int lane = (int)(g & 63), ks = (int)((g >> 6) & 1);
  long long r = g >> 7;
  const int js = (int)(r % KWe); r /= KWe;
  const int c = (int)(r % nchunk);
  const int t = (int)(r / nchunk);
  const int n = t * 32 + (lane & 31);
  const int j = js + ((tap_split > 0 && t * 32 >= tap_split) ? 1 : 0);
  // one x3 row = 64 bf16 = 8 groups of 16 bytes: [hi k 0-7, 8-15, 16-23, 24-31 | lo ...]
  img[g] = x3[(((long long)n * KW + j) * nchunk + c) * 8 + ks * 2 + (lane >> 5)];
}

struct BcCfg {
  int MT, NT, WGM, WGN;
};

// Tile shape.  What bounds a plain-bf16 convolution on this chip is the path weight fragments take into the registers: 64 B / clk per CU
// through the L1, whether they hit there or not (tools/bconv_bench with s_memtime stamps: 259 cycles per (chunk, tap) unit for the four
// wavefronts' 16 KiB, i.e. 62 B / clk, with the fragments pinned in L1 just as from L2).  A 1-KiB fragment must therefore feed several
// MFMAs of its wavefront: MT position blocks per wavefront need 1024 / MT bytes per MFMA against the 512 the L1 can deliver per MFMA at
// full matrix rate -- MT = 4 (a 128-position x 32-channel wavefront tile; the activations come from LDS, whose 256 B / clk the eight
// ds_read_b128 per unit use half of) leaves headroom, MT = 2 sits on the limit, MT = 1 runs at half rate.  So: 32 columns per wavefront,
// and as many position blocks as still leave the launch a useful number of workgroups.
BcCfg bc_choose(const BConvParams& p) {
  BcCfg c{1, 1, 4, 1};                                  // wavefronts stacked on the rows (Cout % 32 == 0)
  if (p.Cout % 128 == 0) c = {1, 1, 1, 4};              // ... side by side on the columns
  else if (p.Cout % 64 == 0) c = {1, 1, 2, 2};
  static const int mt_env = getenv("E2ETTS_BCONV_MT") ? atoi(getenv("E2ETTS_BCONV_MT")) : 0;   // tuning aids
  static const int nt_env = getenv("E2ETTS_BCONV_NT") ? atoi(getenv("E2ETTS_BCONV_NT")) : 0;
  const long long rows = p.rows_hint > 0 ? p.rows_hint : (long long)p.B * p.T;
  auto wgs = [&](int mt) { return ((rows + 32 * mt * c.WGM - 1) / (32 * mt * c.WGM)) * (p.Cout / (32 * c.WGN)); };
  const int halo = p.dil * (p.KW - 1);
  auto lds = [&](int mt) { return (long long)(32 * mt * c.WGM + halo) * (((p.Cin + 31) / 32) * 64 + 16) + 64; };
  c.MT = 1;
  for (int mt : {4, 2})   // the largest wavefront tile that leaves >= 48 workgroups, and two of them per CU once there is more than one per CU
    if (wgs(mt) >= 48 && lds(mt) <= (wgs(mt) <= 256 ? 160 : 80) * 1024) { c.MT = mt; break; }
  if (mt_env > 0 && lds(mt_env) <= 160 * 1024) c.MT = mt_env;
  // two column blocks per wavefront (MT <= 2, Cout % 256 == 0): half the workgroups stage the same slab; only on request
  if (nt_env == 2 && c.MT <= 2 && c.WGN == 4 && p.Cout % 256 == 0 && (p.tap_split % 64) == 0) c.NT = 2;
  return c;
}

template <int MT, int NT, int WGM, int WGN, int D>
const char* bc_launch(const BConvParams* ps, int n, hipStream_t s) {
  constexpr int BM = 32 * MT * WGM, BN = 32 * NT * WGN;
  const BConvParams& p0 = ps[0];
  const int NCH = (p0.Cin + 31) / 32;
  BConvGroup g;
  g.n = n;
  g.rtiles = (p0.T + BM - 1) / BM;
  g.ncg = p0.Cout / BN;
  size_t lds = (size_t)4 * 32 * (32 * NT + 4) * 4;   // the epilogue's patches
  long long nwg = 0;
  for (int k = 0; k < n; ++k) {
    lds = std::max(lds, (size_t)(BM + ps[k].dil * (ps[k].KW - 1)) * (NCH * 64 + 16) + 64);   // slab + the read-ahead past the last unit
    nwg += (long long)g.rtiles * p0.B * g.ncg;
    g.wg_end[k] = (int)nwg;
    g.p[k] = ps[k];
  }
  for (int k = n; k < BC_GROUP_MAX; ++k) g.wg_end[k] = (int)nwg;
  if (lds > 160 * 1024) return "conv_bf16: slab exceeds the CU's LDS";
  if (nwg >= (1LL << 31)) return "conv_bf16: grid too large";
  static const long lds_min = getenv("E2ETTS_BCONV_LDS_MIN") ? atol(getenv("E2ETTS_BCONV_LDS_MIN")) : 0;   // tuning aid: occupancy experiments
  lds = std::max(lds, (size_t)std::min(lds_min, 160L * 1024));
  static bool attr_done = false;   // > 64 KiB of dynamic LDS needs the opt-in, once per instantiation
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MT, NT, WGM, WGN, D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<MT, NT, WGM, WGN, D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (p0.in_bf16) hipLaunchKernelGGL((conv_bf16_kernel<MT, NT, WGM, WGN, D, true>), dim3((unsigned)nwg), dim3(256), lds, s, g);
  else hipLaunchKernelGGL((conv_bf16_kernel<MT, NT, WGM, WGN, D, false>), dim3((unsigned)nwg), dim3(256), lds, s, g);
  return hipGetLastError() == hipSuccess ? nullptr : "conv_bf16: launch failed";
}

}  // namespace

bool conv_bf16_supported(const BConvParams& p) {
  const int halo = p.dil * (p.KW - 1);
  if (!(p.B > 0 && p.T > 0 && p.Cin >= 8 && (p.Cin % 8) == 0 && p.Cout >= 32 && (p.Cout % 32) == 0 && p.KW >= 1 && p.dil >= 1 && halo <= BC_MAX_HALO &&
        p.pad >= 0 && p.pad <= halo))
    return false;
  if (p.tap_split > 0 && !(p.KW == 3 && p.KWe == 2 && (p.tap_split % 32) == 0 && p.dil == 1)) return false;
  if (p.tap_split == 0 && p.KWe != p.KW) return false;
  if ((long long)p.T * std::max(p.Cin, p.Cout) * 4 >= (1LL << 31)) return false;   // 32-bit buffer offsets per utterance
  const BcCfg c = bc_choose(p);
  const long long lds = (long long)(32 * c.MT * c.WGM + halo) * (((p.Cin + 31) / 32) * 64 + 16);
  return lds <= 160 * 1024;
}

const char* conv_bf16_class(const BConvParams& p) {
  const BcCfg c = bc_choose(p);
  static char names[16][24];
  static int next = 0;
  char nm[24];
  snprintf(nm, sizeof nm, "conv_bf16_%dx%d", 32 * c.MT * c.WGM, 32 * c.NT * c.WGN);
  for (int i = 0; i < next; ++i)
    if (!strcmp(names[i], nm)) return names[i];
  if (next == 16) return "conv_bf16";
  strcpy(names[next], nm);
  return names[next++];
}

static const char* bc_check(const BConvParams& p) {
  if (!p.in || !p.wimg || (!p.out && !p.out_b)) return "conv_bf16: null pointer";
  if (!conv_bf16_supported(p)) return "conv_bf16: unsupported shape";
  if (p.accumulate && !p.out) return "conv_bf16: accumulate needs the fp32 output";
  if (p.in_add[0] && p.in_bf16) return "conv_bf16: in_add needs an fp32 input";
  if ((p.in_add[1] && !p.in_add[0]) || (p.in_add[2] && !p.in_add[1])) return "conv_bf16: in_add must be filled from the front";
  if (((uintptr_t)p.in_add[0] | (uintptr_t)p.in_add[1] | (uintptr_t)p.in_add[2]) & 15) return "conv_bf16: pointers must be 16-byte aligned";
  if (p.out_div != 1.0f && !p.accumulate) return "conv_bf16: out_div needs accumulate";
  if (((uintptr_t)p.in | (uintptr_t)p.wimg | (uintptr_t)p.out | (uintptr_t)p.out_b | (uintptr_t)p.res | (uintptr_t)p.bias) & 15)
    return "conv_bf16: pointers must be 16-byte aligned";
  return nullptr;
}

const char* launch_conv_bf16_group(const BConvParams* ps, int n, hipStream_t s) {
  if (!ps || n < 1 || n > BC_GROUP_MAX) return "conv_bf16: a group has 1 .. BC_GROUP_MAX members";
  // one tile shape for the launch: the one the members' common geometry asks for, with the rows of ALL members counted (the group fills
  // the chip together); the deepest member decides whether the slab still fits
  BConvParams ref = ps[0];
  ref.rows_hint = (int)std::min<long long>((long long)ps[0].B * ps[0].T * n, 0x7fffffffLL);
  for (int k = 0; k < n; ++k) {
    if (const char* m = bc_check(ps[k])) return m;
    if (ps[k].B != ps[0].B || ps[k].T != ps[0].T || ps[k].Cin != ps[0].Cin || ps[k].Cout != ps[0].Cout || ps[k].in_bf16 != ps[0].in_bf16 ||
        (ps[k].tap_split > 0) != (ps[0].tap_split > 0))
      return "conv_bf16: the members of a group share B, T, Cin, Cout and the input type";
    if (ps[k].dil * (ps[k].KW - 1) > ref.dil * (ref.KW - 1)) { ref.KW = ps[k].KW; ref.dil = ps[k].dil; }
  }
  const BcCfg c = bc_choose(ref);
#define E2ETTS_BC_CASE(mt, nt, wgm, wgn, d) \
  if (c.MT == mt && c.NT == nt && c.WGM == wgm && c.WGN == wgn) return bc_launch<mt, nt, wgm, wgn, d>(ps, n, s);
  E2ETTS_BC_CASE(1, 2, 1, 4, 8)
  E2ETTS_BC_CASE(2, 2, 1, 4, 6)
  E2ETTS_BC_CASE(1, 1, 1, 4, 8)
  E2ETTS_BC_CASE(2, 1, 1, 4, 8)
  E2ETTS_BC_CASE(4, 1, 1, 4, 6)
  E2ETTS_BC_CASE(4, 1, 2, 2, 6)
  E2ETTS_BC_CASE(4, 1, 4, 1, 6)
  E2ETTS_BC_CASE(1, 1, 2, 2, 8)
  E2ETTS_BC_CASE(2, 1, 2, 2, 8)
  E2ETTS_BC_CASE(1, 1, 4, 1, 8)
  E2ETTS_BC_CASE(2, 1, 4, 1, 8)
#undef E2ETTS_BC_CASE
  return "conv_bf16: no instantiation for the chosen tile";
}

const char* launch_conv_bf16(const BConvParams& p, hipStream_t s) { return launch_conv_bf16_group(&p, 1, s); }

namespace {

template <int MT, int WGM, int WGN, int D>
const char* pb_launch(const PairParams* ps, int n, hipStream_t s) {
  constexpr int BMI = 32 * MT * WGM, RS = WGN * 64 + 16;
  BPairGroup g;
  g.n = n;
  size_t lds = (size_t)WGM * WGN * 32 * 36 * 4;
  long long nwg = 0;
  bool any_acc = false;
  for (int k = 0; k < n; ++k) {
    const int BMO = BMI - (ps[k].KW - 1);
    lds = std::max(lds, (size_t)(BMI + ps[k].dil * (ps[k].KW - 1)) * RS + 64);
    g.rtiles[k] = (ps[k].T + BMO - 1) / BMO;
    nwg += (long long)g.rtiles[k] * ps[k].B;
    g.wg_end[k] = (int)nwg;
    g.p[k] = ps[k];
    any_acc = any_acc || ps[k].accumulate;
  }
  for (int k = n; k < BC_GROUP_MAX; ++k) { g.wg_end[k] = (int)nwg; g.rtiles[k] = 1; }
  if (lds > 160 * 1024) return "pair_bf16: slab exceeds the CU's LDS";
  if (nwg >= (1LL << 31)) return "pair_bf16: grid too large";
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_bf16_kernel<MT, WGM, WGN, D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pair_bf16_kernel<MT, WGM, WGN, D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (any_acc) hipLaunchKernelGGL((pair_bf16_kernel<MT, WGM, WGN, D, true>), dim3((unsigned)nwg), dim3(64 * WGM * WGN), lds, s, g);
  else hipLaunchKernelGGL((pair_bf16_kernel<MT, WGM, WGN, D, false>), dim3((unsigned)nwg), dim3(64 * WGM * WGN), lds, s, g);
  return hipGetLastError() == hipSuccess ? nullptr : "pair_bf16: launch failed";
}

// position blocks per wavefront: 4 (128 x 32 wavefront tiles) unless that leaves the launch under ~half a round of workgroups
long long pb_tiles4(const PairParams& p) {
  const int wgm = p.C >= 128 ? 1 : 4 / (p.C / 32);
  return (long long)p.B * ((p.T + 128 * wgm - p.KW) / (128 * wgm - (p.KW - 1)));
}
int pb_mt(long long tiles4) {
  static const int mt_env = getenv("E2ETTS_BPAIR_MT") ? atoi(getenv("E2ETTS_BPAIR_MT")) : 0;   // tuning aid
  if (mt_env == 2 || mt_env == 4) return mt_env;
  return tiles4 >= 128 ? 4 : 2;
}

}  // namespace

bool pair_bf16_supported(const PairParams& p) {
  static const bool on = !(getenv("E2ETTS_BPAIR") && atoi(getenv("E2ETTS_BPAIR")) == 0);   // tuning aid: 0 keeps resblock_pair.hip
  if (!on || p.mode != 2 || !p.bimg1 || !p.bimg2 || p.act_rows) return false;
  if (!(p.C == 32 || p.C == 64 || p.C == 128 || p.C == 256) || !(p.KW & 1) || p.KW < 3 || p.KW > 15 || p.dil < 1 || p.dil * (p.KW - 1) > BC_MAX_HALO) return false;
  static const bool on256 = !(getenv("E2ETTS_BPAIR256") && atoi(getenv("E2ETTS_BPAIR256")) == 0);   // tuning aid
  if (p.C == 256 && !on256) return false;
  if (p.x_bs != (long long)p.T * p.C || p.out_bs != (long long)p.T * p.C) return false;
  return (long long)p.T * p.C * 4 < (1LL << 31);
}

const char* launch_pair_bf16_group(const PairParams* ps, int n, hipStream_t s) {
  if (!ps || n < 1 || n > BC_GROUP_MAX) return "pair_bf16: a group has 1 .. BC_GROUP_MAX members";
  long long tiles4 = 0;
  for (int k = 0; k < n; ++k) {
    const PairParams& p = ps[k];
    if (!p.x || !p.b1 || !p.b2 || !p.out) return "pair_bf16: null pointer";
    if (p.B <= 0 || p.T <= 0) return "pair_bf16: bad dims";
    if (!pair_bf16_supported(p)) return "pair_bf16: unsupported launch";
    if (p.out_div != 1.0f && !p.accumulate) return "pair_bf16: out_div needs accumulate";
    if (((uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.bimg1 | (uintptr_t)p.bimg2 | (uintptr_t)p.b1 | (uintptr_t)p.b2) & 15) return "pair_bf16: pointers must be 16-byte aligned";
    if (p.x == p.out) return "pair_bf16: in-place is not possible (tiles read their neighbours' rows)";
    if (p.B != ps[0].B || p.T != ps[0].T || p.C != ps[0].C) return "pair_bf16: the members of a group share B, T and C";
    tiles4 += pb_tiles4(p);
  }
  const int mt = pb_mt(tiles4);
  const int C = ps[0].C;
  // 256 channels: eight wavefronts side by side on the channels, one workgroup per CU (the slab alone is 60-94 KB); 64-position blocks
  // per wavefront while the launch has fewer tiles than CUs (a 542-frame window: 226 tiles of 64 rows for the three kernel sizes together)
  // (same-box A/B at a 542-frame window, whole pass: 64-position blocks 10.64 ms, 128-position blocks 10.95, two convolution launches 10.76)
  static const int mt256 = getenv("E2ETTS_BPAIR256_MT") ? atoi(getenv("E2ETTS_BPAIR256_MT")) : 0;   // tuning aid (0: by grid size)
  if (C == 256) return (mt256 == 4 || (mt256 == 0 && tiles4 >= 512)) ? pb_launch<4, 1, 8, 6>(ps, n, s) : pb_launch<2, 1, 8, 8>(ps, n, s);
  if (C == 128) return mt == 4 ? pb_launch<4, 1, 4, 6>(ps, n, s) : pb_launch<2, 1, 4, 8>(ps, n, s);
  if (C == 64) return mt == 4 ? pb_launch<4, 2, 2, 6>(ps, n, s) : pb_launch<2, 2, 2, 8>(ps, n, s);
  return mt == 4 ? pb_launch<4, 4, 1, 6>(ps, n, s) : pb_launch<2, 4, 1, 8>(ps, n, s);
}

const char* launch_pair_bf16(const PairParams& p, hipStream_t s) { return launch_pair_bf16_group(&p, 1, s); }

namespace {

void rb_geometry(const RbParams& p, int& H, int& gx) {
  const int hk = (p.KW - 1) / 2;
  H = 0; gx = 0;
  for (int m = 0; m < p.n_pairs; ++m) {
    H += hk * (p.dil[m] + 1);
    gx = std::max(gx, hk * p.dil[m]);
  }
}

template <int MT, int WGM, int WGN, int D>
const char* rb_launch(const RbParams* ps, int n, hipStream_t s) {
  constexpr int R = 32 * MT * WGM, RS = WGN * 64 + 16;
  BRbGroup g;
  g.n = n;
  size_t lds = 0;
  long long nwg = 0;
  bool any_acc = false;
  for (int k = 0; k < n; ++k) {
    int H, gx;
    rb_geometry(ps[k], H, gx);
    const int RO = R - 2 * H, hk = (ps[k].KW - 1) / 2;
    lds = std::max(lds, (size_t)((R + 2 * gx) + (R + 2 * hk)) * RS + 64);
    g.rtiles[k] = (ps[k].T + RO - 1) / RO;
    nwg += (long long)g.rtiles[k] * ps[k].B;
    g.wg_end[k] = (int)nwg;
    g.p[k] = ps[k];
    any_acc = any_acc || ps[k].accumulate;
  }
  for (int k = n; k < BC_GROUP_MAX; ++k) { g.wg_end[k] = (int)nwg; g.rtiles[k] = 1; }
  if (lds > 160 * 1024) return "rb_bf16: LDS images exceed the CU's 160 KiB";
  if (nwg >= (1LL << 31)) return "rb_bf16: grid too large";
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_bf16_kernel<MT, WGM, WGN, D, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_bf16_kernel<MT, WGM, WGN, D, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  if (any_acc) hipLaunchKernelGGL((rb_bf16_kernel<MT, WGM, WGN, D, true, false>), dim3((unsigned)nwg), dim3(64 * WGM * WGN), lds, s, g);
  else hipLaunchKernelGGL((rb_bf16_kernel<MT, WGM, WGN, D, false, false>), dim3((unsigned)nwg), dim3(64 * WGM * WGN), lds, s, g);
  return hipGetLastError() == hipSuccess ? nullptr : "rb_bf16: launch failed";
}

// STAGE form: one workgroup per tile computes every member and writes their sum / n to ps[0].out
template <int MT, int WGM, int WGN, int D>
const char* rb_launch_stage(const RbParams* ps, int n, hipStream_t s) {
  constexpr int R = 32 * MT * WGM, RS = WGN * 64 + 16;
  BRbGroup g;
  g.n = n;
  g.st_H = g.st_gx = g.st_hk = 0;
  for (int k = 0; k < n; ++k) {
    int H, gx;
    rb_geometry(ps[k], H, gx);
    g.st_H = std::max(g.st_H, H); g.st_gx = std::max(g.st_gx, gx); g.st_hk = std::max(g.st_hk, (ps[k].KW - 1) / 2);
    g.p[k] = ps[k];
  }
  const int RO = R - 2 * g.st_H;
  if (RO < R / 2) return "rb_bf16: receptive field too wide for the tile";
  const size_t lds = (size_t)((R + 2 * g.st_gx) + (R + 2 * g.st_hk)) * RS + 64;
  if (lds > 160 * 1024) return "rb_bf16: LDS images exceed the CU's 160 KiB";
  const int rtiles = (ps[0].T + RO - 1) / RO;
  const long long nwg = (long long)rtiles * ps[0].B;
  if (nwg >= (1LL << 31)) return "rb_bf16: grid too large";
  for (int k = 0; k < BC_GROUP_MAX; ++k) { g.wg_end[k] = (int)nwg; g.rtiles[k] = rtiles; }
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rb_bf16_kernel<MT, WGM, WGN, D, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL((rb_bf16_kernel<MT, WGM, WGN, D, false, true>), dim3((unsigned)nwg), dim3(64 * WGM * WGN), lds, s, g);
  return hipGetLastError() == hipSuccess ? nullptr : "rb_bf16: launch failed";
}

}  // namespace

bool rb_bf16_supported(const RbParams& p) {
  static const bool on = !(getenv("E2ETTS_BRB") && atoi(getenv("E2ETTS_BRB")) == 0);   // tuning aid: 0 keeps the pair launches
  if (!on || !(p.C == 32 || p.C == 64) || !(p.KW & 1) || p.KW < 3 || p.KW > 15 || p.n_pairs < 1 || p.n_pairs > RB_MAX_PAIRS) return false;
  int H, gx;
  rb_geometry(p, H, gx);
  for (int m = 0; m < p.n_pairs; ++m)
    if (p.dil[m] < 1 || !p.bimg[m][0] || !p.bimg[m][1] || !p.b1[m] || !p.b2[m]) return false;
  if (512 - 2 * H < 256) return false;   // more than half a tile recomputed
  const size_t lds = (size_t)((512 + 2 * gx) + (512 + (p.KW - 1))) * ((p.C / 32) * 64 + 16) + 64;
  if (lds > 160 * 1024) return false;
  if (p.x_bs != (long long)p.T * p.C || p.out_bs != (long long)p.T * p.C) return false;
  return (long long)p.T * p.C * 4 < (1LL << 31);
}

double rb_bf16_flops(const RbParams& p) { return p.n_pairs * 2.0 * 2.0 * p.B * (double)p.T * p.C * p.KW * p.C; }
double rb_bf16_bytes(const RbParams& p) { return 4.0 * ((double)p.B * p.T * p.C * (2.0 + (p.accumulate ? 1 : 0)) + p.n_pairs * 2.0 * p.C * p.KW * p.C / 2.0); }

const char* launch_rb_bf16_group(const RbParams* ps, int n, hipStream_t s) {
  if (!ps || n < 1 || n > BC_GROUP_MAX) return "rb_bf16: a group has 1 .. BC_GROUP_MAX members";
  for (int k = 0; k < n; ++k) {
    const RbParams& p = ps[k];
    if (!p.x || !p.out) return "rb_bf16: null pointer";
    if (p.B <= 0 || p.T <= 0) return "rb_bf16: bad dims";
    if (!rb_bf16_supported(p)) return "rb_bf16: unsupported launch";
    if (p.slope < 0.f || p.slope > 1.f) return "rb_bf16: slope must lie in [0, 1]";
    if (p.out_div != 1.0f && !p.accumulate) return "rb_bf16: out_div needs accumulate";
    if (((uintptr_t)p.x | (uintptr_t)p.out) & 15) return "rb_bf16: pointers must be 16-byte aligned";
    for (int m = 0; m < p.n_pairs; ++m)
      if (((uintptr_t)p.bimg[m][0] | (uintptr_t)p.bimg[m][1] | (uintptr_t)p.b1[m] | (uintptr_t)p.b2[m]) & 15) return "rb_bf16: weights and biases must be 16-byte aligned";
    if (p.x == p.out) return "rb_bf16: in-place is not possible (tiles read their neighbours' rows)";
    if (p.B != ps[0].B || p.T != ps[0].T || p.C != ps[0].C) return "rb_bf16: the members of a group share B, T and C";
  }
  // 32 channels: 8 wavefronts of 64 positions each (two per SIMD: one covers the other's image writes and barriers) or 4 of 128
  static const int w32 = getenv("E2ETTS_BRB_W32") ? atoi(getenv("E2ETTS_BRB_W32")) : 8;   // tuning aid (3: tiles of 384 positions, two workgroups per CU)
  if (ps[0].C == 32 && w32 == 3) {
    for (int k = 0; k < n; ++k) {
      int H, gx;
      rb_geometry(ps[k], H, gx);
      if (384 - 2 * H < 192) return "rb_bf16: receptive field too wide for the 384-position tile";
    }
    return rb_launch<3, 4, 1, 6>(ps, n, s);
  }
  if (ps[0].C == 32) return w32 == 4 ? rb_launch<4, 4, 1, 4>(ps, n, s) : rb_launch<2, 8, 1, 8>(ps, n, s);
  return rb_launch<4, 4, 2, 4>(ps, n, s);
}

bool rb_bf16_stage_supported(const RbParams* ps, int n) {
  static const bool on = !(getenv("E2ETTS_BRB_STAGE") && atoi(getenv("E2ETTS_BRB_STAGE")) == 0);   // tuning aid: 0 keeps one workgroup per ResBlock
  if (!on || !ps || n < 1 || n > BC_GROUP_MAX || ps[0].C != 32) return false;
  int Hm = 0, gxm = 0, hkm = 0;
  for (int k = 0; k < n; ++k) {
    if (!rb_bf16_supported(ps[k]) || ps[k].accumulate || ps[k].x != ps[0].x || ps[k].B != ps[0].B || ps[k].T != ps[0].T || ps[k].C != ps[0].C ||
        ps[k].slope != ps[0].slope)
      return false;
    int H, gx;
    rb_geometry(ps[k], H, gx);
    Hm = std::max(Hm, H); gxm = std::max(gxm, gx); hkm = std::max(hkm, (ps[k].KW - 1) / 2);
  }
  return 512 - 2 * Hm >= 256 && (size_t)((512 + 2 * gxm) + (512 + 2 * hkm)) * ((ps[0].C / 32) * 64 + 16) + 64 <= 160 * 1024;
}

const char* launch_rb_bf16_stage(const RbParams* ps, int n, hipStream_t s) {
  if (!rb_bf16_stage_supported(ps, n)) return "rb_bf16: unsupported stage launch";
  if (!ps[0].x || !ps[0].out || ps[0].x == ps[0].out) return "rb_bf16: bad pointers";
  if (((uintptr_t)ps[0].x | (uintptr_t)ps[0].out) & 15) return "rb_bf16: pointers must be 16-byte aligned";
  return rb_launch_stage<2, 8, 1, 4>(ps, n, s);
}

size_t bf16_image_bytes(int Cout, int KW, int Cin, int tap_split) {
  return (size_t)((Cout + 31) / 32) * ((Cin + 31) / 32) * (tap_split > 0 ? 2 : KW) * 2048;
}

const char* launch_bf16_image(const float* x3, void* img, int Cout, int KW, int Cin, int tap_split, hipStream_t s) {
  if (!x3 || !img) return "bf16_image: null pointer";
  if (Cout <= 0 || (Cout % 32) || KW <= 0 || Cin <= 0) return "bf16_image: Cout must be a positive multiple of 32";
  if (tap_split > 0 && (KW != 3 || (tap_split % 32))) return "bf16_image: a polyphase image needs KW == 3 and tap_split % 32 == 0";
  const int KWe = tap_split > 0 ? 2 : KW;
  const long long groups = (long long)(bf16_image_bytes(Cout, KW, Cin, tap_split) / 16);
  hipLaunchKernelGGL(bf16_image_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const uint4*>(x3),
                     reinterpret_cast<uint4*>(img), Cout, KW, KWe, (Cin + 31) / 32, tap_split, groups);
  return hipGetLastError() == hipSuccess ? nullptr : "bf16_image: launch failed";
}

}  // namespace e2etts
