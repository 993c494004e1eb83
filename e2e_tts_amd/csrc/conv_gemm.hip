// conv_gemm: "same" 1-D convolution / Linear as an implicit GEMM on the gfx950 fp32 matrix cores.
//
// This one kernel carries ~99 % of the hot path's FLOPs: every Conv1d / Linear of the FastSpeech2 FFT
// blocks (reference U/blocks/transformer.py:213-240, 289-297), the predictors (U/layers.py:410-420,
// 491-505), mel_linear (U/model.py:186), the Postnet (U/layers.py:556-563), and HiFi-GAN's conv_pre,
// polyphase-rewritten ConvTranspose1d upsamplers and ResBlock1 dilated convolutions
// (V/generator.py:37-53, V/layers.py:33-40).
//
// Mapping (channels-last activations [B, T, C], weights [Cout, KW*Cin] tap-major):
//   M = time positions of one utterance, N = Cout, K = KW * Cin.
//   D[t][n] += A[t][k] * B[k][n] with v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD).
//
// Structure.  A workgroup (4 wavefronts of 64) walks a run of consecutive BM x BN output tiles of one utterance
// (persistent over M).  The unit of staging is a WORK ITEM = (tile, 32-channel chunk of Cin): ONE activation slab
// of BM + dil*(KW-1) rows goes to LDS -- the taps of a dilated convolution are row-shifted views of that slab, so
// the input is read once per chunk, not once per tap -- and the KW per-tap weight tiles stream through a
// double-buffered LDS tile.  Both streams are software-pipelined through registers: the next weight tile is
// fetched one tap ahead and the next slab (next chunk, or the next tile's first chunk) one work item ahead,
// so the global-load latency sits behind MFMAs instead of in front of them; the loads of the next tile also fly
// during the epilogue.  LDS rows are 36 floats (32 + 4 pad): a ds_read_b128 of 16 consecutive rows covers all 64
// banks once.  Operand fragments are float4 = 4 consecutive k; lane half h supplies k = 8q + 4h + r to MFMA r for
// both operands, so the k order inside an 8-wide group is permuted identically for A and B.
// Epilogue: accumulators are transposed through a wave-private LDS patch (carved from the idle slab) so that
// bias / residual / accumulate reads and the result store are float4, 256 B contiguous per 16 lanes.
//
// Weights reach the MFMA through that LDS tile (256 x 64 and 256 x 32 tiles) or -- BFRAG: the 128-column tiles and the 64 x 64 few-rows
// tile, in every arithmetic mode since round 2 -- as MFMA fragments straight from L2, in an order written once at load time
// (launch_x3_to_frag / launch_f32_to_frag): no LDS weight tile, no barrier per tap.  The grid is one-dimensional and XCD-aware: the
// column tiles of a row group run next to each other on one XCD's L2 (see the kernel's prologue).
#include <algorithm>
#include <cstdlib>

#include "host_logic.h"
#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifdef E2ETTS_DIAG
__device__ unsigned long long g_conv_diag[8];
#define DIAG_STAMP(var)                                                                        \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
#define DIAG_ADD(slot, t0v, t1v) dsum[slot] += (t1v) - (t0v)
#else
#define DIAG_STAMP(var) do {} while (0)
#define DIAG_ADD(slot, t0v, t1v) do {} while (0)
#endif

namespace {

constexpr int BK = 32;   // channels per K chunk
constexpr int LDK = 36;  // padded LDS row stride (floats)
constexpr int MAX_HALO = 64;

__device__ __forceinline__ float4 lrelu4(float4 v, float slope) {  // 0 <= slope <= 1: lrelu(x) = max(x, slope * x)
  v.x = fmaxf(v.x, v.x * slope);
  v.y = fmaxf(v.y, v.y * slope);
  v.z = fmaxf(v.z, v.z * slope);
  v.w = fmaxf(v.w, v.w * slope);
  return v;
}

// Split-precision ("bf16x3") helpers.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 significant bits.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const bf16x2 r = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32 (round to nearest even)
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

__device__ __forceinline__ float act1(float v, int act, float slope) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_LRELU) return v >= 0.f ? v : v * slope;
  if (act == ACT_SWISH) return v * (1.0f / (1.0f + expf(-v)));
  return v;
}

// X3 = false: operands fp32, v_mfma_f32_32x32x2_f32.
// X3 = true : split precision.  Each fp32 operand is hi + lo (two bf16); the product keeps hi*hi + hi*lo + lo*hi in
//             three v_mfma_f32_32x32x16_bf16 with fp32 accumulation (relative error ~2^-16 per product instead of
//             bf16's 2^-8; measured on the vocoder: wav mean-L1 9e-7 vs fp64, against 6e-8 for fp32 and 5e-4 for
//             plain bf16).  An LDS row holds one 32-channel chunk as [32 bf16 hi | 32 bf16 lo | pad] = the same 144
//             bytes as the fp32 row; activations are split while staging, weights arrive pre-split from the packer.
// MODE 0: exact fp32; 1: bf16x3 split precision; 2: plain bf16 (hi x hi only).  A compile-time mode keeps the MFMA block one
// straight basic block (a runtime flag put a branch between every few MFMAs).
// ACCUM: the epilogue adds what the output rows hold already (requested early, next to the residual rows); a kernel
// parameter of its own so that the launches without it keep their register budget (3 waves / SIMD at 32 channels).
// VEC:   float4 epilogue (channel counts and strides multiples of 4, 16-byte aligned pointers -- every launch of the
//        engine).  The scalar epilogue for odd shapes lives in instantiations of its own: next to the vector one, its
//        conditional loads made hipcc drain vmcnt(0) after every barrier of the main loop, i.e. wait for the weight
//        tile it had just requested before issuing the first MFMA.
// CPI > 0 (KW == 1 launches on the fragment path, Cin a multiple of 32 CPI): a work item stages CPI 32-channel chunks at once, as
//        CPI planes of the slab, and the loop walks them like taps.  A plain Linear has ONE tap per chunk, i.e. two workgroup
//        barriers and a staging pass per 24 MFMAs of a wave; with CPI planes it is one per 24 CPI.  Same chunk order, same bits.
// OCC: waves per SIMD the register allocation must allow.  The exact-fp32 128 x 128 kernel on fragment-order weights keeps only its
// slab in LDS (<= 28 KB), so a third workgroup fits on the CU if the kernel stays within 168 VGPRs.  At OCC = 3 ("LEAN") the weight
// fragments wait in a ring of two k-groups instead of four and the epilogue requests residual rows per 16-row half: no spill left in
// the MFMA loop (36 B / lane outside it; with the full ring the cap spilled a float4 per k-group and lost 13 % on the k = 3 layers).
// Same-box A/B on the headline step: k = 11 layers 139.5 -> 141.5 (128 channels) / 139.9 -> 143.7 (256) TFLOP/s, the FFN's k = 9
// convolution 139 -> 142.8, k = 7 136 -> 140; the class 59.1 -> 58.6 ms/step.  The polyphase upsamplers (structural-zero taps skipped:
// short, irregular iterations) LOSE 8-15 % at three waves, so launch_cfg keeps them -- and every other tile shape -- at OCC = 2.
template <int BM, int BN, int WM, int WN, int MODE, bool ACCUM, bool VEC, bool BFRAG, int CPI = 0, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void conv_gemm_kernel(const ConvParams p, const int tiles_per_block,
                                                                                                      const int gx, const int ny, const RowMap rm) {
  constexpr bool K1 = CPI > 0;
  static_assert(!K1 || BFRAG, "multi-chunk items exist on the fragment path only");
  constexpr bool X3 = MODE != 0;
  constexpr bool SPLIT = MODE == 1;
  constexpr int NWN = BN / WN;
  constexpr int MT = WM / 32, NT = WN / 32;
  static_assert((BM / WM) * NWN == 4, "4 wavefronts per workgroup");
  constexpr int BROWS = BN / 32;                      // weight-tile rows staged per thread
  constexpr int AROWS1 = K1 ? BM / 32 : (BM + MAX_HALO + 31) / 32;  // slab rows staged per thread and plane (upper bound)
  constexpr int AROWS = AROWS1 * (K1 ? CPI : 1);
  constexpr int ELD = WN + 4;                         // epilogue patch row stride (floats)
  constexpr bool LEAN = OCC >= 3;
  static_assert(!LEAN || (MODE == 0 && BFRAG && CPI == 0), "the three-wave form exists for the exact-fp32 fragment path");
  static_assert(4 * 16 * ELD <= BM * LDK, "epilogue patches must fit in the slab");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int halo = p.dil * (p.KW - 1);
  const int arows = BM + halo;
  float* As = smem;
  float* Bs = smem + arows * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: lives in an SGPR
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

  // XCD-aware work mapping (1-D grid).  Workgroups are dealt round-robin to the 8 XCDs, each with its own L2: linear id L runs on XCD
  // L % 8 (observed placement: speed only, never correctness).  A unit of work = (row group g = `tiles_per_block` consecutive tiles of one
  // utterance, column tile y); on each XCD consecutive ids walk the COLUMN TILES of one row group first, so the ny workgroups that read
  // the same activation slab run on the same L2 at about the same time (with an (x, y, z) grid they were a whole grid row apart, on
  // whatever XCD that happened to be), and row groups are interleaved over the XCDs, which keeps them evenly loaded when ragged batches
  // make some utterances short.  PMC: FETCH_SIZE of the decoder's Linear layers (8-9 column tiles) -38 %; speed unchanged (these
  // kernels wait on the matrix pipe, not on memory).  Tried and not taken: a contiguous eighth of each utterance's row groups per XCD
  // (so that neighbouring tiles find their halo rows in the same L2): -5 % FETCH_SIZE on the dilated layers, but 4 x the workgroups
  // for the short sequences of the encoder (+0.45 ms/step).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int by = slot % ny;
  const int g = (slot / ny) * 8 + xcd;
  int b, bx;
  if (rm.n > 0) {  // ragged batch, compact grid: every row group of the grid has rows to compute (kernels.h: RowMap)
    if (!rowmap_find(rm, g, b, bx)) return;  // the padding of the grid to a multiple of 8 row groups
  } else {
    if (g >= gx * p.B) return;
    b = g / gx;
    bx = g - b * gx;
  }
  const int n0 = by * BN;
  // ragged batches: rows >= act_rows[b] are not needed by anyone (see engine.hip) -- whole tiles beyond them are skipped
  const int t_act = p.act_rows ? min(p.act_rows[b], p.T) : p.T;
  const int mtiles = (t_act + BM - 1) / BM;
  const int tile0 = bx * tiles_per_block;
  const int ntile = min(tiles_per_block, mtiles - tile0);
  if (ntile <= 0) return;  // uniform for the workgroup, before any barrier
  const float* in_b = p.in + (long long)b * p.in_bs;
  const int nchunk = (p.Cin + BK - 1) / BK;
  const int KC = X3 ? p.KW * nchunk * BK : p.KW * p.Cin;  // X3 weights: [Cout][KW][nchunk][32 words], chunk-padded
  const int ntap = K1 ? CPI : p.KW;             // loop steps per work item: taps, or the planes of a K1 item
  const int nci = K1 ? nchunk / CPI : nchunk;   // work items per tile
  const int nitem = ntile * nci;
  const int niter = nitem * ntap;
  const int plane = arows * LDK;                // floats per slab plane (K1)

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  float4 breg[BROWS];
  float4 areg[AROWS];

  // Staging loads.  They are UNCONDITIONAL (a load under a runtime predicate makes hipcc branch around it and
  // serialise) and the zero fill is applied when the registers are written to LDS.  They are buffer loads with a
  // loop-invariant per-lane offset and a scalar per-iteration offset: no vector address arithmetic in the loop --
  // next to a wave that streams MFMAs every extra vector instruction costs about one MFMA slot -- and no address
  // temporaries for hipcc to guard with vmcnt waits against the loads still in flight.
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)((long long)p.Cout * KC * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t in_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b), 0, (int)((long long)p.T * p.in_ld * 4), 0x00020000);
  // BFRAG: weights in fragment order [n_tile32][tap][chunk][k-step][hi|lo][lane][8 bf16] (bf16 modes) or
  // [n_tile32][tap][chunk][q 0..3][lane][4 floats] (fp32: lane (li, lh) holds k = 8 q + 4 lh .. + 3 of column li, the float4 the LDS
  // tile would have given it) -- either way 4 KiB per (tile, tap, chunk) in 1 KiB pieces: one fully coalesced 1 KiB
  // buffer_load_dwordx4 per fragment, per wave, served by L2 (the tensor is <= a few MB and shared by every workgroup).
  // No LDS tile, no ds_write pass and -- the point -- no workgroup barrier per tap: barriers remain per slab only.
  const int ntile32 = (p.Cout + 31) / 32;
  const __amdgpu_buffer_rsrc_t wf_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(BFRAG ? p.wfrag : p.w), 0, BFRAG ? (int)((long long)ntile32 * p.KW * nchunk * 4096) : 16, 0x00020000);
  // structural zeros of the polyphase upsampler: this wave's columns lie in one half -> one of the three taps contributes nothing
  int skip_tap = -1;
  if (BFRAG && p.zero_tap_split > 0) {
    const int c0 = n0 + wn * WN;
    if (c0 + WN <= p.zero_tap_split) skip_tap = 2;
    else if (c0 >= p.zero_tap_split) skip_tap = 0;
  }
  int fnt[NT];  // this wave's 32-column tiles (clamped: columns beyond Cout are never stored)
#pragma unroll
  for (int n = 0; n < NT; ++n) fnt[n] = min((n0 + wn * WN) / 32 + n, ntile32 - 1);
  constexpr int KS = X3 ? BK / 16 : BK / 8;  // k-steps per 32-channel chunk: 2 x 16 (bf16 MFMA) or 4 x (4 MFMAs of k = 2) (fp32)
  constexpr int HL = X3 ? 2 : 1;
  float4 bfr[KS][NT][HL];  // [k-step][n tile][hi | lo] fragments of the CURRENT iteration (BFRAG)
  auto load_frag_to = [&](int slot, int chunk, int j, int ks) {
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int hl = 0; hl < HL; ++hl) {
        const int soff = ((((fnt[n] * p.KW + j) * nchunk + chunk) * KS + ks) * HL + hl) * 1024;
        bfr[slot][n][hl] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wf_rsrc, lane * 16, soff, 0));
      }
  };
  auto load_frag = [&](int chunk, int j, int ks) { load_frag_to(ks, chunk, j, ks); };
  int wvoff[BROWS];
  bool wok[BROWS];
  bool w_all_ok = true;
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int n = n0 + lrow + i * 32;
    wok[i] = n < p.Cout;
    wvoff[i] = (min(n, p.Cout - 1) * KC + lc4) * 4;
  }
  w_all_ok = n0 + BN <= p.Cout;               // uniform: no row of the weight tile needs zeroing
  const bool ragged = (p.Cin % BK) != 0;      // uniform: the last chunk is partial (fp32 layout only)
  int avoff[AROWS1];
#pragma unroll
  for (int i = 0; i < AROWS1; ++i) avoff[i] = ((lrow + i * 32) * p.in_ld + lc4) * 4;
  const int cmax = p.Cin - 4;  // Cin % 4 == 0: last float4 of a row
  bool b_cok = true, a_cok = true, b_mask = false, a_edge = false;
  int a_tbase = 0;

  auto load_b = [&](int chunk, int j) {
    const int c = chunk * BK + lc4;
    const bool partial = !X3 && ragged && chunk == nchunk - 1;  // X3 rows are zero-padded to whole chunks by the packer
    b_cok = !partial || c < p.Cin;
    b_mask = partial || !w_all_ok;
    // a partial chunk reads past the row end into the next tap / row (or past the tensor: the buffer returns 0); masked below
    const int soff = (X3 ? (j * nchunk + chunk) * BK : j * p.Cin + chunk * BK) * 4;
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      breg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wvoff[i], soff, 0));
  };
  auto store_b = [&](int buf) {
    float* dst = Bs + buf * (BN * LDK);
    if (b_mask) {
#pragma unroll
      for (int i = 0; i < BROWS; ++i)
        if (!(wok[i] && b_cok)) breg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *reinterpret_cast<float4*>(dst + (lrow + i * 32) * LDK + lc4) = breg[i];
  };
  auto load_a = [&](int tile, int chunk) {
    if constexpr (K1) {  // CPI whole chunks (Cin % (32 CPI) == 0, checked by the launcher); only the last tile can cross T
      a_cok = true;
      a_tbase = tile * BM - p.pad;
      a_edge = a_tbase < 0 || a_tbase + BM > p.T;
#pragma unroll
      for (int pl = 0; pl < CPI; ++pl) {
        const int ch = chunk * CPI + pl;
        if (!a_edge) {
          const int soff = (a_tbase * p.in_ld + ch * BK) * 4;
#pragma unroll
          for (int i = 0; i < AROWS1; ++i)
            areg[pl * AROWS1 + i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, avoff[i], soff, 0));
        } else {
#pragma unroll
          for (int i = 0; i < AROWS1; ++i) {
            const int t = min(max(a_tbase + lrow + i * 32, 0), p.T - 1);
            areg[pl * AROWS1 + i] =
                __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (t * p.in_ld + ch * BK + lc4) * 4, 0, 0));
          }
        }
      }
      return;
    }
    const int c = chunk * BK + lc4;
    a_cok = c < p.Cin;
    a_tbase = tile * BM - p.pad;
    // interior slab of a full chunk: every row and channel exists -> scalar offset, invariant lane offsets
    a_edge = a_tbase < 0 || a_tbase + AROWS1 * 32 > p.T || (ragged && chunk == nchunk - 1);
    if (!a_edge) {
      const int soff = (a_tbase * p.in_ld + chunk * BK) * 4;
#pragma unroll
      for (int i = 0; i < AROWS; ++i)
        areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, avoff[i], soff, 0));
    } else {
      const int cc = min(c, cmax);
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const int t = min(max(a_tbase + lrow + i * 32, 0), p.T - 1);
        areg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (t * p.in_ld + cc) * 4, 0, 0));
      }
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = lrow + (K1 ? i % AROWS1 : i) * 32;
      float* As = smem + (K1 ? (i / AROWS1) * plane : 0);  // plane of this register (shadows the slab base)
      float4 v = areg[i];
      if (a_edge) {
        const int t = a_tbase + r;
        if (!(a_cok && t >= 0 && t < p.T)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (p.in_slope != 1.0f) v = lrelu4(v, p.in_slope);
      if (r < arows) {
        if constexpr (X3) {
          uint2 hi, lo;
          split4(v, hi, lo);
          *reinterpret_cast<uint2*>(As + r * LDK + (lc4 >> 1)) = hi;        // bf16 channels lc4 .. lc4+3 of the hi half
          *reinterpret_cast<uint2*>(As + r * LDK + 16 + (lc4 >> 1)) = lo;   // same channels of the lo half
        } else {
          *reinterpret_cast<float4*>(As + r * LDK + lc4) = v;
        }
      }
    }
  };

  // Epilogue of one output tile.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  // Vector form: accumulators are transposed through a wave-private LDS patch so that every global access is a float4,
  // 256 B contiguous per 16 lanes.  It is written branch-light on purpose: bias is fetched before the main loop, the
  // residual tile is fetched up front with unconditional (clamped) loads, NONE / ReLU / leaky-ReLU are one
  // max(v, v * slope), and only the stores are predicated -- straight-line code lets hipcc count its vmcnt waits instead
  // of draining every load and store one at a time (which cost ~16k cycles per tile).
  const int len = p.lens ? p.lens[b] : p.T;
  float* out_b = p.out + (long long)b * p.out_bs;
  const float* res_b = p.res ? p.res + (long long)b * p.res_bs : nullptr;
  constexpr int LPR = WN / 4;     // lanes per row when the patch is read back as float4
  constexpr int RPP = 64 / LPR;   // rows per pass
  constexpr int PASSES = 16 / RPP;
  const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
  const int ecol = n0 + wn * WN + pc4;
  const bool ecol_ok = ecol < p.Cout;
  const int ecol_c = min(ecol, max(p.Cout - 4, 0));
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (VEC && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + ecol_c);
  const float eslope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LRELU ? p.act_slope : 1.f);

  auto epilogue_vec = [&](int tile, auto has_res) {
    constexpr bool RES = decltype(has_res)::value;
    constexpr bool ACC = ACCUM;
    const int t0 = tile * BM + wm * WM;
    float* patch = As + wave * (16 * ELD);  // wave-private: no workgroup barrier between its write and read
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      // LEAN (the instantiations compiled for three waves per SIMD, 168 registers): residual / accumulate rows are requested per
      // 16-row half, right before that half's transposes, instead of both halves up front -- half the registers, less cover
      float4 resv[LEAN ? 1 : 2][PASSES];  // residual rows of this 32-row block, requested before its transposes
      float4 accv[LEAN ? 1 : 2][PASSES];  // accumulate mode: what the output rows hold now, requested together with the residual
      auto request_rows = [&](int hh) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int t = min(t0 + m * 32 + hh * 16 + ps * RPP + prow, p.T - 1);
          if constexpr (RES) resv[LEAN ? 0 : hh][ps] = *reinterpret_cast<const float4*>(res_b + (long long)t * p.res_ld + ecol_c);
          if constexpr (ACC) accv[LEAN ? 0 : hh][ps] = *reinterpret_cast<const float4*>(out_b + (long long)t * p.out_ld + ecol_c);
        }
      };
      if constexpr (!LEAN) {
        request_rows(0);
        request_rows(1);
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if constexpr (LEAN) request_rows(hh);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) {
            const int r = hh * 8 + rr;
            const int row = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lh;  // 0..15 inside this half
            patch[row * ELD + n * 32 + li] = acc[m][n][r];
          }
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
          const int row = ps * RPP + prow;
          const int t = t0 + m * 32 + hh * 16 + row;
          float4 v = *reinterpret_cast<const float4*>(patch + row * ELD + pc4);
          v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
          v.x = fmaxf(v.x, v.x * eslope); v.y = fmaxf(v.y, v.y * eslope);
          v.z = fmaxf(v.z, v.z * eslope); v.w = fmaxf(v.w, v.w * eslope);
          if (p.act == ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
          if (p.act == ACT_SWISH) {
            v.x *= 1.0f / (1.0f + expf(-v.x)); v.y *= 1.0f / (1.0f + expf(-v.y));
            v.z *= 1.0f / (1.0f + expf(-v.z)); v.w *= 1.0f / (1.0f + expf(-v.w));
          }
          if constexpr (RES) {
            const float4 rv = resv[LEAN ? 0 : hh][ps];
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
          }
          if (t >= len) v = make_float4(0.f, 0.f, 0.f, 0.f);
          const bool ok = t < p.T && ecol_ok;
          float4* o = reinterpret_cast<float4*>(out_b + (long long)min(t, p.T - 1) * p.out_ld + ecol_c);
          if constexpr (ACC) {
            const float4 ov = accv[LEAN ? 0 : hh][ps];
            v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
          }
          if (ACC && p.out_div != 1.0f) {
            v.x = v.x / p.out_div; v.y = v.y / p.out_div; v.z = v.z / p.out_div; v.w = v.w / p.out_div;
          }
          if (ok) *o = v;
        }
      }
    }
  };
  auto epilogue = [&](int tile) {
    if constexpr (VEC) {
      if (res_b) epilogue_vec(tile, std::true_type{});
      else epilogue_vec(tile, std::false_type{});
    } else {
      const int t0 = tile * BM;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int col = n0 + wn * WN + n * 32 + li;
        if (col >= p.Cout) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int t = t0 + wm * WM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (t >= p.T) continue;
            float v = act1(acc[m][n][r] + bias, p.act, p.act_slope);
            if (res_b) v += res_b[(long long)t * p.res_ld + col];
            if (t >= len) v = 0.f;
            float* o = out_b + (long long)t * p.out_ld + col;
            if (p.accumulate) v += *o;
            if (p.out_div != 1.0f) v = v / p.out_div;
            *o = v;
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  };

  __builtin_amdgcn_s_setprio(3);
  load_a(tile0, 0);
  store_a();
  if constexpr (BFRAG) {
#pragma unroll
    for (int ks = 0; ks < (LEAN ? 2 : KS); ++ks) load_frag(0, 0, ks);  // LEAN: a ring of two k-groups (see the fp32 MFMA block)
    // hipcc may issue these in any order; whatever is still in flight at the loop head would force the head's wait for
    // the first fragment down to vmcnt(0) in EVERY iteration.  Drained here, the wait inside the loop stays counted.
    __builtin_amdgcn_s_waitcnt(0x0F70);
  } else {
    load_b(0, 0);
    store_b(0);
  }
  int cur = 0;
  int tl = 0, chunk = 0, j = 0;  // work item = (tile0 + tl, chunk); tap j
#ifdef E2ETTS_DIAG
  unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta = 0, tb = 0;
#endif
  for (int it = 0; it < niter; ++it) {
    DIAG_STAMP(ta);
    const bool last_tap = j == ntap - 1;
    const bool tile_done = last_tap && chunk == nci - 1;
    const bool more_items = !(tile_done && tl == ntile - 1);
    const int nchk = chunk + 1 == nci ? 0 : chunk + 1;
    // Weight tile first, slab second: hipcc reuses the registers of earlier loads for address arithmetic and then
    // waits (vmcnt) for whatever is in flight -- with the slab loads issued first that was a full HBM round trip.
    if constexpr (!BFRAG) {
      if (it + 1 < niter) {  // next weight tile: next tap, or tap 0 of the next item's chunk
        if (!last_tap) load_b(chunk, j + 1);
        else load_b(nchk, 0);
      }
    }
    if (last_tap && more_items) load_a(tile0 + tl + (nchk == 0 ? 1 : 0), nchk);  // next slab: next chunk, or next tile
    DIAG_STAMP(tb); DIAG_ADD(0, ta, tb);
    if (!BFRAG || j == 0) __syncthreads();  // slab (+ Bs[cur]) visible; with BFRAG only a new slab needs the barrier
    DIAG_STAMP(ta); DIAG_ADD(1, tb, ta);

    // Between two waves of a SIMD the one streaming MFMAs starves the other's vector issue (measured: ~1 MFMA
    // slot per VALU / VMEM / LDS instruction).  Staging code therefore runs at raised priority, MFMA blocks at 0.
    __builtin_amdgcn_s_setprio(0);
    const float* a_base = K1 ? As + j * plane + (wm * WM + li) * LDK + lh * 4 : As + (wm * WM + li + j * p.dil) * LDK + lh * 4;
    const float* b_base = Bs + cur * (BN * LDK) + (wn * WN + li) * LDK + lh * 4;
    if constexpr (X3) {
      // lane (row li, half lh) holds k = 16 s + 8 lh .. + 7 of its row: one 16-byte read per operand half
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        if (!BFRAG || j != skip_tap) {  // wave-uniform; the fragment path has no barrier per tap, so a wave may run ahead
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          ah[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + ks * 8));
          al[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + 16 + ks * 8));
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if constexpr (BFRAG) {
            bh[n] = __builtin_bit_cast(bf16x8, bfr[ks][n][0]);
            bl[n] = __builtin_bit_cast(bf16x8, bfr[ks][n][1]);
          } else {
            bh[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(b_base + n * 32 * LDK + ks * 8));
            bl[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(b_base + n * 32 * LDK + 16 + ks * 8));
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (SPLIT) {  // the two cross terms; absent in plain-bf16 mode
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            }
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          }
        }
        if constexpr (BFRAG) {  // fragments of this k-step are consumed: request them for the next iteration (one k-step of cover)
          // unconditional (the very last iteration re-requests fragments nobody uses): with a branch around them
          // hipcc has to assume "first k-step requested, second not" and waits for all eight loads at the loop head
          if constexpr (K1) {  // plane j of item `chunk` is 32-channel chunk chunk * CPI + j of the (single) tap
            load_frag(last_tap ? nchk * CPI : chunk * CPI + j + 1, 0, ks);
          } else {
            if (!last_tap) load_frag(chunk, j + 1, ks);
            else load_frag(nchk, 0, ks);
          }
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        if (!BFRAG || j != skip_tap) {  // wave-uniform (see the bf16 branch)
        float4 af[MT], bf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + q * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if constexpr (BFRAG) bf[n] = bfr[LEAN ? (q & 1) : q][n][0];
          else bf[n] = *reinterpret_cast<const float4*>(b_base + n * 32 * LDK + q * 8);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].x, bf[n].x, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].y, bf[n].y, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].z, bf[n].z, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m].w, bf[n].w, acc[m][n], 0, 0, 0);
          }
        }
        if constexpr (BFRAG && LEAN) {
          // three waves per SIMD (168 registers): the fragments wait in a ring of TWO k-groups instead of a whole iteration's four --
          // 16 registers less, and two k-groups (32 MFMAs, 2 048 cycles) are still several L2 round trips of cover.  Slot q & 1 is
          // consumed: it takes k-group q + 2 of this iteration, or k-group q - 2 of the next one.
          if (q < 2) {
            load_frag_to(q & 1, chunk, j, q + 2);
          } else {
            if (!last_tap) load_frag_to(q & 1, chunk, j + 1, q - 2);
            else load_frag_to(q & 1, nchk, 0, q - 2);
          }
        } else if constexpr (BFRAG) {  // as in the bf16 branch: this k-step's fragments for the NEXT iteration, unconditionally
          if constexpr (K1) {
            load_frag(last_tap ? nchk * CPI : chunk * CPI + j + 1, 0, q);
          } else {
            if (!last_tap) load_frag(chunk, j + 1, q);
            else load_frag(nchk, 0, q);
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(3);
    DIAG_STAMP(tb); DIAG_ADD(2, ta, tb);
    if constexpr (!BFRAG) {
      if (it + 1 < niter) store_b(cur ^ 1);  // that buffer was last read before this iteration's barrier
    }
    cur ^= 1;
    DIAG_STAMP(ta); DIAG_ADD(3, tb, ta);
    if (last_tap) {
      if (tile_done || more_items) __syncthreads();  // every wave is done reading the slab
      DIAG_STAMP(tb); DIAG_ADD(4, ta, tb);
      if (tile_done) {
        epilogue(tile0 + tl);
        if (more_items) __syncthreads();  // patches read back before the slab is overwritten
        ++tl;
      }
      DIAG_STAMP(ta); DIAG_ADD(5, tb, ta);
      if (more_items) store_a();
      DIAG_STAMP(tb); DIAG_ADD(6, ta, tb);
      j = 0;
      chunk = nchk;
    } else {
      ++j;
    }
  }
#ifdef E2ETTS_DIAG
  if (bx == 1 && by == 0 && b == 0 && tid == 0) {
    for (int i = 0; i < 7; ++i) g_conv_diag[i] = dsum[i];
    g_conv_diag[7] = niter;
  }
#endif
}

template <int BM, int BN, int WM, int WN, int MODE, bool ACCUM, bool VEC, bool BFRAG, int CPI = 0, int OCC = 2>
const char* launch_cfg_impl(const ConvParams& p, hipStream_t s) {
  const int halo = p.dil * (p.KW - 1);
  static const int lds_pad = getenv("E2ETTS_LDS_PAD") ? atoi(getenv("E2ETTS_LDS_PAD")) : 0;  // tuning aid: occupancy experiments
  const size_t lds = (size_t)((BM + halo) * LDK * (CPI > 0 ? CPI : 1) + (BFRAG ? 0 : 2 * BN * LDK)) * sizeof(float) + (BFRAG ? lds_pad : 0);
  if (lds > 80 * 1024) return "conv_gemm: LDS tile exceeds 80 KiB";
  if (lds > 64 * 1024) {  // the multi-chunk Linear form: opt in once per instantiation (two workgroups per CU still fit)
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, WM, WN, MODE, ACCUM, VEC, BFRAG, CPI, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (attr != hipSuccess) return "conv_gemm: cannot raise the dynamic LDS limit";
  }
  const int mtiles = (p.T + BM - 1) / BM;
  const int ntiles = (p.Cout + BN - 1) / BN;
  // Persistent over M: enough workgroups for ~24 per CU, each walking consecutive tiles of one utterance.
  // (ragged batches: the tiles that will really run -- sized on the padded total the row groups came out too long and too few, and the
  // last round of workgroups ran a third empty: 110-118 instead of 137-140 TFLOP/s on the mixed-length batch.  24 rather than the
  // earlier 8 for the same reason: a mixed-length batch ends every utterance in a partial group, and with 2-3 workgroups resident per
  // CU eight groups per CU are only three or four rounds -- 140.5 -> 135.4 ms/step on BASELINE config 3, the fixed-length batch unchanged)
  const bool compact = p.act_rows && p.act_rows_host && p.B <= ROWMAP_MAX;  // ragged, lengths known on the host: no empty workgroups
  long long real_tiles = 0;  // row tiles that have rows to compute
  if (compact)
    for (int b = 0; b < p.B; ++b) real_tiles += (std::min(std::max(p.act_rows_host[b], 0), p.T) + BM - 1) / BM;
  const long long total = compact ? real_tiles * ntiles
                                  : (long long)((double)mtiles * ntiles * p.B * (p.act_frac > 0.0 && p.act_frac <= 1.0 ? p.act_frac : 1.0));
  static const int wg_per_cu = getenv("E2ETTS_WG_PER_CU") ? atoi(getenv("E2ETTS_WG_PER_CU")) : 24;  // tuning aid
  int tpb = (int)(total / (256 * wg_per_cu));
  tpb = tpb < 1 ? 1 : (tpb > 64 ? 64 : tpb);
  if (tpb > mtiles) tpb = mtiles;
  const int gx = (mtiles + tpb - 1) / tpb;                        // row groups per utterance (padded length)
  RowMap rm;
  long long groups = (long long)gx * p.B;
  if (compact) {  // row groups per utterance from ITS rows; the kernel re-derives the same counts from act_rows[b] on the device
    rm.n = p.B;
    rm.identity();
    rm.cum[0] = 0;
    for (int b = 0; b < p.B; ++b) {
      const int mt = (std::min(std::max(p.act_rows_host[b], 0), p.T) + BM - 1) / BM;
      rm.cum[b + 1] = rm.cum[b] + (mt + tpb - 1) / tpb;
    }
    groups = rm.cum[p.B];
    if (groups == 0) return nullptr;  // no utterance has a row to compute
  }
  const long long groups8 = (groups + 7) / 8;                     // row groups per XCD
  if (groups8 * 8 * ntiles >= (1LL << 31)) return "conv_gemm: grid too large";
  dim3 grid((unsigned)(groups8 * 8 * ntiles));
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, MODE, ACCUM, VEC, BFRAG, CPI, OCC>), grid, dim3(256), lds, s, p, tpb, gx, ntiles, rm);
  return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm: launch failed";
}

// Fragment-order weights (ConvParams::wfrag) only for the 128-column tile: +7..13 % there (A/B in tools/conv_bench), nothing
// at 64 and 32 columns, where the weight tile is small next to the slab.
template <int BM, int BN, int WM, int WN, int MODE>
const char* launch_cfg(const ConvParams& p, hipStream_t s) {
  // (fp32 fragments for the 256 x 64 / 256 x 32 tiles were measured too: 0 .. -6 %, the LDS tile stays there.)  The 64 x 64 few-rows tile
  // -- small batches, the B = 1 latency path, the encoder -- is a chain of short iterations per wave, where the barrier per tap of the
  // LDS weight tile is a large part of each link: fragments there too (E2ETTS_FRAG64=0: off, tuning aid).
  static const bool frag64 = !(getenv("E2ETTS_FRAG64") && atoi(getenv("E2ETTS_FRAG64")) == 0);
  if constexpr (BN == 128 || (BM == 64 && BN == 64)) {
    if (BN == 64 && !frag64) {
      return p.accumulate ? launch_cfg_impl<BM, BN, WM, WN, MODE, true, true, false>(p, s)
                          : launch_cfg_impl<BM, BN, WM, WN, MODE, false, true, false>(p, s);
    }
    // plain Linear layers (q | k | v, fc, the k = 1 FFN conv, every Conformer GEMM): several chunks per work item
    static const int cpi_env = getenv("E2ETTS_K1_CPI") ? atoi(getenv("E2ETTS_K1_CPI")) : 4;  // tuning aid: 0 = off, 2, 4
    if constexpr (BM == 128 || BM == 64) {
      if (p.wfrag && p.KW == 1 && !p.accumulate && cpi_env > 0) {
        if (cpi_env >= 4 && p.Cin % (4 * BK) == 0) return launch_cfg_impl<BM, BN, WM, WN, MODE, false, true, true, 4>(p, s);
        if (p.Cin % (2 * BK) == 0) return launch_cfg_impl<BM, BN, WM, WN, MODE, false, true, true, 2>(p, s);
      }
    }
    if constexpr (MODE == 0 && BM == 128 && BN == 128) {  // three waves per SIMD (see OCC above); E2ETTS_F32_OCC=2: off (tuning aid)
      static const bool occ3 = !(getenv("E2ETTS_F32_OCC") && atoi(getenv("E2ETTS_F32_OCC")) == 2);
      if (p.wfrag && occ3 && p.zero_tap_split == 0)
        return p.accumulate ? launch_cfg_impl<BM, BN, WM, WN, MODE, true, true, true, 0, 3>(p, s)
                            : launch_cfg_impl<BM, BN, WM, WN, MODE, false, true, true, 0, 3>(p, s);
    }
    if (p.wfrag)
      return p.accumulate ? launch_cfg_impl<BM, BN, WM, WN, MODE, true, true, true>(p, s)
                          : launch_cfg_impl<BM, BN, WM, WN, MODE, false, true, true>(p, s);
  }
  return p.accumulate ? launch_cfg_impl<BM, BN, WM, WN, MODE, true, true, false>(p, s)
                      : launch_cfg_impl<BM, BN, WM, WN, MODE, false, true, false>(p, s);
}

// Tile choice for Cout > 64.  few: so few rows (small batches, the B = 1 latency path, the encoder) that 128 x 128 tiles would leave
// CUs idle -> 64 x 64.  half: on the fragment path, a 128 x 128 grid whose last round of workgroups (2 per CU = 512 at a time) is
// mostly empty -- 576 tiles for a 24576 x 384 Linear are one full round and an eighth of a second -- runs as 64 x 128 tiles: twice
// the workgroups at half the work, e.g. 3 half-rounds instead of 2 full ones.  Same per-wave MFMA : fragment ratio (MT = 2).
// (under one round of 128 x 128 tiles but many rows -- host_logic.h -- the fragment path takes 64 x 128, accumulate or not; without
// fragment-order weights such a launch stays on 64 x 64)
// (the counts are those of the tiles that really run: a ragged batch whose lengths the host knows is sized on them, host_logic.h)
static void launch_counts(const ConvParams& p, long long* t128, long long* rows) {
  if (p.act_rows && p.act_rows_host) {
    ragged_counts(p.act_rows_host, p.B, p.T, p.Cout, t128, rows);
  } else {
    *t128 = tiles_128(p.B, p.T, p.Cout);
    *rows = (long long)p.B * p.T;
  }
}
static bool under_round_many_rows(const ConvParams& p) {
  long long t128, rows;
  launch_counts(p, &t128, &rows);
  return p.Cout > 64 && t128 < 2 * 256 && tile_many_rows_n(rows);
}
bool few_rows(const ConvParams& p) {  // host_logic.h
  long long t128, rows;
  launch_counts(p, &t128, &rows);
  return tile_few_rows_n(t128, rows, p.Cout) || (!p.wfrag && under_round_many_rows(p));
}
bool half_rows(const ConvParams& p) {
  static const bool on = !(getenv("E2ETTS_HALF_ROWS") && atoi(getenv("E2ETTS_HALF_ROWS")) == 0);  // tuning aid
  if (!p.wfrag) return false;
  if (under_round_many_rows(p)) return true;
  if (!on || p.accumulate) return false;
  long long t128, rows;
  launch_counts(p, &t128, &rows);
  return tile_half_rows_n(t128, rows, p.Cout);
}

bool epilogue_vec_ok(const ConvParams& p) {
  return (p.Cout % 4 == 0) && (p.out_ld % 4 == 0) && ((p.out_bs & 3) == 0) && (((uintptr_t)p.out & 15) == 0) &&
         (!p.res || ((p.res_ld % 4 == 0) && ((p.res_bs & 3) == 0) && (((uintptr_t)p.res & 15) == 0))) &&
         (!p.bias || (((uintptr_t)p.bias & 15) == 0));
}

}  // namespace

#ifdef E2ETTS_DIAG
void conv_gemm_read_diag(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_diag), sizeof(g_conv_diag)); }
#endif

// the profile class of a launch = the tile configuration launch_conv_gemm picks for it (vector-epilogue shapes)
// The 64-column polyphase upsampler (HiFi-GAN V1's last) in exact fp32: as two 32-column tiles on fragment-order weights each workgroup's
// columns lie in ONE half of the phases, so a third of the MFMAs -- the structurally zero tap of that half -- is skipped
// (zero_tap_split): 0.68 -> 0.605 ms/step.  In the bf16 modes the launch is bound by HBM, not by the matrix pipe, and two column tiles
// read the input twice: 0.395 -> 0.42 ms, so those keep the 256 x 64 tile and multiply the zeros.
static bool narrow_upsampler(const ConvParams& p) {
  return p.x3 == 0 && p.zero_tap_split == 32 && p.Cout == 64 && p.KW == 3 && p.wfrag && !p.accumulate && !p.res;
}

const char* conv_gemm_class(const ConvParams& p) {
  if (narrow_upsampler(p)) return "conv_gemm_256x32";
  if (p.x3) {
    if (few_rows(p)) return "conv_x3_64x64";
    if (half_rows(p)) return "conv_x3_64x128";
    return p.Cout > 64 ? "conv_x3_128x128" : (p.Cout > 32 ? "conv_x3_256x64" : "conv_x3_256x32");
  }
  if (p.Cout > 64) return few_rows(p) ? "conv_gemm_64x64" : (half_rows(p) ? "conv_gemm_64x128" : "conv_gemm_128x128");
  return p.Cout > 32 ? "conv_gemm_256x64" : "conv_gemm_256x32";
}

double conv_gemm_flops(const ConvParams& p) { return 2.0 * p.B * (double)p.T * p.act_frac * p.Cout * p.KW * p.Cin; }

double conv_gemm_bytes(const ConvParams& p) {
  double e = (double)p.B * p.T * p.act_frac * (p.Cin + p.Cout * (1.0 + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0)));
  e += (double)p.Cout * p.KW * p.Cin;
  return 4.0 * e;
}

const char* launch_conv_gemm(const ConvParams& p, hipStream_t s) {
  if (!p.in || !p.w || !p.out) return "conv_gemm: null pointer";
  if (p.B <= 0 || p.T <= 0 || p.Cin <= 0 || p.Cout <= 0 || p.KW <= 0 || p.dil <= 0) return "conv_gemm: bad dims";
  if (p.Cin % 4 || p.in_ld % 4) return "conv_gemm: Cin and in_ld must be multiples of 4";
  if (((uintptr_t)p.in | (uintptr_t)p.w) & 15) return "conv_gemm: in / w must be 16-byte aligned";
  if ((p.in_bs % 4) != 0) return "conv_gemm: in_bs must be a multiple of 4";
  if (p.in_ld < p.Cin || p.out_ld < p.Cout || (p.res && p.res_ld < p.Cout)) return "conv_gemm: row stride < channels";
  if (p.dil * (p.KW - 1) > MAX_HALO) return "conv_gemm: dilation * (KW - 1) exceeds the slab halo limit";
  if (p.pad < 0 || p.pad > p.dil * (p.KW - 1)) return "conv_gemm: pad out of range";
  if (p.out_div != 1.0f && !p.accumulate) return "conv_gemm: out_div needs accumulate (it closes a sum of branches)";
  if (p.zero_tap_split != 0 && (p.KW != 3 || p.zero_tap_split < 0 || p.zero_tap_split >= p.Cout)) return "conv_gemm: zero_tap_split needs KW == 3 and 0 < split < Cout";
  if (p.x3 < 0 || p.x3 > 2) return "conv_gemm: x3 must be 0 (fp32), 1 (bf16x3) or 2 (bf16)";
  if (p.in_slope < 0.f || p.in_slope > 1.f) return "conv_gemm: in_slope must lie in [0, 1]";
  if ((long long)p.T * p.in_ld * 4 >= (1LL << 31) || (long long)p.Cout * p.KW * ((p.Cin + 31) / 32 * 32) * 4 >= (1LL << 31))
    return "conv_gemm: one utterance / the weight matrix must stay below 2 GiB (32-bit buffer offsets)";
  if (!epilogue_vec_ok(p)) {  // odd channel counts / strides: one tile shape per arithmetic mode, scalar epilogue
    if (p.x3 == 1) return launch_cfg_impl<128, 128, 64, 64, 1, false, false, false>(p, s);
    if (p.x3 == 2) return launch_cfg_impl<128, 128, 64, 64, 2, false, false, false>(p, s);
    return launch_cfg_impl<128, 128, 64, 64, 0, false, false, false>(p, s);
  }
  if (narrow_upsampler(p)) return launch_cfg_impl<256, 32, 64, 32, 0, false, true, true>(p, s);
  // few rows (small batches, the B = 1 latency path): 64 x 64 tiles give 4x the workgroups of 128 x 128
  const bool few = few_rows(p), half = half_rows(p);
  if (p.x3 == 1) {
    if (few) return launch_cfg<64, 64, 32, 32, 1>(p, s);
    if (half) return launch_cfg<64, 128, 64, 32, 1>(p, s);
    if (p.Cout > 64) return launch_cfg<128, 128, 64, 64, 1>(p, s);
    if (p.Cout > 32) return launch_cfg<256, 64, 64, 64, 1>(p, s);
    return launch_cfg<256, 32, 64, 32, 1>(p, s);
  }
  if (p.x3 == 2) {
    if (few) return launch_cfg<64, 64, 32, 32, 2>(p, s);
    if (half) return launch_cfg<64, 128, 64, 32, 2>(p, s);
    if (p.Cout > 64) return launch_cfg<128, 128, 64, 64, 2>(p, s);
    if (p.Cout > 32) return launch_cfg<256, 64, 64, 64, 2>(p, s);
    return launch_cfg<256, 32, 64, 32, 2>(p, s);
  }
  if (p.Cout > 64) {
    // (the encoder and the variance adaptor see B * L phonemes, not B * T frames: always few rows)
    if (few) return launch_cfg<64, 64, 32, 32, 0>(p, s);
    if (half) return launch_cfg<64, 128, 64, 32, 0>(p, s);
    return launch_cfg<128, 128, 64, 64, 0>(p, s);
  }
  if (p.Cout > 32) return launch_cfg<256, 64, 64, 64, 0>(p, s);
  return launch_cfg<256, 32, 64, 32, 0>(p, s);
}

}  // namespace e2etts
