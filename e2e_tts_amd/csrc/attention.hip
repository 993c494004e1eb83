// Fused masked multi-head self-attention for one FFT block (fp32, flash-style, never materialises [N, N]).
//
// Replaces reference U/blocks/transformer.py:224-236 + 251-261: head split, bmm(q, k^T) / sqrt(d_k),
// masked_fill(key padding, -inf), softmax over keys, bmm(attn, v), head merge.  The reference materialises
// (2B) x N x N scores (151 MB at B = 32, T = 768); here a workgroup of 4 wavefronts owns 128 queries of one
// (utterance, head) and walks the keys in chunks of 32 staged in LDS, keeping running max / sum per query.
//
// MFMA orientation (v_mfma_f32_32x32x2_f32): S^T = K . Q^T puts the QUERY on the lane (column) and the 32 keys
// of the chunk in the 16 accumulator registers x 2 lane halves.  Then
//   - the softmax row reduction is 16 in-lane ops + one cross-half shuffle (wavefront shuffle, no LDS);
//   - the probabilities are already the B operand of O^T += V^T . P^T: register r of lane half h holds key
//     k0(r) + 4h, k0(r) = (r & 3) + 8 (r >> 2), and MFMA step r consumes exactly the key pair {k0(r), k0(r) + 4}.
#include <math.h>

#include <algorithm>
#include <stdlib.h>

#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

namespace {

// Two workgroups per CU (2 waves / SIMD): at DK = 192 the Q fragments (96 registers) and the O accumulators (96) leave little, and with a
// one-argument __launch_bounds__ hipcc took 380 registers -- ONE wave per SIMD, one workgroup per CU: the 384 workgroups of the decoder
// (B = 32, T = 768) then ran as a round of 256 and a round of 128 with every staging round trip exposed (0.5 ms per layer).
// Workgroup barrier for LDS traffic only.  __syncthreads() is fence + s_barrier and the fence drains EVERY outstanding memory operation
// (s_waitcnt vmcnt(0)) -- including the next chunk's K / V rows the split kernels have just requested, which is the round trip the
// request was issued early to hide.  LDS writes and reads are counted by lgkmcnt alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS written by other lanes of THIS wavefront: order later reads behind the writes (and later writes behind earlier reads) without a
// workgroup barrier.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Key SEGMENTS (round 3).  The exact-fp32 attention kernels run the online softmax per segment of ATT_SEG_CHUNKS x 32 keys -- each segment
// from scratch: running max -inf, sum 0, O 0 -- and merge the finished segments in key order:
//     m' = max(m, m_s);  a = exp(m - m'), b = exp(m_s - m');  l' = l a + l_s b;  O' = O a + O_s b
// (explicit mul / fma: every kernel that merges produces the same bits).  A sequence of up to one segment (the encoder) is computed exactly as
// before (the merge of a single segment multiplies by exp(-inf) = 0 and exp(0) = 1).  What the segments buy: they are independent until
// the merge, so a SMALL grid -- the B = 1 latency path, 24 workgroups walking 24 chunks each -- can hand every segment to a workgroup of
// its own (attention_split_kernel with par_nseg > 0 writes (O_s, m_s, l_s) to a workspace, attention_combine_kernel merges them), while a
// large grid keeps one workgroup per query block that merges in registers: the same operations either way, hence the same bits at every
// batch size (tests: a B = 1 utterance equals its row of the B = 32 batch).
constexpr int ATT_SEG_CHUNKS = 8;
__device__ __forceinline__ void seg_merge_ml(float& m, float& l, const float m_s, const float l_s, float& a, float& b) {
  const float m_new = fmaxf(m, m_s);
  a = expf(m - m_new);      // first segment: m = -inf -> 0
  b = expf(m_s - m_new);
  l = __fmaf_rn(l_s, b, __fmul_rn(l, a));
  m = m_new;
}
__device__ __forceinline__ float seg_merge_o(const float o, const float o_s, const float a, const float b) { return __fmaf_rn(o_s, b, __fmul_rn(o, a)); }

// (utterance b, head, query block qblk) of this workgroup.  Padded grid: (x, y, z) = (query block, head, utterance).  Compact grid of a
// ragged batch whose lengths the host knows (kernels.h: RowMap): 1-D, utterance b owns ceil(len_b / rows) x n_head consecutive blocks,
// heads fastest -- no block of padded queries exists (their output rows keep what they held: every consumer masks them, engine.hip).
#define ATT_BLOCK_OF_GRID(rows)                                    \
  int b, head, qblk;                                               \
  if (rm.n > 0) {                                                  \
    int u_;                                                        \
    if (!rowmap_find(rm, (int)blockIdx.x, b, u_)) return;          \
    const int nh_ = H / DK;                                        \
    head = u_ % nh_;                                               \
    qblk = u_ / nh_;                                               \
  } else {                                                         \
    b = blockIdx.z;                                                \
    head = blockIdx.y;                                             \
    qblk = blockIdx.x;                                             \
  }

template <int DK>
__global__ __launch_bounds__(256, 2) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        const int32_t* __restrict__ lens, int N, int H, float temperature, const RowMap rm) {
  constexpr int LDS_LD = DK + 4;
  constexpr int DT = DK / 32;  // 32-wide tiles of the head dimension
  constexpr int QQ = DK / 8;   // float4 fragments per query row and lane half
  constexpr bool HALVES = DT % 2 == 0;
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDS_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  ATT_BLOCK_OF_GRID(128)
  const int q0 = qblk * 128 + wave * 32;
  const int len = min(lens ? lens[b] : N, N);
  if (qblk * 128 >= len) {  // a tile of padded queries only: their rows are zero (padded grids; a compact grid has no such block)
    for (int i = tid; i < 128 * (DK / 4); i += 256) {
      const int q = qblk * 128 + i / (DK / 4);
      if (q < N) *reinterpret_cast<float4*>(out + ((long long)b * N + q) * H + head * DK + (i % (DK / 4)) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int ld = 3 * H;
  const float* base = qkv + (long long)b * N * ld;
  const float* qp = base + head * DK;
  const float* kp = base + H + head * DK;
  const float* vp = base + 2 * H + head * DK;

  // Q fragments: lane (query li, half lh) holds Q[q][8 qq + 4 lh + r]
  float4 qf[QQ];
  {
    const int qrow = min(q0 + li, N - 1);
    const float* qr = qp + (long long)qrow * ld + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) qf[qq] = *reinterpret_cast<const float4*>(qr + qq * 8);
  }

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  // (o, m_run, l_run) accumulate the current key segment, (ot, m_tot, l_tot) the merged segments before it -- attention_split_kernel's order
  f32x16 ot[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float m_tot = -INFINITY, l_tot = 0.f;

  const int nchunks = (len + 31) / 32;
  for (int sg0 = 0; sg0 < nchunks; sg0 += ATT_SEG_CHUNKS) {   // (two loops: see attention_split_kernel)
  const int sg1 = min(nchunks, sg0 + ATT_SEG_CHUNKS);
  for (int kc = sg0; kc < sg1; ++kc) {
    __syncthreads();
    for (int i = tid; i < 32 * (DK / 4); i += 256) {
      const int r = i / (DK / 4), c = (i % (DK / 4)) * 4;
      const int key = kc * 32 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        kv = *reinterpret_cast<const float4*>(kp + (long long)key * ld + c);
        vv = *reinterpret_cast<const float4*>(vp + (long long)key * ld + c);
      }
      *reinterpret_cast<float4*>(Ks + r * LDS_LD + c) = kv;
      *reinterpret_cast<float4*>(Vs + r * LDS_LD + c) = vv;
    }
    __syncthreads();

    // S^T[key][query] = sum_d K[key][d] Q[query][d].  With an even number of 32-wide head-dim tiles the sum is taken as
    // (d < DK / 2) + (d >= DK / 2): the order attention_split_kernel below produces with one wavefront per half, so the two kernels
    // give the same bits and the launcher may pick by grid size (two independent MFMA chains here, as a side effect).
    f32x16 s, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f, s1[r] = 0.f;
    const float* ka = Ks + li * LDS_LD + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) {
      const float4 a = *reinterpret_cast<const float4*>(ka + qq * 8);
      if (!HALVES || qq < QQ / 2) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[qq].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[qq].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[qq].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[qq].w, s, 0, 0, 0);
      } else {
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[qq].x, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[qq].y, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[qq].z, s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[qq].w, s1, 0, 0, 0);
      }
    }
    if constexpr (HALVES) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = s[r] + s1[r];
    }
    // scale, key-padding mask, online softmax (per query = per lane, both lane halves hold the same query)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kc * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] / temperature;
      v = key < len ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: chunk kc has at least one valid key
    const float corr = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
    // O^T[d][query] = O^T * corr + sum_key V[key][d] P[key][query]
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float a = Vs[key * LDS_LD + d * 32 + li];
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[d], 0, 0, 0);
      }
    }
  }
    {   // the key segment ends (see ATT_SEG_CHUNKS): merge it, start the next from scratch
      if (sg0 == 0) {   // the first segment is the total so far
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[d][r] = o[d][r];
        m_tot = m_run;
        l_tot = l_run;
      } else {
        float ma, mb;
        seg_merge_ml(m_tot, l_tot, m_run, l_run, ma, mb);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[d][r] = seg_merge_o(ot[d][r], o[d][r], ma, mb);
      }
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
      m_run = -INFINITY;
      l_run = 0.f;
    }
  }
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = ot[d][r];
  l_run = l_tot;

  // O^T tile d: column = query (lane & 31), row = head-dim offset (r & 3) + 8 (r >> 2) + 4 lh -> 4 x float4 per tile
  const int q = q0 + li;
  if (q < N) {
    const bool valid = q < len;
    const float inv = valid ? 1.0f / l_run : 0.f;
    float* orow = out + ((long long)b * N + q) * H + head * DK + 4 * lh;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = valid ? o[d][4 * g + 0] * inv : 0.f;
        v.y = valid ? o[d][4 * g + 1] * inv : 0.f;
        v.z = valid ? o[d][4 * g + 2] * inv : 0.f;
        v.w = valid ? o[d][4 * g + 3] * inv : 0.f;
        *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = v;
      }
  }
}

// ---- the same attention for SMALL grids (B = 1: the latency path; small batches): two wavefronts per 32-query tile, each owning HALF of
// the head dimension, and QT = 2 query tiles per workgroup -- four wavefronts, one per SIMD, so that the halved chains really run side
// by side (with 4 tiles x 2 halves on one CU each SIMD carried two half-chains: nothing gained) and twice the workgroups spread over CUs.  In attention_kernel one wavefront walks the keys of its 32 queries alone -- per 32-key chunk
// 2 x DK / 2 dependent v_mfma_f32_32x32x2_f32 (192 at DK = 192: 12 k cycles = 5 us) -- and at B = 1, T = 768 only 48 wavefronts exist:
// 24 chunks x 9 us = 225 us per decoder layer on 12 of 256 CUs.  Here the pair (w, w ^ 1) splits it: each computes the partial S^T over
// its DK / 2 channels (Q fragments: half the registers), the two partial tiles meet in LDS and BOTH add them in the order
// (d < DK / 2) + (d >= DK / 2) -- the order attention_kernel uses -- run the same softmax, and accumulate O^T for their own DK / 64
// head-dim tiles.  Same bits as attention_kernel, half the chain, twice the wavefronts; and with half the registers there is room to
// request the next chunk's K / V rows before this chunk's arithmetic (the staging round trip was exposed once per chunk).
template <int DK, int QT>
__global__ __launch_bounds__(QT * 128, 2) void attention_split_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                 const int32_t* __restrict__ lens, int N, int H, float temperature, const RowMap rm,
                                                                 float* __restrict__ ws_o, float* __restrict__ ws_ml, const int par_nseg) {
  constexpr int LDS_LD = DK + 4;
  constexpr int DH = DK / 2;    // head-dim channels per wavefront
  constexpr int DTH = DH / 32;  // its 32-wide tiles
  constexpr int QQH = DH / 8;   // its float4 Q fragments per lane
  static_assert(DK % 64 == 0, "the split form needs an even number of 32-wide head-dim tiles");
  constexpr int NF4 = 32 * (DK / 4);          // float4 per staged K (or V) chunk
  constexpr int NTH = QT * 128;            // threads: QT query tiles x 2 halves x 64 lanes
  constexpr int NLD = (NF4 + NTH - 1) / NTH;      // ... per thread
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Sx[QT * 2][16 * 64];  // each wavefront's partial S^T tile, register-major

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int qt = wave >> 1, part = wave & 1;
  ATT_BLOCK_OF_GRID(QT * 32)
  int seg_id = 0;   // par_nseg > 0 (padded grids only): x = query block x segments; this workgroup computes ONE key segment's partial result
  if (par_nseg > 0) {
    seg_id = qblk % par_nseg;
    qblk /= par_nseg;
  }
  const int q0 = qblk * (QT * 32) + qt * 32;
  const int len = min(lens ? lens[b] : N, N);
  if (par_nseg > 0 && qblk * (QT * 32) >= len) return;   // attention_combine_kernel writes the zero rows
  if (qblk * (QT * 32) >= len) {  // a tile of padded queries only: their rows are zero (padded grids)
    for (int i = tid; i < QT * 32 * (DK / 4); i += NTH) {
      const int q = qblk * (QT * 32) + i / (DK / 4);
      if (q < N) *reinterpret_cast<float4*>(out + ((long long)b * N + q) * H + head * DK + (i % (DK / 4)) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int ld = 3 * H;
  const float* base = qkv + (long long)b * N * ld;
  const float* qp = base + head * DK + part * DH;
  const float* kp = base + H + head * DK;
  const float* vp = base + 2 * H + head * DK;

  float4 qf[QQH];
  {
    const int qrow = min(q0 + li, N - 1);
    const float* qr = qp + (long long)qrow * ld + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQH; ++qq) qf[qq] = *reinterpret_cast<const float4*>(qr + qq * 8);
  }
  f32x16 o[DTH];
#pragma unroll
  for (int d = 0; d < DTH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  float4 kreg[NLD], vreg[NLD];
  auto fetch = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = min(tid + i * NTH, NF4 - 1);
      const int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
      const int key = min(kc * 32 + r, N - 1);   // clamped: rows past N are zeroed when they are written to LDS
      kreg[i] = *reinterpret_cast<const float4*>(kp + (long long)key * ld + c);
      vreg[i] = *reinterpret_cast<const float4*>(vp + (long long)key * ld + c);
    }
  };
  auto stash = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * NTH;
      if (idx < NF4) {
        const int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
        const bool ok = kc * 32 + r < N;
        // (component by component: a select between two float4 objects went through scratch memory)
        const float4 kv = make_float4(ok ? kreg[i].x : 0.f, ok ? kreg[i].y : 0.f, ok ? kreg[i].z : 0.f, ok ? kreg[i].w : 0.f);
        const float4 vv = make_float4(ok ? vreg[i].x : 0.f, ok ? vreg[i].y : 0.f, ok ? vreg[i].z : 0.f, ok ? vreg[i].w : 0.f);
        *reinterpret_cast<float4*>(Ks + r * LDS_LD + c) = kv;
        *reinterpret_cast<float4*>(Vs + r * LDS_LD + c) = vv;
      }
    }
  };

  const int nchunks = (len + 31) / 32;
  // (o, m_run, l_run) accumulate the CURRENT key segment; (ot, m_tot, l_tot) the merged segments before it (serial form)
  const int kc_begin = par_nseg > 0 ? seg_id * ATT_SEG_CHUNKS : 0;
  const int kc_end = par_nseg > 0 ? min(nchunks, (seg_id + 1) * ATT_SEG_CHUNKS) : nchunks;
  if (kc_begin >= kc_end) return;   // a segment beyond this utterance's keys (uniform for the workgroup, before any barrier)
  f32x16 ot[DTH];
#pragma unroll
  for (int d = 0; d < DTH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float m_tot = -INFINITY, l_tot = 0.f;
  fetch(kc_begin);
  // (two loops on purpose: the inner one never touches `ot`, so the register allocator may park the merged segments outside it -- as one
  // flat loop with the merge under a condition it spilled and reloaded them in EVERY chunk, and each scratch access drains the K / V prefetch)
  for (int sg0 = kc_begin; sg0 < kc_end; sg0 += ATT_SEG_CHUNKS) {
  const int sg1 = min(kc_end, sg0 + ATT_SEG_CHUNKS);
  for (int kc = sg0; kc < sg1; ++kc) {
    lds_barrier();  // every wavefront is done with the previous chunk's K / V
    stash(kc);
    lds_barrier();
    if (kc + 1 < kc_end) fetch(kc + 1);

    // partial S^T over this wavefront's half of the head dimension
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* ka = Ks + li * LDS_LD + part * DH + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQH; ++qq) {
      const float4 a = *reinterpret_cast<const float4*>(ka + qq * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[qq].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[qq].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[qq].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[qq].w, s, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Sx[wave][r * 64 + lane] = s[r];
    lds_barrier();
    {
      const float* s0 = Sx[wave & ~1];
      const float* s1 = Sx[wave | 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = s0[r * 64 + lane] + s1[r * 64 + lane];
    }
    // scale, key-padding mask, online softmax: as in attention_kernel, computed by both wavefronts of the pair
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kc * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] / temperature;
      v = key < len ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DTH; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float a = Vs[key * LDS_LD + (part * DTH + d) * 32 + li];
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[d], 0, 0, 0);
      }
    }
  }
    if (par_nseg == 0) {   // the segment ends: merge it, start the next one from scratch
      if (sg0 == kc_begin) {   // the first segment IS the total so far (every merging kernel assigns here: no parked registers to fetch)
#pragma unroll
        for (int d = 0; d < DTH; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[d][r] = o[d][r];
        m_tot = m_run;
        l_tot = l_run;
      } else {
        float ma, mb;
        seg_merge_ml(m_tot, l_tot, m_run, l_run, ma, mb);
#pragma unroll
        for (int d = 0; d < DTH; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[d][r] = seg_merge_o(ot[d][r], o[d][r], ma, mb);
      }
#pragma unroll
      for (int d = 0; d < DTH; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
      m_run = -INFINITY;
      l_run = 0.f;
    }
  }

  const int q = q0 + li;
  if (par_nseg > 0) {  // this segment's unnormalised partial result: O_s [.., N, DK] and (m_s, l_s) [.., N, 2] per (utterance, head, segment)
    if (q < N) {
      const long long row = (((long long)b * (H / DK) + head) * par_nseg + seg_id) * N + q;
      float* orow = ws_o + row * DK + part * DH + 4 * lh;
#pragma unroll
      for (int d = 0; d < DTH; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = make_float4(o[d][4 * g], o[d][4 * g + 1], o[d][4 * g + 2], o[d][4 * g + 3]);
      if (part == 0 && lh == 0) *reinterpret_cast<float2*>(ws_ml + row * 2) = make_float2(m_run, l_run);
    }
    return;
  }
#pragma unroll
  for (int d = 0; d < DTH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = ot[d][r];
  l_run = l_tot;
  if (q < N) {
    const bool valid = q < len;
    const float inv = valid ? 1.0f / l_run : 0.f;
    float* orow = out + ((long long)b * N + q) * H + head * DK + part * DH + 4 * lh;
#pragma unroll
    for (int d = 0; d < DTH; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = valid ? o[d][4 * g + 0] * inv : 0.f;
        v.y = valid ? o[d][4 * g + 1] * inv : 0.f;
        v.z = valid ? o[d][4 * g + 2] * inv : 0.f;
        v.w = valid ? o[d][4 * g + 3] * inv : 0.f;
        *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = v;
      }
  }
}

// Merges the key segments attention_split_kernel left in the workspace (par_nseg > 0): per (utterance, head, query) the segments in key
// order with seg_merge_*, then the normalisation and the zero rows of padded queries, as the kernels' own epilogues write them.
__global__ void attention_combine_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml, float* __restrict__ out,
                                         const int32_t* __restrict__ lens, int N, int H, int DK, int par_nseg, long long total4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (b, q, head, float4 of the head dimension)
  if (i >= total4) return;
  const int d4 = DK / 4, nh = H / DK;
  const int c4 = (int)(i % d4);
  const int head = (int)((i / d4) % nh);
  const long long bq = i / ((long long)d4 * nh);
  const int q = (int)(bq % N), b = (int)(bq / N);
  const int len = min(lens ? lens[b] : N, N);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < len) {
    const int nseg = ((len + 31) / 32 + ATT_SEG_CHUNKS - 1) / ATT_SEG_CHUNKS;
    float m = -INFINITY, l = 0.f;
    for (int sg = 0; sg < nseg; ++sg) {
      const long long row = (((long long)b * nh + head) * par_nseg + sg) * N + q;
      const float2 ml = *reinterpret_cast<const float2*>(ws_ml + row * 2);
      const float4 os = *reinterpret_cast<const float4*>(ws_o + row * DK + c4 * 4);
      if (sg == 0) {   // the first segment is the total so far (as in the kernels that merge in registers)
        m = ml.x; l = ml.y; o = os;
      } else {
        float ma, mb;
        seg_merge_ml(m, l, ml.x, ml.y, ma, mb);
        o.x = seg_merge_o(o.x, os.x, ma, mb); o.y = seg_merge_o(o.y, os.y, ma, mb);
        o.z = seg_merge_o(o.z, os.z, ma, mb); o.w = seg_merge_o(o.w, os.w, ma, mb);
      }
    }
    const float inv = 1.0f / l;
    o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
  }
  *reinterpret_cast<float4*>(out + ((long long)b * N + q) * H + head * DK + c4 * 4) = o;
}

// ---- Conformer: relative-position attention (reference U/blocks/conformer.py:399-440), fp32.  Same flash-style structure and MFMA
// orientation as attention_kernel; differences: u_bias is added to the query fragments; nothing is masked (the block calls the module
// without a mask, :252, so padded keys take part and padded queries are computed); the score gets the SHIFTED position term before it
// is divided by sqrt(d_model) (:384, :418).  DK need not be a multiple of 32: the P . V product runs on ceil(DK / 32) tiles with V
// zero-padded in LDS.
//
// The position term, computed HERE (round 3; rounds 1-2 materialised the unshifted scores (q + v) . P^T as a [B, heads, N, N] tensor --
// 604 MB at B = 32, T = 768 -- with one GEMM per head and gathered the shifted entries 4 bytes at a time).  _relative_shift (:432-440)
// is the Transformer-XL reshape applied to a non-causal score matrix; index for index it reads
//     (i, j <= i)     -> (q_i + v)     . P[N - 1 - i + j]
//     (i, j == i + 1) -> 0
//     (i, j >= i + 2) -> (q_{i+1} + v) . P[j - i - 2]
// i.e. the P row is a function of j - i alone.  For a tile of 32 queries i0 + a and 32 keys j0 + c the rows needed are a BAND of 63
// consecutive rows of P, indexed by t = c - a + 31: the wavefront stages that band (64 rows) in its own LDS region, computes
// G^T[t][a] = P[band row t] . (q_a + v) with 2 x DK / 2 MFMAs -- the same orientation as S^T = K . Q^T -- and reads the entry each
// (query, key) pair needs back through LDS with the skew t = c - a + 31.  Tiles below the diagonal take the first form only, tiles
// above it the third (with the NEXT query's vector, a second set of fragments), the diagonal tile computes both bands and selects.
template <int DK>
__global__ __launch_bounds__(256, 2) void rel_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ pos, int pos_rows,
                                                               const float* __restrict__ ub, const float* __restrict__ vb,
                                                               float* __restrict__ out, int N, int H, float temperature) {
  static_assert(DK % 8 == 0, "head dim must be a multiple of 8");
  constexpr int DT = (DK + 31) / 32;
  constexpr int LDS_LD = DT * 32 + 4;
  constexpr int QQ = DK / 8;
  constexpr int GLD = 68;               // G row stride (query-major [a][t], 64 + 4): float4 writes of 16 rows cover the 64 banks once, and
                                        // the skewed read (row a, column c - a + 31) walks 3 a + const: distinct banks
  constexpr int NPRE = QQ;              // float4 fragments of one band tile per lane
  __shared__ __attribute__((aligned(16))) float Ks[32 * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Vs[32 * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Gt[4][32 * GLD];   // per wavefront: G of the current band, [query a][t 0..63]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int i0 = blockIdx.x * 128 + wave * 32;
  const int ld = 3 * H;
  const float* base = qkv + (long long)b * N * ld;
  const float* qp = base + head * DK;
  const float* kp = base + H + head * DK;
  const float* vp = base + 2 * H + head * DK;
  const float* ph = pos + (long long)head * pos_rows * DK;   // this head's projected position table [pos_rows][DK]
  float* const gt = Gt[wave];

  const int qi = min(i0 + li, N - 1);  // this lane's query (clamped: rows >= N are computed and dropped)
  // q + u (content term) and ONE set of position fragments: q + v while the chunks lie at or below the diagonal (j <= i), reloaded as
  // q_{i+1} + v when the upper form starts at the diagonal tile (j >= i + 2) -- two sets at once cost 24 registers and spilled
  float4 qf[QQ], qx[QQ];
  auto load_qx = [&](const int row) __attribute__((always_inline)) {
    const float* qr = qp + (row * ld + lh * 4);
    const float* vr = vb + head * DK + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) {
      const float4 a = *reinterpret_cast<const float4*>(qr + qq * 8), v4 = *reinterpret_cast<const float4*>(vr + qq * 8);
      qx[qq] = make_float4(a.x + v4.x, a.y + v4.y, a.z + v4.z, a.w + v4.w);
    }
  };
  {
    const float* qr = qp + (qi * ld + lh * 4);
    const float* ur = ub + head * DK + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) {
      const float4 a = *reinterpret_cast<const float4*>(qr + qq * 8), u4 = *reinterpret_cast<const float4*>(ur + qq * 8);
      qf[qq] = make_float4(a.x + u4.x, a.y + u4.y, a.z + u4.z, a.w + u4.w);
    }
  }
  load_qx(qi);
  // zero the V padding columns once (they are never overwritten)
  if (DK % 32) {
    for (int i = tid; i < 32 * (DT * 32 - DK); i += 256) {
      const int r = i / (DT * 32 - DK), c = DK + i % (DT * 32 - DK);
      Vs[r * LDS_LD + c] = 0.f;
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // The band SLIDES: the next key chunk's band is this one's shifted by 32 table rows, so its first G^T tile (t < 32) is this chunk's
  // second one (t >= 32) -- P[row] . x does not depend on the chunk.  Per chunk ONE new tile of 32 table rows is multiplied (both tiles
  // only where a sequence starts: the first chunk, and the upper form at the diagonal).  The table rows are the A operand of that
  // product, and each lane fetches exactly ITS fragments -- row li of the tile, channels 8 qq + 4 lh .. + 3 -- straight from the table
  // (L2-resident: N x DK floats per head) into registers, one chunk ahead (`pre`): no LDS staging, and the trip through L2 sits behind
  // the previous chunk's arithmetic.
  typedef f32x4v PRE_T;   // a native vector (float4 is a struct of unions: an array of them carried around the chunk loop stayed in scratch memory)
  auto band_fetch = [&](PRE_T (&pre)[NPRE], const int row_start) __attribute__((always_inline)) {
    const int row = min(max(row_start + li, 0), pos_rows - 1);   // rows outside [0, N) feed entries nobody selects
    const float* pr = ph + (row * DK + lh * 4);
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) pre[qq] = *reinterpret_cast<const f32x4v*>(pr + qq * 8);
  };
  // G^T tile of the 32 table rows in `pre`: [t][a] = P[row_start + t] . x_a (32 x 32, the orientation of S^T = K . Q^T)
  auto band_tile = [&](const PRE_T (&pre)[NPRE], const float4 (&x)[QQ]) __attribute__((always_inline)) -> f32x16 {
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) {
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(pre[qq].x, x[qq].x, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(pre[qq].y, x[qq].y, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(pre[qq].z, x[qq].z, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(pre[qq].w, x[qq].w, g, 0, 0, 0);
    }
    return g;
  };
  // lane (query a = li, half lh) gets the 16 entries of its keys c = (r & 3) + 8 (r >> 2) + 4 lh: G[a][c - a + 31], through LDS.  A lane
  // holds G^T[t][a] for t = (r & 3) + 8 (r >> 2) + 4 lh: four consecutive t per register quad -> one float4 write into row a.
  // diag: the upper form on the diagonal tile -- keep the lower entry where j <= i, 0 at j == i + 1
  auto skew = [&](const f32x16& g0, const f32x16& g1, float (&term)[16], const bool diag) __attribute__((always_inline)) {
    wave_lds_fence();   // the previous band's skewed reads are complete
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int t = 8 * q4 + 4 * lh;
      *reinterpret_cast<float4*>(gt + li * GLD + t) = make_float4(g0[4 * q4], g0[4 * q4 + 1], g0[4 * q4 + 2], g0[4 * q4 + 3]);
      *reinterpret_cast<float4*>(gt + li * GLD + 32 + t) = make_float4(g1[4 * q4], g1[4 * q4 + 1], g1[4 * q4 + 2], g1[4 * q4 + 3]);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = gt[li * GLD + (c - li + 31)];
      const bool keep = diag && c <= li, zero = diag && c == li + 1;
      term[r] = keep ? term[r] : (zero ? 0.f : v);
    }
  };
  // first table row of the NEW tile (t >= 32) of the band of key chunk j0: lower form (j0 <= i0) rows N - 1 - i + j, upper form j - i - 2
  auto new_tile_row = [&](const int j0, const bool upper) { return upper ? j0 - i0 - 1 : N + j0 - i0; };
  PRE_T pre[NPRE];   // the table rows requested for the next band tile (registers: handed to the lambdas as an argument -- captured by
                     // reference it stayed in scratch memory)
  f32x16 ghi;   // the second tile of the last band of the running sequence = the first tile of the next band
#pragma unroll
  for (int r = 0; r < 16; ++r) ghi[r] = 0.f;

  const int nchunks = (N + 31) / 32;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int j0 = kc * 32;
    __syncthreads();
    for (int i = tid; i < 32 * (DK / 4); i += 256) {
      const int r = i / (DK / 4), c = (i % (DK / 4)) * 4;
      const int key = j0 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        kv = *reinterpret_cast<const float4*>(kp + (key * ld + c));
        vv = *reinterpret_cast<const float4*>(vp + (key * ld + c));
      }
      *reinterpret_cast<float4*>(Ks + r * LDS_LD + c) = kv;
      *reinterpret_cast<float4*>(Vs + r * LDS_LD + c) = vv;
    }
    // shifted position term of this lane's 16 (query, key) pairs (wave-uniform case split on j0 - i0; both are multiples of 32)
    float pterm[16];
    if (j0 <= i0) {         // lower form, P row N - 1 - i + j = (N - 1 + j0 - i0 - 31) + t  (every pair below the diagonal; at j0 == i0: j <= i)
      if (kc == 0) {        // the sequence starts: both tiles
        band_fetch(pre, new_tile_row(j0, false) - 32);
        ghi = band_tile(pre, qx);
        band_fetch(pre, new_tile_row(j0, false));
      }
      const f32x16 g1 = band_tile(pre, qx);
      if (j0 < i0) band_fetch(pre, new_tile_row(j0 + 32, false));   // the next chunk continues the lower sequence
      skew(ghi, g1, pterm, false);
      ghi = g1;
    }
    if (j0 >= i0) {         // upper form, P row j - i - 2 = (j0 - i0 - 33) + t  (j >= i + 2; j == i + 1 -> 0)
      if (j0 == i0) {       // the sequence starts at the diagonal tile: the NEXT query's vector from here on, both tiles
        load_qx(min(qi + 1, N - 1));
        band_fetch(pre, new_tile_row(j0, true) - 32);
        ghi = band_tile(pre, qx);
        band_fetch(pre, new_tile_row(j0, true));
      }
      const f32x16 g1 = band_tile(pre, qx);
      if (kc + 1 < nchunks) band_fetch(pre, new_tile_row(j0 + 32, true));
      skew(ghi, g1, pterm, j0 == i0);
      ghi = g1;
      if (j0 == i0 + 32 && li == 31 && lh == 0) pterm[0] = 0.f;   // (a, c) = (31, 0): j == i + 1
    }
    __syncthreads();

    f32x16 s;   // the content scores accumulate on top of the position term (one array instead of two)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = pterm[r];
    const float* ka = Ks + li * LDS_LD + lh * 4;
#pragma unroll
    for (int qq = 0; qq < QQ; ++qq) {
      const float4 a = *reinterpret_cast<const float4*>(ka + qq * 8);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[qq].x, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[qq].y, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[qq].z, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[qq].w, s, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] / temperature;
      v = key < N ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: every chunk has a key < N
    const float corr = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float a = Vs[key * LDS_LD + d * 32 + li];
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[d], 0, 0, 0);
      }
    }
  }

  const int q = i0 + li;
  if (q < N) {
    const float inv = 1.0f / l_run;
    float* orow = out + ((long long)b * N + q) * H + head * DK + 4 * lh;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (d * 32 + 8 * g + 4 * lh < DK) {
          float4 v;
          v.x = o[d][4 * g + 0] * inv; v.y = o[d][4 * g + 1] * inv; v.z = o[d][4 * g + 2] * inv; v.w = o[d][4 * g + 3] * inv;
          *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = v;
        }
      }
  }
}

// ---- split-precision ("bf16x3") form, for the decoder when its FFT blocks run in that arithmetic (the encoder keeps the
// fp32 kernel above: its output decides durations and buckets).  Same flash-style structure and the same orientation trick,
// on v_mfma_f32_32x32x16_bf16: every fp32 operand is hi + lo (two bf16) and a product keeps lo*hi + hi*lo + hi*hi.
//   S^T = K . Q^T : A = K rows from LDS as [key][hi d | lo d], B = Q fragments in registers (lane = query, 8 consecutive d).
//   O^T += V^T . P^T : the lane already holds P for keys k(r) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the order of k inside an
//   MFMA is free as long as A agrees, so V is staged TRANSPOSED with its keys permuted into that order:
//   Vt[d][slot], slot = 16 s + 8 (lane >> 5) + e  <->  key k(8 s + e) + 4 (lane >> 5): the A fragment is one 16-byte read.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
  const bf16x2_t r = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, r);
}
// 8 floats -> hi and lo fragments (8 bf16 each)
__device__ __forceinline__ void split8(const float* v, bf16x8_t& hi, bf16x8_t& lo) {
  uint4 h, l;
  unsigned* hp = reinterpret_cast<unsigned*>(&h);
  unsigned* lp = reinterpret_cast<unsigned*>(&l);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hp[i] = pk_bf16(v[2 * i], v[2 * i + 1]);
    const float h0 = __builtin_bit_cast(float, hp[i] << 16), h1 = __builtin_bit_cast(float, hp[i] & 0xffff0000u);
    lp[i] = pk_bf16(v[2 * i] - h0, v[2 * i + 1] - h1);
  }
  hi = __builtin_bit_cast(bf16x8_t, h);
  lo = __builtin_bit_cast(bf16x8_t, l);
}

// NW: wavefronts (32 queries each) per workgroup.  Every workgroup of an (utterance, head) pulls ALL its K / V chunks through L2, and that
// traffic, not arithmetic, is what the staging costs: 8 waves = 256 queries per workgroup halve it against 4.
template <int DK, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attention_x3_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           const int32_t* __restrict__ lens, int N, int H, float temperature, const RowMap rm) {
  constexpr int KS = DK + 4;   // words per K row: DK/2 (hi bf16) + DK/2 (lo bf16) + 4 pad; KS mod 64 == 4 -> conflict-free b128
  constexpr int VS = 36;       // words per Vt row: 16 (32 hi slots) + 16 (32 lo slots) + 4 pad
  constexpr int DT = DK / 32;  // 32-wide tiles of the head dimension
  constexpr int NS = DK / 16;  // k-steps of S^T = K . Q^T
  constexpr bool HALVES = DT % 2 == 0;
  static_assert(DK % 32 == 0, "head dim must be a multiple of 32");
  __shared__ __attribute__((aligned(16))) unsigned Kh[32 * KS];
  __shared__ __attribute__((aligned(16))) unsigned Vt[DK * VS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  ATT_BLOCK_OF_GRID(NW * 32)
  const int q0 = qblk * (NW * 32) + wave * 32;
  const int len = min(lens ? lens[b] : N, N);
  if (qblk * (NW * 32) >= len) {  // a tile of padded queries only: their rows are zero (padded grids)
    for (int i = tid; i < NW * 32 * (DK / 4); i += NW * 64) {
      const int q = qblk * (NW * 32) + i / (DK / 4);
      if (q < N) *reinterpret_cast<float4*>(out + ((long long)b * N + q) * H + head * DK + (i % (DK / 4)) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int ld = 3 * H;
  const float* base = qkv + (long long)b * N * ld;
  const float* qp = base + head * DK;
  const float* kp = base + H + head * DK;
  const float* vp = base + 2 * H + head * DK;

  // Q fragments: lane (query li, half lh) holds Q[q][16 s + 8 lh .. + 7] for every k-step s, split once
  bf16x8_t qh[NS], ql[NS];
  {
    const int qrow = min(q0 + li, N - 1);
    const float* qr = qp + (long long)qrow * ld + lh * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(qr + s * 16), c = *reinterpret_cast<const float4*>(qr + s * 16 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      split8(v, qh[s], ql[s]);
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float inv_temp = 1.0f / temperature;  // split-precision path: a multiply and the hardware exp are within its error budget

  const int nchunks = (len + 31) / 32;
  for (int kc = 0; kc < nchunks; ++kc) {
    __syncthreads();
    for (int i = tid; i < 32 * (DK / 4); i += NW * 64) {
      const int r = i / (DK / 4), c4 = i % (DK / 4);
      const int key = kc * 32 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        kv = *reinterpret_cast<const float4*>(kp + (key * ld + c4 * 4));
        vv = *reinterpret_cast<const float4*>(vp + (key * ld + c4 * 4));
      }
      // K row: [hi d 0 .. DK-1 | lo d 0 .. DK-1] as bf16
      const unsigned k0 = pk_bf16(kv.x, kv.y), k1 = pk_bf16(kv.z, kv.w);
      const float kx = __builtin_bit_cast(float, k0 << 16), ky = __builtin_bit_cast(float, k0 & 0xffff0000u);
      const float kz = __builtin_bit_cast(float, k1 << 16), kw = __builtin_bit_cast(float, k1 & 0xffff0000u);
      *reinterpret_cast<uint2*>(Kh + r * KS + c4 * 2) = make_uint2(k0, k1);
      *reinterpret_cast<uint2*>(Kh + r * KS + DK / 2 + c4 * 2) = make_uint2(pk_bf16(kv.x - kx, kv.y - ky), pk_bf16(kv.z - kz, kv.w - kw));
      // V transposed, key r -> slot: r = k(rr) + 4 h with rr = (r & 3) + 4 (r >> 3), h = (r >> 2) & 1; slot = 16 (rr >> 3) + 8 h + (rr & 7)
      const int rr = (r & 3) + 4 * (r >> 3), h = (r >> 2) & 1;
      const int slot = 16 * (rr >> 3) + 8 * h + (rr & 7);
      // head-dim row d = 4 c4 + e lives at physical row (d & 3) (DK / 4) + (d >> 2) = e (DK / 4) + c4: consecutive lanes
      // (consecutive c4) then write consecutive rows, 36 words apart.  Stored by row d they were 4 rows = 144 words = 16 banks apart.
      unsigned short* vt = reinterpret_cast<unsigned short*>(Vt) + c4 * (VS * 2) + slot;
      const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = (__bf16)vs[e];
        const __bf16 lb = (__bf16)(vs[e] - (float)hb);
        vt[e * (DK / 4) * (VS * 2)] = __builtin_bit_cast(unsigned short, hb);
        vt[e * (DK / 4) * (VS * 2) + 32] = __builtin_bit_cast(unsigned short, lb);
      }
    }
    __syncthreads();

    // S^T[key][query] = sum_d K[key][d] Q[query][d]
    // (with an even number of head-dim tiles: as (d < DK / 2) + (d >= DK / 2), the order of attention_x3_split_kernel -- see the fp32 kernels)
    f32x16 s, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f, s1[r] = 0.f;
    const unsigned* ka = Kh + li * KS + lh * 4;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ka + ks * 8));
      const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ka + DK / 2 + ks * 8));
      if (!HALVES || ks < NS / 2) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[ks], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[ks], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[ks], s, 0, 0, 0);
      } else {
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[ks], s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[ks], s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[ks], s1, 0, 0, 0);
      }
    }
    if constexpr (HALVES) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = s[r] + s1[r];
    }
    // scale, key-padding mask, online softmax (per query = per lane, both lane halves hold the same query)
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kc * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] * inv_temp;
      v = key < len ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: chunk kc has at least one valid key
    const float corr = __expf(m_run - m_new);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __expf(s[r] - m_new);
      psum += pv[r];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
    // P fragments: k-step s2 takes registers 8 s2 .. 8 s2 + 7 (their keys are what the Vt slots of that step hold)
    bf16x8_t ph[2], pl[2];
    split8(pv, ph[0], pl[0]);
    split8(pv + 8, ph[1], pl[1]);
    // O^T[d][query] = O^T * corr + sum_key V[key][d] P[key][query]
#pragma unroll
    for (int d = 0; d < DT; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= corr;
      const unsigned* va = Vt + ((li & 3) * (DK / 4) + d * 8 + (li >> 2)) * VS + lh * 4;  // physical row of head-dim row d * 32 + li
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(va + s2 * 8));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(va + 16 + s2 * 8));
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pl[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph[s2], o[d], 0, 0, 0);
      }
    }
  }

  // O^T tile d: column = query (lane & 31), row = head-dim offset (r & 3) + 8 (r >> 2) + 4 lh -> 4 x float4 per tile
  const int q = q0 + li;
  if (q < N) {
    const bool valid = q < len;
    const float inv = valid ? 1.0f / l_run : 0.f;
    float* orow = out + ((long long)b * N + q) * H + head * DK + 4 * lh;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = valid ? o[d][4 * g + 0] * inv : 0.f;
        v.y = valid ? o[d][4 * g + 1] * inv : 0.f;
        v.z = valid ? o[d][4 * g + 2] * inv : 0.f;
        v.w = valid ? o[d][4 * g + 3] * inv : 0.f;
        *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = v;
      }
  }
}

// ---- Conformer relative-position attention in split precision (the decoder when its GEMMs run in bf16x3): rel_attention_kernel's
// structure -- content scores, the position BAND of 64 table rows per (32-query, 32-key) tile, G^T = band . (q + v)^T, the skewed read --
// on attention_x3_kernel's operands: every fp32 operand is hi + lo (two bf16) and a product keeps lo*hi + hi*lo + hi*hi on
// v_mfma_f32_32x32x16_bf16.  The projected position table arrives pre-split from the packer (`att.pos.x3`: rows of DK bf16 hi | DK bf16
// lo), so staging a band is a copy.  DK % 16 == 0; P . V runs on ceil(DK / 32) tiles (the rows past DK are never stored).
template <int DK>
__global__ __launch_bounds__(256, 2) void rel_attention_x3_kernel(const float* __restrict__ qkv, const unsigned* __restrict__ posx, int pos_rows,
                                                                  const float* __restrict__ ub, const float* __restrict__ vb,
                                                                  float* __restrict__ out, int N, int H, float temperature) {
  static_assert(DK % 16 == 0, "head dim must be a multiple of 16");
  constexpr int KS = DK + 4;            // words per K / band row: DK/2 (hi bf16) + DK/2 (lo bf16) + 4 pad
  constexpr int VS = 36;                // words per Vt row: 16 (32 hi slots) + 16 (32 lo slots) + 4 pad
  constexpr int DTP = (DK + 31) / 32;   // 32-wide tiles of the (padded) head dimension
  constexpr int NS = DK / 16;           // k-steps over the head dimension
  constexpr int GLD = 68;               // rel_attention_kernel's exchange layout: [query a][t 0..63], 64 + 4
  constexpr int NPRE = 2 * NS;          // uint4 fragments of one band tile per lane: hi and lo of every k-step
  __shared__ __attribute__((aligned(16))) unsigned Kh[32 * KS];
  __shared__ __attribute__((aligned(16))) unsigned Vt[DTP * 32 * VS];
  __shared__ __attribute__((aligned(16))) float Gt[4][32 * GLD];     // per wavefront: G of the current band

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y;
  const int i0 = blockIdx.x * 128 + wave * 32;
  const int ld = 3 * H;
  const float* base = qkv + (long long)b * N * ld;
  const float* qp = base + head * DK;
  const float* kp = base + H + head * DK;
  const float* vp = base + 2 * H + head * DK;
  const unsigned* ph = posx + (long long)head * pos_rows * DK;   // this head's table: [pos_rows][DK words]
  float* const gt = Gt[wave];

  // query fragments: lane (query li, half lh) holds X[q][16 s + 8 lh .. + 7] for every k-step s, split once: q + u, and ONE set of
  // position fragments -- q + v up to the diagonal, q_{i+1} + v from the diagonal tile's upper form on (rel_attention_kernel)
  const int qi = min(i0 + li, N - 1);
  bf16x8_t fh[NS], fl[NS], xh[NS], xl[NS];
  auto load_frag = [&](const int row, const float* bias, bf16x8_t (&dh)[NS], bf16x8_t (&dl)[NS]) __attribute__((always_inline)) {
    const float* qr = qp + (row * ld + lh * 8);
    const float* br = bias + head * DK + lh * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(qr + s * 16), a1 = *reinterpret_cast<const float4*>(qr + s * 16 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(br + s * 16), b1 = *reinterpret_cast<const float4*>(br + s * 16 + 4);
      const float t8[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w, a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
      split8(t8, dh[s], dl[s]);
    }
  };
  load_frag(qi, ub, fh, fl);
  load_frag(qi, vb, xh, xl);

  f32x16 o[DTP];
#pragma unroll
  for (int d = 0; d < DTP; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float inv_temp = 1.0f / temperature;

  // the sliding band of rel_attention_kernel: one new tile of 32 (pre-split) table rows per key chunk, each lane fetching its own A
  // fragments (row li: bf16 hi d = 16 ks + 8 lh .. + 7 and the matching lo) from the table into registers a chunk ahead
  typedef u32x4v PRE_T;
  auto band_fetch = [&](PRE_T (&pre)[NPRE], const int row_start) __attribute__((always_inline)) {
    const int row = min(max(row_start + li, 0), pos_rows - 1);
    const unsigned* pr = ph + (row * DK + lh * 4);
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      pre[2 * ks] = *reinterpret_cast<const u32x4v*>(pr + ks * 8);
      pre[2 * ks + 1] = *reinterpret_cast<const u32x4v*>(pr + DK / 2 + ks * 8);
    }
  };
  auto band_tile = [&](const PRE_T (&pre)[NPRE]) __attribute__((always_inline)) -> f32x16 {
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, pre[2 * ks]);
      const bf16x8_t al = __builtin_bit_cast(bf16x8_t, pre[2 * ks + 1]);
      g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[ks], g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[ks], g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[ks], g, 0, 0, 0);
    }
    return g;
  };
  auto skew = [&](const f32x16& g0, const f32x16& g1, float (&term)[16], const bool diag) __attribute__((always_inline)) {
    wave_lds_fence();
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int t = 8 * q4 + 4 * lh;
      *reinterpret_cast<float4*>(gt + li * GLD + t) = make_float4(g0[4 * q4], g0[4 * q4 + 1], g0[4 * q4 + 2], g0[4 * q4 + 3]);
      *reinterpret_cast<float4*>(gt + li * GLD + 32 + t) = make_float4(g1[4 * q4], g1[4 * q4 + 1], g1[4 * q4 + 2], g1[4 * q4 + 3]);
    }
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = gt[li * GLD + (c - li + 31)];
      const bool keep = diag && c <= li, zero = diag && c == li + 1;
      term[r] = keep ? term[r] : (zero ? 0.f : v);
    }
  };
  auto new_tile_row = [&](const int j0, const bool upper) { return upper ? j0 - i0 - 1 : N + j0 - i0; };
  PRE_T pre[NPRE];
  f32x16 ghi;
#pragma unroll
  for (int r = 0; r < 16; ++r) ghi[r] = 0.f;

  const int nchunks = (N + 31) / 32;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int j0 = kc * 32;
    __syncthreads();
    for (int i = tid; i < 32 * (DK / 4); i += 256) {
      const int r = i / (DK / 4), c4 = i % (DK / 4);
      const int key = j0 + r;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (key < N) {
        kv = *reinterpret_cast<const float4*>(kp + (key * ld + c4 * 4));
        vv = *reinterpret_cast<const float4*>(vp + (key * ld + c4 * 4));
      }
      // K row: [hi d 0 .. DK-1 | lo d 0 .. DK-1] as bf16 (attention_x3_kernel's staging)
      const unsigned k0 = pk_bf16(kv.x, kv.y), k1 = pk_bf16(kv.z, kv.w);
      const float kx = __builtin_bit_cast(float, k0 << 16), ky = __builtin_bit_cast(float, k0 & 0xffff0000u);
      const float kz = __builtin_bit_cast(float, k1 << 16), kw = __builtin_bit_cast(float, k1 & 0xffff0000u);
      *reinterpret_cast<uint2*>(Kh + r * KS + c4 * 2) = make_uint2(k0, k1);
      *reinterpret_cast<uint2*>(Kh + r * KS + DK / 2 + c4 * 2) = make_uint2(pk_bf16(kv.x - kx, kv.y - ky), pk_bf16(kv.z - kz, kv.w - kw));
      // V transposed, key r -> slot (attention_x3_kernel); head-dim row d = 4 c4 + e at physical row (d & 3) (8 DTP) + (d >> 2): the
      // padded rows DK .. 32 DTP - 1 get rows of their own, which nobody writes and whose products land in output rows nobody stores
      const int rr = (r & 3) + 4 * (r >> 3), h = (r >> 2) & 1;
      const int slot = 16 * (rr >> 3) + 8 * h + (rr & 7);
      unsigned short* vt = reinterpret_cast<unsigned short*>(Vt) + c4 * (VS * 2) + slot;
      const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 hb = (__bf16)vs[e];
        const __bf16 lb = (__bf16)(vs[e] - (float)hb);
        vt[e * (8 * DTP) * (VS * 2)] = __builtin_bit_cast(unsigned short, hb);
        vt[e * (8 * DTP) * (VS * 2) + 32] = __builtin_bit_cast(unsigned short, lb);
      }
    }
    float pterm[16];
    if (j0 <= i0) {         // lower form (see rel_attention_kernel)
      if (kc == 0) {
        band_fetch(pre, new_tile_row(j0, false) - 32);
        ghi = band_tile(pre);
        band_fetch(pre, new_tile_row(j0, false));
      }
      const f32x16 g1 = band_tile(pre);
      if (j0 < i0) band_fetch(pre, new_tile_row(j0 + 32, false));
      skew(ghi, g1, pterm, false);
      ghi = g1;
    }
    if (j0 >= i0) {         // upper form
      if (j0 == i0) {
        load_frag(min(qi + 1, N - 1), vb, xh, xl);
        band_fetch(pre, new_tile_row(j0, true) - 32);
        ghi = band_tile(pre);
        band_fetch(pre, new_tile_row(j0, true));
      }
      const f32x16 g1 = band_tile(pre);
      if (kc + 1 < nchunks) band_fetch(pre, new_tile_row(j0 + 32, true));
      skew(ghi, g1, pterm, j0 == i0);
      ghi = g1;
      if (j0 == i0 + 32 && li == 31 && lh == 0) pterm[0] = 0.f;
    }
    __syncthreads();

    f32x16 s;   // the content scores accumulate on top of the position term
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = pterm[r];
    const unsigned* ka = Kh + li * KS + lh * 4;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ka + ks * 8));
      const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ka + DK / 2 + ks * 8));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, fh[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fl[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fh[ks], s, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] * inv_temp;
      v = key < N ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __expf(m_run - m_new);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __expf(s[r] - m_new);
      psum += pv[r];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
    bf16x8_t pfh[2], pfl[2];
    split8(pv, pfh[0], pfl[0]);
    split8(pv + 8, pfh[1], pfl[1]);
#pragma unroll
    for (int d = 0; d < DTP; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= corr;
      const unsigned* va = Vt + ((li & 3) * (8 * DTP) + d * 8 + (li >> 2)) * VS + lh * 4;  // physical row of head-dim row d * 32 + li
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(va + s2 * 8));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(va + 16 + s2 * 8));
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, pfh[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pfl[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pfh[s2], o[d], 0, 0, 0);
      }
    }
  }

  const int q = i0 + li;
  if (q < N) {
    const float inv = 1.0f / l_run;
    float* orow = out + ((long long)b * N + q) * H + head * DK + 4 * lh;
#pragma unroll
    for (int d = 0; d < DTP; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (d * 32 + 8 * g + 4 * lh < DK) {
          float4 v;
          v.x = o[d][4 * g + 0] * inv; v.y = o[d][4 * g + 1] * inv; v.z = o[d][4 * g + 2] * inv; v.w = o[d][4 * g + 3] * inv;
          *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = v;
        }
      }
  }
}

// ---- split-precision attention for small grids: attention_split_kernel's arrangement (two wavefronts per 32-query tile, half the head
// dimension each, partial S^T tiles added in LDS in attention_x3_kernel's order, next chunk's rows requested before this chunk's
// arithmetic) on attention_x3_kernel's operands.  Same bits as attention_x3_kernel.
template <int DK, int QT>
__global__ __launch_bounds__(QT * 128, 2) void attention_x3_split_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                    const int32_t* __restrict__ lens, int N, int H, float temperature, const RowMap rm) {
  constexpr int KS = DK + 4;
  constexpr int VS = 36;
  constexpr int DH = DK / 2, DTH = DH / 32, NSH = DH / 16;
  static_assert(DK % 64 == 0, "the split form needs an even number of 32-wide head-dim tiles");
  constexpr int NF4 = 32 * (DK / 4);
  constexpr int NTH = QT * 128;            // threads: QT query tiles x 2 halves x 64 lanes
  constexpr int NLD = (NF4 + NTH - 1) / NTH;
  __shared__ __attribute__((aligned(16))) unsigned Kh[32 * KS];
  __shared__ __attribute__((aligned(16))) unsigned Vt[DK * VS];
  __shared__ __attribute__((aligned(16))) float Sx[QT * 2][16 * 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int qt = wave >> 1, part = wave & 1;
  ATT_BLOCK_OF_GRID(QT * 32)
  const int q0 = qblk * (QT * 32) + qt * 32;
  const int len = min(lens ? lens[b] : N, N);
  if (qblk * (QT * 32) >= len) {
    for (int i = tid; i < QT * 32 * (DK / 4); i += NTH) {
      const int q = qblk * (QT * 32) + i / (DK / 4);
      if (q < N) *reinterpret_cast<float4*>(out + ((long long)b * N + q) * H + head * DK + (i % (DK / 4)) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int ld = 3 * H;
  const float* base = qkv + (long long)b * N * ld;
  const float* qp = base + head * DK + part * DH;
  const float* kp = base + H + head * DK;
  const float* vp = base + 2 * H + head * DK;

  bf16x8_t qh[NSH], ql[NSH];
  {
    const int qrow = min(q0 + li, N - 1);
    const float* qr = qp + (long long)qrow * ld + lh * 8;
#pragma unroll
    for (int s = 0; s < NSH; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(qr + s * 16), c = *reinterpret_cast<const float4*>(qr + s * 16 + 4);
      const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      split8(v, qh[s], ql[s]);
    }
  }
  f32x16 o[DTH];
#pragma unroll
  for (int d = 0; d < DTH; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float inv_temp = 1.0f / temperature;

  float4 kreg[NLD], vreg[NLD];
  auto fetch = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = min(tid + i * NTH, NF4 - 1);
      const int r = idx / (DK / 4), c4 = idx % (DK / 4);
      const int key = min(kc * 32 + r, N - 1);
      kreg[i] = *reinterpret_cast<const float4*>(kp + (long long)key * ld + c4 * 4);
      vreg[i] = *reinterpret_cast<const float4*>(vp + (long long)key * ld + c4 * 4);
    }
  };
  auto stash = [&](int kc) __attribute__((always_inline)) {  // attention_x3_kernel's staging, from the registers
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * NTH;
      if (idx < NF4) {
        const int r = idx / (DK / 4), c4 = idx % (DK / 4);
        const bool ok = kc * 32 + r < N;
        const float4 kv = make_float4(ok ? kreg[i].x : 0.f, ok ? kreg[i].y : 0.f, ok ? kreg[i].z : 0.f, ok ? kreg[i].w : 0.f);
        const float4 vv = make_float4(ok ? vreg[i].x : 0.f, ok ? vreg[i].y : 0.f, ok ? vreg[i].z : 0.f, ok ? vreg[i].w : 0.f);
        const unsigned k0 = pk_bf16(kv.x, kv.y), k1 = pk_bf16(kv.z, kv.w);
        const float kx = __builtin_bit_cast(float, k0 << 16), ky = __builtin_bit_cast(float, k0 & 0xffff0000u);
        const float kz = __builtin_bit_cast(float, k1 << 16), kw = __builtin_bit_cast(float, k1 & 0xffff0000u);
        *reinterpret_cast<uint2*>(Kh + r * KS + c4 * 2) = make_uint2(k0, k1);
        *reinterpret_cast<uint2*>(Kh + r * KS + DK / 2 + c4 * 2) = make_uint2(pk_bf16(kv.x - kx, kv.y - ky), pk_bf16(kv.z - kz, kv.w - kw));
        const int rr = (r & 3) + 4 * (r >> 3), h = (r >> 2) & 1;
        const int slot = 16 * (rr >> 3) + 8 * h + (rr & 7);
        unsigned short* vt = reinterpret_cast<unsigned short*>(Vt) + c4 * (VS * 2) + slot;
        const float vs[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 hb = (__bf16)vs[e];
          const __bf16 lb = (__bf16)(vs[e] - (float)hb);
          vt[e * (DK / 4) * (VS * 2)] = __builtin_bit_cast(unsigned short, hb);
          vt[e * (DK / 4) * (VS * 2) + 32] = __builtin_bit_cast(unsigned short, lb);
        }
      }
    }
  };

  const int nchunks = (len + 31) / 32;
  fetch(0);
  for (int kc = 0; kc < nchunks; ++kc) {
    lds_barrier();
    stash(kc);
    lds_barrier();
    if (kc + 1 < nchunks) fetch(kc + 1);

    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const unsigned* ka = Kh + li * KS + part * (DH / 2) + lh * 4;   // this half's hi words; its lo words are DK / 2 further
#pragma unroll
    for (int ks = 0; ks < NSH; ++ks) {
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ka + ks * 8));
      const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ka + DK / 2 + ks * 8));
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[ks], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[ks], s, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Sx[wave][r * 64 + lane] = s[r];
    lds_barrier();
    {
      const float* s0 = Sx[wave & ~1];
      const float* s1 = Sx[wave | 1];
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = s0[r * 64 + lane] + s1[r * 64 + lane];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kc * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = s[r] * inv_temp;
      v = key < len ? v : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float corr = __expf(m_run - m_new);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __expf(s[r] - m_new);
      psum += pv[r];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * corr + psum;
    m_run = m_new;
    bf16x8_t ph[2], pl[2];
    split8(pv, ph[0], pl[0]);
    split8(pv + 8, ph[1], pl[1]);
#pragma unroll
    for (int d = 0; d < DTH; ++d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= corr;
      const unsigned* va = Vt + ((li & 3) * (DK / 4) + (part * DTH + d) * 8 + (li >> 2)) * VS + lh * 4;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(va + s2 * 8));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(va + 16 + s2 * 8));
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ph[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, pl[s2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ph[s2], o[d], 0, 0, 0);
      }
    }
  }

  const int q = q0 + li;
  if (q < N) {
    const bool valid = q < len;
    const float inv = valid ? 1.0f / l_run : 0.f;
    float* orow = out + ((long long)b * N + q) * H + head * DK + part * DH + 4 * lh;
#pragma unroll
    for (int d = 0; d < DTH; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = valid ? o[d][4 * g + 0] * inv : 0.f;
        v.y = valid ? o[d][4 * g + 1] * inv : 0.f;
        v.z = valid ? o[d][4 * g + 2] * inv : 0.f;
        v.w = valid ? o[d][4 * g + 3] * inv : 0.f;
        *reinterpret_cast<float4*>(orow + d * 32 + 8 * g) = v;
      }
  }
}

}  // namespace

const char* launch_rel_attention(const float* qkv, const float* pos, int pos_rows, const float* u, const float* v, float* out, int B, int N,
                                 int H, int n_head, hipStream_t s, const float* pos_x3) {
  if (!qkv || !pos || !u || !v || !out) return "rel_attention: null pointer";
  if (B <= 0 || N <= 0 || n_head <= 0 || H % n_head || pos_rows < N) return "rel_attention: bad dims (the position table must cover N rows)";
  if (B > 65535 || n_head > 65535) return "rel_attention: batch / heads exceed the grid limit";
  if ((long long)N * 3 * H >= (1LL << 31) || (long long)pos_rows * (H / n_head) >= (1LL << 31)) return "rel_attention: one utterance must stay below 2^31 elements (32-bit offsets)";
  if (((uintptr_t)qkv | (uintptr_t)pos | (uintptr_t)u | (uintptr_t)v | (uintptr_t)out) & 15) return "rel_attention: buffers must be 16-byte aligned";
  const int dk = H / n_head;
  const float temperature = sqrtf((float)H);  // sqrt(d_model), not sqrt(d_head) (conformer.py:384)
  dim3 grid((N + 127) / 128, n_head, B);
  if (pos_x3) {  // split precision: the table pre-split by the packer
    if ((uintptr_t)pos_x3 & 15) return "rel_attention: buffers must be 16-byte aligned";
    const unsigned* px = reinterpret_cast<const unsigned*>(pos_x3);
    switch (dk) {
      case 16: hipLaunchKernelGGL(rel_attention_x3_kernel<16>, grid, dim3(256), 0, s, qkv, px, pos_rows, u, v, out, N, H, temperature); break;
      case 32: hipLaunchKernelGGL(rel_attention_x3_kernel<32>, grid, dim3(256), 0, s, qkv, px, pos_rows, u, v, out, N, H, temperature); break;
      case 48: hipLaunchKernelGGL(rel_attention_x3_kernel<48>, grid, dim3(256), 0, s, qkv, px, pos_rows, u, v, out, N, H, temperature); break;
      case 64: hipLaunchKernelGGL(rel_attention_x3_kernel<64>, grid, dim3(256), 0, s, qkv, px, pos_rows, u, v, out, N, H, temperature); break;
      case 96: hipLaunchKernelGGL(rel_attention_x3_kernel<96>, grid, dim3(256), 0, s, qkv, px, pos_rows, u, v, out, N, H, temperature); break;
      default: return "rel_attention: the split-precision form needs a head dim of 16, 32, 48, 64 or 96";
    }
    return hipGetLastError() == hipSuccess ? nullptr : "rel_attention: launch failed";
  }
  switch (dk) {
    case 8: hipLaunchKernelGGL(rel_attention_kernel<8>, grid, dim3(256), 0, s, qkv, pos, pos_rows, u, v, out, N, H, temperature); break;
    case 16: hipLaunchKernelGGL(rel_attention_kernel<16>, grid, dim3(256), 0, s, qkv, pos, pos_rows, u, v, out, N, H, temperature); break;
    case 32: hipLaunchKernelGGL(rel_attention_kernel<32>, grid, dim3(256), 0, s, qkv, pos, pos_rows, u, v, out, N, H, temperature); break;
    case 48: hipLaunchKernelGGL(rel_attention_kernel<48>, grid, dim3(256), 0, s, qkv, pos, pos_rows, u, v, out, N, H, temperature); break;
    case 64: hipLaunchKernelGGL(rel_attention_kernel<64>, grid, dim3(256), 0, s, qkv, pos, pos_rows, u, v, out, N, H, temperature); break;
    case 96: hipLaunchKernelGGL(rel_attention_kernel<96>, grid, dim3(256), 0, s, qkv, pos, pos_rows, u, v, out, N, H, temperature); break;
    default: return "rel_attention: head dim must be one of 8, 16, 32, 48, 64, 96";
  }
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? nullptr : hipGetErrorString(err);
}

// largest padded grid (64-query workgroups, before the segment split) that computes its key segments in parallel; E2ETTS_ATT_PAR_MAX (tuning aid)
long long attention_par_max_grid() {
  // default: up to one round of the serial form's workgroups (2 per CU).  Same box, T = 768: B = 8 (192 workgroups) 1.12 -> 1.04 ms/step,
  // B = 32 (768) 2.41 -> 2.47: above one round the serial form already fills the chip and the workspace round trip costs more than it buys
  static const long long v = getenv("E2ETTS_ATT_PAR_MAX") ? atoll(getenv("E2ETTS_ATT_PAR_MAX")) : 512;
  return v;
}

size_t attention_workspace_bytes(int B, int N, int H, int n_head) {
  const long long nseg = ((N + 31) / 32 + ATT_SEG_CHUNKS - 1) / ATT_SEG_CHUNKS;
  return (size_t)((long long)B * n_head * nseg * N * (H / n_head + 2) * 4);
}

const char* launch_attention(const float* qkv, float* out, const int32_t* lens, int B, int N, int H, int n_head, int x3,
                             hipStream_t s, const int32_t* lens_host, float* ws, size_t ws_bytes) {
  if (!qkv || !out) return "attention: null pointer";
  if (B <= 0 || N <= 0 || n_head <= 0 || H % n_head) return "attention: bad dims";
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) return "attention: buffers must be 16-byte aligned";
  const int dk = H / n_head;
  // reference: temperature = np.power(d_k, 0.5), scores divided by it in fp32 (U/blocks/transformer.py:201,254)
  const float temperature = (float)sqrt((double)dk);
  dim3 grid((N + 127) / 128, n_head, B);
  // The split form on 64-query workgroups (same bits, see attention_split_kernel).  fp32: at every grid size -- B = 1: 225 -> 135 us per
  // decoder layer, B = 32: 2.50 -> 2.22 ms/step.  bf16x3: up to 512 workgroups (B = 16: 0.82 -> 0.75 ms/step); beyond, the 256-query
  // workgroups of attention_x3_kernel win (B = 32: 0.95 against 1.33 ms), their K / V conversion shared by four times the queries.
  // E2ETTS_ATT_SPLIT_MAX overrides both limits (0: never split -- what the bit-identity test runs its child process with).
  static const long long split_env = getenv("E2ETTS_ATT_SPLIT_MAX") ? atoll(getenv("E2ETTS_ATT_SPLIT_MAX")) : -1;
  const long long split_max = split_env >= 0 ? split_env : (x3 ? 512 : (1LL << 62));
  dim3 gs((N + 63) / 64, n_head, B);
  // ragged batch, lengths known on the host: the compact 1-D grid of the query blocks that exist (kernels.h: RowMap)
  RowMap rm;
  auto shape = [&](dim3 padded, int qrows) -> dim3 {
    rm.n = 0;
    if (!lens || !lens_host || B > ROWMAP_MAX) return padded;
    rm.n = B;
    rm.identity();   // slots ordered by length, longest first (stable): a unit's cost grows with its utterance's length
    std::stable_sort(rm.idx, rm.idx + B, [&](unsigned char x, unsigned char y) { return lens_host[x] > lens_host[y]; });
    rm.cum[0] = 0;
    for (int k = 0; k < B; ++k) rm.cum[k + 1] = rm.cum[k] + (std::min(std::max(lens_host[rm.idx[k]], 0), N) + qrows - 1) / qrows * n_head;
    return dim3((unsigned)std::max(rm.cum[B], 1));
  };
  const bool split = (dk == 64 || dk == 128 || dk == 192) && (long long)gs.x * gs.y * gs.z <= split_max;
  if (x3 && split) {
    const dim3 gc = shape(gs, 64);
    switch (dk) {
      case 64: hipLaunchKernelGGL((attention_x3_split_kernel<64, 2>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      case 128: hipLaunchKernelGGL((attention_x3_split_kernel<128, 2>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      default: hipLaunchKernelGGL((attention_x3_split_kernel<192, 2>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
    }
    return hipGetLastError() == hipSuccess ? nullptr : "attention: launch failed";
  }
  if (x3) {
    static const int nw = getenv("E2ETTS_ATT_NW") ? atoi(getenv("E2ETTS_ATT_NW")) : 8;  // tuning aid
    if (nw == 8) {
      dim3 g8((N + 255) / 256, n_head, B);
      const dim3 gc = shape(g8, 256);
      switch (dk) {
        case 32: hipLaunchKernelGGL((attention_x3_kernel<32, 8>), gc, dim3(512), 0, s, qkv, out, lens, N, H, temperature, rm); break;
        case 64: hipLaunchKernelGGL((attention_x3_kernel<64, 8>), gc, dim3(512), 0, s, qkv, out, lens, N, H, temperature, rm); break;
        case 96: hipLaunchKernelGGL((attention_x3_kernel<96, 8>), gc, dim3(512), 0, s, qkv, out, lens, N, H, temperature, rm); break;
        case 128: hipLaunchKernelGGL((attention_x3_kernel<128, 8>), gc, dim3(512), 0, s, qkv, out, lens, N, H, temperature, rm); break;
        case 192: hipLaunchKernelGGL((attention_x3_kernel<192, 8>), gc, dim3(512), 0, s, qkv, out, lens, N, H, temperature, rm); break;
        default: return "attention: head dim must be one of 32, 64, 96, 128, 192";
      }
      return hipGetLastError() == hipSuccess ? nullptr : "attention: launch failed";
    }
    const dim3 gc = shape(grid, 128);
    switch (dk) {
      case 32: hipLaunchKernelGGL((attention_x3_kernel<32, 4>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      case 64: hipLaunchKernelGGL((attention_x3_kernel<64, 4>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      case 96: hipLaunchKernelGGL((attention_x3_kernel<96, 4>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      case 128: hipLaunchKernelGGL((attention_x3_kernel<128, 4>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      case 192: hipLaunchKernelGGL((attention_x3_kernel<192, 4>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
      default: return "attention: head dim must be one of 32, 64, 96, 128, 192";
    }
    return hipGetLastError() == hipSuccess ? nullptr : "attention: launch failed";
  }
  if (split) {
    dim3 gc = shape(gs, 64);
    // Small padded grids (the B = 1 latency path): every key segment of a query block in a workgroup of its own, merged by
    // attention_combine_kernel -- the same operations as the in-register merge, so the same bits (ATT_SEG_CHUNKS).  B = 1, T = 768: 24
    // workgroups walking 24 chunks -> 72 walking 8, 133 -> ~55 us per decoder layer.  E2ETTS_ATT_PAR=0: never (tuning aid).
    static const bool par_on = !(getenv("E2ETTS_ATT_PAR") && atoi(getenv("E2ETTS_ATT_PAR")) == 0);
    const int nseg = ((N + 31) / 32 + ATT_SEG_CHUNKS - 1) / ATT_SEG_CHUNKS;
    int par_nseg = 0;
    float *ws_o = nullptr, *ws_ml = nullptr;
    if (par_on && rm.n == 0 && nseg >= 2 && ws && ws_bytes >= attention_workspace_bytes(B, N, H, n_head) && (((uintptr_t)ws) & 15) == 0 &&
        (long long)gs.x * gs.y * gs.z <= attention_par_max_grid() && (long long)gs.x * nseg < (1LL << 31)) {
      par_nseg = nseg;
      ws_o = ws;
      ws_ml = ws + (long long)B * n_head * nseg * N * dk;
      gc = dim3(gs.x * nseg, gs.y, gs.z);
    }
    switch (dk) {
      case 64: hipLaunchKernelGGL((attention_split_kernel<64, 2>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm, ws_o, ws_ml, par_nseg); break;
      case 128: hipLaunchKernelGGL((attention_split_kernel<128, 2>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm, ws_o, ws_ml, par_nseg); break;
      default: hipLaunchKernelGGL((attention_split_kernel<192, 2>), gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm, ws_o, ws_ml, par_nseg); break;
    }
    if (hipGetLastError() != hipSuccess) return "attention: launch failed";
    if (par_nseg > 0) {
      const long long total4 = (long long)B * N * (H / 4);
      hipLaunchKernelGGL(attention_combine_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ws_o, ws_ml, out, lens, N, H, dk, par_nseg, total4);
    }
    return hipGetLastError() == hipSuccess ? nullptr : "attention: launch failed";
  }
  const dim3 gc = shape(grid, 128);
  switch (dk) {
    case 32: hipLaunchKernelGGL(attention_kernel<32>, gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
    case 64: hipLaunchKernelGGL(attention_kernel<64>, gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
    case 96: hipLaunchKernelGGL(attention_kernel<96>, gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
    case 128: hipLaunchKernelGGL(attention_kernel<128>, gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
    case 192: hipLaunchKernelGGL(attention_kernel<192>, gc, dim3(256), 0, s, qkv, out, lens, N, H, temperature, rm); break;
    default: return "attention: head dim must be one of 32, 64, 96, 128, 192";
  }
  return hipGetLastError() == hipSuccess ? nullptr : "attention: launch failed";
}

}  // namespace e2etts
