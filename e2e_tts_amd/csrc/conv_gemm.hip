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
#include "kernels.h"

namespace e2etts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 32;   // channels per K chunk
constexpr int LDK = 36;  // padded LDS row stride (floats)
constexpr int MAX_HALO = 64;

__device__ __forceinline__ float4 lrelu4(float4 v, float slope) {
  v.x = v.x >= 0.f ? v.x : v.x * slope;
  v.y = v.y >= 0.f ? v.y : v.y * slope;
  v.z = v.z >= 0.f ? v.z : v.z * slope;
  v.w = v.w >= 0.f ? v.w : v.w * slope;
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
  return v;
}

// X3 = false: operands fp32, v_mfma_f32_32x32x2_f32.
// X3 = true : split precision.  Each fp32 operand is hi + lo (two bf16); the product keeps hi*hi + hi*lo + lo*hi in
//             three v_mfma_f32_32x32x16_bf16 with fp32 accumulation (relative error ~2^-16 per product instead of
//             bf16's 2^-8; measured on the vocoder: wav mean-L1 9e-7 vs fp64, against 6e-8 for fp32 and 5e-4 for
//             plain bf16).  An LDS row holds one 32-channel chunk as [32 bf16 hi | 32 bf16 lo | pad] = the same 144
//             bytes as the fp32 row; activations are split while staging, weights arrive pre-split from the packer.
template <int BM, int BN, int WM, int WN, bool X3>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvParams p, const int tiles_per_block) {
  constexpr int NWN = BN / WN;
  constexpr int MT = WM / 32, NT = WN / 32;
  static_assert((BM / WM) * NWN == 4, "4 wavefronts per workgroup");
  constexpr int BROWS = BN / 32;                      // weight-tile rows staged per thread
  constexpr int AROWS = (BM + MAX_HALO + 31) / 32;    // slab rows staged per thread (upper bound)
  constexpr int ELD = WN + 4;                         // epilogue patch row stride (floats)
  static_assert(4 * 16 * ELD <= BM * LDK, "epilogue patches must fit in the slab");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int halo = p.dil * (p.KW - 1);
  const int arows = BM + halo;
  float* As = smem;
  float* Bs = smem + arows * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int li = lane & 31, lh = lane >> 5;
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;

  const int b = blockIdx.z;
  const int n0 = blockIdx.y * BN;
  const int mtiles = (p.T + BM - 1) / BM;
  const int tile0 = blockIdx.x * tiles_per_block;
  const int ntile = min(tiles_per_block, mtiles - tile0);
  const float* in_b = p.in + (long long)b * p.in_bs;
  const int nchunk = (p.Cin + BK - 1) / BK;
  const int KC = X3 ? p.KW * nchunk * BK : p.KW * p.Cin;  // X3 weights: [Cout][KW][nchunk][32 words], chunk-padded
  const int nitem = ntile * nchunk;
  const int niter = nitem * p.KW;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  float4 breg[BROWS];
  float4 areg[AROWS];

  // Staging loads are UNCONDITIONAL (addresses clamped into the tensor) and the zero fill is applied when the
  // registers are written to LDS: a load under a runtime predicate makes hipcc branch around it and serialise.
  const float* wrow[BROWS];
  bool wok[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int n = n0 + lrow + i * 32;
    wok[i] = n < p.Cout;
    wrow[i] = p.w + (long long)min(n, p.Cout - 1) * KC;
  }
  const int cmax = p.Cin - 4;  // Cin % 4 == 0: last float4 of a row
  bool b_cok = true, a_cok = true;
  int a_tbase = 0;

  auto load_b = [&](int chunk, int j) {
    const int c = chunk * BK + lc4;
    b_cok = X3 ? true : c < p.Cin;  // the packed X3 rows are zero-padded to whole chunks
    const int off = X3 ? (j * nchunk + chunk) * BK + lc4 : j * p.Cin + min(c, cmax);
#pragma unroll
    for (int i = 0; i < BROWS; ++i) breg[i] = *reinterpret_cast<const float4*>(wrow[i] + off);
  };
  auto store_b = [&](int buf) {
    float* dst = Bs + buf * (BN * LDK);
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      float4 v = breg[i];
      if (!(wok[i] && b_cok)) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(dst + (lrow + i * 32) * LDK + lc4) = v;
    }
  };
  auto load_a = [&](int tile, int chunk) {
    const int c = chunk * BK + lc4;
    a_cok = c < p.Cin;
    a_tbase = tile * BM - p.pad;
    const float* src = in_b + min(c, cmax);
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int t = min(max(a_tbase + lrow + i * 32, 0), p.T - 1);
      areg[i] = *reinterpret_cast<const float4*>(src + (long long)t * p.in_ld);
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = lrow + i * 32;
      const int t = a_tbase + r;
      float4 v = areg[i];
      if (!(a_cok && t >= 0 && t < p.T)) v = make_float4(0.f, 0.f, 0.f, 0.f);
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
  const bool vec_ok = (p.Cout % 4 == 0) && (p.out_ld % 4 == 0) && ((p.out_bs & 3) == 0) && (((uintptr_t)p.out & 15) == 0) &&
                      (!p.res || ((p.res_ld % 4 == 0) && ((p.res_bs & 3) == 0) && (((uintptr_t)p.res & 15) == 0))) &&
                      (!p.bias || (((uintptr_t)p.bias & 15) == 0));
  const int len = p.lens ? p.lens[b] : p.T;
  float* out_b = p.out + (long long)b * p.out_bs;
  const float* res_b = p.res ? p.res + (long long)b * p.res_bs : nullptr;
  auto epilogue = [&](int tile) {
    const int t0 = tile * BM;
    if (vec_ok) {
      float* patch = As + wave * (16 * ELD);  // wave-private: no workgroup barrier between its write and read
      constexpr int LPR = WN / 4;             // lanes per row when reading float4
      constexpr int RPP = 64 / LPR;           // rows per pass
      const int prow = lane / LPR, pc4 = (lane % LPR) * 4;
      const int col = n0 + wn * WN + pc4;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && col < p.Cout) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          __builtin_amdgcn_sched_barrier(0);  // keep the passes apart: hoisting their loads together costs ~100 VGPRs
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              const int r = hh * 8 + rr;
              const int row = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lh;  // 0..15 inside this half
              patch[row * ELD + n * 32 + li] = acc[m][n][r];
            }
#pragma unroll
          for (int ps = 0; ps < 16 / RPP; ++ps) {
            const int row = ps * RPP + prow;
            const int t = t0 + wm * WM + m * 32 + hh * 16 + row;
            float4 v = *reinterpret_cast<const float4*>(patch + row * ELD + pc4);
            if (t < p.T && col < p.Cout) {
              v.x = act1(v.x + bias4.x, p.act, p.act_slope);
              v.y = act1(v.y + bias4.y, p.act, p.act_slope);
              v.z = act1(v.z + bias4.z, p.act, p.act_slope);
              v.w = act1(v.w + bias4.w, p.act, p.act_slope);
              if (res_b) {
                const float4 rv = *reinterpret_cast<const float4*>(res_b + (long long)t * p.res_ld + col);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
              }
              if (t >= len) v = make_float4(0.f, 0.f, 0.f, 0.f);
              float4* o = reinterpret_cast<float4*>(out_b + (long long)t * p.out_ld + col);
              if (p.accumulate) {
                const float4 ov = *o;
                v.x += ov.x; v.y += ov.y; v.z += ov.z; v.w += ov.w;
              }
              if (p.out_div != 1.0f) {
                v.x = v.x / p.out_div; v.y = v.y / p.out_div; v.z = v.z / p.out_div; v.w = v.w / p.out_div;
              }
              *o = v;
            }
          }
        }
      }
    } else {
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

  load_a(tile0, 0);
  store_a();
  load_b(0, 0);
  store_b(0);
  int cur = 0;
  int tl = 0, chunk = 0, j = 0;  // work item = (tile0 + tl, chunk); tap j
  for (int it = 0; it < niter; ++it) {
    const bool last_tap = j == p.KW - 1;
    const bool tile_done = last_tap && chunk == nchunk - 1;
    const bool more_items = !(tile_done && tl == ntile - 1);
    const int nchk = chunk + 1 == nchunk ? 0 : chunk + 1;
    if (last_tap && more_items) load_a(tile0 + tl + (nchk == 0 ? 1 : 0), nchk);  // next slab: next chunk, or next tile
    if (it + 1 < niter) {  // next weight tile: next tap, or tap 0 of the next item's chunk
      if (!last_tap) load_b(chunk, j + 1);
      else load_b(nchk, 0);
    }
    __syncthreads();  // slab + Bs[cur] visible

    const float* a_base = As + (wm * WM + li + j * p.dil) * LDK + lh * 4;
    const float* b_base = Bs + cur * (BN * LDK) + (wn * WN + li) * LDK + lh * 4;
    if constexpr (X3) {
      // lane (row li, half lh) holds k = 16 s + 8 lh .. + 7 of its row: one 16-byte read per operand half
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          ah[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + ks * 8));
          al[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + 16 + ks * 8));
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          bh[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(b_base + n * 32 * LDK + ks * 8));
          bl[n] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4*>(b_base + n * 32 * LDK + 16 + ks * 8));
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
          }
      }
    } else {
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        float4 af[MT], bf[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const float4*>(a_base + m * 32 * LDK + q * 8);
#pragma unroll
        for (int n = 0; n < NT; ++n) bf[n] = *reinterpret_cast<const float4*>(b_base + n * 32 * LDK + q * 8);
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
    }
    if (it + 1 < niter) store_b(cur ^ 1);  // that buffer was last read before this iteration's barrier
    cur ^= 1;
    if (last_tap) {
      if (tile_done || more_items) __syncthreads();  // every wave is done reading the slab
      if (tile_done) {
        epilogue(tile0 + tl);
        if (more_items) __syncthreads();  // patches read back before the slab is overwritten
        ++tl;
      }
      if (more_items) store_a();
      j = 0;
      chunk = nchk;
    } else {
      ++j;
    }
  }
}

template <int BM, int BN, int WM, int WN, bool X3>
const char* launch_cfg(const ConvParams& p, hipStream_t s) {
  const int halo = p.dil * (p.KW - 1);
  const size_t lds = (size_t)((BM + halo) * LDK + 2 * BN * LDK) * sizeof(float);
  if (lds > 64 * 1024) return "conv_gemm: LDS tile exceeds 64 KiB";
  const int mtiles = (p.T + BM - 1) / BM;
  const int ntiles = (p.Cout + BN - 1) / BN;
  // Persistent over M: enough workgroups for ~8 per CU, each walking up to 8 consecutive tiles.
  const long long total = (long long)mtiles * ntiles * p.B;
  int tpb = (int)(total / (256 * 8));
  tpb = tpb < 1 ? 1 : (tpb > 8 ? 8 : tpb);
  if (tpb > mtiles) tpb = mtiles;
  dim3 grid((mtiles + tpb - 1) / tpb, ntiles, p.B);
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, X3>), grid, dim3(256), lds, s, p, tpb);
  return hipGetLastError() == hipSuccess ? nullptr : "conv_gemm: launch failed";
}

}  // namespace

double conv_gemm_flops(const ConvParams& p) { return 2.0 * p.B * (double)p.T * p.Cout * p.KW * p.Cin; }

double conv_gemm_bytes(const ConvParams& p) {
  double e = (double)p.B * p.T * (p.Cin + p.Cout * (1.0 + (p.res ? 1 : 0) + (p.accumulate ? 1 : 0)));
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
  if (p.x3) {
    if (p.Cout > 64) return launch_cfg<128, 128, 64, 64, true>(p, s);
    if (p.Cout > 32) return launch_cfg<256, 64, 64, 64, true>(p, s);
    return launch_cfg<256, 32, 64, 32, true>(p, s);
  }
  if (p.Cout > 64) return launch_cfg<128, 128, 64, 64, false>(p, s);
  if (p.Cout > 32) return launch_cfg<256, 64, 64, 64, false>(p, s);
  return launch_cfg<256, 32, 64, 32, false>(p, s);
}

}  // namespace e2etts
