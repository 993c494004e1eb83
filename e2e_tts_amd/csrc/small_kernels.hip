// The memory-bound / integer kernels around the two MFMA kernels: embedding, LayerNorm, the variance
// adaptor's integer decisions (duration rounding, f0 / energy buckets), the length regulator, the
// vocoder's 1-channel output convolution.  fp32 throughout; float4 (16 B / lane) global accesses.
#include <math.h>

#include <algorithm>

#include "kernels.h"

namespace e2etts {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------- LayerNorm (one wavefront per row)
template <int NV>  // float4 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const int32_t* __restrict__ lens, int rows, int N, int C, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nv = C / 4;
  float4* yr = reinterpret_cast<float4*>(y + (long long)row * C);
  if (lens) {
    const int b = row / N, t = row - b * N;
    if (t >= lens[b]) {  // masked_fill(mask, 0) after the norm (reference U/blocks/transformer.py:182-187)
      for (int i = lane; i < nv; i += 64) yr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
  }
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * C);
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + i * 64;
    v[i] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float a = v[i].x - mean, b2 = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b2 * b2) + (c * c + d * d);
    }
  }
  const float var = wave_sum(q) / (float)C;  // biased, as torch.nn.LayerNorm
  const float rstd = 1.0f / sqrtf(var + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + i * 64;
    if (idx < nv) {
      const float4 g = g4[idx], bb = b4[idx];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + bb.x;
      o.y = (v[i].y - mean) * rstd * g.y + bb.y;
      o.z = (v[i].z - mean) * rstd * g.z + bb.z;
      o.w = (v[i].w - mean) * rstd * g.w + bb.w;
      yr[idx] = o;
    }
  }
}

// ---------------------------------------------------------------- embedding + position table
__global__ void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb, const float* __restrict__ pos,
                             float* __restrict__ x, int L, int H, int n_rows) {
  const int row = blockIdx.x;  // b * L + l
  const int l = row % L;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= n_rows ? n_rows - 1 : id);  // host validates; clamp keeps the access in bounds
  const float4* e = reinterpret_cast<const float4*>(emb + id * H);
  const float4* p = reinterpret_cast<const float4*>(pos + (long long)l * H);
  float4* o = reinterpret_cast<float4*>(x + (long long)row * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const float4 a = e[i], b = p[i];
    o[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

__global__ void add_speaker_kernel(const float* __restrict__ xin, float* __restrict__ x, const float* __restrict__ spk,
                                   const int64_t* __restrict__ speaker, int n_spk_ids, int n_speakers, int L, int H) {
  const int row = blockIdx.x;
  const int b = row / L;
  long long sid = speaker[n_spk_ids == 1 ? 0 : b];
  sid = sid < 0 ? 0 : (sid >= n_speakers ? n_speakers - 1 : sid);
  const float4* e = reinterpret_cast<const float4*>(spk + sid * H);
  float4* o = reinterpret_cast<float4*>(x + (long long)row * H);
  const float4* in = reinterpret_cast<const float4*>(xin + (long long)row * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const float4 a = in[i], c = e[i];
    o[i] = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
  }
}

// ---------------------------------------------------------------- variance-predictor positions
// positions = cumsum(x[..., 0] != 0) * (x[..., 0] != 0)  (reference U/function.py:28-38 via U/sublayers.py:64),
// y = x + alpha * table[positions] (U/layers.py:497-498).  One wavefront per utterance: shuffle scan + carry.
__global__ __launch_bounds__(64) void var_positions_kernel(const float* __restrict__ x, int32_t* __restrict__ posbuf, int L, int H) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int carry = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    const int nz = (l < L && x[((long long)b * L + l) * H] != 0.f) ? 1 : 0;
    int v = nz;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (l < L) posbuf[b * L + l] = nz ? carry + v : 0;
    carry += __shfl(v, 63);
  }
}

__global__ void var_pos_add_kernel(const float* __restrict__ x, const int32_t* __restrict__ posbuf, const float* __restrict__ table,
                                   int table_rows, const float* __restrict__ alpha, float* __restrict__ y, int H) {
  const int row = blockIdx.x;
  int pos = posbuf[row];
  pos = pos >= table_rows ? table_rows - 1 : pos;  // host guarantees L + 1 <= table_rows
  const float a = alpha[0];
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * H);
  const float4* tr = reinterpret_cast<const float4*>(table + (long long)pos * H);
  float4* yr = reinterpret_cast<float4*>(y + (long long)row * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const float4 v = xr[i], t = tr[i];
    // x + (alpha * table): two roundings, as the reference computes it
    yr[i] = make_float4(v.x + __fmul_rn(a, t.x), v.y + __fmul_rn(a, t.y), v.z + __fmul_rn(a, t.z), v.w + __fmul_rn(a, t.w));
  }
}

// ---------------------------------------------------------------- small-odim Linear (one wavefront per row)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     const int32_t* __restrict__ lens, int rows, int L, int C, int O) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * C);
  bool masked = false;
  if (lens) {
    const int b = row / L;
    masked = (row - b * L) >= lens[b];
  }
  for (int o = 0; o < O; ++o) {
    const float4* wr = reinterpret_cast<const float4*>(w + (long long)o * C);
    float s = 0.f;
    for (int i = lane; i < C / 4; i += 64) {
      const float4 a = xr[i], c = wr[i];
      s += (a.x * c.x + a.y * c.y) + (a.z * c.z + a.w * c.w);
    }
    s = wave_sum(s);
    if (lane == 0) out[(long long)row * O + o] = masked ? 0.f : s + bias[o];
  }
}

// ---------------------------------------------------------------- duration rounding + inclusive scan
// duration_rounded = clamp(round(exp(log_d) - 1) * d_control, min = 0)   (reference U/layers.py:218-221;
// torch.round = half-to-even = rintf).  Repeat count = max(int(d), 0) (U/layers.py:448-449).
// One wavefront per utterance: __shfl_up inclusive scan over 64 phonemes at a time with a running carry.
__global__ __launch_bounds__(64) void duration_kernel(const float* __restrict__ log_d, float d_control, float* __restrict__ dur,
                                                      int32_t* __restrict__ cum, int64_t* __restrict__ mel64,
                                                      int32_t* __restrict__ mel32, int L) {
  const int b = blockIdx.x, lane = threadIdx.x;
  // Repeat counts are capped at 2^20 frames per phoneme (inf / huge exp(log_d) would overflow the int cast) and the running sum is kept
  // in 64 bits, so a runaway prediction reaches the host as a huge mel_lens value that its T check rejects instead of a wrapped one.
  long long carry = 0;
  for (int l0 = 0; l0 < L; l0 += 64) {
    const int l = l0 + lane;
    float d = 0.f;
    if (l < L) {
      const float e = __fsub_rn(expf(log_d[b * L + l]), 1.0f);
      d = fmaxf(__fmul_rn(rintf(e), d_control), 0.f);
      dur[b * L + l] = d;
    }
    int v = d == d ? (int)fminf(d, 1048576.f) : 0;  // <= 2^20 each, 64 lanes: the wave's scan stays below 2^26
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (l < L) cum[b * L + l] = (int32_t)min(carry + v, 0x7fffffffLL);
    carry += __shfl(v, 63);
  }
  if (lane == 0) {
    mel64[b] = carry;
    mel32[b] = (int32_t)min(carry, 0x7fffffffLL);
  }
}

// ---------------------------------------------------------------- pitch / energy buckets + embedding add
// f0 bucket: reference U/layers.py:145-154 + U/function.py:178-187 (constants :9-13); energy: torch.bucketize
// (right = False) on linspace bins (U/layers.py:169).  Every fp32 operation is kept a separate rounding
// (__f*_rn: no FMA contraction) so that the integer results track the reference's elementwise torch ops.
__global__ __launch_bounds__(128) void variance_embed_kernel(float* __restrict__ x, float* __restrict__ pitch_pred,
                                                             const float* __restrict__ energy_pred, float p_control,
                                                             float e_control, float f0_mean, float f0_std,
                                                             const float* __restrict__ energy_bins, int n_bins,
                                                             const float* __restrict__ pitch_emb, const float* __restrict__ energy_emb,
                                                             int32_t* __restrict__ pitch_idx, int32_t* __restrict__ energy_idx, int H,
                                                             float mel_min, float mel_range, int pitch_mode,
                                                             const float* __restrict__ pitch_bins, int feat) {
  const int row = blockIdx.x;
  float f0 = 0.f, uvl = 0.f;
  int pidx = 0;
  // feat: bit 0 = pitch, bit 1 = energy take part in this pass (the other one lives at the other level: phoneme / frame, U/layers.py:226-257)
  if (!(feat & 1)) {
  } else if (pitch_mode == 2) {  // use_uv False (U/layers.py:155-157): bucketize(prediction * control, pitch_bins), right = False
    f0 = __fmul_rn(pitch_pred[row], p_control);
    int lo = 0, hi = n_bins - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (pitch_bins[mid] < f0) lo = mid + 1; else hi = mid;
    }
    pidx = lo;
  } else {
    f0 = __fmul_rn(pitch_pred[2 * row], p_control);
    uvl = __fmul_rn(pitch_pred[2 * row + 1], p_control);
    float f0d = pitch_mode == 1 ? powf(2.0f, f0) : __fadd_rn(__fmul_rn(f0, f0_std), f0_mean);  // U/layers.py:148-151
    if (uvl > 0.f) f0d = 0.f;
    float mel = __fmul_rn(1127.0f, logf(__fadd_rn(1.0f, __fdiv_rn(f0d, 700.0f))));
    if (mel > 0.f) mel = __fadd_rn(__fdiv_rn(__fmul_rn(__fsub_rn(mel, mel_min), 254.0f), mel_range), 1.0f);
    if (mel <= 1.f) mel = 1.f;
    if (mel > 255.f) mel = 255.f;
    pidx = (int)__fadd_rn(mel, 0.5f);
    pidx = pidx < 0 ? 0 : (pidx > n_bins - 1 ? n_bins - 1 : pidx);  // NaN guard only; the clamps above bound it
  }
  int eidx = 0;
  if (feat & 2) {
    const float e = __fmul_rn(energy_pred[row], e_control);
    int lo = 0, hi = n_bins - 1;  // first index with bins[idx] >= e; n_bins - 1 boundaries
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (energy_bins[mid] < e) lo = mid + 1; else hi = mid;
    }
    eidx = lo;
  }
  if (threadIdx.x == 0) {
    if ((feat & 1) && pitch_mode != 2) {  // (mode 2 hands back the prediction before the control, as the reference does: U/layers.py:156,162)
      pitch_pred[2 * row] = f0;
      pitch_pred[2 * row + 1] = uvl;
    }
    if (feat & 1) pitch_idx[row] = pidx;
    if (feat & 2) energy_idx[row] = eidx;
  }
  float4* xr = reinterpret_cast<float4*>(x + (long long)row * H);
  const float4* pe = reinterpret_cast<const float4*>(pitch_emb + (long long)pidx * H);
  const float4* ee = reinterpret_cast<const float4*>(energy_emb + (long long)eidx * H);
  if (feat == 3) {
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
      const float4 a = xr[i], p = pe[i], q = ee[i];
      xr[i] = make_float4((a.x + p.x) + q.x, (a.y + p.y) + q.y, (a.z + p.z) + q.z, (a.w + p.w) + q.w);
    }
  } else {
    const float4* oe = (feat & 1) ? pe : ee;
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
      const float4 a = xr[i], p = oe[i];
      xr[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
  }
}

// ---------------------------------------------------------------- length regulator (+ decoder position add)
// Reference U/layers.py:429-451: repeat phoneme row i duration[i] times, concatenate, zero-pad to the batch
// maximum; here frame t finds its phoneme by binary search in the inclusive duration scan.
__global__ __launch_bounds__(128) void length_regulate_kernel(const float* __restrict__ x, const int32_t* __restrict__ cum,
                                                              const int32_t* __restrict__ mel_lens, const float* __restrict__ pos,
                                                              float* __restrict__ y, int L, int T, int H) {
  const int t = blockIdx.x, b = blockIdx.y;
  float4* yr = reinterpret_cast<float4*>(y + ((long long)b * T + t) * H);
  if (!pos) {  // plain length regulator (frame-level pitch / energy follow before the decoder's positions, U/layers.py:241-257)
    if (t >= mel_lens[b]) {
      for (int i = threadIdx.x; i < H / 4; i += blockDim.x) yr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
    const int32_t* c = cum + b * L;
    int lo = 0, hi = L - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (c[mid] > t) hi = mid; else lo = mid + 1;
    }
    const float4* xr = reinterpret_cast<const float4*>(x + ((long long)b * L + lo) * H);
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) yr[i] = xr[i];
    return;
  }
  const float4* pr = reinterpret_cast<const float4*>(pos + (long long)t * H);
  if (t >= mel_lens[b]) {
    for (int i = threadIdx.x; i < H / 4; i += blockDim.x) yr[i] = pr[i];
    return;
  }
  const int32_t* c = cum + b * L;
  int lo = 0, hi = L - 1;  // smallest i with cum[i] > t
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (c[mid] > t) hi = mid; else lo = mid + 1;
  }
  const float4* xr = reinterpret_cast<const float4*>(x + ((long long)b * L + lo) * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const float4 a = xr[i], p = pr[i];
    yr[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

// y[b, t, :] += pos[t, :] (the decoder's position add, when the length regulator could not carry it)
__global__ __launch_bounds__(128) void add_positions_kernel(float* __restrict__ y, const float* __restrict__ pos, int T, int H) {
  const int t = blockIdx.x, b = blockIdx.y;
  const float4* pr = reinterpret_cast<const float4*>(pos + (long long)t * H);
  float4* yr = reinterpret_cast<float4*>(y + ((long long)b * T + t) * H);
  for (int i = threadIdx.x; i < H / 4; i += blockDim.x) {
    const float4 a = yr[i], p = pr[i];
    yr[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

// ---------------------------------------------------------------- ragged-batch row limits
__global__ void act_rows_kernel(const int32_t* __restrict__ lens, int32_t* __restrict__ out, int B, int add, int mul, long long cap, long long add_rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    const long long v = ((long long)lens[i] + add) * mul + add_rows;
    out[i] = (int32_t)(v < cap ? v : cap);
  }
}

// ---------------------------------------------------------------- [B, C, T] -> [B, T, C]
// Split-precision weight image [Cout][KW][nchunk][32 bf16 hi | 32 bf16 lo] (packer.pack_x3) -> MFMA-fragment order
// [ceil(Cout/32)][KW][nchunk][k-step 0..1][hi | lo][lane 0..63][8 bf16]: the 16 bytes lane l of a wave needs as B operand of
// v_mfma_f32_32x32x16_bf16 for output column 32 t + (l & 31), k = 16 ks + 8 (l >> 5) .. + 7.  One thread moves one such
// 16-byte group; columns >= Cout are zero.  Runs once per weight tensor at load time.
__global__ void x3_to_frag_kernel(const uint4* __restrict__ x3, uint4* __restrict__ frag, int Cout, int KW, int nchunk, long long groups) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const int lane = (int)(g & 63);
  const int hl = (int)((g >> 6) & 1), ks = (int)((g >> 7) & 1);
  long long r = g >> 8;
  const int c = (int)(r % nchunk); r /= nchunk;
  const int j = (int)(r % KW);
  const int t = (int)(r / KW);
  const int n = t * 32 + (lane & 31);
  uint4 v = make_uint4(0, 0, 0, 0);
  // one x3 row = 64 bf16 = 8 groups of 16 bytes: [hi k 0-7, 8-15, 16-23, 24-31 | lo ...]
  if (n < Cout) v = x3[(((long long)n * KW + j) * nchunk + c) * 8 + hl * 4 + ks * 2 + (lane >> 5)];
  frag[g] = v;
}

// fp32 weights [Cout][KW * Cin] (tap-major) -> MFMA-fragment order [ceil(Cout/32)][KW][nchunk][q 0..3][lane 0..63][4 floats]: the float4
// lane l = (li, lh) of a wave feeds to the four v_mfma_f32_32x32x2_f32 of k-group q as B operand of column 32 t + li:
// k = 32 c + 8 q + 4 lh .. + 3 (conv_gemm.hip: the same float4 its LDS weight tile hands that lane).  Columns >= Cout and channels
// >= Cin are zero.  Runs once per weight tensor at load time.
__global__ void f32_to_frag_kernel(const float* __restrict__ w, float4* __restrict__ frag, int Cout, int KW, int Cin, int nchunk, long long groups) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= groups) return;
  const int lane = (int)(g & 63), q = (int)((g >> 6) & 3);
  long long r = g >> 8;
  const int c = (int)(r % nchunk); r /= nchunk;
  const int j = (int)(r % KW);
  const int t = (int)(r / KW);
  const int n = t * 32 + (lane & 31);
  const int k0 = c * 32 + q * 8 + (lane >> 5) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < Cout) {
    const float* row = w + ((long long)n * KW + j) * Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (k0 + i < Cin) v[i] = row[k0 + i];
  }
  frag[g] = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------- tempo change (WSOLA) for `audio_speed_change`
// Reference API/utils.py:163-172 shells out to ffmpeg's `atempo` filter (tempo without pitch change).  ffmpeg is not part of this build,
// so this is waveform-similarity overlap-add (Verhelst & Roelands 1993), the algorithm family behind that filter, restated from the
// published method -- PARITY UNPINNED against the reference; it mirrors e2e_tts_amd/api.py: time_stretch_wsola step for step (float64).
// One workgroup walks the output frames of one signal in order (frame i's best offset fixes the template of frame i + 1):
//   target_i = rint(i hop_in) + delta + n;  best_i = target_i - delta + argmax_lag sum_k seg[lag + k] tmpl[k]  (tmpl = natural continuation
//   of frame i - 1);  out[i hop_out .. + n) += xp[best_i ..] * hann;   xp = [0 (delta + n) | x | 0 ...].
__global__ __launch_bounds__(1024) void wsola_kernel(const int16_t* __restrict__ x, long long n_in, int16_t* __restrict__ out, long long n_out,
                                                     double speed, int n, int delta, int n_frames) {
  constexpr int NT = 1024;
  extern __shared__ __attribute__((aligned(16))) double ws_smem[];
  double* seg = ws_smem;                 // [n + 2 delta]  candidate region around the target
  double* tmpl = seg + n + 2 * delta;    // [n]            natural continuation of the previous frame
  double* win = tmpl + n;                // [n]            periodic Hann window, np.hanning(n + 1)[:n]
  double* open = win + n;                // [n / 2]        the half frame of output that the next frame will still add to
  double* cval = open + n;               // [NT] reduction scratch
  int* cidx = reinterpret_cast<int*>(cval + NT);
  const int tid = threadIdx.x;
  const int hop_out = n / 2;
  const double hop_in = hop_out * speed;
  const long long lead = delta + n;
  auto xp = [&](long long idx) -> double {
    const long long j = idx - lead;
    return (j >= 0 && j < n_in) ? (double)x[j] : 0.0;
  };
  auto emit = [&](long long o, double v) {   // out = clip(rint(sum)) as int16 (api.py: audio_speed_change)
    if (o < n_out) {
      v = rint(v);
      v = v < -32768.0 ? -32768.0 : (v > 32767.0 ? 32767.0 : v);
      out[o] = (int16_t)v;
    }
  };
  for (int k = tid; k < n; k += NT) {
    win[k] = 0.5 - 0.5 * cos(6.283185307179586476925286766559 * (double)k / (double)n);
    if (k < hop_out) open[k] = 0.0;
  }
  __syncthreads();
  long long pos = lead;
  for (int i = 0; i < n_frames; ++i) {
    const long long target = llrint((double)i * hop_in) + lead;   // round-half-even, as Python's round()
    long long best = target;
    if (i > 0) {
      const long long lo = target - delta;
      for (int k = tid; k < n + 2 * delta; k += NT) seg[k] = xp(lo + k);
      for (int k = tid; k < n; k += NT) tmpl[k] = xp(pos + k);
      __syncthreads();
      double bv = -1.0e300;
      int bi = 0x7fffffff;
      for (int lag = tid; lag <= 2 * delta; lag += NT) {
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;   // four chains: the fp64 FMA latency, not its rate, bounds a single one
        int k = 0;
        for (; k + 3 < n; k += 4) {
          c0 += seg[lag + k] * tmpl[k];
          c1 += seg[lag + k + 1] * tmpl[k + 1];
          c2 += seg[lag + k + 2] * tmpl[k + 2];
          c3 += seg[lag + k + 3] * tmpl[k + 3];
        }
        for (; k < n; ++k) c0 += seg[lag + k] * tmpl[k];
        const double c = (c0 + c1) + (c2 + c3);
        if (c > bv) { bv = c; bi = lag; }   // lags visited in increasing order: the first maximum of this thread's lags
      }
      cval[tid] = bv;
      cidx[tid] = bi;
      __syncthreads();
      for (int st = NT / 2; st > 0; st >>= 1) {
        if (tid < st) {
          const double ov = cval[tid + st];
          const int oi = cidx[tid + st];
          if (ov > cval[tid] || (ov == cval[tid] && oi < cidx[tid])) { cval[tid] = ov; cidx[tid] = oi; }   // np.argmax: first maximum
        }
        __syncthreads();
      }
      best = lo + cidx[0];
      __syncthreads();  // cidx[0] read by everyone before the next frame overwrites it
    }
    // out[i hop_out .. + n) += xp[best ..] * win: with 50 % overlap the first half is final now (previous frame's second half + this
    // frame's first), the second half waits for frame i + 1
    for (int k = tid; k < hop_out; k += NT) {
      emit((long long)i * hop_out + k, open[k] + xp(best + k) * win[k]);
      open[k] = xp(best + k + hop_out) * win[k + hop_out];
    }
    __syncthreads();
    pos = best + hop_out;
  }
  for (int k = tid; k < hop_out; k += NT) emit((long long)n_frames * hop_out + k, open[k]);   // the tail of the last frame
}

// ---- iSTFTNet tail (reference V/generator.py:107-113 + src/tools/stft.py:138-148)
// x = leaky_relu(x, 0.01); x = ReflectionPad1d((1, 0))(x): frame 0 of the padded signal is frame 1 of the input.
__global__ void reflect_lrelu_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long n, int c4, float slope) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index inside one padded utterance
  const int b = blockIdx.y;
  if (i >= (n + 1) * c4) return;
  const long long f = i / c4;
  const int c = (int)(i - f * c4);
  const long long src = f == 0 ? 1 : f - 1;
  float4 v = in[((long long)b * n + src) * c4 + c];
  v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);
  out[((long long)b * (n + 1) + f) * c4 + c] = v;
}

// ---- Conformer block pieces (reference U/blocks/conformer.py, U/blocks/utils.py:196-219) ----
__device__ __forceinline__ float swish1(float v) { return v * (1.0f / (1.0f + expf(-v))); }

// GLU over channels (conformer.py:472, utils.py:217-219): out[row, c] = in[row, c] * sigmoid(in[row, C + c])
__global__ void glu_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long rows, int c4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * c4) return;
  const long long row = i / c4;
  const int c = (int)(i - row * c4);
  const float4 a = in[row * 2 * c4 + c], g = in[row * 2 * c4 + c4 + c];
  float4 v;
  v.x = a.x * (1.0f / (1.0f + expf(-g.x))); v.y = a.y * (1.0f / (1.0f + expf(-g.y)));
  v.z = a.z * (1.0f / (1.0f + expf(-g.z))); v.w = a.w * (1.0f / (1.0f + expf(-g.w)));
  out[i] = v;
}

// Depthwise Conv1d(k, "same", no bias) -> BatchNorm1d (eval; folded into w / b by the packer) -> Swish (conformer.py:473-475) on
// channels-last rows.  The convolution runs over the whole padded length N: nothing masks padded frames inside a Conformer block.
// w: [k][C] tap-major.  One thread = 4 channels of one frame; the k re-reads of a frame come from L1 / L2.
__global__ void dwconv_swish_kernel(const float4* __restrict__ in, const float4* __restrict__ w, const float4* __restrict__ bias,
                                    float4* __restrict__ out, int N, int c4, int k) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (n, c) inside one utterance
  const int b = blockIdx.y;
  if (i >= (long long)N * c4) return;
  const int n = (int)(i / c4), c = (int)(i - (long long)n * c4);
  const float4* xb = in + (long long)b * N * c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int half = (k - 1) / 2;
  for (int j = 0; j < k; ++j) {
    const int t = n - half + j;
    if (t < 0 || t >= N) continue;
    const float4 x = xb[(long long)t * c4 + c], ww = w[j * c4 + c];
    acc.x = fmaf(x.x, ww.x, acc.x); acc.y = fmaf(x.y, ww.y, acc.y); acc.z = fmaf(x.z, ww.z, acc.z); acc.w = fmaf(x.w, ww.w, acc.w);
  }
  const float4 bb = bias[c];
  float4 v;
  v.x = swish1(acc.x + bb.x); v.y = swish1(acc.y + bb.y); v.z = swish1(acc.z + bb.z); v.w = swish1(acc.w + bb.w);
  out[(long long)b * N * c4 + i] = v;
}

// GLU -> depthwise Conv1d(K) -> BatchNorm (folded) -> Swish in ONE pass over the pointwise convolution's 2 C output channels (round 3; the
// two kernels above cost 21 + 54 us per layer at B = 32, T = 768, against ~20 us of HBM time for what they move).  A workgroup owns a
// tile of 64 frames x CT channels: the GLU of the tile's rows and of (K - 1) / 2 halo rows either side goes to LDS once (every input
// element's sigmoid computed once, not K times); a thread then owns 4 channels of FPT consecutive frames, holds its K weight float4s in
// registers, and streams the FPT + K - 1 rows it needs past its FPT accumulators -- one LDS read per row, the products added in tap
// order as dwconv_swish_kernel adds them (rows outside [0, N) are zeros: fma(0, w, acc) = acc), so the two forms give the same bits.
template <int K, int CT>
__global__ __launch_bounds__(256) void dwconv_glu_swish_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out, int N, int C) {
  constexpr int FR = 64, NC4 = CT / 4, NFG = 256 / NC4, FPT = FR / NFG, HALF = (K - 1) / 2, ROWS = FR + K - 1, LD = CT + 4;
  static_assert(256 % NC4 == 0 && FR % NFG == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float dw_xs[];   // [ROWS][LD]
  const int tid = threadIdx.x, c4 = tid % NC4, fg = tid / NC4;
  const int b = blockIdx.z, c0 = blockIdx.y * CT, t0 = blockIdx.x * FR;
  const float* inb = in + (long long)b * N * 2 * C;
  for (int i = tid; i < ROWS * NC4; i += 256) {
    const int r = i / NC4, cc = i % NC4;
    const int t = t0 - HALF + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < N) {
      const float4 a = *reinterpret_cast<const float4*>(inb + (long long)t * 2 * C + c0 + cc * 4);
      const float4 g = *reinterpret_cast<const float4*>(inb + (long long)t * 2 * C + C + c0 + cc * 4);
      v.x = a.x * (1.0f / (1.0f + expf(-g.x))); v.y = a.y * (1.0f / (1.0f + expf(-g.y)));   // glu_kernel's formula
      v.z = a.z * (1.0f / (1.0f + expf(-g.z))); v.w = a.w * (1.0f / (1.0f + expf(-g.w)));
    }
    *reinterpret_cast<float4*>(dw_xs + r * LD + cc * 4) = v;
  }
  float4 wk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wk[k] = *reinterpret_cast<const float4*>(w + (long long)k * C + c0 + c4 * 4);
  __syncthreads();
  float4 acc[FPT];
#pragma unroll
  for (int n = 0; n < FPT; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* xr = dw_xs + (fg * FPT) * LD + c4 * 4;
#pragma unroll
  for (int j = 0; j < FPT + K - 1; ++j) {
    const float4 x = *reinterpret_cast<const float4*>(xr + j * LD);
#pragma unroll
    for (int n = 0; n < FPT; ++n) {
      const int k = j - n;   // compile-time after unrolling
      if (k >= 0 && k < K) {
        acc[n].x = fmaf(x.x, wk[k].x, acc[n].x); acc[n].y = fmaf(x.y, wk[k].y, acc[n].y);
        acc[n].z = fmaf(x.z, wk[k].z, acc[n].z); acc[n].w = fmaf(x.w, wk[k].w, acc[n].w);
      }
    }
  }
  const float4 bb = *reinterpret_cast<const float4*>(bias + c0 + c4 * 4);
#pragma unroll
  for (int n = 0; n < FPT; ++n) {
    const int t = t0 + fg * FPT + n;
    if (t < N) {
      float4 v;
      v.x = swish1(acc[n].x + bb.x); v.y = swish1(acc[n].y + bb.y); v.z = swish1(acc[n].z + bb.z); v.w = swish1(acc[n].w + bb.w);
      *reinterpret_cast<float4*>(out + ((long long)b * N + t) * C + c0 + c4 * 4) = v;
    }
  }
}

// spec = exp(q[:bins]), phase = sin(q[bins:2 bins]) (V/generator.py:110-111); X = spec * exp(j phase) (stft.py:141)
__global__ void istft_prep_kernel(const float* __restrict__ q, int ldq, float* __restrict__ specphase, float2* __restrict__ ri,
                                  long long total, int bins) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (b * F + f) * bins + k
  if (i >= total) return;
  const long long row = i / bins;
  const int k = (int)(i - row * bins);
  const float spec = expf(q[row * ldq + k]);
  const float phase = sinf(q[row * ldq + bins + k]);
  specphase[row * (2 * bins) + k] = spec;
  specphase[row * (2 * bins) + bins + k] = phase;
  ri[i] = make_float2(spec * cosf(phase), spec * sinf(phase));
}

// torch.istft(X, n_fft, hop, win_length = n_fft, window = hann_window(n_fft) [periodic], center = True): per frame the real
// inverse DFT (1/N normalisation, imaginary parts of DC and Nyquist ignored), times the window, overlap-added, divided by the
// overlap-added squared window, with n_fft/2 samples trimmed on both sides -> hop * (F - 1) samples.  One thread per sample.
__global__ void istft_ola_kernel(const float2* __restrict__ ri, float* __restrict__ wav, int16_t* __restrict__ pcm, long long F,
                                 int nfft, int hop, long long nsamp) {
  extern __shared__ float tw[];  // cos(2 pi m / N), sin(2 pi m / N), window[m]
  for (int m = threadIdx.x; m < nfft; m += blockDim.x) {
    const float a = 6.283185307179586f * (float)m / (float)nfft;
    tw[m] = cosf(a);
    tw[nfft + m] = sinf(a);
    tw[2 * nfft + m] = 0.5f - 0.5f * cosf(a);
  }
  __syncthreads();
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= nsamp) return;
  const int bins = nfft / 2 + 1;
  const long long np = n + nfft / 2;  // index before the centre trim
  long long f_hi = np / hop;
  if (f_hi > F - 1) f_hi = F - 1;
  long long f_lo = (np - nfft + hop) / hop;  // smallest f with np - f hop <= nfft - 1
  if (f_lo < 0) f_lo = 0;
  float num = 0.f, den = 0.f;
  for (long long f = f_lo; f <= f_hi; ++f) {
    const int m = (int)(np - f * hop);
    const float2* x = ri + ((long long)b * F + f) * bins;
    float acc = x[0].x + ((m & 1) ? -x[bins - 1].x : x[bins - 1].x);
    for (int k = 1; k < bins - 1; ++k) {
      const int idx = (k * m) & (nfft - 1);  // n_fft is a power of two
      acc += 2.0f * (x[k].x * tw[idx] - x[k].y * tw[nfft + idx]);
    }
    const float w = tw[2 * nfft + m];
    num += w * (acc / (float)nfft);
    den += w * w;
  }
  const float y = num / den;
  if (wav) wav[(long long)b * nsamp + n] = y;
  if (pcm) pcm[(long long)b * nsamp + n] = (int16_t)(int32_t)(y * 32768.0f);
}

__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    tile[i][tx] = (c < C && t < T) ? in[((long long)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < T && c < C) out[((long long)b * T + t) * C + c] = tile[tx][i];
  }
}

// ---------------------------------------------------------------- conv_post: lrelu(0.01) -> Conv1d(C -> 1, k7) -> tanh
// Reference V/generator.py:49-51 (F.leaky_relu default slope 0.01).  Also emits the int16 PCM exactly as
// TTS.combine_audio does (API/utils.py:112,117): fp32 wav * 32768.0, then numpy's float -> int16 cast,
// i.e. truncation toward zero through a 32-bit integer (so exactly +1.0 wraps to -32768, as numpy does on x86).
template <int TPB>
__global__ __launch_bounds__(TPB) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ wav,
                                                        int16_t* __restrict__ pcm, long long N, int C, int KW,
                                                        const int32_t* __restrict__ act_rows, const RowMap rm,
                                                        const float* __restrict__ x2, const float* __restrict__ x3, const float* __restrict__ x4,
                                                        const float x_div) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldx = C + 4;
  const int pad = (KW - 1) / 2;
  const int rows = TPB + KW - 1;
  float* xs = sm;
  float* ws = sm + rows * ldx;
  int b, blk;   // ragged batch, lengths known on the host: a compact 1-D grid of the blocks that have samples to write (kernels.h: RowMap)
  if (rm.n > 0) {
    if (!rowmap_find(rm, (int)blockIdx.x, b, blk)) return;
  } else {
    b = blockIdx.y;
    blk = blockIdx.x;
  }
  const long long t0 = (long long)blk * TPB;
  if (act_rows && t0 >= act_rows[b]) return;   // samples nobody needs (uniform for the block, before the barrier)
  const float* xb = x + (long long)b * N * C;
  const int c4n = C / 4;
  for (int i = threadIdx.x; i < rows * c4n; i += TPB) {
    const int r = i / c4n, c = (i - r * c4n) * 4;
    const long long t = t0 - pad + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < N) {
      v = *reinterpret_cast<const float4*>(xb + t * C + c);
      if (x2) {   // the deferred join of the last stage's ResBlocks: accum_div_kernel's additions and division, in its order
        const long long o = (long long)b * N * C + t * C + c;
        const float4 a2 = *reinterpret_cast<const float4*>(x2 + o);
        v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w;
        if (x3) { const float4 a3 = *reinterpret_cast<const float4*>(x3 + o); v.x += a3.x; v.y += a3.y; v.z += a3.z; v.w += a3.w; }
        if (x4) { const float4 a4 = *reinterpret_cast<const float4*>(x4 + o); v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
        if (x_div != 1.0f) { v.x = v.x / x_div; v.y = v.y / x_div; v.z = v.z / x_div; v.w = v.w / x_div; }
      }
      v.x = v.x >= 0.f ? v.x : v.x * 0.01f;
      v.y = v.y >= 0.f ? v.y : v.y * 0.01f;
      v.z = v.z >= 0.f ? v.z : v.z * 0.01f;
      v.w = v.w >= 0.f ? v.w : v.w * 0.01f;
    }
    *reinterpret_cast<float4*>(xs + r * ldx + c) = v;
  }
  for (int i = threadIdx.x; i < KW * C; i += TPB) ws[i] = w[i];
  __syncthreads();
  const long long t = t0 + threadIdx.x;
  if (t >= N) return;
  float acc = 0.f;
  for (int j = 0; j < KW; ++j) {
    const float* xr = xs + (threadIdx.x + j) * ldx;
    const float* wr = ws + j * C;
    for (int c = 0; c < C; c += 4) {
      const float4 a = *reinterpret_cast<const float4*>(xr + c);
      const float4 ww = *reinterpret_cast<const float4*>(wr + c);
      acc += (a.x * ww.x + a.y * ww.y) + (a.z * ww.z + a.w * ww.w);
    }
  }
  const float v = tanhf(acc + bias[0]);
  if (wav) wav[(long long)b * N + t] = v;
  if (pcm) pcm[(long long)b * N + t] = (int16_t)(int32_t)__fmul_rn(v, 32768.0f);
}

}  // namespace

#define CHECK_LAUNCH(name) (hipGetLastError() == hipSuccess ? nullptr : name ": launch failed")

const char* launch_layernorm(const float* x, float* y, const float* gamma, const float* beta, const int32_t* lens,
                             int B, int N, int C, float eps, hipStream_t s) {
  if (!x || !y || !gamma || !beta) return "layernorm: null pointer";
  if (C % 4 || C <= 0 || C > 1024) return "layernorm: C must be a multiple of 4 in (0, 1024]";
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) return "layernorm: unaligned pointer";
  const int rows = B * N;
  if (rows <= 0) return "layernorm: bad dims";
  dim3 grid((rows + 3) / 4);
  const int nv = (C / 4 + 63) / 64;
  switch (nv) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, rows, N, C, eps); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, rows, N, C, eps); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, rows, N, C, eps); break;
    default: hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, x, y, gamma, beta, lens, rows, N, C, eps); break;
  }
  return CHECK_LAUNCH("layernorm");
}

const char* launch_embed(const int64_t* ids, const float* emb, const float* pos, float* x, int B, int L, int H,
                         int n_rows, hipStream_t s) {
  if (!ids || !emb || !pos || !x) return "embed: null pointer";
  if (H % 4 || B <= 0 || L <= 0) return "embed: bad dims";
  hipLaunchKernelGGL(embed_kernel, dim3(B * L), dim3(128), 0, s, ids, emb, pos, x, L, H, n_rows);
  return CHECK_LAUNCH("embed");
}

const char* launch_add_speaker(const float* xin, float* x, const float* spk, const int64_t* speaker, int n_spk_ids, int n_speakers,
                               int B, int L, int H, hipStream_t s) {
  if (!xin || !x || !spk || !speaker) return "add_speaker: null pointer";
  if (n_spk_ids != 1 && n_spk_ids != B) return "add_speaker: speaker must have 1 or B entries";
  hipLaunchKernelGGL(add_speaker_kernel, dim3(B * L), dim3(128), 0, s, xin, x, spk, speaker, n_spk_ids, n_speakers, L, H);
  return CHECK_LAUNCH("add_speaker");
}

const char* launch_rowdot(const float* x, const float* w, const float* b, float* out, const int32_t* lens, int B, int L,
                          int C, int O, hipStream_t s) {
  if (!x || !w || !b || !out) return "rowdot: null pointer";
  if (C % 4 || O <= 0 || O > 4) return "rowdot: bad dims";
  const int rows = B * L;
  hipLaunchKernelGGL(rowdot_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, out, lens, rows, L, C, O);
  return CHECK_LAUNCH("rowdot");
}

const char* launch_duration(const float* log_d, float d_control, float* dur, int32_t* cum, int64_t* mel_lens64,
                            int32_t* mel_lens32, int B, int L, hipStream_t s) {
  if (!log_d || !dur || !cum || !mel_lens64 || !mel_lens32) return "duration: null pointer";
  hipLaunchKernelGGL(duration_kernel, dim3(B), dim3(64), 0, s, log_d, d_control, dur, cum, mel_lens64, mel_lens32, L);
  return CHECK_LAUNCH("duration");
}

const char* launch_variance_embed(float* x, float* pitch_pred, const float* energy_pred, float p_control, float e_control,
                                  float f0_mean, float f0_std, const float* energy_bins, int n_bins, const float* pitch_emb,
                                  const float* energy_emb, int32_t* pitch_idx, int32_t* energy_idx, int B, int L, int H,
                                  hipStream_t s, int pitch_mode, const float* pitch_bins, int feat) {
  if (!x || !pitch_pred || !energy_pred || !energy_bins || !pitch_emb || !energy_emb || !pitch_idx || !energy_idx)
    return "variance_embed: null pointer";
  if (feat < 1 || feat > 3) return "variance_embed: bad feature mask";
  if (pitch_mode < 0 || pitch_mode > 2 || (pitch_mode == 2 && !pitch_bins)) return "variance_embed: bad pitch mode";
  if (n_bins != 256) return "variance_embed: the f0 coarse coding is defined for 256 bins (reference U/function.py:9)";
  // f0_mel_min / max: numpy float64 constants, used by torch as fp32 scalars (U/function.py:12-13,180)
  const double mel_min = 1127.0 * log(1.0 + 50.0 / 700.0), mel_max = 1127.0 * log(1.0 + 1100.0 / 700.0);
  hipLaunchKernelGGL(variance_embed_kernel, dim3(B * L), dim3(128), 0, s, x, pitch_pred, energy_pred, p_control, e_control,
                     f0_mean, f0_std, energy_bins, n_bins, pitch_emb, energy_emb, pitch_idx, energy_idx, H, (float)mel_min,
                     (float)(mel_max - mel_min), pitch_mode, pitch_bins, feat);
  return CHECK_LAUNCH("variance_embed");
}

const char* launch_length_regulate(const float* x, const int32_t* cum, const int32_t* mel_lens, const float* pos, float* y,
                                   int B, int L, int T, int H, hipStream_t s) {
  if (!x || !cum || !mel_lens || !y) return "length_regulate: null pointer";   // pos == nullptr: no position add
  if (T <= 0 || H % 4) return "length_regulate: bad dims";
  hipLaunchKernelGGL(length_regulate_kernel, dim3(T, B), dim3(128), 0, s, x, cum, mel_lens, pos, y, L, T, H);
  return CHECK_LAUNCH("length_regulate");
}

const char* launch_add_positions(float* y, const float* pos, int B, int T, int H, hipStream_t s) {
  if (!y || !pos) return "add_positions: null pointer";
  if (T <= 0 || H % 4) return "add_positions: bad dims";
  hipLaunchKernelGGL(add_positions_kernel, dim3(T, B), dim3(128), 0, s, y, pos, T, H);
  return CHECK_LAUNCH("add_positions");
}

// S = (S + Sj) [/ div]: joins the stage sums of ResBlocks that ran side by side (engine.hip, small batches); the operations the
// accumulating epilogues of conv_gemm / resblock_pair / resblock_chain perform, in their order.
__global__ void accum_div_kernel(float* __restrict__ S, const float* __restrict__ Sj, const float* __restrict__ Sk, long long n4, float div) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<const float4*>(S)[i];
  const float4 b = reinterpret_cast<const float4*>(Sj)[i];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  if (Sk) {  // (S + Sj) + Sk: the order of two passes, in one
    const float4 c = reinterpret_cast<const float4*>(Sk)[i];
    a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
  }
  if (div != 1.0f) { a.x = a.x / div; a.y = a.y / div; a.z = a.z / div; a.w = a.w / div; }
  reinterpret_cast<float4*>(S)[i] = a;
}

const char* launch_accum_div(float* S, const float* Sj, long long n, float div, hipStream_t s, const float* Sk) {
  if (!S || !Sj || n <= 0 || (n & 3)) return "accum_div: bad arguments";
  const long long n4 = n / 4;
  hipLaunchKernelGGL(accum_div_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, S, Sj, Sk, n4, div);
  return hipGetLastError() == hipSuccess ? nullptr : "accum_div: launch failed";
}

const char* launch_act_rows(const int32_t* lens, int32_t* out, int B, int add, int mul, long long cap, hipStream_t s, long long add_rows) {
  if (!lens || !out || B <= 0) return "act_rows: bad arguments";
  hipLaunchKernelGGL(act_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, s, lens, out, B, add, mul, cap, add_rows);
  return CHECK_LAUNCH("act_rows");
}

size_t x3_frag_bytes(int Cout, int KW, int Cin) { return (size_t)((Cout + 31) / 32) * KW * ((Cin + 31) / 32) * 4096; }

const char* launch_x3_to_frag(const float* x3, float* frag, int Cout, int KW, int Cin, hipStream_t s) {
  if (!x3 || !frag) return "x3_to_frag: null pointer";
  if (Cout <= 0 || KW <= 0 || Cin <= 0) return "x3_to_frag: bad dims";
  const int nchunk = (Cin + 31) / 32;
  const long long groups = (long long)((Cout + 31) / 32) * KW * nchunk * 256;
  hipLaunchKernelGGL(x3_to_frag_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(x3), reinterpret_cast<uint4*>(frag), Cout, KW, nchunk, groups);
  return CHECK_LAUNCH("x3_to_frag");
}

const char* launch_f32_to_frag(const float* w, float* frag, int Cout, int KW, int Cin, hipStream_t s) {
  if (!w || !frag) return "f32_to_frag: null pointer";
  if (Cout <= 0 || KW <= 0 || Cin <= 0) return "f32_to_frag: bad dims";
  const int nchunk = (Cin + 31) / 32;
  const long long groups = (long long)((Cout + 31) / 32) * KW * nchunk * 256;
  hipLaunchKernelGGL(f32_to_frag_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<float4*>(frag), Cout, KW,
                     Cin, nchunk, groups);
  return CHECK_LAUNCH("f32_to_frag");
}

const char* launch_wsola(const int16_t* x, long long n_in, int16_t* out, long long n_out, double speed, int n, int delta, int n_frames,
                         hipStream_t s) {
  if (!x || !out) return "wsola: null pointer";
  if (n_in <= 0 || n_out <= 0 || n < 64 || (n & 1) || delta < 1 || n_frames < 1 || !(speed >= 0.25 && speed <= 4.0)) return "wsola: bad arguments";
  const size_t lds = ((size_t)(n + 2 * delta) + 3 * (size_t)n + 1024) * sizeof(double) + 1024 * sizeof(int);
  if (lds > 160 * 1024) return "wsola: frame too long for LDS";
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wsola_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
  hipLaunchKernelGGL(wsola_kernel, dim3(1), dim3(1024), lds, s, x, n_in, out, n_out, speed, n, delta, n_frames);
  return CHECK_LAUNCH("wsola");
}

const char* launch_reflect_lrelu(const float* in, float* out, int B, long long n, int C, float slope, hipStream_t s) {
  if (!in || !out) return "reflect_lrelu: null pointer";
  if (C % 4 || n < 2) return "reflect_lrelu: channels must be a multiple of 4 and n >= 2";
  const long long items = (n + 1) * (C / 4);
  hipLaunchKernelGGL(reflect_lrelu_kernel, dim3((unsigned)((items + 255) / 256), B), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(in), reinterpret_cast<float4*>(out), n, C / 4, slope);
  return CHECK_LAUNCH("reflect_lrelu");
}

const char* launch_glu(const float* in, float* out, long long rows, int C, hipStream_t s) {
  if (!in || !out || rows <= 0 || C <= 0 || C % 4 || (((uintptr_t)in | (uintptr_t)out) & 15)) return "glu: bad arguments";
  const long long items = rows * (C / 4);
  hipLaunchKernelGGL(glu_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                     reinterpret_cast<float4*>(out), rows, C / 4);
  return CHECK_LAUNCH("glu");
}

const char* launch_dwconv_swish(const float* in, const float* w, const float* bias, float* out, int B, int N, int C, int k, hipStream_t s) {
  if (!in || !w || !bias || !out || in == out) return "dwconv: null or aliased pointer";
  if (B <= 0 || N <= 0 || C <= 0 || C % 4 || k <= 0 || k % 2 == 0) return "dwconv: channels must be a multiple of 4, kernel odd";
  if (((uintptr_t)in | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out) & 15) return "dwconv: buffers must be 16-byte aligned";
  const long long items = (long long)N * (C / 4);
  hipLaunchKernelGGL(dwconv_swish_kernel, dim3((unsigned)((items + 255) / 256), B), dim3(256), 0, s, reinterpret_cast<const float4*>(in),
                     reinterpret_cast<const float4*>(w), reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(out), N, C / 4, k);
  return CHECK_LAUNCH("dwconv_swish");
}

template <int K, int CT>
static const char* launch_dwglu_cfg(const float* in, const float* w, const float* bias, float* out, int B, int N, int C, hipStream_t s) {
  const size_t lds = (size_t)(64 + K - 1) * (CT + 4) * sizeof(float);
  if (lds > 64 * 1024) return "dwconv_glu: LDS tile exceeds 64 KiB";
  hipLaunchKernelGGL((dwconv_glu_swish_kernel<K, CT>), dim3((unsigned)((N + 63) / 64), (unsigned)(C / CT), (unsigned)B), dim3(256), lds, s, in, w, bias,
                     out, N, C);
  return CHECK_LAUNCH("dwconv_glu_swish");
}

// out = Swish(BN(dwconv_k(GLU(in)))): in [B, N, 2C], out [B, N, C]; scratch [B, N, C] is used only when (k, C) has no fused form
// (then: launch_glu into scratch, launch_dwconv_swish from it).  fused_out (optional) reports which path ran.
const char* launch_dwconv_glu_swish(const float* in, const float* w, const float* bias, float* out, float* scratch, int B, int N, int C, int k,
                                    hipStream_t s, bool* fused_out) {
  if (!in || !w || !bias || !out || !scratch || in == out) return "dwconv_glu: null or aliased pointer";
  if (B <= 0 || N <= 0 || C <= 0 || C % 4 || k <= 0 || k % 2 == 0) return "dwconv_glu: channels must be a multiple of 4, kernel odd";
  if (((uintptr_t)in | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)out | (uintptr_t)scratch) & 15) return "dwconv_glu: buffers must be 16-byte aligned";
  if (B > 65535) return "dwconv_glu: batch exceeds the grid limit";
  static const bool off = getenv("E2ETTS_DWGLU") && atoi(getenv("E2ETTS_DWGLU")) == 0;   // tuning aid: the two-kernel form
  if (fused_out) *fused_out = true;
  if (!off) {
    if (k == 31 && C % 128 == 0) return launch_dwglu_cfg<31, 128>(in, w, bias, out, B, N, C, s);
    if (k == 15 && C % 128 == 0) return launch_dwglu_cfg<15, 128>(in, w, bias, out, B, N, C, s);
    if (k == 7 && C % 64 == 0) return launch_dwglu_cfg<7, 64>(in, w, bias, out, B, N, C, s);
  }
  if (fused_out) *fused_out = false;
  if (const char* m = launch_glu(in, scratch, (long long)B * N, C, s)) return m;
  return launch_dwconv_swish(scratch, w, bias, out, B, N, C, k, s);
}

const char* launch_istft(const float* q, int ldq, float* specphase, float* ri, float* wav, int16_t* pcm, int B, long long F, int nfft,
                         int hop, hipStream_t s) {
  if (!q || !specphase || !ri) return "istft: null pointer";
  if (nfft < 4 || nfft > 256 || (nfft & (nfft - 1)) || hop <= 0 || nfft % hop || F < 2) return "istft: n_fft must be a power of two in [4, 256], hop must divide it";
  const int bins = nfft / 2 + 1;
  const long long total = (long long)B * F * bins;
  hipLaunchKernelGGL(istft_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, q, ldq, specphase,
                     reinterpret_cast<float2*>(ri), total, bins);
  const long long nsamp = (long long)hop * (F - 1);
  hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((nsamp + 255) / 256), B), dim3(256), 3 * nfft * sizeof(float), s,
                     reinterpret_cast<const float2*>(ri), wav, pcm, F, nfft, hop, nsamp);
  return CHECK_LAUNCH("istft");
}

const char* launch_transpose_bct_btc(const float* in, float* out, int B, int C, int T, hipStream_t s) {
  if (!in || !out) return "transpose: null pointer";
  hipLaunchKernelGGL(transpose_kernel, dim3((T + 31) / 32, (C + 31) / 32, B), dim3(256), 0, s, in, out, C, T);
  return CHECK_LAUNCH("transpose");
}

const char* launch_conv_post(const float* x, const float* w, const float* bias, float* wav, int16_t* pcm, int B,
                             long long N, int C, int KW, hipStream_t s, const int32_t* act_rows, const int32_t* act_rows_host,
                             const float* const* x_add, float x_div) {
  if (!x || !w || !bias) return "conv_post: null pointer";
  if (C % 4 || C <= 0 || C > 128 || KW <= 0 || KW > 15 || !(KW & 1)) return "conv_post: bad dims";
  constexpr int TPB = 256;
  const size_t lds = ((size_t)(TPB + KW - 1) * (C + 4) + (size_t)KW * C) * sizeof(float);
  if (lds > 64 * 1024) return "conv_post: LDS tile exceeds 64 KiB";
  dim3 grid((unsigned)((N + TPB - 1) / TPB), B);
  RowMap rm;
  if (act_rows && act_rows_host && B <= ROWMAP_MAX) {
    rm.n = B;
    rm.identity();
    rm.cum[0] = 0;
    for (int b = 0; b < B; ++b) {
      const long long r = std::min<long long>(std::max(act_rows_host[b], 0), N);
      rm.cum[b + 1] = rm.cum[b] + (int)((r + TPB - 1) / TPB);
    }
    if (rm.cum[B] == 0) return nullptr;
    grid = dim3((unsigned)rm.cum[B]);
  }
  const float* xa[3] = {x_add ? x_add[0] : nullptr, x_add ? x_add[1] : nullptr, x_add ? x_add[2] : nullptr};
  if ((xa[1] && !xa[0]) || (xa[2] && !xa[1])) return "conv_post: x_add must be filled from the front";
  hipLaunchKernelGGL(conv_post_kernel<TPB>, grid, dim3(TPB), lds, s, x, w, bias, wav, pcm, N, C, KW, act_rows, rm, xa[0], xa[1], xa[2], x_div);
  return CHECK_LAUNCH("conv_post");
}

const char* launch_var_positions(const float* x, int32_t* posbuf, const float* table, int table_rows, const float* alpha,
                                  float* y, int B, int L, int H, hipStream_t s, bool compute_positions) {
  if (!x || !posbuf || !table || !alpha || !y) return "var_positions: null pointer";
  if (L + 1 > table_rows) return "var_positions: sequence longer than the shipped position table";
  if (compute_positions) hipLaunchKernelGGL(var_positions_kernel, dim3(B), dim3(64), 0, s, x, posbuf, L, H);
  hipLaunchKernelGGL(var_pos_add_kernel, dim3(B * L), dim3(128), 0, s, x, posbuf, table, table_rows, alpha, y, H);
  return CHECK_LAUNCH("var_positions");
}

}  // namespace e2etts
