// Launch wrappers of the hand-written gfx950 kernels.  Every wrapper validates its shapes on the host
// (returning a message instead of launching) because an out-of-bounds access on this pool can reset the
// whole node; kernels themselves bounds-check every global access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace e2etts {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_LRELU = 3, ACT_SWISH = 4 /* v * sigmoid(v), Conformer FFN */ };

// Compact grid of a RAGGED batch.  A launch whose utterances need different numbers of rows must not contain workgroups that find nothing
// to do: the hardware places workgroups on CUs in a fixed rotation, not by load, so runs of workgroups that exit at once leave some
// CUs with a multiple of the others' work (tools/dispatch_rate.hip: 3072 busy workgroups with 60 % more empties among them take 1.25 x
// to 1.75 x as long as without them; a mixed-length batch lost 10 % on the 256-channel vocoder stage this way).  The launcher therefore
// counts each utterance's units (row groups, tiles, query tiles ...) on the HOST -- the engine holds the mel lengths there after its one
// synchronisation -- and the kernel finds its utterance by a scan over this table, which travels in the kernel arguments.
constexpr int ROWMAP_MAX = 64;  // utterances per launch the table holds; larger ragged batches keep the padded grid
struct RowMap {
  int n = 0;                  // 0: uniform grid (every utterance has the same number of units)
  int cum[ROWMAP_MAX + 1];    // cum[k] = units of the slots < k; cum[n] = all units
  unsigned char idx[ROWMAP_MAX];  // slot k holds utterance idx[k].  Identity for the convolutions (their units cost the same wherever they
                              // lie); the attention launcher orders the slots by length, longest first: its units cost ~ the utterance's
                              // length, and with the long ones dispatched first the short ones fill the tail of the launch
  void identity() { for (int i = 0; i < ROWMAP_MAX; ++i) idx[i] = (unsigned char)i; }
};
// (b, local unit) of unit g; false when g lies beyond the last unit (the grid is padded to a multiple of 8)
__device__ __forceinline__ bool rowmap_find(const RowMap& rm, int g, int& b, int& local) {
  if (g >= rm.cum[rm.n]) return false;
  int k = 0;
  for (int i = 1; i < rm.n; ++i) k += g >= rm.cum[i] ? 1 : 0;
  b = rm.idx[k];
  local = g - rm.cum[k];
  return true;
}

// out[b, t, n] = epilogue( sum_{j < KW} sum_{c < Cin} f(in[b, t - pad + j*dil, c]) * w[n, j*Cin + c] )
// "same" 1-D convolution as an implicit GEMM on the fp32 MFMA (M = time, N = Cout, K = KW*Cin);
// KW == 1 is a plain Linear.  Rows outside [0, T) read as zero (zero padding).
struct ConvParams {
  const float* in = nullptr;     // [B, T, Cin]  (row stride in_ld, batch stride in_bs)
  const float* w = nullptr;      // [Cout, KW*Cin] tap-major fp32; x3: [Cout, KW, ceil(Cin/32), 32 bf16 hi | 32 bf16 lo]
  int zero_tap_split = 0;        // polyphase upsampler (KW == 3): columns < split have an all-zero tap 2, columns >= split an all-zero
                                 // tap 0 (packer.polyphase_upsampler); the fragment path skips those MFMAs.  0: no structural zeros
  const float* wfrag = nullptr;  // optional: the same weights in MFMA-fragment order (launch_x3_to_frag for the bf16 modes,
                                 // launch_f32_to_frag for fp32); when set, each wave of the 128-column kernels loads its B
                                 // fragments straight from global / L2 and the weight tile skips LDS
  int x3 = 0;                    // 1: split-precision bf16x3 MFMA path (w pre-split by the packer); 2: plain bf16 (hi x hi only)
  const float* bias = nullptr;   // [Cout] or null
  const float* res = nullptr;    // optional residual, same indexing as out
  float* out = nullptr;          // [B, T, Cout]
  const int32_t* lens = nullptr; // optional [B]: output rows t >= lens[b] are written as 0
  const int32_t* act_rows = nullptr; // optional [B]: only output rows < act_rows[b] are computed at all (ragged batches)
  const int32_t* act_rows_host = nullptr;  // the same B values in HOST memory (launcher only): with them the grid holds no workgroup
                                     // without rows (RowMap above) and the tile shape is chosen on the tiles that really run
  int B = 0, T = 0, Cin = 0, Cout = 0, KW = 1, dil = 1, pad = 0;
  long long in_bs = 0, out_bs = 0, res_bs = 0;
  int in_ld = 0, out_ld = 0, res_ld = 0;
  float in_slope = 1.0f;  // leaky-ReLU slope applied to the input while staging (1 = identity)
  int act = ACT_NONE;     // applied after bias
  float act_slope = 0.0f;
  int accumulate = 0;     // out = out_old + value
  float out_div = 1.0f;   // then value / out_div
  double act_frac = 1.0;  // host-side bookkeeping only: the fraction of the B x T rows that act_rows leaves to compute (ragged
                          // batches), so that conv_gemm_flops / _bytes count the work really done
  // plain-bf16 mode of the vocoder (x3 == 2), used by the engine's router only (conv_gemm ignores them): the weights in conv_bf16.hip's
  // order, and bf16 hand-over between two launches (the input given / the result also written as the bf16 image the next layer stages)
  const void* bimg = nullptr;
  int bimg_tap_split = 0;
  int in_bf16 = 0;        // `in` points at bf16 [B, T, Cin]
  const float* in_add[3] = {nullptr, nullptr, nullptr};   // see BConvParams::in_add / in_div
  float in_div = 1.0f;
  void* out_b = nullptr;  // bf16(max(v, v * outb_slope)) [B, T, Cout]; `out` may then be null
  float outb_slope = 1.0f;
};
// returns nullptr on success, else a static error string
const char* launch_conv_gemm(const ConvParams& p, hipStream_t s);
const char* conv_gemm_class(const ConvParams& p);  // profile class = the tile configuration the launch will use
double conv_gemm_flops(const ConvParams& p);
double conv_gemm_bytes(const ConvParams& p);

// The same convolution for the phoneme-level layers (encoder, predictors), exact fp32, K split four ways inside the workgroup with a
// fixed reduction order (conv_ksplit.hip); needs p.wfrag in launch_f32_to_frag's order.  Used at every batch size for those layers.
bool conv_ksplit_supported(const ConvParams& p);
const char* launch_conv_ksplit(const ConvParams& p, hipStream_t s);

// conv_gemm's few-rows launches (small batches, the B = 1 latency path) with one wavefront per 32-row tile, private slabs, no workgroup
// barrier and a deep ring of weight fragments (conv_ksplit.hip); the same bits as conv_gemm.  Needs p.wfrag (either fragment order).
bool conv_rows_supported(const ConvParams& p);
const char* launch_conv_rows(const ConvParams& p, hipStream_t s);

// Fused masked self-attention on the packed QKV buffer of one FFT block.
// qkv [B, N, 3H] (q | k | v, head h at columns h*dk .. (h+1)*dk of each third); keys >= lens[b] masked (-inf);
// out [B, N, H]; query rows >= lens[b] are written as 0 (they are zeroed after the LayerNorm anyway).
// lens_host: the same B lengths in host memory (optional): the grid then holds only the query blocks that exist (RowMap), and the output
// rows of queries >= lens[b] beyond the last block keep their old contents instead of being zeroed.
// ws (optional, attention_workspace_bytes(B, N, H, n_head) bytes, 16-byte aligned): lets small padded fp32 grids compute the key segments
// of a query block in workgroups of their own and merge them in a second launch (same bits as the in-register merge).
const char* launch_attention(const float* qkv, float* out, const int32_t* lens, int B, int N, int H, int n_head, int x3,
                             hipStream_t s, const int32_t* lens_host = nullptr, float* ws = nullptr, size_t ws_bytes = 0);
size_t attention_workspace_bytes(int B, int N, int H, int n_head);
long long attention_par_max_grid();

// y[row, :] = LayerNorm(x[row, :]) * gamma + beta; rows t >= lens[b] -> 0 when lens != null (C <= 1024, C % 4 == 0)
const char* launch_layernorm(const float* x, float* y, const float* gamma, const float* beta, const int32_t* lens,
                             int B, int N, int C, float eps, hipStream_t s);

// x[b, l, :] = emb[ids[b, l], :] + pos[l, :]
const char* launch_embed(const int64_t* ids, const float* emb, const float* pos, float* x, int B, int L, int H,
                         int n_rows, hipStream_t s);
// x[b, l, :] = xin[b, l, :] + spk[speaker[b or 0], :]   (xin may be x)
const char* launch_add_speaker(const float* xin, float* x, const float* spk, const int64_t* speaker, int n_spk_ids, int n_speakers,
                               int B, int L, int H, hipStream_t s);
// y = x + alpha[0] * table[pos(b, l)], pos = running count of x[b, l, 0] != 0 (0 where it is 0)
// (posbuf: [B, L] int32 scratch; compute_positions = false reuses what an earlier call on the same x left there)
const char* launch_var_positions(const float* x, int32_t* posbuf, const float* table, int table_rows, const float* alpha,
                                 float* y, int B, int L, int H, hipStream_t s, bool compute_positions = true);
// out[row, o] = dot(x[row, :], w[o, :]) + b[o], o < O <= 2; rows >= lens[b] -> 0 when lens != null
const char* launch_rowdot(const float* x, const float* w, const float* b, float* out, const int32_t* lens, int B, int L,
                          int C, int O, hipStream_t s);
// duration_rounded, integer durations, inclusive scan -> mel_lens; one block per utterance
const char* launch_duration(const float* log_d, float d_control, float* dur, int32_t* cum, int64_t* mel_lens64,
                            int32_t* mel_lens32, int B, int L, hipStream_t s);
// pitch / energy bucket indices + x += pitch_emb[pidx] + energy_emb[eidx]
// pitch_mode 0: f0 = pred[0] * std + mean, zeroed where pred[1] > 0 (uv), tensor_f0_to_coarse; 1: the same with f0 = 2 ** pred[0]
// (pitch_quantization "log"); 2: use_uv False -- ONE prediction per row, bucketize(pred * p_control, pitch_bins) (U/layers.py:136-160)
const char* launch_variance_embed(float* x, float* pitch_pred /*[B,L,2], scaled in place by p_control; mode 2: [B,L], left as it is*/,
                                  const float* energy_pred, float p_control, float e_control, float f0_mean,
                                  float f0_std, const float* energy_bins, int n_bins, const float* pitch_emb,
                                  const float* energy_emb, int32_t* pitch_idx, int32_t* energy_idx, int B, int L, int H,
                                  hipStream_t s, int pitch_mode = 0, const float* pitch_bins = nullptr, int feat = 3 /* bit 0 pitch, bit 1 energy */);
// y[b, t, :] += pos[t, :]
const char* launch_add_positions(float* y, const float* pos, int B, int T, int H, hipStream_t s);
// length regulator fused with the decoder position add: y[b, t, :] = (t < mel_len[b] ? x[b, ph(t), :] : 0) + pos[t, :] (pos == nullptr: without it)
const char* launch_length_regulate(const float* x, const int32_t* cum, const int32_t* mel_lens, const float* pos,
                                   float* y, int B, int L, int T, int H, hipStream_t s);
// out[b] = min(cap, (lens[b] + add) * mul + add_rows): rows a layer has to compute for utterance b in ragged mode
const char* launch_act_rows(const int32_t* lens, int32_t* out, int B, int add, int mul, long long cap, hipStream_t s, long long add_rows = 0);
// S = (S + Sj) [+ Sk] [/ div] over n floats (n % 4 == 0): joins the sums of ResBlocks run on side streams
const char* launch_accum_div(float* S, const float* Sj, long long n, float div, hipStream_t s, const float* Sk = nullptr);
// [B, C, T] -> [B, T, C]
// One (conv k, dilation d -> leaky ReLU -> conv k, dilation 1 -> + x) pair of a HiFi-GAN ResBlock1 in one launch
// (resblock_pair.hip): out = c2(lrelu(c1(lrelu(x)) + b1)) + b2 + x, optionally (out_old + that) / out_div.
struct PairParams {
  const float* x = nullptr;      // [B, T, C] channels-last; also the residual
  const float* wfrag = nullptr;  // conv1's fragment-order image followed by conv2's (C x KW x C each): launch_x3_to_frag's order for
                                 // mode 1 / 2, launch_f32_to_frag's for mode 0
  const float* b1 = nullptr;     // [C]
  const float* b2 = nullptr;     // [C]
  float* out = nullptr;          // [B, T, C], must not alias x
  const int32_t* act_rows = nullptr;  // optional [B]: only output rows < act_rows[b] are needed (whole tiles beyond are skipped)
  const int32_t* act_rows_host = nullptr;  // the same values in host memory (launcher only): compact grid, see RowMap
  int B = 0, T = 0, C = 0, KW = 0, dil = 1;
  long long x_bs = 0, out_bs = 0;     // batch strides in floats
  float slope = 0.1f;            // leaky-ReLU slope of both activations
  int accumulate = 0;            // out = out_old + result
  float out_div = 1.0f;          // then / out_div (needs accumulate)
  int mode = 1;                  // 0: exact fp32, 1: bf16x3 split precision, 2: plain bf16
  double act_frac = 1.0;         // host-side bookkeeping only (see ConvParams::act_frac)
  const void* bimg1 = nullptr;   // mode 2, optional: conv1's and conv2's weights in conv_bf16.hip's order (launch_bf16_image): the pair then
  const void* bimg2 = nullptr;   // may run on launch_pair_bf16 (same bits as launch_resblock_pair in mode 2)
};
bool resblock_pair_supported(int C, int KW, int dil);
// the same pair in plain bf16 on conv_bf16.hip's machinery (whole-slab staging, ring-prefetched weights, 128 x 32 wavefront tiles);
// padded batches (act_rows == null), C = 32 / 64 / 128
bool pair_bf16_supported(const PairParams& p);
const char* launch_pair_bf16(const PairParams& p, hipStream_t s);
const char* launch_pair_bf16_group(const PairParams* p, int n, hipStream_t s);   // as launch_conv_bf16_group: same B, T, C; KW / dil / buffers per member

// A WHOLE ResBlock1 (reference V/layers.py:33-40: n_pairs x [lrelu -> conv k, dilation d_m -> lrelu -> conv k -> + x]) in one launch, plain
// bf16, any odd kernel size, 32 or 64 channels: resblock_chain.hip's scheme (residual stream in registers, two operand images in LDS, every
// pair computed on all R = 512 positions of a tile and the sum_m (k - 1) / 2 (d_m + 1) positions per edge that saw a neighbour's rows
// discarded) on conv_bf16.hip's machinery.  HBM / Infinity-Cache traffic per ResBlock: x in, out out (+ out in) -- against three passes per
// PAIR as pair launches, which is what bounds the 32- and 64-channel stages (tensors of 35 MB per 542-frame window at 48 kHz, far beyond
// an XCD's 4 MB of L2).  Same arithmetic in the same order as launch_pair_bf16 / launch_resblock_pair in mode 2: same bits.
constexpr int RB_MAX_PAIRS = 4;
struct RbParams {
  const float* x = nullptr;      // [B, T, C] channels-last
  float* out = nullptr;          // [B, T, C], must not alias x
  const void* bimg[RB_MAX_PAIRS][2] = {};   // conv1 / conv2 weights of pair m in launch_bf16_image's order
  const float* b1[RB_MAX_PAIRS] = {};
  const float* b2[RB_MAX_PAIRS] = {};
  int dil[RB_MAX_PAIRS] = {1, 1, 1, 1};
  int n_pairs = 3;
  int B = 0, T = 0, C = 0, KW = 3;
  long long x_bs = 0, out_bs = 0;
  float slope = 0.1f;
  int accumulate = 0;            // out = out_old + result
  float out_div = 1.0f;          // then / out_div (needs accumulate)
};
bool rb_bf16_supported(const RbParams& p);
const char* launch_rb_bf16_group(const RbParams* p, int n, hipStream_t s);   // members share B, T, C, n_pairs; KW / dilations / buffers per member
inline const char* launch_rb_bf16(const RbParams& p, hipStream_t s) { return launch_rb_bf16_group(&p, 1, s); }
// ALL the ResBlocks of a stage on the same input (reference V/generator.py:44-48) in one launch, each workgroup computing every member on
// its tile from ONE load of x and writing out = (((rb_0 + rb_1) + rb_2) ...) / n to p[0].out: no partial sums in memory, no join (32 channels)
bool rb_bf16_stage_supported(const RbParams* p, int n);
const char* launch_rb_bf16_stage(const RbParams* p, int n, hipStream_t s);
double rb_bf16_flops(const RbParams& p);
double rb_bf16_bytes(const RbParams& p);
const char* launch_resblock_pair(const PairParams& p, hipStream_t s);
double resblock_pair_flops(const PairParams& p);
double resblock_pair_bytes(const PairParams& p);

// A whole ResBlock1 of kernel size 3 (three pairs, dilations dil[0..2]) in one launch (resblock_chain.hip):
// out = x_3 with x_{m+1} = c2_m(lrelu(c1_m(lrelu(x_m)) + b1_m)) + b2_m + x_m, optionally (out_old + x_3) / out_div.
struct ChainParams {
  const float* x = nullptr;        // [B, T, C] channels-last
  const float* wfrag = nullptr;    // fragment-order images, contiguous: conv1 | conv2 of pair 0, then pair 1, then pair 2
  const float* b1[3] = {nullptr, nullptr, nullptr};
  const float* b2[3] = {nullptr, nullptr, nullptr};
  float* out = nullptr;            // [B, T, C], must not alias x
  const int32_t* act_rows = nullptr;
  const int32_t* act_rows_host = nullptr;  // as in PairParams
  int B = 0, T = 0, C = 0, KW = 3;
  int dil[3] = {1, 3, 5};
  long long x_bs = 0, out_bs = 0;
  float slope = 0.1f;
  int accumulate = 0;
  float out_div = 1.0f;
  int mode = 1;                    // 1: bf16x3 split precision, 2: plain bf16
  double act_frac = 1.0;           // host-side bookkeeping only (see ConvParams::act_frac)
};
bool resblock_chain_supported(int C, int KW, const int* dil, int n_dil);
const char* launch_resblock_chain(const ChainParams& p, hipStream_t s);
double resblock_chain_flops(const ChainParams& p);
double resblock_chain_bytes(const ChainParams& p);

// ---- conv_bf16.hip: the convolutions of the vocoder in plain bf16 (precision "bf16", BASELINE config 5: the 48 kHz long-form stream).
// out[b, t, n] = epilogue( sum_c sum_j bf16(f(in[b, t - pad + j dil, c])) * bf16(w[n, j, c]) ), fp32 accumulation on
// v_mfma_f32_32x32x16_bf16 in conv_gemm's order of terms (chunk-major, tap, k-step), so the result is BIT-IDENTICAL to conv_gemm /
// conv_rows / the fused ResBlock kernels in mode 2 and the engine may pick per launch.  What differs is the shape of the work: the whole
// [rows + halo] x Cin slab of a row tile sits in LDS as bf16 (converted ONCE while staging, or copied when the producer already wrote
// bf16), so there is one workgroup barrier per tile instead of two per 32-channel chunk, and the weights stream from L2 through a ring of
// D (chunk, tap) units per wavefront.  Made for what mode 2's shapes need: few rows with a long K (the 256-channel stage of a 542-frame
// streaming window: 4 336 rows, K up to 2 816) and MFMA : fragment ratios that the L1's 64 B / clk can feed with a third of bf16x3's
// MFMAs per fragment.
struct BConvParams {
  const void* in = nullptr;       // [B, T, Cin] channels-last, dense: fp32 (in_bf16 = 0) or bf16 (in_bf16 = 1)
  int in_bf16 = 0;
  float in_slope = 1.0f;          // fp32 input: leaky ReLU applied while staging (1 = identity); a bf16 input is taken as it is
  const float* in_add[3] = {nullptr, nullptr, nullptr};   // fp32 input: further tensors of the same shape, summed while staging in this order --
  float in_div = 1.0f;            // x = (((in + in_add[0]) + in_add[1]) + in_add[2]) / in_div: the join of a stage's parallel ResBlocks
                                  // (reference V/generator.py:44-48: xs / num_kernels) folded into the next layer's input
  const void* wimg = nullptr;     // launch_bf16_image's order
  int KWe = 0;                    // taps the image holds per 32-column tile: KW, or 2 for a polyphase upsampler (tap_split != 0)
  int tap_split = 0;              // polyphase upsampler (KW == 3, packer.polyphase_upsampler): columns < tap_split never use tap 2, columns
                                  // >= tap_split never tap 0; the image holds the two live taps of each 32-column tile.  0: all taps
  const float* bias = nullptr;    // [Cout] or null
  float act_slope = 1.0f;         // v = max(v, v * act_slope) after the bias (1 = none, 0 = ReLU)
  const float* res = nullptr;     // optional fp32 residual [B, T, Cout], added after the activation
  int accumulate = 0;             // v = out_old + v   (needs out)
  float out_div = 1.0f;           // then v / out_div  (needs accumulate)
  float* out = nullptr;           // fp32 result [B, T, Cout] or null
  void* out_b = nullptr;          // bf16 image of the result for the NEXT convolution, or null: bf16(max(v, v * outb_slope))
  float outb_slope = 1.0f;
  int B = 0, T = 0, Cin = 0, Cout = 0, KW = 1, dil = 1, pad = 0;
  int rows_hint = 0;              // 0: tile shape by B x T; tuning aid otherwise (bench)
};
bool conv_bf16_supported(const BConvParams& p);
const char* launch_conv_bf16(const BConvParams& p, hipStream_t s);
// Up to BC_GROUP_MAX INDEPENDENT convolutions of the same geometry class (B, T, Cin, Cout, input type; kernel size, dilation, weights and
// buffers may differ) in ONE launch: the parallel ResBlocks of a vocoder stage (reference V/generator.py:44-48) at small windows, where a
// single convolution cannot fill the chip and three streams of short launches are bound by the host's enqueue rate.  Members are
// dispatched in the order given (longest first is best).  Each member's result is what launch_conv_bf16 gives for it alone.
constexpr int BC_GROUP_MAX = 4;
const char* launch_conv_bf16_group(const BConvParams* p, int n, hipStream_t s);
const char* conv_bf16_class(const BConvParams& p);
// split-precision weight image [Cout][KW][ceil(Cin/32)][32 bf16 hi | 32 bf16 lo] (packer.pack_x3) -> the hi halves alone in the order
// conv_bf16 streams them: [Cout/32][chunk][tap slot 0..KWe-1][k-step 0..1][lane 0..63][8 bf16] (2 KiB per (tile, chunk, tap), units of a
// tile contiguous).  tap_split != 0 (KW == 3): tap slot s of tile t is tap s + (32 t >= tap_split ? 1 : 0), KWe = 2.
size_t bf16_image_bytes(int Cout, int KW, int Cin, int tap_split);
const char* launch_bf16_image(const float* x3, void* img, int Cout, int KW, int Cin, int tap_split, hipStream_t s);

// split-precision weight image -> MFMA-fragment order (ConvParams::wfrag)
const char* launch_x3_to_frag(const float* x3, float* frag, int Cout, int KW, int Cin, hipStream_t s);
size_t x3_frag_bytes(int Cout, int KW, int Cin);
// fp32 weights [Cout][KW * Cin] -> the fp32 fragment order (same size as the bf16 one: 4 KiB per (32 columns, tap, 32 channels))
const char* launch_f32_to_frag(const float* w, float* frag, int Cout, int KW, int Cin, hipStream_t s);

// iSTFTNet tail: leaky ReLU + ReflectionPad1d((1, 0)) on channels-last frames; exp / sin heads, inverse STFT with overlap-add
const char* launch_reflect_lrelu(const float* in, float* out, int B, long long n, int C, float slope, hipStream_t s);
const char* launch_istft(const float* q, int ldq, float* specphase, float* ri, float* wav, int16_t* pcm, int B, long long F, int nfft,
                         int hop, hipStream_t s);

// Conformer block pieces (reference U/blocks/conformer.py:273-304, 443-481; U/blocks/utils.py:196-219)
const char* launch_glu(const float* in, float* out, long long rows, int C, hipStream_t s);          // [rows, 2C] -> [rows, C]
// depthwise conv (k taps, zero "same" padding over [0, N), w [k][C], BatchNorm folded into w / bias) + Swish; channels-last
const char* launch_dwconv_swish(const float* in, const float* w, const float* bias, float* out, int B, int N, int C, int k, hipStream_t s);
// The two above in one pass: out = Swish(BN(dwconv_k(GLU(in)))), in [B, N, 2C] -> out [B, N, C]; same bits as the two-kernel form, which
// it falls back to (through `scratch`, [B, N, C]) for kernel sizes / channel counts without a fused instantiation
const char* launch_dwconv_glu_swish(const float* in, const float* w, const float* bias, float* out, float* scratch, int B, int N, int C, int k,
                                    hipStream_t s, bool* fused_out = nullptr);
// Relative-position self-attention of RelativeMultiHeadAttention (conformer.py:399-440): no key mask (nn.Sequential passes no mask,
// :252), score = ((q + u) . k + shift((q + v) . P)) / sqrt(H), where _relative_shift (:432-440) re-indexes the position scores as
// (i, j <= i) -> (q_i + v) . P[N - 1 - i + j]; (i, i + 1) -> 0; (i, j > i + 1) -> (q_{i+1} + v) . P[j - i - 2].  The kernel computes those
// entries itself from the projected position table (no [B, heads, N, N] tensor).  qkv [B, N, 3H] as in launch_attention; u, v [H] (heads
// flattened); pos [n_head][pos_rows][H / n_head] = pos_proj(table) per head (pos_rows >= N); out [B, N, H].
// pos_x3 (optional): the same table as bf16 hi | lo halves per row (packer: `att.pos.x3`) -> the split-precision kernel (decoder, bf16x3 mode).
const char* launch_rel_attention(const float* qkv, const float* pos, int pos_rows, const float* u, const float* v, float* out, int B, int N,
                                 int H, int n_head, hipStream_t s, const float* pos_x3 = nullptr);

// tempo change without pitch change (WSOLA; small_kernels.hip): x int16 [n_in] -> out int16 [n_out <= n_frames * n / 2 + n / 2]
const char* launch_wsola(const int16_t* x, long long n_in, int16_t* out, long long n_out, double speed, int n, int delta, int n_frames,
                         hipStream_t s);

const char* launch_transpose_bct_btc(const float* in, float* out, int B, int C, int T, hipStream_t s);
// wav = tanh(conv7(lrelu_0.01(x))) with Cout = 1; pcm = (int16)(int32)(wav * 32768)
// act_rows / act_rows_host (optional, device / host copies of the same B values): only samples < act_rows[b] are written (ragged batches)
// x_add (optional, up to 3 more tensors like x) / x_div: x = (((x + x_add[0]) + x_add[1]) + x_add[2]) / x_div, formed while staging (the
// join of the last stage's parallel ResBlocks, see BConvParams::in_add)
const char* launch_conv_post(const float* x, const float* w, const float* bias, float* wav, int16_t* pcm, int B,
                             long long N, int C, int KW, hipStream_t s, const int32_t* act_rows = nullptr, const int32_t* act_rows_host = nullptr,
                             const float* const* x_add = nullptr, float x_div = 1.0f);

}  // namespace e2etts
