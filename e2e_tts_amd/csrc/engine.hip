// Engine + C ABI (include/e2etts.h): owns the weights, the workspace, one HIP stream, and sequences the
// kernels of kernels.h into the reference's inference path
//   UnsupervisedFastSpeech2.inference (reference U/model.py:155-194)  ->  HifiGan.forward (V/generator.py:37-53)
// exactly as TTS.inference drives them per batch (API/utils.py:130-148).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <dlfcn.h>

#include "../../include/e2etts.h"
#include "host_logic.h"
#ifndef E2ETTS_ST_DEPTH
#define E2ETTS_ST_DEPTH 2   // chunks the streaming vocoder keeps in flight (tuning builds: 3)
#endif
#include "kernels.h"

using namespace e2etts;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct FFTLayer {
  const float *wqkv, *bqkv, *wo, *bo, *ln1g, *ln1b, *w1, *b1, *w2, *b2, *ln2g, *ln2b;
  const float *wqkv_x3 = nullptr, *wo_x3 = nullptr, *w1_x3 = nullptr, *w2_x3 = nullptr;  // optional split-precision images
};
// Conformer block (U/blocks/conformer.py:171-255), as packed by packer._pack_conformer
struct CfGemm {
  const float *w = nullptr, *b = nullptr, *wx3 = nullptr;
};
struct CfLayer {
  const float *ff_lng[2], *ff_lnb[2];
  CfGemm ff_a[2], ff_b[2];  // FeedForwardModule x 2: Linear(H -> F), Linear(F -> H) (half-step factor folded in)
  const float *att_lng, *att_lnb, *att_u, *att_v;
  CfGemm att_qkv, att_o;
  const float* pos[2];  // pos_proj(table) per head [n_head][rows][d_head]: stored / regenerated table
  const float* posx[2] = {nullptr, nullptr};  // the same as bf16 hi | lo rows (decoder, optional): the split-precision attention kernel's operand
  uint64_t pos_rows[2];
  const float *cv_lng, *cv_lnb, *dw_w, *dw_b;
  CfGemm pw1, pw2;
  const float *lng, *lnb;
};
struct PredLayer {
  const float *w, *b, *g, *beta;
};
struct Predictor {
  std::vector<PredLayer> layers;
  const float *lin_w = nullptr, *lin_b = nullptr, *alpha = nullptr;
  int kernel = 0, chans = 0, odim = 0;
};
struct ConvW {
  const float *w = nullptr, *b = nullptr;
  const float* wx3 = nullptr;  // split-precision (bf16 hi | lo) image, when the blob carries one
};

struct ProfRec {
  hipEvent_t start, stop;
  int cls;
  double flops, bytes;
};

}  // namespace

struct e2etts_engine {
  e2etts_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::string err;
  size_t dev_bytes = 0;

  std::vector<DevBuf*> owned;  // every buffer ensure() has ever allocated (members of this struct); e2etts_destroy frees them

  // weights
  DevBuf blob;
  std::map<std::string, std::pair<const float*, uint64_t>> tensors;
  // split-precision image -> the same weights in MFMA-fragment order (built on the device at load time for the layers
  // the 128-column kernel serves: their waves read weight fragments straight from L2, no LDS tile, no barrier per tap)
  std::map<const float*, float*> frag_of;
  // split-precision image -> its hi halves in conv_bf16.hip's order (launch_bf16_image), for the vocoder's plain-bf16 mode; value: image
  // and the tap_split it was built with (polyphase upsamplers keep two taps per 32-column tile)
  std::map<const float*, std::pair<void*, int>> bimg_of;
  size_t frag_bytes = 0;
  bool ac_loaded = false, voc_loaded = false;
  std::vector<FFTLayer> enc, dec;
  std::vector<CfLayer> cf_enc, cf_dec;  // cfg.block_type == 1
  Predictor dur, pitch, energy;
  const float *emb = nullptr, *enc_pos = nullptr, *dec_pos = nullptr, *pos_regen = nullptr, *spk_emb = nullptr;
  const float *var_pos = nullptr, *pitch_emb = nullptr, *energy_emb = nullptr, *energy_bins = nullptr, *pitch_bins = nullptr;
  uint64_t var_pos_rows = 0;
  ConvW mel_lin, voc_pre, voc_post;
  std::vector<ConvW> postnet, voc_up;
  std::vector<std::vector<ConvW>> rb_c1, rb_c2;  // [stage * n_kernels + j][dilation index]
  // fused ResBlock pairs (resblock_pair.hip): conv1's fragment-order image followed by conv2's, per [stage * n_kernels + j][m];
  // stage_fused[i]: every pair of stage i can run fused (channels 32 / 64 / 128)
  std::vector<std::vector<float*>> rb_pair_frag;
  std::vector<float*> rb_frag_base;  // the allocations rb_pair_frag points into (one per ResBlock)
  // the same for the exact-fp32 mode (launch_f32_to_frag order), built for the stages of <= 64 channels only: wider stages are bound by
  // the fp32 matrix pipe, where the fused kernel's (KW - 1) / BMI recompute costs more than its saved HBM traffic buys
  std::vector<std::vector<float*>> rb_pair_frag32;
  std::vector<char> stage_fused;
  int fuse_pairs = 2;  // e2etts_set_fused_resblocks: 0 off, 1 pairs, 2 pairs + whole k = 3 ResBlocks

  // workspace
  DevBuf ids, lens64, lens32, spk, xa, xb, xs, xp, tmp, qkv, att, hid, p1, p2, attws;
  DevBuf logd, durf, cum, mel64, mel32, posbuf, ppred, epred, pidx, eidx;
  DevBuf dx, dxb, mel, melpost, pn1, pn2;
  DevBuf melin, v0, v1, v2, v3, wav, pcm;
  // small batches: the ResBlocks of a vocoder stage run side by side, ResBlock j > 0 on side stream j - 1 with buffers of its own
  // (stage sum, two ping-pong buffers); created on first use
  DevBuf vside[E2ETTS_MAX_RB_KERNELS - 1][3];
  hipStream_t side[E2ETTS_MAX_RB_KERNELS - 1] = {};
  hipEvent_t ev_fork = nullptr, ev_join[E2ETTS_MAX_RB_KERNELS - 1] = {};
  DevBuf tempo_in, tempo_out;  // e2etts_tempo
  DevBuf istft_q, istft_ri, istft_sp;  // iSTFTNet tail: conv_post output, Re/Im per bin, exp / sin heads (tap "istft_spec_phase")
  int istft_B = 0;
  long long istft_F = 0;
  int64_t* h_mel = nullptr;  // pinned
  int last_B = 0, last_L = 0, last_T = 0, voc_B = 0, voc_T = 0;
  bool have_acoustic = false, have_wav = false;
  int voc_precision = 0;  // 0: fp32 MFMA (default: the reference's arithmetic), 1: bf16x3 split-precision MFMA, 2: plain bf16 (E2ETTS_PRECISION_*)
  // streaming vocoder (e2etts_vocoder_stream_*): trailing mel frames kept as context / not yet emitted
  DevBuf st_carry, st_win;
  int st_B = 0, st_carry_n = 0, st_halo = 0;
  long long st_emitted = 0;
  bool st_open = false, st_done = false;
  // Two chunks may be in flight: _push enqueues and returns, _fetch takes the OLDEST unfetched chunk.  A chunk's pass depends on the mel
  // stream only (the carried context is mel frames), so the two passes are independent: each slot has a compute stream, a workspace and
  // output buffers of its own, and the kernels of chunk i + 1 fill the CUs that chunk i's launch tails and small launches leave idle.
  // Window assembly, uploads and downloads run on the engine's own stream, which carries no kernels while a stream is open (and is the
  // one e2etts_order_after orders behind the caller's work).  That makes three streams busy at a time: a process has FOUR hardware
  // queues by default (GPU_MAX_HW_QUEUES), the null stream holds one, and streams beyond that share a queue, i.e. run in turn --
  // measured: with a copy stream of its own the two passes landed on one queue and did not overlap at all.
  static constexpr int ST_DEPTH = E2ETTS_ST_DEPTH;
  struct VocCtx {   // what one vocoder pass works in besides the weights: swapped into the engine around vocoder_impl
    hipStream_t stream = nullptr;
    DevBuf v0, v1, v2, v3, vside[E2ETTS_MAX_RB_KERNELS - 1][3];
    hipStream_t side[E2ETTS_MAX_RB_KERNELS - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[E2ETTS_MAX_RB_KERNELS - 1] = {};
  };
  struct StSlot {
    VocCtx ctx;
    DevBuf wav, pcm, win;
    void* pin = nullptr;
    size_t pin_cap = 0;
    hipEvent_t done = nullptr, win_ready = nullptr;
    int emit_n = 0, emit_off = 0, win_n = 0;
  } st_slot[ST_DEPTH];
  int st_head = 0, st_pending = 0;
  int ragged = 1;         // synthesize(): skip rows of shorter utterances that no valid output sample depends on
  bool rag_short = false; // the last acoustic pass had an utterance shorter than T (else ragged mode has nothing to skip)
  DevBuf actbuf;          // [5 + voc_stages][B] int32 row limits: decoder, mel_linear / postnet, vocoder stage 0 .. voc_stages, encoder (unused on the
                          // device: lens32 serves), variance predictors
  std::vector<int32_t> h_act;  // the same limits on the host, same layout (valid for the call that computed them): the launchers build
                               // their compact grids from them (kernels.h: RowMap)
  double rag_frac_dec = 1.0, rag_frac_post = 1.0, rag_frac_voc = 1.0;  // fraction of the padded rows those limits leave (profile FLOP counts)
  int dec_precision = 0;  // same choice for decoder + mel_linear + postnet (encoder / variance adaptor: always fp32)

  // profiling
  bool prof_on = false;
  std::string prof_filter;  // non-empty: only launches of this class are bracketed by events (e2etts_profile_filter)
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> ev_pool;
  std::vector<e2etts_kernel_stat> prof_stats;

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

namespace {

#define HIPCHK(e, expr)                                                                     \
  do {                                                                                      \
    hipError_t _s = (expr);                                                                 \
    if (_s != hipSuccess) return (e)->fail(E2ETTS_EHIP, "%s: %s", #expr, hipGetErrorString(_s)); \
  } while (0)

#define RET(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != E2ETTS_OK) return _r; \
  } while (0)

int ensure(e2etts_engine* e, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return E2ETTS_OK;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (b.p) {
    HIPCHK(e, hipFree(b.p));
    e->dev_bytes -= b.cap;
    b.p = nullptr;
    b.cap = 0;
  }
  if (std::find(e->owned.begin(), e->owned.end(), &b) == e->owned.end()) e->owned.push_back(&b);
  size_t want = (bytes + 255) & ~size_t(255);
  if (hipMalloc(&b.p, want) != hipSuccess) {
    b.p = nullptr;
    return e->fail(E2ETTS_ENOMEM, "hipMalloc(%zu bytes) failed", want);
  }
  b.cap = want;
  e->dev_bytes += want;
  // ragged mode leaves rows nobody needs uncomputed: make sure what they hold is at least finite
  HIPCHK(e, hipMemsetAsync(b.p, 0, want, e->stream));
  return E2ETTS_OK;
}

template <typename T>
T* ptr(DevBuf& b) { return reinterpret_cast<T*>(b.p); }

int prof_class(e2etts_engine* e, const char* name) {
  for (size_t i = 0; i < e->prof_stats.size(); ++i)
    if (!strcmp(e->prof_stats[i].name, name)) return (int)i;
  e2etts_kernel_stat s{};
  snprintf(s.name, sizeof s.name, "%s", name);
  e->prof_stats.push_back(s);
  return (int)e->prof_stats.size() - 1;
}

hipEvent_t get_event(e2etts_engine* e) {
  if (!e->ev_pool.empty()) {
    hipEvent_t ev = e->ev_pool.back();
    e->ev_pool.pop_back();
    return ev;
  }
  hipEvent_t ev = nullptr;
  (void)hipEventCreate(&ev);
  return ev;
}

struct ProfScope {
  e2etts_engine* e;
  ProfRec rec{};
  bool on;
  ProfScope(e2etts_engine* e_, const char* name, double flops, double bytes) : e(e_), on(e_->prof_on) {
    if (on && !e->prof_filter.empty() && e->prof_filter != name) on = false;
    if (!on) return;
    rec.cls = prof_class(e, name);
    rec.flops = flops;
    rec.bytes = bytes;
    rec.start = get_event(e);
    rec.stop = get_event(e);
    (void)hipEventRecord(rec.start, e->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(rec.stop, e->stream);
    e->prof_recs.push_back(rec);
  }
};

int prof_collect(e2etts_engine* e) {
  if (e->prof_recs.empty()) return E2ETTS_OK;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (auto& r : e->prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      auto& s = e->prof_stats[r.cls];
      s.launches += 1;
      s.ms += ms;
      s.flops += r.flops;
      s.bytes += r.bytes;
    }
    e->ev_pool.push_back(r.start);
    e->ev_pool.push_back(r.stop);
  }
  e->prof_recs.clear();
  return E2ETTS_OK;
}

#define KCHK(e, expr)                                              \
  do {                                                             \
    const char* _m = (expr);                                       \
    if (_m) return (e)->fail(E2ETTS_EINVAL, "%s", _m);             \
  } while (0)

// The plain-bf16 mode's convolutions on conv_bf16.hip (bit-identical to conv_gemm's mode 2, so the choice is per launch): a padded
// batch, dense rows, an activation that is a slope.  E2ETTS_BCONV=0 (tuning aid) keeps them on conv_gemm.
bool bconv_params(const ConvParams& p, BConvParams& q) {
  static const bool on = !(getenv("E2ETTS_BCONV") && atoi(getenv("E2ETTS_BCONV")) == 0);
  if (!on || p.x3 != 2 || !p.bimg || p.act_rows || p.lens) return false;
  if (p.act != ACT_NONE && p.act != ACT_RELU && p.act != ACT_LRELU) return false;
  if (p.in_ld != p.Cin || p.out_ld != p.Cout || (p.res && p.res_ld != p.Cout)) return false;
  if (p.in_bs != (long long)p.T * p.Cin || p.out_bs != (long long)p.T * p.Cout || (p.res && p.res_bs != p.out_bs)) return false;
  if (p.bimg_tap_split != p.zero_tap_split) return false;
  q = BConvParams();
  q.in = p.in; q.in_bf16 = p.in_bf16; q.in_slope = p.in_slope; q.wimg = p.bimg; q.tap_split = p.bimg_tap_split;
  for (int k = 0; k < 3; ++k) q.in_add[k] = p.in_add[k];
  q.in_div = p.in_div;
  q.KWe = p.bimg_tap_split > 0 ? 2 : p.KW;
  q.bias = p.bias; q.act_slope = p.act == ACT_LRELU ? p.act_slope : (p.act == ACT_RELU ? 0.f : 1.f);
  q.res = p.res; q.accumulate = p.accumulate; q.out_div = p.out_div; q.out = p.out; q.out_b = p.out_b; q.outb_slope = p.outb_slope;
  q.B = p.B; q.T = p.T; q.Cin = p.Cin; q.Cout = p.Cout; q.KW = p.KW; q.dil = p.dil; q.pad = p.pad;
  return conv_bf16_supported(q);
}

// One conv / linear launch.  alg_scale < 1 when part of the packed weight is structural zeros
// (the polyphase upsampler) so that the recorded FLOPs stay the algorithmic ones.
// ksplit: a phoneme-level layer (encoder FFT blocks, predictors): served by conv_ksplit.hip at every batch size when its fp32 fragment
// image exists (E2ETTS_KSPLIT=0 routes them through conv_gemm again: tuning aid, changes the summation order of those layers)
int conv(e2etts_engine* e, ConvParams p, double alg_scale = 1.0, bool ksplit = false) {
  if (p.in_ld == 0) p.in_ld = p.Cin;
  if (p.out_ld == 0) p.out_ld = p.Cout;
  if (p.res && p.res_ld == 0) p.res_ld = p.Cout;
  if (p.in_bs == 0) p.in_bs = (long long)p.T * p.in_ld;
  if (p.out_bs == 0) p.out_bs = (long long)p.T * p.out_ld;
  if (p.res && p.res_bs == 0) p.res_bs = (long long)p.T * p.res_ld;
  if (!p.wfrag) {  // fragment-order image of these weights (split-precision image or fp32 tensor: the map is keyed by pointer)
    auto it = e->frag_of.find(p.w);
    if (it != e->frag_of.end()) p.wfrag = it->second;
  }
  // E2ETTS_PROFILE_FINE=1: one profile class per layer shape instead of per kernel configuration (a tuning aid)
  static const bool fine = getenv("E2ETTS_PROFILE_FINE") != nullptr;
  char fname[48];
  if (fine && e->prof_on)
    snprintf(fname, sizeof fname, "%s %d>%d k%d d%d r%lld%s%s", p.x3 ? "x3" : "f32", p.Cin, p.Cout, p.KW, p.dil,
             (long long)p.B * p.T, p.res ? "+r" : "", p.accumulate ? "+a" : "");
  {
    BConvParams q;
    if (bconv_params(p, q)) {
      ProfScope ps(e, fine && e->prof_on ? fname : conv_bf16_class(q), conv_gemm_flops(p) * alg_scale, conv_gemm_bytes(p));
      KCHK(e, launch_conv_bf16(q, e->stream));
      return E2ETTS_OK;
    }
    if (p.in_bf16 || p.out_b || p.in_add[0]) return e->fail(E2ETTS_EINVAL, "bf16 hand-over / joined input asked of a launch conv_bf16 does not serve");
  }
  static const bool ksplit_on = !(getenv("E2ETTS_KSPLIT") && atoi(getenv("E2ETTS_KSPLIT")) == 0);
  // ... and only where the chain is long (K = KW x Cin >= 768: the FFN convolutions, the predictors): a q | k | v or fc projection
  // (K = 384) is a chain of 12 short links, and at B = 32 the split costs those more (three idle waves per epilogue) than it saves.
  // The choice is by layer shape, never by batch size, so every batch size sums every layer in the same order.
  if (ksplit && ksplit_on && (long long)((p.Cin + 31) / 32) * p.KW >= 24 && conv_ksplit_supported(p)) {
    ProfScope ps(e, fine && e->prof_on ? fname : "conv_ksplit", conv_gemm_flops(p) * alg_scale, conv_gemm_bytes(p));
    KCHK(e, launch_conv_ksplit(p, e->stream));
    return E2ETTS_OK;
  }
  // few rows (small batches, the B = 1 latency path) on fragment-order weights: conv_rows -- conv_gemm's arithmetic, MFMA for MFMA, with
  // one wavefront per 32-row tile and no workgroup barrier (conv_ksplit.hip).  Same bits: E2ETTS_ROWS=0 (tuning aid) changes speed only.
  // Taken where it measured ahead: convolutions (KW >= 3: a k = 1 Linear stages a slab per 6 MFMAs here, conv_gemm's multi-chunk items do
  // better) on grids of at most one 64 x 64 workgroup per CU -- B = 1: the decoder's k = 9 convolution 63 -> 40 us, the postnet's 52 -> 36;
  // at B = 4 conv_gemm's several workgroups per CU already hide what this kernel avoids.
  static const bool rows_on = !(getenv("E2ETTS_ROWS") && atoi(getenv("E2ETTS_ROWS")) == 0);
  const long long wg64 = (long long)((p.T + 63) / 64) * p.B * ((p.Cout + 63) / 64);
  if (rows_on && p.KW >= 3 && wg64 <= 256 && tile_few_rows(p.B, p.T, p.Cout) && conv_rows_supported(p)) {
    ProfScope ps(e, fine && e->prof_on ? fname : (p.x3 ? "conv_x3_rows" : "conv_rows"), conv_gemm_flops(p) * alg_scale, conv_gemm_bytes(p));
    KCHK(e, launch_conv_rows(p, e->stream));
    return E2ETTS_OK;
  }
  ProfScope ps(e, fine && e->prof_on ? fname : conv_gemm_class(p), conv_gemm_flops(p) * alg_scale, conv_gemm_bytes(p));
  KCHK(e, launch_conv_gemm(p, e->stream));
  return E2ETTS_OK;
}

void free_frags(e2etts_engine* e) {
  for (auto& kv : e->frag_of) {
    (void)hipFree(kv.second);
  }
  e->frag_of.clear();
  for (auto& kv : e->bimg_of) (void)hipFree(kv.second.first);
  e->bimg_of.clear();
  for (float* f : e->rb_frag_base)
    if (f) (void)hipFree(f);
  e->rb_frag_base.clear();
  e->rb_pair_frag.clear();
  e->rb_pair_frag32.clear();
  e->stage_fused.clear();
  e->dev_bytes -= e->frag_bytes;
  e->frag_bytes = 0;
}

int make_frag(e2etts_engine* e, const float* wx3, uint64_t cout, uint64_t kw, uint64_t cin) {
  if (!wx3 || cout <= 64 || e->frag_of.count(wx3)) return E2ETTS_OK;  // <= 64 columns: the LDS-tile kernels measured no gain
  float* f = nullptr;
  const size_t bytes = x3_frag_bytes((int)cout, (int)kw, (int)cin);
  HIPCHK(e, hipMalloc(&f, bytes));
  e->dev_bytes += bytes;
  e->frag_bytes += bytes;
  e->frag_of[wx3] = f;
  KCHK(e, launch_x3_to_frag(wx3, f, (int)cout, (int)kw, (int)cin, e->stream));
  return E2ETTS_OK;
}

// fp32 weights of a layer the 128-column kernels serve (Cout > 64) -> fp32 fragment order, for the exact-fp32 mode.
// phoneme_level: a layer conv_ksplit.hip serves -- that kernel has no other weight format, so its image is made whatever the tuning aid says.
int make_frag32(e2etts_engine* e, const float* w, uint64_t cout, uint64_t kw, uint64_t cin, bool narrow = false, bool phoneme_level = false) {
  static const bool off = getenv("E2ETTS_NO_FRAG32") != nullptr;  // tuning aid: A/B against the LDS weight tile
  if ((off && !phoneme_level) || !w || (cout <= 64 && !narrow && !phoneme_level) || e->frag_of.count(w)) return E2ETTS_OK;
  float* f = nullptr;
  const size_t bytes = x3_frag_bytes((int)cout, (int)kw, (int)cin);
  HIPCHK(e, hipMalloc(&f, bytes));
  e->dev_bytes += bytes;
  e->frag_bytes += bytes;
  e->frag_of[w] = f;
  KCHK(e, launch_f32_to_frag(w, f, (int)cout, (int)kw, (int)cin, e->stream));
  return E2ETTS_OK;
}

// hi halves of a vocoder convolution's split-precision image in conv_bf16.hip's order (plain-bf16 mode).  tap_split: see BConvParams.
int make_bimg(e2etts_engine* e, const float* wx3, uint64_t cout, uint64_t kw, uint64_t cin, int tap_split = 0) {
  if (!wx3 || (cout % 32) || e->bimg_of.count(wx3)) return E2ETTS_OK;
  if (tap_split > 0 && (kw != 3 || (tap_split % 32))) tap_split = 0;   // no polyphase image: the kernel multiplies the zeros
  void* img = nullptr;
  const size_t bytes = bf16_image_bytes((int)cout, (int)kw, (int)cin, tap_split);
  HIPCHK(e, hipMalloc(&img, bytes));
  e->dev_bytes += bytes;
  e->frag_bytes += bytes;
  e->bimg_of[wx3] = {img, tap_split};
  KCHK(e, launch_bf16_image(wx3, img, (int)cout, (int)kw, (int)cin, tap_split, e->stream));
  return E2ETTS_OK;
}

int get_tensor(e2etts_engine* e, const std::string& name, uint64_t numel, const float** out) {
  auto it = e->tensors.find(name);
  if (it == e->tensors.end()) return e->fail(E2ETTS_EKEY, "weight blob has no tensor '%s'", name.c_str());
  if (numel && it->second.second != numel)
    return e->fail(E2ETTS_EINVAL, "tensor '%s' has %llu elements, expected %llu", name.c_str(),
                   (unsigned long long)it->second.second, (unsigned long long)numel);
  *out = it->second.first;
  return E2ETTS_OK;
}

int bind_fft(e2etts_engine* e, const char* side, int layers, std::vector<FFTLayer>& v) {
  const auto& c = e->cfg;
  const uint64_t H = c.hidden, F = c.ffn_dim;
  v.resize(layers);
  for (int l = 0; l < layers; ++l) {
    std::string p = std::string(side) + "." + std::to_string(l) + ".";
    FFTLayer& f = v[l];
    RET(get_tensor(e, p + "wqkv", 3 * H * H, &f.wqkv));
    RET(get_tensor(e, p + "bqkv", 3 * H, &f.bqkv));
    RET(get_tensor(e, p + "wo", H * H, &f.wo));
    RET(get_tensor(e, p + "bo", H, &f.bo));
    RET(get_tensor(e, p + "ln1.g", H, &f.ln1g));
    RET(get_tensor(e, p + "ln1.b", H, &f.ln1b));
    RET(get_tensor(e, p + "w1", F * c.ffn_k1 * H, &f.w1));
    RET(get_tensor(e, p + "b1", F, &f.b1));
    RET(get_tensor(e, p + "w2", H * F, &f.w2));
    RET(get_tensor(e, p + "b2", H, &f.b2));
    RET(get_tensor(e, p + "ln2.g", H, &f.ln2g));
    RET(get_tensor(e, p + "ln2.b", H, &f.ln2b));
    const bool ph = side[0] == 'e';  // the encoder runs over phonemes
    RET(make_frag32(e, f.wqkv, 3 * H, 1, H));
    RET(make_frag32(e, f.wo, H, 1, H));
    RET(make_frag32(e, f.w1, F, c.ffn_k1, H, false, ph));
    RET(make_frag32(e, f.w2, H, 1, F, false, ph));
    const uint64_t Hc = (H + 31) / 32 * 32, Fc = (F + 31) / 32 * 32;
    if (e->tensors.count(p + "wqkv.x3")) {
      RET(get_tensor(e, p + "wqkv.x3", 3 * H * Hc, &f.wqkv_x3));
      RET(get_tensor(e, p + "wo.x3", H * Hc, &f.wo_x3));
      RET(get_tensor(e, p + "w1.x3", F * c.ffn_k1 * Hc, &f.w1_x3));
      RET(get_tensor(e, p + "w2.x3", H * Fc, &f.w2_x3));
      RET(make_frag(e, f.wqkv_x3, 3 * H, 1, H));
      RET(make_frag(e, f.wo_x3, H, 1, H));
      RET(make_frag(e, f.w1_x3, F, c.ffn_k1, H));
      RET(make_frag(e, f.w2_x3, H, 1, F));
    }
  }
  return E2ETTS_OK;
}

int bind_cf_gemm(e2etts_engine* e, const std::string& name, const char* bias, uint64_t cout, uint64_t cin, CfGemm& g) {
  RET(get_tensor(e, name, cout * cin, &g.w));
  RET(make_frag32(e, g.w, cout, 1, cin));
  g.b = nullptr;
  if (bias) RET(get_tensor(e, bias, cout, &g.b));
  g.wx3 = nullptr;
  if (e->tensors.count(name + ".x3")) {
    RET(get_tensor(e, name + ".x3", cout * ((cin + 31) / 32 * 32), &g.wx3));
    RET(make_frag(e, g.wx3, cout, 1, cin));
  }
  return E2ETTS_OK;
}

int bind_conformer(e2etts_engine* e, const char* side, int layers, std::vector<CfLayer>& v, int n_head) {
  const auto& c = e->cfg;
  const uint64_t H = c.hidden, F = c.ffn_dim, nh = n_head;
  v.resize(layers);
  for (int l = 0; l < layers; ++l) {
    const std::string p = std::string(side) + "." + std::to_string(l) + ".";
    CfLayer& f = v[l];
    for (int i = 0; i < 2; ++i) {
      const std::string q = p + (i ? "ff2." : "ff1.");
      RET(get_tensor(e, q + "ln.g", H, &f.ff_lng[i]));
      RET(get_tensor(e, q + "ln.b", H, &f.ff_lnb[i]));
      RET(bind_cf_gemm(e, q + "w1", (q + "b1").c_str(), F, H, f.ff_a[i]));
      RET(bind_cf_gemm(e, q + "w2", (q + "b2").c_str(), H, F, f.ff_b[i]));
    }
    RET(get_tensor(e, p + "att.ln.g", H, &f.att_lng));
    RET(get_tensor(e, p + "att.ln.b", H, &f.att_lnb));
    RET(get_tensor(e, p + "att.u", H, &f.att_u));
    RET(get_tensor(e, p + "att.v", H, &f.att_v));
    RET(bind_cf_gemm(e, p + "att.wqkv", nullptr, 3 * H, H, f.att_qkv));  // LinearNorm default: no bias (U/blocks/utils.py:182)
    RET(bind_cf_gemm(e, p + "att.wo", nullptr, H, H, f.att_o));
    const uint64_t rows[2] = {(uint64_t)c.max_seq_len + 1, (uint64_t)c.pos_table_rows};
    const char* tag[2] = {"att.pos", "att.posr"};
    for (int i = 0; i < 2; ++i) {
      f.pos_rows[i] = rows[i];
      RET(get_tensor(e, p + tag[i], nh * rows[i] * (H / nh), &f.pos[i]));
      f.posx[i] = nullptr;
      if (e->tensors.count(p + tag[i] + ".x3")) RET(get_tensor(e, p + tag[i] + ".x3", nh * rows[i] * (H / nh), &f.posx[i]));
    }
    RET(get_tensor(e, p + "cv.ln.g", H, &f.cv_lng));
    RET(get_tensor(e, p + "cv.ln.b", H, &f.cv_lnb));
    RET(bind_cf_gemm(e, p + "cv.pw1.w", (p + "cv.pw1.b").c_str(), 2 * H, H, f.pw1));
    RET(get_tensor(e, p + "cv.dw.w", (uint64_t)c.ffn_k1 * H, &f.dw_w));
    RET(get_tensor(e, p + "cv.dw.b", H, &f.dw_b));
    RET(bind_cf_gemm(e, p + "cv.pw2.w", (p + "cv.pw2.b").c_str(), H, H, f.pw2));
    RET(get_tensor(e, p + "ln.g", H, &f.lng));
    RET(get_tensor(e, p + "ln.b", H, &f.lnb));
  }
  return E2ETTS_OK;
}

int bind_pred(e2etts_engine* e, const char* name, int layers, int kernel, int chans, int odim, bool alpha, Predictor& pr) {
  const uint64_t H = e->cfg.hidden;
  pr.layers.resize(layers);
  pr.kernel = kernel;
  pr.chans = chans;
  pr.odim = odim;
  for (int i = 0; i < layers; ++i) {
    std::string p = std::string(name) + "." + std::to_string(i) + ".";
    const uint64_t cin = i == 0 ? H : (uint64_t)chans;
    RET(get_tensor(e, p + "w", (uint64_t)chans * kernel * cin, &pr.layers[i].w));
    RET(make_frag32(e, pr.layers[i].w, chans, kernel, cin, false, true));
    RET(get_tensor(e, p + "b", chans, &pr.layers[i].b));
    RET(get_tensor(e, p + "g", chans, &pr.layers[i].g));
    RET(get_tensor(e, p + "beta", chans, &pr.layers[i].beta));
  }
  RET(get_tensor(e, std::string(name) + ".lin.w", (uint64_t)odim * chans, &pr.lin_w));
  RET(get_tensor(e, std::string(name) + ".lin.b", odim, &pr.lin_b));
  if (alpha) RET(get_tensor(e, std::string(name) + ".alpha", 1, &pr.alpha));
  return E2ETTS_OK;
}

int bind_acoustic(e2etts_engine* e) {
  const auto& c = e->cfg;
  const uint64_t H = c.hidden;
  RET(get_tensor(e, "enc.emb", (uint64_t)(c.n_symbols + 1) * H, &e->emb));
  RET(get_tensor(e, "enc.pos", (uint64_t)(c.max_seq_len + 1) * H, &e->enc_pos));
  RET(get_tensor(e, "dec.pos", (uint64_t)(c.max_seq_len + 1) * H, &e->dec_pos));
  RET(get_tensor(e, "pos.regen", (uint64_t)c.pos_table_rows * H, &e->pos_regen));
  RET(get_tensor(e, "spk.emb", (uint64_t)c.n_speakers * H, &e->spk_emb));
  if (c.block_type == 1) {
    RET(bind_conformer(e, "enc", c.enc_layers, e->cf_enc, c.n_head));
    RET(bind_conformer(e, "dec", c.dec_layers, e->cf_dec, c.dec_n_head ? c.dec_n_head : c.n_head));
  } else {
    RET(bind_fft(e, "enc", c.enc_layers, e->enc));
    RET(bind_fft(e, "dec", c.dec_layers, e->dec));
  }
  RET(bind_pred(e, "dur", c.dur_layers, c.dur_kernel, c.dur_chans, 1, false, e->dur));
  RET(bind_pred(e, "pitch", c.var_layers, c.var_kernel, c.var_chans, c.pitch_no_uv ? 1 : 2, true, e->pitch));
  RET(bind_pred(e, "energy", c.energy_layers ? c.energy_layers : c.var_layers, c.energy_kernel ? c.energy_kernel : c.var_kernel, c.var_chans, 1, true,
                e->energy));
  {
    auto it = e->tensors.find("var.pos");
    if (it == e->tensors.end()) return e->fail(E2ETTS_EKEY, "weight blob has no tensor 'var.pos'");
    if (it->second.second % H) return e->fail(E2ETTS_EINVAL, "var.pos size is not a multiple of hidden");
    e->var_pos = it->second.first;
    e->var_pos_rows = it->second.second / H;
  }
  RET(get_tensor(e, "pitch.emb", (uint64_t)(c.pitch_emb_rows ? c.pitch_emb_rows : c.n_bins) * H, &e->pitch_emb));
  e->pitch_bins = nullptr;
  if (c.pitch_no_uv) RET(get_tensor(e, "pitch.bins", (uint64_t)c.n_bins - 1, &e->pitch_bins));
  RET(get_tensor(e, "energy.emb", (uint64_t)c.n_bins * H, &e->energy_emb));
  RET(get_tensor(e, "energy.bins", (uint64_t)c.n_bins - 1, &e->energy_bins));
  RET(get_tensor(e, "mel.w", (uint64_t)c.n_mel * H, &e->mel_lin.w));
  RET(make_frag32(e, e->mel_lin.w, c.n_mel, 1, H));
  RET(get_tensor(e, "mel.b", c.n_mel, &e->mel_lin.b));
  e->mel_lin.wx3 = nullptr;
  if (e->tensors.count("mel.w.x3")) RET(get_tensor(e, "mel.w.x3", (uint64_t)c.n_mel * ((H + 31) / 32 * 32), &e->mel_lin.wx3));
  RET(make_frag(e, e->mel_lin.wx3, c.n_mel, 1, H));
  e->postnet.resize(c.postnet_layers);
  for (int i = 0; i < c.postnet_layers; ++i) {
    const uint64_t cin = i == 0 ? c.n_mel : c.postnet_dim, cout = i == c.postnet_layers - 1 ? c.n_mel : c.postnet_dim;
    std::string p = "post." + std::to_string(i) + ".";
    RET(get_tensor(e, p + "w", cout * c.postnet_kernel * cin, &e->postnet[i].w));
    RET(make_frag32(e, e->postnet[i].w, cout, c.postnet_kernel, cin));
    RET(get_tensor(e, p + "b", cout, &e->postnet[i].b));
    e->postnet[i].wx3 = nullptr;
    if (e->tensors.count(p + "w.x3")) RET(get_tensor(e, p + "w.x3", cout * c.postnet_kernel * ((cin + 31) / 32 * 32), &e->postnet[i].wx3));
    RET(make_frag(e, e->postnet[i].wx3, cout, c.postnet_kernel, cin));
  }
  return E2ETTS_OK;
}

// fp32 weight + bias (required) and the split-precision image (optional: older blobs do not carry it)
int bind_conv(e2etts_engine* e, const std::string& name, uint64_t cout, uint64_t kw, uint64_t cin, ConvW& w) {
  RET(get_tensor(e, name + ".w", cout * kw * cin, &w.w));
  RET(make_frag32(e, w.w, cout, kw, cin));
  RET(get_tensor(e, name + ".b", cout, &w.b));
  w.wx3 = nullptr;
  if (e->tensors.count(name + ".wx3")) RET(get_tensor(e, name + ".wx3", cout * kw * ((cin + 31) / 32) * 32, &w.wx3));
  RET(make_frag(e, w.wx3, cout, kw, cin));
  return E2ETTS_OK;
}

int bind_vocoder(e2etts_engine* e) {
  const auto& c = e->cfg;
  const uint64_t C0 = c.voc_init_ch;
  RET(bind_conv(e, "voc.pre", C0, 7, c.n_mel, e->voc_pre));
  RET(make_bimg(e, e->voc_pre.wx3, C0, 7, c.n_mel));
  e->voc_up.resize(c.voc_stages);
  e->rb_c1.assign((size_t)c.voc_stages * c.voc_n_kernels, {});
  e->rb_c2.assign((size_t)c.voc_stages * c.voc_n_kernels, {});
  e->rb_pair_frag.assign((size_t)c.voc_stages * c.voc_n_kernels, {});
  e->rb_pair_frag32.assign((size_t)c.voc_stages * c.voc_n_kernels, {});
  e->stage_fused.clear();
  uint64_t ch = C0;
  for (int i = 0; i < c.voc_stages; ++i) {
    const uint64_t cin = ch, cout = ch / 2, s = c.voc_up_rate[i];
    RET(bind_conv(e, "voc.up." + std::to_string(i), s * cout, 3, cin, e->voc_up[i]));
    // the last upsampler of HiFi-GAN V1 (64 columns), exact fp32: fragments for the zero-tap-skipping 256 x 32 tile (conv_gemm.hip)
    if (s * cout == 64) RET(make_frag32(e, e->voc_up[i].w, s * cout, 3, cin, true));
    RET(make_bimg(e, e->voc_up[i].wx3, s * cout, 3, cin, (int)(s * cout / 2)));
    ch = cout;
    for (int j = 0; j < c.voc_n_kernels; ++j) {
      const int idx = i * c.voc_n_kernels + j;
      const uint64_t k = c.voc_rb_kernel[j];
      e->rb_c1[idx].resize(c.voc_n_dil);
      e->rb_c2[idx].resize(c.voc_n_dil);
      for (int m = 0; m < c.voc_n_dil; ++m) {
        std::string q = "voc.rb." + std::to_string(idx) + ".";
        if (c.voc_resblock == 2) {  // ResBlock2: one convolution per dilation (V/layers.py:52-56)
          RET(bind_conv(e, q + "c." + std::to_string(m), ch, k, ch, e->rb_c1[idx][m]));
          continue;
        }
        RET(bind_conv(e, q + "c1." + std::to_string(m), ch, k, ch, e->rb_c1[idx][m]));
        RET(bind_conv(e, q + "c2." + std::to_string(m), ch, k, ch, e->rb_c2[idx][m]));
        RET(make_bimg(e, e->rb_c1[idx][m].wx3, ch, k, ch));
        RET(make_bimg(e, e->rb_c2[idx][m].wx3, ch, k, ch));
      }
    }
    // fused pairs: all of a stage or none (the two forms use the stage's scratch buffers differently)
    bool fusable = c.voc_resblock == 1;
    for (int j = 0; j < c.voc_n_kernels && fusable; ++j)
      for (int m = 0; m < c.voc_n_dil; ++m)
        fusable = fusable && resblock_pair_supported((int)ch, c.voc_rb_kernel[j], c.voc_rb_dil[j][m]) &&
                  e->rb_c1[i * c.voc_n_kernels + j][m].wx3 && e->rb_c2[i * c.voc_n_kernels + j][m].wx3;
    e->stage_fused.push_back(fusable ? 1 : 0);
    for (int j = 0; j < c.voc_n_kernels && fusable; ++j) {
      const int idx = i * c.voc_n_kernels + j;
      const int k = c.voc_rb_kernel[j];
      // ONE allocation per ResBlock: conv1 | conv2 of pair 0, pair 1, ... contiguous, so that the whole-ResBlock kernel
      // (resblock_chain.hip) walks all of them through one buffer descriptor; rb_pair_frag[idx][m] points at pair m
      const size_t one = x3_frag_bytes((int)ch, k, (int)ch);
      float* base = nullptr;
      HIPCHK(e, hipMalloc(&base, 2 * one * c.voc_n_dil));
      e->dev_bytes += 2 * one * c.voc_n_dil;
      e->frag_bytes += 2 * one * c.voc_n_dil;
      e->rb_frag_base.push_back(base);
      for (int m = 0; m < c.voc_n_dil; ++m) {
        float* f = base + (size_t)m * 2 * (one / 4);
        e->rb_pair_frag[idx].push_back(f);
        KCHK(e, launch_x3_to_frag(e->rb_c1[idx][m].wx3, f, (int)ch, k, (int)ch, e->stream));
        KCHK(e, launch_x3_to_frag(e->rb_c2[idx][m].wx3, f + one / 4, (int)ch, k, (int)ch, e->stream));
      }
      // at 128 / 256 channels the fp32 kernels are bound by the matrix pipe and the fused form recomputes (KW - 1) / 128 of its rows:
      // it pays up to a kernel size that the environment can move (tuning aid; same bits either way)
      static const int f32_k128 = getenv("E2ETTS_F32_FUSE_K128") ? atoi(getenv("E2ETTS_F32_FUSE_K128")) : 7;
      static const int f32_k256 = getenv("E2ETTS_F32_FUSE_K256") ? atoi(getenv("E2ETTS_F32_FUSE_K256")) : 0;
      if (ch <= 64 || (ch == 128 && k <= f32_k128) || (ch == 256 && k <= f32_k256)) {  // fp32 images of the same pairs
        float* b32 = nullptr;
        HIPCHK(e, hipMalloc(&b32, 2 * one * c.voc_n_dil));
        e->dev_bytes += 2 * one * c.voc_n_dil;
        e->frag_bytes += 2 * one * c.voc_n_dil;
        e->rb_frag_base.push_back(b32);
        for (int m = 0; m < c.voc_n_dil; ++m) {
          float* f = b32 + (size_t)m * 2 * (one / 4);
          e->rb_pair_frag32[idx].push_back(f);
          KCHK(e, launch_f32_to_frag(e->rb_c1[idx][m].w, f, (int)ch, k, (int)ch, e->stream));
          KCHK(e, launch_f32_to_frag(e->rb_c2[idx][m].w, f + one / 4, (int)ch, k, (int)ch, e->stream));
        }
      }
    }
  }
  if (c.voc_istft_nfft) {  // conv_post to n_fft + 2 channels (padded to a multiple of 4 by the packer), V/generator.py:92
    const uint64_t pc = ((uint64_t)c.voc_istft_nfft + 2 + 3) / 4 * 4;
    RET(bind_conv(e, "voc.post", pc, 7, ch, e->voc_post));
  } else {
    RET(get_tensor(e, "voc.post.w", 7 * ch, &e->voc_post.w));
    RET(get_tensor(e, "voc.post.b", 1, &e->voc_post.b));
  }
  return E2ETTS_OK;
}

// 6 x FFTBlock (reference U/blocks/transformer.py:178-189), in place on x ([B, N, H]); lens32: device [B]
int fft_stack(e2etts_engine* e, const std::vector<FFTLayer>& layers, int n_head, float* x, float* xalt, const int32_t* lens, int B, int N, bool x3,
              const int32_t* act = nullptr, double act_frac = 1.0, bool ksplit = false, const int32_t* act_host = nullptr) {
  const auto& c = e->cfg;
  const int H = c.hidden, F = c.ffn_dim;
  float* qkv = ptr<float>(e->qkv);
  float* att = ptr<float>(e->att);
  float* tmp = ptr<float>(e->tmp);
  float* hid = ptr<float>(e->hid);
  for (const FFTLayer& f : layers) {
    ConvParams p;
    p.B = B; p.T = N; p.act_rows = act; p.act_rows_host = act_host; p.act_frac = act_frac;
    // q | k | v projections as one GEMM (U/blocks/transformer.py:220-222)
    const bool sx = x3 && f.wqkv_x3;
    p.in = x; p.w = sx ? f.wqkv_x3 : f.wqkv; p.x3 = sx; p.bias = f.bqkv; p.out = qkv; p.Cin = H; p.Cout = 3 * H;
    RET(conv(e, p, 1.0, ksplit));
    {
      const double fl = 4.0 * B * n_head * (double)N * N * (H / n_head);
      ProfScope ps(e, sx ? "attention_x3" : "attention", fl, 4.0 * 4.0 * B * N * H);
      float* aws = nullptr;   // workspace for the parallel key segments of small fp32 grids (attention.hip); 3.6 MB at B = 1, T = 768
      size_t aws_bytes = 0;
      if (!sx && !act_host && (long long)B * ((N + 63) / 64) * n_head <= attention_par_max_grid() && N > 32 * 8) {
        aws_bytes = attention_workspace_bytes(B, N, H, n_head);
        RET(ensure(e, e->attws, aws_bytes));
        aws = ptr<float>(e->attws);
      }
      KCHK(e, launch_attention(qkv, att, lens, B, N, H, n_head, sx ? 1 : 0, e->stream, act_host, aws, aws_bytes));
    }
    // fc + residual (:238-239), LayerNorm eps 1e-5, masked_fill (:182-183)
    p = ConvParams(); p.B = B; p.T = N; p.act_rows = act; p.act_rows_host = act_host; p.act_frac = act_frac;
    p.in = att; p.w = sx ? f.wo_x3 : f.wo; p.x3 = sx; p.bias = f.bo; p.res = x; p.out = tmp; p.Cin = H; p.Cout = H;
    RET(conv(e, p, 1.0, ksplit));
    {
      ProfScope ps(e, "layernorm", 0, 8.0 * B * N * H);
      KCHK(e, launch_layernorm(tmp, xalt, f.ln1g, f.ln1b, lens, B, N, H, 1e-5f, e->stream));
    }
    // conv k9 + ReLU, conv k1 + residual, LayerNorm, masked_fill (:289-297, :185-187)
    p = ConvParams(); p.B = B; p.T = N; p.act_rows = act; p.act_rows_host = act_host; p.act_frac = act_frac;
    p.in = xalt; p.w = sx ? f.w1_x3 : f.w1; p.x3 = sx; p.bias = f.b1; p.out = hid; p.Cin = H; p.Cout = F; p.KW = c.ffn_k1; p.pad = (c.ffn_k1 - 1) / 2;
    p.act = ACT_RELU;
    RET(conv(e, p, 1.0, ksplit));
    p = ConvParams(); p.B = B; p.T = N; p.act_rows = act; p.act_rows_host = act_host; p.act_frac = act_frac;
    p.in = hid; p.w = sx ? f.w2_x3 : f.w2; p.x3 = sx; p.bias = f.b2; p.res = xalt; p.out = tmp; p.Cin = F; p.Cout = H;
    RET(conv(e, p, 1.0, ksplit));
    {
      ProfScope ps(e, "layernorm", 0, 8.0 * B * N * H);
      KCHK(e, launch_layernorm(tmp, x, f.ln2g, f.ln2b, lens, B, N, H, 1e-5f, e->stream));
    }
  }
  return E2ETTS_OK;
}

// n x ConformerBlock (reference U/blocks/conformer.py:214-255), in place on x ([B, N, H]).  Every sub-module is a pre-norm residual
// unit; nothing inside a block is masked (nn.Sequential hands the attention module no mask, :252, and the depthwise convolution
// runs over the padded length), so all N rows are computed; only the block's final LayerNorm output is zeroed at rows >= lens[b].
int conformer_stack(e2etts_engine* e, const std::vector<CfLayer>& layers, int n_head, float* x, float* xalt, const int32_t* lens, int B, int N, bool x3) {
  const auto& c = e->cfg;
  const int H = c.hidden, F = c.ffn_dim, nh = n_head, dh = H / nh;
  float* qkv = ptr<float>(e->qkv);
  float* att = ptr<float>(e->att);
  float* tmp = ptr<float>(e->tmp);
  float* hid = ptr<float>(e->hid);
  float *cur = x, *oth = xalt;
  const int tsel = N > c.max_seq_len ? 1 : 0;  // eval-time regenerated table (:339-344)
  auto gemm = [&](const CfGemm& g, const float* in, float* out, int cin, int cout, const float* res, int act = ACT_NONE) -> int {
    ConvParams p;
    p.B = B; p.T = N; p.in = in; p.out = out; p.Cin = cin; p.Cout = cout; p.bias = g.b; p.res = res; p.act = act;
    const bool sx = x3 && g.wx3;
    p.w = sx ? g.wx3 : g.w; p.x3 = sx;
    return conv(e, p);
  };
  auto ln = [&](const float* in, float* out, const float* g, const float* b, const int32_t* mask) -> int {
    ProfScope ps(e, "layernorm", 0, 8.0 * B * N * H);
    KCHK(e, launch_layernorm(in, out, g, b, mask, B, N, H, 1e-5f, e->stream));
    return E2ETTS_OK;
  };
  for (const CfLayer& f : layers) {
    if ((uint64_t)N > f.pos_rows[tsel])
      return e->fail(E2ETTS_EINVAL, "sequence of %d rows exceeds the Conformer position table (%llu rows)", N, (unsigned long long)f.pos_rows[tsel]);
    for (int half = 0; half < 2; ++half) {
      if (half == 1) {
        // MultiHeadedSelfAttentionModule (:335-353) + RelativeMultiHeadAttention (:399-440)
        RET(ln(cur, tmp, f.att_lng, f.att_lnb, nullptr));
        RET(gemm(f.att_qkv, tmp, qkv, H, 3 * H, nullptr));
        {  // the shifted position scores are computed inside the kernel, from this layer's projected table (attention.hip)
          // FLOPs: content scores and P . V (4 N^2 d_h per head) + the position bands (2 x 32 x 64 x d_h per 32 x 32 tile = 4 N^2 d_h)
          const double fl = 8.0 * B * nh * (double)N * N * dh;
          ProfScope ps(e, "rel_attention", fl, 4.0 * B * (4.0 * N * H) + 4.0 * nh * N * dh);
          KCHK(e, launch_rel_attention(qkv, f.pos[tsel], (int)f.pos_rows[tsel], f.att_u, f.att_v, att, B, N, H, nh, e->stream,
                                       x3 ? f.posx[tsel] : nullptr));
        }
        RET(gemm(f.att_o, att, oth, H, H, cur));
        std::swap(cur, oth);
        // ConformerConvModule (:468-481): LayerNorm, pointwise 2H + GLU, depthwise k + BatchNorm + Swish, pointwise
        RET(ln(cur, tmp, f.cv_lng, f.cv_lnb, nullptr));
        RET(gemm(f.pw1, tmp, hid, H, 2 * H, nullptr));
        {  // GLU + depthwise k + BatchNorm + Swish in one pass (small_kernels.hip); `att` is scratch for shapes without a fused form
          ProfScope ps(e, "dwconv_glu_swish", 2.0 * B * N * (double)H * c.ffn_k1, 12.0 * B * N * H);
          KCHK(e, launch_dwconv_glu_swish(hid, f.dw_w, f.dw_b, tmp, att, B, N, H, c.ffn_k1, e->stream));
        }
        RET(gemm(f.pw2, tmp, oth, H, H, cur));
        std::swap(cur, oth);
      }
      // FeedForwardModule (:294-301): LayerNorm, Linear + Swish (GEMM epilogue), Linear; x + factor * ff(x) with the factor folded into the weights
      RET(ln(cur, tmp, f.ff_lng[half], f.ff_lnb[half], nullptr));
      RET(gemm(f.ff_a[half], tmp, hid, H, F, nullptr, ACT_SWISH));
      RET(gemm(f.ff_b[half], hid, oth, F, H, cur));
      std::swap(cur, oth);
    }
    RET(ln(cur, oth, f.lng, f.lnb, lens));  // final LayerNorm (:248) + masked_fill (:253-254)
    std::swap(cur, oth);
  }
  if (cur != x) HIPCHK(e, hipMemcpyAsync(x, cur, (size_t)B * N * H * 4, hipMemcpyDeviceToDevice, e->stream));
  return E2ETTS_OK;
}

// conv -> ReLU -> channel LayerNorm(eps 1e-12) [-> x (1 - mask)] stack + small Linear
// (reference DurationPredictor U/layers.py:410-420, VariancePredictor :499-503)
int predictor(e2etts_engine* e, const Predictor& pr, const float* x, float* out, const int32_t* mask_lens, int B, int L,
              const int32_t* act = nullptr, const int32_t* act_host = nullptr, double act_frac = 1.0, bool frame_level = false) {
  const int H = e->cfg.hidden;
  float* a = ptr<float>(e->p1);  // conv output
  float* b = ptr<float>(e->p2);  // LayerNorm output = next layer's input
  const float* in = x;
  int cin = H;
  for (const PredLayer& l : pr.layers) {
    ConvParams p;
    p.B = B; p.T = L; p.in = in; p.w = l.w; p.bias = l.b; p.out = a; p.Cin = cin; p.Cout = pr.chans;
    p.act_rows = act; p.act_rows_host = act_host; p.act_frac = act_frac;
    p.KW = pr.kernel; p.pad = e->cfg.pred_pad_left ? pr.kernel - 1 : (pr.kernel - 1) / 2; p.act = ACT_RELU;  // ConstantPad1d, U/layers.py:400-402
    RET(conv(e, p, 1.0, !frame_level));  // phoneme-level layer: conv_ksplit.hip at every batch size; frame-level (B x T rows): conv_gemm at every batch size
    {
      ProfScope ps(e, "layernorm", 0, 8.0 * B * L * pr.chans);
      KCHK(e, launch_layernorm(a, b, l.g, l.beta, mask_lens, B, L, pr.chans, 1e-12f, e->stream));
    }
    in = b;
    cin = pr.chans;
  }
  {
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_rowdot(in, pr.lin_w, pr.lin_b, out, mask_lens, B, L, pr.chans, pr.odim, e->stream));
  }
  return E2ETTS_OK;
}

bool is_device_pointer(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}

int copy_in(e2etts_engine* e, void* dst, const void* src, size_t bytes) {
  HIPCHK(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, e->stream));
  return E2ETTS_OK;
}

int copy_out(e2etts_engine* e, void* dst, const void* src, size_t bytes) {
  HIPCHK(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, e->stream));
  return E2ETTS_OK;
}

__global__ void lens_to_i32_kernel(const int64_t* in, int32_t* out, int B, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    long long v = in[i];
    out[i] = (int32_t)(v < 0 ? 0 : (v > L ? L : v));
  }
}

int acoustic_impl(e2etts_engine* e, const int64_t* ids, const int64_t* lens, int B, int L, const int64_t* speaker,
                  int n_spk_ids, float d_control, float p_control, float e_control, bool ragged = false) {
  const auto& c = e->cfg;
  if (!e->ac_loaded) return e->fail(E2ETTS_ESTATE, "acoustic weights not loaded");
  if (!ids || !lens || !speaker) return e->fail(E2ETTS_EINVAL, "ids / lens / speaker must not be NULL");
  if (B <= 0 || L <= 0) return e->fail(E2ETTS_EINVAL, "B and L must be positive (got B=%d L=%d)", B, L);
  if (n_spk_ids != 1 && n_spk_ids != B) return e->fail(E2ETTS_EINVAL, "speaker must hold 1 or B ids");
  if (L > c.max_seq_len && L > c.pos_table_rows)
    return e->fail(E2ETTS_EINVAL, "L=%d exceeds the shipped position table (%d rows)", L, c.pos_table_rows);
  if ((uint64_t)L + 1 > e->var_pos_rows)
    return e->fail(E2ETTS_EINVAL, "L=%d exceeds the variance-predictor position table (%llu rows)", L,
                   (unsigned long long)e->var_pos_rows);
  // host-side validation when the caller's buffers are host memory (KeyError / IndexError in the reference)
  if (!is_device_pointer(ids) && !is_device_pointer(lens)) {
    for (int b = 0; b < B; ++b)
      if (lens[b] < 1 || lens[b] > L) return e->fail(E2ETTS_EINVAL, "lens[%d]=%lld outside [1, %d]", b, (long long)lens[b], L);
    for (long long i = 0; i < (long long)B * L; ++i)
      if (ids[i] < 0 || ids[i] > c.n_symbols) return e->fail(E2ETTS_EINVAL, "symbol id %lld out of range", (long long)ids[i]);
  }
  if (!is_device_pointer(speaker))
    for (int i = 0; i < n_spk_ids; ++i)
      if (speaker[i] < 0 || speaker[i] >= c.n_speakers) return e->fail(E2ETTS_EINVAL, "speaker id %lld out of range", (long long)speaker[i]);

  const int H = c.hidden, F = c.ffn_dim;
  const size_t BL = (size_t)B * L;
  RET(ensure(e, e->ids, BL * 8));
  RET(ensure(e, e->lens64, (size_t)B * 8));
  RET(ensure(e, e->lens32, (size_t)B * 4));
  RET(ensure(e, e->spk, (size_t)B * 8));
  RET(ensure(e, e->xa, BL * H * 4));
  RET(ensure(e, e->xb, BL * H * 4));
  RET(ensure(e, e->xs, BL * H * 4));
  RET(ensure(e, e->xp, BL * H * 4));
  RET(ensure(e, e->tmp, BL * H * 4));
  RET(ensure(e, e->qkv, BL * 3 * H * 4));
  RET(ensure(e, e->att, BL * H * 4));
  RET(ensure(e, e->hid, BL * F * 4));
  const size_t pc = std::max(c.dur_chans, c.var_chans);
  RET(ensure(e, e->p1, BL * pc * 4));
  RET(ensure(e, e->p2, BL * pc * 4));
  RET(ensure(e, e->logd, BL * 4));
  RET(ensure(e, e->durf, BL * 4));
  RET(ensure(e, e->cum, BL * 4));
  RET(ensure(e, e->mel64, (size_t)B * 8));
  RET(ensure(e, e->mel32, (size_t)B * 4));
  RET(ensure(e, e->posbuf, BL * 4));
  RET(ensure(e, e->ppred, BL * 2 * 4));
  RET(ensure(e, e->epred, BL * 4));
  RET(ensure(e, e->pidx, BL * 4));
  RET(ensure(e, e->eidx, BL * 4));

  e->have_acoustic = false;
  RET(copy_in(e, e->ids.p, ids, BL * 8));
  RET(copy_in(e, e->lens64.p, lens, (size_t)B * 8));
  RET(copy_in(e, e->spk.p, speaker, (size_t)n_spk_ids * 8));
  hipLaunchKernelGGL(lens_to_i32_kernel, dim3((B + 63) / 64), dim3(64), 0, e->stream, ptr<int64_t>(e->lens64),
                     ptr<int32_t>(e->lens32), B, L);
  const int32_t* tl = ptr<int32_t>(e->lens32);

  // Encoder (U/blocks/transformer.py:58-86): embedding + position table, 6 x FFTBlock
  const float* pos = L > c.max_seq_len ? e->pos_regen : e->enc_pos;  // eval-time regeneration branch (:68-73)
  float* x = ptr<float>(e->xa);
  {
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_embed(ptr<int64_t>(e->ids), e->emb, pos, x, B, L, H, c.n_symbols + 1, e->stream));
  }
  // Ragged mode at the phoneme level (synthesize, lengths in HOST memory -- the compact grids are built from them).  FFT blocks and
  // duration predictor: every consumer of a row >= len masks it (keys masked, LayerNorm kernels write zeros there), so their convolutions
  // compute rows < len only.  Pitch / energy predictors are unmasked stacks of n convolutions of kernel k: the rows < len that the bucket
  // kernel reads depend on rows < len + (n - 1)(k - 1)/2 of the layers before, so every layer computes that many (cap L).
  const int32_t *act_enc = nullptr, *act_enc_h = nullptr, *act_var = nullptr, *act_var_h = nullptr;
  double frac_enc = 1.0, frac_var = 1.0;
  bool enc_short = false;  // some utterance is shorter than the padded length (else every limit equals L: nothing to skip, and the
                           // launches keep their plain grids -- the B = 1 latency path pays no act_rows kernels)
  if (ragged && c.block_type == 0 && !is_device_pointer(lens))
    for (int b = 0; b < B; ++b) enc_short = enc_short || lens[b] < L;
  if (enc_short) {
    RET(ensure(e, e->actbuf, (size_t)(5 + c.voc_stages) * B * 4));
    e->h_act.resize((size_t)(5 + c.voc_stages) * B, 0);
    int32_t* he = e->h_act.data() + (size_t)(3 + c.voc_stages) * B;
    int32_t* hv = he + B;
    // (one limit for both variance predictors: the deeper reach of the two)
    const int add_var = std::max((c.var_layers - 1) * ((c.var_kernel - 1) / 2),
                                 ((c.energy_layers ? c.energy_layers : c.var_layers) - 1) * (((c.energy_kernel ? c.energy_kernel : c.var_kernel) - 1) / 2));
    double se = 0, sv = 0;
    for (int b = 0; b < B; ++b) {
      he[b] = (int32_t)std::min<long long>(std::max<long long>(lens[b], 0), L);   // lens_to_i32_kernel's clamp
      hv[b] = (int32_t)std::min<long long>((long long)he[b] + add_var, L);        // act_rows_kernel's arithmetic
      se += he[b];
      sv += hv[b];
    }
    int32_t* dv = ptr<int32_t>(e->actbuf) + (size_t)(4 + c.voc_stages) * B;
    KCHK(e, launch_act_rows(tl, dv, B, add_var, 1, L, e->stream));
    act_enc = tl; act_enc_h = he; act_var = dv; act_var_h = hv;
    frac_enc = se / ((double)B * L);
    frac_var = sv / ((double)B * L);
  }
  if (c.block_type == 1) RET(conformer_stack(e, e->cf_enc, c.n_head, x, ptr<float>(e->xb), tl, B, L, false));
  else RET(fft_stack(e, e->enc, c.n_head, x, ptr<float>(e->xb), tl, B, L, false, act_enc, frac_enc, true, act_enc_h));  // encoder: always exact fp32, K-split kernel

  // Variance adaptor, inference branch (U/layers.py:195-258)
  float* xs = ptr<float>(e->xs);
  {
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_add_speaker(x, xs, e->spk_emb, ptr<int64_t>(e->spk), n_spk_ids, c.n_speakers, B, L, H, e->stream));
  }
  RET(predictor(e, e->dur, xs, ptr<float>(e->logd), tl, B, L, act_enc, act_enc_h, frac_enc));
  {
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_duration(ptr<float>(e->logd), d_control, ptr<float>(e->durf), ptr<int32_t>(e->cum),
                            ptr<int64_t>(e->mel64), ptr<int32_t>(e->mel32), B, L, e->stream));
  }
  HIPCHK(e, hipMemcpyAsync(e->h_mel, e->mel64.p, (size_t)B * 8, hipMemcpyDeviceToHost, e->stream));
  hipEvent_t mel_ready = get_event(e);
  HIPCHK(e, hipEventRecord(mel_ready, e->stream));
  // pitch / energy predictors run while the mel lengths travel to the host
  float* xp = ptr<float>(e->xp);
  const bool p_frame = c.pitch_frame != 0, e_frame = c.energy_frame != 0;   // frame_level features wait for the length regulator (below)
  const int pitch_mode = c.pitch_no_uv ? 2 : (c.pitch_log2 ? 1 : 0);
  if (!p_frame) {
    {
      ProfScope ps(e, "misc", 0, 0);
      KCHK(e, launch_var_positions(xs, ptr<int32_t>(e->posbuf), e->var_pos, (int)e->var_pos_rows, e->pitch.alpha, xp, B, L, H, e->stream));
    }
    RET(predictor(e, e->pitch, xp, ptr<float>(e->ppred), nullptr, B, L, act_var, act_var_h, frac_var));
  }
  if (!e_frame) {
    {
      ProfScope ps(e, "misc", 0, 0);
      KCHK(e, launch_var_positions(xs, ptr<int32_t>(e->posbuf), e->var_pos, (int)e->var_pos_rows, e->energy.alpha, xp, B, L, H, e->stream,
                                   p_frame));  // the positions depend on xs alone: the pitch predictor's pass (if it ran) left them in posbuf
    }
    RET(predictor(e, e->energy, xp, ptr<float>(e->epred), nullptr, B, L, act_var, act_var_h, frac_var));
  }
  if (!p_frame || !e_frame) {
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_variance_embed(xs, ptr<float>(e->ppred), ptr<float>(e->epred), p_control, e_control, c.f0_mean, c.f0_std,
                                  e->energy_bins, c.n_bins, e->pitch_emb, e->energy_emb, ptr<int32_t>(e->pidx),
                                  ptr<int32_t>(e->eidx), B, L, H, e->stream, pitch_mode, e->pitch_bins, (p_frame ? 0 : 1) | (e_frame ? 0 : 2)));
  }
  // the one host synchronisation of the acoustic model: T = max(mel_lens) sizes everything downstream
  HIPCHK(e, hipEventSynchronize(mel_ready));
  e->ev_pool.push_back(mel_ready);
  long long T = 0;
  for (int b = 0; b < B; ++b) T = std::max<long long>(T, e->h_mel[b]);
  if (T <= 0) return e->fail(E2ETTS_EINVAL, "every predicted duration is zero: nothing to synthesise");
  if (T > c.max_seq_len && T > c.pos_table_rows)
    return e->fail(E2ETTS_EINVAL, "T=%lld exceeds the shipped position table (%d rows)", T, c.pos_table_rows);
  if (T > (1 << 20)) return e->fail(E2ETTS_EINVAL, "T=%lld is unreasonably large", T);
  const size_t BT = (size_t)B * T;
  RET(ensure(e, e->dx, BT * H * 4));
  RET(ensure(e, e->dxb, BT * H * 4));
  RET(ensure(e, e->tmp, BT * H * 4));
  RET(ensure(e, e->qkv, BT * 3 * H * 4));
  RET(ensure(e, e->att, BT * H * 4));
  RET(ensure(e, e->hid, BT * F * 4));
  RET(ensure(e, e->mel, BT * c.n_mel * 4));
  RET(ensure(e, e->melpost, BT * c.n_mel * 4));
  RET(ensure(e, e->pn1, BT * c.postnet_dim * 4));
  RET(ensure(e, e->pn2, BT * c.postnet_dim * 4));

  // Length regulator (U/layers.py:423-457) fused with the decoder's position add (U/blocks/transformer.py:138-153)
  const float* dpos = T > c.max_seq_len ? e->pos_regen : e->dec_pos;
  float* dx = ptr<float>(e->dx);
  const int32_t* ml = ptr<int32_t>(e->mel32);
  {
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_length_regulate(xs, ptr<int32_t>(e->cum), ml, (p_frame || e_frame) ? nullptr : dpos, dx, B, L, (int)T, H, e->stream));
  }
  if (p_frame || e_frame) {
    // frame_level pitch / energy (U/layers.py:249-257): both predictors read the length regulator's output, their embeddings are added
    // to it -- on every row, padded ones included, as the reference does; the decoder masks those -- and the decoder's positions follow
    if (T + 1 > (long long)e->var_pos_rows)
      return e->fail(E2ETTS_EINVAL, "T=%lld exceeds the variance predictors' position table (%lld rows)", T, (long long)e->var_pos_rows);
    const size_t pcw = (size_t)std::max(c.dur_chans, c.var_chans);
    RET(ensure(e, e->p1, BT * pcw * 4));
    RET(ensure(e, e->p2, BT * pcw * 4));
    RET(ensure(e, e->posbuf, BT * 4));
    if (p_frame) { RET(ensure(e, e->ppred, BT * 2 * 4)); RET(ensure(e, e->pidx, BT * 4)); }
    if (e_frame) { RET(ensure(e, e->epred, BT * 4)); RET(ensure(e, e->eidx, BT * 4)); }
    float* xpf = ptr<float>(e->dxb);   // [B, T, H]: free until the decoder starts
    if (p_frame) {
      {
        ProfScope ps(e, "misc", 0, 0);
        KCHK(e, launch_var_positions(dx, ptr<int32_t>(e->posbuf), e->var_pos, (int)e->var_pos_rows, e->pitch.alpha, xpf, B, (int)T, H, e->stream));
      }
      RET(predictor(e, e->pitch, xpf, ptr<float>(e->ppred), nullptr, B, (int)T, nullptr, nullptr, 1.0, true));
    }
    if (e_frame) {
      {
        ProfScope ps(e, "misc", 0, 0);
        KCHK(e, launch_var_positions(dx, ptr<int32_t>(e->posbuf), e->var_pos, (int)e->var_pos_rows, e->energy.alpha, xpf, B, (int)T, H, e->stream,
                                     !p_frame));
      }
      RET(predictor(e, e->energy, xpf, ptr<float>(e->epred), nullptr, B, (int)T, nullptr, nullptr, 1.0, true));
    }
    ProfScope ps(e, "misc", 0, 0);
    KCHK(e, launch_variance_embed(dx, ptr<float>(e->ppred), ptr<float>(e->epred), p_control, e_control, c.f0_mean, c.f0_std,
                                  e->energy_bins, c.n_bins, e->pitch_emb, e->energy_emb, ptr<int32_t>(e->pidx),
                                  ptr<int32_t>(e->eidx), B, (int)T, H, e->stream, pitch_mode, e->pitch_bins, (p_frame ? 1 : 0) | (e_frame ? 2 : 0)));
    KCHK(e, launch_add_positions(dx, dpos, B, (int)T, H, e->stream));
  }
  // Ragged mode (synthesize only).  Decoder: every consumer of a row >= mel_len masks it (keys are masked, the LayerNorm
  // kernels write zeros there), so its convolutions compute rows < mel_len only and valid rows stay bit-identical.
  // mel_linear / postnet are unmasked and the vocoder reads `halo` frames past the end, so they compute rows
  // < mel_len + halo + 2 * postnet_layers; rows beyond hold stale finite values that no valid sample depends on.
  const int32_t *act_dec = nullptr, *act_post = nullptr, *act_dec_h = nullptr, *act_post_h = nullptr;
  // (equal lengths -- B = 1, a fixed-length batch --: every limit would equal T; nothing is skipped and no act_rows kernel is launched)
  e->rag_short = false;
  for (int b = 0; b < B; ++b) e->rag_short = e->rag_short || e->h_mel[b] < T;
  if (ragged && e->rag_short) {
    RET(ensure(e, e->actbuf, (size_t)(5 + c.voc_stages) * B * 4));
    int32_t* ab = ptr<int32_t>(e->actbuf);
    const int halo = vocoder_halo_frames(c);
    KCHK(e, launch_act_rows(ml, ab, B, 0, 1, T, e->stream));
    KCHK(e, launch_act_rows(ml, ab + B, B, halo + 2 * c.postnet_layers * ((c.postnet_kernel - 1) / 2), 1, T, e->stream));
    act_dec = ab;
    act_post = ab + B;
    double sd = 0, sp = 0, sv = 0;  // the same limits on the host (mel lengths are in e->h_mel since the sync above)
    const long long add_post = halo + 2 * c.postnet_layers * ((c.postnet_kernel - 1) / 2);
    e->h_act.resize((size_t)(5 + c.voc_stages) * B, 0);
    for (int b = 0; b < B; ++b) {
      const long long m32 = std::min<long long>(e->h_mel[b], 0x7fffffffLL);  // what duration_kernel wrote to mel32 (act_rows_kernel's input)
      e->h_act[b] = (int32_t)std::min<long long>(m32, T);
      e->h_act[(size_t)B + b] = (int32_t)std::min<long long>(m32 + add_post, T);
      sd += (double)std::min<long long>(e->h_mel[b], T);
      sp += (double)std::min<long long>(e->h_mel[b] + add_post, T);
      sv += (double)std::min<long long>(e->h_mel[b] + halo, T);
    }
    act_dec_h = e->h_act.data();
    act_post_h = e->h_act.data() + B;
    e->rag_frac_dec = sd / ((double)B * T);
    e->rag_frac_post = sp / ((double)B * T);
    e->rag_frac_voc = sv / ((double)B * T);
  }
  // (a Conformer decoder computes every row: its attention is unmasked and its depthwise convolution crosses into the padding)
  const int dec_heads = c.dec_n_head ? c.dec_n_head : c.n_head;
  if (c.block_type == 1) RET(conformer_stack(e, e->cf_dec, dec_heads, dx, ptr<float>(e->dxb), ml, B, (int)T, e->dec_precision == 1));
  else RET(fft_stack(e, e->dec, dec_heads, dx, ptr<float>(e->dxb), ml, B, (int)T, e->dec_precision == 1, act_dec, act_dec ? e->rag_frac_dec : 1.0, false, act_dec_h));
  // mel_linear (U/model.py:186)
  ConvParams p;
  auto setw = [&](ConvParams& q, const ConvW& w) {
    q.bias = w.b;
    if (e->dec_precision == 1 && w.wx3) { q.w = w.wx3; q.x3 = 1; }
    else { q.w = w.w; q.x3 = 0; }
  };
  p.B = B; p.T = (int)T; p.act_rows = act_post; p.act_rows_host = act_post_h; p.act_frac = act_post ? e->rag_frac_post : 1.0; p.in = dx; setw(p, e->mel_lin); p.out = ptr<float>(e->mel); p.Cin = H; p.Cout = c.n_mel;
  RET(conv(e, p));
  // Postnet (U/layers.py:556-563; BatchNorm folded at pack time) + residual (U/model.py:188); unmasked
  const float* pin = ptr<float>(e->mel);
  int cin = c.n_mel;
  float* bufs[2] = {ptr<float>(e->pn1), ptr<float>(e->pn2)};
  for (int i = 0; i < c.postnet_layers; ++i) {
    const bool last = i == c.postnet_layers - 1;
    p = ConvParams();
    p.B = B; p.T = (int)T; p.act_rows = act_post; p.act_rows_host = act_post_h; p.act_frac = act_post ? e->rag_frac_post : 1.0; p.in = pin; setw(p, e->postnet[i]); p.Cin = cin;
    p.Cout = last ? c.n_mel : c.postnet_dim; p.KW = c.postnet_kernel; p.pad = (c.postnet_kernel - 1) / 2;
    if (last) { p.out = ptr<float>(e->melpost); p.res = ptr<float>(e->mel); }
    else { p.out = bufs[i & 1]; p.act = ACT_TANH; }
    RET(conv(e, p));
    pin = p.out;
    cin = p.Cout;
  }
  e->last_B = B; e->last_L = L; e->last_T = (int)T;
  e->have_acoustic = true;
  return E2ETTS_OK;
}

// HifiGan.forward (V/generator.py:37-53) on channels-last mel [B, T, n_mel] already in HBM
int vocoder_impl(e2etts_engine* e, const float* mel_btc, int B, int T, bool want_wav, bool want_pcm, const int32_t* ragged_lens = nullptr,
                 const int64_t* ragged_lens_host = nullptr, DevBuf* out_wav = nullptr, DevBuf* out_pcm = nullptr) {
  // out_wav / out_pcm: the streaming path's slot buffers; the resident one-shot result (e->wav, e->pcm, have_wav) is then left alone
  const bool own_out = out_wav != nullptr;
  DevBuf& WAV = own_out ? *out_wav : e->wav;
  DevBuf& PCM = own_out ? *out_pcm : e->pcm;
  const auto& c = e->cfg;
  if (!e->voc_loaded) return e->fail(E2ETTS_ESTATE, "vocoder weights not loaded");
  if (B <= 0 || T <= 0) return e->fail(E2ETTS_EINVAL, "B and T must be positive");
  // largest activation of any stage, in floats per utterance
  long long len = T, ch = c.voc_init_ch;
  long long maxv = len * ch;
  for (int i = 0; i < c.voc_stages; ++i) {
    len *= c.voc_up_rate[i];
    ch /= 2;
    maxv = std::max(maxv, len * ch);
  }
  if (ch < 4 || (ch % 4)) return e->fail(E2ETTS_EINVAL, "final vocoder width %lld must be a positive multiple of 4", ch);
  const bool istft = c.voc_istft_nfft != 0;
  const long long nsamp = istft ? len * c.voc_istft_hop : len;
  if (nsamp != (long long)T * c.hop_length) return e->fail(E2ETTS_EINVAL, "upsample product != hop_length");
  if (istft) maxv += ch;  // the reflection-padded frame (the tail itself is always computed in full: it is 18 channels wide)
  const size_t vb = (size_t)B * maxv * 4;
  RET(ensure(e, e->v0, vb));
  RET(ensure(e, e->v1, vb));
  RET(ensure(e, e->v2, vb));
  RET(ensure(e, e->v3, vb));
  RET(ensure(e, WAV, (size_t)B * nsamp * 4));
  RET(ensure(e, PCM, (size_t)B * nsamp * 2));
  if (!own_out) e->have_wav = false;
  float *S = ptr<float>(e->v0), *XU = ptr<float>(e->v1), *T1 = ptr<float>(e->v2), *CUR = ptr<float>(e->v3);

  // weight selection: split-precision image when requested and present, else fp32
  auto setw = [&](ConvParams& q, const ConvW& w) {
    q.bias = w.b;
    if (e->voc_precision != E2ETTS_PRECISION_FP32 && w.wx3) { q.w = w.wx3; q.x3 = e->voc_precision; }
    else { q.w = w.w; q.x3 = 0; }
    if (q.x3 == E2ETTS_PRECISION_BF16) {   // plain bf16: the same weights in conv_bf16.hip's order (the router in conv() decides)
      auto it = e->bimg_of.find(w.wx3);
      if (it != e->bimg_of.end()) { q.bimg = it->second.first; q.bimg_tap_split = it->second.second; }
    }
  };
  // plain bf16: a whole ResBlock1 of stage `stage`, kernel index j, as ONE launch (conv_bf16.hip: rb_bf16) where that applies
  auto rb_params = [&](int stage, int j, int co, long long n, const float* x, float* out, RbParams& q) -> bool {
    if (e->voc_precision != E2ETTS_PRECISION_BF16 || c.voc_resblock != 1 || c.voc_n_dil > RB_MAX_PAIRS) return false;
    const int idx = stage * c.voc_n_kernels + j;
    q = RbParams();
    q.x = x; q.out = out; q.n_pairs = c.voc_n_dil; q.B = B; q.T = (int)n; q.C = co; q.KW = c.voc_rb_kernel[j];
    q.x_bs = q.out_bs = n * co; q.slope = 0.1f;
    for (int m = 0; m < c.voc_n_dil; ++m) {
      auto i1 = e->bimg_of.find(e->rb_c1[idx][m].wx3), i2 = e->bimg_of.find(e->rb_c2[idx][m].wx3);
      if (i1 == e->bimg_of.end() || i2 == e->bimg_of.end()) return false;
      q.bimg[m][0] = i1->second.first; q.bimg[m][1] = i2->second.first;
      q.b1[m] = e->rb_c1[idx][m].b; q.b2[m] = e->rb_c2[idx][m].b; q.dil[m] = c.voc_rb_dil[j][m];
    }
    return rb_bf16_supported(q);
  };
  // Ragged mode: the layers of stage i compute rows < mel_len * rate_i + halo_i only (host_logic.h: vocoder_stage_halo_rows -- the
  // reach of what is still to come, in that stage's rows: 12 frames after conv_pre, 76 / 109 / 94 / 63 rows in the four stages of
  // HiFi-GAN V1).  What lies beyond is stale but finite and out of the reach of every valid sample.
  const int32_t* act_stage[E2ETTS_MAX_STAGES + 1] = {nullptr};
  const int32_t* act_stage_h[E2ETTS_MAX_STAGES + 1] = {nullptr};  // the same limits in host memory (compact grids), when the caller has the lengths there
  double vfs[E2ETTS_MAX_STAGES + 1];  // fraction of the padded rows the limits of stage i leave (profile FLOP counts, tile choice without host lengths)
  for (int i = 0; i <= E2ETTS_MAX_STAGES; ++i) vfs[i] = ragged_lens ? e->rag_frac_voc : 1.0;
  if (ragged_lens) {
    RET(ensure(e, e->actbuf, (size_t)(5 + c.voc_stages) * B * 4));
    int32_t* ab = ptr<int32_t>(e->actbuf) + 2 * B;
    long long halo[E2ETTS_MAX_STAGES + 1];
    vocoder_stage_halo_rows(c, halo);
    if (ragged_lens_host) e->h_act.resize((size_t)(5 + c.voc_stages) * B, 0);
    long long rate = 1;
    for (int i = 0; i <= c.voc_stages; ++i) {
      KCHK(e, launch_act_rows(ragged_lens, ab + (size_t)i * B, B, 0, (int)rate, (long long)T * rate, e->stream, halo[i]));
      act_stage[i] = ab + (size_t)i * B;
      if (ragged_lens_host) {  // act_rows_kernel's arithmetic on the host copy of its input
        int32_t* h = e->h_act.data() + (size_t)(2 + i) * B;
        double sum = 0;
        for (int b = 0; b < B; ++b) {
          const long long m32 = std::min<long long>(ragged_lens_host[b], 0x7fffffffLL);
          h[b] = (int32_t)std::min<long long>(m32 * rate + halo[i], (long long)T * rate);
          sum += (double)std::max(h[b], 0);
        }
        act_stage_h[i] = h;
        vfs[i] = sum / ((double)B * T * rate);
      }
      if (i < c.voc_stages) rate *= c.voc_up_rate[i];
    }
  }
  // Small batches (the B = 1 latency path): the ResBlocks of a stage are independent until their sum (V/generator.py:44-48), and one
  // ResBlock's launches -- 384 tiles on 256 CUs at B = 1, chains of 40-70 us kernels -- leave CUs idle at every tail.  They run side by
  // side: ResBlock 0 on the engine's stream into S, ResBlock j > 0 on a side stream into a sum buffer of its own, joined by
  // S = (S + S_j) [/ num_kernels] in the accumulating epilogue's order -- same operations, same bits.  Off while the full class profile is taken
  // and above E2ETTS_VOC_CONC_FRAMES frames (default 2 048; 4 096 in exact fp32), where every launch fills the chip.
  // (measured per batch size and mode: bf16x3 gains up to B = 2 at T = 768 and loses 1-2 % from B = 4, exact fp32 still gains 2 % at
  // B = 4 and loses from B = 8; a per-stage limit on the tile count instead showed nothing beyond noise)
  static const long long conc_env = getenv("E2ETTS_VOC_CONC_FRAMES") ? atoll(getenv("E2ETTS_VOC_CONC_FRAMES")) : -1;
  const long long conc_frames = conc_env >= 0 ? conc_env : (e->voc_precision == E2ETTS_PRECISION_FP32 ? 4096 : 2048);
  const int nk = c.voc_n_kernels;
  // (a profile of ALL classes wants one kernel at a time; a profile filtered to one class -- bench.py's timed region -- records its
  // events on whichever stream the launch goes to and is fine)
  const bool full_profile = e->prof_on && e->prof_filter.empty();
  // Plain bf16 at such sizes goes one step further (conv_bf16.hip: launch_*_group): pair m of ALL ResBlocks of the stage is ONE launch on
  // the engine's stream -- a third of the launches, no fork / join events, and a grid the three kernel sizes fill together (one ResBlock's
  // pair is 1.15 rounds of workgroups at a 542-frame window, the k = 11 stream the critical path of every stage, and the host needed
  // longer to enqueue three streams' launches than the GPU to run them).  Each ResBlock keeps its own buffers as on the side streams,
  // and the join below is the same: same bits.  E2ETTS_VOC_GROUP=0 (tuning aid) keeps the streams.
  static const bool group_env = !(getenv("E2ETTS_VOC_GROUP") && atoi(getenv("E2ETTS_VOC_GROUP")) == 0);
  const bool small_window = nk > 1 && nk <= E2ETTS_MAX_RB_KERNELS && nk <= BC_GROUP_MAX && (long long)B * T <= conc_frames;
  const bool group_mode = group_env && small_window && e->voc_precision == E2ETTS_PRECISION_BF16 && !ragged_lens && c.voc_resblock == 1 && e->fuse_pairs;
  // A pass in a stream slot (e2etts_vocoder_stream_push) overlaps with the other slot's pass instead: side streams of its own would only
  // crowd the process's four hardware queues (E2ETTS_STREAM_SIDE=1, tuning aid, keeps them).
  static const bool side_in_slot = getenv("E2ETTS_STREAM_SIDE") && atoi(getenv("E2ETTS_STREAM_SIDE")) != 0;
  const bool conc = (!full_profile || group_mode) && nk > 1 && nk <= E2ETTS_MAX_RB_KERNELS && (long long)B * T <= conc_frames &&
                    (!own_out || side_in_slot || group_mode);
  if (conc) {
    for (int j = 0; j + 1 < nk; ++j) {
      if (!e->side[j]) HIPCHK(e, hipStreamCreateWithFlags(&e->side[j], hipStreamNonBlocking));
      if (!e->ev_join[j]) HIPCHK(e, hipEventCreateWithFlags(&e->ev_join[j], hipEventDisableTiming));
      for (int k = 0; k < 3; ++k) RET(ensure(e, e->vside[j][k], vb));
    }
    if (!e->ev_fork) HIPCHK(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  }
  hipStream_t const main_stream = e->stream;
  // A failure between a stage's fork and its join must not leave ResBlocks queued on the side streams: the next call could overwrite
  // XU or vside[*] -- or ensure() free them -- under work still in flight.  Whatever path leaves this function while
  // `armed`, the side streams are drained first.
  struct SideDrain {
    e2etts_engine* e; int n; bool armed;
    ~SideDrain() {
      if (!armed) return;
      for (int j = 0; j < n; ++j)
        if (e->side[j]) (void)hipStreamSynchronize(e->side[j]);
    }
  } drain{e, conc ? nk - 1 : 0, false};
  ConvParams p;
  p.B = B; p.T = T; p.act_rows = act_stage[0]; p.act_rows_host = act_stage_h[0]; p.act_frac = vfs[0]; p.in = mel_btc; setw(p, e->voc_pre); p.out = S; p.Cin = c.n_mel; p.Cout = c.voc_init_ch;
  p.KW = 7; p.pad = 3;
  RET(conv(e, p));
  long long n = T;
  ch = c.voc_init_ch;
  // stage i's upsampler on n_in rows of ch_in channels (input S)
  auto make_up = [&](int i, long long n_in, long long ch_in, ConvParams& q) {
    const int s = c.voc_up_rate[i];
    const int co = (int)ch_in / 2;
    q = ConvParams();
    q.B = B; q.T = (int)n_in; q.act_rows = act_stage[i]; q.act_rows_host = act_stage_h[i]; q.act_frac = vfs[i]; q.in = S; setw(q, e->voc_up[i]); q.out = XU; q.Cin = (int)ch_in; q.Cout = s * co;
    q.KW = 3; q.pad = 1; q.in_slope = 0.1f;
    q.zero_tap_split = s * co / 2;  // phases < s/2 never use tap 2, the others never tap 0 (packer.polyphase_upsampler)
  };
  // a join that the NEXT layer performs while staging its input (grouped stages, below): S = (((S + pend_add[0]) + ...) / pend_div
  const float* pend_add[3] = {nullptr, nullptr, nullptr};
  float pend_div = 1.0f;
  for (int i = 0; i < c.voc_stages; ++i) {
    const int s = c.voc_up_rate[i];
    const int co = (int)ch / 2;
    // leaky_relu(0.1) -> ConvTranspose1d(k = 2s, stride s, pad s/2)  (V/generator.py:40-41) as a 3-tap convolution
    // with s * co output channels: row q of the [n, s*co] result IS rows q*s .. q*s+s-1 of the [n*s, co] signal.
    make_up(i, n, ch, p);
    for (int k = 0; k < 3; ++k) p.in_add[k] = pend_add[k];   // the previous stage's join, folded into this layer's input (see below)
    p.in_div = pend_div;
    pend_add[0] = pend_add[1] = pend_add[2] = nullptr; pend_div = 1.0f;
    RET(conv(e, p, 2.0 / 3.0));
    n *= s;
    ch = co;
    if (n > 0x7fffffffLL / 2) return e->fail(E2ETTS_EINVAL, "utterance too long");
    // ---- grouped launches (see group_mode above): pair m of every ResBlock in one launch; at 256 channels, where a pair runs as two
    // convolutions with a bf16 hand-over, conv1 of every ResBlock and then conv2 of every ResBlock
    bool grouped_stage = false;
    bool joined_in_kernel = false;   // the stage's launch already wrote (S_0 + S_1 + ...) / n to S
    if (group_mode) {
      float* Sj[E2ETTS_MAX_RB_KERNELS]; float* T1j[E2ETTS_MAX_RB_KERNELS]; float* CURj[E2ETTS_MAX_RB_KERNELS];
      int order[E2ETTS_MAX_RB_KERNELS];   // dispatch order inside a launch: largest kernel size first
      for (int j = 0; j < nk; ++j) {
        Sj[j] = j ? ptr<float>(e->vside[j - 1][0]) : S; T1j[j] = j ? ptr<float>(e->vside[j - 1][1]) : T1; CURj[j] = j ? ptr<float>(e->vside[j - 1][2]) : CUR;
        order[j] = j;
      }
      std::sort(order, order + nk, [&](int a, int b2) { return c.voc_rb_kernel[a] > c.voc_rb_kernel[b2]; });
      bool as_pairs = co <= 128;
      if (co == 256) {   // fused pairs at 256 channels too where conv_bf16.hip has them (eight-wavefront workgroups)
        PairParams q;
        q.x = XU; q.out = Sj[0]; q.B = B; q.T = (int)n; q.C = co; q.KW = c.voc_rb_kernel[0]; q.dil = 1; q.x_bs = q.out_bs = (long long)n * co; q.mode = 2;
        const int idx0 = i * nk;
        auto i1 = e->bimg_of.find(e->rb_c1[idx0][0].wx3), i2 = e->bimg_of.find(e->rb_c2[idx0][0].wx3);
        if (i1 != e->bimg_of.end() && i2 != e->bimg_of.end()) { q.bimg1 = i1->second.first; q.bimg2 = i2->second.first; as_pairs = pair_bf16_supported(q); }
      }
      // 32 / 64 channels: the WHOLE ResBlock of every kernel size in one launch (these stages' tensors, 35 MB per 542-frame window at 48 kHz,
      // stream through the Infinity Cache: pair by pair each is read twice and written once per pair)
      if (e->fuse_pairs >= 2) {   // ... and, where one workgroup can hold it (32 channels), ALL the ResBlocks of the stage and their sum
        RbParams rq[E2ETTS_MAX_RB_KERNELS];
        bool st_ok = true;
        double fl = 0;
        for (int j = 0; j < nk && st_ok; ++j) {   // in j order: the kernel adds the ResBlocks' results as the join does, (S_0 + S_1) + S_2
          st_ok = rb_params(i, j, co, n, XU, S, rq[j]);
          if (st_ok) fl += rb_bf16_flops(rq[j]);
        }
        if (st_ok && rb_bf16_stage_supported(rq, nk)) {
          char nm[48];
          snprintf(nm, sizeof nm, "rb_bf16_stage_%d", co);
          ProfScope ps(e, nm, fl, 4.0 * 2.0 * B * (double)n * co);
          KCHK(e, launch_rb_bf16_stage(rq, nk, e->stream));
          grouped_stage = true;
          joined_in_kernel = true;
        }
      }
      if (!grouped_stage) {
        RbParams rq[E2ETTS_MAX_RB_KERNELS];
        bool rb_ok = e->fuse_pairs >= 2;
        double fl = 0, by = 0;
        for (int t = 0; t < nk && rb_ok; ++t) {
          rb_ok = rb_params(i, order[t], co, n, XU, Sj[order[t]], rq[t]);
          if (rb_ok) { fl += rb_bf16_flops(rq[t]); by += rb_bf16_bytes(rq[t]); }
        }
        if (rb_ok) {
          char nm[48];
          snprintf(nm, sizeof nm, "rb_bf16_group_%d", co);
          ProfScope ps(e, nm, fl, by);
          KCHK(e, launch_rb_bf16_group(rq, nk, e->stream));
          grouped_stage = true;
        }
      }
      // every member of every launch must be servable: decided before anything is launched
      bool ok = !grouped_stage;
      for (int j = 0; j < nk && ok; ++j)
        for (int m = 0; m < c.voc_n_dil && ok; ++m) {
          const int idx = i * nk + j;
          auto i1 = e->bimg_of.find(e->rb_c1[idx][m].wx3), i2 = e->bimg_of.find(e->rb_c2[idx][m].wx3);
          ok = i1 != e->bimg_of.end() && i2 != e->bimg_of.end();
          if (!ok) break;
          if (as_pairs) {
            PairParams q;
            q.x = XU; q.out = Sj[j]; q.b1 = e->rb_c1[idx][m].b; q.b2 = e->rb_c2[idx][m].b; q.bimg1 = i1->second.first; q.bimg2 = i2->second.first;
            q.B = B; q.T = (int)n; q.C = co; q.KW = c.voc_rb_kernel[j]; q.dil = c.voc_rb_dil[j][m]; q.x_bs = q.out_bs = (long long)n * co; q.mode = 2;
            ok = pair_bf16_supported(q);
          } else {
            BConvParams q;
            q.in = XU; q.wimg = i1->second.first; q.KWe = q.KW = c.voc_rb_kernel[j]; q.dil = c.voc_rb_dil[j][m]; q.pad = (q.KW * q.dil - q.dil) / 2;
            q.out_b = T1j[j]; q.B = B; q.T = (int)n; q.Cin = q.Cout = co;
            ok = conv_bf16_supported(q);
          }
        }
      if (ok) {
        grouped_stage = true;
        const float* curj[E2ETTS_MAX_RB_KERNELS];
        for (int j = 0; j < nk; ++j) curj[j] = XU;
        for (int m = 0; m < c.voc_n_dil; ++m) {
          const bool last = m == c.voc_n_dil - 1;
          if (as_pairs) {
            PairParams qs[E2ETTS_MAX_RB_KERNELS];
            double fl = 0, by = 0;
            for (int t = 0; t < nk; ++t) {
              const int j = order[t], idx = i * nk + j;
              PairParams& q = qs[t];
              q = PairParams();
              q.x = curj[j]; q.b1 = e->rb_c1[idx][m].b; q.b2 = e->rb_c2[idx][m].b;
              q.bimg1 = e->bimg_of[e->rb_c1[idx][m].wx3].first; q.bimg2 = e->bimg_of[e->rb_c2[idx][m].wx3].first;
              q.out = last ? Sj[j] : (curj[j] == CURj[j] ? T1j[j] : CURj[j]);   // x and out must differ: the running x ping-pongs CUR / T1
              q.B = B; q.T = (int)n; q.C = co; q.KW = c.voc_rb_kernel[j]; q.dil = c.voc_rb_dil[j][m];
              q.x_bs = q.out_bs = (long long)n * co; q.slope = 0.1f; q.mode = 2;
              fl += resblock_pair_flops(q); by += resblock_pair_bytes(q);
            }
            char nm[48];
            snprintf(nm, sizeof nm, "pair_bf16_group_%d", co);
            ProfScope ps(e, nm, fl, by);
            KCHK(e, launch_pair_bf16_group(qs, nk, e->stream));
            for (int t = 0; t < nk; ++t) curj[order[t]] = qs[t].out;
          } else {
            BConvParams q1[E2ETTS_MAX_RB_KERNELS], q2[E2ETTS_MAX_RB_KERNELS];
            double fl = 0, by = 0;
            for (int t = 0; t < nk; ++t) {
              const int j = order[t], idx = i * nk + j;
              const int k = c.voc_rb_kernel[j], d = c.voc_rb_dil[j][m];
              BConvParams& a = q1[t];   // xt = lrelu(c1(lrelu(x)) + b1), handed over as conv2's bf16 operand (V/layers.py:35-38)
              a = BConvParams();
              a.in = curj[j]; a.in_slope = 0.1f; a.wimg = e->bimg_of[e->rb_c1[idx][m].wx3].first; a.KWe = a.KW = k; a.dil = d; a.pad = (k * d - d) / 2;
              a.bias = e->rb_c1[idx][m].b; a.act_slope = 0.1f; a.out_b = T1j[j]; a.B = B; a.T = (int)n; a.Cin = a.Cout = co;
              BConvParams& z = q2[t];   // x = c2(xt) + x (:38-39)
              z = BConvParams();
              z.in = T1j[j]; z.in_bf16 = 1; z.wimg = e->bimg_of[e->rb_c2[idx][m].wx3].first; z.KWe = z.KW = k; z.dil = 1; z.pad = (k - 1) / 2;
              z.bias = e->rb_c2[idx][m].b; z.res = curj[j]; z.out = last ? Sj[j] : CURj[j]; z.B = B; z.T = (int)n; z.Cin = z.Cout = co;
              fl += 2.0 * B * (double)n * co * k * co; by += 4.0 * ((double)B * n * co * 2.0 + (double)co * k * co);
            }
            char nm[48];
            snprintf(nm, sizeof nm, "conv_bf16_group_%d", co);
            {
              ProfScope ps(e, nm, fl, by);
              KCHK(e, launch_conv_bf16_group(q1, nk, e->stream));
            }
            {
              ProfScope ps(e, nm, fl, by * 1.5);
              KCHK(e, launch_conv_bf16_group(q2, nk, e->stream));
            }
            for (int t = 0; t < nk; ++t) curj[order[t]] = q2[t].out;
          }
        }
      }
    }
    if (conc && !grouped_stage) {  // fork: the side streams may start once the upsampler's output (and everything before it) is complete
      HIPCHK(e, hipEventRecord(e->ev_fork, main_stream));
      drain.armed = true;
      for (int j = 0; j + 1 < nk; ++j) HIPCHK(e, hipStreamWaitEvent(e->side[j], e->ev_fork, 0));
    }
    float* const S_main = S;
    float* const T1_main = T1;
    float* const CUR_main = CUR;
    // Side by side, the ResBlocks are ENQUEUED last one first: at small windows (a 542-frame streaming window's 256-channel stage: six
    // launches of ~15 us per ResBlock) the host needs longer to enqueue a ResBlock's launches than the GPU to run them, so the stream
    // enqueued last starts late by the others' enqueue time -- and the last ResBlock has the largest kernel size, the stage's critical
    // path.  Which stream a ResBlock runs on, and every result bit, are unchanged.
    // Large grids, plain bf16, padded batch: the stage form of rb_bf16 (every ResBlock of the stage and their sum per workgroup: x read once,
    // the sum written once) where it exists (32 channels) -- there these stages run on the HBM roofline as pair launches.
    bool whole_stage = false;
    {
      static const int stage_env = getenv("E2ETTS_BRB_STAGE_BIG") ? atoi(getenv("E2ETTS_BRB_STAGE_BIG")) : 1;   // tuning aid
      if (stage_env && !grouped_stage && !conc && e->fuse_pairs >= 2 && !act_stage[i + 1] && nk <= BC_GROUP_MAX) {
        RbParams rq[E2ETTS_MAX_RB_KERNELS];
        bool st_ok = true;
        double fl = 0;
        for (int j = 0; j < nk && st_ok; ++j) {
          st_ok = rb_params(i, j, co, n, XU, S, rq[j]);
          if (st_ok) fl += rb_bf16_flops(rq[j]);
        }
        if (st_ok && rb_bf16_stage_supported(rq, nk)) {
          char nm[48];
          snprintf(nm, sizeof nm, "rb_bf16_stage_%d", co);
          ProfScope ps(e, nm, fl, 4.0 * 2.0 * B * (double)n * co);
          KCHK(e, launch_rb_bf16_stage(rq, nk, e->stream));
          whole_stage = true;
        }
      }
    }
    for (int jj = 0; jj < c.voc_n_kernels && !grouped_stage && !whole_stage; ++jj) {
      const int j = conc ? c.voc_n_kernels - 1 - jj : jj;
      const int idx = i * c.voc_n_kernels + j;
      const int k = c.voc_rb_kernel[j];
      // this ResBlock's stream and buffers (shadowing the stage-wide names below)
      const bool aside = conc && j > 0;
      float* const S = aside ? ptr<float>(e->vside[j - 1][0]) : S_main;
      float* const T1 = aside ? ptr<float>(e->vside[j - 1][1]) : T1_main;
      float* const CUR = aside ? ptr<float>(e->vside[j - 1][2]) : CUR_main;
      e->stream = aside ? e->side[j - 1] : main_stream;
      struct StreamGuard {  // whatever path leaves this iteration, the engine's stream is its own again
        e2etts_engine* e; hipStream_t s;
        ~StreamGuard() { e->stream = s; }
      } guard{e, main_stream};
      const float* cur = XU;
      const bool f32 = e->voc_precision == E2ETTS_PRECISION_FP32;
      bool fused = e->fuse_pairs && e->stage_fused[i] && (!f32 || !e->rb_pair_frag32[idx].empty());
      // 256 channels: the fused pair runs ONE 512-thread workgroup per CU on tiles of ~118-126 rows; with fewer tiles than CUs (small
      // batches: 52 tiles at B = 1, T = 768) the two-launch form on 64 x 64 tiles fills the chip better (B = 1: 5.99 vs 6.45 ms per
      // utterance).  Both forms give the same bits, so the choice is invisible in the output.
      if (fused && co == 256 && (long long)B * (n / 128) < 256) fused = false;
      // plain bf16, padded batch: the whole ResBlock in one launch on conv_bf16.hip's machinery, any kernel size at 32 / 64 channels
      bool whole_rb = false;
      // (at large grids the pair / chain kernels, two or three workgroups per CU, are ahead of this one-workgroup-per-CU kernel:
      // measured 7.9 against 8.4 ms for the 60 s utterance in one call; E2ETTS_BRB_ALWAYS=1 takes it there too -- same bits)
      static const bool rb_always = getenv("E2ETTS_BRB_ALWAYS") && atoi(getenv("E2ETTS_BRB_ALWAYS")) != 0;
      if (rb_always && fused && e->fuse_pairs >= 2 && !act_stage[i + 1]) {
        RbParams rq;
        if (rb_params(i, j, co, n, XU, S, rq)) {
          rq.accumulate = !conc && j > 0;
          if (j == c.voc_n_kernels - 1 && rq.accumulate) rq.out_div = (float)c.voc_n_kernels;
          char nm[48];
          snprintf(nm, sizeof nm, "rb_bf16_%d", co);
          ProfScope ps(e, nm, rb_bf16_flops(rq), rb_bf16_bytes(rq));
          KCHK(e, launch_rb_bf16(rq, e->stream));
          whole_rb = true;
          fused = false;   // nothing left for the pair / chain / two-launch forms below
        }
      }
      // the whole ResBlock in one launch (resblock_chain.hip) where it exists: kernel size 3 at 32 / 64 channels
      const bool chained = fused && !f32 && e->fuse_pairs >= 2 && resblock_chain_supported(co, k, c.voc_rb_dil[j], c.voc_n_dil);
      if (chained) {
        ChainParams q;
        q.x = XU; q.wfrag = e->rb_pair_frag[idx][0]; q.out = S;
        for (int m = 0; m < 3; ++m) { q.b1[m] = e->rb_c1[idx][m].b; q.b2[m] = e->rb_c2[idx][m].b; q.dil[m] = c.voc_rb_dil[j][m]; }
        q.act_rows = act_stage[i + 1]; q.act_rows_host = act_stage_h[i + 1]; q.act_frac = vfs[i + 1];
        q.B = B; q.T = (int)n; q.C = co; q.KW = k;
        q.x_bs = q.out_bs = (long long)n * co;
        q.slope = 0.1f; q.mode = e->voc_precision;
        q.accumulate = !conc && j > 0;
        if (j == c.voc_n_kernels - 1 && q.accumulate) q.out_div = (float)c.voc_n_kernels;
        char nm[48];
        snprintf(nm, sizeof nm, "resblock_chain_%d", co);
        ProfScope ps(e, nm, resblock_chain_flops(q), resblock_chain_bytes(q));
        KCHK(e, launch_resblock_chain(q, e->stream));
      }
      for (int m = 0; m < c.voc_n_dil && fused && !chained; ++m) {
        // the whole pair in one launch (resblock_pair.hip); x and out must differ, so the running x ping-pongs CUR / T1
        const bool last = m == c.voc_n_dil - 1;
        PairParams q;
        q.x = cur; q.wfrag = f32 ? e->rb_pair_frag32[idx][m] : e->rb_pair_frag[idx][m]; q.b1 = e->rb_c1[idx][m].b; q.b2 = e->rb_c2[idx][m].b;
        q.out = last ? S : (cur == CUR ? T1 : CUR);
        q.act_rows = act_stage[i + 1]; q.act_rows_host = act_stage_h[i + 1]; q.act_frac = vfs[i + 1];
        q.B = B; q.T = (int)n; q.C = co; q.KW = k; q.dil = c.voc_rb_dil[j][m];
        q.x_bs = q.out_bs = (long long)n * co;
        q.slope = 0.1f; q.mode = e->voc_precision;
        if (last) {
          q.accumulate = !conc && j > 0;
          if (j == c.voc_n_kernels - 1 && q.accumulate) q.out_div = (float)c.voc_n_kernels;
        }
        {
          static const bool fine = getenv("E2ETTS_PROFILE_FINE") != nullptr;
          char nm[48];
          if (fine) snprintf(nm, sizeof nm, "pair %d k%d d%d r%lld%s", co, k, q.dil, (long long)B * n, q.accumulate ? "+a" : "");
          else snprintf(nm, sizeof nm, f32 ? "resblock_pair_f32_%d" : "resblock_pair_%d", co);
          if (q.mode == E2ETTS_PRECISION_BF16) {   // plain bf16: the same pair on conv_bf16.hip's machinery where it applies (same bits)
            auto i1 = e->bimg_of.find(e->rb_c1[idx][m].wx3), i2 = e->bimg_of.find(e->rb_c2[idx][m].wx3);
            if (i1 != e->bimg_of.end() && i2 != e->bimg_of.end()) { q.bimg1 = i1->second.first; q.bimg2 = i2->second.first; }
          }
          // (at large grids the two kernel families are level pair by pair; the whole 60 s utterance in one call: 7.69 ms with pair_bf16
          // everywhere against 7.91 with resblock_pair.hip there, same box; E2ETTS_BPAIR_BIG=0 keeps the latter -- same bits)
          static const bool bpair_big = !(getenv("E2ETTS_BPAIR_BIG") && atoi(getenv("E2ETTS_BPAIR_BIG")) == 0);
          const bool bpair = (bpair_big || small_window) && pair_bf16_supported(q);
          if (bpair && !fine) snprintf(nm, sizeof nm, "pair_bf16_%d", co);
          ProfScope ps(e, nm, resblock_pair_flops(q), resblock_pair_bytes(q));
          KCHK(e, bpair ? launch_pair_bf16(q, e->stream) : launch_resblock_pair(q, e->stream));
        }
        cur = q.out;
      }
      for (int m = 0; m < c.voc_n_dil && c.voc_resblock == 2; ++m) {
        // ResBlock2 (V/layers.py:59-63): x = c(lrelu(x)) + x per dilation; the last one adds into the stage sum
        const bool last = m == c.voc_n_dil - 1;
        p = ConvParams();
        p.B = B; p.T = (int)n; p.act_rows = act_stage[i + 1]; p.act_rows_host = act_stage_h[i + 1]; p.act_frac = vfs[i + 1]; p.in = cur; setw(p, e->rb_c1[idx][m]); p.res = cur; p.Cin = co; p.Cout = co;
        p.KW = k; p.dil = c.voc_rb_dil[j][m]; p.pad = (k * p.dil - p.dil) / 2; p.in_slope = 0.1f;
        if (last) {
          p.out = S;
          p.accumulate = !conc && j > 0;
          if (j == c.voc_n_kernels - 1 && p.accumulate) p.out_div = (float)c.voc_n_kernels;
        } else {
          p.out = cur == CUR ? T1 : CUR;  // never in place: other workgroups still read the rows around this tile
        }
        RET(conv(e, p));
        cur = p.out;
      }
      for (int m = 0; m < c.voc_n_dil && !fused && !whole_rb && c.voc_resblock == 1; ++m) {
        const int d = c.voc_rb_dil[j][m];
        // xt = c1(lrelu(x)); the lrelu that feeds c2 is applied here, in c1's epilogue (V/layers.py:35-38)
        p = ConvParams();
        p.B = B; p.T = (int)n; p.act_rows = act_stage[i + 1]; p.act_rows_host = act_stage_h[i + 1]; p.act_frac = vfs[i + 1]; p.in = cur; setw(p, e->rb_c1[idx][m]); p.out = T1; p.Cin = co; p.Cout = co;
        p.KW = k; p.dil = d; p.pad = (k * d - d) / 2; p.in_slope = 0.1f; p.act = ACT_LRELU; p.act_slope = 0.1f;
        // plain bf16 on conv_bf16.hip: xt is only ever read as conv2's bf16 operand, so conv1 writes THAT (2 bytes per element, rounded
        // once, here) and conv2 copies it into its slab -- the values conv2's staging would have formed from an fp32 xt
        bool handover = false;
        {
          ConvParams p2 = ConvParams();
          p2.B = B; p2.T = (int)n; p2.in = T1; setw(p2, e->rb_c2[idx][m]); p2.res = cur; p2.out = CUR; p2.Cin = co; p2.Cout = co; p2.KW = k; p2.pad = (k - 1) / 2;
          p2.in_ld = p2.out_ld = p2.res_ld = co; p2.in_bs = p2.out_bs = p2.res_bs = (long long)n * co;
          ConvParams p1 = p; p1.in_ld = p1.out_ld = co; p1.in_bs = p1.out_bs = (long long)n * co;
          BConvParams q1, q2;
          handover = !p.act_rows && bconv_params(p1, q1) && bconv_params(p2, q2);
        }
        if (handover) { p.out = nullptr; p.out_b = T1; p.outb_slope = 1.0f; }
        RET(conv(e, p));
        // x = c2(xt) + x (:38-39); the last pair adds into the stage sum, and the last ResBlock divides by num_kernels
        // (V/generator.py:44-48)
        const bool last = m == c.voc_n_dil - 1;
        p = ConvParams();
        p.B = B; p.T = (int)n; p.act_rows = act_stage[i + 1]; p.act_rows_host = act_stage_h[i + 1]; p.act_frac = vfs[i + 1]; p.in = T1; setw(p, e->rb_c2[idx][m]); p.res = cur; p.Cin = co; p.Cout = co;
        p.in_bf16 = handover ? 1 : 0;
        p.KW = k; p.dil = 1; p.pad = (k - 1) / 2;
        if (last) {
          p.out = S;
          p.accumulate = !conc && j > 0;
          if (j == c.voc_n_kernels - 1 && !conc) p.out_div = (float)c.voc_n_kernels;
        } else {
          p.out = CUR;
        }
        RET(conv(e, p));
        cur = CUR;
      }
      if (aside) HIPCHK(e, hipEventRecord(e->ev_join[j - 1], e->stream));
    }
    // A grouped stage leaves its join to the layer that reads the sum -- the next upsampler (conv_bf16.hip stages (S_0 + S_1 + S_2) / 3 from
    // the three tensors) or conv_post -- where that layer can take it: accum_div's additions and division in its order (same bits), without
    // its launch and its pass over four tensors (142 MB at the 32- and 64-channel stages of a 542-frame window at 48 kHz).
    bool deferred = false;
    static const bool defer_env = !(getenv("E2ETTS_VOC_DEFER_JOIN") && atoi(getenv("E2ETTS_VOC_DEFER_JOIN")) == 0);   // tuning aid
    if (joined_in_kernel) deferred = true;   // nothing left to join
    if (conc && grouped_stage && !joined_in_kernel && defer_env && nk <= 4) {
      bool can = false;
      if (i + 1 < c.voc_stages) {
        ConvParams q;
        make_up(i + 1, n, ch, q);
        q.in_ld = q.Cin; q.out_ld = q.Cout; q.in_bs = (long long)q.T * q.Cin; q.out_bs = (long long)q.T * q.Cout;
        q.in_add[0] = ptr<float>(e->vside[0][0]);
        BConvParams bq;
        can = bconv_params(q, bq);
      } else {
        can = !istft;
      }
      if (can) {
        for (int j = 1; j < nk; ++j) pend_add[j - 1] = ptr<float>(e->vside[j - 1][0]);
        pend_div = (float)nk;
        deferred = true;
      }
    }
    if (conc && !deferred) {  // join: S = ((S_0 + S_1) + S_2 ...) / num_kernels, the accumulating epilogues' order
      for (int j = 1; j < nk && !grouped_stage; ++j) HIPCHK(e, hipStreamWaitEvent(main_stream, e->ev_join[j - 1], 0));
      for (int j = 1; j < nk; j += 2) {  // two side sums per pass: (S + S_j) + S_j+1, the same additions in the same order
        const bool two = j + 1 < nk;
        const bool closes = (two ? j + 1 : j) == nk - 1;
        KCHK(e, launch_accum_div(S_main, ptr<float>(e->vside[j - 1][0]), (long long)B * n * co, closes ? (float)nk : 1.0f, main_stream,
                                 two ? ptr<float>(e->vside[j][0]) : nullptr));
      }
      drain.armed = false;  // every side stream's work is now ordered before the main stream's next launch
    }
  }
  if (istft) {
    // x = leaky_relu(x) [slope 0.01]; ReflectionPad1d((1, 0)); conv_post; spec = exp(..), phase = sin(..) (V/generator.py:107-111);
    // wav = istft(spec * exp(j phase), n_fft, hop, window = hann) (src/tools/stft.py:138-148)
    const long long F = n + 1;
    const int pc = (c.voc_istft_nfft + 2 + 3) / 4 * 4, bins = c.voc_istft_nfft / 2 + 1;
    RET(ensure(e, e->istft_q, (size_t)B * F * pc * 4));
    RET(ensure(e, e->istft_ri, (size_t)B * F * bins * 8));
    RET(ensure(e, e->istft_sp, (size_t)B * F * bins * 2 * 4));
    {
      ProfScope ps(e, "misc", 0, 0);
      KCHK(e, launch_reflect_lrelu(S, XU, B, n, (int)ch, 0.01f, e->stream));
    }
    p = ConvParams();
    p.B = B; p.T = (int)F; p.in = XU; setw(p, e->voc_post); p.out = ptr<float>(e->istft_q); p.Cin = (int)ch; p.Cout = pc; p.KW = 7; p.pad = 3;
    RET(conv(e, p));
    {
      ProfScope ps(e, "istft", 0, (double)B * F * pc * 4.0 + (double)B * nsamp * 6.0);
      KCHK(e, launch_istft(ptr<float>(e->istft_q), pc, ptr<float>(e->istft_sp), ptr<float>(e->istft_ri), ptr<float>(WAV),
                           ptr<int16_t>(PCM), B, F, c.voc_istft_nfft, c.voc_istft_hop, e->stream));
    }
    e->istft_B = B;
    e->istft_F = F;
  } else {
    ProfScope ps(e, "conv_post", 2.0 * B * (double)n * 7 * ch, (double)B * n * (ch * 4.0 + 6.0));
    KCHK(e, launch_conv_post(S, e->voc_post.w, e->voc_post.b, ptr<float>(WAV), ptr<int16_t>(PCM), B, n, (int)ch, 7, e->stream,
                             act_stage[c.voc_stages], act_stage_h[c.voc_stages], pend_add[0] ? pend_add : nullptr, pend_div));
  }
  (void)want_wav; (void)want_pcm;
  if (!own_out) {
    e->voc_B = B; e->voc_T = T;
    e->have_wav = true;
  }
  return E2ETTS_OK;
}

}  // namespace

extern "C" {

// E2ETTS_SRC_HASH: sha256 (first 16 hex digits) of csrc/* + include/e2etts.h, passed by __graft_entry__.build(); the marker string
// "E2ETTS_SRC_HASH=" lets build() read it from the file without loading the library, so a stale shipped .so is rebuilt, not reused.
#ifndef E2ETTS_SRC_HASH
#define E2ETTS_SRC_HASH "unknown"
#endif
const char* e2etts_version(void) { return "e2etts-hip 0.4 (gfx950; fp32 MFMA + bf16x3 split-precision MFMA) E2ETTS_SRC_HASH=" E2ETTS_SRC_HASH; }

// The struct layout is part of the ABI: a change of its size must come with a new E2ETTS_ABI_VERSION (and a new row in the sizes
// e2etts_create accepts), never silently.
static_assert(sizeof(e2etts_config) == 320, "e2etts_config changed size: bump E2ETTS_ABI_VERSION and review e2etts_create's size check");
static_assert(offsetof(e2etts_config, struct_size) == 0, "struct_size must stay the first field");
int e2etts_abi_version(void) { return E2ETTS_ABI_VERSION; }
size_t e2etts_config_size(void) { return sizeof(e2etts_config); }

const char* e2etts_last_error(const e2etts_engine* engine) { return engine ? engine->err.c_str() : g_create_error.c_str(); }

int e2etts_create(int device_id, const e2etts_config* cfg, e2etts_engine** out) {
  if (!cfg || !out) { g_create_error = "cfg / out must not be NULL"; return E2ETTS_EINVAL; }
  *out = nullptr;
  // Nothing of *cfg is read before its size is known to be this header's: a host compiled against another revision of the struct
  // would otherwise be read out of bounds (or have its fields mistaken for others).
  if (cfg->struct_size != (uint32_t)sizeof(e2etts_config)) {
    char m[192];
    snprintf(m, sizeof m, "e2etts_config.struct_size = %u is not a size this library knows (sizeof(e2etts_config) = %zu, ABI version %d): "
             "the caller was built against another revision of e2etts.h", (unsigned)cfg->struct_size, sizeof(e2etts_config), E2ETTS_ABI_VERSION);
    g_create_error = m;
    return E2ETTS_EINVAL;
  }
  const e2etts_config& c = *cfg;
  auto bad = [&](const char* m) { g_create_error = m; return E2ETTS_EINVAL; };
  if (const char* m = config_check(c)) return bad(m);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device visible: the e2etts engine has no CPU fallback";
    return E2ETTS_EHIP;
  }
  if (device_id < 0 || device_id >= ndev) return bad("device_id out of range");
  if (hipSetDevice(device_id) != hipSuccess) { g_create_error = "hipSetDevice failed"; return E2ETTS_EHIP; }
  e2etts_engine* e = new e2etts_engine();
  e->cfg = c;
  e->device = device_id;
  if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc((void**)&e->h_mel, 4096 * sizeof(int64_t)) != hipSuccess) {
    g_create_error = "stream / pinned buffer creation failed";
    delete e;
    return E2ETTS_EHIP;
  }
  *out = e;
  return E2ETTS_OK;
}

void e2etts_destroy(e2etts_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipStreamSynchronize(e->stream);
  for (auto& sl : e->st_slot)
    if (sl.ctx.stream) (void)hipStreamSynchronize(sl.ctx.stream);
  for (DevBuf* b : e->owned)  // registered by ensure(): no hand-kept list to fall behind when a buffer is added
    if (b->p) (void)hipFree(b->p);
  for (auto& r : e->prof_recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
  for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
  free_frags(e);
  if (e->h_mel) (void)hipHostFree(e->h_mel);
  for (auto& sl : e->st_slot) {
    if (sl.pin) (void)hipHostFree(sl.pin);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.win_ready) (void)hipEventDestroy(sl.win_ready);
    for (auto& st : sl.ctx.side)
      if (st) (void)hipStreamDestroy(st);
    if (sl.ctx.ev_fork) (void)hipEventDestroy(sl.ctx.ev_fork);
    for (auto& ev : sl.ctx.ev_join)
      if (ev) (void)hipEventDestroy(ev);
    if (sl.ctx.stream) (void)hipStreamDestroy(sl.ctx.stream);
  }
  for (auto& st : e->side)
    if (st) (void)hipStreamDestroy(st);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  for (auto& ev : e->ev_join)
    if (ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

// Binds the blob resident in e->blob (nbytes bytes, copy possibly still in flight on the engine's stream): header and directory are
// read back from HBM, validated by host_logic.h, and the tensors located.
static int bind_resident_blob(e2etts_engine* e, size_t nbytes) {
  BlobHeader h;
  HIPCHK(e, hipMemcpyAsync(&h, e->blob.p, sizeof h, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (const char* m = blob_check_header(h, nbytes)) return e->fail(E2ETTS_EINVAL, "%s", m);
  std::vector<BlobEntry> dir(h.n_entries);
  if (h.n_entries)
    HIPCHK(e, hipMemcpy(dir.data(), (const char*)e->blob.p + sizeof h, (size_t)h.n_entries * sizeof(BlobEntry), hipMemcpyDeviceToHost));
  std::vector<BlobTensor> ts;
  std::string bad;
  if (const char* m = blob_check_directory(h, dir.data(), nbytes, ts, bad)) return e->fail(E2ETTS_EINVAL, "tensor '%s': %s", bad.c_str(), m);
  e->tensors.clear();
  free_frags(e);
  for (const BlobTensor& t : ts) e->tensors[t.name] = {reinterpret_cast<const float*>((const char*)e->blob.p + t.offset), t.numel};
  // a blob may carry the acoustic model, the vocoder, or both (the reference loads them from two checkpoints)
  const bool has_ac = e->tensors.count("enc.emb") != 0, has_voc = e->tensors.count("voc.pre.w") != 0;
  if (!has_ac && !has_voc) return e->fail(E2ETTS_EKEY, "blob holds neither acoustic ('enc.emb') nor vocoder ('voc.pre.w') tensors");
  if (has_ac) RET(bind_acoustic(e));
  if (has_voc) RET(bind_vocoder(e));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->ac_loaded = has_ac;
  e->voc_loaded = has_voc;
  return E2ETTS_OK;
}

// New weights replace the tensors the stream slots' kernels read: whatever they still have in flight finishes first, and an open stream
// is closed (its chunks were computed with the old weights; the next _push says so).
static int drain_stream_slots(e2etts_engine* e) {
  for (auto& sl : e->st_slot)
    if (sl.ctx.stream) HIPCHK(e, hipStreamSynchronize(sl.ctx.stream));
  e->st_open = false;
  e->st_pending = 0;
  return E2ETTS_OK;
}

int e2etts_load_weights(e2etts_engine* e, const void* blob, size_t nbytes) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!blob || nbytes < sizeof(BlobHeader)) return e->fail(E2ETTS_EINVAL, "weight blob too small");
  BlobHeader h;
  HIPCHK(e, hipMemcpy(&h, blob, sizeof h, hipMemcpyDefault));
  if (const char* m = blob_check_header(h, nbytes)) return e->fail(E2ETTS_EINVAL, "%s", m);  // before anything resident is touched
  RET(drain_stream_slots(e));
  e->ac_loaded = e->voc_loaded = false;
  RET(ensure(e, e->blob, nbytes));
  HIPCHK(e, hipMemcpyAsync(e->blob.p, blob, nbytes, hipMemcpyDefault, e->stream));
  return bind_resident_blob(e, nbytes);
}

// RCCL entry points, resolved at the first call: from the process image when the host already links or has loaded RCCL (the
// communicator handed in must come from THAT copy), else from librccl.so.1 (or the file named by E2ETTS_RCCL_LIB).
extern "C++" {
namespace {
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, int, void* /*ncclComm_t*/, hipStream_t);
typedef int (*nccl_rank_fn)(void*, int*);
typedef const char* (*nccl_errstr_fn)(int);
struct RcclApi {
  nccl_bcast_fn bcast = nullptr;
  nccl_rank_fn user_rank = nullptr;
  nccl_errstr_fn errstr = nullptr;
  std::string err;
};
RcclApi& rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    void* h = RTLD_DEFAULT;
    if (!dlsym(h, "ncclBroadcast")) {
      const char* path = getenv("E2ETTS_RCCL_LIB");
      h = dlopen(path ? path : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) {
        const char* why = dlerror();
        a.err = std::string("RCCL not loadable: ") + (why ? why : "?");
        return a;
      }
    }
    a.bcast = (nccl_bcast_fn)dlsym(h, "ncclBroadcast");
    a.user_rank = (nccl_rank_fn)dlsym(h, "ncclCommUserRank");
    a.errstr = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
    if (!a.bcast || !a.user_rank) a.err = "RCCL library lacks ncclBroadcast / ncclCommUserRank";
    return a;
  }();
  return api;
}
}  // namespace
}  // extern "C++"

int e2etts_load_weights_bcast(e2etts_engine* e, const void* blob_or_null, size_t nbytes, void* rccl_comm, int root) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!rccl_comm) return e->fail(E2ETTS_EINVAL, "rccl_comm must not be NULL");
  if (nbytes < sizeof(BlobHeader)) return e->fail(E2ETTS_EINVAL, "weight blob too small");
  RcclApi& api = rccl_api();
  if (!api.err.empty()) return e->fail(E2ETTS_ESTATE, "%s", api.err.c_str());
  int rank = -1;
  int rc = api.user_rank(rccl_comm, &rank);
  if (rc != 0) return e->fail(E2ETTS_EHIP, "ncclCommUserRank: %s", api.errstr ? api.errstr(rc) : "error");
  if (rank == root && !blob_or_null) return e->fail(E2ETTS_EINVAL, "the root rank must supply the blob");
  RET(drain_stream_slots(e));
  e->ac_loaded = e->voc_loaded = false;
  RET(ensure(e, e->blob, nbytes));
  if (rank == root) HIPCHK(e, hipMemcpyAsync(e->blob.p, blob_or_null, nbytes, hipMemcpyDefault, e->stream));
  rc = api.bcast(e->blob.p, e->blob.p, nbytes, 1 /* ncclUint8 */, root, rccl_comm, e->stream);  // in place, on the engine's stream
  if (rc != 0) return e->fail(E2ETTS_EHIP, "ncclBroadcast: %s", api.errstr ? api.errstr(rc) : "error");
  return bind_resident_blob(e, nbytes);
}

int e2etts_order_after(e2etts_engine* e, void* caller_stream) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  hipEvent_t ev = get_event(e);
  if (!ev) return e->fail(E2ETTS_EHIP, "hipEventCreate failed");
  HIPCHK(e, hipEventRecord(ev, (hipStream_t)caller_stream));
  HIPCHK(e, hipStreamWaitEvent(e->stream, ev, 0));  // the wait refers to the record above even if the event is re-recorded later
  e->ev_pool.push_back(ev);
  return E2ETTS_OK;
}

int e2etts_acoustic(e2etts_engine* e, const int64_t* ids, const int64_t* lens, int B, int L, const int64_t* speaker,
                    int n_spk_ids, float d_control, float p_control, float e_control, float* dur_out, int64_t* mel_lens_out,
                    int* T_out, int32_t* pitch_idx_out, int32_t* energy_idx_out, float* log_dur_out, float* pitch_pred_out,
                    float* energy_pred_out) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (B > 4096) return e->fail(E2ETTS_EINVAL, "B > 4096");
  RET(acoustic_impl(e, ids, lens, B, L, speaker, n_spk_ids, d_control, p_control, e_control));
  const size_t BL = (size_t)B * L;
  if (dur_out) RET(copy_out(e, dur_out, e->durf.p, BL * 4));
  if (mel_lens_out) RET(copy_out(e, mel_lens_out, e->mel64.p, (size_t)B * 8));
  if ((e->cfg.pitch_frame && (pitch_idx_out || pitch_pred_out)) || (e->cfg.energy_frame && (energy_idx_out || energy_pred_out)))
    return e->fail(E2ETTS_EINVAL, "a frame_level feature's index / prediction has T columns: pass NULL here and read it with e2etts_fetch_tap[_i32]");
  if (pitch_idx_out) RET(copy_out(e, pitch_idx_out, e->pidx.p, BL * 4));
  if (energy_idx_out) RET(copy_out(e, energy_idx_out, e->eidx.p, BL * 4));
  if (log_dur_out) RET(copy_out(e, log_dur_out, e->logd.p, BL * 4));
  if (pitch_pred_out) RET(copy_out(e, pitch_pred_out, e->ppred.p, BL * (e->cfg.pitch_no_uv ? 4 : 8)));
  if (energy_pred_out) RET(copy_out(e, energy_pred_out, e->epred.p, BL * 4));
  if (T_out) *T_out = e->last_T;
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_fetch_mel(e2etts_engine* e, float* mel, float* mel_post) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_acoustic) return e->fail(E2ETTS_ESTATE, "no acoustic result resident");
  const size_t n = (size_t)e->last_B * e->last_T * e->cfg.n_mel * 4;
  if (mel) RET(copy_out(e, mel, e->mel.p, n));
  if (mel_post) RET(copy_out(e, mel_post, e->melpost.p, n));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_fetch_tap(e2etts_engine* e, const char* which, float* out, size_t n_floats) {
  if (!e || !which || !out) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  const void* src = nullptr;
  size_t n = 0;
  if (!strcmp(which, "istft_spec_phase")) {
    if (!e->cfg.voc_istft_nfft || !e->have_wav) return e->fail(E2ETTS_ESTATE, "no iSTFT vocoder result resident");
    n = (size_t)e->istft_B * e->istft_F * (e->cfg.voc_istft_nfft + 2);
    if (n_floats != n) return e->fail(E2ETTS_EINVAL, "tap '%s' holds %zu floats, caller asked for %zu", which, n, n_floats);
    RET(copy_out(e, out, e->istft_sp.p, n * 4));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return E2ETTS_OK;
  }
  if (!e->have_acoustic) return e->fail(E2ETTS_ESTATE, "no acoustic result resident");
  const size_t np = (size_t)e->last_B * (e->cfg.pitch_frame ? e->last_T : e->last_L), ne = (size_t)e->last_B * (e->cfg.energy_frame ? e->last_T : e->last_L);
  if (!strcmp(which, "pitch_pred")) { src = e->ppred.p; n = np * (e->cfg.pitch_no_uv ? 1 : 2); }
  else if (!strcmp(which, "energy_pred")) { src = e->epred.p; n = ne; }
  else if (!strcmp(which, "enc_out")) { src = e->xa.p;  /* the encoder's output stays in its own buffer: nothing later in the pass writes it */ n = (size_t)e->last_B * e->last_L * e->cfg.hidden; }
  else if (!strcmp(which, "dec_out")) { src = e->dx.p; n = (size_t)e->last_B * e->last_T * e->cfg.hidden; }
  else return e->fail(E2ETTS_EKEY, "unknown tap '%s'", which);
  if (n_floats != n) return e->fail(E2ETTS_EINVAL, "tap '%s' holds %zu floats, caller asked for %zu", which, n, n_floats);
  RET(copy_out(e, out, src, n * 4));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_fetch_tap_i32(e2etts_engine* e, const char* which, int32_t* out, size_t n_values) {
  if (!e || !which || !out) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_acoustic) return e->fail(E2ETTS_ESTATE, "no acoustic result resident");
  const void* src = nullptr;
  size_t n = 0;
  if (!strcmp(which, "pitch_idx")) { src = e->pidx.p; n = (size_t)e->last_B * (e->cfg.pitch_frame ? e->last_T : e->last_L); }
  else if (!strcmp(which, "energy_idx")) { src = e->eidx.p; n = (size_t)e->last_B * (e->cfg.energy_frame ? e->last_T : e->last_L); }
  else return e->fail(E2ETTS_EKEY, "unknown tap '%s'", which);
  if (n_values != n) return e->fail(E2ETTS_EINVAL, "tap '%s' holds %zu values, caller asked for %zu", which, n, n_values);
  RET(copy_out(e, out, src, n * 4));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

static int vocoder_entry(e2etts_engine* e, const float* mel, bool channels_first, int B, int T, float* wav_out, int16_t* pcm_out) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  const float* src = nullptr;
  if (!mel) {
    if (!e->have_acoustic) return e->fail(E2ETTS_ESTATE, "mel == NULL but no acoustic result is resident");
    if (B != e->last_B || T != e->last_T) return e->fail(E2ETTS_EINVAL, "resident mel is [%d, %d, n_mel], caller said [%d, %d]", e->last_B, e->last_T, B, T);
    src = ptr<float>(e->melpost);
  } else {
    if (B <= 0 || T <= 0) return e->fail(E2ETTS_EINVAL, "B and T must be positive");
    const size_t n = (size_t)B * T * e->cfg.n_mel * 4;
    RET(ensure(e, e->melin, 2 * n));
    float* stage = ptr<float>(e->melin);
    float* btc = stage + (size_t)B * T * e->cfg.n_mel;
    if (channels_first) {
      RET(copy_in(e, stage, mel, n));
      KCHK(e, launch_transpose_bct_btc(stage, btc, B, e->cfg.n_mel, T, e->stream));
    } else {
      RET(copy_in(e, btc, mel, n));
    }
    src = btc;
  }
  RET(vocoder_impl(e, src, B, T, wav_out != nullptr, pcm_out != nullptr));
  const size_t ns = (size_t)B * T * e->cfg.hop_length;
  if (wav_out) RET(copy_out(e, wav_out, e->wav.p, ns * 4));
  if (pcm_out) RET(copy_out(e, pcm_out, e->pcm.p, ns * 2));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_vocoder(e2etts_engine* e, const float* mel_bct, int B, int T, float* wav_out, int16_t* pcm_out) {
  return vocoder_entry(e, mel_bct, true, B, T, wav_out, pcm_out);
}

int e2etts_vocoder_btc(e2etts_engine* e, const float* mel_btc, int B, int T, float* wav_out, int16_t* pcm_out) {
  return vocoder_entry(e, mel_btc, false, B, T, wav_out, pcm_out);
}

int e2etts_synthesize(e2etts_engine* e, const int64_t* ids, const int64_t* lens, int B, int L, const int64_t* speaker, int n_spk_ids,
                      float d_control, float p_control, float e_control, int16_t* pcm_out, size_t pcm_capacity,
                      int64_t* mel_lens_out, int* T_out) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (B > 4096) return e->fail(E2ETTS_EINVAL, "B > 4096");
  bool ragged = e->ragged != 0;
  RET(acoustic_impl(e, ids, lens, B, L, speaker, n_spk_ids, d_control, p_control, e_control, ragged));
  ragged = ragged && e->rag_short;
  if (T_out) *T_out = e->last_T;
  RET(vocoder_impl(e, ptr<float>(e->melpost), B, e->last_T, false, true, ragged ? ptr<int32_t>(e->mel32) : nullptr, ragged ? e->h_mel : nullptr));
  if (mel_lens_out) RET(copy_out(e, mel_lens_out, e->mel64.p, (size_t)B * 8));
  const size_t ns = (size_t)B * e->last_T * e->cfg.hop_length;
  if (pcm_out) {
    if (pcm_capacity < ns) {
      HIPCHK(e, hipStreamSynchronize(e->stream));
      return e->fail(E2ETTS_EINVAL, "pcm buffer holds %zu samples, result has %zu (fetch it with e2etts_fetch_pcm)", pcm_capacity, ns);
    }
    RET(copy_out(e, pcm_out, e->pcm.p, ns * 2));
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_fetch_pcm(e2etts_engine* e, int16_t* pcm_out, size_t capacity) {
  if (!e || !pcm_out) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_wav) return e->fail(E2ETTS_ESTATE, "no vocoder result resident");
  const size_t ns = (size_t)e->voc_B * e->voc_T * e->cfg.hop_length;
  if (capacity < ns) return e->fail(E2ETTS_EINVAL, "buffer holds %zu samples, result has %zu", capacity, ns);
  RET(copy_out(e, pcm_out, e->pcm.p, ns * 2));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_fetch_wav(e2etts_engine* e, float* wav_out, size_t capacity) {
  if (!e || !wav_out) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->have_wav) return e->fail(E2ETTS_ESTATE, "no vocoder result resident");
  const size_t ns = (size_t)e->voc_B * e->voc_T * e->cfg.hop_length;
  if (capacity < ns) return e->fail(E2ETTS_EINVAL, "buffer holds %zu samples, result has %zu", capacity, ns);
  RET(copy_out(e, wav_out, e->wav.p, ns * 4));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

// The engine works in ctx (stream, activations, side streams) while this object lives.
struct CtxSwap {
  e2etts_engine* e; e2etts_engine::VocCtx& c;
  void swap() {
    std::swap(e->stream, c.stream);
    std::swap(e->v0, c.v0); std::swap(e->v1, c.v1); std::swap(e->v2, c.v2); std::swap(e->v3, c.v3);
    for (int j = 0; j < E2ETTS_MAX_RB_KERNELS - 1; ++j) {
      for (int k = 0; k < 3; ++k) std::swap(e->vside[j][k], c.vside[j][k]);
      std::swap(e->side[j], c.side[j]);
      std::swap(e->ev_join[j], c.ev_join[j]);
    }
    std::swap(e->ev_fork, c.ev_fork);
  }
  CtxSwap(e2etts_engine* e_, e2etts_engine::VocCtx& c_) : e(e_), c(c_) { swap(); }
  ~CtxSwap() { swap(); }
};

int e2etts_vocoder_stream_begin(e2etts_engine* e, int B) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->voc_loaded) return e->fail(E2ETTS_ESTATE, "vocoder weights not loaded");
  if (B <= 0 || B > 4096) return e->fail(E2ETTS_EINVAL, "B out of range");
  for (auto& sl : e->st_slot) {
    if (!sl.done) HIPCHK(e, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    if (!sl.win_ready) HIPCHK(e, hipEventCreateWithFlags(&sl.win_ready, hipEventDisableTiming));
    if (!sl.ctx.stream) {
      HIPCHK(e, hipStreamCreateWithFlags(&sl.ctx.stream, hipStreamNonBlocking));
      // a slot's workspace is grown by ensure() while it is swapped into the engine, i.e. under the engine members' addresses: the
      // slot's own DevBufs are what e2etts_destroy must free
      DevBuf* mine[] = {&sl.ctx.v0, &sl.ctx.v1, &sl.ctx.v2, &sl.ctx.v3};
      for (DevBuf* b : mine) e->owned.push_back(b);
      for (auto& row : sl.ctx.vside)
        for (auto& b : row) e->owned.push_back(&b);
    }
    // chunks of an abandoned stream may still be in flight: nothing of theirs is delivered
    HIPCHK(e, hipStreamSynchronize(sl.ctx.stream));
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->st_head = 0;
  e->st_pending = 0;
  e->st_B = B;
  e->st_carry_n = 0;
  e->st_emitted = 0;
  e->st_halo = vocoder_halo_frames(e->cfg);
  e->st_open = true;
  e->st_done = false;
  return e->st_halo;
}

// Whether a caller's pointer is ordinary (pageable) host memory: such a chunk is copied into the slot's pinned buffer before _push
// returns, so the caller may reuse it at once; device memory and memory the caller pinned is read by the copy engine later.
static bool st_pageable_host(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return true;
  }
  return a.type == hipMemoryTypeUnregistered;
}

int e2etts_vocoder_stream_push(e2etts_engine* e, const float* mel_btc, int n, int last, int* n_frames_out) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->st_open || e->st_done) return e->fail(E2ETTS_ESTATE, "no open vocoder stream");
  if (n < 0 || (n > 0 && !mel_btc)) return e->fail(E2ETTS_EINVAL, "bad chunk");
  if (e->st_pending >= e2etts_engine::ST_DEPTH) return e->fail(E2ETTS_ESTATE, "%d chunks await e2etts_vocoder_stream_fetch", e2etts_engine::ST_DEPTH);
  const int B = e->st_B, H = e->st_halo, M = e->cfg.n_mel;
  const int total = e->st_carry_n + n;
  if (n_frames_out) *n_frames_out = 0;
  auto& sl = e->st_slot[(e->st_head + e->st_pending) % e2etts_engine::ST_DEPTH];   // free: whatever used it last has been fetched, hence has finished
  hipStream_t const cp = e->stream;   // assembly and copies: see the note at st_slot
  // window = [carry | new]; carry = up to H already-emitted frames (left context) followed by the frames not yet emitted
  const int left_ctx = (int)std::min<long long>(H, std::min<long long>(e->st_emitted, e->st_carry_n));
  const int emit_end = last ? total : total - H;   // frames before emit_end have their full right context in the window
  RET(ensure(e, sl.win, (size_t)B * std::max(total, 1) * M * 4));
  const size_t row_new = (size_t)n * M * 4, row_carry = (size_t)e->st_carry_n * M * 4, row_win = (size_t)total * M * 4;
  const void* src = mel_btc;
  if (n && st_pageable_host(mel_btc)) {   // taken now, so that the caller may reuse the chunk; sent up by the copy engine
    const size_t bytes = (size_t)B * row_new;
    if (sl.pin_cap < bytes) {
      if (sl.pin) HIPCHK(e, hipHostFree(sl.pin));
      sl.pin = nullptr;
      sl.pin_cap = 0;
      HIPCHK(e, hipHostMalloc(&sl.pin, bytes + bytes / 4, hipHostMallocDefault));
      sl.pin_cap = bytes + bytes / 4;
    }
    memcpy(sl.pin, mel_btc, bytes);
    src = sl.pin;
  }
  if (e->st_carry_n)
    HIPCHK(e, hipMemcpy2DAsync(sl.win.p, row_win, e->st_carry.p, row_carry, row_carry, B, hipMemcpyDeviceToDevice, cp));
  if (n)
    HIPCHK(e, hipMemcpy2DAsync((char*)sl.win.p + row_carry, row_win, src, row_new, row_new, B, hipMemcpyDefault, cp));
  int n_emit = emit_end - left_ctx;
  if (n_emit < 0) n_emit = 0;
  // next carry: H frames of emitted context + everything not yet emitted (it depends on the window alone, not on the pass)
  const int keep_from = std::max(0, (n_emit > 0 ? emit_end : left_ctx) - H);
  const int keep_n = last ? 0 : total - keep_from;
  if (keep_n > 0) {
    RET(ensure(e, e->st_carry, (size_t)B * keep_n * M * 4));
    HIPCHK(e, hipMemcpy2DAsync(e->st_carry.p, (size_t)keep_n * M * 4, (char*)sl.win.p + (size_t)keep_from * M * 4, row_win,
                               (size_t)keep_n * M * 4, B, hipMemcpyDeviceToDevice, cp));
  }
  e->st_carry_n = keep_n;
  if (last) e->st_done = true;
  if (n_emit > 0) {
    HIPCHK(e, hipEventRecord(sl.win_ready, cp));
    HIPCHK(e, hipStreamWaitEvent(sl.ctx.stream, sl.win_ready, 0));
    hipStream_t const cs = sl.ctx.stream;
    int rc;
    {
      CtxSwap in_slot(e, sl.ctx);
      rc = vocoder_impl(e, ptr<float>(sl.win), B, total, true, true, nullptr, nullptr, &sl.wav, &sl.pcm);
    }
    if (rc != E2ETTS_OK) {   // the carried context has moved on already: this stream cannot be continued
      e->st_open = false;
      (void)hipStreamSynchronize(cs);
      return rc;
    }
    HIPCHK(e, hipEventRecord(sl.done, cs));
    if (e->prof_on) HIPCHK(e, hipStreamSynchronize(cs));   // the profile's events are read after a wait on the engine's stream alone
    sl.win_n = total;
    sl.emit_off = left_ctx;
    sl.emit_n = n_emit;
    e->st_emitted += n_emit;
    ++e->st_pending;
  } else {
    HIPCHK(e, hipStreamSynchronize(cp));   // nothing to fetch: this call is the only place its errors can surface
  }
  if (n_frames_out) *n_frames_out = n_emit;
  return E2ETTS_OK;
}

int e2etts_vocoder_stream_fetch(e2etts_engine* e, float* wav_out, int16_t* pcm_out, size_t capacity) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!e->st_open || e->st_pending <= 0) return e->fail(E2ETTS_ESTATE, "no pushed chunk awaits a fetch (the last push emitted nothing?)");
  auto& sl = e->st_slot[e->st_head];
  const size_t hop = e->cfg.hop_length, ns = (size_t)sl.emit_n * hop;
  if (capacity < (size_t)e->st_B * ns) return e->fail(E2ETTS_EINVAL, "buffer holds %zu samples, chunk has %zu", capacity, (size_t)e->st_B * ns);
  const size_t src_row = (size_t)sl.win_n * hop, off = (size_t)sl.emit_off * hop;
  HIPCHK(e, hipStreamWaitEvent(e->stream, sl.done, 0));
  if (wav_out)
    HIPCHK(e, hipMemcpy2DAsync(wav_out, ns * 4, ptr<float>(sl.wav) + off, src_row * 4, ns * 4, e->st_B, hipMemcpyDefault, e->stream));
  if (pcm_out)
    HIPCHK(e, hipMemcpy2DAsync(pcm_out, ns * 2, ptr<int16_t>(sl.pcm) + off, src_row * 2, ns * 2, e->st_B, hipMemcpyDefault, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->st_head = (e->st_head + 1) % e2etts_engine::ST_DEPTH;
  --e->st_pending;
  return E2ETTS_OK;
}

int e2etts_tempo(e2etts_engine* e, const int16_t* pcm_in, size_t n_in, double speed, int sample_rate, int16_t* pcm_out, size_t capacity,
                 size_t* n_out) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  if (!pcm_in || !n_out) return e->fail(E2ETTS_EINVAL, "pcm_in / n_out must not be NULL");
  if (!(speed >= 0.25 && speed <= 4.0)) return e->fail(E2ETTS_EINVAL, "speed must lie in [0.25, 4]");
  if (sample_rate < 4000 || sample_rate > 96000) return e->fail(E2ETTS_EINVAL, "sample_rate must lie in [4000, 96000]");
  if (n_in == 0 || n_in > (size_t)1 << 30) return e->fail(E2ETTS_EINVAL, "n_in out of range");
  // the geometry of e2e_tts_amd/api.py: time_stretch_wsola: 40 ms frames (even), 50 % overlap, +-10 ms search
  const int n = std::max((int)(sample_rate * 40.0 / 1000.0) / 2 * 2, 64);
  const int hop_out = n / 2, delta = std::max((int)(sample_rate * 10.0 / 1000.0), 1);
  const double sp = speed;
  const int n_frames = std::max((int)std::ceil(((double)n_in / sp) / hop_out), 1);
  const long long want = std::llrint((double)n_in / sp);   // Python round(): half to even
  const long long avail = (long long)n_frames * hop_out + hop_out;   // what the frames cover; want <= n_frames * hop_out + 1 by the choice of n_frames
  const long long nout = std::min(want, avail);
  *n_out = (size_t)nout;
  if (!pcm_out) return E2ETTS_OK;  // size query
  if (capacity < (size_t)nout) return e->fail(E2ETTS_EINVAL, "output holds %zu samples, result has %lld", capacity, nout);
  RET(ensure(e, e->tempo_in, n_in * 2));
  RET(ensure(e, e->tempo_out, (size_t)std::max<long long>(nout, 1) * 2));
  RET(copy_in(e, e->tempo_in.p, pcm_in, n_in * 2));
  if (speed == 1.0) {
    RET(copy_out(e, pcm_out, e->tempo_in.p, n_in * 2));
  } else {
    KCHK(e, launch_wsola(ptr<int16_t>(e->tempo_in), (long long)n_in, ptr<int16_t>(e->tempo_out), nout, sp, n, delta, n_frames, e->stream));
    RET(copy_out(e, pcm_out, e->tempo_out.p, (size_t)nout * 2));
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}

int e2etts_set_precision(e2etts_engine* e, int vocoder_precision, int decoder_precision) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  if (vocoder_precision != E2ETTS_PRECISION_FP32 && vocoder_precision != E2ETTS_PRECISION_BF16X3 &&
      vocoder_precision != E2ETTS_PRECISION_BF16)
    return e->fail(E2ETTS_EINVAL, "unknown vocoder precision %d", vocoder_precision);
  if (decoder_precision != E2ETTS_PRECISION_FP32 && decoder_precision != E2ETTS_PRECISION_BF16X3)
    return e->fail(E2ETTS_EINVAL, "decoder precision must be fp32 or bf16x3 (got %d)", decoder_precision);
  e->voc_precision = vocoder_precision;
  e->dec_precision = decoder_precision;
  return E2ETTS_OK;
}

int e2etts_set_ragged(e2etts_engine* e, int enable) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  e->ragged = enable != 0;
  return E2ETTS_OK;
}

#ifdef E2ETTS_TEST_HOOKS
int e2etts_debug_poison_workspace(e2etts_engine* e) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  // what the workspaces held is gone: taps, fetches and vocoder(NULL) must say so instead of handing out the pattern
  e->have_acoustic = false;
  e->have_wav = false;
  DevBuf* bufs[] = {&e->xa, &e->xb, &e->xs, &e->xp, &e->tmp, &e->qkv, &e->att, &e->hid, &e->p1, &e->p2, &e->attws, &e->dx, &e->dxb, &e->mel,
                    &e->melpost, &e->pn1, &e->pn2, &e->melin, &e->v0, &e->v1, &e->v2, &e->v3, &e->wav, &e->pcm, &e->istft_q, &e->istft_ri, &e->istft_sp};
  for (DevBuf* b : bufs)
    if (b->p) HIPCHK(e, hipMemsetAsync(b->p, 0x4B, b->cap, e->stream));
  for (auto& row : e->vside)
    for (DevBuf& b : row)
      if (b.p) HIPCHK(e, hipMemsetAsync(b.p, 0x4B, b.cap, e->stream));
  // the stream slots' workspaces too (their outputs and windows only when nothing awaits a fetch)
  for (auto& sl : e->st_slot) {
    if (sl.ctx.stream) HIPCHK(e, hipStreamSynchronize(sl.ctx.stream));
    DevBuf* ws[] = {&sl.ctx.v0, &sl.ctx.v1, &sl.ctx.v2, &sl.ctx.v3};
    for (DevBuf* b : ws)
      if (b->p) HIPCHK(e, hipMemsetAsync(b->p, 0x4B, b->cap, e->stream));
    for (auto& row : sl.ctx.vside)
      for (DevBuf& b : row)
        if (b.p) HIPCHK(e, hipMemsetAsync(b.p, 0x4B, b.cap, e->stream));
    if (e->st_pending == 0) {
      DevBuf* outs[] = {&sl.wav, &sl.pcm, &sl.win};
      for (DevBuf* b : outs)
        if (b->p) HIPCHK(e, hipMemsetAsync(b->p, 0x4B, b->cap, e->stream));
    }
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return E2ETTS_OK;
}
#endif  // E2ETTS_TEST_HOOKS

int e2etts_set_fused_resblocks(e2etts_engine* e, int level) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  if (level < 0 || level > 2) return e->fail(E2ETTS_EINVAL, "fusion level must be 0 (off), 1 (pairs) or 2 (pairs + whole k = 3 ResBlocks)");
  e->fuse_pairs = level;
  return E2ETTS_OK;
}

int e2etts_profile_filter(e2etts_engine* e, const char* kernel_class) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  e->prof_filter = kernel_class ? kernel_class : "";
  return E2ETTS_OK;
}

int e2etts_profile_enable(e2etts_engine* e, int enable) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  RET(prof_collect(e));
  e->prof_stats.clear();
  e->prof_on = enable != 0;
  return E2ETTS_OK;
}

int e2etts_profile_read(e2etts_engine* e, e2etts_kernel_stat* out, int cap) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  RET(prof_collect(e));
  const int n = (int)e->prof_stats.size();
  for (int i = 0; i < n && i < cap && out; ++i) out[i] = e->prof_stats[i];
  return n;
}

size_t e2etts_device_bytes(const e2etts_engine* e) { return e ? e->dev_bytes : 0; }

void* e2etts_stream(e2etts_engine* e) { return e ? (void*)e->stream : nullptr; }

int e2etts_sync(e2etts_engine* e) {
  if (!e) return E2ETTS_EINVAL;
  std::lock_guard<std::mutex> lk(e->mu);
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (auto& sl : e->st_slot)
    if (sl.ctx.stream) HIPCHK(e, hipStreamSynchronize(sl.ctx.stream));
  return E2ETTS_OK;
}

}  // extern "C"
