/*
 * e2etts.h -- C ABI of the MI355X-native FastSpeech2 + HiFi-GAN inference engine.
 *
 * The reference (InterlinkLabs/e2e-tts) has no FFI / plugin interface: its hot path is a stack of
 * Python classes calling torch.  This header is the boundary this project puts
 * underneath Python mirrors of those classes; every entry point names the reference call it serves.
 * Paths are relative to the reference root; U/ = e2e_tts/models/acoustic/unsupervised_fastspeech2/,
 * V/ = e2e_tts/models/vocoder/, API/ = e2e_tts/src/api/.
 *
 * Conventions
 *  - plain C, no torch types; every function returns 0 on success or a negative E2ETTS_E* code, and
 *    e2etts_last_error() gives the message (no exceptions cross the boundary);
 *  - data pointers may be host OR device memory: copies use hipMemcpyDefault, so a caller that keeps
 *    its buffers in HBM (bench.py, the torch-tensor mirrors) pays no PCIe transfer;
 *  - STREAM ORDERING of device buffers: the engine works on its own non-blocking stream and every entry point
 *    returns only after that stream has drained, so results are complete on return.  What the engine cannot
 *    see is work the CALLER still has queued on a stream of its own that writes an input buffer (or still reads
 *    a buffer handed in as an output): call e2etts_order_after(engine, that_stream) first -- the engine's stream
 *    then waits for everything queued on that stream so far -- or synchronise that stream yourself.  The Python
 *    binding does this for every torch CUDA tensor argument (torch.cuda.current_stream());
 *  - one engine = one GPU + one HIP stream (at small batches the vocoder forks the ResBlocks of a stage onto internal side
 *    streams and joins them back into that stream before anything else reads their sum: invisible to the caller, same
 *    results bit for bit); calls on one engine are serialised by an internal mutex;
 *    distinct engines are independent (one process per GPU in multi-GPU runs);
 *  - all activations are fp32, channels-last ([B, N, C]); weights come packed by
 *    e2e_tts_amd/packer.py (weight-norm and BatchNorm folded, conv weights tap-major).
 */
#ifndef E2ETTS_H
#define E2ETTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#define E2ETTS_API __attribute__((visibility("default")))
#else
#define E2ETTS_API
#endif

/* ABI revision of this header: bumped whenever a struct layout or a signature changes.  A host checks e2etts_abi_version() ==
 * E2ETTS_ABI_VERSION (the Python binding does at load) and fills e2etts_config.struct_size, which e2etts_create checks. */
#define E2ETTS_ABI_VERSION 4

#define E2ETTS_MAX_STAGES 8
#define E2ETTS_MAX_RB_KERNELS 4
#define E2ETTS_MAX_DILATIONS 4

#define E2ETTS_OK 0
#define E2ETTS_EINVAL (-1)   /* bad argument / shape (the Python mirror raises ValueError) */
#define E2ETTS_EHIP (-2)     /* HIP runtime error */
#define E2ETTS_ESTATE (-3)   /* call order (e.g. vocoder before weights are loaded) */
#define E2ETTS_ENOMEM (-4)
#define E2ETTS_EKEY (-5)     /* tensor missing from the weight blob (KeyError) */

/* Dimensions parsed from the reference's config.yaml (API/utils.py:34; e2e_tts/config/model_config.yaml). */
typedef struct e2etts_config {
  uint32_t struct_size;   /* = sizeof(e2etts_config) of the header the CALLER was compiled against.  e2etts_create reads nothing beyond
                             it and refuses (E2ETTS_EINVAL) a size it does not know: a host built against an older or newer header gets an
                             error instead of garbage fields.  This revision knows exactly one size, e2etts_config_size(). */
  int32_t n_symbols;      /* len(symbols) = 131; the embedding has n_symbols + 1 rows (U/blocks/transformer.py:25) */
  int32_t n_speakers;
  int32_t n_mel;          /* audio.mel.channels */
  int32_t hidden;         /* encoder_hidden == decoder_hidden */
  int32_t enc_layers, dec_layers;
  int32_t n_head;
  int32_t ffn_dim;        /* conv_filter_size */
  int32_t ffn_k1, ffn_k2; /* conv_kernel_size */
  int32_t max_seq_len;
  int32_t dur_layers, dur_kernel, dur_chans;
  int32_t var_layers, var_kernel, var_chans; /* pitch predictor (and the energy predictor unless energy_layers / energy_kernel say otherwise); filter_size */
  int32_t n_bins;
  int32_t postnet_layers, postnet_dim, postnet_kernel;
  int32_t voc_init_ch;
  int32_t voc_stages;
  int32_t voc_up_rate[E2ETTS_MAX_STAGES];
  int32_t voc_up_kernel[E2ETTS_MAX_STAGES];
  int32_t voc_n_kernels;
  int32_t voc_rb_kernel[E2ETTS_MAX_RB_KERNELS];
  int32_t voc_n_dil;
  int32_t voc_rb_dil[E2ETTS_MAX_RB_KERNELS][E2ETTS_MAX_DILATIONS];
  int32_t hop_length;
  int32_t sample_rate;
  int32_t pos_table_rows; /* rows of the regenerated sinusoid table shipped in the blob */
  float f0_mean, f0_std;  /* stats.json "f0" (U/layers.py:152) */
  int32_t voc_resblock;   /* 1: ResBlock1 (V/layers.py:11-40), voc_n_dil pairs; 2: ResBlock2 (V/layers.py:49-66), voc_n_dil = 2 single convs */
  int32_t voc_istft_nfft; /* 0: HiFi-GAN tail (conv_post -> tanh, V/generator.py:49-51); else the iSTFTNet tail (V/generator.py:107-113 +
                             src/tools/stft.py:138-148): conv_post to n_fft + 2 channels, exp / sin, inverse STFT */
  int32_t voc_istft_hop;  /* iSTFT hop; prod(voc_up_rate) * voc_istft_hop == hop_length */
  int32_t block_type;     /* encoder / decoder block: 0 = FFT block (U/blocks/transformer.py:178-189); 1 = Conformer block
                             (U/blocks/conformer.py:171-255): n_head relative-position heads, ffn_dim = hidden x ffn_expansion_factor,
                             ffn_k1 = depthwise kernel size */
  int32_t energy_layers;  /* energy predictor depth / kernel (U/layers.py:92,96: ener_predictor_layers / ener_predictor_kernel); 0 = the pitch */
  int32_t energy_kernel;  /* predictor's (var_layers / var_kernel, U/layers.py:54,58), which is what the shipped model_config.yaml has */
  int32_t dec_n_head;     /* attention heads of the decoder's blocks (decoder_head, U/blocks/transformer.py:105, conformer.py:108); 0 = n_head, which then
                             is both encoder_head and decoder_head */
  int32_t pitch_no_uv;    /* variance_embedding.use_uv == False (U/layers.py:136,155-157): the pitch predictor has ONE output, the bucket is
                             torch.bucketize(prediction * p_control, pitch_bins) and the embedding table has pitch_emb_rows rows; 0 = use_uv */
  int32_t pitch_log2;     /* with use_uv: pitch_quantization == "log" (U/layers.py:148-149): f0 = 2 ** prediction instead of prediction * std + mean */
  int32_t pitch_emb_rows; /* rows of pitch_embedding (U/layers.py:60-63: n_bins with use_uv, f0_bins without); 0 = n_bins */
  int32_t pred_pad_left;  /* variance_predictor.ffn_padding != "SAME" (U/layers.py:400-402,479-481): the three predictors' convolutions are padded
                             (k - 1, 0) -- causal -- instead of ((k - 1) / 2, (k - 1) / 2); 0 = SAME */
  int32_t pitch_frame;    /* variance_embedding.pitch_feature == "frame_level" (U/layers.py:226-257): the pitch predictor and its embedding run on the */
  int32_t energy_frame;   /* length regulator's output ([B, T] rows) instead of on the phonemes; likewise energy_feature.  0 = phoneme_level.  The
                             frame-level feature's pitch_idx / pitch_pred / energy_idx / energy_pred then have T columns: e2etts_acoustic takes NULL
                             for them and e2etts_fetch_tap / e2etts_fetch_tap_i32 hand them out afterwards (T is only known then) */
} e2etts_config;

/* Threading: every entry point takes the engine's internal mutex, so single calls are safe from any thread and distinct engines are
 * independent.  Results that stay RESIDENT between calls (e2etts_acoustic -> e2etts_fetch_mel / e2etts_vocoder(NULL, ...),
 * e2etts_synthesize with pcm_out == NULL -> e2etts_fetch_pcm, taps, the vocoder stream) belong to the last such call: threads sharing one
 * engine serialise those SEQUENCES themselves (the Python binding holds Engine.lock around them). */
typedef struct e2etts_engine e2etts_engine;

/* Library / build identification. */
E2ETTS_API const char* e2etts_version(void);
/* E2ETTS_ABI_VERSION of the header the library was built from, and its sizeof(e2etts_config): what a binding compares its own mirror
 * of the struct with before it calls e2etts_create (e2e_tts_amd/_lib.py: load_library). */
E2ETTS_API int e2etts_abi_version(void);
E2ETTS_API size_t e2etts_config_size(void);

/* Last error message of this engine (or of a failed e2etts_create when engine == NULL). */
E2ETTS_API const char* e2etts_last_error(const e2etts_engine* engine);

/* Replaces: TTS.__init__ model construction + .to(device) (API/utils.py:41-56). */
E2ETTS_API int e2etts_create(int device_id, const e2etts_config* cfg, e2etts_engine** out);
E2ETTS_API void e2etts_destroy(e2etts_engine* engine);

/* Replaces: load_state_dict for both models (API/utils.py:48-49,54-55).  `blob` is the packed weight
 * image (host or device memory; a device image is what a RCCL broadcast from rank 0 leaves behind --
 * SURVEY.md 8(e)); it is copied into engine-owned HBM. */
E2ETTS_API int e2etts_load_weights(e2etts_engine* engine, const void* blob, size_t nbytes);

/* Multi-GPU start-up (no reference line: the reference is single-device): ONE RCCL broadcast of the packed
 * image from rank `root` of `rccl_comm` (an ncclComm_t of the RCCL copy the host process links or has loaded; every rank's
 * engine sits on that communicator's device), issued on the engine's stream straight into engine-owned HBM, then the same
 * binding as e2etts_load_weights.  Collective: every rank of the communicator calls it with the same nbytes; `blob_or_null`
 * (host or device memory) is read on the root only.  RCCL is resolved at the first call (process image, else librccl.so.1 or
 * $E2ETTS_RCCL_LIB); E2ETTS_ESTATE if it cannot be.  Hosts that drive RCCL through torch.distributed broadcast a tensor and
 * call e2etts_load_weights on it instead (e2e_tts_amd/dist.py) -- torch does not expose its ncclComm_t. */
E2ETTS_API int e2etts_load_weights_bcast(e2etts_engine* engine, const void* blob_or_null, size_t nbytes, void* rccl_comm, int root);

/* Replaces: UnsupervisedFastSpeech2.inference (U/model.py:155-194).
 *   ids  [B, L] int64, lens [B] int64 (1 <= lens[b] <= L), speaker [n_spk_ids] int64 with
 *   n_spk_ids == 1 (broadcast, as API/utils.py:133 does) or == B.
 * Outputs (each may be NULL): dur [B, L] fp32 (duration_rounded), mel_lens [B] int64, T_out = max mel
 * length, pitch_idx / energy_idx [B, L] int32 (the bucket indices of U/function.py:178-187 and
 * U/layers.py:169), log_dur [B, L], pitch_pred [B, L, 2] ([B, L] when pitch_no_uv), energy_pred [B, L].
 * mel / mel_post stay resident; read them with e2etts_fetch_mel.  One host sync (for T). */
E2ETTS_API int e2etts_acoustic(e2etts_engine* engine, const int64_t* ids, const int64_t* lens, int B, int L,
                    const int64_t* speaker, int n_spk_ids, float d_control, float p_control, float e_control,
                    float* dur_out, int64_t* mel_lens_out, int* T_out, int32_t* pitch_idx_out, int32_t* energy_idx_out,
                    float* log_dur_out, float* pitch_pred_out, float* energy_pred_out);

/* Copies of the resident results of the last e2etts_acoustic: mel, mel_post [B, T, n_mel] (either may be NULL). */
E2ETTS_API int e2etts_fetch_mel(e2etts_engine* engine, float* mel, float* mel_post);

/* Debug / parity taps of the last e2etts_acoustic: which = "enc_out" [B, L, H] | "dec_out" [B, T, H]. */
/* taps: "enc_out" [B, L, hidden], "dec_out" [B, T, hidden] (after e2etts_acoustic); "istft_spec_phase" [B, T * prod(voc_up_rate) + 1,
 * n_fft + 2] = exp / sin heads of the iSTFTNet generator (reference iSTFT.forward's return values), after a vocoder call. */
E2ETTS_API int e2etts_fetch_tap(e2etts_engine* engine, const char* which, float* out, size_t n_floats);
/* more taps of the last e2etts_acoustic, at the level the feature lives at (N = L for phoneme_level, T for frame_level):
 * e2etts_fetch_tap: "pitch_pred" [B, N, 2] ([B, N] when pitch_no_uv), "energy_pred" [B, N]; e2etts_fetch_tap_i32: "pitch_idx", "energy_idx" [B, N]. */
E2ETTS_API int e2etts_fetch_tap_i32(e2etts_engine* engine, const char* which, int32_t* out, size_t n_values);

/* Replaces: HifiGan.forward (V/generator.py:37-53) on mel [B, n_mel, T] (the reference layout,
 * channels-first, as API/utils.py:144 passes it) or, when mel == NULL, on the resident mel_post of the
 * last e2etts_acoustic.  Outputs (each may be NULL): wav [B, T*hop] fp32 in (-1, 1);
 * pcm [B, T*hop] int16 = trunc(wav * 32768) as TTS.combine_audio computes it (API/utils.py:111-117). */
E2ETTS_API int e2etts_vocoder(e2etts_engine* engine, const float* mel_bct, int B, int T, float* wav_out, int16_t* pcm_out);

/* Same, mel given channels-last [B, T, n_mel] (the engine's native layout; no transpose). */
E2ETTS_API int e2etts_vocoder_btc(e2etts_engine* engine, const float* mel_btc, int B, int T, float* wav_out, int16_t* pcm_out);

/* Replaces one iteration of the TTS.inference batch loop (API/utils.py:130-148): acoustic -> vocoder.
 * pcm_out [B, T*hop] int16 (rows padded to the batch maximum; valid samples of row b = mel_lens[b]*hop),
 * mel_lens_out [B]. T_out receives T.  pcm_capacity = number of int16 the caller's buffer holds; if it is
 * too small E2ETTS_EINVAL is returned with T_out set, and the result can be fetched with e2etts_fetch_pcm. */
E2ETTS_API int e2etts_synthesize(e2etts_engine* engine, const int64_t* ids, const int64_t* lens, int B, int L,
                      const int64_t* speaker, int n_spk_ids, float d_control, float p_control, float e_control,
                      int16_t* pcm_out, size_t pcm_capacity, int64_t* mel_lens_out, int* T_out);
E2ETTS_API int e2etts_fetch_pcm(e2etts_engine* engine, int16_t* pcm_out, size_t capacity);
E2ETTS_API int e2etts_fetch_wav(e2etts_engine* engine, float* wav_out, size_t capacity);

/* Long-form / streaming vocoder (BASELINE config 5).  The mel stream of B parallel utterances is pushed in chunks of any
 * size, channels-last [B, n, n_mel] (host or device memory); the engine keeps the receptive-field halo (returned by
 * _begin, in frames; 15 for the default HiFi-GAN V1) and the not-yet-emittable tail internally, so HBM use is bounded by
 * the chunk size, and the concatenated output is bit-identical to one e2etts_vocoder call on the whole mel.
 *   _begin(B)                      -> halo in frames (>= 0) or a negative error
 *   _push(mel, n, last, &n_emit)   -> enqueues the vocoder on [context | pending | new] and RETURNS; n_emit frames became final
 *   _fetch(wav, pcm, capacity)     -> waits for the OLDEST unfetched push that emitted frames and copies its n_emit * hop samples per
 *                                     utterance, [B, n_emit * hop] compact
 * Up to two emitting pushes may await their fetch (a third fails with E2ETTS_ESTATE), so a caller either alternates push / fetch
 * as before, or keeps the GPU busy across chunk boundaries: push(i + 1), then fetch(i) -- chunk i's samples travel to the host on a
 * copy stream while chunk i + 1 computes.  Each in-flight chunk has its own output buffers; the resident one-shot result
 * (e2etts_fetch_wav / _pcm) is not touched by the stream.  A chunk in pageable host memory has been consumed when _push returns; one in
 * device memory, or in host memory the caller pinned, must stay valid until the fetch of that push (or e2etts_sync) returns.  Errors of
 * the enqueued work surface at the fetch. */
E2ETTS_API int e2etts_vocoder_stream_begin(e2etts_engine* engine, int B);
E2ETTS_API int e2etts_vocoder_stream_push(e2etts_engine* engine, const float* mel_btc, int n_frames, int last, int* n_frames_out);
E2ETTS_API int e2etts_vocoder_stream_fetch(e2etts_engine* engine, float* wav_out, int16_t* pcm_out, size_t capacity);

/* Replaces: audio_speed_change (API/utils.py:163-172), which shells out to ffmpeg's `atempo` filter: tempo change without pitch change
 * of an int16 PCM signal (host or device memory), on the GPU.  ffmpeg is not part of this build: the kernel is waveform-similarity
 * overlap-add (WSOLA), the algorithm family behind that filter, restated from the published method -- PARITY UNPINNED against the
 * reference; it mirrors e2e_tts_amd/api.py: time_stretch_wsola (40 ms frames, 50 % overlap, +-10 ms search, float64) step for step.
 * n_out receives round(n_in / speed); with pcm_out == NULL the call only reports that size.  speed in [0.25, 4].  (The Python mirror's
 * DEFAULT for speed != 1 is the model's own duration control -- no post-processing at all; this entry serves speed_mode="wsola".) */
/* `speed` is a double: n_out and the per-frame analysis positions are computed from it exactly as the Python mirror computes them from
 * its float (a C float 0.8f is 0.800000012, which rounds n_in / speed the other way at exact .5 ties).  sample_rate in [4000, 96000]: the
 * kernel keeps one analysis frame, its search region and three work frames in LDS (0.18 s of float64 samples <= 160 KB). */
E2ETTS_API int e2etts_tempo(e2etts_engine* engine, const int16_t* pcm_in, size_t n_in, double speed, int sample_rate, int16_t* pcm_out,
                 size_t capacity, size_t* n_out);

/* Arithmetic of the convolutions / projections of (a) the vocoder and (b) the decoder + mel_linear + postnet.
 * The encoder and the variance adaptor are always exact fp32: the duration / pitch / energy decisions taken there
 * must be bit-exact, and nothing downstream of the length regulator is discrete.
 * E2ETTS_PRECISION_FP32 (default -- the reference's arithmetic): v_mfma_f32_32x32x2_f32, an exact fp32 FMA chain.
 * E2ETTS_PRECISION_BF16X3 (opt-in fast mode, 2.6 x the throughput): every fp32 operand is split into bf16 hi + lo and the
 * product keeps hi*hi + hi*lo + lo*hi on the bf16 matrix pipe with fp32 accumulation -- ~16 significant bits per operand;
 * measured against the reference's own outputs (bench.py `split_precision_mode`): wav mean-L1 1.1e-6 .. 1.3e-6 (fp32: 1.3e-7;
 * plain bf16: 5e-4; parity bar: 1e-4), int16 PCM within 1 LSB on 100 % of samples, discrete outputs untouched. */
#define E2ETTS_PRECISION_FP32 0
#define E2ETTS_PRECISION_BF16X3 1
#define E2ETTS_PRECISION_BF16 2 /* vocoder only: hi x hi product alone (plain bf16 operands, fp32 accumulation): the arithmetic
                                   BASELINE config 5 (long-form streaming) names; waveform error ~5e-4, above the fp32 bar */
E2ETTS_API int e2etts_set_precision(e2etts_engine* engine, int vocoder_precision, int decoder_precision);

/* Ragged batches (default on).  e2etts_synthesize hands back, per utterance, only mel_lens[b] * hop valid samples; with
 * ragged != 0 it therefore skips, layer by layer, the rows of shorter utterances that no valid sample depends on (rows past
 * mel_len + the receptive field of the layers still to come, stage by stage), instead of computing the whole padded batch as the reference does.  Valid samples are
 * bit-identical either way; what lies beyond them in the padded PCM rows is then unspecified.  e2etts_acoustic /
 * e2etts_vocoder always compute the full padded tensors (their padded rows match the reference's).
 * The frame level (decoder, postnet, vocoder) is skipped from the mel lengths the engine computes itself; the phoneme level (encoder,
 * predictors) additionally when `lens` is HOST memory (the launch grids are built from the lengths on the host; with `lens` in device
 * memory the phoneme level computes the padded batch -- same results).  Batches of up to 64 utterances launch grids without idle
 * workgroups; larger ones still skip the rows but keep the padded grid. */
E2ETTS_API int e2etts_set_ragged(e2etts_engine* engine, int enable);

/* Test hook for ragged mode, present only in the TEST build of the library (compiled with -DE2ETTS_TEST_HOOKS into
 * libe2etts_hip_test.so; the product library libe2etts_hip.so does not export it): overwrite the activation workspaces (not the
 * weights, not the index buffers) with a large finite pattern (every byte 0x4B: 1.3e7 as fp32) and forget the resident results, so
 * that a test can show that no valid sample of the next call depends on what an earlier call left in the rows that ragged compute
 * skips. */
#ifdef E2ETTS_TEST_HOOKS
E2ETTS_API int e2etts_debug_poison_workspace(e2etts_engine* engine);
#endif

/* Fused ResBlocks (bf16 modes).  level 1: each (conv k, dilation d -> leaky ReLU -> conv k -> + x) pair of HiFi-GAN's ResBlock1
 * (reference V/layers.py:33-40) at 32 / 64 / 128 / 256 channels runs as ONE kernel whose intermediate stays in LDS.  level 2
 * (default): additionally a whole kernel-size-3 ResBlock1 (its three pairs) at 32 / 64 channels runs as one kernel whose residual
 * stream stays in registers.  level 0: two convolution launches per pair.  All three produce bit-identical output. */
E2ETTS_API int e2etts_set_fused_resblocks(e2etts_engine* engine, int level);

/* Per-kernel-class timing with HIP events on the engine's stream (bench.py roofline leg).
 * enable != 0 starts recording (and clears counters); e2etts_profile_read fills up to `cap` records. */
typedef struct e2etts_kernel_stat {
  char name[48];
  uint64_t launches;
  double ms;      /* sum of launch durations */
  double flops;   /* algorithmic FLOPs (2 x MAC) of those launches */
  double bytes;   /* algorithmic bytes (operands read once + result written once) */
} e2etts_kernel_stat;
/* Restrict the event bracketing to one kernel class (NULL or "": all classes).  Two hipEventRecord per launch cost ~8 us; with ~155
 * launches per step that is 2 % of a step, so a timed region that only needs the dominant kernel's duration brackets that class alone. */
E2ETTS_API int e2etts_profile_filter(e2etts_engine* engine, const char* kernel_class);
E2ETTS_API int e2etts_profile_enable(e2etts_engine* engine, int enable);
E2ETTS_API int e2etts_profile_read(e2etts_engine* engine, e2etts_kernel_stat* out, int cap);

/* Bytes of HBM currently owned by the engine (weights + workspace). */
E2ETTS_API size_t e2etts_device_bytes(const e2etts_engine* engine);

/* The engine's stream as a hipStream_t cast to void* (so a caller can order its own work after ours). */
E2ETTS_API void* e2etts_stream(e2etts_engine* engine);
/* Orders everything the engine does from now on after the work queued so far on `caller_stream` (a hipStream_t cast to
 * void*; NULL = the legacy default stream): an event recorded there, waited for on the engine's stream.  See "STREAM
 * ORDERING" at the top.  Costs two API calls, no host synchronisation. */
E2ETTS_API int e2etts_order_after(e2etts_engine* engine, void* caller_stream);
E2ETTS_API int e2etts_sync(e2etts_engine* engine);

#ifdef __cplusplus
}
#endif
#endif /* E2ETTS_H */
