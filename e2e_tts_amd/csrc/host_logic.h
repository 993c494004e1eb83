// Host-side logic of the engine that touches no HIP API: weight-blob header / directory validation, engine-config validation,
// the vocoder's receptive field, and the tile choice of conv_gemm.  Kept free of <hip/...> so that tests/csrc/host_logic_test.cc can
// build it with gcc -fsanitize=address,undefined and feed it truncated / corrupt blobs (tests/test_host_sanitizer.py); engine.hip and
// conv_gemm.hip include the same header, so what the sanitizer run exercises is the code the library ships.
#pragma once
#include <stdint.h>
#include <string.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/e2etts.h"

namespace e2etts {

// Blob layout (e2e_tts_amd/packer.py: build_blob), little endian: 32-byte header, n_entries x 80-byte directory entries, then the
// fp32 tensors, each at a 256-byte aligned offset.
struct BlobHeader {
  char magic[8];
  uint32_t version;
  uint32_t n_entries;
  uint64_t data_offset;
  uint64_t total_bytes;
};
struct BlobEntry {
  char name[64];
  uint64_t offset;
  uint64_t numel;
};
static_assert(sizeof(BlobHeader) == 32 && sizeof(BlobEntry) == 80, "blob layout is fixed by packer.py");

struct BlobTensor {
  std::string name;
  uint64_t offset = 0, numel = 0;
};

// nullptr when the header is plausible for a blob of `nbytes` bytes, else a static message
inline const char* blob_check_header(const BlobHeader& h, size_t nbytes) {
  if (nbytes < sizeof(BlobHeader)) return "weight blob too small";
  if (memcmp(h.magic, "E2ETTSW1", 8) != 0 || h.version != 1) return "not an e2etts weight blob (bad magic / version)";
  if (h.total_bytes != (uint64_t)nbytes) return "blob size differs from the header's total";
  if (h.n_entries > 100000) return "corrupt blob directory (entry count)";
  const uint64_t dir_end = sizeof(BlobHeader) + (uint64_t)h.n_entries * sizeof(BlobEntry);
  if (dir_end > h.data_offset || h.data_offset > (uint64_t)nbytes || (h.data_offset & 255)) return "corrupt blob directory (data offset)";
  return nullptr;
}

// Validates the n_entries directory records (`dir` holds exactly h.n_entries of them) against the blob size and appends them to `out`.
// Every arithmetic step is overflow-checked: numel comes from the file.
inline const char* blob_check_directory(const BlobHeader& h, const BlobEntry* dir, size_t nbytes, std::vector<BlobTensor>& out, std::string& bad_name) {
  out.clear();
  out.reserve(h.n_entries);
  for (uint32_t i = 0; i < h.n_entries; ++i) {
    BlobEntry en = dir[i];
    en.name[sizeof en.name - 1] = 0;
    bad_name = en.name;
    if ((en.offset & 255) || en.offset < h.data_offset || en.offset > (uint64_t)nbytes) return "tensor lies outside the blob";
    if (en.numel > ((uint64_t)nbytes - en.offset) / 4) return "tensor lies outside the blob";
    BlobTensor t;
    t.name = en.name;
    t.offset = en.offset;
    t.numel = en.numel;
    out.push_back(t);
  }
  bad_name.clear();
  return nullptr;
}

// e2etts_create's argument validation (nullptr = valid)
inline const char* config_check(const e2etts_config& c) {
  // ranges first: every later expression (2 * rate, 4 << stages, hidden / n_head ...) then stays far inside int32
  auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  const int BIG = 1 << 20;
  if (!in(c.hidden, 4, BIG) || !in(c.n_head, 1, BIG) || c.hidden % 4 || c.hidden % c.n_head) return "hidden must be a positive multiple of 4 and of n_head";
  if (!in(c.n_symbols, 1, BIG) || !in(c.n_speakers, 1, BIG)) return "n_symbols and n_speakers must be positive";
  if (!in(c.enc_layers, 0, 64) || !in(c.dec_layers, 0, 64)) return "layer counts out of range";
  if (!in(c.n_mel, 4, BIG) || c.n_mel % 4) return "n_mel must be a positive multiple of 4";
  if (!in(c.ffn_dim, 4, BIG) || !in(c.dur_chans, 4, BIG) || !in(c.var_chans, 4, BIG) || !in(c.postnet_dim, 4, BIG)) return "channel counts must be positive";
  if (c.ffn_dim % 4 || c.dur_chans % 4 || c.var_chans % 4 || c.postnet_dim % 4) return "channel counts must be multiples of 4";
  if (c.ffn_k2 != 1 || !in(c.ffn_k1, 1, 255) || !(c.ffn_k1 & 1)) return "FFN kernels must be (odd, 1)";
  if (!in(c.dur_layers, 1, 16) || !in(c.var_layers, 1, 16) || !in(c.dur_kernel, 1, 255) || !in(c.var_kernel, 1, 255) || !(c.dur_kernel & 1) || !(c.var_kernel & 1))
    return "predictor layer counts / kernels out of range";
  if ((c.energy_layers != 0 && !in(c.energy_layers, 1, 16)) || (c.energy_kernel != 0 && (!in(c.energy_kernel, 1, 255) || !(c.energy_kernel & 1))))
    return "energy predictor layer count / kernel out of range";
  if (!in(c.postnet_layers, 1, 16) || !in(c.postnet_kernel, 1, 255) || !(c.postnet_kernel & 1)) return "postnet layers / kernel out of range";
  if (!in(c.voc_stages, 1, E2ETTS_MAX_STAGES) || !in(c.voc_n_kernels, 1, E2ETTS_MAX_RB_KERNELS) || !in(c.voc_n_dil, 1, E2ETTS_MAX_DILATIONS))
    return "vocoder stage / kernel / dilation counts out of range";
  for (int i = 0; i < c.voc_stages; ++i)
    if (!in(c.voc_up_rate[i], 2, 1024) || c.voc_up_kernel[i] != 2 * c.voc_up_rate[i] || (c.voc_up_rate[i] & 1)) return "upsample kernel must be 2 x rate, rate even";
  for (int j = 0; j < c.voc_n_kernels; ++j) {
    if (!in(c.voc_rb_kernel[j], 1, 255) || !(c.voc_rb_kernel[j] & 1)) return "ResBlock kernel sizes must be odd";
    for (int m = 0; m < c.voc_n_dil; ++m)
      if (!in(c.voc_rb_dil[j][m], 1, 1024)) return "ResBlock dilations must be positive";
  }
  if (!in(c.voc_init_ch, 4, BIG) || (c.voc_init_ch >> c.voc_stages) < 4 || (c.voc_init_ch % (4 << c.voc_stages))) return "upsample_initial_channel too small for the stage count";
  if (c.voc_resblock != 1 && c.voc_resblock != 2) return "voc_resblock must be 1 or 2";
  if (c.voc_resblock == 2 && c.voc_n_dil != 2) return "ResBlock2 has exactly two dilated convolutions (voc_n_dil == 2)";
  if (c.voc_istft_nfft != 0) {
    const int n = c.voc_istft_nfft;
    if (n < 4 || n > 256 || (n & (n - 1)) || !in(c.voc_istft_hop, 1, 256) || n % c.voc_istft_hop) return "iSTFT: n_fft must be a power of two in [4, 256] and a multiple of the hop";
  }
  if (c.block_type != 0 && c.block_type != 1) return "block_type must be 0 (FFT block) or 1 (Conformer block)";
  if (c.dec_n_head != 0 && (!in(c.dec_n_head, 1, BIG) || c.hidden % c.dec_n_head)) return "hidden must be a multiple of dec_n_head";
  if (c.block_type == 1) {
    const int dh = c.hidden / c.n_head, dhd = c.hidden / (c.dec_n_head ? c.dec_n_head : c.n_head);
    for (int d : {dh, dhd})
      if (d != 8 && d != 16 && d != 32 && d != 48 && d != 64 && d != 96) return "Conformer head dim must be one of 8, 16, 32, 48, 64, 96";
    if (c.ffn_dim < 2 * c.hidden) return "Conformer ffn_expansion_factor must be at least 2";
  }
  if (c.n_bins != 256) return "n_bins must be 256";
  if (c.pitch_emb_rows != 0 && !in(c.pitch_emb_rows, c.n_bins, BIG)) return "pitch_emb_rows must cover n_bins";
  if ((c.pitch_no_uv != 0 && c.pitch_no_uv != 1) || (c.pitch_log2 != 0 && c.pitch_log2 != 1) || (c.pred_pad_left != 0 && c.pred_pad_left != 1) ||
      (c.pitch_frame != 0 && c.pitch_frame != 1) || (c.energy_frame != 0 && c.energy_frame != 1))
    return "pitch_no_uv / pitch_log2 / pred_pad_left / pitch_frame / energy_frame must be 0 or 1";
  if (!in(c.max_seq_len, 1, BIG) || !in(c.pos_table_rows, c.max_seq_len + 1, 1 << 24)) return "pos_table_rows must cover max_seq_len + 1";
  if (!in(c.hop_length, 1, BIG) || !in(c.sample_rate, 1, 1 << 24)) return "hop_length and sample_rate must be positive";
  return nullptr;
}

// Rows past the last valid one (mel_len * rate_i) that the layers of vocoder stage i have to compute so that every valid sample sees
// exactly the inputs the padded computation gives it: out[i], i = 0 .. voc_stages (i = 0: conv_pre's output, rate 1; i >= 1: the
// upsampler output and the ResBlocks of stage i, rate = product of the first i upsampling factors).  Conservative bound from the
// layer geometry, back to front: the tail reads `tail` rows of the trunk's output past a sample (conv_post: 3; iSTFTNet: conv_post 3
// frames + the reflection pad's shift of 1 + the inverse STFT's overlap -- a sample is the sum of n_fft / hop frames, n_fft / (2 hop) to
// either side); every layer of a stage computes the rows its output needs plus the widest ResBlock's reach
// sum_m ((k-1)/2 d_m + (k-1)/2) (one limit per stage covers the intermediate layers too); an upsampler reads +-2 input positions.
inline void vocoder_stage_halo_rows(const e2etts_config& c, long long out[E2ETTS_MAX_STAGES + 1]) {
  long long need = c.voc_istft_nfft ? 4 + (c.voc_istft_nfft + 2 * c.voc_istft_hop - 1) / (2 * c.voc_istft_hop) : 3;
  for (int i = c.voc_stages; i >= 1; --i) {
    long long worst = 0;
    for (int j = 0; j < c.voc_n_kernels; ++j) {
      long long sum = 0;
      for (int m = 0; m < c.voc_n_dil; ++m) sum += (long long)(c.voc_rb_kernel[j] - 1) / 2 * (c.voc_rb_dil[j][m] + 1);
      worst = sum > worst ? sum : worst;
    }
    out[i] = need + worst;
    const long long r = c.voc_up_rate[i - 1];
    need = (out[i] + r - 1) / r + 2;
  }
  out[0] = need;
}

// Frames of mel context one output frame depends on, each side: conv_pre's output limit plus its own 3 frames.
inline int vocoder_halo_frames(const e2etts_config& c) {
  long long h[E2ETTS_MAX_STAGES + 1];
  vocoder_stage_halo_rows(c, h);
  return (int)std::min<long long>(h[0] + 3, 1 << 24);
}

// conv_gemm tile choice for Cout > 64 (rows = T per utterance).  few: so few tiles (small batches, the B = 1 latency path, the encoder)
// that 128 x 128 tiles would leave CUs idle -> 64 x 64.  half (fragment path only): a 128 x 128 grid whose last round of workgroups
// (2 per CU = 512 at a time) is mostly empty runs as 64 x 128 tiles: twice the workgroups at half the work.
inline long long tiles_128(int B, int T, int Cout) {  // 128 x 128 tiles of the launch, saturating (the callers only compare with ~512)
  const long long t = ((long long)T + 127) / 128 * (((long long)Cout + 127) / 128);
  return t > (1LL << 40) / (B > 0 ? B : 1) ? (1LL << 40) : t * B;
}
// Under one round of 128 x 128 tiles but >= 6144 rows (the first vocoder stage at B = 1 .. 2, 6144 x 256: 96 tiles): 64 x 64 tiles are
// 384 workgroups whose waves each fetch their own weight fragments for ONE 32 x 32 MFMA tile -- that launch waits on L2 bandwidth
// (540 MB of fragments in 55 us) -- so those run as 64 x 128 too: 192 workgroups, two MFMA row tiles per fragment (B = 1 step: 9.46 ->
// 9.05 ms fp32, 5.21 -> 5.11 ms bf16x3; B = 4: -4 %).  With fewer rows (the decoder at B <= 4, the encoder) 64 x 64 stayed ahead.
// The predicates on counts (t128 = 128 x 128 tiles of the launch, rows = output rows of the launch): a ragged batch passes the tiles and
// rows that really run (conv_gemm.hip counts them from ConvParams::act_rows_host), a padded one those of B x T.
inline bool tile_many_rows_n(long long rows) { return rows >= 6144; }
inline bool tile_few_rows_n(long long t128, long long rows, int Cout) { return Cout > 64 && t128 < 2 * 256 && !tile_many_rows_n(rows); }
inline bool tile_half_rows_n(long long t128, long long rows, int Cout) {
  if (Cout <= 64 || tile_few_rows_n(t128, rows, Cout)) return false;
  if (t128 < 2 * 256) return true;  // under one round, many rows
  const long long slots = 2 * 256;
  const long long r128 = (t128 + slots - 1) / slots, r64 = (2 * t128 + slots - 1) / slots;
  return 0.5 * 1.06 * (double)r64 < (double)r128;
}
inline bool tile_many_rows(int B, int T) { return tile_many_rows_n((long long)B * T); }
inline bool tile_few_rows(int B, int T, int Cout) { return tile_few_rows_n(tiles_128(B, T, Cout), (long long)B * T, Cout); }
inline bool tile_half_rows(int B, int T, int Cout) { return tile_half_rows_n(tiles_128(B, T, Cout), (long long)B * T, Cout); }

// 128 x 128 tiles and rows of a ragged launch: utterance b computes min(rows[b], T) rows (rows = ConvParams::act_rows_host)
inline void ragged_counts(const int32_t* rows, int B, int T, int Cout, long long* t128, long long* nrows) {
  long long t = 0, r = 0;
  for (int b = 0; b < B; ++b) {
    const long long v = rows[b] < 0 ? 0 : (rows[b] > T ? T : rows[b]);
    t += (v + 127) / 128;
    r += v;
  }
  *t128 = t * (((long long)Cout + 127) / 128);
  *nrows = r;
}

}  // namespace e2etts
