"""Configuration for the hot path.

The reference merges three YAML files into ``{"audio", "models", "train"}``
(reference: e2e_tts/src/tools/tools_for_model.py:14-20) and stores the result
as ``config.yaml`` next to the acoustic checkpoint, where ``TTS.__init__``
reads it back (reference: e2e_tts/src/api/utils.py:34-36).  This module holds

* ``default_config()`` -- the same dictionary *shape* with the hot-path keys
  only (values: reference e2e_tts/config/model_config.yaml:1-92 and
  e2e_tts/config/preprocessing_config.yaml:1-14), used for synthetic weights;
* ``EngineDims`` -- the flat, integer view of that dictionary that crosses the
  C ABI as ``e2etts_config`` (include/e2etts.h).
"""
from __future__ import annotations

import copy
import ctypes
from dataclasses import dataclass, field
from typing import List

N_SYMBOLS = 131  # reference: e2e_tts/models/g2p/symbols.py:19-50

_DEFAULT = {
    "audio": {
        "signal": {"sampling_rate": 22050, "max_wav_value": 32768.0},
        "stft": {"filter_length": 1024, "hop_length": 256, "win_length": 1024},
        "mel": {"channels": 80},
    },
    "models": {
        "fastspeech2": {
            "max_seq_len": 1000,
            "encoder_layers": 6,
            "encoder_hidden": 384,
            "decoder_layers": 6,
            "decoder_hidden": 384,
            "building_block": {
                "block_type": "transformer",
                "transformer": {
                    "encoder_head": 2,
                    "decoder_head": 2,
                    "conv_filter_size": 1024,
                    "conv_kernel_size": [9, 1],
                    "encoder_dropout": 0.1,
                    "decoder_dropout": 0.1,
                },
                # reference config/model_config.yaml:16-24 (selected by block_type: "conformer", U/model.py:26-27)
                "conformer": {
                    "encoder_head": 8,
                    "decoder_head": 8,
                    "ffn_expansion_factor": 4,
                    "conv_kernel_size": 31,
                    "conv_expansion_factor": 2,
                    "half_step_residual": True,
                    "encoder_dropout": 0.1,
                    "decoder_dropout": 0.1,
                },
            },
            "variance": {
                "duration_modelling": {
                    "learn_alignment": True,
                    "aligner_temperature": 0.0005,
                    "binarization_start_steps": 6000,
                },
                "variance_predictor": {
                    "predictor_grad": 0.1,
                    "filter_size": 256,
                    "kernel_size": 3,
                    "dropout": 0.5,
                    "dur_predictor_layers": 2,
                    "dur_predictor_kernel": 3,
                    "pit_predictor_layers": 2,
                    "pit_predictor_kernel": 5,
                    "ener_predictor_layers": 2,
                    "ener_predictor_kernel": 5,
                    "ffn_padding": "SAME",
                    "ffn_act": "gelu",
                },
                "variance_embedding": {
                    "use_uv": True,
                    "n_bins": 256,
                    "pitch_feature": "phoneme_level",
                    "pitch_quantization": "linear",
                    "energy_feature": "phoneme_level",
                    "energy_quantization": "linear",
                    "f0_bins": 300,
                },
            },
            "postnet": {"embedding_dim": 512, "conv_layers": 5, "kernel_size": 5},
        },
        "hifigan": {
            "resblock": 1,
            "upsample_rates": [8, 8, 2, 2],
            "upsample_kernel_sizes": [16, 16, 4, 4],
            "upsample_initial_channel": 512,
            "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        },
        # iSTFTNet (reference config/model_config.yaml:83-92): two upsampling stages, then an inverse STFT (n_fft 16, hop 4)
        "istft": {
            "resblock": 1,
            "gen_istft_n_fft": 16,
            "gen_istft_hop_size": 4,
            "gen_istft_win_size": 16,
            "upsample_rates": [8, 8],
            "upsample_kernel_sizes": [16, 16],
            "upsample_initial_channel": 512,
            "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        },
    },
}

#: stats.json of the synthetic model; the reference reads
#: these keys at U/layers.py:77-84,115-122,152.
DEFAULT_STATS = {
    "f0": {"mean": 191.463, "std": 67.695},
    "pitch": {"min": -2.047, "max": 10.332},
    "energy": {"min": -1.258, "max": 7.351},
}

DEFAULT_SPEAKERS = {"hn_minhphuong": 0, "spk_b": 1, "spk_c": 2, "spk_d": 3}


def default_config() -> dict:
    return copy.deepcopy(_DEFAULT)


def tiny_config() -> dict:
    """A small model (hidden 64, 2+2 layers, vocoder width 64) for fast tests.

    Every structural feature of the default model is kept (2 heads, k9/k1 FFN,
    k3/k5 predictors, 5-layer postnet, 4 upsampling stages x 3 ResBlock1).
    """
    c = default_config()
    fs = c["models"]["fastspeech2"]
    fs["max_seq_len"] = 60
    fs["encoder_layers"] = 2
    fs["decoder_layers"] = 2
    fs["encoder_hidden"] = 64
    fs["decoder_hidden"] = 64
    fs["building_block"]["transformer"]["conv_filter_size"] = 96
    fs["building_block"]["conformer"].update(encoder_head=4, decoder_head=4, conv_kernel_size=7)
    fs["variance"]["variance_predictor"]["filter_size"] = 48
    fs["postnet"]["embedding_dim"] = 48
    c["models"]["hifigan"]["upsample_initial_channel"] = 64
    c["models"]["istft"]["upsample_initial_channel"] = 64
    return c


MAX_STAGES = 8
MAX_RESBLOCK_KERNELS = 4
MAX_DILATIONS = 4


class CEngineConfig(ctypes.Structure):
    """Mirror of ``e2etts_config`` in include/e2etts.h.  Kept in sync by two checks, not by hand alone: ``_lib.load_library`` refuses a
    library whose ``e2etts_config_size()`` differs from ``ctypes.sizeof`` of this class, and tests/test_host_logic.py compares the field
    names and order with the header's text."""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_symbols", ctypes.c_int32),
        ("n_speakers", ctypes.c_int32),
        ("n_mel", ctypes.c_int32),
        ("hidden", ctypes.c_int32),
        ("enc_layers", ctypes.c_int32),
        ("dec_layers", ctypes.c_int32),
        ("n_head", ctypes.c_int32),
        ("ffn_dim", ctypes.c_int32),
        ("ffn_k1", ctypes.c_int32),
        ("ffn_k2", ctypes.c_int32),
        ("max_seq_len", ctypes.c_int32),
        ("dur_layers", ctypes.c_int32),
        ("dur_kernel", ctypes.c_int32),
        ("dur_chans", ctypes.c_int32),
        ("var_layers", ctypes.c_int32),
        ("var_kernel", ctypes.c_int32),
        ("var_chans", ctypes.c_int32),
        ("n_bins", ctypes.c_int32),
        ("postnet_layers", ctypes.c_int32),
        ("postnet_dim", ctypes.c_int32),
        ("postnet_kernel", ctypes.c_int32),
        ("voc_init_ch", ctypes.c_int32),
        ("voc_stages", ctypes.c_int32),
        ("voc_up_rate", ctypes.c_int32 * MAX_STAGES),
        ("voc_up_kernel", ctypes.c_int32 * MAX_STAGES),
        ("voc_n_kernels", ctypes.c_int32),
        ("voc_rb_kernel", ctypes.c_int32 * MAX_RESBLOCK_KERNELS),
        ("voc_n_dil", ctypes.c_int32),
        ("voc_rb_dil", (ctypes.c_int32 * MAX_DILATIONS) * MAX_RESBLOCK_KERNELS),
        ("hop_length", ctypes.c_int32),
        ("sample_rate", ctypes.c_int32),
        ("pos_table_rows", ctypes.c_int32),
        ("f0_mean", ctypes.c_float),
        ("f0_std", ctypes.c_float),
        ("voc_resblock", ctypes.c_int32),
        ("voc_istft_nfft", ctypes.c_int32),
        ("voc_istft_hop", ctypes.c_int32),
        ("block_type", ctypes.c_int32),
        ("energy_layers", ctypes.c_int32),
        ("energy_kernel", ctypes.c_int32),
        ("dec_n_head", ctypes.c_int32),
        ("pitch_no_uv", ctypes.c_int32),
        ("pitch_log2", ctypes.c_int32),
        ("pitch_emb_rows", ctypes.c_int32),
        ("pred_pad_left", ctypes.c_int32),
        ("pitch_frame", ctypes.c_int32),
        ("energy_frame", ctypes.c_int32),
    ]


@dataclass
class EngineDims:
    n_symbols: int
    n_speakers: int
    n_mel: int
    hidden: int
    enc_layers: int
    dec_layers: int
    n_head: int
    ffn_dim: int
    ffn_k1: int
    ffn_k2: int
    max_seq_len: int
    dur_layers: int
    dur_kernel: int
    dur_chans: int
    var_layers: int
    var_kernel: int
    var_chans: int
    n_bins: int
    postnet_layers: int
    postnet_dim: int
    postnet_kernel: int
    voc_init_ch: int
    voc_up_rate: List[int] = field(default_factory=list)
    voc_up_kernel: List[int] = field(default_factory=list)
    voc_rb_kernel: List[int] = field(default_factory=list)
    voc_rb_dil: List[List[int]] = field(default_factory=list)
    hop_length: int = 256
    sample_rate: int = 22050
    pos_table_rows: int = 4096
    f0_mean: float = 0.0
    f0_std: float = 1.0
    voc_resblock: int = 1     # 1: ResBlock1, 2: ResBlock2 (reference V/layers.py)
    voc_istft_nfft: int = 0   # 0: HiFi-GAN tail; else iSTFTNet (reference V/generator.py:65-113)
    voc_istft_hop: int = 0
    energy_layers: int = 0    # energy predictor depth / kernel when they differ from the pitch predictor's (0: the same; U/layers.py:92,96)
    energy_kernel: int = 0
    dec_n_head: int = 0       # decoder_head when it differs from encoder_head (0: the same; U/blocks/transformer.py:105)
    pitch_no_uv: int = 0      # variance_embedding.use_uv False: one pitch output, bucketize on pitch_bins, f0_bins embedding rows (U/layers.py:136-160)
    pitch_log2: int = 0       # use_uv with pitch_quantization "log": f0 = 2 ** prediction (U/layers.py:148-149)
    pitch_emb_rows: int = 0   # rows of pitch_embedding when not n_bins
    pred_pad_left: int = 0    # variance_predictor.ffn_padding "LEFT": causal predictor convolutions (U/layers.py:400-402)
    pitch_frame: int = 0      # variance_embedding.pitch_feature / energy_feature "frame_level": predictor + embedding on the length regulator's
    energy_frame: int = 0     # output instead of on the phonemes (U/layers.py:226-257)
    cf_ffn_factor: float = 0.5  # Conformer half_step_residual (U/blocks/conformer.py:209-212); folded into the weights by the packer
    block_type: int = 0       # 0: FFT block (U/blocks/transformer.py), 1: Conformer block (U/blocks/conformer.py); then ffn_dim =
                              # hidden x ffn_expansion_factor and ffn_k1 = the depthwise kernel size

    @property
    def upsample_total(self) -> int:
        t = 1
        for r in self.voc_up_rate:
            t *= r
        return t * (self.voc_istft_hop if self.voc_istft_nfft else 1)

    @property
    def voc_post_channels(self) -> int:
        """conv_post output channels as stored in the blob: 1 (HiFi-GAN), or n_fft + 2 padded to a multiple of 4 (iSTFTNet)."""
        return (self.voc_istft_nfft + 2 + 3) // 4 * 4 if self.voc_istft_nfft else 1

    def to_c(self) -> CEngineConfig:
        c = CEngineConfig()
        c.struct_size = ctypes.sizeof(CEngineConfig)
        for name in (
            "n_symbols n_speakers n_mel hidden enc_layers dec_layers n_head ffn_dim ffn_k1 ffn_k2 "
            "max_seq_len dur_layers dur_kernel dur_chans var_layers var_kernel var_chans n_bins "
            "postnet_layers postnet_dim postnet_kernel voc_init_ch hop_length sample_rate pos_table_rows"
        ).split():
            setattr(c, name, int(getattr(self, name)))
        c.f0_mean = float(self.f0_mean)
        c.f0_std = float(self.f0_std)
        c.voc_resblock, c.voc_istft_nfft, c.voc_istft_hop = int(self.voc_resblock), int(self.voc_istft_nfft), int(self.voc_istft_hop)
        c.block_type = int(self.block_type)
        c.energy_layers, c.energy_kernel = int(self.energy_layers), int(self.energy_kernel)
        c.dec_n_head = int(self.dec_n_head)
        c.pitch_no_uv, c.pitch_log2, c.pitch_emb_rows = int(self.pitch_no_uv), int(self.pitch_log2), int(self.pitch_emb_rows)
        c.pred_pad_left = int(self.pred_pad_left)
        c.pitch_frame, c.energy_frame = int(self.pitch_frame), int(self.energy_frame)
        if len(self.voc_up_rate) > MAX_STAGES or len(self.voc_rb_kernel) > MAX_RESBLOCK_KERNELS:
            raise ValueError("vocoder config exceeds the C-ABI limits")
        c.voc_stages = len(self.voc_up_rate)
        for i, (r, k) in enumerate(zip(self.voc_up_rate, self.voc_up_kernel)):
            c.voc_up_rate[i] = r
            c.voc_up_kernel[i] = k
        c.voc_n_kernels = len(self.voc_rb_kernel)
        c.voc_n_dil = len(self.voc_rb_dil[0])
        for j, k in enumerate(self.voc_rb_kernel):
            c.voc_rb_kernel[j] = k
            if len(self.voc_rb_dil[j]) != c.voc_n_dil or c.voc_n_dil > MAX_DILATIONS:
                raise ValueError("ragged / oversize resblock dilation list")
            for m, d in enumerate(self.voc_rb_dil[j]):
                c.voc_rb_dil[j][m] = d
        return c


def dims_from_config(config: dict, stats: dict, n_speakers: int, n_symbols: int = N_SYMBOLS,
                     pos_table_rows: int = 4096, vocoder: str = "hifigan") -> EngineDims:
    """Flatten the reference-style config dict into the engine dims.

    Raises for every configuration the hot path does not implement, naming the
    reference location that selects it, instead of silently computing
    something else.
    """
    fs = config["models"]["fastspeech2"]
    if vocoder not in ("hifigan", "istft"):
        raise ValueError("vocoder must be 'hifigan' or 'istft'")
    hg = config["models"][vocoder]
    bt = fs["building_block"]["block_type"]
    if bt not in ("transformer", "conformer"):
        raise NotImplementedError(
            f"building_block.block_type={bt!r}: only the 'transformer' FFT block and the 'conformer' block are implemented "
            "(reference U/model.py:24-33; SURVEY.md section 8(f))")
    tr = fs["building_block"][bt]
    if bt == "conformer":
        # U/blocks/conformer.py:31-36: heads, FFN expansion, depthwise kernel; conv expansion is asserted to be 2 there (:466)
        if tr["conv_expansion_factor"] != 2:
            raise NotImplementedError("conv_expansion_factor must be 2 (reference U/blocks/conformer.py:466)")
        if tr["conv_kernel_size"] % 2 != 1:
            raise ValueError("conformer conv_kernel_size must be odd (reference U/blocks/conformer.py:465)")
        tr = dict(tr, conv_filter_size=fs["encoder_hidden"] * tr["ffn_expansion_factor"], conv_kernel_size=[tr["conv_kernel_size"], 1])
    if fs["encoder_hidden"] != fs["decoder_hidden"]:
        raise NotImplementedError("encoder_hidden != decoder_hidden")
    var = fs["variance"]
    if not var["duration_modelling"]["learn_alignment"]:
        raise NotImplementedError("SupervisedFastSpeech2 is out of scope (reference API/utils.py:37-40)")
    ve = var["variance_embedding"]
    for k in ("pitch_feature", "energy_feature"):
        if ve[k] not in ("phoneme_level", "frame_level"):
            raise ValueError(f"{k} must be 'phoneme_level' or 'frame_level' (reference U/layers.py:48,88)")
    if ve["pitch_quantization"] not in ("linear", "log") or ve.get("energy_quantization", "linear") not in ("linear", "log"):
        raise ValueError("pitch_quantization / energy_quantization must be 'linear' or 'log' (reference U/layers.py:66,104)")
    vp = var["variance_predictor"]
    # HifiGan picks ResBlock1 for `config['resblock'] == 1` (V/generator.py:19); iSTFT compares with the STRING '1'
    # (V/generator.py:71), so the shipped yaml (an int) gives it ResBlock2 -- reproduced, checkpoints depend on it
    rb1 = (hg["resblock"] == "1") if vocoder == "istft" else (hg["resblock"] == 1)
    rb_dil = [list(d) for d in hg["resblock_dilation_sizes"]]
    if not rb1:
        rb_dil = [d[:2] for d in rb_dil]  # ResBlock2 uses dilation[0], dilation[1] only (V/layers.py:52-56)
        if any(len(d) != 2 for d in rb_dil):
            raise ValueError("ResBlock2 needs at least two dilations per kernel size")
    for k, u in zip(hg["upsample_kernel_sizes"], hg["upsample_rates"]):
        if k != 2 * u or u % 2:
            raise NotImplementedError("upsample kernel must be 2 x rate with even rate (polyphase 3-tap form)")
    n_mel = config["audio"]["mel"]["channels"]
    hop = config["audio"]["stft"]["hop_length"]
    dims = EngineDims(
        n_symbols=n_symbols, n_speakers=n_speakers, n_mel=n_mel,
        hidden=fs["encoder_hidden"], enc_layers=fs["encoder_layers"], dec_layers=fs["decoder_layers"],
        n_head=tr["encoder_head"], ffn_dim=tr["conv_filter_size"],
        ffn_k1=tr["conv_kernel_size"][0], ffn_k2=tr["conv_kernel_size"][1],
        max_seq_len=fs["max_seq_len"],
        dur_layers=vp["dur_predictor_layers"], dur_kernel=vp["dur_predictor_kernel"], dur_chans=n_mel,
        var_layers=vp["pit_predictor_layers"], var_kernel=vp["pit_predictor_kernel"], var_chans=vp["filter_size"],
        n_bins=ve["n_bins"],
        postnet_layers=fs["postnet"]["conv_layers"], postnet_dim=fs["postnet"]["embedding_dim"],
        postnet_kernel=fs["postnet"]["kernel_size"],
        voc_init_ch=hg["upsample_initial_channel"],
        voc_up_rate=list(hg["upsample_rates"]), voc_up_kernel=list(hg["upsample_kernel_sizes"]),
        voc_rb_kernel=list(hg["resblock_kernel_sizes"]),
        voc_rb_dil=rb_dil, voc_resblock=1 if rb1 else 2,
        voc_istft_nfft=int(hg["gen_istft_n_fft"]) if vocoder == "istft" else 0,
        voc_istft_hop=int(hg["gen_istft_hop_size"]) if vocoder == "istft" else 0,
        hop_length=hop, sample_rate=config["audio"]["signal"]["sampling_rate"],
        pos_table_rows=pos_table_rows,
        f0_mean=float(stats["f0"]["mean"]), f0_std=float(stats["f0"]["std"]),
        block_type=1 if bt == "conformer" else 0,
        energy_layers=vp["ener_predictor_layers"], energy_kernel=vp["ener_predictor_kernel"],
        dec_n_head=tr["decoder_head"],
        pred_pad_left=0 if vp["ffn_padding"] == "SAME" else 1,
        pitch_frame=1 if ve["pitch_feature"] == "frame_level" else 0, energy_frame=1 if ve["energy_feature"] == "frame_level" else 0,
        pitch_no_uv=0 if ve["use_uv"] else 1,
        pitch_log2=1 if (ve["use_uv"] and ve["pitch_quantization"] == "log") else 0,   # (without uv the log / linear choice lives in the checkpoint's pitch_bins)
        pitch_emb_rows=ve["n_bins"] if ve["use_uv"] else ve["f0_bins"],
        cf_ffn_factor=(0.5 if tr.get("half_step_residual", True) else 1.0),
    )
    if dims.ffn_k2 != 1:
        raise NotImplementedError("second FFN conv must be k=1")
    if vocoder == "istft":
        if hg["gen_istft_win_size"] != hg["gen_istft_n_fft"]:
            raise NotImplementedError("gen_istft_win_size must equal gen_istft_n_fft")
        n = dims.voc_istft_nfft
        if n < 4 or n > 256 or n & (n - 1) or n % dims.voc_istft_hop:
            raise NotImplementedError("gen_istft_n_fft must be a power of two in [4, 256] and a multiple of gen_istft_hop_size")
    if dims.upsample_total != hop:
        raise ValueError(f"product of upsample_rates (x iSTFT hop) ({dims.upsample_total}) != hop_length ({hop})")
    if dims.hidden % dims.n_head or dims.hidden % (dims.dec_n_head or dims.n_head):
        raise ValueError("hidden not divisible by heads")
    return dims
