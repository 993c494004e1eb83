"""The engine's HIP-free host logic (e2e_tts_amd/csrc/host_logic.h: weight-blob header / directory validation, config
validation, vocoder halo, conv_gemm tile choice) built with AddressSanitizer + UBSan on the CPU (the reference
has no sanitizer target; GPU ASan is not available on this pool) and fed valid, truncated and corrupt inputs.  engine.hip and
conv_gemm.hip include the same header, so this is the code the library ships."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw

SRC = os.path.join(ROOT, "tests", "csrc", "host_logic_test.cc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("san") / "host_logic_test")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", SRC, "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, f"sanitizer report or crash (rc={r.returncode}):\n{r.stderr[-3000:]}"
    return r.stdout.strip()


@pytest.fixture(scope="module")
def tiny_blob():
    cfg = cfgmod.tiny_config()
    dims = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, n_speakers=4)
    ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=1234, mode="varied")
    voc = sw.make_vocoder_state(cfg, seed=4321)
    return dims, packer.pack(dims, ac, voc)


def test_valid_blob_parses(harness, tiny_blob, tmp_path):
    _, blob = tiny_blob
    f = tmp_path / "ok.blob"
    blob.tofile(f)
    out = run(harness, "blob", f)
    n_entries = struct.unpack_from("<I", blob, 12)[0]
    assert out.startswith(f"OK {n_entries} "), out


def test_truncated_blobs_are_rejected_without_overreads(harness, tiny_blob, tmp_path):
    _, blob = tiny_blob
    data_offset = struct.unpack_from("<Q", blob, 16)[0]
    for n in (0, 1, 31, 32, 33, 111, 112, int(data_offset) - 1, int(data_offset), int(data_offset) + 255, blob.size // 2, blob.size - 1):
        f = tmp_path / f"trunc{n}.blob"
        blob[:n].tofile(f)
        assert run(harness, "blob", f).startswith("ERR"), n
    # truncated AND the header patched to claim the shorter size: the directory entries that now point past the end must be caught
    for n in (int(data_offset), int(data_offset) + 256, blob.size // 2 // 256 * 256):
        b = blob[:n].copy()
        struct.pack_into("<Q", b, 24, n)
        f = tmp_path / f"relabel{n}.blob"
        b.tofile(f)
        assert run(harness, "blob", f).startswith("ERR tensor"), n


def test_corrupt_headers_and_directories_are_rejected(harness, tiny_blob, tmp_path):
    _, blob = tiny_blob
    n_entries = struct.unpack_from("<I", blob, 12)[0]
    data_offset = struct.unpack_from("<Q", blob, 16)[0]

    def variant(name, patch):
        b = blob.copy()
        patch(b)
        f = tmp_path / f"{name}.blob"
        b.tofile(f)
        return run(harness, "blob", f)

    assert variant("magic", lambda b: b.__setitem__(0, 0x58)).startswith("ERR not an e2etts")
    assert variant("version", lambda b: struct.pack_into("<I", b, 8, 2)).startswith("ERR not an e2etts")
    assert variant("entries_huge", lambda b: struct.pack_into("<I", b, 12, 0xFFFFFFFF)).startswith("ERR corrupt blob directory")
    assert variant("entries_past_data", lambda b: struct.pack_into("<I", b, 12, n_entries + 100000 - n_entries % 100000 - 1)).startswith("ERR")
    assert variant("entries_plus_some", lambda b: struct.pack_into("<I", b, 12, (data_offset - 32) // 80 + 1)).startswith("ERR corrupt blob directory")
    assert variant("data_offset_unaligned", lambda b: struct.pack_into("<Q", b, 16, data_offset + 4)).startswith("ERR corrupt blob directory")
    assert variant("data_offset_past_end", lambda b: struct.pack_into("<Q", b, 16, (blob.size + 255) // 256 * 256 + 256)).startswith("ERR corrupt blob directory")
    assert variant("total", lambda b: struct.pack_into("<Q", b, 24, blob.size + 1)).startswith("ERR blob size")
    first = 32  # first directory entry: name[64], offset, numel
    for name, off, numel in (("off_unaligned", data_offset + 4, 1), ("off_before_data", 0, 1), ("off_past_end", 1 << 40, 1),
                             ("numel_huge", data_offset, (1 << 64) - 1), ("numel_wraps", data_offset, (1 << 62) + 7),
                             ("numel_one_too_many", data_offset, (blob.size - data_offset) // 4 + 1)):
        out = variant(name, lambda b: struct.pack_into("<QQ", b, first + 64, off, numel))
        assert out.startswith("ERR tensor"), (name, out)
    # an entry name without a terminating NUL must not be read past its 64 bytes
    out = variant("name_unterminated", lambda b: b.__setitem__(slice(first, first + 64), 0x41))
    assert out.startswith("OK") or out.startswith("ERR"), out
    # the largest tensor that still fits is accepted
    out = variant("numel_exact", lambda b: struct.pack_into("<QQ", b, first + 64, data_offset, (blob.size - data_offset) // 4))
    assert out.startswith("OK"), out


def test_config_validation_and_halo(harness, tiny_blob, tmp_path):
    dims, _ = tiny_blob
    good = dims.to_c()
    raw = bytes(ctypes.string_at(ctypes.addressof(good), ctypes.sizeof(good)))
    f = tmp_path / "cfg.bin"
    f.write_bytes(raw)
    assert run(harness, "config", f).startswith("OK halo=")
    d = cfgmod.dims_from_config(cfgmod.default_config(), cfgmod.DEFAULT_STATS, n_speakers=4).to_c()
    f.write_bytes(bytes(ctypes.string_at(ctypes.addressof(d), ctypes.sizeof(d))))
    # HiFi-GAN V1 (include/e2etts.h: streaming vocoder; ragged limits per stage: conv_post 3 + ResBlock reach 60 = 63 rows at the
    # output rate, then (rows / rate + 2) + 60 stage by stage towards the mel)
    assert run(harness, "config", f) == "OK halo=15 stage_rows=12,76,109,94,63"
    # every int32 field forced to hostile values: rejected or accepted, never a sanitizer report (shift / overflow / index)
    n_fields = len(raw) // 4
    for val in (0, -1, 0x7FFFFFFF, -0x80000000, 1 << 20, 3):
        for i in range(n_fields):
            b = bytearray(raw)
            struct.pack_into("<i", b, 4 * i, val)
            f.write_bytes(bytes(b))
            out = run(harness, "config", f)
            assert out.startswith("OK") or out.startswith("ERR"), (i, val, out)
    rng = np.random.Generator(np.random.PCG64(7))
    for _ in range(200):
        f.write_bytes(rng.integers(0, 256, size=len(raw), dtype=np.uint8).tobytes())
        out = run(harness, "config", f)
        assert out.startswith("OK") or out.startswith("ERR"), out


def test_tile_choice(harness):
    # the headline workload's shapes: decoder Linear 24576 x 384 -> 64 x 128 tiles; big vocoder layers stay 128 x 128;
    # the encoder (B * L phonemes) and B = 1 take the few-rows tile
    assert run(harness, "tiles", 32, 768, 384) == "few=0 half=1"
    assert run(harness, "tiles", 32, 6144, 256) == "few=0 half=0"
    assert run(harness, "tiles", 32, 128, 384) == "few=1 half=0"
    assert run(harness, "tiles", 1, 768, 1024) == "few=1 half=0"
    assert run(harness, "tiles", 1, 6144, 256) == "few=0 half=1"      # under one round of 128 x 128 tiles, many rows: 64 x 128
    assert run(harness, "tiles", 32, 768, 64) == "few=0 half=0"
    for args in ((0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF), (1, 1, 65), (4096, 1 << 20, 8192)):
        assert run(harness, "tiles", *args).startswith("few=")


def test_ragged_counts(harness):
    """The counts a ragged launch is sized on (host_logic.h: ragged_counts, fed from ConvParams::act_rows_host): rows are clamped to
    [0, T] per utterance, tiles counted per utterance (each ends in a partial tile), and the tile choice follows the REAL counts -- the
    mixed-length decoder Linear takes the 64-row tile where its padded shape (32 x 1200) would not."""
    assert run(harness, "ragged", 1200, 384, 240, 1200, 600) == "t128=51 rows=2040 few=1 half=0"          # (2 + 10 + 5) row tiles x 3 column tiles
    assert run(harness, "ragged", 1200, 384, -5, 5000, 0) == "t128=30 rows=1200 few=1 half=0"            # clamped: 0, T, 0
    lens = [6 * n for n in (40, 45, 50, 56, 61, 66, 71, 76, 81, 86, 92, 97, 102, 107, 112, 117, 123, 128, 133, 138, 143, 148, 154, 159, 164,
                            169, 174, 179, 184, 190, 195, 200)]
    out = run(harness, "ragged", 1200, 1024, *lens)
    assert out.startswith("t128=") and out.endswith("few=0 half=1"), out
    assert run(harness, "tiles", 32, 1200, 1024) == "few=0 half=0"
    assert run(harness, "ragged", 0x7FFFFFFF, 0x7FFFFFFF, *([0x7FFFFFFF] * 64)).startswith("t128=")


def _exact_stage_rows(rates, kernels, dils, resblock=1, tail=3):
    """Rows past the last valid one that each vocoder stage must hold CORRECT values for, from the layer index maps alone
    (V/generator.py:37-53, V/layers.py:33-40,59-62): conv_post reads +-3 samples; a ResBlock1 pair (conv k dilation d, conv k) reaches
    (k - 1) / 2 * (d + 1) rows, a ResBlock2 convolution (k - 1) / 2 * d; ConvTranspose1d(k = 2 r, stride r, padding r / 2) output t
    reads inputs up to floor((t + r / 2) / r)."""
    need = tail
    rows = [0] * (len(rates) + 1)
    for i in range(len(rates), 0, -1):
        reach = max(sum((k - 1) // 2 * (d + (1 if resblock == 1 else 0)) for d in dl) for k, dl in zip(kernels, dils))
        rows[i] = need + reach
        r = rates[i - 1]
        need = (rows[i] - 1 + r // 2) // r + 1
    rows[0] = need
    return rows


def test_stage_limits_cover_the_exact_receptive_field(harness, tmp_path):
    """host_logic.h: vocoder_stage_halo_rows (the ragged limits of every vocoder stage) against an independent derivation from the layer
    geometry, for every generator the fixtures use: never less than the exact reach, never more than a few rows above it."""
    cases = []
    for make in (cfgmod.tiny_config, cfgmod.default_config):
        cfg = make()
        cases.append((cfg, "hifigan"))
    c48 = cfgmod.default_config()   # BASELINE config 5's generator (tests/golden/hifigan_48k.npz)
    c48["models"]["hifigan"].update(upsample_rates=[8, 8, 4, 2], upsample_kernel_sizes=[16, 16, 8, 4])
    c48["audio"]["stft"]["hop_length"] = 512
    cases.append((c48, "hifigan"))
    f = tmp_path / "cfg.bin"
    for cfg, voc in cases:
        hg = cfg["models"][voc]
        d = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, n_speakers=4, vocoder=voc).to_c()
        f.write_bytes(bytes(ctypes.string_at(ctypes.addressof(d), ctypes.sizeof(d))))
        out = run(harness, "config", f)
        assert out.startswith("OK halo="), out
        got = [int(x) for x in out.split("stage_rows=")[1].split(",")]
        halo = int(out.split("halo=")[1].split()[0])
        want = _exact_stage_rows(hg["upsample_rates"], hg["resblock_kernel_sizes"], hg["resblock_dilation_sizes"], int(hg.get("resblock", 1)))
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert w <= g <= w + 4, (got, want)
        assert halo == got[0] + 3   # conv_pre's own reach


def test_halo_frames_hold_numerically_on_the_oracle_vocoder(harness, tmp_path):
    """The frame halo the engine uses (streaming context, ragged limits) against the numpy HifiGan restatement: changing the mel from frame
    t0 + halo on must leave every sample of the first t0 frames untouched bit for bit, and one frame earlier than the exact reach must not
    (so the bound is tight to within its stated slack, not vacuous)."""
    from e2e_tts_amd import synth_weights as sw
    from oracle import ref_numpy as orc
    cfg = cfgmod.tiny_config()
    d = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, n_speakers=4).to_c()
    f = tmp_path / "cfg.bin"
    f.write_bytes(bytes(ctypes.string_at(ctypes.addressof(d), ctypes.sizeof(d))))
    out = run(harness, "config", f)
    halo = int(out.split("halo=")[1].split()[0])
    voc = orc.VocoderOracle(sw.make_vocoder_state(cfg, seed=5), cfg)
    hop = cfg["audio"]["stft"]["hop_length"]
    n_mel = cfg["audio"]["mel"]["channels"]
    rng = np.random.Generator(np.random.PCG64(11))
    t0, T = 9, 9 + halo + 6
    mel = rng.standard_normal((1, n_mel, T)).astype(np.float32)
    base = voc.forward(mel)[0, 0]
    far = mel.copy()
    far[:, :, t0 + halo:] += 3.0
    np.testing.assert_array_equal(voc.forward(far)[0, 0][: t0 * hop], base[: t0 * hop])
    near = mel.copy()
    near[:, :, t0 + 2:] += 3.0   # two frames past the end: well inside the receptive field
    assert not np.array_equal(voc.forward(near)[0, 0][: t0 * hop], base[: t0 * hop])
