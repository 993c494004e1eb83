"""CPU tests of the product's host side (no GPU, no oracle in the product path): text arrangement / batching /
PCM assembly against the fixture produced by the reference's own methods, the weight packer, the C ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden
from e2e_tts_amd import config as cfgmod, packer, synth_weights as sw
from e2e_tts_amd.api import TTS


def bare_tts(max_len=300, hop=256):
    t = TTS.__new__(TTS)  # host-logic methods only: no checkpoint, no engine
    t.max_len, t.hop_length, t.max_wav_value, t.sample_rate = max_len, hop, 32768.0, 22050
    return t


def toy_tokenizer(txt):
    return [4 + (ord(c) % 127) for c in txt]


def test_arrange_and_batch_match_reference_fixture():
    g = load_golden("host_loop")
    t = bare_tts()
    texts = [str(x) for x in g["texts"]]
    arranged = t.arrange_text(list(texts))
    assert arranged == [str(x) for x in g["arranged"]]
    t.text_to_sequence = toy_tokenizer
    batches, revert = t.input_parse(list(texts))
    np.testing.assert_array_equal(revert.numpy(), g["revert"])
    assert len(batches) == int(g["n_batches"])
    for i, (ids, lens) in enumerate(batches):
        np.testing.assert_array_equal(ids.numpy(), g[f"ids{i}"])
        np.testing.assert_array_equal(lens.numpy(), g[f"lens{i}"])


def test_token_budget_quirk_matches_reference_fixture():
    g = load_golden("host_loop")
    lens = g["stress_lens"]
    batches, revert = TTS.pack_sequences([[5] * int(n) for n in lens], 300)
    np.testing.assert_array_equal([len(l) for _, l in batches], g["stress_batch_sizes"])
    np.testing.assert_array_equal([int(l[0]) for _, l in batches], g["stress_batch_first_len"])
    order = np.argsort(revert)
    ref_order = np.argsort(g["stress_revert"])
    np.testing.assert_array_equal(lens[order], lens[ref_order])  # equal up to the tie order torch.sort leaves open


def test_combine_audio_matches_reference_fixture():
    g = load_golden("host_loop")
    t = bare_tts()
    audios = [g["ca_audio0"], g["ca_audio1"], g["ca_audio2"]]
    pcm = t.combine_audio(audios, g["ca_lengths"], int(g["ca_distance"]))
    np.testing.assert_array_equal(pcm, g["ca_pcm"])
    # the GPU emits (int16)(int32)(wav * 32768); assembling those must give the same stream
    gpu_like = [(a * np.float32(32768.0)).astype(np.int32).astype(np.int16) for a in audios]
    np.testing.assert_array_equal(t._combine_pcm(gpu_like, list(g["ca_lengths"]), int(g["ca_distance"])), g["ca_pcm"])


def test_empty_and_single_inputs():
    batches, revert = TTS.pack_sequences([[7, 8, 9]], 300)
    assert len(batches) == 1 and batches[0][0].shape == (1, 3) and revert.tolist() == [0]
    batches, _ = TTS.pack_sequences([[5] * 400, [5] * 10], 300)  # an over-budget sentence still forms a batch
    assert [len(l) for _, l in batches][0] == 1
    t = bare_tts()
    assert t.arrange_text([]) == []


def test_polyphase_upsampler_equals_conv_transpose():
    from oracle import ref_numpy as orc
    rng = np.random.Generator(np.random.PCG64(3))
    for s, cin, cout in ((8, 12, 8), (2, 8, 4)):
        w = rng.standard_normal((cin, cout, 2 * s)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        x = rng.standard_normal((2, cin, 13)).astype(np.float32)
        ref = orc.conv_transpose1d(x, w, b, s, s // 2)                      # [B, cout, 13 * s]
        w3, b3 = packer.polyphase_upsampler(w, b, s)                       # [s*cout, 3*cin]
        y = orc.conv1d(x, w3.reshape(s * cout, 3, cin).transpose(0, 2, 1), b3, padding=1)  # [B, s*cout, 13]
        y = y.reshape(2, s, cout, 13).transpose(0, 2, 3, 1).reshape(2, cout, 13 * s)
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5)


def test_blob_roundtrip_and_batchnorm_fold():
    cfg = cfgmod.tiny_config()
    stats = cfgmod.DEFAULT_STATS
    ac = sw.make_acoustic_state(cfg, stats, 4, mode="varied")
    voc = sw.make_vocoder_state(cfg)
    dims = cfgmod.dims_from_config(cfg, stats, 4)
    tensors = packer.pack_tensors(dims, ac, voc)
    blob = packer.build_blob(tensors)
    magic, version, n, data_off, total = np.frombuffer(blob[:8], "S8")[0], *np.frombuffer(blob[8:16], "<u4"), *np.frombuffer(blob[16:32], "<u8")
    assert magic == b"E2ETTSW1" and version == 1 and n == len(tensors) and total == blob.size and data_off % 256 == 0
    for i, (name, t) in enumerate(tensors.items()):
        ent = blob[32 + 80 * i: 32 + 80 * (i + 1)]
        assert ent[:64].tobytes().rstrip(b"\0").decode() == name
        off, numel = np.frombuffer(ent[64:80], "<u8")
        assert off % 256 == 0 and numel == t.size
        np.testing.assert_array_equal(blob[off:off + 4 * numel].view(np.float32), t.reshape(-1))
    # BN fold: conv + eval BatchNorm == folded conv (postnet layer 0)
    from oracle import ref_numpy as orc
    rng = np.random.Generator(np.random.PCG64(4))
    x = rng.standard_normal((1, 80, 9)).astype(np.float32)
    p = "postnet.convolutions.0"
    y = orc.conv1d(x, ac[p + ".0.conv.weight"], ac[p + ".0.conv.bias"], padding=2)
    y = (y - ac[p + ".1.running_mean"][None, :, None]) / np.sqrt(ac[p + ".1.running_var"][None, :, None] + 1e-5)
    y = y * ac[p + ".1.weight"][None, :, None] + ac[p + ".1.bias"][None, :, None]
    k = dims.postnet_kernel
    wf = tensors["post.0.w"].reshape(-1, k, 80).transpose(0, 2, 1)
    np.testing.assert_allclose(orc.conv1d(x, wf, tensors["post.0.b"], padding=2), y, rtol=0, atol=2e-5)
    # halves
    assert "voc.pre.w" not in packer.pack_tensors(dims, ac, None) and "enc.emb" not in packer.pack_tensors(dims, None, voc)


def test_config_rejects_unimplemented_variants():
    cfg = cfgmod.default_config()
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = "lstransformer"
    with pytest.raises(NotImplementedError):
        cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4)
    cfg["models"]["fastspeech2"]["building_block"]["block_type"] = "conformer"   # implemented: 8 relative-position heads, FFN x4, k31
    d = cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4)
    assert (d.block_type, d.n_head, d.ffn_dim, d.ffn_k1, d.ffn_k2, d.cf_ffn_factor) == (1, 8, 1536, 31, 1, 0.5)
    assert cfgmod.dims_from_config(cfgmod.default_config(), cfgmod.DEFAULT_STATS, 4).block_type == 0
    cfg = cfgmod.default_config()
    cfg["models"]["hifigan"]["resblock"] = 2   # ResBlock2: implemented (two dilations per kernel size)
    assert cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4).voc_rb_dil == [[1, 3]] * 3
    d = cfgmod.dims_from_config(cfgmod.default_config(), cfgmod.DEFAULT_STATS, 4, vocoder="istft")
    assert (d.voc_resblock, d.voc_istft_nfft, d.voc_istft_hop, d.voc_post_channels, d.upsample_total) == (2, 16, 4, 20, 256)
    cfg = cfgmod.default_config()
    cfg["models"]["istft"]["gen_istft_win_size"] = 8
    with pytest.raises(NotImplementedError):
        cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4, vocoder="istft")
    cfg = cfgmod.default_config()
    cfg["audio"]["stft"]["hop_length"] = 300
    with pytest.raises(ValueError):
        cfgmod.dims_from_config(cfg, cfgmod.DEFAULT_STATS, 4)


def _dynamic_symbols(path):
    """Defined symbols of the dynamic symbol table (`nm -D --defined-only`), i.e. everything a host could bind."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_c_abi_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/e2etts.h declares -- no C++ symbol, no test hook
    (VERDICT r3 item 3): built with -fvisibility=hidden, every entry point marked E2ETTS_API.  The test build adds the one hook."""
    from e2e_tts_amd import _lib
    import __graft_entry__ as g
    if g.built_hash() != g.source_hash() or g.built_hash(g.TEST_LIB) != g.source_hash():
        g.build()
    header = open(os.path.join(ROOT, "include", "e2etts.h")).read()
    hooks_block = re.search(r"#ifdef E2ETTS_TEST_HOOKS\n(.*?)#endif", header, re.S).group(1)
    hook_syms = sorted(set(re.findall(r"\b(e2etts_[a-z0-9_]+)\s*\(", hooks_block)))
    declared_all = sorted(set(re.findall(r"^E2ETTS_API [^;]*?\b(e2etts_[a-z0-9_]+)\s*\(", header, re.M)))
    declared = [d for d in declared_all if d not in hook_syms]
    assert hook_syms == sorted(_lib.TEST_HOOK_SYMBOLS)
    assert declared == sorted(_lib.EXPORTED_SYMBOLS)
    # every function the header mentions carries the export macro (nothing declared but hidden)
    assert sorted(set(re.findall(r"^(?:E2ETTS_API )?(?:const char\*|int|void\*?|size_t) (e2etts_[a-z0-9_]+)\s*\(", header, re.M))) == declared_all
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    exported = [s for s in _dynamic_symbols(_lib.LIB_PATH) if not s.startswith(("_init", "_fini", "__"))]
    assert exported == declared, sorted(set(exported) ^ set(declared))
    exported_test = [s for s in _dynamic_symbols(_lib.TEST_LIB_PATH) if not s.startswith(("_init", "_fini", "__"))]
    assert exported_test == sorted(declared + hook_syms)


def test_config_struct_mirror_matches_the_header_and_the_library():
    """e2etts_config is mirrored by hand in config.CEngineConfig: its size must be what the LIBRARY reports (e2etts_config_size, the
    C compiler's sizeof), its ABI version the header's, and its fields the header's fields in the header's order."""
    from e2e_tts_amd import _lib
    import __graft_entry__ as g
    if g.built_hash() != g.source_hash():
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.e2etts_config_size.restype = ctypes.c_size_t
    assert lib.e2etts_config_size() == ctypes.sizeof(cfgmod.CEngineConfig) == 320
    header = open(os.path.join(ROOT, "include", "e2etts.h")).read()
    assert lib.e2etts_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define E2ETTS_ABI_VERSION (\d+)", header).group(1))
    body = re.search(r"typedef struct e2etts_config \{(.*?)\} e2etts_config;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        assert typ in ("int32_t", "uint32_t", "float"), decl
        fields += [re.sub(r"\[.*", "", n.strip()) for n in names.split(",")]
    assert fields == [f[0] for f in cfgmod.CEngineConfig._fields_]
    assert cfgmod.CEngineConfig._fields_[0][0] == "struct_size" and cfgmod.CEngineConfig.struct_size.offset == 0
    dims = cfgmod.dims_from_config(cfgmod.tiny_config(), cfgmod.DEFAULT_STATS, 4)
    assert dims.to_c().struct_size == 320


def test_create_rejects_a_config_of_unknown_size():
    """A host compiled against another revision of the header (struct_size not this library's) gets E2ETTS_EINVAL and a message, before
    any other field is read and before a GPU is looked for."""
    from e2e_tts_amd import _lib
    import __graft_entry__ as g
    if g.built_hash() != g.source_hash():
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.e2etts_last_error.restype = ctypes.c_char_p
    lib.e2etts_last_error.argtypes = [ctypes.c_void_p]
    dims = cfgmod.dims_from_config(cfgmod.tiny_config(), cfgmod.DEFAULT_STATS, 4)
    for size in (0, 316, 324, 4096):   # 316 = round 3's struct (no struct_size field: its n_symbols = 131 would be read as a size, too)
        c = dims.to_c()
        c.struct_size = size
        h = ctypes.c_void_p()
        assert lib.e2etts_create(0, ctypes.byref(c), ctypes.byref(h)) == _lib.E_INVAL
        assert not h.value
        msg = lib.e2etts_last_error(None).decode()
        assert "struct_size" in msg and str(size) in msg and "320" in msg, msg


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from e2e_tts_amd._lib import Engine
    dims = cfgmod.dims_from_config(cfgmod.tiny_config(), cfgmod.DEFAULT_STATS, 4)
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        Engine(dims, 0)


def test_wsola_time_stretch_and_wav_io(tmp_path):
    """audio_speed_change (reference API/utils.py:163-172 shells out to ffmpeg atempo): duration scales by 1/speed, pitch
    stays, naming follows the reference."""
    from e2e_tts_amd.api import audio_speed_change, read_wav, time_stretch_wsola, write_wav
    sr = 22050
    t = np.arange(2 * sr) / sr
    x = 8000.0 * np.sin(2 * np.pi * 220.0 * t) * (1 + 0.3 * np.sin(2 * np.pi * 3 * t))
    np.testing.assert_array_equal(time_stretch_wsola(x, 1.0, sr), x)
    for speed in (0.5, 0.8, 1.25, 2.0):
        y = time_stretch_wsola(x, speed, sr)
        assert y.size == round(x.size / speed)
        spec = np.abs(np.fft.rfft(y * np.hanning(y.size)))
        assert abs(np.fft.rfftfreq(y.size, 1 / sr)[int(np.argmax(spec))] - 220.0) < 2.0
        assert 0.8 < np.abs(y).max() / np.abs(x).max() < 1.2
    with pytest.raises(ValueError):
        time_stretch_wsola(x, 8.0, sr)
    src = str(tmp_path / "a.wav")
    write_wav(src, x.astype(np.int16), sr)
    back, sr2 = read_wav(src)
    np.testing.assert_array_equal(back, x.astype(np.int16))
    out = audio_speed_change(src, speed_rate=1.5)
    assert out == str(tmp_path / "a_1.5.wav") and sr2 == sr
    assert read_wav(out)[0].size == round(x.size / 1.5)


def test_speed_paths_exist_and_facade_keeps_the_reference_path_by_default(tmp_path):
    """Synthesizer.synthesis with speed != 1 (reference API/inference.py:44-49 writes the file, THEN derives <file>_<speed>.wav from
    it): both paths exist afterwards, the returned one carries the tempo; the top-level facade hands on the path IT generated, as the
    reference does, or the returned one with use_returned_path=True."""
    from e2e_tts_amd import api, synthesizer as top

    class StubTTS:  # stands in for the engine-backed TTS: the file handling is what is under test
        calls = []

        def inference(self, texts, speaker_id, pitch_control, energy_control, duration_control, silence_distance):
            StubTTS.calls.append(duration_control)
            n = int(22050 * 0.2 * duration_control)
            return (np.arange(n) % 100).astype(np.int16)

    s = api.Synthesizer.__new__(api.Synthesizer)
    s.model, s.output_dir = StubTTS(), str(tmp_path)
    want = str(tmp_path / "x.wav")
    got = s.synthesis("xin chao", want, speed=1.25)
    assert got == str(tmp_path / "x_1.25.wav") and os.path.exists(got) and os.path.exists(want)
    assert StubTTS.calls[-1] == pytest.approx(0.8)          # tempo applied in the model: duration_control = 1 / speed
    np.testing.assert_array_equal(api.read_wav(got)[0], api.read_wav(want)[0])
    got = s.synthesis("xin chao", str(tmp_path / "y.wav"), speed=1.25, speed_mode="wsola")
    assert got == str(tmp_path / "y_1.25.wav") and os.path.exists(got) and os.path.exists(str(tmp_path / "y.wav"))
    assert api.read_wav(got)[0].size == round(api.read_wav(str(tmp_path / "y.wav"))[0].size / 1.25)
    assert s.synthesis("xin chao", str(tmp_path / "z.wav"), speed=1) == str(tmp_path / "z.wav")

    f = top.Synthesizer.__new__(top.Synthesizer)
    f.output_dir = str(tmp_path)
    f.model_dict = {"vie": s}
    # default: the reference's own behaviour (synthesizer.py:47 ignores tts_to_file's return value): the path it generated comes back
    path, vc = f.synthesis("xin chao", "vie Vietnamese", speed=1.5)
    assert vc is None and not path.endswith("_1.5.wav") and os.path.exists(path)
    sped = path[:-4] + "_1.5.wav"
    assert os.path.exists(sped)                                  # the service wrote it next to it, as the reference does
    np.testing.assert_array_equal(api.read_wav(path)[0], api.read_wav(sped)[0])   # duration mode: both carry the tempo
    # opt-in: hand on the path tts_to_file reports
    f.use_returned_path = True
    path, vc = f.synthesis("xin chao", "vie Vietnamese", speed=1.25)
    assert vc is None and path.endswith("_1.25.wav") and os.path.exists(path)
    path, _ = f.synthesis("xin chao", "vie Vietnamese", speed=1.0)
    assert os.path.exists(path) and not path.endswith("_1.0.wav")


def test_binding_refuses_wrong_dtypes_and_short_buffers():
    """_lib hands raw addresses to C, which reads B * L * 8 bytes of ids etc.: dtype / element-count mismatches are refused in the
    binding (no GPU needed: the checks run before any C call)."""
    import torch
    from e2e_tts_amd import _lib
    ok = np.zeros((2, 5), np.int64)
    _lib._expect(ok, "ids", "int64", 10)
    _lib._expect(torch.zeros(2, 5, dtype=torch.int64), "ids", "int64", 10)
    _lib._expect(None, "ids", "int64", 10)
    with pytest.raises(TypeError, match="dtype int32"):
        _lib._expect(ok.astype(np.int32), "ids", "int64", 10)
    with pytest.raises(TypeError, match="dtype float64"):
        _lib._expect(torch.zeros(3, dtype=torch.float64), "mel", "float32", 3)
    with pytest.raises(ValueError, match="9 elements"):
        _lib._expect(np.zeros(9, np.int64), "lens", "int64", 10)
    with pytest.raises(ValueError, match="12 elements, expected 10"):
        _lib._expect(np.zeros(12, np.int64), "lens", "int64", 10)
    _lib._expect(np.zeros(12, np.int16), "out_pcm", "int16", 10, at_least=True)
    with pytest.raises(TypeError):
        _lib._expect([1, 2, 3], "ids", "int64", 3)

    class E(_lib.Engine):  # the argument checks of the public methods, reached without constructing an engine
        def __init__(self):
            import threading
            self.lock = threading.RLock()
            self.dims = cfgmod.dims_from_config(cfgmod.tiny_config(), cfgmod.DEFAULT_STATS, 4)
            self.device = 0

        def __del__(self):
            pass

    e = E()
    ids, lens, spk = np.zeros((2, 5), np.int64), np.full((2,), 5, np.int64), np.array([1], np.int64)
    with pytest.raises(TypeError, match="ids: dtype int32"):
        e.synthesize(ids.astype(np.int32), lens, spk)
    with pytest.raises(ValueError, match="lens: 1 elements, expected 2"):
        e.acoustic(ids, lens[:1], spk)
    with pytest.raises(ValueError, match="speaker holds 3 ids"):
        e.acoustic(ids, lens, np.array([1, 1, 1], np.int64))
    with pytest.raises(TypeError, match="mel: dtype float64"):
        e.vocoder(np.zeros((1, e.dims.n_mel, 4)), 1, 4)
    with pytest.raises(ValueError, match="out_wav"):
        e.vocoder(np.zeros((1, e.dims.n_mel, 4), np.float32), 1, 4, out_wav=np.zeros(7, np.float32))
