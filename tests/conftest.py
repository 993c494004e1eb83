import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")

# The tests load the TEST build of the library (libe2etts_hip_test.so: the product sources + the e2etts_debug_poison_workspace hook that the
# ragged-mode tests need); child processes inherit the choice.  tests/test_gpu_dropin.py: test_product_library_gives_the_test_builds_bits
# runs the product library next to it.
os.environ.setdefault("E2ETTS_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


def load_golden(name):
    path = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name} not generated")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def config_for(name):
    from e2e_tts_amd import config as cfgmod
    cfg = cfgmod.tiny_config() if name.startswith("tiny") or name.startswith("voc_micro") else cfgmod.default_config()
    if "_cf_" in name:  # fixtures made with building_block.block_type = "conformer" (oracle/make_goldens.py)
        cfg["models"]["fastspeech2"]["building_block"]["block_type"] = "conformer"
    if "_hv_" in name:  # decoder_head != encoder_head, energy predictor of its own depth / kernel (oracle/make_goldens.py: hv_variant)
        from oracle.make_goldens import hv_variant
        cfg = hv_variant(cfg)
    for tag in ("nouv", "plog", "lpad", "frame", "pframe", "eframe"):  # use_uv False / pitch_quantization "log" / ffn_padding "LEFT" / frame_level features (oracle/make_goldens.py: pv_variant)
        if f"_{tag}_" in name:
            from oracle.make_goldens import pv_variant
            cfg = pv_variant(cfg, tag)
    return cfg


_STATE_CACHE = {}


def states_for(g, name):
    """Regenerate the synthetic state dicts a fixture was made with (seeds + mode are stored in it)."""
    from e2e_tts_amd import config as cfgmod, synth_weights as sw
    cfg = config_for(name)
    key = (name.startswith("tiny"), "_cf_" in name, "_hv_" in name, "_nouv_" in name, "_plog_" in name, "_lpad_" in name, "_frame_" in name, "_pframe_" in name, "_eframe_" in name, str(g["mode"]), tuple(int(x) for x in g["weight_seeds"]))
    if key not in _STATE_CACHE:
        ac = sw.make_acoustic_state(cfg, cfgmod.DEFAULT_STATS, 4, seed=int(g["weight_seeds"][0]), mode=str(g["mode"]))
        voc = sw.make_vocoder_state(cfg, seed=int(g["weight_seeds"][1]))
        _STATE_CACHE[key] = (ac, voc)
    return (cfg,) + _STATE_CACHE[key]
