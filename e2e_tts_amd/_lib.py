"""ctypes binding of include/e2etts.h (libe2etts_hip.so).  There is no CPU fallback: if the
library is missing or no GPU is visible, construction raises."""
from __future__ import annotations

import collections
import ctypes as C
import functools
import threading
import os
from typing import Optional

import numpy as np

from .config import CEngineConfig, EngineDims

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libe2etts_hip.so")
# The TEST build of the same sources (-DE2ETTS_TEST_HOOKS: one more export, e2etts_debug_poison_workspace).  Loaded instead of the product
# library only when E2ETTS_TEST_HOOKS=1 is in the environment (tests/conftest.py sets it); nothing in the product path asks for it.
TEST_LIB_PATH = os.path.join(_HERE, "lib", "libe2etts_hip_test.so")
ABI_VERSION = 4   # E2ETTS_ABI_VERSION of the include/e2etts.h this binding mirrors

E_OK, E_INVAL, E_HIP, E_STATE, E_NOMEM, E_KEY = 0, -1, -2, -3, -4, -5


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_lib = None


def load_library() -> C.CDLL:
    """dlopen the in-tree HIP library (built by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    hooks = os.environ.get("E2ETTS_TEST_HOOKS", "") not in ("", "0")
    path = TEST_LIB_PATH if hooks else LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  e2e_tts_amd has no CPU fallback.")
    # ONE HIP runtime per process: torch ships its own libamdhip64.so (soname libamdhip64.so.7) and resolves it by file name, so if this
    # library -- which needs "libamdhip64.so.7" -- were loaded first it would pull in /opt/rocm's copy and a later `import torch` would
    # bring a second runtime that finds no GPU ("No HIP GPUs are available").  With torch loaded first the soname matches its copy.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    P, I, F, SZ = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.e2etts_version.restype = C.c_char_p
    lib.e2etts_version.argtypes = []
    # ABI guard before anything else is bound: the library's header revision and its sizeof(e2etts_config) must be the ones this
    # binding's hand-written mirror (config.CEngineConfig) was written for -- a stale .so or a drifted mirror fails here, loudly
    try:
        lib.e2etts_abi_version.restype = I
        lib.e2etts_abi_version.argtypes = []
        lib.e2etts_config_size.restype = SZ
        lib.e2etts_config_size.argtypes = []
    except AttributeError as ex:
        raise ImportError(f"{path} predates the ABI guard (no e2etts_abi_version / e2etts_config_size): rebuild it") from ex
    if lib.e2etts_abi_version() != ABI_VERSION or lib.e2etts_config_size() != C.sizeof(CEngineConfig):
        raise ImportError(f"{path}: ABI version {lib.e2etts_abi_version()} / sizeof(e2etts_config) {lib.e2etts_config_size()}, this binding "
                          f"mirrors version {ABI_VERSION} / {C.sizeof(CEngineConfig)} bytes: rebuild the library or update config.CEngineConfig")
    lib.e2etts_last_error.restype = C.c_char_p
    lib.e2etts_last_error.argtypes = [P]
    lib.e2etts_create.restype = I
    lib.e2etts_create.argtypes = [I, C.POINTER(CEngineConfig), C.POINTER(P)]
    lib.e2etts_destroy.restype = None
    lib.e2etts_destroy.argtypes = [P]
    lib.e2etts_load_weights.restype = I
    lib.e2etts_load_weights.argtypes = [P, P, SZ]
    lib.e2etts_load_weights_bcast.restype = I
    lib.e2etts_load_weights_bcast.argtypes = [P, P, SZ, P, I]
    lib.e2etts_order_after.restype = I
    lib.e2etts_order_after.argtypes = [P, P]
    lib.e2etts_acoustic.restype = I
    lib.e2etts_acoustic.argtypes = [P, P, P, I, I, P, I, F, F, F, P, P, C.POINTER(I), P, P, P, P, P]
    lib.e2etts_fetch_mel.restype = I
    lib.e2etts_fetch_mel.argtypes = [P, P, P]
    lib.e2etts_fetch_tap.restype = I
    lib.e2etts_fetch_tap.argtypes = [P, C.c_char_p, P, SZ]
    lib.e2etts_fetch_tap_i32.restype = I
    lib.e2etts_fetch_tap_i32.argtypes = [P, C.c_char_p, P, SZ]
    lib.e2etts_vocoder.restype = I
    lib.e2etts_vocoder.argtypes = [P, P, I, I, P, P]
    lib.e2etts_vocoder_btc.restype = I
    lib.e2etts_vocoder_btc.argtypes = [P, P, I, I, P, P]
    lib.e2etts_synthesize.restype = I
    lib.e2etts_synthesize.argtypes = [P, P, P, I, I, P, I, F, F, F, P, SZ, P, C.POINTER(I)]
    lib.e2etts_fetch_pcm.restype = I
    lib.e2etts_fetch_pcm.argtypes = [P, P, SZ]
    lib.e2etts_fetch_wav.restype = I
    lib.e2etts_fetch_wav.argtypes = [P, P, SZ]
    lib.e2etts_vocoder_stream_begin.restype = I
    lib.e2etts_vocoder_stream_begin.argtypes = [P, I]
    lib.e2etts_vocoder_stream_push.restype = I
    lib.e2etts_vocoder_stream_push.argtypes = [P, P, I, I, C.POINTER(I)]
    lib.e2etts_vocoder_stream_fetch.restype = I
    lib.e2etts_vocoder_stream_fetch.argtypes = [P, P, P, SZ]
    lib.e2etts_tempo.restype = I
    lib.e2etts_tempo.argtypes = [P, P, SZ, C.c_double, I, P, SZ, C.POINTER(SZ)]
    lib.e2etts_set_precision.restype = I
    lib.e2etts_set_precision.argtypes = [P, I, I]
    lib.e2etts_set_ragged.restype = I
    lib.e2etts_set_ragged.argtypes = [P, I]
    if hooks:
        lib.e2etts_debug_poison_workspace.restype = I
        lib.e2etts_debug_poison_workspace.argtypes = [P]
    lib.e2etts_set_fused_resblocks.restype = I
    lib.e2etts_set_fused_resblocks.argtypes = [P, I]
    lib.e2etts_profile_enable.restype = I
    lib.e2etts_profile_enable.argtypes = [P, I]
    lib.e2etts_profile_filter.restype = I
    lib.e2etts_profile_filter.argtypes = [P, C.c_char_p]
    lib.e2etts_profile_read.restype = I
    lib.e2etts_profile_read.argtypes = [P, C.POINTER(KernelStat), I]
    lib.e2etts_device_bytes.restype = SZ
    lib.e2etts_device_bytes.argtypes = [P]
    lib.e2etts_stream.restype = P
    lib.e2etts_stream.argtypes = [P]
    lib.e2etts_sync.restype = I
    lib.e2etts_sync.argtypes = [P]
    _lib = lib
    return lib


# every entry point include/e2etts.h declares for the product library -- and, with -fvisibility=hidden, ALL it exports
# (tests/test_host_logic.py compares this list with the header and with `nm -D`); the test build adds TEST_HOOK_SYMBOLS
TEST_HOOK_SYMBOLS = ["e2etts_debug_poison_workspace"]
EXPORTED_SYMBOLS = [
    "e2etts_version", "e2etts_abi_version", "e2etts_config_size", "e2etts_last_error", "e2etts_create", "e2etts_destroy", "e2etts_load_weights", "e2etts_acoustic",
    "e2etts_fetch_mel", "e2etts_fetch_tap", "e2etts_fetch_tap_i32", "e2etts_vocoder", "e2etts_vocoder_btc", "e2etts_synthesize", "e2etts_fetch_pcm",
    "e2etts_fetch_wav", "e2etts_vocoder_stream_begin", "e2etts_vocoder_stream_push", "e2etts_vocoder_stream_fetch",
    "e2etts_set_precision", "e2etts_set_ragged", "e2etts_set_fused_resblocks", "e2etts_profile_enable", "e2etts_profile_filter", "e2etts_profile_read", "e2etts_device_bytes", "e2etts_stream", "e2etts_sync",
    "e2etts_load_weights_bcast", "e2etts_order_after", "e2etts_tempo",
]


def _addr(x) -> Optional[int]:
    """Raw address of a numpy array (host) or torch tensor (host or HBM); None stays NULL."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    raise TypeError(f"cannot take the address of {type(x)}")


def _is_cuda(x) -> bool:
    return x is not None and not isinstance(x, np.ndarray) and bool(getattr(x, "is_cuda", False))


def _expect(x, name: str, dtype: str, count: int, at_least: bool = False):
    """The C side reads / writes `count` elements of `dtype` behind the raw address: a wrong dtype or a short buffer would be an
    out-of-bounds access there, so it is refused here (numpy arrays and torch tensors alike)."""
    if x is None:
        return
    if isinstance(x, np.ndarray):
        dt, n = x.dtype.name, x.size
    elif hasattr(x, "data_ptr"):
        dt, n = str(x.dtype).replace("torch.", ""), x.numel()
    else:
        raise TypeError(f"{name}: expected a numpy array or torch tensor, got {type(x)}")
    if dt != dtype:
        raise TypeError(f"{name}: dtype {dt}, expected {dtype}")
    if n < count or (not at_least and n != count):
        raise ValueError(f"{name}: {n} elements, expected {'at least ' if at_least else ''}{count}")


def _locked(fn):
    """Run a method under the engine's re-entrant lock.  Every C entry point takes the engine's mutex, but several results are
    RESIDENT (`synthesize` then `fetch_pcm`, `acoustic` then `fetch_mel` / `vocoder(None)`): a method that is a sequence of C calls,
    and the error text read after a failing one, must not interleave with another thread's calls on the same engine."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        with self.lock:
            return fn(self, *a, **k)
    return wrapper


class Engine:
    """One GPU, one stream, one set of weights.  Thin, typed wrapper over the C ABI.

    Thread safety: every method is atomic with respect to the other methods of the same Engine (`self.lock`, an RLock).  A caller that
    relies on resident results ACROSS methods from several threads -- `acoustic()` ... `fetch_mel()` ... `vocoder(None, ...)` -- holds
    `with engine.lock:` around the sequence, as the model mirrors do.  Distinct engines are independent."""

    def __init__(self, dims: EngineDims, device: int = 0):
        self.lock = threading.RLock()
        self.lib = load_library()
        self.dims = dims
        self._h = C.c_void_p()
        cfg = dims.to_c()
        rc = self.lib.e2etts_create(int(device), C.byref(cfg), C.byref(self._h))
        if rc != E_OK:
            msg = self.lib.e2etts_last_error(None).decode()
            self._h = C.c_void_p()
            raise (ValueError if rc == E_INVAL else RuntimeError)(f"e2etts_create: {msg}")
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.e2etts_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc >= 0:
            return rc
        msg = f"{what}: {self.lib.e2etts_last_error(self._h).decode()}"
        if rc == E_INVAL:
            raise ValueError(msg)
        if rc == E_KEY:
            raise KeyError(msg)
        if rc == E_NOMEM:
            raise MemoryError(msg)
        raise RuntimeError(msg)

    def _order(self, *xs):
        """Stream ordering of device buffers (include/e2etts.h, "STREAM ORDERING"): torch may still have kernels queued on its current
        stream that write an input tensor (a `.contiguous()` copy, the caller's own producer) or read a recycled allocator block handed
        in as an output.  The engine's stream is made to wait for that stream before the C call touches the buffers."""
        cuda = [x for x in xs if _is_cuda(x)]
        if not cuda:
            return
        import torch
        for x in cuda:
            if x.device.index != self.device:
                raise ValueError(f"tensor on {x.device}, engine on cuda:{self.device}")
        st = torch.cuda.current_stream(cuda[0].device).cuda_stream
        self._check(self.lib.e2etts_order_after(self._h, C.c_void_p(st)), "e2etts_order_after")

    # ---- weights
    @_locked
    def load_weights(self, blob) -> None:
        """blob: uint8 numpy array (host) or uint8 torch tensor (host / HBM, e.g. after a RCCL broadcast)."""
        n = blob.nbytes if isinstance(blob, np.ndarray) else blob.numel() * blob.element_size()
        _expect(blob, "blob", "uint8", n)
        self._order(blob)
        self._check(self.lib.e2etts_load_weights(self._h, _addr(blob), n), "e2etts_load_weights")

    @_locked
    def load_weights_bcast(self, blob, nbytes: int, rccl_comm: int, root: int = 0) -> None:
        """Collective: one RCCL broadcast of the packed image from `root` into this engine's HBM, then bind (include/e2etts.h).
        `rccl_comm` is a raw ncclComm_t (an integer address) of the RCCL copy loaded in this process; `blob` may be None off-root."""
        if blob is not None:
            _expect(blob, "blob", "uint8", nbytes)
            self._order(blob)
        self._check(self.lib.e2etts_load_weights_bcast(self._h, _addr(blob), int(nbytes), C.c_void_p(rccl_comm), int(root)),
                    "e2etts_load_weights_bcast")

    # ---- acoustic model
    @_locked
    def acoustic(self, ids, lens, speaker, d_control=1.0, p_control=1.0, e_control=1.0, want=("dur", "mel_lens")):
        """ids [B, L] int64, lens [B] int64, speaker [1 or B] int64 (numpy or torch, host or device).
        Returns dict with T and the requested host arrays."""
        B, L = int(ids.shape[0]), int(ids.shape[1])
        n_spk = int(speaker.shape[0])
        _expect(ids, "ids", "int64", B * L)
        _expect(lens, "lens", "int64", B)
        _expect(speaker, "speaker", "int64", n_spk)
        if n_spk not in (1, B):
            raise ValueError(f"speaker holds {n_spk} ids, expected 1 or {B}")
        self._order(ids, lens, speaker)
        out = {}
        pf, ef = bool(self.dims.pitch_frame), bool(self.dims.energy_frame)   # frame_level features: T columns, fetched after the call (below)
        bufs = dict(
            dur=np.empty((B, L), np.float32) if "dur" in want else None,
            mel_lens=np.empty((B,), np.int64) if "mel_lens" in want else None,
            pitch_idx=np.empty((B, L), np.int32) if "pitch_idx" in want and not pf else None,
            energy_idx=np.empty((B, L), np.int32) if "energy_idx" in want and not ef else None,
            log_d=np.empty((B, L), np.float32) if "log_d" in want else None,
            pitch_pred=np.empty((B, L) if self.dims.pitch_no_uv else (B, L, 2), np.float32) if "pitch_pred" in want and not pf else None,
            energy_pred=np.empty((B, L), np.float32) if "energy_pred" in want and not ef else None,
        )
        T = C.c_int(0)
        rc = self.lib.e2etts_acoustic(self._h, _addr(ids), _addr(lens), B, L, _addr(speaker), n_spk,
                                      float(d_control), float(p_control), float(e_control),
                                      _addr(bufs["dur"]), _addr(bufs["mel_lens"]), C.byref(T), _addr(bufs["pitch_idx"]),
                                      _addr(bufs["energy_idx"]), _addr(bufs["log_d"]), _addr(bufs["pitch_pred"]),
                                      _addr(bufs["energy_pred"]))
        self._check(rc, "e2etts_acoustic")
        out.update({k: v for k, v in bufs.items() if v is not None})
        out["T"] = T.value
        out["B"] = B
        for name, frame, shape, dt in (("pitch_idx", pf, (B, T.value), np.int32), ("energy_idx", ef, (B, T.value), np.int32),
                                       ("pitch_pred", pf, (B, T.value) if self.dims.pitch_no_uv else (B, T.value, 2), np.float32),
                                       ("energy_pred", ef, (B, T.value), np.float32)):
            if frame and name in want:
                a = np.empty(shape, dt)
                fn = self.lib.e2etts_fetch_tap_i32 if dt is np.int32 else self.lib.e2etts_fetch_tap
                self._check(fn(self._h, name.encode(), _addr(a), a.size), "e2etts_fetch_tap")
                out[name] = a
        return out

    @_locked
    def fetch_mel(self, B: int, T: int, mel=True, mel_post=True, out_mel=None, out_mel_post=None):
        m = out_mel if out_mel is not None else (np.empty((B, T, self.dims.n_mel), np.float32) if mel else None)
        mp = out_mel_post if out_mel_post is not None else (np.empty((B, T, self.dims.n_mel), np.float32) if mel_post else None)
        _expect(m, "mel", "float32", B * T * self.dims.n_mel)
        _expect(mp, "mel_post", "float32", B * T * self.dims.n_mel)
        self._order(m, mp)
        self._check(self.lib.e2etts_fetch_mel(self._h, _addr(m), _addr(mp)), "e2etts_fetch_mel")
        return m, mp

    @_locked
    def fetch_tap(self, which: str, shape) -> np.ndarray:
        out = np.empty(shape, np.float32)
        self._check(self.lib.e2etts_fetch_tap(self._h, which.encode(), _addr(out), out.size), "e2etts_fetch_tap")
        return out

    @_locked
    def fetch_tap_into(self, which: str, out) -> None:
        """Like fetch_tap, into a caller buffer (numpy array or torch tensor, host or HBM)."""
        n = out.size if isinstance(out, np.ndarray) else out.numel()
        _expect(out, "out", "float32", n)
        self._order(out)
        self._check(self.lib.e2etts_fetch_tap(self._h, which.encode(), _addr(out), n), "e2etts_fetch_tap")

    # ---- vocoder
    @_locked
    def vocoder(self, mel, B: int, T: int, channels_first=True, wav=True, pcm=False, out_wav=None, out_pcm=None):
        """mel: [B, n_mel, T] (channels_first) or [B, T, n_mel], or None for the resident mel_post."""
        n = B * T * self.dims.hop_length
        w = out_wav if out_wav is not None else (np.empty((B, T * self.dims.hop_length), np.float32) if wav else None)
        p = out_pcm if out_pcm is not None else (np.empty((B, T * self.dims.hop_length), np.int16) if pcm else None)
        _expect(mel, "mel", "float32", B * T * self.dims.n_mel)
        _expect(w, "out_wav", "float32", n)
        _expect(p, "out_pcm", "int16", n)
        self._order(mel, w, p)
        fn = self.lib.e2etts_vocoder if channels_first else self.lib.e2etts_vocoder_btc
        self._check(fn(self._h, _addr(mel), B, T, _addr(w), _addr(p)), "e2etts_vocoder")
        return w, p

    # ---- end to end
    @_locked
    def synthesize(self, ids, lens, speaker, d_control=1.0, p_control=1.0, e_control=1.0, fetch_pcm=True,
                   out_pcm=None, out_mel_lens=None):
        """One batch of TTS.inference: returns (pcm [B, T*hop] int16 or None, mel_lens [B], T)."""
        B, L = int(ids.shape[0]), int(ids.shape[1])
        mel_lens = out_mel_lens if out_mel_lens is not None else np.empty((B,), np.int64)
        T = C.c_int(0)
        cap = 0
        if out_pcm is not None:
            cap = out_pcm.size if isinstance(out_pcm, np.ndarray) else out_pcm.numel()
            _expect(out_pcm, "out_pcm", "int16", cap)
        n_spk = int(speaker.shape[0])
        _expect(ids, "ids", "int64", B * L)
        _expect(lens, "lens", "int64", B)
        _expect(speaker, "speaker", "int64", n_spk)
        _expect(mel_lens, "out_mel_lens", "int64", B)
        if n_spk not in (1, B):
            raise ValueError(f"speaker holds {n_spk} ids, expected 1 or {B}")
        self._order(ids, lens, speaker, out_pcm, mel_lens)
        rc = self.lib.e2etts_synthesize(self._h, _addr(ids), _addr(lens), B, L, _addr(speaker), int(speaker.shape[0]),
                                        float(d_control), float(p_control), float(e_control), _addr(out_pcm), cap,
                                        _addr(mel_lens), C.byref(T))
        self._check(rc, "e2etts_synthesize")
        pcm = out_pcm
        if out_pcm is None and fetch_pcm:
            pcm = np.empty((B, T.value * self.dims.hop_length), np.int16)
            self._check(self.lib.e2etts_fetch_pcm(self._h, _addr(pcm), pcm.size), "e2etts_fetch_pcm")
        return pcm, mel_lens, T.value

    @_locked
    def fetch_wav(self, B: int, T: int) -> np.ndarray:
        w = np.empty((B, T * self.dims.hop_length), np.float32)
        self._check(self.lib.e2etts_fetch_wav(self._h, _addr(w), w.size), "e2etts_fetch_wav")
        return w

    @_locked
    def tempo(self, pcm: np.ndarray, speed: float, sample_rate: int = 22050) -> np.ndarray:
        """Tempo change without pitch change of an int16 signal on the GPU (WSOLA; include/e2etts.h: e2etts_tempo -- parity unpinned,
        the reference shells out to ffmpeg).  Returns round(len / speed) samples."""
        pcm = np.ascontiguousarray(pcm)
        _expect(pcm, "pcm", "int16", pcm.size)
        n_out = C.c_size_t(0)
        self._check(self.lib.e2etts_tempo(self._h, _addr(pcm), pcm.size, float(speed), int(sample_rate), None, 0, C.byref(n_out)), "e2etts_tempo")
        out = np.empty((n_out.value,), np.int16)
        self._check(self.lib.e2etts_tempo(self._h, _addr(pcm), pcm.size, float(speed), int(sample_rate), _addr(out), out.size, C.byref(n_out)),
                    "e2etts_tempo")
        return out

    @_locked
    def set_precision(self, vocoder: str = "fp32", decoder: Optional[str] = None):
        """'fp32' (exact fp32 MFMA: the engine's default and the reference's arithmetic) or 'bf16x3' (split-precision bf16 MFMA: the
        opt-in fast mode, PCM within 1 LSB of the reference's) for the vocoder and for the decoder + mel_linear + postnet (defaults
        to the vocoder's choice).  Encoder / variance adaptor: always fp32."""
        modes = {"fp32": 0, "bf16x3": 1, "bf16": 2}   # 'bf16' (plain, vocoder only) is the long-form streaming config's arithmetic
        dec = decoder if decoder is not None else ("bf16x3" if vocoder == "bf16" else vocoder)
        self._check(self.lib.e2etts_set_precision(self._h, modes[vocoder], modes[dec]), "e2etts_set_precision")

    @_locked
    def set_ragged(self, on: bool = True):
        """synthesize(): skip the rows of shorter utterances that no valid sample depends on (default on)."""
        self._check(self.lib.e2etts_set_ragged(self._h, 1 if on else 0), "e2etts_set_ragged")

    @_locked
    def poison_workspace(self):
        """Test hook (test build of the library only, E2ETTS_TEST_HOOKS=1): fill the activation workspaces with a large finite pattern
        (ragged-mode tests: nothing valid may depend on stale rows)."""
        if not hasattr(self.lib, "e2etts_debug_poison_workspace"):
            raise RuntimeError("e2etts_debug_poison_workspace is a test hook: set E2ETTS_TEST_HOOKS=1 before the library is loaded "
                               "(libe2etts_hip_test.so); the product library does not export it")
        self._check(self.lib.e2etts_debug_poison_workspace(self._h), "e2etts_debug_poison_workspace")

    @_locked
    def set_fused_resblocks(self, on=True):
        """bf16 modes.  True / 2 (default): ResBlock conv pairs at 32 .. 256 channels as one kernel each, and whole k = 3 ResBlocks at
        32 / 64 channels as one kernel; 1: pairs only; False / 0: two convolution launches per pair.  Results are bit-identical."""
        level = 2 if on is True else (0 if on is False else int(on))
        self._check(self.lib.e2etts_set_fused_resblocks(self._h, level), "e2etts_set_fused_resblocks")

    # ---- long-form / streaming vocoder
    def vocoder_stream(self, chunks, B: int, want_pcm: bool = False):
        """Generator: feed an iterable of mel chunks [B, n, n_mel] (numpy / torch, channels-last), yield the waveform (or
        int16 PCM) pieces [B, n_emit * hop] as they become final.  Concatenated along axis 1 they equal
        ``vocoder(whole_mel)`` bit for bit, while HBM use stays bounded by the chunk size."""
        with self.lock:
            halo = self._check(self.lib.e2etts_vocoder_stream_begin(self._h, B), "e2etts_vocoder_stream_begin")
        self.stream_halo = halo
        hop = self.dims.hop_length

        def fetch(n_emit):   # the OLDEST unfetched chunk (the engine keeps at most two in flight)
            out = np.empty((B, n_emit * hop), np.int16 if want_pcm else np.float32)
            self._check(self.lib.e2etts_vocoder_stream_fetch(self._h, None if want_pcm else _addr(out), _addr(out) if want_pcm else None,
                                                             out.size), "e2etts_vocoder_stream_fetch")
            return out

        chunks = iter(chunks)
        cur = next(chunks, None)
        pend = collections.deque()   # (frames made final, the chunk: device memory must outlive the copy the push enqueued)
        while cur is not None:
            nxt = next(chunks, None)
            n_emit = C.c_int(0)
            out = None
            with self.lock:  # one step: push chunk i (returns once enqueued), then take chunk i - 1's samples while i computes.  The
                # lock is NOT held across the yield below; the stream's context and its two output slots live in buffers of their own,
                # so one-shot calls may run between steps
                _expect(cur, "mel chunk", "float32", B * int(cur.shape[1]) * self.dims.n_mel)
                self._order(cur)   # a chunk torch is still writing on its own stream: the copy into the window waits for it
                self._check(self.lib.e2etts_vocoder_stream_push(self._h, _addr(cur), int(cur.shape[1]), 1 if nxt is None else 0,
                                                                C.byref(n_emit)), "e2etts_vocoder_stream_push")
                if n_emit.value > 0:
                    pend.append((n_emit.value, cur))
                if len(pend) == 2:
                    out = fetch(pend.popleft()[0])
            if out is not None:
                yield out
            cur = nxt
        while pend:
            with self.lock:
                out = fetch(pend.popleft()[0])
            yield out

    # ---- profiling
    @_locked
    def profile_enable(self, on: bool = True):
        self._check(self.lib.e2etts_profile_enable(self._h, 1 if on else 0), "e2etts_profile_enable")

    @_locked
    def profile_filter(self, kernel_class=None):
        """Bracket only launches of this kernel class with events (None: all)."""
        self._check(self.lib.e2etts_profile_filter(self._h, kernel_class.encode() if kernel_class else None), "e2etts_profile_filter")

    @_locked
    def profile_read(self):
        arr = (KernelStat * 256)()
        n = self._check(self.lib.e2etts_profile_read(self._h, arr, 256), "e2etts_profile_read")
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), ms=float(arr[i].ms), flops=float(arr[i].flops),
                     bytes=float(arr[i].bytes)) for i in range(min(n, 256))]

    def device_bytes(self) -> int:
        return int(self.lib.e2etts_device_bytes(self._h))

    @_locked
    def sync(self):
        self._check(self.lib.e2etts_sync(self._h), "e2etts_sync")

    def stream(self) -> int:
        return int(self.lib.e2etts_stream(self._h) or 0)
