"""Fixed-seed synthetic corpora: log-like (kind "log") and JSON-like (kind "json").

Backed by libzcorpus.so (zstd_amd/csrc/corpus.c, host C).  Frames are generated independently from
(seed, frame index), so ranks can produce disjoint slices of the same corpus.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libzcorpus.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("zstd_amd: %s not found; run __graft_entry__.build()" % _PATH)
        _lib = ctypes.CDLL(_PATH)
        _lib.zc_generate.restype = None
        _lib.zc_generate.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_uint64, ctypes.c_int, ctypes.c_int]
    return _lib


KINDS = {"log": 0, "json": 1}


def generate(kind, frame_size, n_frames, seed, first_frame=0, out=None, threads=8):
    """Return a uint8 numpy array of n_frames*frame_size bytes (frames first_frame.. of the corpus)."""
    lib = _load()
    total = frame_size * n_frames
    if out is None:
        out = np.empty(total, dtype=np.uint8)
    assert out.dtype == np.uint8 and out.size >= total and out.flags["C_CONTIGUOUS"]
    if total:
        lib.zc_generate(out.ctypes.data, frame_size, first_frame, n_frames, seed, KINDS[kind], threads)
    return out[:total]


def frames(kind, frame_size, n_frames, seed, first_frame=0):
    """List of bytes objects, one per frame."""
    a = generate(kind, frame_size, n_frames, seed, first_frame)
    return [a[i * frame_size:(i + 1) * frame_size].tobytes() for i in range(n_frames)]
