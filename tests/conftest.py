import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class _Lib:
    """ctypes view of a libzstd-shaped library (prefix 'ZSTD_' for oracle/_ref, 'zo_' for the oracle)."""

    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.prefix = prefix
        S, V, I = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int
        self.compress_ = self._fn("compress", S, [V, S, ctypes.c_char_p, S, I])
        self.decompress_ = self._fn("decompress", S, [V, S, ctypes.c_char_p, S])
        self.bound_ = self._fn("compressBound", S, [S])
        self.getErrorName_ = self._fn("getErrorName", ctypes.c_char_p, [S])
        self.isError_ = self._fn("isError", ctypes.c_uint, [S])
        self.fcs_ = self._fn("getFrameContentSize", ctypes.c_ulonglong, [ctypes.c_char_p, S])

    def _fn(self, name, res, args):
        f = getattr(self.lib, self.prefix + name)
        f.restype = res
        f.argtypes = args
        return f

    def compress(self, data, level=1, cap=None):
        cap = self.bound_(len(data)) if cap is None else cap
        buf = ctypes.create_string_buffer(cap + 1)
        r = self.compress_(buf, cap, data, len(data), level)
        if self.isError_(r):
            return -(2 ** 64 - r)
        return buf.raw[:r]

    def decompress(self, data, cap):
        buf = ctypes.create_string_buffer(cap + 1)
        r = self.decompress_(buf, cap, data, len(data))
        if self.isError_(r):
            return -(2 ** 64 - r)
        return buf.raw[:r]

    # ---- dictionary path (zstd_bulk.go): ZSTD_createCDict / compress_usingCDict / createDDict / decompress_usingDDict
    def cdict(self, d, level):
        S, V = ctypes.c_size_t, ctypes.c_void_p
        f = self._fn("createCDict", V, [ctypes.c_char_p, S, ctypes.c_int])
        return f(d, len(d), level)

    def ddict(self, d):
        S, V = ctypes.c_size_t, ctypes.c_void_p
        f = self._fn("createDDict", V, [ctypes.c_char_p, S])
        return f(d, len(d))

    def free_cdict(self, h):
        self._fn("freeCDict", None if self.prefix == "zo_" else ctypes.c_size_t, [ctypes.c_void_p])(h)

    def free_ddict(self, h):
        self._fn("freeDDict", None if self.prefix == "zo_" else ctypes.c_size_t, [ctypes.c_void_p])(h)

    def compress_cdict(self, data, h):
        S, V = ctypes.c_size_t, ctypes.c_void_p
        cap = self.bound_(len(data))
        buf = ctypes.create_string_buffer(cap + 1)
        if self.prefix == "zo_":
            r = self._fn("compress_usingCDict", S, [V, S, ctypes.c_char_p, S, V])(buf, cap, data, len(data), h)
        else:
            if not hasattr(self, "_cctx"):
                self._cctx = self._fn("createCCtx", V, [])()
            r = self._fn("compress_usingCDict", S, [V, V, S, ctypes.c_char_p, S, V])(self._cctx, buf, cap, data, len(data), h)
        if self.isError_(r):
            return -(2 ** 64 - r)
        return buf.raw[:r]

    def decompress_ddict(self, data, cap, h):
        S, V = ctypes.c_size_t, ctypes.c_void_p
        buf = ctypes.create_string_buffer(cap + 1)
        if self.prefix == "zo_":
            r = self._fn("decompress_usingDDict", S, [V, S, ctypes.c_char_p, S, V])(buf, cap, data, len(data), h)
        else:
            if not hasattr(self, "_dctx"):
                self._dctx = self._fn("createDCtx", V, [])()
            r = self._fn("decompress_usingDDict", S, [V, V, S, ctypes.c_char_p, S, V])(self._dctx, buf, cap, data, len(data), h)
        if self.isError_(r):
            return -(2 ** 64 - r)
        return buf.raw[:r]

    def error_name(self, code):
        return self.getErrorName_(ctypes.c_size_t((-code) if code > 0 else (2 ** 64 + code) if code < 0 else 0)).decode()


@pytest.fixture(scope="session")
def oracle():
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        import subprocess
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle"], check=True)
    return _Lib(path, "zo_")


@pytest.fixture(scope="session")
def ref():
    """The reference's own C (oracle/_ref), when it has been built in the dev container; else skip."""
    path = os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libzstd_ref.so not built (needs /root/reference)")
    return _Lib(path, "ZSTD_")
