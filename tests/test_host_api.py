"""CPU tests of the boundary: libzstd_amd.so loads and exports every symbol include/zstd_amd.h declares,
and the host-only entry points (bound, error names, header peek, parameter resolution) answer like the reference.
No kernel is launched here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "zstd_amd.h")).read()
    declared = set(re.findall(r"\b(zsa_[A-Za-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(os.path.join(ROOT, "zstd_amd", "libzstd_amd.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export " + name
    import zstd_amd._lib as L
    assert set(L.EXPORTS) == declared, set(L.EXPORTS) ^ declared


def test_compress_bound(ref):
    # zstd_test.go:32-40 TestCompressBound
    import zstd_amd
    for n in (0, 1, 2, 10, 456, 15468, 1313, 512, 2147483632):
        assert zstd_amd.CompressBound(n) == zstd_amd.api.cCompressBound(n) == ref.bound_(n)


def test_error_codes(ref):
    # zstd_test.go:43-57 TestErrorCode: same names for every code in -105..105
    from zstd_amd._lib import lib
    for code in range(-125, 125):
        v = ctypes.c_size_t(code & (2 ** 64 - 1))
        assert lib.zsa_getErrorName(v) == ref.getErrorName_(v), code
        assert bool(lib.zsa_isError(v)) == bool(ref.isError_(v)), code


def test_dst_too_small_error_string():
    # errors_test.go:15-29: exactly one code maps to "Destination buffer is too small"
    import zstd_amd
    hits = [c for c in range(-200, 0) if zstd_amd.IsDstSizeTooSmallError(zstd_amd.ErrorCode(c))]
    assert hits == [-70]


def test_frame_content_size(oracle, ref):
    import zstd_amd
    from zstd_amd._lib import lib
    from zstd_amd import corpus
    samples = [oracle.compress(b, 1) for b in (b"", b"a", b"x" * 300, corpus.frames("log", 70000, 1, 1)[0], bytes(300000))]
    samples += [b"", b"\x28", b"\x28\xb5\x2f\xfd", b"\x28\xb5\x2f\xfd\x20", bytes(12), (0x184D2A50).to_bytes(4, "little") + bytes(8)]
    for s in samples:
        for cut in (len(s), max(len(s) - 1, 0), 5, 6):
            x = s[:cut]
            want = ref.fcs_(x, len(x))
            if len(x) >= 5 or want >= 2 ** 64 - 2:
                got = lib.zsa_getFrameContentSize(ctypes.c_char_p(x), len(x))
                # sizes below 5 bytes: the reference answers "error" for all of them, so does the engine
                assert got == want or (len(x) < 5 and got == 2 ** 64 - 2), (len(x), got, want)
    # zstd.go:62-85 decompressSizeHint
    assert zstd_amd.api.decompressSizeHint(oracle.compress(b"", 1)) == 1
    assert zstd_amd.api.decompressSizeHint(oracle.compress(bytes(5000), 1)) == 5000
    assert zstd_amd.api.decompressSizeHint(b"\x00" * 9) == 1000 * 1000


def test_cparams_match_oracle(oracle):
    import zstd_amd
    q = oracle.lib.zo_cparams_query
    q.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]
    out = (ctypes.c_uint * 7)()
    for level in (-5, -1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13):
        for size in (0, 1, 63, 64, 1000, 1024, 8192, 8193, 16384, 16385, 65536, 131072, 131073, 262144, 262145, 1 << 20, 1 << 23, 2 ** 64 - 1):
            for dsize, mode in ((0, 0), (256, 2), (40000, 2), (256, 1)):
                rc = q(level, size, dsize, mode if mode != 2 else 2, out)
                got = zstd_amd.getCParams(level, size, dsize, mode)
                assert (got is None) == (rc != 0)
                if got is not None:
                    assert got == tuple(out), (level, size, dsize, mode)


def test_cparams_match_reference(ref):
    """the engine's clevels.h rows and ZSTD_adjustCParams_internal restatement (zsa_params.h) against the reference's own
    ZSTD_getCParams (zstd_compress.c:7011-7016), every level it serves, across the size classes"""
    import zstd_amd

    class CP(ctypes.Structure):
        _fields_ = [(n, ctypes.c_uint) for n in ("wlog", "clog", "hlog", "slog", "mml", "tlen", "strat")]
    f = ref.lib.ZSTD_getCParams
    f.restype = CP
    f.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_size_t]
    n = 0
    for level in (-7, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
        for size in (1, 63, 64, 1000, 1024, 8192, 16384, 16385, 65536, 131072, 131073, 262144, 262145, 1 << 20, 1 << 23):
            got = zstd_amd.getCParams(level, size)
            r = f(level, size, 0)
            want = (r.wlog, r.clog, r.hlog, r.slog, r.mml, r.tlen, r.strat)
            if got is None:
                assert r.strat > 5, (level, size, want)      # a btlazy2 / btopt row: outside the engine
            else:
                assert got == want, (level, size, got, want)
                n += 1
    assert n > 150
