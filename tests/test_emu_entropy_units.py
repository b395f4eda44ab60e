"""Unit tests of the wave-level entropy-table routines of zsa_encode.h (share distribution incl. its second-chance path,
table description, Huffman code construction with the depth limit and the large-count bucket order) under the lane emulator,
against the functions of the same job in the reference build (oracle/_ref/libzstd_ref.so: FSE_normalizeCount,
FSE_writeNCount, HUF_buildCTable_wksp).  CPU only; needs the reference build, which exists where /root/reference does."""
import ctypes
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference build absent (oracle/_ref is built where /root/reference exists)")


@pytest.fixture(scope="module")
def libs():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", d], check=True)
    emu = ctypes.CDLL(os.path.join(d, "libzsa_emu.so"))
    ref = ctypes.CDLL(REF)
    U, S = ctypes.c_uint, ctypes.c_size_t
    emu.emu_fse_ncount.restype = S
    emu.emu_fse_ncount.argtypes = [ctypes.POINTER(U), U, U, U, U, ctypes.POINTER(ctypes.c_short), ctypes.c_void_p]
    emu.emu_huf_ctable.restype = U
    emu.emu_huf_ctable.argtypes = [ctypes.POINTER(U), U, U, ctypes.POINTER(ctypes.c_ubyte), ctypes.POINTER(ctypes.c_ushort)]
    ref.FSE_normalizeCount.restype = S
    ref.FSE_normalizeCount.argtypes = [ctypes.POINTER(ctypes.c_short), U, ctypes.POINTER(U), S, U, U]
    ref.FSE_writeNCount.restype = S
    ref.FSE_writeNCount.argtypes = [ctypes.c_void_p, S, ctypes.POINTER(ctypes.c_short), U, U]
    ref.FSE_optimalTableLog.restype = U
    ref.FSE_optimalTableLog.argtypes = [U, S, U]
    ref.HUF_buildCTable_wksp.restype = S
    ref.HUF_buildCTable_wksp.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(U), U, U, ctypes.c_void_p, S]
    ref.ZSTD_isError.restype = U
    ref.ZSTD_isError.argtypes = [S]
    return emu, ref


def _fse_shapes(rnd):
    """count vectors over up to 53 symbols (the sequence alphabets) and up to 13 (Huffman weights)"""
    for _ in range(400):
        n = rnd.choice([2, 3, 5, 13, 20, 29, 36, 53])
        kind = rnd.randrange(7)
        if kind == 0:      # near-uniform small counts: many shares round up together -> second chance
            c = [rnd.choice([1, 1, 2, 2, 3]) for _ in range(n)]
        elif kind == 1:    # geometric
            c = [max(0, int(2000 * (0.6 ** i)) + rnd.randrange(3) - 1) for i in range(n)]
        elif kind == 2:    # a few heavy symbols and many singletons
            c = [rnd.choice([1, 1, 1, 0]) for _ in range(n)]
            for _ in range(rnd.randrange(1, 4)):
                c[rnd.randrange(n)] = rnd.randrange(50, 5000)
        elif kind == 3:    # long runs of unused symbols (run-length codes of 3 and 24)
            c = [0] * n
            for _ in range(rnd.randrange(2, 6)):
                c[rnd.randrange(n)] = rnd.randrange(1, 400)
        elif kind == 4:    # all symbols just above the one-cell threshold
            base = rnd.randrange(3, 40)
            c = [base + rnd.randrange(2) for _ in range(n)]
        elif kind == 5:    # random
            c = [rnd.randrange(0, rnd.choice([4, 30, 1000])) for _ in range(n)]
        else:              # one dominant, rest tiny, large total (low-probability marks)
            c = [rnd.randrange(0, 3) for _ in range(n)]
            c[rnd.randrange(n)] = rnd.randrange(3000, 100000)
        if sum(1 for x in c if x) < 2:
            continue
        while c[-1] == 0:
            c.pop()
        yield c


def test_share_distribution_and_table_description_match_reference(libs):
    emu, ref = libs
    rnd = random.Random(2024)
    U = ctypes.c_uint
    cases = errors = 0
    stat = (ctypes.c_ulonglong * 8).in_dll(emu, "g_zsa_stat")
    second0 = stat[7]
    for c in _fse_shapes(rnd):
        maxSV, total = len(c) - 1, sum(c)
        cnt = (U * 256)(*c)
        hb = lambda x: x.bit_length() - 1
        min_log = min(hb(total) + 1, hb(maxSV) + 2) if maxSV else 1
        logs = {ref.FSE_optimalTableLog(9, total, maxSV), ref.FSE_optimalTableLog(6, total, maxSV)} | set(range(max(5, min_log - 1), 10))
        for tl in sorted(logs):
            for low in (0, 1):
                rn = (ctypes.c_short * 256)()
                r = ref.FSE_normalizeCount(rn, tl, cnt, total, maxSV, low)
                en = (ctypes.c_short * 256)()
                eb = ctypes.create_string_buffer(512)
                e = emu.emu_fse_ncount(cnt, maxSV, total, tl, low, en, eb)
                cases += 1
                if ref.ZSTD_isError(r):
                    errors += 1
                    assert e == 2 ** 64 - 1, (c, tl, low)
                    continue
                assert e != 2 ** 64 - 1, (c, tl, low)
                assert list(en)[:maxSV + 1] == list(rn)[:maxSV + 1], (c, tl, low)
                rb = ctypes.create_string_buffer(512)
                w = ref.FSE_writeNCount(rb, 512, rn, maxSV, tl)
                assert not ref.ZSTD_isError(w)
                assert e == w and eb.raw[:e] == rb.raw[:w], (c, tl, low)
    second = stat[7] - second0          # times the second-chance distribution ran (counter of the test build)
    assert cases > 3000 and second > 100 and errors > 20, (cases, second, errors)


def _huf_shapes(rnd):
    for _ in range(300):
        n = rnd.choice([2, 3, 16, 60, 100, 180, 256])
        kind = rnd.randrange(6)
        if kind == 0:      # fibonacci-like: a degenerate, very deep tree (depth limit with a long repayment)
            a, b, c = 1, 1, []
            for _ in range(min(n, 40)):
                c.append(a); a, b = b, a + b
            rnd.shuffle(c)
        elif kind == 1:    # powers of two
            c = [1 << rnd.randrange(0, 17) for _ in range(n)]
        elif kind == 2:    # many equal large counts: the bucket order of equal counts decides the codes
            vals = [rnd.randrange(200, 60000) for _ in range(rnd.randrange(1, 6))]
            c = [rnd.choice(vals) for _ in range(n)]
        elif kind == 3:    # text-like
            c = [int(50000 / (i + 1) ** 1.2) + rnd.randrange(3) for i in range(n)]
            rnd.shuffle(c)
        elif kind == 4:    # sparse
            c = [rnd.choice([0, 0, 0, 1, 2, 300]) for _ in range(n)]
        else:
            c = [rnd.randrange(0, rnd.choice([3, 200, 70000])) for _ in range(n)]
        while c and c[-1] == 0:
            c.pop()
        if sum(1 for x in c if x) < 2 or sum(c) > 131072 * 4:
            continue
        yield c


def test_huffman_codes_match_reference(libs):
    emu, ref = libs
    rnd = random.Random(77)
    U = ctypes.c_uint
    cases = limited = 0
    wksp = ctypes.create_string_buffer(16384 + 64)
    wk = (ctypes.addressof(wksp) + 63) & ~63
    for c in _huf_shapes(rnd):
        maxSV = len(c) - 1
        cnt = (U * 256)(*c)
        for limit in (11, 8, 7, 5):
            if (1 << limit) < sum(1 for x in c if x):     # (the limit cannot hold that many symbols: the callers never ask)
                continue
            ct = (ctypes.c_size_t * 257)()
            r = ref.HUF_buildCTable_wksp(ct, cnt, maxSV, limit, wk, 16384)
            assert not ref.ZSTD_isError(r)
            nb = (ctypes.c_ubyte * 256)()
            val = (ctypes.c_ushort * 256)()
            e = emu.emu_huf_ctable(cnt, maxSV, limit, nb, val)
            assert e == r, (c, limit, e, r)
            for s in range(maxSV + 1):
                rb = ct[s + 1] & 0xFF
                rv = (ct[s + 1] >> (64 - rb)) if rb else 0
                assert (nb[s], val[s]) == (rb, rv), (c, limit, s)
            cases += 1
            if r == limit:
                limited += 1
    assert cases > 500 and limited > 100, (cases, limited)
