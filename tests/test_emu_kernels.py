"""CPU tests that run the HIP kernel BODIES (zsa_decode.h / zsa_encode.h) under the lock-step lane emulator
(tests/emu/).  A debugging aid that keeps kernel logic honest without a GPU; GPU parity itself is proven by the
`-m gpu` tests through the C ABI."""
import ctypes
import os
import random
import subprocess

import pytest

from zstd_amd import corpus
import zstd_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", d], check=True)
    lib = ctypes.CDLL(os.path.join(d, "libzsa_emu.so"))
    lib.emu_compress.restype = ctypes.c_size_t
    lib.emu_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t] + [ctypes.c_uint] * 4
    lib.emu_decompress.restype = ctypes.c_size_t
    lib.emu_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.emu_compress_level.restype = ctypes.c_size_t
    lib.emu_compress_level.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    V, S = ctypes.c_void_p, ctypes.c_size_t
    lib.emu_create_cdict.restype = V; lib.emu_create_cdict.argtypes = [ctypes.c_char_p, S, ctypes.c_int]
    lib.emu_create_ddict.restype = V; lib.emu_create_ddict.argtypes = [ctypes.c_char_p, S]
    lib.emu_compress_cdict.restype = S; lib.emu_compress_cdict.argtypes = [V, S, ctypes.c_char_p, S, V]
    lib.emu_decompress_ddict.restype = S; lib.emu_decompress_ddict.argtypes = [V, S, ctypes.c_char_p, S, V]
    lib.emu_free_cdict.argtypes = [V]; lib.emu_free_ddict.argtypes = [V]
    return lib


def _inputs():
    rnd = random.Random(99)
    out = [b"", b"a", b"abcdefg", bytes(5000), rnd.randbytes(3000), b"ab" * 5000]
    out += corpus.frames("log", 131072, 2, 42) + corpus.frames("json", 65536, 2, 1234) + corpus.frames("log", 262144, 1, 43)
    out += corpus.frames("json", 1024, 8, 7) + corpus.frames("log", 12000, 2, 6)
    out.append(rnd.randbytes(131072) + bytes(100000))
    out += corpus.frames("log", 600000, 1, 44)      # 16384-cell table (2-bit LDS filter), frame longer than the window
    return out


def test_emulated_kernels_match_oracle(emu, oracle):
    for d in _inputs():
        for level in (1, -3):
            cp = zstd_amd.getCParams(level, len(d))
            cap = zstd_amd.CompressBound(len(d))
            buf = ctypes.create_string_buffer(cap + 1)
            n = emu.emu_compress(buf, cap, d, len(d), cp[0], cp[2], cp[4], cp[5])
            want = oracle.compress(d, level)
            assert buf.raw[:n] == want, (len(d), level)
            out = ctypes.create_string_buffer(len(d) + 1)
            m = emu.emu_decompress(out, len(d), want, len(want))
            assert m == len(d) and out.raw[:m] == d


def test_emulated_huffman_tables_seeded(emu, oracle):
    """Literal-heavy inputs with shaped symbol statistics: the wave-level Huffman code construction (zsa_encode.h:
    zsa_huf_buildCTable_wave -- key ranking, quick sort of the large-count buckets, code values by ballots; weight description
    by the wave FSE routines) and the decoder's table (zsa_decode.h: zsa_huf_buildTable) against the oracle.  Equal counts
    (the sort's order among them reaches the output), deep trees (length limiting), 4-bit and FSE-coded weight descriptions."""
    import numpy as np
    rng = np.random.default_rng(5)
    cases = []
    for n, nsym, kind in ((300, 3, 0), (1500, 17, 1), (4000, 40, 2), (9000, 90, 3), (20000, 180, 4), (50000, 256, 2),
                          (131072, 256, 3), (140000, 90, 1), (700, 8, 2), (60000, 256, 0)):
        if kind == 0: p = rng.dirichlet(np.ones(nsym) * 0.3)
        elif kind == 1: p = 0.5 ** np.arange(1, nsym + 1); p = p / p.sum()
        elif kind == 2: p = np.ones(nsym) / nsym
        elif kind == 3: p = np.repeat(rng.dirichlet(np.ones(nsym // 8 + 1)), 8)[:nsym]; p = p / p.sum()
        else: p = 1.0 / np.arange(1, nsym + 1); p = p / p.sum()
        syms = rng.permutation(256)[:nsym]
        if kind in (2, 3):   # exact ties: a shuffled multiset
            d = np.repeat(syms, np.maximum(1, (p * n).astype(int))); rng.shuffle(d)
        else:
            d = rng.choice(syms, size=n, p=p)
        cases.append(d.astype(np.uint8).tobytes())
    for d in cases:
        for level in (1, 5):
            want = oracle.compress(d, level)
            cap = zstd_amd.CompressBound(len(d))
            buf = ctypes.create_string_buffer(cap + 1)
            n = emu.emu_compress_level(buf, cap, d, len(d), level)
            assert buf.raw[:n] == want, (len(d), level)
            out = ctypes.create_string_buffer(len(d) + 1)
            m = emu.emu_decompress(out, len(d), want, len(want))
            assert m == len(d) and out.raw[:m] == d


def test_emulated_decoder_errors(emu, oracle):
    d = corpus.frames("json", 5000, 1, 3)[0]
    c = oracle.compress(d, 1)
    for x, cap in ((c[:-1], 6000), (c[:10], 6000), (b"\x00" * 20, 6000), (c + b"\x01", 6000), (c, len(d) - 1), (c + c[:5], 12000)):
        out = ctypes.create_string_buffer(cap + 1)
        m = emu.emu_decompress(out, cap, x, len(x))
        want = oracle.decompress(x, cap)
        assert isinstance(want, int) and m == 2 ** 64 + want, (len(x), cap, m, want)


def test_emulated_dictionary_path_matches_golden(emu):
    """digest kernels (zsa_dict.h) + attached / copied / re-loaded dictionary parses, against the reference-generated
    fixture (tests/golden/dict_cases.json)"""
    import hashlib
    from test_oracle_dict import load_dict_cases
    dicts, payloads, cases = load_dict_cases()
    handles = {}
    for c in cases:
        if c["level"] != 1 and c["dict"] not in ("ref_test_dict", "raw_log_4k"):
            continue
        key = (c["dict"], c["level"])
        if key not in handles:
            d = dicts[c["dict"]]
            handles[key] = (emu.emu_create_cdict(d, len(d), c["level"]), emu.emu_create_ddict(d, len(d)))
            assert handles[key][0] and handles[key][1], key
        cd, dd = handles[key]
        p = payloads[c["payload"]]
        cap = zstd_amd.CompressBound(len(p))
        buf = ctypes.create_string_buffer(cap + 1)
        n = emu.emu_compress_cdict(buf, cap, p, len(p), cd)
        assert n == c["c_len"] and hashlib.sha256(buf.raw[:n]).hexdigest()[:32] == c["c_sha256"], (c["dict"], c["level"], c["payload"])
        out = ctypes.create_string_buffer(len(p) + 1)
        m = emu.emu_decompress_ddict(out, len(p), buf.raw[:n], n, dd)
        assert m == len(p) and out.raw[:m] == p
    for cd, dd in handles.values():
        emu.emu_free_cdict(cd); emu.emu_free_ddict(dd)
    bad = dicts["ref_test_dict"]
    bad = bad[:40] + bytes([bad[40] ^ 0xFF]) + bad[41:60]
    assert not emu.emu_create_cdict(bad, len(bad), 1) and not emu.emu_create_ddict(bad, len(bad))


def test_emulated_dictionary_path_above_fast_matches_golden(emu):
    """CDict digests for ZSTD_dfast (tagged double hash), the hash chain and the row records; the attached-dictionary parsers
    of zsa_dict_hi.h and zsa_lazy.h (levels 3 .. 10) -- against the reference-generated fixture tests/golden/dict_cases_hi.json.
    Payloads above the attach cut-off answer parameter_unsupported."""
    import hashlib
    from test_oracle_dict import load_dict_cases_hi
    dicts, payloads, cases = load_dict_cases_hi()
    handles = {}
    served = declined = 0
    for i, c in enumerate(cases):
        if i % 3 != (len(c["payload"]) + c["level"]) % 3 and len(payloads[c["payload"]]) < 16000:
            continue   # a third of the small cases (the GPU test runs them all)
        key = (c["dict"], c["level"])
        if key not in handles:
            d = dicts[c["dict"]]
            handles[key] = emu.emu_create_cdict(d, len(d), c["level"])
            assert handles[key] or (c["level"] == 10 and len(d) < 16000), key
        cd = handles[key]
        if not cd:
            continue
        p = payloads[c["payload"]]
        cap = zstd_amd.CompressBound(len(p))
        buf = ctypes.create_string_buffer(cap + 1)
        n = emu.emu_compress_cdict(buf, cap, p, len(p), cd)
        if n == 2 ** 64 - 40:
            assert len(p) > 16384, (c["dict"], c["level"], c["payload"])
            declined += 1
            continue
        served += 1
        assert n == c["c_len"] and hashlib.sha256(buf.raw[:n]).hexdigest()[:32] == c["c_sha256"], (c["dict"], c["level"], c["payload"], n, c["c_len"])
    for cd in handles.values():
        if cd:
            emu.emu_free_cdict(cd)
    assert served > 500 and declined > 10, (served, declined)


def test_emulated_dictionary_parsers_seeded(emu, oracle):
    """zsa_parse_dict.h against the oracle on seeded payloads the fixture does not hold: the external-segment parser with
    steps that have grown over an incompressible stretch before dictionary content turns up (where the position after a pair
    lies beyond the match: the insert of zstd_fast.c:870-873 must not happen), multi-block payloads, matches that run from
    the dictionary into the payload; the attached parser on small ones."""
    from test_oracle_dict import load_dict_cases
    dicts = load_dict_cases()[0]
    rnd = random.Random(8)
    d = dicts["raw_log_112k"]
    n = 201820
    big = bytes(rnd.choice(b"abc") for _ in range(100000)) + rnd.randbytes(100000) + d[:n - 200000]
    cases = [("raw_log_112k", big), ("raw_log_4k", (dicts["raw_log_4k"][-700:] + rnd.randbytes(50)) * 30),
             ("trained_json_8k", corpus.frames("json", 9000, 1, 5)[0]), ("trained_json_8k", corpus.frames("json", 1000, 1, 6)[0]),
             ("raw_json_40k", rnd.randbytes(9000) + dicts["raw_json_40k"][:9000])]
    for dn, p in cases:
        cd = emu.emu_create_cdict(dicts[dn], len(dicts[dn]), 1)
        ocd = oracle.cdict(dicts[dn], 1)
        cap = zstd_amd.CompressBound(len(p))
        buf = ctypes.create_string_buffer(cap + 1)
        r = emu.emu_compress_cdict(buf, cap, p, len(p), cd)
        assert buf.raw[:r] == oracle.compress_cdict(p, ocd), (dn, len(p))
        emu.emu_free_cdict(cd); oracle.free_cdict(ocd)


def test_emulated_dfast_parse_matches_oracle(emu, oracle):
    """zsa_dfast.h (levels 2-4 where the strategy is ZSTD_dfast) through the whole frame encoder"""
    cases = [b"", b"abc", corpus.frames("json", 1000, 1, 7)[0], corpus.frames("log", 20000, 1, 5)[0], corpus.frames("json", 70000, 1, 3)[0],
             corpus.frames("log", 140000, 1, 4)[0], bytes(300) + b"xyz" * 500 + bytes(range(256)) * 4]
    for d in cases:
        for level in (3, 4, 2, 0):
            want = oracle.compress(d, level)
            cap = zstd_amd.CompressBound(len(d))
            buf = ctypes.create_string_buffer(cap + 1)
            r = emu.emu_compress_level(buf, cap, d, len(d), level)
            if isinstance(want, int):
                assert r == 2 ** 64 + want, (len(d), level)
            else:
                assert r == len(want) and buf.raw[:r] == want, (len(d), level)


def test_emulated_greedy_chunks_match_oracle(emu, oracle):
    """zsa_greedy_row_chunk (level 5 above 16 KiB: 64 searches per set-up).  Besides ordinary text: a periodic stretch that runs
    across a block boundary -- the next block opens with a long repcode match, so the positions pending from the last block
    must enter the table by the rule for long gaps (both ends only), not all of them as a chunk's first search would enter
    them --, runs of equal bytes (neighbouring positions share rows), literal stretches long enough for the search step to
    grow, and the three clevels.h tables (minMatch 4 / 5, 16- and 32-entry rows)."""
    rnd = random.Random(12)
    text = lambda n, seed: corpus.frames("log", n, 1, seed)[0]
    unit = rnd.randbytes(24)
    crossing = text(131072 - 2000, 1) + unit * 170 + text(60000, 2) + unit * 40 + text(80000, 3)      # > 256 KB: minMatch 5
    mixed = text(30000, 4) + bytes(3000) + rnd.randbytes(2500) + text(20000, 5) + b"ab" * 900 + rnd.randbytes(700) + text(9000, 6)
    cases = [(crossing, 5), (crossing[:262144], 4), (mixed, 5), (text(131072, 7), 5), (corpus.frames("json", 65536, 1, 8)[0], 5),
             (text(200000, 9), 5), (text(131072 - 900, 10) + unit * 80 + text(5000, 11), 5)]
    stat = (ctypes.c_ulonglong * 8).in_dll(emu, "g_zsa_stat")
    chunks0 = stat[0]
    for d, level in cases:
        want = oracle.compress(d, level)
        cap = zstd_amd.CompressBound(len(d))
        buf = ctypes.create_string_buffer(cap + 1)
        r = emu.emu_compress_level(buf, cap, d, len(d), level)
        assert r == len(want) and buf.raw[:r] == want, (len(d), level)
    assert stat[0] - chunks0 > 5000        # (the chunk path did the work)


def test_emulated_lazy_parse_matches_oracle(emu, oracle):
    """zsa_lazy.h (greedy / lazy / lazy2 with the hash-chain and the row-hash match finders: levels 4..12 where clevels.h
    selects them; level 5 is the reference's default) through the whole frame encoder, cost-based sequence modes and
    multi-block repeat tables included"""
    rnd = random.Random(5)
    # (sizes kept small and the large inputs at two levels each: under the emulator all 64 lanes walk the parser)
    small = [b"", b"abc", b"abcdefg" * 3, corpus.frames("json", 1000, 1, 7)[0], corpus.frames("log", 16384, 1, 5)[0],
             corpus.frames("log", 20000, 1, 5)[0], bytes(300) + b"xyz" * 500 + bytes(range(256)) * 4,
             rnd.randbytes(5000) + bytes(3000) + rnd.randbytes(40)]
    words = b"".join(rnd.choice([b"alpha ", b"beta ", b"gamma ", b"delta=%d " % rnd.randrange(50)]) for _ in range(24000))
    cases = [(d, (5, 6, 7, 4, 9)) for d in small]
    cases += [(corpus.frames("json", 40000, 1, 3)[0], (5, 7)), (corpus.frames("log", 140000, 1, 4)[0], (5, 4)),   # two blocks
              (bytes(140000), (5,)), (words, (6, 8))]                          # two blocks at ZSTD_lazy / ZSTD_lazy2: priced repeat tables
    n_checked = 0
    for d, levels in cases:
        for level in levels:
            want = oracle.compress(d, level)
            cap = zstd_amd.CompressBound(len(d))
            buf = ctypes.create_string_buffer(cap + 1)
            r = emu.emu_compress_level(buf, cap, d, len(d), level)
            if isinstance(want, int):
                assert r == 2 ** 64 + want, (len(d), level)
            else:
                assert r == len(want) and buf.raw[:r] == want, (len(d), level)
                n_checked += 1
    assert n_checked >= 40


def test_emulated_fingerprint_collisions(oracle):
    """zsa_parse.h keeps a fingerprint of the 4 bytes at each position in the spare bits of a match-table cell and only
    reads a candidate's bytes when it wins its window; with 2-bit fingerprints (test build) one probe in four collides,
    which drives the take-the-next-event path.  Output must not change.  zsa_dfast.h does the same in both of its tables
    (levels 3 and 4, level 2 at these sizes)."""
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-s", "-C", d, os.path.join(d, "libzsa_emu_fp2.so")], check=True)
    lib = ctypes.CDLL(os.path.join(d, "libzsa_emu_fp2.so"))
    lib.emu_compress_level.restype = ctypes.c_size_t
    lib.emu_compress_level.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    rnd = random.Random(9)
    cases = [corpus.frames("log", 40000, 1, 5)[0], corpus.frames("json", 131073, 1, 6)[0], corpus.frames("log", 300000, 1, 7)[0],
             bytes(rnd.choice(b"abcde") for _ in range(30000)) + b"ab" * 3000 + bytes(5000)]
    for data in cases:
        for level in (1, -3, 2, 3, 4):
            want = oracle.compress(data, level)
            if isinstance(want, int):
                continue
            cap = zstd_amd.CompressBound(len(data))
            buf = ctypes.create_string_buffer(cap + 1)
            r = lib.emu_compress_level(buf, cap, data, len(data), level)
            assert r == len(want) and buf.raw[:r] == want, (len(data), level)
