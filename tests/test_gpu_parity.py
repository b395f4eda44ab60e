"""GPU parity tests: everything goes through the C ABI of libzstd_amd.so and is compared with the oracle
(oracle/liboracle.so, itself pinned to the reference by tests/test_oracle.py) and with the reference-generated
golden fixtures.  Bar: byte-identical compressed frames, bit-exact round trips, same error codes."""
import ctypes
import hashlib
import json
import os
import random

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def z():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd
    return zstd_amd


def _check_compress(z, oracle, frames, level=1):
    got = z.CompressBatch(frames, level)
    for f, c in zip(frames, got):
        want = oracle.compress(f, level)
        assert isinstance(c, bytes), (len(f), c)
        assert c == want, "frame of %d bytes differs from the oracle (got %d, want %d bytes)" % (len(f), len(c), len(want))
    return got


def test_golden_small(z, oracle):
    from golden_inputs import small_inputs
    inputs = dict(small_inputs())
    cases = json.load(open(os.path.join(HERE, "golden", "kat_small.json")))["cases"]
    by_level = {}
    for k in cases:
        src = bytes.fromhex(k["src_hex"]) if "src_hex" in k else inputs[k["name"]]
        by_level.setdefault(k["level"], []).append((k, src))
    for level, items in by_level.items():
        got = z.CompressBatch([s for _, s in items], level)
        for (k, src), c in zip(items, got):
            assert isinstance(c, bytes), (k["name"], c)
            assert len(c) == k["c_len"] and hashlib.sha256(c).hexdigest() == k["c_sha256"], k["name"]
        back = z.DecompressBatch(got, [max(len(s), 1) for _, s in items])
        assert back == [s for _, s in items]


def test_golden_corpora(z):
    from zstd_amd import corpus
    for t in json.load(open(os.path.join(HERE, "golden", "corpus_frames.json")))["tables"]:
        frames = corpus.frames(t["kind"], t["frame_size"], t["n"], t["seed"])
        got = z.CompressBatch(frames, 1)
        for c, (clen, chash, _) in zip(got, t["rows"]):
            assert isinstance(c, bytes) and len(c) == clen and hashlib.sha256(c).hexdigest()[:32] == chash, (t["kind"], t["frame_size"])
        assert z.DecompressBatch(got, [t["frame_size"]] * t["n"]) == frames


def test_reference_go_cases(z, oracle):
    # zstd_test.go:60-88 (Hello World round trip), :119-142 (payload matrix at the levels this engine implements),
    # :216-254 (empty input / output), :325-342 (small payload)
    payloads = [b"", b"\x00", b"Hello World!", b"a" * 3, bytes(range(256)) * 3]
    for level in (1, -1, -3, 2, 3, 4, 5, 6, 7, 8):
        got = _check_compress(z, oracle, payloads, level)
        assert z.DecompressBatch(got, [max(len(p), 1) for p in payloads]) == payloads
    c = z.CompressLevel(None, b"Hello World!", 1)
    assert z.Decompress(None, c) == b"Hello World!"
    assert z.Decompress(bytearray(3), c) == b"Hello World!"          # dst too small: reallocated
    with pytest.raises(z.api.ZstdError):
        z.Decompress(None, b"")                                       # ErrEmptySlice, zstd.go:135-137
    # zstd.go:90-92: Compress == CompressLevel(DefaultCompression = 5), ZSTD_lazy with the hash chain at this size
    assert z.Compress(None, b"Hello World!") == oracle.compress(b"Hello World!", 5)
    # levels whose strategy is ZSTD_btlazy2 or above are outside the engine and say so
    r = z.CompressBatch([b"x" * 100], 13)
    assert isinstance(r[0], z.ErrorCode) and str(r[0]) == "Unsupported parameter"


def test_decompress_into_semantics(z, oracle):
    # zstd_test.go:90-117: exact fit works, one byte short errors with the string Go matches on
    from zstd_amd import corpus
    d = corpus.frames("json", 3000, 1, 5)[0]
    c = oracle.compress(d, 1)
    buf = bytearray(len(d))
    assert z.DecompressInto(buf, c) == len(d) and bytes(buf) == d
    with pytest.raises(z.ErrorCode) as e:
        z.DecompressInto(bytearray(len(d) - 1), c)
    assert z.IsDstSizeTooSmallError(e.value)
    # zstd_test.go:256-275: 130 KB payload, too-small dst -> fallback still returns the content
    big = corpus.frames("log", 130 * 1024, 1, 8)[0]
    assert z.Decompress(bytearray(10), z.CompressLevel(None, big, 1)) == big
    # zstd_test.go:317-323: header announcing an absurd size on a tiny input
    bomb = bytes.fromhex("28b52ffd") + bytes([0xE0]) + (10 ** 12).to_bytes(8, "little") + b"\x01\x00\x00"
    with pytest.raises(z.ErrorCode) as e:
        z.DecompressInto(bytearray(100), bomb)
    assert e.value.code == oracle.decompress(bomb, 100)


def test_seeded_sizes_and_shapes(z, oracle):
    rnd = random.Random(2024)
    frames = []
    for n in (0, 1, 2, 5, 6, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 8191, 8192, 8193, 16383, 16384, 16385,
              65535, 65536, 131071, 131072, 131073, 200000, 262144, 262145, 400000, 600000):
        kind = rnd.randrange(5)
        if kind == 0:
            frames.append(rnd.randbytes(n))
        elif kind == 1:
            frames.append(bytes(n))
        elif kind == 2:
            frames.append((rnd.randbytes(rnd.randrange(1, 40)) * (n + 1))[:n])
        elif kind == 3:
            frames.append(bytes(rnd.choice(b"aaaabbbcd\n") for _ in range(n)))
        else:
            frames.append(b"".join(b"k%d=%x;" % (rnd.randrange(50), rnd.getrandbits(16)) for _ in range(n // 6 + 1))[:n])
    frames.append(rnd.randbytes(131072) + bytes(131072))          # second block is RLE
    frames.append(bytes(131072) + rnd.randbytes(131072))          # raw second block
    frames.append(b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 2000) for _ in range(300)))
    got = _check_compress(z, oracle, frames, 1)
    assert z.DecompressBatch(got, [max(len(f), 1) for f in frames]) == frames
    _check_compress(z, oracle, frames[:24], -5)


def test_decode_foreign_frames(z, oracle, ref):
    # frames the engine did not produce: other levels/strategies, checksums, skippable + concatenated frames
    from zstd_amd import corpus
    srcs = corpus.frames("log", 100000, 2, 77) + corpus.frames("json", 300000, 1, 78) + [bytes(50000), random.Random(5).randbytes(40000)]
    frames, want = [], []
    for lvl in (1, 3, 5, 9, 19):
        for s in srcs:
            frames.append(ref.compress(s, lvl))
            want.append(s)
    lib = ref.lib
    lib.ZSTD_createCCtx.restype = ctypes.c_void_p
    lib.ZSTD_CCtx_setParameter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.ZSTD_compress2.restype = ctypes.c_size_t
    lib.ZSTD_compress2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    lib.ZSTD_freeCCtx.argtypes = [ctypes.c_void_p]
    cctx = lib.ZSTD_createCCtx()
    lib.ZSTD_CCtx_setParameter(cctx, 201, 1)
    buf = ctypes.create_string_buffer(400000)
    n = lib.ZSTD_compress2(cctx, buf, 400000, srcs[0], len(srcs[0]))
    lib.ZSTD_freeCCtx(cctx)
    chk = buf.raw[:n]
    skip = (0x184D2A5F).to_bytes(4, "little") + (3).to_bytes(4, "little") + b"abc"
    frames += [chk, skip + chk + skip, frames[0] + frames[1]]
    want += [srcs[0], srcs[0], srcs[0] + srcs[1]]
    got = z.DecompressBatch(frames, [len(w) for w in want])
    assert got == want
    bad = chk[:-1] + bytes([chk[-1] ^ 0x80])
    r = z.DecompressBatch([bad], [len(srcs[0])])[0]
    assert isinstance(r, z.ErrorCode) and r.code == -22


def test_corrupted_inputs_error_like_reference(z, oracle):
    from zstd_amd import corpus
    d = corpus.frames("json", 20000, 1, 3)[0]
    c = oracle.compress(d, 1)
    rnd = random.Random(11)
    cases = [c[:-1], c[:10], c[:4], c[:5], b"\x00" * 20, c + b"\x01", c + c[:5], c[:6] + bytes([c[6] ^ 0x40]) + c[7:], b"\x28\xb5\x2f\xfd"]
    for _ in range(40):
        x = bytearray(c)
        for _ in range(rnd.randrange(1, 4)):
            x[rnd.randrange(9, len(x))] ^= 1 << rnd.randrange(8)
        cases.append(bytes(x))
    got = z.DecompressBatch(cases, [25000] * len(cases))
    for x, g in zip(cases, got):
        w = oracle.decompress(x, 25000)
        if isinstance(w, int):
            assert isinstance(g, z.ErrorCode) and g.code == w, (len(x), g, w)
        else:
            assert g == w


def test_full_size_round_trip_properties(z):
    # BASELINE configs[1] at full size: 8192 x 128 KiB device-resident; checks size-independent properties
    import numpy as np
    import torch
    from zstd_amd import corpus
    N, FS = 8192, 131072
    host = corpus.generate("log", FS, N, 42, threads=16)
    src = torch.from_numpy(host).cuda()
    bound = z.CompressBound(FS)
    stride = (bound + 255) // 256 * 256
    cbuf = torch.zeros(N * stride, dtype=torch.uint8, device="cuda")
    out = torch.empty(N * FS, dtype=torch.uint8, device="cuda")
    enc, dec = z.DeviceBatch(N), z.DeviceBatch(N)
    enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound)
    dec.set_frames_strided(N, cbuf.data_ptr(), stride, None, out.data_ptr(), FS, FS)
    enc.compress(1)
    dec.link_src_sizes(enc)
    dec.decompress()
    cs, ds = enc.results(), dec.results()
    assert all(c < 2 ** 63 for c in cs) and all(d == FS for d in ds)
    assert bool(torch.equal(out, src))
    # golden rows of the first 64 frames of this very corpus (reference-generated): checksum of checksums
    t = [t for t in json.load(open(os.path.join(HERE, "golden", "corpus_frames.json")))["tables"] if t["frame_size"] == FS and t["seed"] == 42][0]
    ch = cbuf[:64 * stride].cpu().numpy()
    for i, (clen, chash, _) in enumerate(t["rows"]):
        assert cs[i] == clen and hashlib.sha256(ch[i * stride:i * stride + clen].tobytes()).hexdigest()[:32] == chash
    # idempotence: a second pass over the same inputs yields the same sizes and bytes
    before = cbuf.clone()
    enc.compress(1)
    assert enc.results() == cs and bool(torch.equal(before, cbuf))


def test_bulk_processor_matches_golden(z):
    """BulkProcessor (zstd_bulk.go): CDict/DDict digested on the device; every reference-generated case of
    tests/golden/dict_cases.json byte-identical, in one batch launch per dictionary and through the single-call ABI."""
    from test_oracle_dict import load_dict_cases
    dicts, payloads, cases = load_dict_cases()
    groups = {}
    for c in cases:
        groups.setdefault((c["dict"], c["level"]), []).append(c)
    for (dn, level), cs in groups.items():
        bp = z.NewBulkProcessor(dicts[dn], level)
        srcs = [payloads[c["payload"]] for c in cs]
        got = bp.CompressBatch(srcs)
        for c, g in zip(cs, got):
            assert isinstance(g, bytes), (dn, level, c["payload"], g)
            assert len(g) == c["c_len"] and hashlib.sha256(g).hexdigest()[:32] == c["c_sha256"], (dn, level, c["payload"])
        assert bp.DecompressBatch(got, [max(len(s), 1) for s in srcs]) == srcs
        one = bp.Compress(None, srcs[2])
        assert one == got[2] and bp.Decompress(None, one) == srcs[2]


def test_bulk_processor_above_fast_matches_golden(z, oracle):
    """NewBulkProcessor(dict, level) with level >= 3 (zstd_bulk.go:33-58): CDicts for ZSTD_dfast, the hash chain and the row finder
    digested on the device, payloads compressed against the ATTACHED dictionary (zsa_dict_hi.h, zsa_lazy.h) -- every
    reference-generated case of tests/golden/dict_cases_hi.json up to the attach cut-off byte-identical; payloads above it answer
    parameter_unsupported (-40: the binding's cue for its CPU path), exactly where the oracle declines too."""
    from test_oracle_dict import load_dict_cases_hi
    dicts, payloads, cases = load_dict_cases_hi()
    groups = {}
    for c in cases:
        groups.setdefault((c["dict"], c["level"]), []).append(c)
    served = declined = 0
    for (dn, level), cs in groups.items():
        ocd = oracle.cdict(dicts[dn], level)
        if not ocd:   # a ZSTD_btlazy2 row (level 10 with a dictionary below 16 KB): no digest, the binding stays on the CPU
            with pytest.raises(z.api.ZstdError):
                z.NewBulkProcessor(dicts[dn], level)
            continue
        bp = z.NewBulkProcessor(dicts[dn], level)
        srcs = [payloads[c["payload"]] for c in cs]
        got = bp.CompressBatch(srcs)
        ok, oks = [], []
        for c, s, g in zip(cs, srcs, got):
            want = oracle.compress_cdict(s, ocd)
            if not isinstance(want, bytes):
                assert isinstance(g, z.ErrorCode) and g.code == -40 and len(s) > 16384, (dn, level, c["payload"], g)
                declined += 1
                continue
            assert isinstance(g, bytes), (dn, level, c["payload"], g)
            assert len(g) == c["c_len"] and hashlib.sha256(g).hexdigest()[:32] == c["c_sha256"], (dn, level, c["payload"])
            ok.append(g); oks.append(s)
            served += 1
        oracle.free_cdict(ocd)
        assert bp.DecompressBatch(ok, [max(len(s), 1) for s in oks]) == oks
        one = bp.Compress(None, srcs[2])
        assert one == got[2] and bp.Decompress(None, one) == srcs[2]
    assert served >= 1600 and declined > 50, (served, declined)


def test_bulk_processor_seeded_payloads_above_fast(z, oracle):
    """Seeded payloads against every fixture dictionary at levels 3 .. 9: dictionary content repeated in the payload (matches that
    start in the dictionary and run on into the payload, repcodes into the dictionary), incompressible stretches, runs."""
    from test_oracle_dict import load_dict_cases
    from zstd_amd import corpus
    dicts = load_dict_cases()[0]
    rnd = random.Random(6)
    compared = 0
    for dn in sorted(dicts):
        d = dicts[dn]
        srcs = []
        for n in (1, 9, 300, 1024, 4000, 8192, 16384, 16385, 30000, 32768):
            k = rnd.randrange(5)
            if k == 0: p = corpus.frames("json", n, 1, rnd.randrange(1000))[0]
            elif k == 1: p = corpus.frames("log", n, 1, rnd.randrange(1000))[0]
            elif k == 2: p = ((d[-rnd.randrange(1, min(len(d), 3000) + 1):] + rnd.randbytes(50)) * (n // 200 + 1))[:n]
            elif k == 3: p = (bytes(rnd.choice(b"abc") for _ in range(n // 3)) + rnd.randbytes(n // 3) + d[:n - 2 * (n // 3)])[:n]
            else: p = (d[len(d) // 3:len(d) // 3 + 40] * 3 + bytes(100) + d[:60]) * (n // 280 + 1)
            srcs.append(p[:n])
        for level in (3, 4, 5, 6, 7, 9):
            ocd = oracle.cdict(d, level)
            if not ocd:
                continue
            bp = z.NewBulkProcessor(d, level)
            got = bp.CompressBatch(srcs)
            ok, oks = [], []
            for p, g in zip(srcs, got):
                want = oracle.compress_cdict(p, ocd)
                if isinstance(want, bytes):
                    assert g == want, (dn, level, len(p))
                    ok.append(g); oks.append(p); compared += 1
                else:
                    assert isinstance(g, z.ErrorCode) and g.code == -40, (dn, level, len(p), g)
            oracle.free_cdict(ocd)
            assert bp.DecompressBatch(ok, [max(len(p), 1) for p in oks]) == oks
    assert compared > 350


def test_streams_longer_than_their_input_stay_inside_the_frame(z, oracle):
    """A Huffman table taken over from a dictionary (up to 1 KiB of literals it is used without an estimate,
    huf_compress.c:1336-1341) can code literals a third LONGER than they are; the reference's stream writers stop at the end of
    their buffer and the literals go out raw.  The device writers must stop as well: every frame sits in a slot of exactly its
    bound, with guard bytes behind it that have to survive, and equals the oracle's frame."""
    import torch
    from test_oracle_dict import load_dict_cases
    dicts = load_dict_cases()[0]
    rnd = random.Random(41)
    FS, N = 1000, 256
    payloads = [rnd.randbytes(FS) if i % 3 == 0 else bytes(rnd.randrange(160, 256) for _ in range(FS)) if i % 3 == 1
                else (dicts["trained_json_8k"][20 * i:20 * i + 150] + rnd.randbytes(FS))[:FS] for i in range(N)]
    assert all(len(p) == FS for p in payloads)
    src = torch.frombuffer(bytearray(b"".join(payloads)), dtype=torch.uint8).cuda()
    bound = z.CompressBound(FS)
    stride = bound + 512
    for dn in ("trained_json_8k", "trained_log_4k", "ref_test_dict"):
        for level in (1, 3, 5):
            bp = z.NewBulkProcessor(dicts[dn], level)
            cbuf = torch.full((N * stride,), 0xA5, dtype=torch.uint8, device="cuda")
            enc = z.DeviceBatch(N)
            enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound)
            enc.compress(level, bp.cDict)
            cs = enc.results()
            ch = cbuf.cpu().numpy()
            ocd = oracle.cdict(dicts[dn], level)
            for i in range(N):
                assert 0 < cs[i] <= bound, (dn, level, i, cs[i])
                assert bool((ch[i * stride + bound:(i + 1) * stride] == 0xA5).all()), (dn, level, i, "bytes behind the frame's capacity were written")
                assert ch[i * stride:i * stride + cs[i]].tobytes() == oracle.compress_cdict(payloads[i], ocd), (dn, level, i)
            oracle.free_cdict(ocd)


def test_bulk_processor_seeded_payloads(z, oracle):
    """Seeded payloads of 1 B .. 300 KB against every fixture dictionary, at the two ZSTD_fast levels the engine serves with a
    dictionary: attached (<= 8 KiB), external segment (above; multi-block payloads, steps that have grown over incompressible
    stretches before dictionary content turns up, matches that run from the dictionary into the payload) and re-loaded
    dictionaries -- each frame byte-identical to the oracle's and round-tripped."""
    from test_oracle_dict import load_dict_cases
    from zstd_amd import corpus
    dicts = load_dict_cases()[0]
    rnd = random.Random(5)
    for dn in sorted(dicts):
        d = dicts[dn]
        srcs = []
        for n in (1, 300, 1024, 8192, 8193, 20000, 131084, 201820, 300000):
            k = rnd.randrange(4)
            if k == 0: p = corpus.frames("json", n, 1, rnd.randrange(1000))[0]
            elif k == 1: p = corpus.frames("log", n, 1, rnd.randrange(1000))[0]
            elif k == 2: p = ((d[-rnd.randrange(1, min(len(d), 3000) + 1):] + rnd.randbytes(50)) * (n // 200 + 1))[:n]
            else: p = (bytes(rnd.choice(b"abc") for _ in range(n // 3)) + rnd.randbytes(n // 3) + d[:n - 2 * (n // 3)])[:n]
            srcs.append(p)
        for level in (1, -1):
            bp = z.NewBulkProcessor(d, level)
            cd = oracle.cdict(d, level)
            got = bp.CompressBatch(srcs)
            for p, g in zip(srcs, got):
                assert g == oracle.compress_cdict(p, cd), (dn, level, len(p))
            oracle.free_cdict(cd)
            assert bp.DecompressBatch(got, [max(len(p), 1) for p in srcs]) == srcs


def test_bulk_processor_go_cases(z, oracle):
    # zstd_bullk_test.go:62-81, 83-99, 101-126, 182-199, 227-250
    import base64
    from golden_inputs import REF_TEST_DICT_B64, REF_TEST_PAYLOAD
    d = base64.b64decode(REF_TEST_DICT_B64)
    bp = z.NewBulkProcessor(d, z.BestSpeed)
    rnd = random.Random(3)
    words = ["We", "are", "building", "a platform", "that", "engineers", "love", "to", "use", "Join", "us", "and", "help", "usher", "in", "the", "future"]
    texts = [" ".join(rnd.choice(words) for _ in range(10 + rnd.randrange(100))).encode() for _ in range(100)]
    comp = bp.CompressBatch(texts)
    cd = oracle.cdict(d, 1)
    for t, c in zip(texts, comp):
        assert c == oracle.compress_cdict(t, cd)
    assert bp.DecompressBatch(list(reversed(comp))) == list(reversed(texts))        # :201-225
    for bad in (b"", None):
        with pytest.raises(z.api.ZstdError):
            z.NewBulkProcessor(bad if bad is not None else b"", z.BestSpeed)        # ErrEmptyDictionary
    e = bp.Compress(None, b"")
    assert len(e) >= 4 and bp.Decompress(None, e) == b""
    with pytest.raises(z.api.ZstdError):
        bp.Decompress(None, b"")                                                    # ErrEmptySlice
    c = bp.Compress(None, REF_TEST_PAYLOAD)
    assert c == oracle.compress_cdict(REF_TEST_PAYLOAD, cd)
    # a frame that names a dictionary fails without it / with another one, like the reference
    r = z.DecompressBatch([c], [200])[0]
    assert isinstance(r, z.ErrorCode) and r.code == oracle.decompress(c, 200)
    other = z.NewBulkProcessor(d[:4] + (77).to_bytes(4, "little") + d[8:], 1)
    r = other.DecompressBatch([c], [200])[0]
    assert isinstance(r, z.ErrorCode) and r.code == -32
    # corrupted dictionary -> ErrBadDictionary (zstd_bulk.go:44-46)
    with pytest.raises(z.api.ZstdError):
        z.NewBulkProcessor(d[:40] + bytes([d[40] ^ 0xFF]) + d[41:60], 1)
    # highly compressible frame made without the dictionary decodes through the DDict path when dst is given (:227-250)
    big = z.CompressLevel(None, bytes(1000 * 1000), 1)
    assert bp.DecompressBatch([big], [1000 * 1000])[0] == bytes(1000 * 1000)


def test_bulk_small_payload_batch(z, oracle):
    # BASELINE configs[2] shape at test size: 1 KiB JSON payloads with a trained dictionary, one launch
    from test_oracle_dict import load_dict_cases
    from zstd_amd import corpus
    dicts, _, _ = load_dict_cases()
    d = dicts["trained_json_8k"]
    bp = z.NewBulkProcessor(d, 1)
    frames = corpus.frames("json", 1024, 4096, 4242)
    got = bp.CompressBatch(frames)
    cd = oracle.cdict(d, 1)
    for i in range(0, 4096, 37):
        assert got[i] == oracle.compress_cdict(frames[i], cd)
    assert all(isinstance(g, bytes) for g in got)
    assert bp.DecompressBatch(got, [1024] * len(frames)) == frames
    assert sum(map(len, got)) < 0.45 * 1024 * 4096


def test_large_host_batch_staged_by_helper_threads(z, oracle):
    # host batches above 8 MiB are staged piece by piece by helper threads and their compressed output is packed on the
    # device before it travels back (zsa_lib.hip: zsa_run_host_batch); ragged sizes, empty payloads, incompressible ones
    # (output fills its capacity: no packing), a failing frame in the middle
    import random
    from zstd_amd import corpus
    rnd = random.Random(77)
    frames = corpus.frames("log", 131072, 90, 901) + corpus.frames("json", 50000, 40, 902)
    frames += [b"", b"x", rnd.randbytes(70000), bytes(200000)] + [f[:rnd.randrange(1, 60000)] for f in corpus.frames("log", 60000, 30, 903)]
    rnd.shuffle(frames)
    assert sum(map(len, frames)) > (12 << 20)
    got = z.CompressBatch(frames, 1)
    for i in range(0, len(frames), 7):
        assert got[i] == oracle.compress(frames[i], 1), i
    assert z.DecompressBatch(got, [len(f) for f in frames]) == frames
    caps = [len(f) for f in frames]
    bad = next(i for i, f in enumerate(frames) if len(f) > 1000)
    caps[bad] -= 1                                             # destination one byte short: that frame alone fails
    back = z.DecompressBatch(got, caps)
    assert isinstance(back[bad], z.ErrorCode)
    assert all(b == f for i, (b, f) in enumerate(zip(back, frames)) if i != bad)
    rand = [rnd.randbytes(131072) for _ in range(80)]          # nothing to pack: output as large as the capacities
    out = z.CompressBatch(rand, 1)
    assert out[3] == oracle.compress(rand[3], 1)
    assert z.DecompressBatch(out, [131072] * 80) == rand


def test_large_host_batches_from_many_threads(z, oracle):
    # four caller threads, each with its own staging, stream and helper threads, large batches at the same time
    import threading
    from zstd_amd import corpus
    sets = [corpus.frames("log", 131072, 80, 910 + t) + corpus.frames("json", 30000, 60, 920 + t) for t in range(4)]
    want = [z.CompressBatch(fs, 1) for fs in sets]
    assert want[0][5] == oracle.compress(sets[0][5], 1)
    errors = []
    def work(t):
        try:
            for _ in range(2):
                got = z.CompressBatch(sets[t], 1)
                assert got == want[t]
                assert z.DecompressBatch(got, [len(f) for f in sets[t]]) == sets[t]
        except BaseException as e:   # noqa: BLE001
            errors.append((t, repr(e)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th: x.start()
    for x in th: x.join()
    assert not errors, errors


def test_dfast_levels_match_golden(z, oracle):
    """levels 3 and 4 (and 2 at <= 256 KB, 0 = default) -- strategy ZSTD_dfast -- byte-identical to the
    reference-generated fixture, multi-block frames included (1 MiB at level 3 is BASELINE config 5's chunk)"""
    from zstd_amd import corpus
    g = json.load(open(os.path.join(HERE, "golden", "dfast_frames.json")))
    by_level = {}
    for k in g["cases"]:
        by_level.setdefault(k["level"], []).append(k)
    for level, items in by_level.items():
        srcs = [bytes.fromhex(k["src_hex"]) for k in items]
        got = z.CompressBatch(srcs, level)
        for k, c in zip(items, got):
            assert isinstance(c, bytes) and c.hex() == k["c_hex"], (k["name"], level)
        assert z.DecompressBatch(got, [max(len(s), 1) for s in srcs]) == srcs
    for t in g["tables"]:
        frames = corpus.frames(t["kind"], t["frame_size"], t["n"], t["seed"])
        got = z.CompressBatch(frames, t["level"])
        for c, (clen, chash, _) in zip(got, t["rows"]):
            assert isinstance(c, bytes) and len(c) == clen and hashlib.sha256(c).hexdigest()[:32] == chash, (t["kind"], t["frame_size"], t["level"])
        assert z.DecompressBatch(got, [t["frame_size"]] * t["n"]) == frames
    # one-shot entry point at the Go default-adjacent level, and ragged sizes against the oracle
    d = corpus.frames("json", 50000, 1, 21)[0]
    assert z.CompressLevel(None, d, 3) == oracle.compress(d, 3)
    rnd = random.Random(5)
    ragged = [corpus.frames("log", rnd.choice([1, 8, 9, 300, 4000, 16384, 16385, 100000, 131073]), 1, 30 + i)[0] for i in range(24)]
    for level in (3, 4):
        got = z.CompressBatch(ragged, level)
        for f, c in zip(ragged, got):
            want = oracle.compress(f, level)
            if isinstance(want, int):
                assert isinstance(c, z.ErrorCode) and str(c) == "Unsupported parameter"
            else:
                assert c == want, (len(f), level)


def test_lazy_levels_match_golden(z, oracle):
    """the levels whose strategy is ZSTD_greedy / ZSTD_lazy / ZSTD_lazy2 -- 5 is the default of the
    reference's Compress() (zstd.go:23,90-92) -- byte-identical to the reference-generated fixture: hash-chain finder
    (windowLog <= 14) and row-hash finder, single- and multi-block frames (repeat / cost-selected sequence tables)"""
    from zstd_amd import corpus
    g = json.load(open(os.path.join(HERE, "golden", "lazy_frames.json")))
    by_level = {}
    for k in g["cases"]:
        by_level.setdefault(k["level"], []).append(k)
    for level, items in by_level.items():
        srcs = [bytes.fromhex(k["src_hex"]) for k in items]
        got = z.CompressBatch(srcs, level)
        for k, c in zip(items, got):
            assert isinstance(c, bytes) and c.hex() == k["c_hex"], (k["name"], level)
        assert z.DecompressBatch(got, [max(len(s), 1) for s in srcs]) == srcs
    for t in g["tables"]:
        frames = corpus.frames(t["kind"], t["frame_size"], t["n"], t["seed"])
        got = z.CompressBatch(frames, t["level"])
        for c, (clen, chash, _) in zip(got, t["rows"]):
            assert isinstance(c, bytes) and len(c) == clen and hashlib.sha256(c).hexdigest()[:32] == chash, (t["kind"], t["frame_size"], t["level"])
        assert z.DecompressBatch(got, [t["frame_size"]] * t["n"]) == frames
    # the default-level one-shot entry point, and ragged sizes and shapes against the oracle
    d = corpus.frames("json", 50000, 1, 21)[0]
    assert z.Compress(None, d) == oracle.compress(d, 5)
    rnd = random.Random(6)
    ragged = [corpus.frames("log", rnd.choice([1, 8, 9, 15, 16, 17, 300, 4000, 16384, 16385, 100000, 131073, 262145]), 1, 30 + i)[0] for i in range(24)]
    ragged += [bytes(200000), rnd.randbytes(70000), rnd.randbytes(3000) + bytes(150000) + rnd.randbytes(3000), b"ab" * 40000,
               bytes(rnd.choice(b"abcde") for _ in range(50000))]
    for level in (5, 6, 8, 12):
        got = z.CompressBatch(ragged, level)
        for f, c in zip(ragged, got):
            want = oracle.compress(f, level)
            if isinstance(want, int):
                assert isinstance(c, z.ErrorCode) and str(c) == "Unsupported parameter"   # level 12 is btlazy2 / btopt at <= 256 KB
            else:
                assert c == want, (len(f), level)
                assert z.Decompress(None, c) == f


def test_high_levels_on_large_inputs(z, oracle):
    # levels 9-12 on inputs above 512 KiB: ZSTD_lazy2 with tables of 8-21 MiB per frame in flight (round 1 refused these);
    # the launch is sized to what its frames need
    from zstd_amd import corpus
    frames = [corpus.frames("log", 600000, 1, 61)[0], corpus.frames("json", 1100000, 1, 62)[0], corpus.frames("log", 3000, 1, 63)[0]]
    for level in (9, 10, 12):
        got = z.CompressBatch(frames, level)
        assert got[0] == oracle.compress(frames[0], level) and got[1] == oracle.compress(frames[1], level), level
        # (at 3000 bytes these levels are ZSTD_btlazy2: outside the engine, and the oracle says the same)
        assert isinstance(got[2], z.ErrorCode) and str(got[2]) == "Unsupported parameter" and oracle.compress(frames[2], level) == -40
        assert z.DecompressBatch(got[:2], [len(f) for f in frames[:2]]) == frames[:2]


def test_frames_longer_than_the_window(z, oracle):
    # 3 and 5 MB frames: longer than the window of every level used here (512 KiB .. 4 MiB), 24-39 blocks each
    from zstd_amd import corpus
    frames = [corpus.frames("log", 3000000, 1, 71)[0], corpus.frames("json", 5000000, 1, 72)[0]]
    for level in (1, 3, 5, 12):
        got = z.CompressBatch(frames, level)
        for f, c in zip(frames, got):
            assert c == oracle.compress(f, level), (len(f), level)
        assert z.DecompressBatch(got, [len(f) for f in frames]) == frames


def test_ctx_go_cases(z, oracle):
    # zstd_ctx_test.go:9-40 (round trip, Decompress with a too-small dst), :42-66 (levels), :75-119 (empty input / output),
    # :121-142 (130 KB payload into a short dst)
    c = z.NewCtx()
    hello = b"Hello World!"
    comp = c.CompressLevel(None, hello, 1)
    assert comp == oracle.compress(hello, 1)
    assert c.Decompress(None, comp) == hello and c.Decompress(bytearray(2), comp) == hello
    for level in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
        for size in (1, 10, 100, 1000, 100000):
            d = bytes((i * 7 + (i >> 3)) & 0xFF for i in range(size))
            want = oracle.compress(d, level)
            if isinstance(want, int):
                continue      # levels 9-10 below 16 KB are ZSTD_btlazy2
            got = c.CompressLevel(bytearray(z.CompressBound(size)), d, level)
            assert bytes(got) == want and c.Decompress(None, bytes(got)) == d
    e = c.CompressLevel(None, b"", 1)
    assert e == oracle.compress(b"", 1) and c.Decompress(None, e) == b""
    with pytest.raises(z.api.ZstdError):
        c.Decompress(None, b"")
    big = bytes(130 * 1024)
    assert c.Decompress(bytearray(1), c.CompressLevel(None, big, 1)) == big
    dflt = c.Compress(None, hello)                # zstd_ctx.go:53-55: DefaultCompression (5)
    assert dflt == oracle.compress(hello, 5) and c.Decompress(None, dflt) == hello


def test_bulk_destination_cases(z, oracle):
    # zstd_bullk_test.go:128-141, 143-156 (Compress into a given / a too-small destination), :158-168, 170-180 (Decompress)
    import base64
    from golden_inputs import REF_TEST_DICT_B64, REF_TEST_PAYLOAD
    bp = z.NewBulkProcessor(base64.b64decode(REF_TEST_DICT_B64), z.BestSpeed)
    want = bp.Compress(None, REF_TEST_PAYLOAD)
    assert bp.Compress(bytearray(z.CompressBound(len(REF_TEST_PAYLOAD))), REF_TEST_PAYLOAD) == want
    assert bp.Compress(bytearray(1), REF_TEST_PAYLOAD) == want                  # dst too small: a new buffer is used
    assert bp.Decompress(bytearray(len(REF_TEST_PAYLOAD)), want) == REF_TEST_PAYLOAD
    assert bp.Decompress(bytearray(1), want) == REF_TEST_PAYLOAD                # dst too small: sized from the frame header


def test_many_threads_single_call_api(z, oracle):
    """zstd.go:95-129: the one-shot calls are lock-free and run on any goroutine.  64 threads call zsa_compress /
    zsa_decompress at once; every result equals the oracle's, and the per-thread staging memory is returned when the
    threads end: after a first round (the HIP runtime keeps what its own stream and queue pools grew to under 64 concurrent
    callers), further rounds of 64 new threads leave hipMemGetInfo where it was."""
    import threading
    import torch
    from zstd_amd import corpus
    payloads = corpus.frames("json", 3000, 64, 11)
    want = [oracle.compress(p, 1) for p in payloads]
    errs = []

    def work(i):
        try:
            for k in range(3):
                p = payloads[(i + k) % 64]
                c = z.CompressLevel(None, p, 1)
                assert c == want[(i + k) % 64]
                assert z.Decompress(None, c) == p
        except Exception as e:   # noqa: BLE001
            errs.append((i, repr(e)))

    def round_of_threads():
        ts = [threading.Thread(target=work, args=(i,)) for i in range(64)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs[:3]
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    free0 = round_of_threads()
    round_of_threads()
    free2 = round_of_threads()
    # (the runtime's pools move by some tens of MiB from round to round; the staging of 128 leaked threads would be > 2 GiB)
    assert free2 >= free0 - (256 << 20), "per-thread staging leaked: %d MiB over 128 threads" % ((free0 - free2) >> 20)


def test_capacity_contract(z, oracle):
    """Host entry points: any capacity the frame fits is served, as ZSTD_compress does (zstd.go:95-129 call-site contract;
    a frame that does not fit gives the reference's "Destination buffer is too small").  Device entry points keep
    dstCapacity >= zsa_compressBound(srcSize) as their contract (include/zstd_amd.h section 3)."""
    import torch
    from zstd_amd._lib import lib
    p = b"abcd" * 1000
    want = oracle.compress(p, 1)
    bound = z.CompressBound(len(p))
    out = ctypes.create_string_buffer(bound)
    src = ctypes.create_string_buffer(p, len(p))
    for cap in (bound, bound - 1, len(want) + 1, len(want)):
        r = lib.zsa_compress(ctypes.cast(out, ctypes.c_void_p), cap, ctypes.cast(src, ctypes.c_void_p), len(p), 1)
        assert not lib.zsa_isError(r) and out.raw[:r] == want, cap
    r = lib.zsa_compress(ctypes.cast(out, ctypes.c_void_p), len(want) - 1, ctypes.cast(src, ctypes.c_void_p), len(p), 1)
    assert lib.zsa_isError(r) and z.IsDstSizeTooSmallError(z.ErrorCode(r - 2 ** 64)) and lib.zsa_engine_failure() == 0
    # device path: below the bound is refused
    dsrc = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
    dout = torch.empty(bound, dtype=torch.uint8, device="cuda")
    b = z.DeviceBatch(1)
    b.set_frames(dsrc.data_ptr(), [0], [len(p)], dout.data_ptr(), [0], [bound - 1])
    b.compress(1)
    r = b.results()[0]
    assert lib.zsa_isError(r) and z.IsDstSizeTooSmallError(z.ErrorCode(r - 2 ** 64))
    b.close()


def test_level1_chunk_and_general_paths(z, oracle):
    """The ZSTD_fast kernel has two routes through one frame (zsa_parse.h): the 64-position chunk path with the 4-bit
    filter in LDS (hashLog <= 13: sources up to 128 KiB at level 1) and the general window evaluation it backs off to
    (and which larger sources use throughout).  Frames on both sides of that limit, ragged ones, degenerate ones and
    a multi-block one: byte-identical at levels 1, 2 and -1."""
    from zstd_amd import corpus
    rnd = random.Random(3)
    frames = corpus.frames("log", 131072, 6, 42) + corpus.frames("json", 65536, 6, 1234) + corpus.frames("log", 262144, 2, 43) \
        + corpus.frames("json", 1024, 16, 7) + [b"", b"a", bytes(5000), rnd.randbytes(3000), b"ab" * 5000,
                                                 b"".join(rnd.choice([b"alpha ", b"beta ", b"gamma "]) for _ in range(9000)),
                                                 corpus.frames("log", 600000, 1, 9)[0]]
    for level in (1, 2, -1):
        _check_compress(z, oracle, frames, level)


def test_chains_of_short_dependent_matches(z, oracle):
    """Decoder hand-over of global memory between lanes (ZSA_DEP_FENCE = wv_wave_fence): batches in which nearly every one of
    the 64 sequences is a short match whose source is the output of the match before it (offset < 10, often below the match
    length), so a missed ordering between a lane's store and the next lane's load would corrupt the output.  Frames from the
    oracle at levels 1 and 5 decoded on the GPU, and compressed on the GPU byte-identically."""
    rnd = random.Random(11)
    frames = []
    for _ in range(12):
        d = bytearray(rnd.randbytes(16))
        while len(d) < 60000:
            off, ln = rnd.randrange(1, 10), rnd.randrange(4, 12)
            for _ in range(ln):
                d.append(d[-off])
            d.append(rnd.randrange(256))
        frames.append(bytes(d))
    for level in (1, 5):
        comp = [oracle.compress(f, level) for f in frames]
        assert all(len(c) < len(f) // 2 for c, f in zip(comp, frames))     # (matches dominate)
        back = z.DecompressBatch(comp, [len(f) for f in frames])
        assert all(b == f for b, f in zip(back, frames))
        _check_compress(z, oracle, frames, level)


_SPLIT = r"""
import sys
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import torch, zstd_amd
from zstd_amd import corpus
from conftest import _Lib
orc = _Lib(%(root)r + "/oracle/liboracle.so", "zo_")
frames = corpus.frames("log", 131072, 700, 42) + [b"", b"x"] + corpus.frames("json", 65536, 100, 3)
comp = zstd_amd.CompressBatch(frames, 1)
assert all(isinstance(c, bytes) for c in comp)
for i in range(0, len(frames), 37):
    assert comp[i] == orc.compress(frames[i], 1), i
back = zstd_amd.DecompressBatch(comp, [max(len(f), 1) for f in frames])
assert back == frames
print("split ok")
"""


def test_host_batch_split_over_devices():
    """ZSTD_AMD_DEVICES: one large host batch cut into per-device shares run by device-bound worker threads.  On this one-GPU
    box the three shares all run on device 0 (ZSTD_AMD_DEVICE_ALIAS, a test knob): what is checked is the split, the workers
    and the reassembly -- frames byte-identical to the oracle's, round trip exact, empty and tiny frames in the mix."""
    import subprocess, sys
    root = os.path.dirname(HERE)
    env = dict(os.environ, ZSTD_AMD_DEVICES="3", ZSTD_AMD_DEVICE_ALIAS="1")
    r = subprocess.run([sys.executable, "-c", _SPLIT % {"root": root}], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "split ok" in r.stdout, r.stdout + r.stderr
