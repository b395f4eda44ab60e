"""Deterministic small inputs shared by tests/golden/make_golden.py and the tests (fixtures store their hashes)."""
import random


def small_inputs():
    rnd = random.Random(20260101)
    cases = [("empty", b""), ("a", b"a"), ("hello", b"Hello World!"), ("zeros1000", bytes(1000)),
             ("six", b"abcdef"), ("seven", b"abcdefg"), ("rep7", b"ab" * 40), ("zeros300k", bytes(300000)),
             ("rand300", rnd.randbytes(300)), ("rand70k", rnd.randbytes(70000)),
             ("text2k", (b"the quick brown fox jumps over the lazy dog. " * 60)[:2000]),
             ("lowent5k", bytes(rnd.choice(b"aaaaabbbcd") for _ in range(5000))),
             ("runs", b"".join(bytes([rnd.randrange(256)]) * rnd.randrange(1, 300) for _ in range(200))),
             ("rle2ndblock", rnd.randbytes(131072) + bytes(100000)),
             ("mixed200k", b"".join(rnd.choice([rnd.randbytes(rnd.randrange(1, 3000)), bytes(rnd.randrange(1, 3000)),
                                                  (b"GET /api/v1/items/%d HTTP/1.1\n" % rnd.randrange(1000)) * rnd.randrange(1, 50)])
                                     for _ in range(200))[:200000])]
    return cases


# ---- dictionary path (BulkProcessor, zstd_bulk.go) ----
# The 256-byte dictionary is DATA held by the reference's own test (zstd_bullk_test.go:12-17).
REF_TEST_DICT_B64 = (
    "N6Qw7IsuFDIdENCSQjr//////4+QlekuNkmXbUBIkIDiVRX7H4AzAFCgQCFCO9oHAAAEQEuSikaK"
    "Dg51OYghBYgBAAAAAAAAAAAAAAAAAAAAANQVpmRQGQAAAAAAAAAAAAAAAAABAAAABAAAAAgAAABo"
    "ZWxwIEpvaW4gZW5naW5lZXJzIGVuZ2luZWVycyBmdXR1cmUgbG92ZSB0aGF0IGFyZWlsZGluZyB1"
    "c2UgaGVscCBoZWxwIHVzaGVyIEpvaW4gdXNlIGxvdmUgdXMgSm9pbiB1bmQgaW4gdXNoZXIgdXNo"
    "ZXIgYSBwbGF0Zm9ybSB1c2UgYW5kIGZ1dHVyZQ==")
REF_TEST_PAYLOAD = b"We're building a platform that engineers love to use. Join us, and help usher in the future."   # zstd_bullk_test.go:31


def dict_payloads(corpus):
    """Deterministic payloads for the dictionary tests: (name, bytes).  Sizes straddle the reference's three ways of
    bringing a CDict in (attach <= 8 KiB < copy < 128 KiB <= reload, zstd_compress.c:2254-2267, 5017-5025)."""
    rnd = random.Random(20260102)
    words = ["We", "are", "building", "a platform", "that", "engineers", "love", "to", "use", "Join", "us", "and", "help",
             "usher", "in", "the", "future"]   # zstd_bullk_test.go:46
    out = [("empty", b""), ("a", b"a"), ("ref_payload", REF_TEST_PAYLOAD)]
    for i in range(6):
        out.append(("words%d" % i, " ".join(rnd.choice(words) for _ in range(10 + rnd.randrange(100))).encode()))
    log = corpus.generate("log", 300 * 1024, 1, 77).tobytes()
    js = corpus.generate("json", 1024, 24, 78).tobytes()
    for n in (7, 8, 9, 64, 200, 1000, 4000, 8192, 8193, 20000, 65536, 131071, 131072, 140000, 270000):
        out.append(("log77_%d" % n, log[:n]))
    for i in range(0, 24, 3):
        out.append(("json78_1k_%d" % i, js[i * 1024:(i + 1) * 1024]))
    out.append(("rand5000", rnd.randbytes(5000)))
    out.append(("zeros50k", bytes(50000)))
    return out


def raw_dicts(corpus):
    """Raw-content dictionaries (no entropy header), named."""
    log = corpus.generate("log", 4096, 28, 5).tobytes()
    js = corpus.generate("json", 4096, 10, 9).tobytes()
    return [("raw_7", b"abcdefg"), ("raw_9", b"abcdefghi"), ("raw_log_4k", log[1000:5096]), ("raw_json_40k", js[:40000]),
            ("raw_log_112k", log[:112 * 1024])]


def dict_payloads_hi(corpus):
    """Payloads for the dictionary tests above ZSTD_fast: (name, bytes), up to 40 KiB -- both sides of the attach cut-offs of
    ZSTD_dfast (16 KiB) and of greedy / lazy / lazy2 (32 KiB), zstd_compress.c:2241-2252."""
    rnd = random.Random(20261111)
    out = [p for p in dict_payloads(corpus) if len(p[1]) <= 20000 and p[0] not in ("words1", "words3", "words4", "json78_1k_9", "json78_1k_15", "json78_1k_21")]
    log = corpus.generate("log", 96 * 1024, 1, 91).tobytes()
    js = corpus.generate("json", 96 * 1024, 1, 92).tobytes()
    for i, n in enumerate((300, 1024, 2500, 6000, 12000, 16384, 16385, 24000, 32768, 32769, 40000)):
        src = log if i & 1 else js
        o = rnd.randrange(0, len(src) - n)
        out.append(("%s9%d_%d" % ("log" if i & 1 else "json", 1 + (i & 1), n), src[o:o + n]))
    out.append(("rep_abc_3000", (b"abcabcabd" * 400)[:3000]))
    out.append(("zeros_9000", bytes(9000)))
    return out
