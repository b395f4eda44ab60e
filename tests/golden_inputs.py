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


