#!/usr/bin/env python3
"""Dev-container check: oracle (zo_compress) vs the reference build (oracle/_ref) at the greedy / lazy / lazy2 levels (4..12 where clevels.h selects them)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import _Lib
from zstd_amd import corpus
orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")
ref = _Lib(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so"), "ZSTD_")
rnd = random.Random(11)
LEVELS = (4, 5, 6, 7, 8, 9, 10, 11, 12)
bad = n_cases = 0
def check(d, level, tag):
    global bad, n_cases
    a = ref.compress(d, level); b = orc.compress(d, level)
    if isinstance(b, int) and orc.error_name(b) == "Unsupported parameter": return   # a btlazy2 / btopt row of clevels.h
    n_cases += 1
    if isinstance(b, int) or a != b:
        bad += 1
        print("MISMATCH", tag, len(d), level, len(a), b if isinstance(b, int) else len(b))
for kind in ("log", "json"):
    for size in (100, 1000, 5000, 16384, 16385, 40000, 65536, 131072, 131073, 200000, 262144, 262145, 400000, 1 << 20, 3 << 20):
        for d in corpus.frames(kind, size, 2, 5):
            for level in LEVELS:
                try: check(d, level, kind)
                except Exception as e: print("ERR", kind, size, level, e); bad += 1
for trial in range(120):
    n = rnd.choice([0, 1, 7, 8, 9, 63, 64, 65, 255, 256, 1000, 1024, 4096, 9000, 16384, 20000, 70000, 131072, 131073, 200000, 300000, 600000])
    parts = []
    while sum(map(len, parts)) < n:
        k = rnd.randrange(5); m = rnd.randrange(1, 4000)
        parts.append([rnd.randbytes(m), bytes(m), rnd.randbytes(rnd.randrange(1, 30)) * (m // 8 + 1),
                      (b"key=%d value=%x\n" % (rnd.randrange(99), rnd.getrandbits(20))) * (m // 16 + 1),
                      bytes(rnd.choice(b"abcde") for _ in range(m))][k])
    d = b"".join(parts)[:n]
    for level in LEVELS:
        check(d, level, "rnd%d" % trial)
print("cases", n_cases, "mismatches", bad)
