#!/usr/bin/env python3
"""GPU box: seeded random inputs through zsa_compress_batch at many levels, every frame compared with the oracle
(byte-identical or the same "Unsupported parameter") and round-tripped.  usage: fuzz_gpu_levels.py [trials [seed]]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import _Lib
import zstd_amd
orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
inputs = []
for trial in range(trials):
    n = rnd.choice([0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 256, 1000, 1024, 4096, 9000, 16384, 16385, 20000, 70000, 131072, 131073, 200000, 300000, 262144, 600000, 1100000])
    parts = []
    while sum(map(len, parts)) < n:
        k = rnd.randrange(6); m = rnd.randrange(1, 4000)
        parts.append([rnd.randbytes(m), bytes(m), rnd.randbytes(rnd.randrange(1, 30)) * (m // 8 + 1),
                      (b"key=%d value=%x\n" % (rnd.randrange(99), rnd.getrandbits(20))) * (m // 16 + 1),
                      bytes(rnd.choice(b"abcde") for _ in range(m)),
                      b"".join(rnd.choice([b"alpha ", b"beta ", b"gamma ", b"delta=%d " % rnd.randrange(50)]) for _ in range(m // 6 + 1))][k])
    inputs.append(b"".join(parts)[:n])
bad = checked = refused = 0
for level in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 0, -4):
    got = zstd_amd.CompressBatch(inputs, level)
    ok = []
    for d, c in zip(inputs, got):
        want = orc.compress(d, level)
        if isinstance(want, int):
            if not (isinstance(c, zstd_amd.ErrorCode) and c.code == want): bad += 1; print("MISMATCH (error)", len(d), level, c)
        else:
            checked += 1
            if isinstance(c, zstd_amd.ErrorCode): refused += 1; bad += 1; print("REFUSED", len(d), level, c)   # every level here is served at every size: a refusal is a failure
            elif c != want: bad += 1; print("MISMATCH", len(d), level, c if isinstance(c, zstd_amd.ErrorCode) else "")
            else: ok.append((d, c))
    back = zstd_amd.DecompressBatch([c for _, c in ok], [max(len(d), 1) for d, _ in ok])
    bad += sum(1 for (d, _), b in zip(ok, back) if b != d)
print("frames checked", checked, "refused", refused, "mismatches (refusals included)", bad)
sys.exit(1 if bad else 0)
