#!/usr/bin/env python3
"""GPU box: seeded random dictionaries and payloads through the BulkProcessor path (zsa_createCDict / zsa_compress_batch /
zsa_decompress_batch) at levels 1 .. 10 and a negative one; every frame is compared with the oracle (byte-identical, or
"Unsupported parameter" exactly where the oracle declines: payloads above the attach cut-off at levels 3 and up) and
round-tripped through the device DDict.  usage: fuzz_gpu_dict.py [dictionaries [seed]]"""
import base64, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import _Lib
import zstd_amd
from zstd_amd import corpus
orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")
ndict = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 23)
g = json.load(open(os.path.join(ROOT, "tests", "golden", "dict_cases.json")))
fixtures = [base64.b64decode(d["b64"]) for d in g["dicts"].values() if "b64" in d]
text = corpus.generate("log", 4096, 100, 61).tobytes() + corpus.generate("json", 4096, 100, 62).tobytes()


def blob(n):
    parts = []
    while sum(map(len, parts)) < n:
        k = rnd.randrange(5); m = rnd.randrange(1, 3000)
        o = rnd.randrange(0, len(text) - m)
        parts.append([text[o:o + m], rnd.randbytes(m), bytes(m), rnd.randbytes(rnd.randrange(1, 20)) * (m // 8 + 1),
                      bytes(rnd.choice(b"abcd") for _ in range(m))][k])
    return b"".join(parts)[:n]


bad = checked = declined = 0
for t in range(ndict):
    if t < len(fixtures) and rnd.random() < 0.5:
        d = fixtures[t]
    else:
        d = blob(rnd.choice([8, 9, 60, 300, 2000, 5000, 17000, 40000, 100000]))
    payloads = []
    for _ in range(40):
        n = rnd.choice([0, 1, 8, 9, 100, 500, 1024, 3000, 8192, 8193, 16384, 16385, 20000, 32768, 32769, 50000])
        k = rnd.random()
        if k < 0.5: p = blob(n)
        elif k < 0.8 and len(d) > 16: p = ((d[rnd.randrange(len(d) - 8):][:rnd.randrange(8, 400)] + rnd.randbytes(rnd.randrange(1, 40))) * (n // 9 + 1))[:n]
        else: p = (d[-min(len(d), n // 2 + 1):] + blob(n))[:n]
        payloads.append(p)
    for level in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -3):
        ocd = orc.cdict(d, level)
        try:
            bp = zstd_amd.NewBulkProcessor(d, level)
        except zstd_amd.api.ZstdError:
            bp = None
        if bool(ocd) != (bp is not None):
            bad += 1; print("MISMATCH (digest)", len(d), level, bool(ocd), bp is not None); continue
        if bp is None:
            continue
        got = bp.CompressBatch(payloads)
        ok, oks = [], []
        for p, c in zip(payloads, got):
            want = orc.compress_cdict(p, ocd)
            checked += 1
            if isinstance(want, int):
                declined += 1
                if not (isinstance(c, zstd_amd.ErrorCode) and c.code == want): bad += 1; print("MISMATCH (error)", len(d), len(p), level, c, want)
            elif c != want:
                bad += 1; print("MISMATCH", len(d), len(p), level, c if not isinstance(c, bytes) else len(c), len(want))
            else:
                ok.append(c); oks.append(p)
        back = bp.DecompressBatch(ok, [max(len(p), 1) for p in oks])
        if back != oks: bad += 1; print("ROUND TRIP", len(d), level)
        orc.free_cdict(ocd)
    print("dictionary %d (%d bytes): %d frames so far, %d declined, %d bad" % (t, len(d), checked, declined, bad), flush=True)
print("checked", checked, "declined", declined, "mismatches", bad)
sys.exit(1 if bad else 0)
