"""Regenerates tests/golden/lazy_frames.json from the REFERENCE's own C (oracle/_ref/libzstd_ref.so; dev container
only): frames compressed at levels whose strategy is ZSTD_greedy / ZSTD_lazy / ZSTD_lazy2 (5 = the reference's default
level, zstd.go:23; 4..12 depending on the size class, clevels.h).  Same conventions as make_golden.py: inputs named by
(corpus kind, seed, frame size, index) or given in hex, outputs by (cSize, sha256) or hex."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import comp, corpus, small_inputs


def main():
    kats = []
    for name, d in small_inputs():
        if len(d) > 2000:
            continue
        for lvl in (5, 7):   # <= 16 KB: lazy and lazy2 with the hash-chain finder
            c = comp(d, lvl)
            kats.append({"name": name, "level": lvl, "src_hex": d.hex(), "c_hex": c.hex()})
    tables = []
    for kind, fs, n, seed, levels in (("log", 131072, 16, 42, (5, 6, 7)), ("json", 65536, 16, 1234, (5, 10)), ("log", 262144, 8, 43, (4, 5, 6, 9)),
                                      ("json", 1024, 32, 7, (5, 4, 8)), ("log", 20000, 8, 5, (5, 6)), ("json", 1 << 20, 2, 11, (5, 6, 8)),
                                      ("log", 300000, 2, 12, (5, 7)), ("log", 16384, 8, 9, (5, 6))):
        frames = corpus.frames(kind, fs, n, seed)
        for lvl in levels:
            rows = []
            for f in frames:
                c = comp(f, lvl)
                rows.append([len(c), hashlib.sha256(c).hexdigest()[:32], hashlib.sha256(f).hexdigest()[:32]])
            tables.append({"kind": kind, "frame_size": fs, "n": n, "seed": seed, "level": lvl, "rows": rows})
    json.dump({"libzstd": "1.5.5", "generator": "tests/golden/make_golden_lazy.py", "columns": ["c_len", "c_sha256_128", "src_sha256_128"],
               "cases": kats, "tables": tables}, open(os.path.join(HERE, "lazy_frames.json"), "w"), indent=0)
    print("wrote", len(kats), "KATs and", sum(t["n"] for t in tables), "corpus rows")


if __name__ == "__main__":
    main()
