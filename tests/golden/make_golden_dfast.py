"""Regenerates tests/golden/dfast_frames.json from the REFERENCE's own C (oracle/_ref/libzstd_ref.so; dev container
only): frames compressed at the levels whose strategy is ZSTD_dfast (3, 4, and 2 at <= 256 KB; 0 = default = 3).
Same conventions as make_golden.py: inputs named by (corpus kind, seed, frame size, index) or given in hex, outputs by
(cSize, sha256) or hex."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import comp, corpus, small_inputs


def main():
    kats = []
    for name, d in small_inputs():
        if len(d) > 2000:
            continue
        for lvl in (3, 0):   # level 4 is ZSTD_greedy at <= 16 KB
            c = comp(d, lvl)
            kats.append({"name": name, "level": lvl, "src_hex": d.hex(), "c_hex": c.hex()})
    tables = []
    for kind, fs, n, seed, levels in (("log", 131072, 16, 42, (3, 4)), ("json", 65536, 16, 1234, (3, 4)), ("log", 262144, 8, 43, (2, 3)),
                                      ("json", 1024, 32, 7, (3,)), ("log", 20000, 8, 5, (3, 4)), ("json", 1 << 20, 2, 11, (3, 4)),
                                      ("log", 300000, 2, 12, (0, 4))):
        frames = corpus.frames(kind, fs, n, seed)
        for lvl in levels:
            rows = []
            for f in frames:
                c = comp(f, lvl)
                rows.append([len(c), hashlib.sha256(c).hexdigest()[:32], hashlib.sha256(f).hexdigest()[:32]])
            tables.append({"kind": kind, "frame_size": fs, "n": n, "seed": seed, "level": lvl, "rows": rows})
    json.dump({"libzstd": "1.5.5", "generator": "tests/golden/make_golden_dfast.py", "columns": ["c_len", "c_sha256_128", "src_sha256_128"],
               "cases": kats, "tables": tables}, open(os.path.join(HERE, "dfast_frames.json"), "w"), indent=0)
    print("wrote", len(kats), "KATs and", sum(t["n"] for t in tables), "corpus rows")


if __name__ == "__main__":
    main()
