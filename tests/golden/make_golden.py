"""Regenerates tests/golden/*.json from the REFERENCE's own C (oracle/_ref/libzstd_ref.so, built by
oracle/Makefile from /root/reference; run in the dev container only).  The fixtures are data: inputs are
either literal (hex) or named by (corpus kind, seed, frame size, index) of zstd_amd/corpus.py, outputs are
the reference's compressed bytes (small cases) or (cSize, sha256) (corpus frames)."""
import ctypes, hashlib, json, os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from zstd_amd import corpus
sys.path.insert(0, os.path.dirname(HERE))
from golden_inputs import small_inputs

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so"))
ref.ZSTD_compress.restype = ctypes.c_size_t
ref.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
ref.ZSTD_compressBound.restype = ctypes.c_size_t
ref.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
ref.ZSTD_versionString.restype = ctypes.c_char_p
assert ref.ZSTD_versionString() == b"1.5.5"


def comp(d, level=1):
    cap = ref.ZSTD_compressBound(len(d))
    buf = ctypes.create_string_buffer(cap + 1)
    r = ref.ZSTD_compress(buf, cap, d, len(d), level)
    assert r < 2 ** 63, r
    return buf.raw[:r]


def main():
    kats = []
    for name, d in small_inputs():
        c = comp(d, 1)
        e = {"name": name, "level": 1, "src_sha256": hashlib.sha256(d).hexdigest(), "src_len": len(d),
             "c_len": len(c), "c_sha256": hashlib.sha256(c).hexdigest()}
        if len(d) <= 2000:
            e["src_hex"] = d.hex()
            e["c_hex"] = c.hex()
        kats.append(e)
    for lvl in (-1, -5, 2):
        d = (b"the quick brown fox jumps over the lazy dog. " * 60)[:2000]
        c = comp(d, lvl)
        kats.append({"name": "text2k_L%d" % lvl, "level": lvl, "src_hex": d.hex(), "src_len": len(d), "src_sha256": hashlib.sha256(d).hexdigest(),
                     "c_hex": c.hex(), "c_len": len(c), "c_sha256": hashlib.sha256(c).hexdigest()})
    json.dump({"libzstd": "1.5.5", "generator": "tests/golden/make_golden.py", "cases": kats},
              open(os.path.join(HERE, "kat_small.json"), "w"), indent=0)

    tables = []
    for kind, fs, n, seed in (("log", 131072, 64, 42), ("json", 65536, 64, 1234), ("log", 262144, 16, 43), ("json", 1024, 64, 7),
                              ("log", 20000, 16, 5), ("log", 12000, 8, 6), ("json", 600000, 2, 9)):
        rows = []
        for i, f in enumerate(corpus.frames(kind, fs, n, seed)):
            c = comp(f, 1)
            rows.append([len(c), hashlib.sha256(c).hexdigest()[:32], hashlib.sha256(f).hexdigest()[:32]])
        tables.append({"kind": kind, "frame_size": fs, "n": n, "seed": seed, "level": 1, "rows": rows})
    json.dump({"libzstd": "1.5.5", "generator": "tests/golden/make_golden.py", "columns": ["c_len", "c_sha256_128", "src_sha256_128"], "tables": tables},
              open(os.path.join(HERE, "corpus_frames.json"), "w"), indent=0)
    print("wrote", len(kats), "KATs and", sum(t["n"] for t in tables), "corpus rows")


if __name__ == "__main__":
    main()
