"""Regenerates tests/golden/dict_cases_hi.json from the REFERENCE's own C (oracle/_ref/libzstd_ref.so; dev container only):
ZSTD_compress_usingCDict at the levels whose dictionary strategy is ZSTD_dfast, ZSTD_greedy, ZSTD_lazy or ZSTD_lazy2
(3 .. 10, NewBulkProcessor(dict, level) with level >= 3, zstd_bulk.go:33-58), payloads of up to 40 KiB -- the sizes around the
reference's attach cut-offs (16 KiB / 32 KiB, zstd_compress.c:2241-2252).  Same dictionaries as make_golden_dict.py (their
bytes are kept in dict_cases.json).  Outputs: compressed bytes (small) or (c_len, sha256)."""
import base64, ctypes, hashlib, json, os, random, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from zstd_amd import corpus
from golden_inputs import dict_payloads, raw_dicts, dict_payloads_hi

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so"))
S, V, C = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p
ref.ZSTD_createCDict.restype = V; ref.ZSTD_createCDict.argtypes = [C, S, ctypes.c_int]
ref.ZSTD_createCCtx.restype = V
ref.ZSTD_compress_usingCDict.restype = S; ref.ZSTD_compress_usingCDict.argtypes = [V, V, S, C, S, V]
ref.ZSTD_freeCDict.argtypes = [V]
ref.ZSTD_compressBound.restype = S; ref.ZSTD_compressBound.argtypes = [S]
ref.ZSTD_versionString.restype = C
assert ref.ZSTD_versionString() == b"1.5.5"

LEVELS = (3, 4, 5, 6, 7, 8, 10)


def main():
    base = json.load(open(os.path.join(HERE, "dict_cases.json")))
    dicts = [(n, base64.b64decode(d["b64"])) for n, d in base["dicts"].items() if "b64" in d] + list(raw_dicts(corpus))
    payloads = dict_payloads_hi(corpus)
    cctx = ref.ZSTD_createCCtx()   # one context for all calls, as a reused Go Ctx would: the row finder's hash salt moves on
    out = {"libzstd": "1.5.5", "generator": "tests/golden/make_golden_dict_hi.py", "levels": list(LEVELS), "cases": []}
    for name, d in dicts:
        for level in LEVELS:
            cd = ref.ZSTD_createCDict(d, len(d), level)
            assert cd
            for pn, p in payloads:
                cap = ref.ZSTD_compressBound(len(p))
                buf = ctypes.create_string_buffer(cap + 1)
                r = ref.ZSTD_compress_usingCDict(cctx, buf, cap, p, len(p), cd)
                assert r < 2 ** 63
                c = buf.raw[:r]
                e = {"dict": name, "level": level, "payload": pn, "c_len": len(c), "c_sha256": hashlib.sha256(c).hexdigest()[:32]}
                if len(c) <= 64:
                    e["c_hex"] = c.hex()
                out["cases"].append(e)
            ref.ZSTD_freeCDict(cd)
    json.dump(out, open(os.path.join(HERE, "dict_cases_hi.json"), "w"), indent=0)
    print("wrote", len(out["cases"]), "dictionary cases over", len(dicts), "dictionaries")


if __name__ == "__main__":
    main()
