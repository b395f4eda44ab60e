"""Regenerates tests/golden/dict_cases.json from the REFERENCE's own C (oracle/_ref/libzstd_ref.so; dev container only).
Dictionaries: the reference test-suite's 256-byte dictionary (data, zstd_bullk_test.go:12-17), two dictionaries trained
here by the reference's ZDICT_trainFromBuffer on the synthetic corpora (stored base64), and raw-content dictionaries
named by corpus slice.  Outputs: ZSTD_compress_usingCDict bytes (small) or (c_len, sha256)."""
import base64, ctypes, hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from zstd_amd import corpus
from golden_inputs import REF_TEST_DICT_B64, dict_payloads, raw_dicts

ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so"))
S, V, C = ctypes.c_size_t, ctypes.c_void_p, ctypes.c_char_p
ref.ZSTD_createCDict.restype = V; ref.ZSTD_createCDict.argtypes = [C, S, ctypes.c_int]
ref.ZSTD_createCCtx.restype = V
ref.ZSTD_compress_usingCDict.restype = S; ref.ZSTD_compress_usingCDict.argtypes = [V, V, S, C, S, V]
ref.ZSTD_freeCDict.argtypes = [V]
ref.ZSTD_compressBound.restype = S; ref.ZSTD_compressBound.argtypes = [S]
ref.ZDICT_trainFromBuffer.restype = S; ref.ZDICT_trainFromBuffer.argtypes = [V, S, C, ctypes.POINTER(S), ctypes.c_uint]
ref.ZSTD_versionString.restype = C
assert ref.ZSTD_versionString() == b"1.5.5"


def train(samples, cap):
    buf = b"".join(samples)
    sizes = (S * len(samples))(*[len(s) for s in samples])
    out = ctypes.create_string_buffer(cap)
    n = ref.ZDICT_trainFromBuffer(out, cap, buf, sizes, len(samples))
    assert n < 2 ** 63
    return out.raw[:n]


def main():
    log = corpus.generate("log", 4096, 300, 5).tobytes()
    js = corpus.generate("json", 4096, 300, 9).tobytes()
    dicts = [("ref_test_dict", base64.b64decode(REF_TEST_DICT_B64), True)]
    dicts.append(("trained_json_8k", train([l + b"\n" for l in js.split(b"\n") if l][:6000], 8192), True))
    dicts.append(("trained_log_4k", train([l + b"\n" for l in log.split(b"\n") if l][:6000], 4096), True))
    dicts += [(n, d, False) for n, d in raw_dicts(corpus)]
    payloads = dict_payloads(corpus)
    cctx = ref.ZSTD_createCCtx()
    out = {"libzstd": "1.5.5", "generator": "tests/golden/make_golden_dict.py", "dicts": {}, "cases": []}
    for name, d, store in dicts:
        out["dicts"][name] = {"len": len(d), "sha256": hashlib.sha256(d).hexdigest()}
        if store:
            out["dicts"][name]["b64"] = base64.b64encode(d).decode()
        for level in (1, -3):
            cd = ref.ZSTD_createCDict(d, len(d), level)
            assert cd
            for pn, p in payloads:
                cap = ref.ZSTD_compressBound(len(p))
                buf = ctypes.create_string_buffer(cap + 1)
                r = ref.ZSTD_compress_usingCDict(cctx, buf, cap, p, len(p), cd)
                assert r < 2 ** 63
                c = buf.raw[:r]
                e = {"dict": name, "level": level, "payload": pn, "c_len": len(c), "c_sha256": hashlib.sha256(c).hexdigest()[:32]}
                if len(c) <= 120:
                    e["c_hex"] = c.hex()
                out["cases"].append(e)
            ref.ZSTD_freeCDict(cd)
    json.dump(out, open(os.path.join(HERE, "dict_cases.json"), "w"), indent=0)
    print("wrote", len(out["cases"]), "dictionary cases over", len(dicts), "dictionaries")


if __name__ == "__main__":
    main()
