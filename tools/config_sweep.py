#!/usr/bin/env python3
"""Device-resident throughput of the BASELINE.json configurations on one MI355X (diagnostics for DESIGN.md section 4.6;
bench.py measures configuration 2 only).  Every line also checks a sample of the compressed frames against the oracle
and the round trip bit for bit."""
import base64, ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, zstd_amd
from zstd_amd import corpus
from conftest import _Lib

orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")


def run(name, kind, fs, n, seed, level, dictionary=None, steps=3):
    host = corpus.generate(kind, fs, n, seed, threads=16)
    src = torch.from_numpy(host).cuda()
    bound = zstd_amd.CompressBound(fs); stride = (bound + 255) // 256 * 256
    cbuf = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); out = torch.zeros(n * fs, dtype=torch.uint8, device="cuda")
    enc, dec = zstd_amd.DeviceBatch(n), zstd_amd.DeviceBatch(n)
    enc.set_frames_strided(n, src.data_ptr(), fs, None, cbuf.data_ptr(), stride, bound)
    dec.set_frames_strided(n, cbuf.data_ptr(), stride, None, out.data_ptr(), fs, fs)
    bp = zstd_amd.NewBulkProcessor(dictionary, level) if dictionary is not None else None
    cd, dd = (bp.cDict, bp.dDict) if bp else (None, None)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tc = td = 0.0
    for it in range(steps + 1):
        ev[0].record(); enc.compress(level, cd); ev[1].record(); dec.link_src_sizes(enc); dec.decompress(dd); ev[2].record()
        torch.cuda.synchronize()
        if it:
            tc += ev[0].elapsed_time(ev[1]); td += ev[1].elapsed_time(ev[2])
    csz = enc.results(); dsz = dec.results()
    assert all(0 < c < 2 ** 63 for c in csz), "compress error"
    assert all(d == fs for d in dsz) and torch.equal(src, out), "round trip differs"
    ch = cbuf.cpu().numpy()
    ocd = orc.cdict(dictionary, level) if dictionary is not None else None
    for i in range(0, n, max(1, n // 16)):
        f = host[i * fs:(i + 1) * fs].tobytes()
        want = orc.compress_cdict(f, ocd) if ocd else orc.compress(f, level)
        assert ch[i * stride:i * stride + csz[i]].tobytes() == want, "frame %d differs from the oracle" % i
    tot = float(n) * fs
    # roofline of each launch as bench.py states it: algorithmic bytes (input + output of the launch) / time against 8 TB/s of HBM
    alg = tot + float(sum(csz))
    kern = "zsa_k_compress_dict" if dictionary is not None else ("zsa_k_compress" if level < 3 else ("zsa_k_compress_dfast" if level < 5 else "zsa_k_compress_lazy"))
    print("%-46s %7.2f GB/s compress %7.2f GB/s decompress  ratio %.3f  (%d x %d B, level %d%s; sample byte-identical, round trip exact)\n"
          "%-46s roofline (hbm, 8000 GB/s): %s %.1f GB/s = %.2f %%, zsa_k_decompress %.1f GB/s = %.2f %%"
          % (name, tot / (tc / steps) / 1e6, tot / (td / steps) / 1e6, tot / sum(csz), n, fs, level, ", dictionary %d B" % len(dictionary) if dictionary else "",
             "", kern, alg / (tc / steps) / 1e6, alg / (tc / steps) / 1e6 / 80.0, alg / (td / steps) / 1e6, alg / (td / steps) / 1e6 / 80.0), flush=True)
    del src, cbuf, out, enc, dec
    torch.cuda.empty_cache()


g = json.load(open(os.path.join(ROOT, "tests", "golden", "dict_cases.json")))
trained = base64.b64decode(g["dicts"]["trained_json_8k"]["b64"])
if len(sys.argv) > 1 and sys.argv[1] == "lazy":   # the greedy / lazy rows only
    run("   128 KiB log frames, level 5 (greedy, row hash)", "log", 131072, 8192, 42, 5, steps=2)
    run("   16 KiB log frames, level 5 (lazy, hash chain)", "log", 16384, 65536, 45, 5, steps=2)
    run("   128 KiB log frames, level 7 (lazy2, row hash)", "log", 131072, 8192, 42, 7, steps=2)
    sys.exit(0)
run("C1 64 KiB JSON blocks, level 1", "json", 65536, 16384, 1234, 1)
run("C2 128 KiB log frames, level 1 (bench.py)", "log", 131072, 8192, 42, 1)
run("C3 1 KiB JSON payloads + trained dictionary", "json", 1024, 1 << 20, 7, 1, dictionary=trained)
run("C4 256 KiB log frames, level 1", "log", 262144, 8192, 43, 1)
run("C5 1 MiB log chunks, level 3 (ZSTD_dfast)", "log", 1 << 20, 4096, 44, 3, steps=2)
run("   128 KiB log frames, level 3 (ZSTD_dfast)", "log", 131072, 8192, 42, 3)
run("   128 KiB log frames, level 5 (greedy, row hash)", "log", 131072, 8192, 42, 5, steps=2)
run("   16 KiB log frames, level 5 (lazy, hash chain)", "log", 16384, 65536, 45, 5, steps=2)
run("   128 KiB log frames, level 7 (lazy2, row hash)", "log", 131072, 8192, 42, 7, steps=2)
