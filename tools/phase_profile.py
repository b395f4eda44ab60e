#!/usr/bin/env python3
"""Per-phase cycle breakdown of the two kernels on the bench workload (diagnostics; uses zsa_batch_profile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zstd_amd
from zstd_amd import corpus
# usage: phase_profile.py [frames [frame_bytes [kind [level [dict]]]]]   (dict: name of a fixture dictionary in tests/golden/dict_cases.json)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
FS = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
KIND = sys.argv[3] if len(sys.argv) > 3 else "log"
LEVEL = int(sys.argv[4]) if len(sys.argv) > 4 else 1
CD = DD = None
if len(sys.argv) > 5:
    import base64, json
    g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dict_cases.json")))
    BP = zstd_amd.NewBulkProcessor(base64.b64decode(g["dicts"][sys.argv[5]]["b64"]), LEVEL)
    CD, DD = BP.cDict, BP.dDict
host = corpus.generate(KIND, FS, N, 42, threads=16)
src = torch.from_numpy(host).cuda()
bound = zstd_amd.CompressBound(FS); stride = (bound + 255) // 256 * 256
cbuf = torch.empty(N * stride, dtype=torch.uint8, device="cuda"); out = torch.empty(N * FS, dtype=torch.uint8, device="cuda")
enc, dec = zstd_amd.DeviceBatch(N), zstd_amd.DeviceBatch(N)
enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound)
dec.set_frames_strided(N, cbuf.data_ptr(), stride, None, out.data_ptr(), FS, FS)
enc.compress(LEVEL, CD); dec.link_src_sizes(enc); dec.decompress(DD); torch.cuda.synchronize()
ENC = ["init/hdr", "parse (rest: post-match, rep2, ring)", "gather lits", "lit stats+huf build", "huf encode", "seq hist+tables", "seq encode", "finalize", "parse: window eval", "parse: winner+table", "parse: extension+store", "fetch work"]
DEC = ["blk hdr/other", "huf table", "huf streams", "seq tables", "seq decode", "exec literals", "exec matches", "last lits", "", "", "", "fetch work"]
for name, b, fn, labels in (("compress", enc, lambda: enc.compress(LEVEL, CD), ENC), ("decompress", dec, lambda: dec.decompress(DD), DEC)):
    b.profile(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    p = b.profile(False, read=True)
    tot = sum(p) or 1
    print("%s: %.2f ms, %.2f GB/s, launch %s, %.0f kcycles/frame (lane-0 busy+wait)" % (name, e0.elapsed_time(e1), N * FS / e0.elapsed_time(e1) / 1e6, b.last_launch(), tot / N / 1e3))
    for l, v in zip(labels, p):
        if v: print("   %-22s %6.2f%%  %9.0f cycles/frame" % (l, 100.0 * v / tot, v / N))
