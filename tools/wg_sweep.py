#!/usr/bin/env python3
"""Compress throughput of one configuration against the resident workgroups per CU (ZSA_WG_PER_CU is read per launch).
usage: wg_sweep.py level [frames [frame_bytes [kind]]] -- wg values from the environment WGS="1 2 4 8 16" """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zstd_amd
from zstd_amd import corpus
LEVEL = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
FS = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
KIND = sys.argv[4] if len(sys.argv) > 4 else "log"
host = corpus.generate(KIND, FS, N, 42, threads=16)
src = torch.from_numpy(host).cuda()
bound = zstd_amd.CompressBound(FS); stride = (bound + 255) // 256 * 256
cbuf = torch.empty(N * stride, dtype=torch.uint8, device="cuda")
for wg in os.environ.get("WGS", "1 2 4 8 16").split():
    os.environ["ZSA_WG_PER_CU"] = wg
    enc = zstd_amd.DeviceBatch(N)
    enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound)
    enc.compress(LEVEL); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); enc.compress(LEVEL); e1.record(); torch.cuda.synchronize()
    print("wg/CU %s: %.2f ms, %.2f GB/s, launch %s" % (wg, e0.elapsed_time(e1), N * FS / e0.elapsed_time(e1) / 1e6, enc.last_launch()), flush=True)
    enc.close()
