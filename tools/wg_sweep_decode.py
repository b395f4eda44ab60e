import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, zstd_amd
from zstd_amd import corpus
N, FS = 8192, 131072
host = corpus.generate("log", FS, N, 42, threads=16)
src = torch.from_numpy(host).cuda()
bound = zstd_amd.CompressBound(FS); stride = (bound + 255) // 256 * 256
cbuf = torch.empty(N * stride, dtype=torch.uint8, device="cuda")
out = torch.empty(N * FS, dtype=torch.uint8, device="cuda")
enc = zstd_amd.DeviceBatch(N)
enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound)
enc.compress(1); torch.cuda.synchronize()
for wg in "4 6 8 10 12".split():
    os.environ["ZSA_WG_PER_CU"] = wg
    dec = zstd_amd.DeviceBatch(N)
    dec.set_frames_strided(N, cbuf.data_ptr(), stride, None, out.data_ptr(), FS, FS)
    dec.link_src_sizes(enc); dec.decompress(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dec.decompress(); e1.record(); torch.cuda.synchronize()
    print("wg/CU %s: %.2f ms, %.2f GB/s, launch %s" % (wg, e0.elapsed_time(e1), N * FS / e0.elapsed_time(e1) / 1e6, dec.last_launch()), flush=True)
    dec.close()
