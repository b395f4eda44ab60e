"""The entry point that allocates nothing (include/zstd_amd.h section 4): zsa_decompress_frames_noalloc, whose workspace
is a device array inside the library, against zsa_decompress_enqueue (batch-owned workspace) -- same kernel, so the
outputs and the result words must be identical.  Two calls in a row, the second with a smaller input, because the array
keeps the previous call's contents.  Collected last (file name) so that a failure here hides no other test."""
import ctypes

import pytest

pytestmark = pytest.mark.gpu


def _run_both(z, torch, frames_c, sizes, ddict=None):
    from zstd_amd._lib import lib
    n = len(frames_c)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    coff, doff, c_total, d_total = [], [], 0, 0
    for c, s in zip(frames_c, sizes):
        coff.append(c_total); c_total += (len(c) + 63) // 64 * 64 + 64
        doff.append(d_total); d_total += (max(s, 1) + 63) // 64 * 64
    import numpy as np
    hostc = np.zeros(c_total, dtype=np.uint8)
    for c, o in zip(frames_c, coff):
        hostc[o:o + len(c)] = np.frombuffer(c, dtype=np.uint8)
    cbuf = torch.from_numpy(hostc).to(dev)
    outs = []
    # (a) the existing entry point
    out_a = torch.zeros(d_total, dtype=torch.uint8, device=dev)
    b = z.DeviceBatch(n)
    b.set_frames(cbuf.data_ptr(), coff, [len(c) for c in frames_c], out_a.data_ptr(), doff, sizes, stream)
    b.decompress(ddict, stream)
    res_a = b.results(stream)
    b.close()
    # (b) the allocation-free one: descriptors and result words are caller-owned device memory
    out_b = torch.zeros(d_total, dtype=torch.uint8, device=dev)
    desc = torch.tensor([[cbuf.data_ptr() + coff[i], out_b.data_ptr() + doff[i], len(frames_c[i]), sizes[i]] for i in range(n)],
                        dtype=torch.int64, device=dev)
    res = torch.full((n,), -1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rc = lib.zsa_decompress_frames_noalloc(desc.data_ptr(), res.data_ptr(), n, ddict, stream)
    assert rc == 0, z.ErrorCode(rc - 2 ** 64)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0, "the call changed the amount of free device memory"
    res_b = [int(x) & (2 ** 64 - 1) for x in res.cpu().tolist()]
    assert res_b == res_a
    assert bool(torch.equal(out_a, out_b))
    return res_a, out_a.cpu().numpy(), doff


def test_no_alloc_entry_point_matches_batch_entry_point(oracle):
    import torch
    assert torch.cuda.is_available()
    import zstd_amd as z
    from zstd_amd import corpus
    # first call: 300 frames (more than the 128 workgroups the array serves), mixed sizes, one corrupted, one too small a dst
    frames = corpus.frames("log", 131072, 40, 42) + corpus.frames("json", 65536, 60, 1234) + corpus.frames("json", 1024, 196, 7) \
        + corpus.frames("log", 262144, 4, 43)
    comp = [oracle.compress(f, 1) for f in frames]
    sizes = [len(f) for f in frames]
    comp[5] = comp[5][:200] + bytes([comp[5][200] ^ 0x5A]) + comp[5][201:]
    sizes[7] -= 1
    res, out, doff = _run_both(z, torch, comp, sizes)
    for i, f in enumerate(frames):
        if i in (5, 7):
            assert res[i] >= 2 ** 63
        else:
            assert res[i] == len(f) and out[doff[i]:doff[i] + len(f)].tobytes() == f
    # second call, smaller: the array still holds the first call's literals, tables and counter
    frames2 = corpus.frames("log", 20000, 9, 5) + [b"", b"a"]
    comp2 = [oracle.compress(f, 1) for f in frames2]
    res2, out2, doff2 = _run_both(z, torch, comp2, [max(len(f), 1) for f in frames2])
    for i, f in enumerate(frames2):
        assert res2[i] == len(f) and out2[doff2[i]:doff2[i] + len(f)].tobytes() == f
    # beyond the stated maximum: a clear error, nothing launched
    from zstd_amd._lib import lib
    rc = lib.zsa_decompress_frames_noalloc(1, 1, (1 << 20) + 1, None, None)
    assert z.api.lib.zsa_isError(rc) and str(z.ErrorCode(rc - 2 ** 64)) == "Parameter is out of bound"


def test_no_alloc_entry_point_with_a_dictionary(oracle):
    """the same comparison for frames that name a dictionary (device digest of zsa_createDDict)"""
    import base64, json, os
    import torch
    import zstd_amd as z
    from zstd_amd import corpus
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dict_cases.json")))
    d = base64.b64decode(g["dicts"]["trained_json_8k"]["b64"])
    frames = corpus.frames("json", 1024, 150, 7) + corpus.frames("json", 9000, 6, 8)
    ocd = oracle.cdict(d, 1)
    comp = [oracle.compress_cdict(f, ocd) for f in frames]
    oracle.free_cdict(ocd)
    bp = z.NewBulkProcessor(d, 1)
    res, out, doff = _run_both(z, torch, comp, [len(f) for f in frames], bp.dDict)
    for i, f in enumerate(frames):
        assert res[i] == len(f) and out[doff[i]:doff[i] + len(f)].tobytes() == f
    # without the dictionary the frames are refused ("Dictionary mismatch"), through either entry point alike
    res2, _, _ = _run_both(z, torch, comp[:3], [len(f) for f in frames[:3]], None)
    assert all(r >= 2 ** 63 for r in res2)


_FIRST_CALL = r"""
import ctypes, sys
sys.path.insert(0, %(root)r)
import torch
from zstd_amd._lib import lib          # (loads the library; nothing is launched or allocated by it yet)
frame = bytes.fromhex(%(frame)r)
want = bytes.fromhex(%(want)r)
dev = torch.device("cuda", 0)
c = torch.frombuffer(bytearray(frame), dtype=torch.uint8).to(dev)
out = torch.zeros(len(want), dtype=torch.uint8, device=dev)
desc = torch.tensor([[c.data_ptr(), out.data_ptr(), len(frame), len(want)]], dtype=torch.int64, device=dev)
res = torch.full((1,), -1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
assert lib.zsa_noalloc_prepare() == 0   # the documented warm-up: resolves the workspace address, launches nothing
rc = lib.zsa_decompress_frames_noalloc(desc.data_ptr(), res.data_ptr(), 1, None, None)
torch.cuda.synchronize()
assert rc == 0, rc
assert int(res.cpu()[0]) == len(want) and bytes(out.cpu().numpy()) == want
# (the HIP runtime itself sets up the kernel's private-memory arena at a code object's first launch; from then on nothing moves)
out.zero_(); res.fill_(-1); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
rc = lib.zsa_decompress_frames_noalloc(desc.data_ptr(), res.data_ptr(), 1, None, None)
torch.cuda.synchronize()
assert rc == 0 and torch.cuda.mem_get_info()[0] == free0
assert int(res.cpu()[0]) == len(want) and bytes(out.cpu().numpy()) == want
print("first-call ok")
"""


def test_no_alloc_entry_point_as_the_first_call_of_a_process(oracle):
    """No zsa_batch, no dictionary and no other launch of the library before it: a fresh process whose first use of the
    library is zsa_noalloc_prepare + the allocation-free call."""
    import os, subprocess, sys
    from zstd_amd import corpus
    want = corpus.frames("log", 50000, 1, 3)[0]
    frame = oracle.compress(want, 1)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FIRST_CALL % {"root": root, "frame": frame.hex(), "want": want.hex()}],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "first-call ok" in r.stdout, r.stdout + r.stderr
