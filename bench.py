#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batch codec (BASELINE.json configs[1]).

Workload per GPU: 1 GiB fixed-seed synthetic log corpus cut into 8192 independent 128 KiB frames (weak scaling:
every rank gets its own 8192 frames of the same corpus family).  A "step" is one pass of the hot path over the
batch: compress all frames, hand the compressed sizes over on the device, decompress what was just produced --
inputs, outputs and all intermediate sizes stay in HBM.  With more than one rank the step also contains the path's
only exchange: an all-gather of the per-frame compressed sizes (taken from the kernel's result words on the device)
from which every rank derives the global output offsets.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N ranks itself (torch.distributed.run,
one process per GPU) before anything touches a GPU, and fails loudly when the box has fewer devices.

Prints ONE JSON line on rank 0: `value` is GB/s of UNCOMPRESSED bytes through the full compress+decompress round
trip, whole job; per-direction rates, ratio, roofline of the dominant kernel, the PCIe-inclusive host-batch rate and
the CPU baseline ride along in the same object.
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak BW 8.0 TB/s (spec)


def shard_plan(n_frames_per_rank, rank, world):
    """Frames [first, first+count) of the corpus that this rank owns (contiguous batch split)."""
    return rank * n_frames_per_rank, n_frames_per_rank


def global_offsets(sizes_per_rank):
    """Exclusive prefix sum over the rank-major concatenation of per-frame compressed sizes: what every rank can
    compute after the single all-gather of sizes (the only exchange step of the path).  Host restatement of
    exchange_sizes(), used by the tests."""
    out, acc = [], 0
    for sizes in sizes_per_rank:
        row = []
        for s in sizes:
            row.append(acc)
            acc += int(s)
        out.append(row)
    return out, acc


def exchange_sizes(sizes, world):
    """The exchange step: `sizes` (int32 tensor, this rank's per-frame compressed sizes, on the device the backend
    works on) -> (gathered [world, n] int32, offsets [world * n] int64 exclusive prefix sum, total).  One collective."""
    import torch
    import torch.distributed as dist
    if world > 1:
        flat_out = torch.empty(world * sizes.numel(), dtype=sizes.dtype, device=sizes.device)
        dist.all_gather_into_tensor(flat_out, sizes)
        gathered = flat_out.view(world, -1)
    else:
        gathered = sizes.view(1, -1)
    flat = gathered.reshape(-1).to(torch.int64)
    incl = torch.cumsum(flat, 0)
    return gathered, incl - flat, incl[-1]


def ranks_agree(offsets, world):
    """True when every rank derived the same offsets (checked with one MIN and one MAX reduction of a digest)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return True
    w = torch.arange(1, offsets.numel() + 1, dtype=torch.int64, device=offsets.device)
    digest = ((offsets * w) % 1000003).sum().reshape(1)
    lo, hi = digest.clone(), digest.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool((lo == hi).item())


def spawn_ranks(n, argv, need_gpus=True):
    """Start n ranks of this script under torch.distributed.run and exit with their status.  Called before any GPU
    call of this process (torch.cuda.device_count() does not initialise the GPU on this image)."""
    if need_gpus:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, but this box has %d GPU(s); not running a smaller job under that name\n" % (n, have))
            sys.exit(3)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd))


def build_id():
    """Digest of the engine's sources: ties a committed PMC summary to the build being timed."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "zstd_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_run(exe, lib, impl, frame_size, frames, seed, threads, level, kind=0):
    out = subprocess.run([exe, lib, impl, str(kind), str(frame_size), str(frames), str(seed), str(threads), str(level)],
                         capture_output=True, text=True, timeout=600, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def cpu_baseline(frame_size, n_frames, seed, level, kind=0, quick=False):
    """Reference C path (oracle/_ref, libzstd 1.5.5 as vendored by the reference) on the host cores of this box,
    one ZSTD_compress/ZSTD_decompress call per frame from a pthread pool over all the frames of the workload
    (Go is not installed).  Threads = every core this process may run on; the 1-thread figure (on a
    quarter of the frames) and, where it differs, the figure at the box's CPU share (OMP_NUM_THREADS) ride along."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cpu_bench")
    lib = os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so")
    kind, impl = "reference", "ref"
    if not (os.path.exists(exe) and os.path.exists(lib)):
        lib = os.path.join(ROOT, "oracle", "liboracle.so")
        kind, impl = "port", "oracle"
        if not (os.path.exists(exe) and os.path.exists(lib)):
            return None
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 256))          # (cpu_bench's thread table)
    try:
        share = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", str(cores)))))
    except ValueError:
        share = cores

    def fold(r):
        tc, td = 1.0 / r["compress_GBps"], 1.0 / r["decompress_GBps"]
        return {"round_trip_GBps": round(1.0 / (tc + td), 4), "compress_GBps": r["compress_GBps"], "decompress_GBps": r["decompress_GBps"]}
    try:
        full = _cpu_run(exe, lib, impl, frame_size, n_frames, seed, cores, level, kind)
        res = {"value": fold(full)["round_trip_GBps"], "unit": "GB/s", "cores": cores, "kind": kind, "cpu_model": cpu_model(),
               "compress_GBps": full["compress_GBps"], "decompress_GBps": full["decompress_GBps"], "ratio": full["ratio"],
               "libzstd": full["libzstd"],
               "sample": "all %d frames (%d KiB each, same corpus/seed), %d pthreads = every core in sched_getaffinity, one call per frame"
                         % (n_frames, frame_size // 1024, cores)}
        if quick:
            return res
        one = _cpu_run(exe, lib, impl, frame_size, max(1, n_frames // 4), seed, 1, level, kind)
        res["one_thread"] = dict(fold(one), frames=max(1, n_frames // 4))
        if share != cores:
            sh = _cpu_run(exe, lib, impl, frame_size, n_frames, seed, share, level, kind)
            res["at_cpu_share"] = dict(fold(sh), threads=share, note="OMP_NUM_THREADS: the CPU share the harness gives one GPU's job")
        return res
    except Exception as e:   # the baseline is a reported number, never a reason to lose the GPU line
        return {"value": None, "unit": "GB/s", "cores": cores, "kind": kind, "sample": "failed: %r" % (e,)}


def end_to_end(zstd_amd, host, frame_size, n, level):
    """PCIe-inclusive rate: the host-buffer entry points (zsa_compress_batch / zsa_decompress_batch) on a bounded
    sample of the same frames -- once with pageable host memory in and out (what a caller's ordinary buffers are), once
    with buffers from zsa_host_alloc (page-locked; the GPU reads and writes them in place).  Reported beside `value`,
    never instead of it."""
    import numpy as np
    from zstd_amd._lib import lib, c_size_p, c_void_pp
    bound = zstd_amd.CompressBound(frame_size)
    P, S = ctypes.c_void_p * n, ctypes.c_size_t * n
    cast = lambda a, t: ctypes.cast(a, t)

    def run(src_addr, c_addr, back_addr):
        srcp = P(*[src_addr + i * frame_size for i in range(n)])
        srcn = S(*([frame_size] * n))
        dstp = P(*[c_addr + i * bound for i in range(n)])
        dstc = S(*([bound] * n))
        outn = S()
        best_c = best_d = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = lib.zsa_compress_batch(cast(srcp, c_void_pp), cast(srcn, c_size_p), cast(dstp, c_void_pp), cast(dstc, c_size_p), cast(outn, c_size_p), n, level, None)
            best_c = min(best_c, time.perf_counter() - t0)
            assert rc == 0
        csz = [outn[i] for i in range(n)]
        assert all(c < 2 ** 63 for c in csz)
        backp = P(*[back_addr + i * frame_size for i in range(n)])
        cszA = S(*csz)
        outd = S()
        for _ in range(3):
            t0 = time.perf_counter()
            rc = lib.zsa_decompress_batch(cast(dstp, c_void_pp), cast(cszA, c_size_p), cast(backp, c_void_pp), cast(srcn, c_size_p), cast(outd, c_size_p), n, None)
            best_d = min(best_d, time.perf_counter() - t0)
            assert rc == 0
        assert all(outd[i] == frame_size for i in range(n))
        return best_c, best_d

    b = float(n) * frame_size
    fold = lambda tc, td: {"compress_GBps": round(b / tc / 1e9, 3), "decompress_GBps": round(b / td / 1e9, 3), "round_trip_GBps": round(b / (tc + td) / 1e9, 3)}
    cbuf = np.empty(n * bound, dtype=np.uint8)
    back = np.empty(n * frame_size, dtype=np.uint8)
    best_c, best_d = run(host.ctypes.data, cbuf.ctypes.data, back.ctypes.data)
    assert bool((back == host[:n * frame_size]).all()), "host-batch round trip is not bit-exact"
    res = fold(best_c, best_d)
    res["sample"] = "%d of the frames through zsa_compress_batch / zsa_decompress_batch, pageable host buffers, best of 3" % n
    del cbuf, back
    # the same sample from / into page-locked buffers of the library (no staging copies)
    try:
        lib.zsa_thread_release()
        hs, hc, hb = lib.zsa_host_alloc(n * frame_size), lib.zsa_host_alloc(n * bound), lib.zsa_host_alloc(n * frame_size)
        assert hs and hc and hb, "zsa_host_alloc failed"
        ctypes.memmove(hs, host.ctypes.data, n * frame_size)
        pc, pd = run(hs, hc, hb)
        got = np.ctypeslib.as_array(ctypes.cast(hb, ctypes.POINTER(ctypes.c_uint8)), shape=(n * frame_size,))
        assert bool((got == host[:n * frame_size]).all()), "pinned host-batch round trip is not bit-exact"
        del got
        res["pinned_buffers"] = fold(pc, pd)
        res["pinned_buffers"]["sample"] = "the same frames, payloads and destinations in memory from zsa_host_alloc"
        for h in (hs, hc, hb):
            lib.zsa_host_free(h)
    except Exception as e:
        res["pinned_buffers"] = {"failed": repr(e)}
    # single-call latency (zstd_bullk_test.go:252 quotes 1.5 us/op for the CPU library on a 92-byte payload)
    small = bytes(host[:1024])
    zstd_amd.CompressLevel(None, small, level)
    t0 = time.perf_counter()
    for _ in range(200):
        zstd_amd.CompressLevel(None, small, level)
    res["single_call_compress_us_1KiB"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    lib.zsa_thread_release()
    return res


def _oracle():
    """The CPU oracle as the CHECKER of sampled frames (never the thing measured); None when it is not built."""
    opath = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(opath):
        return None
    orc = ctypes.CDLL(opath)
    orc.zo_compress.restype = ctypes.c_size_t
    orc.zo_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    orc.zo_createCDict.restype = ctypes.c_void_p
    orc.zo_createCDict.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    orc.zo_compress_usingCDict.restype = ctypes.c_size_t
    orc.zo_compress_usingCDict.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    orc.zo_freeCDict.argtypes = [ctypes.c_void_p]
    return orc


def kernel_of(level, dictionary):
    """the compress kernel a level resolves to at these frame sizes (clevels.h: ZSTD_fast for levels <= 2, ZSTD_dfast for 3-4, greedy/lazy above)"""
    if dictionary is not None:   # (a CDict's strategy comes from the level and the dictionary's size: ZSTD_fast below level 3)
        return "zsa_k_compress_dict" if 0 != level < 3 else "zsa_k_compress_dict_hi"
    return "zsa_k_compress" if level < 3 else "zsa_k_compress_dfast" if level < 5 else "zsa_k_compress_lazy"


def measure_config(zstd_amd, corpus, torch, dev, stream, kind, fs, n, seed, level, dictionary=None, steps=2, decode_only=False, samples=8):
    """One BASELINE configuration, device resident: compress all frames, hand the sizes over on the device, decompress;
    HIP events on the launch stream.  Every entry is checked: all frames decode to the input bit for bit (computed, on the
    device) and a sample of the compressed frames equals the oracle's bytes."""
    host = corpus.generate(kind, fs, n, seed, threads=16)
    src = torch.from_numpy(host).to(dev)
    bound = zstd_amd.CompressBound(fs)
    stride = (bound + 255) // 256 * 256
    cbuf = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    out = torch.zeros(n * fs, dtype=torch.uint8, device=dev)
    enc, dec = zstd_amd.DeviceBatch(n), zstd_amd.DeviceBatch(n)
    enc.set_frames_strided(n, src.data_ptr(), fs, None, cbuf.data_ptr(), stride, bound, stream)
    dec.set_frames_strided(n, cbuf.data_ptr(), stride, None, out.data_ptr(), fs, fs, stream)
    bp = zstd_amd.NewBulkProcessor(dictionary, level) if dictionary is not None else None
    cd, dd = (bp.cDict, bp.dDict) if bp else (None, None)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tc = td = 0.0
    for it in range(steps + 1):          # the first pass is the warm-up (and sizes the workspace)
        ev[0].record(); enc.compress(level, cd, stream); ev[1].record()
        dec.link_src_sizes(enc, stream); dec.decompress(dd, stream); ev[2].record()
        torch.cuda.synchronize()
        if it:
            tc += ev[0].elapsed_time(ev[1]); td += ev[1].elapsed_time(ev[2])
    csz, dsz = enc.results(stream), dec.results(stream)
    ok_c = all(0 < c < 2 ** 63 for c in csz)
    exact = bool(ok_c and all(d == fs for d in dsz) and torch.equal(src, out))
    identical = None
    orc = _oracle()
    if orc is not None and ok_c:
        ch = cbuf.cpu().numpy()
        ocd = orc.zo_createCDict(dictionary, len(dictionary), level) if dictionary is not None else None
        identical = True
        for i in range(0, n, max(1, n // samples)):
            f = host[i * fs:(i + 1) * fs].tobytes()
            buf = ctypes.create_string_buffer(bound + 1)
            m = orc.zo_compress_usingCDict(buf, bound, f, fs, ocd) if ocd else orc.zo_compress(buf, bound, f, fs, level)
            identical = identical and (buf.raw[:m] == ch[i * stride:i * stride + csz[i]].tobytes())
        if ocd:
            orc.zo_freeCDict(ocd)
        del ch
    tot, csum = float(n) * fs, float(sum(csz)) if ok_c else float("nan")
    alg = tot + csum                       # srcSize + cSize per frame, either direction
    kc, kd = tc / steps / 1e3, td / steps / 1e3
    r = {"frames": n, "frame_bytes": fs, "corpus": kind, "seed": seed, "level": level,
         "decompress_GBps": round(tot / kd / 1e9, 3), "ratio": round(tot / csum, 4),
         "roofline_frac_decompress": round(alg / kd / 1e9 / HBM_PEAK_GBPS, 6), "kernel_decompress": "zsa_k_decompress",
         "bit_exact_round_trip": exact, "byte_identical_to_oracle_sample": identical}
    if dictionary is not None:
        r["dictionary_bytes"] = len(dictionary)
    if decode_only:
        r["note"] = "decode-only configuration: frames produced on the device (sample byte-identical to the CPU path), only the decode is the figure"
    else:
        r.update({"compress_GBps": round(tot / kc / 1e9, 3), "roofline_frac_compress": round(alg / kc / 1e9 / HBM_PEAK_GBPS, 6),
                  "kernel_compress": kernel_of(level, dictionary)})
    del src, cbuf, out, enc, dec
    torch.cuda.empty_cache()
    return r


def baseline_configs(zstd_amd, corpus, torch, dev, stream, with_cpu):
    """The other BASELINE.json configurations and the JSON-like corpus at the headline shape, one launch size each (N = 1 only).
    `value` stays configuration 2; these ride along so that every configuration has a driver-visible figure."""
    import base64
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "dict_cases.json")))
    trained = base64.b64decode(g["dicts"]["trained_json_8k"]["b64"])
    out = {}
    t0 = time.perf_counter()
    plan = [("C1_json_64KiB_level1", dict(kind="json", fs=65536, n=16384, seed=1234, level=1)),
            ("C2_json_128KiB_level1", dict(kind="json", fs=131072, n=8192, seed=1234, level=1)),
            ("C3_bulk_1KiB_trained_dictionary_level1", dict(kind="json", fs=1024, n=1 << 20, seed=7, level=1, dictionary=trained)),
            ("bulk_1KiB_trained_dictionary_level5_go_default", dict(kind="json", fs=1024, n=1 << 18, seed=7, level=5, dictionary=trained, steps=1)),
            ("C4_decode_256KiB_level1", dict(kind="log", fs=262144, n=8192, seed=43, level=1, decode_only=True)),
            ("C5_1MiB_chunks_level3_dfast", dict(kind="log", fs=1 << 20, n=4096, seed=44, level=3)),
            ("log_128KiB_level5_go_default", dict(kind="log", fs=131072, n=8192, seed=42, level=5, steps=1))]
    for name, kw in plan:
        try:
            out[name] = measure_config(zstd_amd, corpus, torch, dev, stream, **kw)
        except Exception as e:           # a reported extra, never a reason to lose the line
            out[name] = {"failed": repr(e)}
    if with_cpu:   # configuration 1 IS the CPU configuration: the reference C on this box's cores, same corpus and framing
        try:
            out["C1_json_64KiB_level1"]["cpu_baseline"] = cpu_baseline(65536, 16384, 1234, 1, kind=1, quick=True)
        except Exception as e:
            out["C1_json_64KiB_level1"]["cpu_baseline"] = {"failed": repr(e)}
    out["seconds"] = round(time.perf_counter() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8192, help="frames per GPU")
    ap.add_argument("--frame-size", type=int, default=128 * 1024)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the block with the other BASELINE.json configurations")
    ap.add_argument("--exchange-dry-run", action="store_true",
                    help="CPU-only rehearsal of the rank plumbing (spawn, shard plan, all-gather of sizes, offsets) over gloo with "
                         "made-up sizes; nothing is compressed and no throughput is printed")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args.gpus, sys.argv[1:], need_gpus=not args.exchange_dry_run)      # does not return
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch one rank per GPU (or let bench.py start them: unset WORLD_SIZE)\n" % (args.gpus, world))
        sys.exit(3)

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.exchange_dry_run:
        if world > 1:
            dist.init_process_group("gloo")
        first, count = shard_plan(args.frames, rank, world)
        g = torch.Generator().manual_seed(args.seed)
        all_sizes = torch.randint(1000, 40000, (world * args.frames,), generator=g, dtype=torch.int32)   # same on every rank
        mine = all_sizes[first:first + count].clone()
        gathered, offsets, total = exchange_sizes(mine, world)
        ok = bool(torch.equal(gathered.reshape(-1), all_sizes)) and ranks_agree(offsets, world)
        want, want_total = global_offsets([all_sizes[r * args.frames:(r + 1) * args.frames].tolist() for r in range(world)])
        ok = ok and offsets.tolist() == [o for row in want for o in row] and int(total) == want_total
        if world > 1:
            flag = torch.tensor([1 if ok else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"exchange_dry_run": True, "n_ranks": world, "frames_per_rank": args.frames, "offsets_agree": ok,
                              "total_bytes": int(total)}), flush=True)
        sys.exit(0 if ok else 1)

    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU path)"
    if torch.cuda.device_count() <= local_rank:
        sys.stderr.write("bench.py: rank %d has no GPU (device_count %d)\n" % (local_rank, torch.cuda.device_count()))
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()

    import zstd_amd
    from zstd_amd import corpus

    FS, N = args.frame_size, args.frames
    first, _ = shard_plan(N, rank, world)
    host = corpus.generate("log", FS, N, args.seed, first_frame=first, threads=16)
    src = torch.from_numpy(host).to(dev)
    bound = zstd_amd.CompressBound(FS)
    stride = (bound + 255) // 256 * 256
    cbuf = torch.empty(N * stride, dtype=torch.uint8, device=dev)
    out = torch.empty(N * FS, dtype=torch.uint8, device=dev)
    sizes_dev = torch.zeros(N, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    enc = zstd_amd.DeviceBatch(N)
    dec = zstd_amd.DeviceBatch(N)
    enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound, stream)
    dec.set_frames_strided(N, cbuf.data_ptr(), stride, None, out.data_ptr(), FS, FS, stream)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    last = {}

    def step():
        """one pass: compress -> (device-side hand-over of sizes) -> decompress -> exchange of sizes; HIP events on the
        launch stream (torch's current stream, which is also the stream the collective is enqueued on)"""
        ev[0].record()
        enc.compress(args.level, None, stream)
        ev[1].record()
        dec.link_src_sizes(enc, stream)
        dec.decompress(None, stream)
        ev[2].record()
        enc.export_sizes_i32(sizes_dev.data_ptr(), stream)
        last["gathered"], last["offsets"], last["total"] = exchange_sizes(sizes_dev, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if not args.warmup:
        step()
    torch.cuda.synchronize()

    # correctness gate (outside the timed region): every frame compressed, decoded, and bit-exact
    csizes = enc.results(stream)
    dsizes = dec.results(stream)
    assert all(c < 2 ** 63 for c in csizes), "compress errors: %s" % [zstd_amd.ErrorCode(c - 2 ** 64) for c in csizes if c >= 2 ** 63][:3]
    assert all(d == FS for d in dsizes), "decompress errors"
    assert bool(torch.equal(out, src)), "round trip is not bit-exact"
    assert last["gathered"][rank].tolist() == csizes, "the exchanged sizes are not the kernel's result words"
    assert ranks_agree(last["offsets"], world), "ranks disagree on the global output offsets"
    identical = None
    if rank == 0:
        opath = os.path.join(ROOT, "oracle", "liboracle.so")
        if os.path.exists(opath):
            orc = ctypes.CDLL(opath)
            orc.zo_compress.restype = ctypes.c_size_t
            orc.zo_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
            ch = cbuf.cpu().numpy()
            identical = True
            for i in range(0, N, max(1, N // 32)):
                buf = ctypes.create_string_buffer(bound + 1)
                n = orc.zo_compress(buf, bound, host[i * FS:(i + 1) * FS].tobytes(), FS, args.level)
                identical &= (buf.raw[:n] == ch[i * stride:i * stride + csizes[i]].tobytes())
            del ch
    total_c = sum(csizes)

    # timed region
    t_c = t_d = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ev[2].synchronize()
        t_c += ev[0].elapsed_time(ev[1])
        t_d += ev[1].elapsed_time(ev[2])
    barrier()
    elapsed = time.perf_counter() - t0
    total_c_all = float(last["total"].item())
    # the flags of the line are computed on what the TIMED steps left behind (every rank; folded below)
    exact_after = bool(all(d == FS for d in dec.results(stream)) and torch.equal(out, src))
    agree_after = bool(ranks_agree(last["offsets"], world) and last["gathered"][rank].tolist() == enc.results(stream))
    if world > 1:
        t = torch.tensor([elapsed, t_c, t_d], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_c, t_d = [float(x) for x in t.tolist()]
        f = torch.tensor([1 if exact_after else 0, 1 if agree_after else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        exact_after, agree_after = bool(f[0].item()), bool(f[1].item())

    if rank == 0:
        K = args.steps
        bytes_rank = float(N) * FS
        bytes_all = bytes_rank * world
        ms_step = elapsed / K * 1e3
        kc_ms, kd_ms = t_c / K, t_d / K          # average launch durations (HIP events on the launch stream)
        algo_c = bytes_rank + float(total_c)      # compress = srcSize + cSize per frame
        algo_d = float(total_c) + bytes_rank      #              decompress = cSize + dstSize
        wg, lds = enc.last_launch()
        # the kernel family of the level at this frame size (clevels.h: ZSTD_fast for levels <= 2, ZSTD_dfast for 3-4, greedy/lazy above)
        kernel_c = kernel_of(args.level, None)
        traffic_c, traffic_d, traffic_src = committed_traffic(N, FS, args.level, kernel_c)
        res = {
            "metric": "GB/s compress & decompress round trip (whole job, uncompressed bytes), %d KiB frames, level %d; ratio" % (FS // 1024, args.level),
            "value": round(bytes_all / elapsed * K / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d KiB independent level-%d frames per GPU, fixed-seed log-like corpus (seed %d), compress then decompress, device resident"
                                   % (N, FS // 1024, args.level, args.seed),
                       "frames_per_gpu": N, "frame_bytes": FS, "level": args.level, "parallelism": "batch split x%d" % world,
                       "exchange": "all-gather of per-frame compressed sizes (int32 x %d per rank) every step" % N if world > 1 else "none (one rank)"},
            "compress_GBps": round(bytes_all / (kc_ms / 1e3) / 1e9, 3),
            "decompress_GBps": round(bytes_all / (kd_ms / 1e3) / 1e9, 3),
            "ratio": round(bytes_all / total_c_all, 4),
            "bit_exact_round_trip": exact_after,
            "byte_identical_to_oracle_sample": identical,
            "global_offsets_agree_on_all_ranks": agree_after,
            "build_id": build_id(),
            "roofline": {"bound": "hbm", "kernel": kernel_c, "achieved": round(algo_c / (kc_ms / 1e3) / 1e9, 3),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(algo_c / (kc_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 6),
                         "traffic": traffic_c and traffic_c["raw"], "traffic_x2_upper_bound": traffic_c and traffic_c["x2_upper_bound"], "traffic_source": traffic_src, "avg_launch_ms": round(kc_ms, 3), "algorithmic_bytes_per_launch": int(algo_c),
                         "workgroups": wg, "lds_bytes_per_workgroup": lds},
            "roofline_decompress": {"bound": "hbm", "kernel": "zsa_k_decompress", "achieved": round(algo_d / (kd_ms / 1e3) / 1e9, 3),
                                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(algo_d / (kd_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 6),
                                    "traffic": traffic_d and traffic_d["raw"], "traffic_x2_upper_bound": traffic_d and traffic_d["x2_upper_bound"], "avg_launch_ms": round(kd_ms, 3), "algorithmic_bytes_per_launch": int(algo_d)},
        }
        if not args.no_end_to_end:
            try:
                res["end_to_end"] = end_to_end(zstd_amd, host, FS, min(N, 4096), args.level)
            except Exception as e:   # a reported extra, never a reason to lose the line
                res["end_to_end"] = {"failed": repr(e)}
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(FS, N, args.seed, args.level)
        if world == 1 and not args.no_configs:
            del src, cbuf, out
            torch.cuda.empty_cache()
            res["configs"] = baseline_configs(zstd_amd, corpus, torch, dev, stream, not args.no_cpu_baseline)
        elif not args.no_configs:
            res["configs"] = "reported by the --gpus 1 run only (one launch size per configuration)"
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# HBM bytes per launch need PMC passes of their own (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot ride along with a
# timed run); tools/collect_profiles.sh runs them on this very command and tools/summarize_profiles.py commits the
# per-launch averages, with the guide's gfx950 correction (FETCH_SIZE counts 128-B requests as 64 B: x2), to this file,
# stamped with build_id().  A summary taken on another build, workload or kernel is not quoted: traffic is null then.
PMC_SUMMARY = "profiles/r03_pmc.json"


def committed_traffic(n_frames, frame_bytes, level, kernel_c):
    path = os.path.join(ROOT, PMC_SUMMARY)
    if not os.path.exists(path):
        return None, None, "no PMC summary committed (%s)" % PMC_SUMMARY
    pmc = json.load(open(path))
    if pmc.get("build_id") != build_id():
        return None, None, "%s was collected on build %s, this is build %s: re-run tools/collect_profiles.sh" % (PMC_SUMMARY, pmc.get("build_id"), build_id())
    w = pmc.get("workload", {})
    if (w.get("frames"), w.get("frame_bytes"), w.get("level")) != (n_frames, frame_bytes, level):
        return None, None, "%s was collected on another workload" % PMC_SUMMARY
    def get(k):
        if k not in pmc or "fabric_traffic_bytes_raw" not in pmc[k]:
            return None
        return {"raw": int(pmc[k]["fabric_traffic_bytes_raw"]), "x2_upper_bound": int(pmc[k]["fabric_traffic_bytes_x2_upper_bound"])}
    return get(kernel_c), get("zsa_k_decompress"), (
        PMC_SUMMARY + ": separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command and build, per launch, KiB -> bytes. "
        "The counters are L2-miss (fabric) requests, Infinity-Cache hits included, so this bounds HBM traffic from above. `traffic` is the "
        "raw sum; FETCH_SIZE under-counts wide coalesced reads by 2 on gfx950 (guide), scattered cell reads not, so "
        "`traffic_x2_upper_bound` doubles the read part as the other bound")


if __name__ == "__main__":
    main()
