#!/usr/bin/env python3
"""bench.py -- headline benchmark of the level-1 batch codec (BASELINE.json configs[1]).

Workload per GPU: 1 GiB fixed-seed synthetic log corpus cut into 8192 independent 128 KiB frames (weak scaling:
every rank gets its own 8192 frames of the same corpus family).  A "step" is one pass of the hot path over the
batch: compress all frames at level 1, then decompress what was just produced -- inputs, outputs and all
intermediate sizes stay in HBM (the compressed sizes are handed from the compress kernel to the decompress
launch on the device).

Prints ONE JSON line on rank 0 (see the contract in the task description): `value` is GB/s of UNCOMPRESSED
bytes through the full compress+decompress round trip, whole job; per-direction rates, ratio, roofline of the
dominant kernel and the CPU baseline ride along in the same object.
"""
import argparse
import ctypes
import json
import os
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
    compute after the single all-gather of sizes (the only exchange step of the path)."""
    out, acc = [], 0
    for sizes in sizes_per_rank:
        row = []
        for s in sizes:
            row.append(acc)
            acc += int(s)
        out.append(row)
    return out, acc


def cpu_baseline(frame_size, n_frames, seed, level):
    """Reference C path (oracle/_ref, libzstd 1.5.5 as vendored by the reference) on the host cores of this box,
    one ZSTD_compress/ZSTD_decompress call per frame from a pthread pool (Go is not installed)."""
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
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16")), 64))
    sample = min(n_frames, 4096)
    try:
        out = subprocess.run([exe, lib, impl, "0", str(frame_size), str(sample), str(seed), str(cores), str(level)],
                             capture_output=True, text=True, timeout=240, check=True).stdout
        r = json.loads(out.strip().splitlines()[-1])
    except Exception as e:   # the baseline is a reported number, never a reason to lose the GPU line
        return {"value": None, "unit": "GB/s", "cores": cores, "kind": kind, "sample": "failed: %r" % (e,)}
    tc, td = 1.0 / r["compress_GBps"], 1.0 / r["decompress_GBps"]
    return {"value": round(1.0 / (tc + td), 4), "unit": "GB/s", "cores": cores, "kind": kind,
            "compress_GBps": r["compress_GBps"], "decompress_GBps": r["decompress_GBps"], "ratio": r["ratio"],
            "libzstd": r["libzstd"],
            "sample": "%d of the %d frames (%d KiB each, same corpus/seed), %d pthreads, one call per frame"
                      % (sample, n_frames, frame_size // 1024, cores)}


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
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

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
    stream = torch.cuda.current_stream().cuda_stream

    enc = zstd_amd.DeviceBatch(N)
    dec = zstd_amd.DeviceBatch(N)
    enc.set_frames_strided(N, src.data_ptr(), FS, None, cbuf.data_ptr(), stride, bound, stream)
    dec.set_frames_strided(N, cbuf.data_ptr(), stride, None, out.data_ptr(), FS, FS, stream)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    sizes_dev = None

    def step(timed):
        """one pass: compress -> (device-side hand-over of sizes) -> decompress; HIP events on the launch stream"""
        ev[0].record()
        enc.compress(args.level, None, stream)
        ev[1].record()
        dec.link_src_sizes(enc, stream)
        dec.decompress(None, stream)
        ev[2].record()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()

    # correctness gate (outside the timed region): every frame compressed, decoded, and bit-exact
    csizes = enc.results(stream)
    dsizes = dec.results(stream)
    assert all(c < 2 ** 63 for c in csizes), "compress errors: %s" % [zstd_amd.ErrorCode(c - 2 ** 64) for c in csizes if c >= 2 ** 63][:3]
    assert all(d == FS for d in dsizes), "decompress errors"
    assert bool(torch.equal(out, src)), "round trip is not bit-exact"
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
        step(True)
        if world > 1:   # the path's only exchange: all ranks learn all compressed sizes (global output offsets)
            if sizes_dev is None:
                sizes_dev = torch.tensor(csizes, dtype=torch.int32, device=dev)
                gathered = [torch.empty_like(sizes_dev) for _ in range(world)]
            dist.all_gather(gathered, sizes_dev)
        ev[2].synchronize()
        t_c += ev[0].elapsed_time(ev[1])
        t_d += ev[1].elapsed_time(ev[2])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, t_c, t_d], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_c, t_d = [float(x) for x in t.tolist()]
        tot = torch.tensor([total_c], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        total_c_all = float(tot.item())
    else:
        total_c_all = float(total_c)

    if rank == 0:
        K = args.steps
        bytes_rank = float(N) * FS
        bytes_all = bytes_rank * world
        ms_step = elapsed / K * 1e3
        kc_ms, kd_ms = t_c / K, t_d / K          # average launch durations (HIP events on the launch stream)
        algo_c = bytes_rank + float(total_c)      # compress = srcSize + cSize per frame
        algo_d = float(total_c) + bytes_rank      #              decompress = cSize + dstSize
        wg, lds = enc.last_launch()
        traffic_c, traffic_d, traffic_src = committed_traffic(N, FS, args.level)
        res = {
            "metric": "GB/s compress & decompress round trip (whole job, uncompressed bytes), 128 KiB frames, level 1; ratio",
            "value": round(bytes_all / elapsed * K / 1e9, 3),
            "unit": "GB/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d KiB independent level-%d frames per GPU, fixed-seed log-like corpus (seed %d), compress then decompress, device resident"
                                   % (N, FS // 1024, args.level, args.seed),
                       "frames_per_gpu": N, "frame_bytes": FS, "level": args.level, "parallelism": "batch split x%d" % world},
            "compress_GBps": round(bytes_all / (kc_ms / 1e3) / 1e9, 3),
            "decompress_GBps": round(bytes_all / (kd_ms / 1e3) / 1e9, 3),
            "ratio": round(bytes_all / total_c_all, 4),
            "bit_exact_round_trip": True,
            "byte_identical_to_oracle_sample": identical,
            "roofline": {"bound": "hbm", "kernel": "zsa_k_compress", "achieved": round(algo_c / (kc_ms / 1e3) / 1e9, 3),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(algo_c / (kc_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 6),
                         "traffic": traffic_c, "traffic_source": traffic_src, "avg_launch_ms": round(kc_ms, 3), "algorithmic_bytes_per_launch": int(algo_c),
                         "workgroups": wg, "lds_bytes_per_workgroup": lds},
            "roofline_decompress": {"bound": "hbm", "kernel": "zsa_k_decompress", "achieved": round(algo_d / (kd_ms / 1e3) / 1e9, 3),
                                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(algo_d / (kd_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 6),
                                    "traffic": traffic_d, "avg_launch_ms": round(kd_ms, 3), "algorithmic_bytes_per_launch": int(algo_d)},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(FS, N, args.seed, args.level)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


# HBM bytes per launch need PMC passes of their own (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot ride along with a
# timed run); tools/collect_profiles.sh runs them on this very command and tools/summarize_profiles.py commits the
# per-launch averages, with the guide's gfx950 correction (FETCH_SIZE counts 128-B requests as 64 B: x2), to this file.
PMC_SUMMARY = "profiles/r01_v10_pmc.json"


def committed_traffic(n_frames, frame_bytes, level):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_SUMMARY)
    if not (os.path.exists(path) and n_frames == 8192 and frame_bytes == 131072 and level == 1):
        return None, None, None      # the summary was taken on the default workload only
    pmc = json.load(open(path))
    get = lambda k: (int(pmc[k]["hbm_traffic_bytes"]) if k in pmc and "hbm_traffic_bytes" in pmc[k] else None)
    return get("zsa_k_compress"), get("zsa_k_decompress"), PMC_SUMMARY + " (separate rocprofv3 --pmc passes of this command, per launch)"


if __name__ == "__main__":
    main()
