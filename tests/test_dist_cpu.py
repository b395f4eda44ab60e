"""world_size-2 gloo test of the multi-GPU plan (no GPU needed): contiguous batch split of a corpus across ranks,
one all-gather of per-frame compressed sizes, identical global offsets on every rank."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from conftest import _Lib
    from zstd_amd import corpus
    orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")
    per = 6
    first, count = bench.shard_plan(per, rank, world)
    frames = corpus.frames("log", 16384, count, 42, first_frame=first)
    sizes = torch.tensor([len(orc.compress(f, 1)) for f in frames], dtype=torch.int32)   # stands in for the GPU kernel's result words
    gathered = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(gathered, sizes)
    offs, total = bench.global_offsets([g.tolist() for g in gathered])
    q.put((rank, first, [int(x) for x in sizes], offs, total))
    dist.destroy_process_group()


def test_two_rank_split_and_offsets():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(30) for p in ps]
    assert [r[1] for r in res] == [0, 6]                       # disjoint contiguous shards
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]   # every rank derives the same global layout
    flat = res[0][2] + res[1][2]
    acc = 0
    for row, sizes in zip(res[0][3], (res[0][2], res[1][2])):
        for o, s in zip(row, sizes):
            assert o == acc
            acc += s
    assert acc == res[0][4] == sum(flat)
    # the split corpus equals the unsplit one
    from zstd_amd import corpus
    assert corpus.frames("log", 16384, 12, 42) == corpus.frames("log", 16384, 6, 42) + corpus.frames("log", 16384, 6, 42, first_frame=6)


def _bench(*args, timeout=240):
    import subprocess
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_spawns_ranks_itself():
    """`bench.py --gpus 2` without WORLD_SIZE starts two ranks through torch.distributed.run (the entry the driver's
    N-GPU runs take); rehearsed here over gloo with made-up sizes: shard plan, all-gather of sizes, global offsets equal
    on both ranks and equal to the host restatement."""
    import json
    r = _bench("--gpus", "2", "--exchange-dry-run", "--frames", "257")
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_ranks"] == 2 and d["offsets_agree"] is True and d["frames_per_rank"] == 257


def test_bench_refuses_more_gpus_than_the_box_has():
    """asking for more GPUs than exist must fail loudly, never run a smaller job under the larger name"""
    import torch
    have = torch.cuda.device_count()
    r = _bench("--gpus", str(have + 2), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "GPU(s)" in r.stderr and not any(l.startswith('{"metric"') for l in r.stdout.splitlines())
    # a rank count that contradicts --gpus is refused as well
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
