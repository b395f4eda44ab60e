import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, zstd_amd
from zstd_amd import corpus
import bench
dev=torch.device("cuda:0")
for kind, fs, n, seed in (("json",131072,8192,1234),("json",65536,16384,1234),("log",131072,8192,42)):
    r = bench.measure_config(zstd_amd, corpus, torch, dev, None, kind, fs, n, seed, 1, steps=3)
    print(kind, fs, r["compress_GBps"], r["decompress_GBps"], r["bit_exact_round_trip"], r["byte_identical_to_oracle_sample"], flush=True)
