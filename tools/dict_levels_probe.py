#!/usr/bin/env python3
"""Throughput of the BulkProcessor path by level (device resident): JSON payloads with the trained 8 KiB fixture dictionary.
usage: dict_levels_probe.py [payload bytes ...]   (default 1024 4096 16384)"""
import base64, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import zstd_amd
from zstd_amd import corpus
import bench
dev = torch.device("cuda:0")
g = json.load(open(os.path.join(ROOT, "tests", "golden", "dict_cases.json")))
trained = base64.b64decode(g["dicts"]["trained_json_8k"]["b64"])
sizes = [int(a) for a in sys.argv[1:]] or [1024, 4096, 16384]
for fs in sizes:
    n = max(4096, (1 << 28) // fs // 4)
    for level in (1, 3, 4, 5, 6, 7):
        r = bench.measure_config(zstd_amd, corpus, torch, dev, None, "json", fs, n, 7, level, dictionary=trained, steps=1)
        print("payload %6d x %7d level %d: compress %7.3f GB/s  decompress %7.3f GB/s  ratio %.3f  exact %s identical %s  %s" %
              (fs, n, level, r.get("compress_GBps", 0), r["decompress_GBps"], r["ratio"], r["bit_exact_round_trip"],
               r["byte_identical_to_oracle_sample"], r.get("kernel_compress")), flush=True)
