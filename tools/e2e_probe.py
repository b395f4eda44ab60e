#!/usr/bin/env python3
"""Host-batch (PCIe-inclusive) timing probe: bench.py's end_to_end leg alone.  ZSTD_AMD_TRACE=1 prints the library's timeline."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, zstd_amd
from zstd_amd import corpus
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.cuda.set_device(0)
host = corpus.generate("log", fs, n, 42, threads=16)
print(json.dumps(bench.end_to_end(zstd_amd, host, fs, n, level)))
