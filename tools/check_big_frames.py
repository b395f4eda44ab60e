#!/usr/bin/env python3
"""GPU box: 3, 5 and 9 MB frames (longer than every level's window) at levels 1-9, 12 and -5 against the oracle, and back."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import _Lib
import zstd_amd
from zstd_amd import corpus
orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")
frames = [corpus.frames("log", 3000000, 1, 71)[0], corpus.frames("json", 5000000, 1, 72)[0], corpus.frames("log", 9000000, 1, 73)[0]]
bad = 0
for level in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, -5):
    got = zstd_amd.CompressBatch(frames, level)
    for f, c in zip(frames, got):
        want = orc.compress(f, level)
        ok = (c == want) if not isinstance(want, int) else (isinstance(c, zstd_amd.ErrorCode) and c.code == want)
        print(level, len(f), "OK" if ok else "MISMATCH", c if isinstance(c, zstd_amd.ErrorCode) else len(c), want if isinstance(want, int) else len(want), flush=True)
        bad += not ok
    good = [(f, c) for f, c in zip(frames, got) if isinstance(c, bytes)]
    back = zstd_amd.DecompressBatch([c for _, c in good], [len(f) for f, _ in good])
    bad += sum(1 for (f, _), b in zip(good, back) if b != f)
print("bad", bad)
