#!/usr/bin/env python3
"""Runs the kernel bodies under the lane emulator built with AddressSanitizer (tests/emu/libzsa_emu_asan.so) on inputs placed
flush against the ends of their heap blocks, and compares every frame with the oracle.  An out-of-bounds access of a kernel
body -- which the GPU answers with a memory fault that takes the whole box down -- aborts here with a report naming the line.
usage: check_emu_asan.py [levels ...]   (default 1 3 5; starts itself again under LD_PRELOAD=libasan)"""
import ctypes, os, random, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "ZSA_ASAN_CHILD" not in os.environ:
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), os.path.join(ROOT, "tests", "emu", "libzsa_emu_asan.so")], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, ZSA_ASAN_CHILD="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    sys.exit(subprocess.run([sys.executable] + sys.argv, env=env).returncode)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import _Lib
from zstd_amd import corpus
import zstd_amd
orc = _Lib(os.path.join(ROOT, "oracle", "liboracle.so"), "zo_")
emu = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libzsa_emu_asan.so"))
emu.emu_compress_level.restype = ctypes.c_size_t
emu.emu_compress_level.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
emu.emu_decompress.restype = ctypes.c_size_t
emu.emu_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
levels = [int(a) for a in sys.argv[1:] if a.lstrip("-").isdigit()] or [1, 3, 5]
rnd = random.Random(6)
inputs = [corpus.frames("log", n, 1, 30 + i)[0] for i, n in enumerate([1, 9, 17, 300, 4000, 16385, 100000, 131073, 262145])]
inputs += [bytes(70000), rnd.randbytes(30000), rnd.randbytes(3000) + bytes(50000) + rnd.randbytes(3000), b"ab" * 20000,
           bytes(rnd.choice(b"abcde") for _ in range(30000)), corpus.frames("json", 65536, 1, 2)[0]]
bad = 0
for level in levels:
    for d in inputs:
        want = orc.compress(d, level)
        cap = zstd_amd.CompressBound(len(d))
        buf = ctypes.create_string_buffer(cap)          # (exactly the bound: a write past it is reported)
        n = emu.emu_compress_level(buf, cap, d, len(d), level)
        ok = isinstance(want, bytes) and buf.raw[:n] == want
        if ok:
            out = ctypes.create_string_buffer(max(len(d), 1))
            m = emu.emu_decompress(out, len(d), want, len(want))
            ok = m == len(d) and out.raw[:m] == d
        print("level %d, %d bytes: %s" % (level, len(d), "ok" if ok else "DIFFERS"), flush=True)
        bad += not ok
# dictionaries: digests and the attached-dictionary parsers at every strategy (payloads and dictionaries in exact-size heap blocks)
from test_oracle_dict import load_dict_cases
emu.emu_create_cdict.restype = ctypes.c_void_p
emu.emu_create_cdict.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
emu.emu_compress_cdict.restype = ctypes.c_size_t
emu.emu_compress_cdict.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
emu.emu_free_cdict.argtypes = [ctypes.c_void_p]
dicts = load_dict_cases()[0]
if "nodict" not in sys.argv:
    for dn in sorted(dicts):
        d = dicts[dn]
        pl = [corpus.frames("json", n, 1, 50 + i)[0] for i, n in enumerate([1, 9, 300, 1024, 8192, 16384, 30000])]
        pl += [(d[-min(len(d), 700):] + rnd.randbytes(20)) * 12, d[:min(len(d), 5000)], bytes(3000), rnd.randbytes(2000)]
        # literals that a dictionary's Huffman table makes LONGER than they are (taken over without an estimate up to 1 KiB of
        # literals, huf_compress.c:1336-1341): the stream writers must stop at the frame's end
        pl += [rnd.randbytes(300), rnd.randbytes(700), rnd.randbytes(1000), bytes(rnd.randrange(128, 256) for _ in range(900)),
               d[:min(len(d), 200)] + bytes(rnd.randrange(200, 256) for _ in range(800))]
        for level in (1, 3, 4, 5, 7, 8):
            cd = emu.emu_create_cdict(d, len(d), level)
            ocd = orc.cdict(d, level)
            if not cd or not ocd:
                print("dictionary %s level %d: no digest (emu %s, oracle %s)" % (dn, level, bool(cd), bool(ocd)), flush=True)
                bad += bool(cd) != bool(ocd)
                continue
            nbad = 0
            for p in pl:
                want = orc.compress_cdict(p, ocd)
                cap = zstd_amd.CompressBound(len(p))
                buf = ctypes.create_string_buffer(cap)
                n = emu.emu_compress_cdict(buf, cap, p, len(p), cd)
                ok = (buf.raw[:n] == want) if isinstance(want, bytes) else (n == 2 ** 64 + want)
                nbad += not ok
            print("dictionary %s (%d bytes), level %d: %s" % (dn, len(d), level, "ok" if not nbad else "%d DIFFER" % nbad), flush=True)
            bad += nbad
            emu.emu_free_cdict(cd); orc.free_cdict(ocd)
print("mismatches", bad)
sys.exit(1 if bad else 0)
