"""tests/abi_smoke.c: the C ABI called from plain C with the call sequence of the Go files under shim/ (this image has no
Go toolchain).  Header-only calls run anywhere; the calls that move data need the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "abi_smoke")


def _run(*args, env=None):
    assert os.path.exists(BIN), "tests/abi_smoke is missing: run python -c 'import __graft_entry__ as g; g.build()'"
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([BIN, *args], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_abi_header_only_calls():
    assert "abi_smoke ok" in _run()


@pytest.mark.gpu
def test_abi_calls_that_move_data():
    assert "abi_smoke ok (gpu)" in _run("gpu")


def test_engine_failure_is_flagged_and_falls_back_without_a_device():
    """SURVEY 8b "any device error => CPU re-run": with no usable device (this container has none; on a GPU box the child
    is started with the devices hidden, before it makes any HIP call) every data call fails with a code AND
    zsa_engine_failure() != 0, which is what the Go hooks test before answering with the vendored C."""
    out = _run("enginefail", env={"HIP_VISIBLE_DEVICES": "-1", "ROCR_VISIBLE_DEVICES": "-1"})
    assert out.count("-> vendored C") == 3 and "abi_smoke ok" in out


@pytest.mark.gpu
def test_engine_failure_on_refused_allocations_and_not_on_data_errors():
    out = _run("enginefail", env={"ZSTD_AMD_FAIL": "alloc"})
    assert out.count("-> vendored C") == 3
    assert "data errors are not flagged" in _run("enginefail")


def test_shim_binds_only_declared_entry_points():
    """Every C.zsa_* name the Go files use is declared in include/zstd_amd.h (and so exported: test_host.py checks that)."""
    import re
    header = open(os.path.join(ROOT, "include", "zstd_amd.h")).read()
    declared = set(re.findall(r"\b(zsa_\w+)\s*\(", header)) | set(re.findall(r"\b(zsa_\w+);", header))
    used = set()
    for f in os.listdir(os.path.join(ROOT, "shim")):
        if f.endswith(".go"):
            used |= set(re.findall(r"C\.(zsa_\w+)", open(os.path.join(ROOT, "shim", f)).read()))
    assert used and used <= declared, sorted(used - declared)
