import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle runs torch's CPU kernels: on the pool's 256-thread hosts the default (all hardware threads) is several times SLOWER than
    # 32 (profiles/r6_zm_attn_oracle_threads.txt: the eight S = 16 384 attention checks 2-5 s each on 32 threads, 5-6 s on 64, minutes on
    # 256; bench.py's cpu_baseline measured 9x for the whole step).  Tests that know better set their own count.
    import torch
    n = os.cpu_count() or 1
    if n > 32 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(int(os.environ.get("GAOT_ORACLE_THREADS", 32)))


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


_LOUD_SKIPS = []


def pytest_terminal_summary(terminalreporter):
    """parity tests that skipped themselves (host too small for the whole-step oracle, ...) are named in the summary even under -q"""
    for line in _LOUD_SKIPS:
        terminalreporter.write_line(line, yellow=True)


def pytest_runtest_logreport(report):
    """Keep the achieved errors: every `[parity] ...` line a test prints (max-abs / max-rel / cosine against the oracle or
    the goldens) is appended to $GAOT_PARITY_LOG (default gpurun_out/parity_last.txt on a GPU box); tools/gpu_pass.sh
    copies that file to profiles/parity_<tag>.txt."""
    if report.when != "call":
        return
    lines = [ln for ln in (report.capstdout or "").splitlines() if ln.startswith("[parity]") or ln.startswith("[train]")]
    if not lines:
        return
    _LOUD_SKIPS.extend(ln for ln in lines if ": skipped" in ln)
    path = os.environ.get("GAOT_PARITY_LOG")
    if path is None:
        import torch
        if not torch.cuda.is_available():
            return
        path = os.path.join(REPO, "gpurun_out", "parity_last.txt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        f.write(f"# {report.nodeid} [{report.outcome}]\n")
        for ln in lines:
            f.write(ln + "\n")
