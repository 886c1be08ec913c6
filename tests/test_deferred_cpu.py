"""Bookkeeping of the deferred weight-gradient reductions (gaot_3d_amd.ops.defer_ok / _defer / _task_done), exercised on the CPU
autograd engine with the ONE device launch (gaot_reduce_multi) replaced by a torch sum: the questions here are which pass a
gradient is completed in and whether anything is ever dropped, not the arithmetic (tests/test_deferred_gpu.py has that, bit for
bit).  Cases from ADVICE r5: nested (reentrant) backward passes, a process group of ONE rank, non-contiguous parameters, the
opt-in default.  Reference operators: nn.Linear autograd under torch.utils.checkpoint (src/model/layers/attn.py:146-178 is what a
trainer would wrap)."""
import os

import pytest
import torch
from torch.utils.checkpoint import checkpoint

from gaot_3d_amd import ops


@pytest.fixture()
def sim(monkeypatch):
    """ops._flush_pending without the device: out = sum of the partial table; counts flushes and completed tables"""
    stats = {"flushes": 0, "tables": 0}

    def fake_flush():
        pend, ops._DEFER["pending"], ops._DEFER["pending_bytes"] = ops._DEFER["pending"], [], 0
        if not pend:
            return
        stats["flushes"] += 1
        for part, storage, out_ptr, n, parts, lanes in pend:
            out = torch.empty(0, dtype=torch.float32).set_(storage, (out_ptr - storage.data_ptr()) // 4, (n,))
            out.copy_(part.reshape(parts, n).sum(0))
            stats["tables"] += 1
    monkeypatch.setattr(ops, "_flush_pending", fake_flush)
    prev = ops.defer_reductions(True)
    yield stats
    ops.defer_reductions(prev)


class SplitLinear(torch.autograd.Function):
    """y = x w^T whose weight gradient is produced as 4 partial products: the shape of the split-K dW the GPU path defers"""
    log = []

    @staticmethod
    def forward(ctx, x, w, name):
        ctx.save_for_backward(x, w)
        ctx.name = name
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        rows = x.shape[0] // 4
        part = torch.stack([dy[i * rows:(i + 1) * rows].t() @ x[i * rows:(i + 1) * rows] for i in range(4)])
        defer = ops.defer_ok((w,))
        SplitLinear.log.append((ctx.name, defer))
        if defer:
            dw = torch.full(w.shape, float("nan"))       # values that do not exist until the flush
            ops._defer(part.reshape(4, -1), dw, w.numel(), 4, 4)
        else:
            dw = part.sum(0)
        return dy @ w, dw, None


def _weights(n=3, d=8):
    torch.manual_seed(0)
    return [torch.nn.Parameter(torch.randn(d, d) * 0.3) for _ in range(n)]


def _reference(ws, x, fn):
    prev = ops.defer_reductions(False)
    try:
        for w in ws:
            w.grad = None
        fn(ws, x).sum().backward()
        return [w.grad.clone() for w in ws]
    finally:
        ops.defer_reductions(prev)


def test_default_is_off():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "GAOT_DEFER_REDUCE"}
    out = subprocess.run([sys.executable, "-c", "from gaot_3d_amd import ops; print(ops._DEFER['enabled'])"], env=env,
                         capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.stdout.strip() == "False", out.stdout + out.stderr


def test_plain_pass_defers_everything_to_one_flush(sim):
    ws = _weights()
    x = torch.randn(16, 8)

    def fn(ws, x):
        for i, w in enumerate(ws):
            x = SplitLinear.apply(x, w, f"l{i}")
        return x
    ref = _reference(ws, x, fn)
    for w in ws:
        w.grad = None
    SplitLinear.log.clear()
    fn(ws, x).sum().backward()
    assert all(d for _n, d in SplitLinear.log) and sim["flushes"] == 1 and sim["tables"] == 3
    assert ops.deferred_pending() == 0 and ops._DEFER["stack"] == []
    for w, r in zip(ws, ref):
        assert torch.allclose(w.grad, r, atol=1e-6)


def test_reentrant_checkpoint_never_drops_the_outer_pending_list(sim):
    """w2 sits in a reentrant checkpoint; its backward is a NESTED graph task.  The outer pass deferred w3
    before it: that entry must be completed (not dropped) when the nested pass starts"""
    ws = _weights()
    x = torch.randn(16, 8, requires_grad=True)

    def fn(ws, x):
        x = SplitLinear.apply(x, ws[0], "w1")
        x = checkpoint(lambda t: SplitLinear.apply(t, ws[1], "w2"), x, use_reentrant=True)
        return SplitLinear.apply(x, ws[2], "w3")
    ref = _reference(ws, x, fn)
    for w in ws:
        w.grad = None
    SplitLinear.log.clear()
    fn(ws, x).sum().backward()
    assert [n for n, _d in SplitLinear.log] == ["w3", "w2", "w1"] and all(d for _n, d in SplitLinear.log)
    assert sim["tables"] == 3 and ops.deferred_pending() == 0 and ops._DEFER["stack"] == []
    for w, r in zip(ws, ref):
        assert not torch.isnan(w.grad).any()
        assert torch.allclose(w.grad, r, atol=1e-6)


class InnerGrad(torch.autograd.Function):
    """a custom Function whose backward runs torch.autograd.grad on a graph of its own (a nested task inside the outer pass)"""
    @staticmethod
    def forward(ctx, x, w):
        ctx.x, ctx.w = x.detach(), w
        return x.detach() @ w.detach().t()

    @staticmethod
    def backward(ctx, dy):
        with torch.enable_grad():
            xi = ctx.x.clone().requires_grad_(True)
            y = SplitLinear.apply(xi, ctx.w, "inner")
        dx, dw = torch.autograd.grad(y, [xi, ctx.w], dy)
        assert not torch.isnan(dw).any(), "autograd.grad returned before the nested pass completed its deferred gradient"
        return dx, dw


def test_autograd_grad_inside_a_backward(sim):
    ws = _weights()
    x = torch.randn(16, 8, requires_grad=True)

    def fn(ws, x):
        x = SplitLinear.apply(x, ws[0], "w1")
        x = InnerGrad.apply(x, ws[1])
        return SplitLinear.apply(x, ws[2], "w3")
    ref = _reference(ws, x, fn)
    for w in ws:
        w.grad = None
    fn(ws, x).sum().backward()
    assert ops.deferred_pending() == 0 and ops._DEFER["stack"] == []
    for w, r in zip(ws, ref):
        assert torch.allclose(w.grad, r, atol=1e-6)


def test_parameter_shared_between_outer_and_nested_pass(sim):
    """the same weight inside and outside the checkpoint: the second sighting completes everything and takes the in-call pass"""
    ws = _weights(2)
    x = torch.randn(16, 8, requires_grad=True)

    def fn(ws, x):
        x = SplitLinear.apply(x, ws[0], "a")
        x = checkpoint(lambda t: SplitLinear.apply(t, ws[1], "b_in"), x, use_reentrant=True)
        x = SplitLinear.apply(x, ws[1], "b_out")
        return SplitLinear.apply(x, ws[0], "a2")
    ref = _reference(ws, x, fn)
    for w in ws:
        w.grad = None
    SplitLinear.log.clear()
    fn(ws, x).sum().backward()
    d = dict(SplitLinear.log)
    assert d["a2"] and d["b_out"] and not d["a"]      # a: second sighting in the outer pass -> in-call
    assert not d["b_in"]                               # seen by the ENCLOSING pass -> in-call in the nested one
    for w, r in zip(ws, ref):
        assert torch.allclose(w.grad, r, atol=1e-6)


def test_a_pass_that_dies_leaves_nothing_behind_for_the_next(sim):
    ws = _weights()
    x = torch.randn(16, 8)

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    def bad(ws, x):
        x = SplitLinear.apply(x, ws[0], "w1")
        x = Boom.apply(x)
        return SplitLinear.apply(x, ws[2], "w3")

    def good(ws, x):
        for i, w in enumerate(ws):
            x = SplitLinear.apply(x, w, f"l{i}")
        return x
    ref = _reference(ws, x, good)
    for w in ws:
        w.grad = None
    with pytest.raises(RuntimeError, match="boom"):
        bad(ws, x).sum().backward()
    assert ops.deferred_pending() == 1 and len(ops._DEFER["stack"]) == 1     # w3's entry of the dead pass
    for w in ws:
        w.grad = None
    SplitLinear.log.clear()
    good(ws, x).sum().backward()
    # w3 was seen by the dead pass: conservative in-call pass once; the top-level callback then clears the stale bookkeeping
    assert ops.deferred_pending() == 0 and ops._DEFER["stack"] == []
    for w, r in zip(ws, ref):
        assert torch.allclose(w.grad, r, atol=1e-6)
    SplitLinear.log.clear()
    for w in ws:
        w.grad = None
    good(ws, x).sum().backward()
    assert all(d for _n, d in SplitLinear.log)


def test_declines_beside_any_process_group_and_for_strided_parameters(sim, tmp_path):
    import torch.distributed as dist
    ws = _weights(1)
    x = torch.randn(16, 8)
    strided = torch.nn.Parameter(torch.randn(8, 16)[:, ::2])
    assert not strided.is_contiguous()
    SplitLinear.log.clear()
    SplitLinear.apply(x, strided, "strided").sum().backward()
    assert SplitLinear.log == [("strided", False)]
    dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1)
    try:
        SplitLinear.log.clear()
        SplitLinear.apply(x, ws[0], "one_rank_group").sum().backward()
        assert SplitLinear.log == [("one_rank_group", False)]      # DDP's reducer hooks AccumulateGrad at world size 1 too
    finally:
        dist.destroy_process_group()
    SplitLinear.log.clear()
    ws[0].grad = None
    SplitLinear.apply(x, ws[0], "no_group").sum().backward()
    assert SplitLinear.log == [("no_group", True)]


def test_pending_bytes_cap_flushes_early(sim, monkeypatch):
    monkeypatch.setitem(ops._DEFER, "cap_bytes", 4 * 64 * 4 * 2)     # room for two 4 x 64 fp32 tables
    ws = _weights(5)
    x = torch.randn(16, 8)

    def fn(ws, x):
        for i, w in enumerate(ws):
            x = SplitLinear.apply(x, w, f"l{i}")
        return x
    ref = _reference(ws, x, fn)
    for w in ws:
        w.grad = None
    fn(ws, x).sum().backward()
    assert sim["flushes"] >= 2 and sim["tables"] == 5
    for w, r in zip(ws, ref):
        assert torch.allclose(w.grad, r, atol=1e-6)
