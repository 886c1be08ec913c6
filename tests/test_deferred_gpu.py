"""Deferred completion of the weight-gradient reductions (gaot_gemm_ex_partials / gaot_reduce_multi, ABI 10; gaot_3d_amd.ops.defer_ok):
the split-K, RMSNorm-weight and bias-column-sum passes of a backward pass are handed to ONE launch at its end.  Contract checked here:
the gradients are BIT-identical to the in-call passes (same summation order), through loss.backward() and torch.autograd.grad(),
and every situation in which something could observe a gradient early takes the in-call pass (existing .grad, hooks, a parameter
used twice, calls outside a backward pass).  Reference operators: nn.Linear / RMSNorm autograd, src/model/layers/attn.py:104-106,
146-156, 205-230."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model_and_batch(precision, dropout=0.1):
    import gaot_3d_amd
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model import init_model
    from test_model_gpu import small_config
    cfg = small_config(layers=3, latent=(32, 32, 32), k=4, heads=8, dropout=dropout)     # 4 096 tokens: the weight gradients split K
    cfg.transformer.ffn_config.hidden_size = 1024
    torch.manual_seed(0)
    model = init_model(6, 1, "gaot_3d", cfg).to(DEV).train()
    batch, tokens = make_synthetic_sample(20000, cfg.latent_tokens, k=4, seed=1, device=DEV)
    gaot_3d_amd.set_precision(precision)
    return model, batch, tokens.to(DEV)


def _grads(model, batch, tokens, defer, via="backward"):
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF, ops
    prev = ops.defer_reductions(defer)
    try:
        gaot_3d_amd.clear_graph_cache(batch)
        GF.set_dropout_seed(1234, DEV)
        for p in model.parameters():
            p.grad = None
        ops.launch_count_reset()
        loss = GF.mse_loss(model(batch=batch, tokens_pos=tokens), batch.x)
        if via == "backward":
            loss.backward()
            gs = [p.grad for p in model.parameters()]
        else:
            got = iter(torch.autograd.grad(loss, [p for p in model.parameters() if p.requires_grad], allow_unused=True))
            gs = [next(got) if p.requires_grad else None for p in model.parameters()]
        n = ops.launch_count()
        assert ops.deferred_pending() == 0
        torch.cuda.synchronize()
        return [None if g is None else g.clone() for g in gs], n, float(loss)
    finally:
        ops.defer_reductions(prev)


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("via", ["backward", "grad"])
def test_deferred_reductions_are_bit_identical(precision, via):
    import gaot_3d_amd
    model, batch, tokens = _model_and_batch(precision)
    try:
        g0, n0, l0 = _grads(model, batch, tokens, False, via)
        g1, n1, l1 = _grads(model, batch, tokens, True, via)
        g2, n2, _ = _grads(model, batch, tokens, True, via)
    finally:
        gaot_3d_amd.set_precision("fp32")
    assert l0 == l1
    names = [n for n, _ in model.named_parameters()]
    for name, a, b, c in zip(names, g0, g1, g2):
        assert (a is None) == (b is None), name
        if a is not None:
            assert torch.equal(a, b), f"{name}: deferred != in-call, max diff {(a - b).abs().max().item():.3e}"
            assert torch.equal(b, c), f"{name}: deferred rerun differs"
    print(f"[deferred] {precision} {via}: launches per step {n0} -> {n1}")
    assert n1 < n0 - 8, (n0, n1)      # 3 layers: >= 4 split-K + 2 RMSNorm reductions per layer became one launch


def test_second_backward_accumulates_into_existing_grads():
    """p.grad exists -> AccumulateGrad ADDS the returned tensor at once: the in-call pass must be taken"""
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF, ops
    model, batch, tokens = _model_and_batch("bf16", dropout=0.0)
    try:
        g0, _, _ = _grads(model, batch, tokens, True)
        # second backward on top of the first one's gradients (gradient accumulation): exactly twice the first
        loss = GF.mse_loss(model(batch=batch, tokens_pos=tokens), batch.x)
        loss.backward()
        assert ops.deferred_pending() == 0
        for (name, p), a in zip(model.named_parameters(), g0):
            if a is not None:
                assert torch.equal(p.grad, a + a), name
    finally:
        gaot_3d_amd.set_precision("fp32")


def test_shared_weight_and_hooks_take_the_in_call_pass():
    """a weight used twice in one graph (the engine adds its two gradients as they arrive) and a parameter with a tensor hook"""
    from gaot_3d_amd import functional as GF, ops
    torch.manual_seed(0)
    m, k, n = 16384, 256, 256
    x = torch.randn(m, k, device=DEV)
    w = torch.nn.Parameter(torch.randn(n, k, device=DEV) * 0.05)
    w2 = torch.nn.Parameter(torch.randn(n, n, device=DEV) * 0.05)
    seen = []

    def run(defer, hook):
        prev = ops.defer_reductions(defer)
        try:
            w.grad = w2.grad = None
            h = w2.register_hook(lambda g: seen.append(float(g.abs().sum()))) if hook else None
            y = GF.linear(GF.linear(GF.linear(x, w, precision=1), w2, precision=1), w, precision=1)   # w twice (n == k)
            y.square().mean().backward()
            if h is not None:
                h.remove()
            torch.cuda.synchronize()
            assert ops.deferred_pending() == 0
            return w.grad.clone(), w2.grad.clone()
        finally:
            ops.defer_reductions(prev)

    a0, b0 = run(False, False)
    a1, b1 = run(True, False)
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    a2, b2 = run(True, True)
    assert torch.equal(a0, a2) and torch.equal(b0, b2)
    assert len(seen) == 1 and abs(seen[0] - float(b0.abs().sum())) <= 1e-3 * seen[0]    # the hook saw the COMPLETE gradient


def test_reduce_multi_matches_torch_and_keeps_the_in_call_order():
    """gaot_reduce_multi on odd sizes (scalar path), aligned sizes (16-byte path), all three lane counts, > 64 descriptors"""
    import ctypes as C
    from gaot_3d_amd import _lib, ops
    lib = _lib.load()
    torch.manual_seed(0)
    cases = [(37, 5, 4), (4096, 16, 4), (65536, 8, 4), (16384 + 4, 7, 4), (131072, 32, 4), (1000, 100, 16), (2048, 64, 16), (256, 129, 32), (259, 31, 32)] * 10   # 90 tables
    parts = [torch.randn(p, n, device=DEV) for (n, p, _) in cases]
    outs = [torch.full((n,), float("nan"), device=DEV) for (n, _, _) in cases]
    arr = (ops._ReduceDesc * len(cases))()
    for i, ((n, p, lanes), pt, o) in enumerate(zip(cases, parts, outs)):
        arr[i].part, arr[i].out, arr[i].n, arr[i].parts, arr[i].lanes = pt.data_ptr(), o.data_ptr(), n, p, lanes
    _lib.check(lib.gaot_reduce_multi(arr, len(cases), ops._stream()), "gaot_reduce_multi")
    torch.cuda.synchronize()
    for (n, p, lanes), pt, o in zip(cases, parts, outs):
        ref = torch.zeros(n, device=DEV)
        lane_sums = []
        for sl in range(lanes):                      # the order of k_splitk_reduce / k_reduce_parts, restated
            v = torch.zeros(n, device=DEV)
            for s in range(sl, p, lanes):
                v = v + pt[s]
            lane_sums.append(v)
        ref = lane_sums[0]
        for v in lane_sums[1:]:
            ref = ref + v
        assert torch.equal(o, ref), (n, p, lanes, (o - ref).abs().max().item())
        assert torch.allclose(o, pt.double().sum(0).float(), rtol=1e-4, atol=1e-4)
    # a bad descriptor is refused
    arr[0].lanes = 8
    assert lib.gaot_reduce_multi(arr, 1, ops._stream()) != 0


def _block_and_input(layers=3, s=4096, d=256):
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerBlock
    torch.manual_seed(0)
    blocks = torch.nn.ModuleList([
        TransformerBlock(d, d, attn_config=AttentionConfig(hidden_size=d, num_heads=8, num_kv_heads=8, atten_dropout=0.0),
                         ffn_config=FFNConfig(hidden_size=1024)) for _ in range(layers)]).to(DEV).train()
    x = torch.randn(1, s, d, device=DEV, requires_grad=True)
    return blocks, x


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_reentrant_checkpoint_around_a_block_is_bit_identical(precision):
    """VERDICT r5 #4a / ADVICE r5 (high): the middle block in torch.utils.checkpoint(use_reentrant=True) runs its backward as a
    NESTED graph task; the outer pass's pending partials must be completed, not dropped"""
    import gaot_3d_amd
    from torch.utils.checkpoint import checkpoint
    from gaot_3d_amd import ops
    blocks, x = _block_and_input()
    gaot_3d_amd.set_precision(precision)

    def run(defer):
        prev = ops.defer_reductions(defer)
        try:
            for p in blocks.parameters():
                p.grad = None
            x.grad = None
            h = blocks[0](x)
            h = checkpoint(blocks[1], h, use_reentrant=True)
            h = blocks[2](h)
            h.square().mean().backward()
            assert ops.deferred_pending() == 0
            torch.cuda.synchronize()
            return [p.grad.clone() for p in blocks.parameters()] + [x.grad.clone()]
        finally:
            ops.defer_reductions(prev)
    try:
        g0, g1 = run(False), run(True)
    finally:
        gaot_3d_amd.set_precision("fp32")
    names = [n for n, _ in blocks.named_parameters()] + ["x"]
    for n, a, b in zip(names, g0, g1):
        assert torch.isfinite(b).all(), n
        assert torch.equal(a, b), f"{n}: deferred != in-call under a reentrant checkpoint, max diff {(a - b).abs().max().item():.3e}"


def test_autograd_grad_inside_a_custom_backward_is_bit_identical():
    """a custom Function whose backward calls torch.autograd.grad on a graph holding a gaot block: a nested task that must return
    COMPLETE gradients, and must not lose what the enclosing pass had pending"""
    import gaot_3d_amd
    from gaot_3d_amd import ops
    blocks, x = _block_and_input()
    inner = blocks[1]
    inner_params = list(inner.parameters())

    class Nested(torch.autograd.Function):
        @staticmethod
        def forward(ctx, h, *params):
            ctx.save_for_backward(h)
            with torch.no_grad():
                return inner(h)

        @staticmethod
        def backward(ctx, dy):
            (h,) = ctx.saved_tensors
            with torch.enable_grad():
                hi = h.detach().requires_grad_(True)
                y = inner(hi)
            gs = torch.autograd.grad(y, [hi] + inner_params, dy)
            return gs

    gaot_3d_amd.set_precision("bf16")

    def run(defer):
        prev = ops.defer_reductions(defer)
        try:
            for p in blocks.parameters():
                p.grad = None
            x.grad = None
            h = blocks[0](x)
            h = Nested.apply(h, *inner_params)
            h = blocks[2](h)
            h.square().mean().backward()
            assert ops.deferred_pending() == 0
            torch.cuda.synchronize()
            return [p.grad.clone() for p in blocks.parameters()] + [x.grad.clone()]
        finally:
            ops.defer_reductions(prev)
    try:
        g0, g1 = run(False), run(True)
    finally:
        gaot_3d_amd.set_precision("fp32")
    for n, a, b in zip([n for n, _ in blocks.named_parameters()] + ["x"], g0, g1):
        assert torch.equal(a, b), f"{n}: max diff {(a - b).abs().max().item():.3e}"


def test_one_rank_ddp_gets_complete_gradients(tmp_path):
    """DistributedDataParallel's reducer hooks the gradient accumulators at world size 1 too and copies a gradient
    into its bucket the moment it is accumulated -- beside ANY initialised process group the in-call pass is taken"""
    import torch.distributed as dist
    import gaot_3d_amd
    from gaot_3d_amd import ops
    blocks, x = _block_and_input(layers=2)
    gaot_3d_amd.set_precision("bf16")

    def run(net, defer):
        prev = ops.defer_reductions(defer)
        try:
            for p in blocks.parameters():
                p.grad = None
            net(x.detach()).square().mean().backward()
            torch.cuda.synchronize()
            return [p.grad.clone() for p in blocks.parameters()]
        finally:
            ops.defer_reductions(prev)
    seq = torch.nn.Sequential(*blocks)
    try:
        g0 = run(seq, False)
        dist.init_process_group("gloo", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1)
        try:
            ddp = torch.nn.parallel.DistributedDataParallel(seq)
            g1 = run(ddp, True)
        finally:
            dist.destroy_process_group()
    finally:
        gaot_3d_amd.set_precision("fp32")
    for (n, _), a, b in zip(blocks.named_parameters(), g0, g1):
        assert torch.equal(a, b), f"{n}: max diff {(a - b).abs().max().item():.3e}"
