"""GPU checks of the operator-level drop-in boundary (SURVEY §8b "operator-level signatures to keep"): the reference's
``scatter(src, index, dim, out, dim_size, reduce)`` (src/model/layers/utils/scatter_native.py:4-54) against the golden
captured from it, activations passed as callables (integral_transform.py:35, mlp.py:227-335), ChannelMLP in the
reference's channels-first layout, nn.Dropout inside the channel MLPs, ``apply_neighbor_sampling`` at its reference
import path, and the per-batch neighbour-list cache (never stale across forwards)."""
import os
import sys
import types

import pytest
import torch
import torch.nn.functional as F

import golden_io as gio

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import gaot_oracle as orc  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(name, a, b, rtol, atol):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    peak = b.abs().max().item() if b.numel() else 0.0
    print(f"[parity] {name}: max_abs={err:.3e} ref_peak={peak:.3e} max_rel_to_peak={err / max(peak, 1e-30):.3e}")
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max abs err {err:.3e}"


@pytest.mark.parametrize("index_dtype", [torch.int64, torch.int32])
def test_scatter_matches_reference_golden(index_dtype):
    """the four reductions of scatter_native on the golden's variable-degree index (empty rows, one row of degree > 32),
    through the reference's import path and signature; gradients against the oracle's autograd"""
    from gaot_3d_amd.model.layers.utils.scatter_native import scatter, scatter_native
    assert scatter is scatter_native
    meta, g = gio.load("ops")
    idx_cpu = g["in"]["edge_index"][1].long()
    src_cpu = g["in"]["scatter_src"]
    idx = idx_cpu.to(DEV, index_dtype)
    for red in ("sum", "mean", "max", "min"):
        src = src_cpu.to(DEV).requires_grad_(True)
        out = scatter(src, idx, dim=0, dim_size=meta["nq"], reduce=red)
        close(f"scatter_{red}", out, g["out"][f"scatter_{red}"], 1e-6, 1e-6)
        w = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
        (out * w.to(DEV)).sum().backward()
        src_r = src_cpu.clone().requires_grad_(True)
        (orc.scatter(src_r, idx_cpu, meta["nq"], red) * w).sum().backward()
        close(f"scatter_{red}/grad", src.grad, src_r.grad, 1e-6, 1e-6)
    # aliases, 1-D src, `out=` (overwritten, like the reference's out.fill_(0) + scatter), dim_size=None
    v = src_cpu[:, 0].contiguous().to(DEV)
    close("scatter_add_1d", scatter(v, idx, 0, None, meta["nq"], "add"), g["out"]["scatter_sum"][:, 0], 1e-6, 1e-6)
    buf = torch.full((meta["nq"], 5), 7.0, device=DEV)
    res = scatter(src_cpu.to(DEV), idx, dim=0, out=buf, dim_size=meta["nq"], reduce="amax")
    assert res is buf
    close("scatter_out_amax", buf, g["out"]["scatter_max"], 1e-6, 1e-6)
    auto = scatter(src_cpu.to(DEV), idx, dim=0, reduce="amin")
    assert auto.shape[0] == int(idx_cpu.max()) + 1
    close("scatter_auto_amin", auto, g["out"]["scatter_min"][:auto.shape[0]], 1e-6, 1e-6)
    # empty input and the reference's errors
    e = scatter(torch.empty(0, 5, device=DEV), torch.empty(0, dtype=torch.long, device=DEV), dim=0, dim_size=4, reduce="sum")
    assert e.shape == (4, 5) and float(e.abs().sum()) == 0.0
    with pytest.raises(NotImplementedError):
        scatter(src_cpu.to(DEV), idx, dim=1, dim_size=meta["nq"])
    with pytest.raises(ValueError):
        scatter(src_cpu.to(DEV), idx, dim=0, dim_size=meta["nq"], reduce="prod")


def test_scatter_large_unsorted_bit_reproducible():
    """2 M contributions onto 100 K rows in random order: equals the oracle, and two runs are bit-identical (fixed order)"""
    from gaot_3d_amd.model.layers.utils.scatter_native import scatter
    g = torch.Generator().manual_seed(5)
    e, r = 2_000_000, 100_000
    idx = torch.randint(0, r, (e,), generator=g)
    src = torch.randn(e, 4, generator=g)
    a = scatter(src.to(DEV), idx.to(DEV), dim=0, dim_size=r, reduce="mean")
    b = scatter(src.to(DEV), idx.to(DEV), dim=0, dim_size=r, reduce="mean")
    assert torch.equal(a, b)
    close("scatter_mean_2M", a, orc.scatter(src, idx, r, "mean"), 1e-5, 1e-6)


def test_activations_as_callables_and_channel_first_layout():
    """reference-style construction: IntegralTransform(channel_mlp_non_linearity=F.gelu) takes the fused kernels and equals
    the golden; ChannelMLP(non_linearity=F.gelu) on the reference's [C, N] / [B, C, N] layouts equals torch's own Conv1d"""
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    from gaot_3d_amd.model.layers.mlp import ChannelMLP, LinearChannelMLP, activation_name
    gaot_3d_amd.set_precision("fp32")
    assert activation_name(F.gelu) == "gelu" and activation_name(F.relu) == "relu" and activation_name(torch.nn.SiLU()) == "silu"
    # the reference's activation_fn(name) surface (mlp.py:27-35): any F.<name>, "swish", "none"
    assert activation_name(torch.tanh) == "tanh" and activation_name("swish") == "silu" and activation_name(F.leaky_relu) == "leaky_relu"
    assert activation_name(torch.nn.ELU()) == "elu" and activation_name(None) == "none"
    with pytest.raises(NotImplementedError):
        activation_name(torch.nn.ELU(alpha=0.5))       # only torch's default parameters have a kernel
    with pytest.raises(NotImplementedError):
        activation_name(lambda x: x)                    # an anonymous callable cannot be recognised: pass "none"
    meta, g = gio.load("ops")
    tag = "it_linear_noattn"
    it = IntegralTransform(channel_mlp_layers=[6, 64, 64, 32], channel_mlp_non_linearity=F.gelu, transform_type="linear",
                           coord_dim=3)
    it.load_state_dict(gio.sub(g["sd"], tag), strict=True)
    it = it.to(DEV)
    pos, lat, ei = g["in"]["pos"].to(DEV), g["in"]["lat"].to(DEV), g["in"]["edge_index"].to(DEV)
    f = g["in"]["f_y"].to(DEV)
    assert it._fused_eligible(list(it.channel_mlp.fcs), f)
    close("it_callable_gelu/out", it(y_pos=pos, x_pos=lat, edge_index=ei, f_y=f), g["out"][f"{tag}/out"], 1e-4, 1e-5)

    torch.manual_seed(3)
    for act, tact in ((F.gelu, F.gelu), (F.relu, F.relu), ("silu", F.silu), (torch.tanh, torch.tanh), (F.elu, F.elu),
                      ("leaky_relu", F.leaky_relu), (torch.nn.Mish(), F.mish), ("none", lambda t: t)):
        mlp = ChannelMLP(in_channels=12, out_channels=5, hidden_channels=24, n_layers=3, n_dim=1, non_linearity=act)
        x = torch.randn(12, 301)

        def ref(t):
            for i, fc in enumerate(mlp.fcs):
                t = F.conv1d(t, fc.weight, fc.bias)
                if i < mlp.n_layers - 1:
                    t = tact(t)
            return t
        want2, want3 = ref(x), ref(torch.stack([x, 2 * x]))
        mlp = mlp.to(DEV)
        name = activation_name(act)
        close(f"channel_mlp_{name}_[C,N]", mlp(x.to(DEV)), want2, 1e-4, 1e-5)
        close(f"channel_mlp_{name}_[B,C,N]", mlp(torch.stack([x, 2 * x]).to(DEV)), want3, 1e-4, 1e-5)
        close(f"channel_mlp_{name}_[B,C,7,43]", mlp(torch.stack([x, 2 * x]).view(2, 12, 7, 43).to(DEV)), want3.view(2, 5, 7, 43),
              1e-4, 1e-5)
        close(f"channel_mlp_{name}_rows", mlp.forward_rows(x.t().contiguous().to(DEV)), want2.t(), 1e-4, 1e-5)
    lin = LinearChannelMLP([12, 24, 5], non_linearity=F.relu).to(DEV)
    xr = torch.randn(77, 12)
    want = F.linear(F.relu(F.linear(xr, lin.fcs[0].weight.cpu(), lin.fcs[0].bias.cpu())), lin.fcs[1].weight.cpu(), lin.fcs[1].bias.cpu())
    close("linear_channel_mlp_relu", lin(xr.to(DEV)), want, 1e-4, 1e-5)


def test_mlp_dropout_mask_is_the_oracles_and_backward_reuses_it():
    """nn.Dropout of the channel MLPs (mlp.py:318-322): the draw is a counter-based hash (integer work, bit-exact against
    the oracle's restatement); kept values are x / (1 - p); the gradient passes through the same mask; eval mode is the
    identity; a LinearChannelMLP(dropout=0.3) trains (different masks per call) and is deterministic in eval"""
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd.model.layers.mlp import LinearChannelMLP
    p = 0.3
    seed0 = 0x1234_5678_9ABC_DEF1
    GF.set_dropout_seed(seed0, DEV)
    x = torch.randn(1000, 37, device=DEV).requires_grad_(True)
    y = GF.dropout(x, p, True)
    keep = orc.element_dropout_keep(GF.dropout_seed_sequence(seed0, 1)[0], x.numel(), p).view(1000, 37)
    got_keep = (y != 0).cpu()
    assert torch.equal(got_keep | (x.detach().cpu() == 0), keep | (x.detach().cpu() == 0))
    close("dropout_values", y, torch.where(keep, x.detach().cpu() / (1 - p), torch.zeros(())), 1e-6, 1e-7)
    assert abs(keep.float().mean().item() - (1 - p)) < 0.01
    y.sum().backward()
    close("dropout_grad", x.grad, keep.float() / (1 - p), 1e-6, 1e-7)
    assert GF.dropout(x, p, False) is x
    mlp = LinearChannelMLP([37, 64, 8], non_linearity=F.gelu, dropout=p).to(DEV).train()
    a, b = mlp(x.detach()), mlp(x.detach())
    assert not torch.equal(a, b)
    a.sum().backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in mlp.parameters())
    mlp.eval()
    assert torch.equal(mlp(x.detach()), mlp(x.detach()))


def test_reference_import_paths():
    from gaot_3d_amd.model.layers import magno
    from gaot_3d_amd.graph import apply_neighbor_sampling
    assert magno.apply_neighbor_sampling is apply_neighbor_sampling          # reference magno.py:297-371
    assert callable(magno.get_neighbor_strategy) and callable(magno.parse_neighbor_strategy)


def test_neighbour_list_cache_is_never_stale():
    """the per-batch CSR cache was keyed by the edge tensor's ADDRESS; with edges built inside forward the
    allocator hands the same address to the next, different edge list of equal shape.  Two forwards on ONE batch object
    with different query points of equal count must each equal a forward on a fresh batch."""
    import gaot_3d_amd
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model import init_model
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    gaot_3d_amd.set_precision("fp32")
    cfg = types.SimpleNamespace(
        magno=MAGNOConfig(gno_coord_dim=3, lifting_channels=32, encoder_feature_attr=["pos", "c"], mlp_type="linear",
                          use_geoembed=[True, False], neighbor_strategy="knn", k_neighbors=4, precompute_edges=False),
        transformer=TransformerConfig(patch_size=2, hidden_size=256, num_layers=2, positional_embedding="rope",
                                      attn_config=AttentionConfig(atten_dropout=0.0), ffn_config=FFNConfig()),
        latent_tokens=(8, 8, 4))
    torch.manual_seed(0)
    model = init_model(6, 1, "gaot_3d", cfg).to(DEV).eval()
    batch, tokens = make_synthetic_sample(3000, cfg.latent_tokens, k=4, seed=0, device=DEV)
    tokens = tokens.to(DEV)
    g = torch.Generator().manual_seed(9)
    q1 = (torch.rand(2000, 3, generator=g) * 2 - 1).to(DEV)
    q2 = (torch.rand(2000, 3, generator=g) * 2 - 1).to(DEV)
    qb = torch.zeros(2000, dtype=torch.long, device=DEV)
    with torch.no_grad():
        a1 = model(batch=batch, tokens_pos=tokens, query_coord_pos=q1, query_coord_batch_idx=qb).clone()
        a2 = model(batch=batch, tokens_pos=tokens, query_coord_pos=q2, query_coord_batch_idx=qb).clone()
        fresh, _ = make_synthetic_sample(3000, cfg.latent_tokens, k=4, seed=0, device=DEV)
        b2 = model(batch=fresh, tokens_pos=tokens, query_coord_pos=q2, query_coord_batch_idx=qb)
    assert not torch.equal(a1, a2)
    close("cache/second_query_set", a2, b2, 0.0, 0.0)
    # in-place edits of a batch-owned (precomputed) edge list are seen too: the entry records the version counter
    cfg.magno.precompute_edges = True
    torch.manual_seed(0)
    model2 = init_model(6, 1, "gaot_3d", cfg).to(DEV).eval()
    with torch.no_grad():
        o1 = model2(batch=batch, tokens_pos=tokens).clone()
        batch.decoder_edge_index_s0[0, :100] = batch.decoder_edge_index_s0[0, 100:200]     # same tensor, new contents
        batch.encoder_edge_index_s0[1, :100] = batch.encoder_edge_index_s0[1, 100:200]
        o2 = model2(batch=batch, tokens_pos=tokens).clone()
        gaot_3d_amd.clear_graph_cache(batch)
        o3 = model2(batch=batch, tokens_pos=tokens)
    assert not torch.equal(o1, o2)
    close("cache/in_place_edit", o2, o3, 0.0, 0.0)


@pytest.mark.parametrize("hidden,heads,kv_heads,rope,s", [(64, 4, 2, True, 77), (128, 2, 2, False, 130), (96, 2, 1, True, 33)])
def test_attention_any_head_dim_matches_oracle(hidden, heads, kv_heads, rope, s):
    """the reference accepts every hidden_size % num_heads == 0 (attn.py:66-67); head_dim 16 / 64 / 48 run the unfused
    general path (S x S scores per head in HBM, exact-fp32 GEMMs, row softmax) and equal the oracle's attention, forward
    and every gradient; with dropout the step runs and its masks differ from call to call"""
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.attn import GroupQueryFlashAttention
    gaot_3d_amd.set_precision("fp32")
    torch.manual_seed(hidden + s)
    att = GroupQueryFlashAttention(hidden, hidden, hidden_size=hidden, num_heads=heads, num_kv_heads=kv_heads,
                                   atten_dropout=0.0, positional_embedding="rope" if rope else "absolute")
    assert att.head_dim == hidden // heads and att.head_dim != 32
    sd = {"a." + k: v.detach().clone() for k, v in att.state_dict().items()}
    x = torch.randn(2, s, hidden)
    w = torch.randn(2, s, hidden)
    leaves = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "freqs" not in k) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ref = orc.attention(leaves, "a.", xr, heads, kv_heads, rope)
    (ref * w).sum().backward()
    att = att.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    out = att(xd, relative_positions=True if rope else None)
    (out * w.to(DEV)).sum().backward()
    tag = f"attn_general_hd{att.head_dim}"
    close(f"{tag}/out", out, ref, 1e-4, 1e-5)
    close(f"{tag}/dx", xd.grad, xr.grad, 1e-3, 1e-5)
    for k, p in att.named_parameters():
        if p.requires_grad:
            close(f"{tag}/grad/{k}", p.grad, leaves["a." + k].grad, 1e-3, 1e-5)
    att.atten_dropout = 0.2
    with torch.no_grad():
        a, b = att(xd, relative_positions=True if rope else None), att(xd, relative_positions=True if rope else None)
    assert torch.isfinite(a).all() and not torch.equal(a, b)
