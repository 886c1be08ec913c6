"""CPU: the oracle (oracle/gaot_oracle.py) against golden vectors captured from the reference.
This is what pins the oracle; rope cases are 'third-party, unpinned'."""
import sys

import pytest
import torch

import golden_io as gio

sys.path.insert(0, gio.os.path.join(gio.os.path.dirname(gio.GOLDEN_DIR), "..", "oracle"))
import gaot_oracle as orc  # noqa: E402

MODEL_CASES = ["model_knn_abs", "model_radius_rope", "model_channel_multiscale"]


def close(a, b, rtol=1e-4, atol=1e-5):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"max abs err {err:.3e}"


@pytest.mark.parametrize("case", MODEL_CASES)
def test_model_forward_backward(case):
    meta, g = gio.load(case)
    cfg = gio.ns_config(meta)
    batch = gio.batch_from(meta, g["in"])
    tokens = g["in"].get("tokens_pos")
    sd = g["sd"]
    nb = meta["num_graphs"]
    lat = (sd["latent_tokens"] if tokens is None else tokens).repeat(nb, 1)
    enc = orc.magno_encoder(sd, cfg.magno, batch, lat)
    close(enc, g["out"]["encoder"])
    proc = orc.process(sd, cfg.transformer, cfg.latent_tokens, g["out"]["encoder"])
    close(proc, g["out"]["processor"])
    pred, loss, grads = orc.train_step_grads(sd, cfg, batch, tokens)
    close(pred, g["out"]["pred"])
    close(loss, g["out"]["loss"], rtol=1e-5, atol=1e-7)
    assert set(grads.keys()) == set(g["grad"].keys()), set(grads.keys()) ^ set(g["grad"].keys())
    for k, gr in g["grad"].items():
        close(grads[k], gr, rtol=1e-3, atol=1e-6)


def test_scatter():
    meta, g = gio.load("ops")
    idx = g["in"]["edge_index"][1].long()
    for red in ("sum", "mean", "max", "min"):
        close(orc.scatter(g["in"]["scatter_src"], idx, meta["nq"], red), g["out"][f"scatter_{red}"], 1e-6, 1e-6)


def test_integral_transform_variants():
    meta, g = gio.load("ops")
    pos, lat, ei = g["in"]["pos"], g["in"]["lat"], g["in"]["edge_index"]
    for v in meta["variants"]:
        tag = v["tag"]
        sd = {k: t.clone().requires_grad_(True) for k, t in gio.sub(g["sd"], tag).items()}
        f = g["in"]["f_y"].clone().requires_grad_(True)
        out = orc.integral_transform(sd, "", pos, lat, ei, f, v["transform_type"], bool(v["attn"]), 3,
                                     v["attn"] or "cosine")
        close(out, g["out"][f"{tag}/out"])
        (out * g["in"][f"{tag}/w"]).sum().backward()
        close(f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-6)
        for k, gr in gio.sub(g["grad"], tag).items():
            if k != "f_y":
                close(sd[k].grad, gr, 1e-3, 1e-5)
    sd = {"channel_mlp.fcs.0.weight": torch.zeros(64, 6), "channel_mlp.fcs.0.bias": torch.zeros(64),
          "channel_mlp.fcs.1.weight": torch.zeros(32, 64), "channel_mlp.fcs.1.bias": torch.zeros(32)}
    out = orc.integral_transform(sd, "", pos, lat, torch.zeros(2, 0, dtype=torch.long), g["in"]["f_y"])
    close(out, g["out"]["it_empty"])


@pytest.mark.parametrize("name", ["gno_shapes", "gno_hidden"])
def test_integral_transform_other_shapes(name):
    """the reference's default shapes (lifting_channels 16, gno_coord_dim 2: magno.py:25,28), 64 channels, four hidden
    layers, coord dim 1 (gno_shapes); kernel-MLP hidden widths 8 / 32 / 48-64-16 / 128 (gno_hidden: free lists, magno.py:32,36)"""
    meta, g = gio.load(name)
    ei = g["in"]["edge_index"]
    for v in meta["variants"]:
        tag, cd = v["tag"], v["coord_dim"]
        y, x = g["in"]["pos3"][:, :cd].contiguous(), g["in"]["lat3"][:, :cd].contiguous()
        sd = {k: t.clone().requires_grad_(True) for k, t in gio.sub(g["sd"], tag).items()}
        f = g["in"][f"{tag}/f_y"].clone().requires_grad_(True)
        out = orc.integral_transform(sd, "", y, x, ei, f, "linear", None, cd)
        close(out, g["out"][f"{tag}/out"])
        (out * g["in"][f"{tag}/w"]).sum().backward()
        close(f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-6)
        for k, gr in gio.sub(g["grad"], tag).items():
            if k != "f_y":
                close(sd[k].grad, gr, 1e-3, 1e-5)


def test_integral_transform_attention_low_dim_and_activations():
    """attention weights on coordinates of dimension 2 / 1 and kernel MLPs with the other activations the reference's
    `activation_fn` builds (tests/golden/gno_variants.npz, captured from the reference's IntegralTransform)"""
    meta, g = gio.load("gno_variants")
    ei = g["in"]["edge_index"]
    for v in meta["variants"]:
        tag, cd = v["tag"], v["coord_dim"]
        y, x = g["in"]["pos3"][:, :cd].contiguous(), g["in"]["lat3"][:, :cd].contiguous()
        sd = {k: t.clone().requires_grad_(True) for k, t in gio.sub(g["sd"], tag).items()}
        f = g["in"][f"{tag}/f_y"].clone().requires_grad_(True)
        out = orc.integral_transform(sd, "", y, x, ei, f, v["transform_type"], v["use_attn"] or None, cd, v["attention_type"],
                                     act=orc.activation_fn(v["act"]))
        close(out, g["out"][f"{tag}/out"])
        (out * g["in"][f"{tag}/w"]).sum().backward()
        close(f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-6)
        for k, gr in gio.sub(g["grad"], tag).items():
            if k != "f_y":
                close(sd[k].grad, gr, 1e-3, 1e-5)


def test_geoembed_variants():
    meta, g = gio.load("ops")
    pos, lat, ei = g["in"]["pos"], g["in"]["lat"], g["in"]["edge_index"]
    close(orc.geoembed_stat_features(pos, lat, ei), g["out"]["geo_stat_features"], 1e-4, 1e-5)
    for method, pooling in (("statistical", "max"), ("pointnet", "max"), ("pointnet", "mean")):
        tag = f"geo_{method}_{pooling}"
        sd = {k: t.clone().requires_grad_(True) for k, t in gio.sub(g["sd"], tag).items()}
        out = orc.geoembed(sd, "", pos, lat, ei, method, pooling)
        close(out, g["out"][f"{tag}/out"])
        (out * g["in"][f"{tag}/w"]).sum().backward()
        for k, gr in gio.sub(g["grad"], tag).items():
            close(sd[k].grad, gr, 1e-3, 1e-5)


def test_attention_training_dropout_given_mask():
    """reference attention in training mode (atten_dropout 0.1): the oracle's masked SDPA reproduces the output and
    every gradient once it is handed the keep mask torch drew (replayed by make_goldens.attn_dropout_case)"""
    meta, g = gio.load("attn_dropout")
    sd = {k: t.clone().requires_grad_(True) for k, t in g["sd"].items()}
    x = g["in"]["x"].clone().requires_grad_(True)
    keep = g["in"]["keep"].bool()
    assert 0.8 < keep.float().mean().item() < 0.97
    out = orc.attention(sd, "", x, meta["h"], meta["hkv"], False, keep, meta["p"])
    close(out, g["out"]["out"])
    (out * g["in"]["w"]).sum().backward()
    close(x.grad, g["grad"]["x"], 1e-3, 1e-6)
    for k, gr in g["grad"].items():
        if k != "x":
            close(sd[k].grad, gr, 1e-3, 1e-6)


def test_dropout_mask_function_statistics():
    """the product's counter-based mask (restated in numpy): keep rate, per-row balance, seed sensitivity"""
    p = 0.1
    m1 = orc.dropout_keep_mask(0x0123456789ABCDEF, 1, 4, 512, p).float()
    m2 = orc.dropout_keep_mask(0x0123456789ABCDF0, 1, 4, 512, p).float()
    assert abs(m1.mean().item() - (1 - orc.dropout_threshold(p) / 65536)) < 2e-3
    assert (m1.mean(dim=-1) - 0.9).abs().max().item() < 0.08          # every query row
    assert (m1.mean(dim=-2) - 0.9).abs().max().item() < 0.08          # every key column
    agree = (m1 == m2).float().mean().item()                            # independent masks agree w.p. 0.82
    assert abs(agree - (0.9 * 0.9 + 0.1 * 0.1)) < 5e-3
    # neighbouring elements are uncorrelated (the two keys of a pair share one 32-bit word)
    a, b2 = m1[..., 0::2].flatten(), m1[..., 1::2].flatten()
    corr = ((a - a.mean()) * (b2 - b2.mean())).mean() / (a.std() * b2.std())
    assert abs(corr.item()) < 5e-3


def test_conditioned_norm_in_attention_and_ffn():
    """use_conditional_norm (reference mlp.py:74-128, attn.py:101-102, 158-159) against the golden captured from the
    reference's attention and FFN modules"""
    meta, g = gio.load("cond_norm")
    c = g["in"]["c"]
    for tag in ("attn", "ffn"):
        sd = {k: t.clone().requires_grad_(True) for k, t in gio.sub(g["sd"], tag).items()}
        x = g["in"]["x"].clone().requires_grad_(True)
        out = orc.attention(sd, "", x, meta["heads"], meta["heads"], False, condition=c) if tag == "attn" \
            else orc.ffn(sd, "", x, condition=c)
        close(out, g["out"][f"{tag}/out"])
        (out * g["in"][f"{tag}/w"]).sum().backward()
        close(x.grad, g["grad"][f"{tag}/x"], 1e-3, 1e-5 * g["grad"][f"{tag}/x"].abs().max().item())
        for k, gr in gio.sub(g["grad"], tag).items():
            if k != "x":
                close(sd[k].grad, gr, 1e-3, 1e-5 * gr.abs().max().item())   # sums of 40 rows of O(10) terms
