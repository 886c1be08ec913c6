"""GPU parity of the general per-edge path (csrc/edgeops.hip + GEMM kernels): every IntegralTransform variant of the
reference (transform_type x attention, integral_transform.py:80-175) and the PointNet GeometricEmbedding
(geoembed.py:184-222) against the golden vectors captured from the reference's own modules (tests/golden/ops.npz:
variable-degree radius graph with empty rows and one row of degree > 32), plus a whole-model step against the oracle.
fp32: outputs rtol 1e-4 / atol 1e-5, gradients rtol 1e-3 / atol 1e-5."""
import os
import sys
import types

import pytest
import torch

import golden_io as gio

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import gaot_oracle as orc  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(name, a, b, rtol, atol):
    a, b = a.detach().cpu(), b.detach().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    print(f"[parity] {name}: max_abs={err:.3e} ref_peak={b.abs().max().item() if b.numel() else 0:.3e}")
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max abs err {err:.3e}"


VARIANTS = [(tt, attn) for tt in ("linear", "nonlinear", "nonlinear_kernelonly") for attn in (None, "cosine", "dot_product")]


@pytest.mark.parametrize("tt,attn", VARIANTS)
def test_integral_transform_variant_golden(tt, attn):
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    gaot_3d_amd.set_precision("fp32")
    meta, g = gio.load("ops")
    tag = f"it_{tt}_{attn or 'noattn'}"
    in_dim = 6 + (32 if tt != "linear" else 0)
    it = IntegralTransform(channel_mlp_layers=[in_dim, 64, 64, 32], transform_type=tt, use_attn=bool(attn), coord_dim=3,
                           attention_type=attn or "cosine")
    it.load_state_dict(gio.sub(g["sd"], tag), strict=True)
    it = it.to(DEV)
    pos, lat, ei = g["in"]["pos"].to(DEV), g["in"]["lat"].to(DEV), g["in"]["edge_index"].to(DEV)
    f = g["in"]["f_y"].to(DEV).requires_grad_(True)
    out = it(y_pos=pos, x_pos=lat, edge_index=ei, f_y=f)
    close(f"{tag}/out", out, g["out"][f"{tag}/out"], 1e-4, 1e-5)
    (out * g["in"][f"{tag}/w"].to(DEV)).sum().backward()
    close(f"{tag}/grad_f_y", f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-5)
    grads = gio.sub(g["grad"], tag)
    got = {k: p.grad for k, p in it.named_parameters() if p.grad is not None}
    assert set(got) == set(grads) - {"f_y"}, set(got) ^ (set(grads) - {"f_y"})
    for k, gr in grads.items():
        if k != "f_y":
            close(f"{tag}/grad/{k}", got[k], gr, 1e-3, 1e-5)


def test_integral_transform_empty_edge_list():
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    meta, g = gio.load("ops")
    it = IntegralTransform(channel_mlp_layers=[6, 64, 32]).to(DEV)
    out = it(y_pos=g["in"]["pos"].to(DEV), x_pos=g["in"]["lat"].to(DEV), edge_index=torch.zeros(2, 0, dtype=torch.long, device=DEV),
             f_y=g["in"]["f_y"].to(DEV))
    close("it_empty", out, g["out"]["it_empty"], 0, 0)


@pytest.mark.parametrize("pooling", ["max", "mean"])
def test_geoembed_pointnet_golden(pooling):
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.geoembed import GeometricEmbedding
    gaot_3d_amd.set_precision("fp32")
    meta, g = gio.load("ops")
    tag = f"geo_pointnet_{pooling}"
    ge = GeometricEmbedding(3, 32, method="pointnet", pooling=pooling)
    ge.load_state_dict(gio.sub(g["sd"], tag), strict=True)
    ge = ge.to(DEV)
    out = ge(g["in"]["pos"].to(DEV), g["in"]["lat"].to(DEV), g["in"]["edge_index"].to(DEV))
    close(f"{tag}/out", out, g["out"][f"{tag}/out"], 1e-4, 1e-5)
    (out * g["in"][f"{tag}/w"].to(DEV)).sum().backward()
    for k, gr in gio.sub(g["grad"], tag).items():
        close(f"{tag}/grad/{k}", dict(ge.named_parameters())[k].grad, gr, 1e-3, 1e-5)


def test_general_path_other_mlp_shapes_and_no_features():
    """kernel MLP shapes the fused kernel does not take (hidden 48, 3 layers, 16 output channels) and f_y = None"""
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    gaot_3d_amd.set_precision("fp32")
    meta, g = gio.load("ops")
    pos, lat, ei = g["in"]["pos"], g["in"]["lat"], g["in"]["edge_index"]
    torch.manual_seed(3)
    it = IntegralTransform(channel_mlp_layers=[6, 48, 40, 16])
    sd = {k: v.clone() for k, v in it.state_dict().items()}
    f = torch.randn(pos.shape[0], 16)
    w = torch.randn(lat.shape[0], 16)
    for feats in (f, None):
        fr = None if feats is None else feats.clone().requires_grad_(True)
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ref = orc.integral_transform(sdr, "", pos, lat, ei, fr)
        (ref * w).sum().backward()
        itd = IntegralTransform(channel_mlp_layers=[6, 48, 40, 16])
        itd.load_state_dict(sd)
        itd = itd.to(DEV)
        fd = None if feats is None else feats.to(DEV).requires_grad_(True)
        out = itd(y_pos=pos.to(DEV), x_pos=lat.to(DEV), edge_index=ei.to(DEV), f_y=fd)
        (out * w.to(DEV)).sum().backward()
        close("general/out", out, ref, 1e-4, 1e-5)
        if feats is not None:
            close("general/grad_f", fd.grad, fr.grad, 1e-3, 1e-5)
        for k, p in itd.named_parameters():
            close(f"general/grad/{k}", p.grad, sdr[k].grad, 1e-3, 1e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_model_with_nonlinear_attention_pointnet(precision):
    """whole model: encoder linear transform with dot-product attention weights (general path), decoder "nonlinear"
    with attention weights, PointNet GeoEmbed on both sides: one training step against the oracle.  (The reference
    sizes the encoder's kernel MLP for transform_type nonlinear* with the RAW input width, magno.py:403-405, but
    feeds it the lifted features, :541-552 -- that combination only runs when input_size == lifting_channels, so the
    encoder keeps 'linear' here.)"""
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model import init_model
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    cfg = types.SimpleNamespace(
        magno=MAGNOConfig(gno_coord_dim=3, lifting_channels=32, encoder_feature_attr="pos", mlp_type="linear",
                          use_geoembed=[True, True], embedding_method="pointnet", pooling="max",
                          in_gno_transform_type="linear", out_gno_transform_type="nonlinear",
                          use_attn=True, attention_type="dot_product", neighbor_strategy="knn", k_neighbors=6,
                          precompute_edges=True),
        transformer=TransformerConfig(patch_size=2, hidden_size=256, num_layers=2, positional_embedding="rope",
                                      attn_config=AttentionConfig(hidden_size=256, num_heads=8, num_kv_heads=8,
                                                                  atten_dropout=0.0),
                                      ffn_config=FFNConfig(hidden_size=1024)),
        latent_tokens=(8, 8, 8))
    torch.manual_seed(0)
    model = init_model(3, 2, "gaot_3d", cfg)
    batch, tokens = make_synthetic_sample(3000, cfg.latent_tokens, k=6, in_normals=False, surface=False, seed=1, out_channels=2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    pred_r, loss_r, grads_r = orc.train_step_grads(sd, cfg, batch, tokens)
    gaot_3d_amd.set_precision(precision)
    try:
        model = model.to(DEV).train()
        bd = batch.to(DEV)
        pred = model(batch=bd, tokens_pos=tokens.to(DEV))
        loss = GF.mse_loss(pred, bd.x)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        gaot_3d_amd.set_precision("fp32")
    if precision == "fp32":
        close("variants/pred", pred, pred_r, 1e-4, 2e-5)
        close("variants/loss", loss, loss_r, 1e-5, 1e-7)
        for k, p in model.named_parameters():
            if p.requires_grad:
                close(f"variants/grad/{k}", p.grad, grads_r[k], 1e-3, 1e-5)
    else:
        close("variants_bf16/pred", pred, pred_r, 2e-2, 2e-2)
        num = d1 = d2 = 0.0
        for k, p in model.named_parameters():
            if p.requires_grad:
                a, b = p.grad.cpu().double().flatten(), grads_r[k].double().flatten()
                num += (a * b).sum().item(); d1 += (a * a).sum().item(); d2 += (b * b).sum().item()
        cos = num / (d1 ** 0.5 * d2 ** 0.5)
        print(f"[parity] variants bf16 grad cosine = {cos:.6f}")
        assert cos >= 0.999


def test_model_with_neighbor_sampling():
    """MAGNOConfig.sampling_strategy = 'max_neighbors' (reference magno.py:529-538, 739-748): the model's step equals
    the oracle's step on the edge lists the sampler produces for the same seed words (encoder draw first, then the
    decoder's; the sampler itself is checked bit for bit in test_graph_gpu.py)."""
    import copy

    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.graph import apply_neighbor_sampling
    from gaot_3d_amd.model import init_model
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    cfg = types.SimpleNamespace(
        magno=MAGNOConfig(gno_coord_dim=3, lifting_channels=32, encoder_feature_attr="pos", mlp_type="linear",
                          use_geoembed=[True, False], neighbor_strategy="knn", k_neighbors=8, precompute_edges=True,
                          sampling_strategy="max_neighbors", max_neighbors=5),
        transformer=TransformerConfig(patch_size=2, hidden_size=256, num_layers=2, positional_embedding="rope",
                                      attn_config=AttentionConfig(hidden_size=256, num_heads=8, num_kv_heads=8,
                                                                  atten_dropout=0.0),
                                      ffn_config=FFNConfig(hidden_size=1024)),
        latent_tokens=(8, 8, 8))
    gaot_3d_amd.set_precision("fp32")
    torch.manual_seed(0)
    model = init_model(3, 1, "gaot_3d", cfg)
    batch, tokens = make_synthetic_sample(3000, cfg.latent_tokens, k=8, in_normals=False, surface=False, seed=2)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    seed0 = 424242
    words = GF.dropout_seed_sequence(seed0, 2)

    def dev_seed(v):
        return torch.tensor([v - (1 << 64) if v >= (1 << 63) else v], dtype=torch.int64, device=DEV)

    sampled = copy.copy(batch)
    sampled.encoder_edge_index_s0 = apply_neighbor_sampling(batch.encoder_edge_index_s0.to(DEV), 512, DEV, "max_neighbors",
                                                            max_neighbors=5, seed=dev_seed(words[0])).cpu()
    sampled.decoder_edge_index_s0 = apply_neighbor_sampling(batch.decoder_edge_index_s0.to(DEV), 3000, DEV, "max_neighbors",
                                                            max_neighbors=5, seed=dev_seed(words[1])).cpu()
    assert sampled.decoder_edge_index_s0.shape[1] == 3000 * 5
    ocfg = copy.deepcopy(cfg)
    ocfg.magno.sampling_strategy = None
    pred_r, loss_r, grads_r = orc.train_step_grads(sd, ocfg, sampled, tokens)
    model = model.to(DEV).train()
    bd = batch.to(DEV)
    GF.set_dropout_seed(seed0, DEV)
    pred = model(batch=bd, tokens_pos=tokens.to(DEV))
    loss = GF.mse_loss(pred, bd.x)
    loss.backward()
    close("sampling/pred", pred, pred_r, 1e-4, 2e-5)
    close("sampling/loss", loss, loss_r, 1e-5, 1e-7)
    for k, p in model.named_parameters():
        if p.requires_grad:
            close(f"sampling/grad/{k}", p.grad, grads_r[k], 1e-3, 1e-5)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_conditioned_norm_attention_and_ffn_golden(precision):
    """use_conditional_norm=True (reference mlp.py:74-128; attention input attn.py:101-102, FFN output :158-159) with
    one conditioning scalar per batch element, against the golden captured from the reference's modules"""
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.attn import FFN, GroupQueryFlashAttention
    meta, g = gio.load("cond_norm")
    d = meta["d"]
    c = g["in"]["c"].to(DEV)
    att = GroupQueryFlashAttention(d, d, hidden_size=d, num_heads=meta["heads"], num_kv_heads=meta["heads"],
                                   use_conditional_norm=True, cond_norm_hidden_size=4, atten_dropout=0.0,
                                   positional_embedding="absolute")
    ffn = FFN(d, d, hidden_size=meta["ffn_hidden"], use_conditional_norm=True, cond_norm_hidden_size=4)
    gaot_3d_amd.set_precision(precision)
    try:
        for tag, mod in (("attn", att), ("ffn", ffn)):
            sd = gio.sub(g["sd"], tag)
            assert list(mod.state_dict().keys()) == list(sd.keys())
            mod.load_state_dict(sd, strict=True)
            mod = mod.to(DEV).eval()
            x = g["in"]["x"].to(DEV).requires_grad_(True)
            out = mod(x, condition=c)
            (out * g["in"][f"{tag}/w"].to(DEV)).sum().backward()
            ref_o, ref_dx = g["out"][f"{tag}/out"], g["grad"][f"{tag}/x"]
            if precision == "fp32":
                close(f"cond/{tag}/out", out, ref_o, 1e-4, 1e-5 * ref_o.abs().max().item())
                close(f"cond/{tag}/dx", x.grad, ref_dx, 1e-3, 1e-5 * ref_dx.abs().max().item())
                for k, gr in gio.sub(g["grad"], tag).items():
                    if k != "x":
                        close(f"cond/{tag}/grad/{k}", dict(mod.named_parameters())[k].grad, gr, 1e-3, 1e-5 * gr.abs().max().item())
            else:
                close(f"cond_bf16/{tag}/out", out, ref_o, 2e-2, 2e-2 * ref_o.abs().max().item())
                a, b = x.grad.cpu().double().flatten(), ref_dx.double().flatten()
                assert float(a @ b / (a.norm() * b.norm())) >= 0.999
    finally:
        gaot_3d_amd.set_precision("fp32")
