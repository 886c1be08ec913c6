"""
Golden-vector generator (TEST INFRASTRUCTURE; runs only in the authoring container).

Imports the *reference's own* model code from /root/reference (read-only) with three stub
modules for packages that are absent from this image and records inputs,
parameters, outputs and gradients at tiny sizes into ``tests/golden/*.npz``.  Nothing from
/root/reference is copied: the fixtures are data only.

Stubs:
  * ``omegaconf``            -- names only (never called on the model path).
  * ``torch_geometric``      -- names only (edges are precomputed inputs; no sampling).
  * ``rotary_embedding_torch`` -- a *working* restatement of RotaryEmbedding(dim)
        .rotate_queries_or_keys (third-party, version unpinned: "rope: third-party,
        unpinned").  The reference's SDPA / eigvalsh / GEMM run through the real torch.

Usage:  python oracle/make_goldens.py     (needs /root/reference)
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GAOT_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

from gaot_3d_amd.data import (MeshBatch, coalesce_edges, knn_edges_bruteforce, latent_grid,  # noqa: E402
                              radius_edges_bruteforce)


def install_stubs():
    om = types.ModuleType("omegaconf")

    class OmegaConf:  # noqa
        pass

    class DictConfig:  # noqa
        pass

    om.OmegaConf, om.DictConfig = OmegaConf, DictConfig
    sys.modules["omegaconf"] = om

    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")
    tgn = types.ModuleType("torch_geometric.nn")
    tgu = types.ModuleType("torch_geometric.utils")

    class Batch:  # noqa
        pass

    def _absent(*a, **k):
        raise RuntimeError("torch_geometric stub: graph construction is not available here")

    tgd.Batch = Batch
    tgn.knn = _absent
    tgn.radius = _absent
    tgu.coalesce = _absent
    tgu.dropout_edge = _absent
    tg.data, tg.nn, tg.utils = tgd, tgn, tgu
    sys.modules.update({"torch_geometric": tg, "torch_geometric.data": tgd,
                        "torch_geometric.nn": tgn, "torch_geometric.utils": tgu})

    rot = types.ModuleType("rotary_embedding_torch")

    def rotate_half(x):
        x = x.reshape(*x.shape[:-1], -1, 2)
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)

    def apply_rotary_emb(freqs, t, start_index=0, scale=1.0, seq_dim=-2):
        rot_dim = freqs.shape[-1]
        end = start_index + rot_dim
        tl, tm, tr = t[..., :start_index], t[..., start_index:end], t[..., end:]
        tm = (tm * freqs.cos() * scale) + (rotate_half(tm) * freqs.sin() * scale)
        return torch.cat((tl, tm, tr), dim=-1)

    class RotaryEmbedding(nn.Module):
        def __init__(self, dim, theta=10000):
            super().__init__()
            freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
            self.freqs = nn.Parameter(freqs, requires_grad=False)

        def forward(self, t):
            f = torch.einsum("..., f -> ... f", t.type(self.freqs.dtype), self.freqs)
            return f.repeat_interleave(2, dim=-1)

        def rotate_queries_or_keys(self, t, seq_dim=-2, offset=0):
            seq_len = t.shape[seq_dim]
            pos = torch.arange(seq_len, device=t.device, dtype=self.freqs.dtype) + offset
            return apply_rotary_emb(self.forward(pos), t, seq_dim=seq_dim)

    rot.RotaryEmbedding, rot.apply_rotary_emb = RotaryEmbedding, apply_rotary_emb
    sys.modules["rotary_embedding_torch"] = rot


def save(name, meta, arrays):
    os.makedirs(OUT, exist_ok=True)
    flat = {"__meta__": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    for k, v in arrays.items():
        flat[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **flat)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def grads_of(module):
    return {k: p.grad for k, p in module.named_parameters() if p.grad is not None}


# ------------------------------------------------------------------------------------------
def variable_degree_graph(phys, lat, radius, heavy_token, n_heavy, gen):
    """radius graph with empty latent rows and one row of degree > 32 (no PyG cap)."""
    phys = phys.clone()
    phys[:n_heavy] = lat[heavy_token] + 0.25 * radius * (torch.rand(n_heavy, 3, generator=gen) - 0.5)
    enc = radius_edges_bruteforce(phys, lat, radius, max_num_neighbors=None, centers="latent")
    return phys, enc


def model_case(name, seed, n_per_graph, latent_tokens, magno_kw, tr_kw, attn_kw, ffn_kw, graph, out_ch,
               feats, token_box=None):
    from src.model import init_model
    from src.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from src.model.layers.magno import MAGNOConfig

    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    lo, hi = token_box if token_box is not None else ((-1.0,) * 3, (1.0,) * 3)
    lat = latent_grid(latent_tokens, lo, hi)
    m = lat.shape[0]
    nscales = len(magno_kw.get("scales", [1.0]))
    samples = []
    for n in n_per_graph:
        pos = torch.rand(n, 3, generator=gen) * 2 - 1
        s = MeshBatch()
        for si in range(nscales):
            if graph == "knn":
                enc = knn_edges_bruteforce(pos, lat, 4 + si)
                dec = enc.flip(0)
            else:
                pos, enc = variable_degree_graph(pos, lat, 0.45 * (1 + si), heavy_token=5, n_heavy=40, gen=gen)
                knn = knn_edges_bruteforce(pos, lat, 3).flip(0)                    # [latent, phys]
                rad = radius_edges_bruteforce(pos, lat, 0.5, None, centers="phys")  # [latent, phys]
                dec = coalesce_edges(torch.cat([knn, rad], dim=1), n)
            setattr(s, f"encoder_edge_index_s{si}", enc.to(torch.int32))
            setattr(s, f"decoder_edge_index_s{si}", dec.to(torch.int32))
        s.pos = pos
        s.x = torch.randn(n, out_ch, generator=gen)
        for a, w in feats.items():
            if a != "pos":
                setattr(s, a, torch.randn(n, w, generator=gen))
        samples.append(s)
    batch = MeshBatch.from_data_list(samples, m)
    in_size = sum(feats.values())
    attr = list(feats.keys()) if len(feats) > 1 else list(feats.keys())[0]

    mc = MAGNOConfig(gno_coord_dim=3, lifting_channels=32, encoder_feature_attr=attr, precompute_edges=True,
                     **magno_kw)
    tc = TransformerConfig(attn_config=AttentionConfig(atten_dropout=0.0, **attn_kw),
                           ffn_config=FFNConfig(**ffn_kw), **tr_kw)
    cfg = types.SimpleNamespace(magno=mc, transformer=tc, latent_tokens=tuple(latent_tokens))
    model = init_model(in_size, out_ch, "gaot_3d", cfg)
    model.train()
    # make biases / norm weights non-trivial so that a dropped bias shows up
    with torch.no_grad():
        for k, p in model.named_parameters():
            if p.requires_grad and (k.endswith("norm.weight") or k.endswith(".bias")):
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    tokens = lat if token_box is not None else None
    enc_out = model.encoder(batch=batch, latent_tokens_pos=lat.repeat(len(samples), 1),
                            latent_tokens_batch_idx=torch.arange(len(samples)).repeat_interleave(m))
    proc_out = model.process(enc_out)
    pred = model(batch=batch, tokens_pos=tokens) if tokens is not None else model(batch)
    loss = nn.MSELoss()(pred, batch.x)
    loss.backward()
    arrays = {}
    for k, v in model.state_dict().items():
        arrays["sd/" + k] = v
    for k in batch.keys():
        v = getattr(batch, k)
        if torch.is_tensor(v):
            arrays["in/" + k] = v
    if tokens is not None:
        arrays["in/tokens_pos"] = tokens
    arrays["out/encoder"] = enc_out
    arrays["out/processor"] = proc_out
    arrays["out/pred"] = pred
    arrays["out/loss"] = loss
    for k, g in grads_of(model).items():
        arrays["grad/" + k] = g
    meta = dict(name=name, seed=seed, num_graphs=len(samples), latent_tokens=list(latent_tokens),
                in_size=in_size, out_size=out_ch, feats=feats,
                magno=dict(gno_coord_dim=3, lifting_channels=32, encoder_feature_attr=attr,
                           precompute_edges=True, **magno_kw),
                transformer=dict(tr_kw), attn=dict(atten_dropout=0.0, **attn_kw), ffn=dict(ffn_kw),
                rope_note="rope: third-party, unpinned" if tr_kw.get("positional_embedding") == "rope" else "",
                nparams=sum(p.numel() for p in model.parameters()))
    save(name, meta, arrays)


def ops_case():
    """Operator-level goldens: scatter, IntegralTransform variants, GeoEmbed variants."""
    from src.model.layers.geoembed import GeometricEmbedding
    from src.model.layers.integral_transform import IntegralTransform
    from src.model.layers.utils.scatter_native import scatter_native

    gen = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    lat = latent_grid((4, 4, 3))
    pos = torch.rand(260, 3, generator=gen) * 2 - 1
    pos, enc = variable_degree_graph(pos, lat, 0.45, heavy_token=9, n_heavy=45, gen=gen)
    arrays = {"in/pos": pos, "in/lat": lat, "in/edge_index": enc.to(torch.int32)}
    nq = lat.shape[0]
    # scatter
    src = torch.randn(enc.shape[1], 5, generator=gen)
    arrays["in/scatter_src"] = src
    for red in ("sum", "mean", "max", "min"):
        arrays[f"out/scatter_{red}"] = scatter_native(src, enc[1].long(), dim=0, dim_size=nq, reduce=red)
    # integral transform variants
    f_y = torch.randn(pos.shape[0], 32, generator=gen)
    arrays["in/f_y"] = f_y
    variants = []
    for tt in ("linear", "nonlinear", "nonlinear_kernelonly"):
        for attn in (None, "cosine", "dot_product"):
            tag = f"it_{tt}_{attn or 'noattn'}"
            in_dim = 6 + (32 if tt != "linear" else 0)
            it = IntegralTransform(channel_mlp_layers=[in_dim, 64, 64, 32], transform_type=tt,
                                   use_attn=bool(attn), coord_dim=3, attention_type=attn or "cosine")
            with torch.no_grad():
                for p in it.parameters():
                    if p.dim() == 1:
                        p.add_(0.1 * torch.randn(p.shape, generator=gen))
            f = f_y.clone().requires_grad_(True)
            out = it(y_pos=pos, x_pos=lat, edge_index=enc, f_y=f)
            w = torch.randn(out.shape, generator=gen)
            (out * w).sum().backward()
            arrays[f"in/{tag}/w"] = w
            arrays[f"out/{tag}/out"] = out
            arrays[f"grad/{tag}/f_y"] = f.grad
            for k, v in it.state_dict().items():
                arrays[f"sd/{tag}/{k}"] = v
            for k, g in grads_of(it).items():
                arrays[f"grad/{tag}/{k}"] = g
            variants.append(dict(tag=tag, transform_type=tt, attn=attn))
    # empty edge list (integral_transform.py:106-112)
    it = IntegralTransform(channel_mlp_layers=[6, 64, 32])
    arrays["out/it_empty"] = it(y_pos=pos, x_pos=lat, edge_index=torch.zeros(2, 0, dtype=torch.long), f_y=f_y)
    # geoembed
    for method, pooling in (("statistical", "max"), ("pointnet", "max"), ("pointnet", "mean")):
        tag = f"geo_{method}_{pooling}"
        ge = GeometricEmbedding(3, 32, method=method, pooling=pooling)
        out = ge(pos, lat, enc)
        w = torch.randn(out.shape, generator=gen)
        (out * w).sum().backward()
        arrays[f"in/{tag}/w"] = w
        arrays[f"out/{tag}/out"] = out
        for k, v in ge.state_dict().items():
            arrays[f"sd/{tag}/{k}"] = v
        for k, g in grads_of(ge).items():
            arrays[f"grad/{tag}/{k}"] = g
        if method == "statistical":
            arrays["out/geo_stat_features"] = ge._compute_statistical_features_pyg(pos, lat, enc)
    save("ops", dict(name="ops", variants=variants, nq=nq), arrays)


def attn_dropout_case():
    """GroupQueryFlashAttention in TRAINING mode with atten_dropout = 0.1 (attn.py:122-127).  torch's CPU SDPA takes its
    math path for dropout_p > 0 and draws the keep mask as empty_like(attn).bernoulli_(1 - p) from the default
    generator; the same draw is replayed here (same seed, nothing drawn in between) and stored, so the golden pins
    the arithmetic GIVEN the mask: P * keep / (1 - p) @ V, and its autograd."""
    from src.model.layers.attn import GroupQueryFlashAttention

    b, s, d, h, hkv, p = 2, 24, 64, 2, 1, 0.1
    torch.manual_seed(5)
    att = GroupQueryFlashAttention(d, d, hidden_size=d, num_heads=h, num_kv_heads=hkv, atten_dropout=p,
                                   positional_embedding="absolute")
    att.train()
    x = torch.randn(b, s, d)
    w = torch.randn(b, s, d)
    xin = x.clone().requires_grad_(True)
    torch.manual_seed(11)
    out = att(xin)
    (out * w).sum().backward()
    torch.manual_seed(11)
    keep = torch.empty(b, h, s, s).bernoulli_(1 - p)
    arrays = {"in/x": x, "in/w": w, "in/keep": keep.to(torch.uint8), "out/out": out, "grad/x": xin.grad}
    for k, v in att.state_dict().items():
        arrays[f"sd/{k}"] = v
    for k, g in grads_of(att).items():
        arrays[f"grad/{k}"] = g
    save("attn_dropout", dict(name="attn_dropout", b=b, s=s, d=d, h=h, hkv=hkv, p=p), arrays)


def cond_norm_case():
    """time-conditioned norm (mlp.py:74-128) inside GroupQueryFlashAttention and FFN (use_conditional_norm=True,
    attn.py:81-84, 101-102, 150-159) in eval mode, condition = one scalar per batch element"""
    from src.model.layers.attn import FFN, GroupQueryFlashAttention

    b, s, d = 2, 20, 64
    torch.manual_seed(9)
    att = GroupQueryFlashAttention(d, d, hidden_size=d, num_heads=2, num_kv_heads=2, use_conditional_norm=True,
                                   cond_norm_hidden_size=4, atten_dropout=0.0, positional_embedding="absolute").eval()
    ffn = FFN(d, d, hidden_size=96, use_conditional_norm=True, cond_norm_hidden_size=4).eval()
    with torch.no_grad():   # the reference initialises these weights with std 0.01: make the correction visible
        for m in (att.correction, ffn.correction):
            for p in m.parameters():
                p.mul_(10.0).add_(0.05 * torch.randn(p.shape))
    x = 0.5 * torch.randn(b, s, d)
    c = torch.tensor([[0.3], [-1.2]])
    arrays = {"in/x": x, "in/c": c}
    for tag, mod in (("attn", att), ("ffn", ffn)):
        xin = x.clone().requires_grad_(True)
        out = mod(xin, condition=c)
        w = torch.randn(out.shape)
        (out * w).sum().backward()
        arrays[f"in/{tag}/w"] = w
        arrays[f"out/{tag}/out"] = out
        arrays[f"grad/{tag}/x"] = xin.grad
        for k, v in mod.state_dict().items():
            arrays[f"sd/{tag}/{k}"] = v
        for k, g in grads_of(mod).items():
            arrays[f"grad/{tag}/{k}"] = g
    save("cond_norm", dict(name="cond_norm", b=b, s=s, d=d, heads=2, ffn_hidden=96), arrays)


def lr_schedule_case():
    """the trainer's 'mix' learning-rate schedule (src/trainer/optimizers.py:40-67, 226-246): per-epoch values of the
    reference's own CustomLRScheduler driven through AdamWOptimizer's phase split, for a few epoch counts"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_optimizers", os.path.join(REF, "src", "trainer", "optimizers.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except Exception as ex:   # relative imports of the trainer package: fall back to the class source only
        raise RuntimeError(f"cannot import the reference scheduler: {ex}")
    arrays, cases = {}, []
    for total in (1, 2, 7, 50, 100, 333):
        cfg = mod.OptimizerargsConfig(epoch=total, lr=3e-4, scheduler="mix", max_lr=1e-2, min_lr=1e-5, final_lr=2e-6)
        opt = mod.AdamWOptimizer([torch.nn.Parameter(torch.zeros(1))], cfg)
        lrs = []
        for _ in range(total + 2):
            lrs.append(opt.optimizer.param_groups[0]["lr"])
            opt.optimizer.step()
            opt.scheduler.step()
        arrays[f"out/lr_{total}"] = np.asarray(lrs, dtype=np.float64)
        cases.append(total)
    save("lr_mix", dict(name="lr_mix", totals=cases, lr=3e-4, max_lr=1e-2, min_lr=1e-5, final_lr=2e-6), arrays)


def gno_shapes_case():
    """IntegralTransform at the shapes the reference's own defaults produce (MAGNOConfig: lifting_channels 16,
    gno_coord_dim 2, magno.py:25,28) and at 64 channels / four hidden layers: linear transform, mean reduction, on a
    variable-degree graph with empty rows and one heavy row"""
    from src.model.layers.integral_transform import IntegralTransform

    gen = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    lat3 = latent_grid((4, 4, 3))
    pos3 = torch.rand(230, 3, generator=gen) * 2 - 1
    pos3, enc = variable_degree_graph(pos3, lat3, 0.45, heavy_token=7, n_heavy=41, gen=gen)
    arrays = {"in/pos3": pos3, "in/lat3": lat3, "in/edge_index": enc.to(torch.int32)}
    variants = []
    for tag, cd, layers in (("c16_cd2", 2, [4, 64, 64, 64, 16]), ("c64_cd3", 3, [6, 64, 64, 64]),
                            ("c32_cd3_nh4", 3, [6, 64, 64, 64, 64, 32]), ("c48_cd1", 1, [2, 64, 48])):
        it = IntegralTransform(channel_mlp_layers=layers, transform_type="linear", coord_dim=cd)
        with torch.no_grad():
            for p in it.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
        y, x = pos3[:, :cd].contiguous(), lat3[:, :cd].contiguous()
        f = torch.randn(pos3.shape[0], layers[-1], generator=gen).requires_grad_(True)
        out = it(y_pos=y, x_pos=x, edge_index=enc, f_y=f)
        w = torch.randn(out.shape, generator=gen)
        (out * w).sum().backward()
        arrays[f"in/{tag}/f_y"] = f.detach()
        arrays[f"in/{tag}/w"] = w
        arrays[f"out/{tag}/out"] = out
        arrays[f"grad/{tag}/f_y"] = f.grad
        for k, v in it.state_dict().items():
            arrays[f"sd/{tag}/{k}"] = v
        for k, g in grads_of(it).items():
            arrays[f"grad/{tag}/{k}"] = g
        variants.append(dict(tag=tag, coord_dim=cd, layers=layers))
    save("gno_shapes", dict(name="gno_shapes", variants=variants), arrays)


def gno_hidden_case():
    """IntegralTransform with kernel-MLP hidden widths other than 64: `in_gno_channel_mlp_hidden_layers` /
    `out_gno_channel_mlp_hidden_layers` are free lists (magno.py:32,36).  Widths <= 64 run the fused kernels through exact zero
    padding, 128 the general per-edge path; same graph family as gno_shapes"""
    from src.model.layers.integral_transform import IntegralTransform

    gen = torch.Generator().manual_seed(31)
    torch.manual_seed(31)
    lat3 = latent_grid((4, 4, 3))
    pos3 = torch.rand(220, 3, generator=gen) * 2 - 1
    pos3, enc = variable_degree_graph(pos3, lat3, 0.45, heavy_token=9, n_heavy=39, gen=gen)
    arrays = {"in/pos3": pos3, "in/lat3": lat3, "in/edge_index": enc.to(torch.int32)}
    variants = []
    for tag, cd, layers in (("h32_cd3", 3, [6, 32, 32, 32]), ("h48_64_16_cd2", 2, [4, 48, 64, 16, 16]),
                            ("h8_cd3_c40", 3, [6, 8, 40]), ("h128_cd3", 3, [6, 128, 32])):
        it = IntegralTransform(channel_mlp_layers=layers, transform_type="linear", coord_dim=cd)
        with torch.no_grad():
            for p in it.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
        y, x = pos3[:, :cd].contiguous(), lat3[:, :cd].contiguous()
        f = torch.randn(pos3.shape[0], layers[-1], generator=gen).requires_grad_(True)
        out = it(y_pos=y, x_pos=x, edge_index=enc, f_y=f)
        w = torch.randn(out.shape, generator=gen)
        (out * w).sum().backward()
        arrays[f"in/{tag}/f_y"] = f.detach()
        arrays[f"in/{tag}/w"] = w
        arrays[f"out/{tag}/out"] = out
        arrays[f"grad/{tag}/f_y"] = f.grad
        for k, v in it.state_dict().items():
            arrays[f"sd/{tag}/{k}"] = v
        for k, g in grads_of(it).items():
            arrays[f"grad/{tag}/{k}"] = g
        variants.append(dict(tag=tag, coord_dim=cd, layers=layers))
    save("gno_hidden", dict(name="gno_hidden", variants=variants), arrays)


def gno_variants_case():
    """IntegralTransform variants the reference's config surface reaches beyond the shipped yaml: segment-softmax attention
    weights on coordinates of dimension 2 / 1 (`gno_coord_dim: 2` is the reference's default, magno.py:28; scores slice
    `[:, :coord_dim]`, integral_transform.py:126-142) and kernel MLPs built with other activations through the reference's own
    `activation_fn(name)` (mlp.py:27-35: "swish", "none", any F.<name>)"""
    from src.model.layers.integral_transform import IntegralTransform
    from src.model.layers.mlp import activation_fn

    gen = torch.Generator().manual_seed(23)
    torch.manual_seed(23)
    lat3 = latent_grid((4, 4, 3))
    pos3 = torch.rand(210, 3, generator=gen) * 2 - 1
    pos3, enc = variable_degree_graph(pos3, lat3, 0.45, heavy_token=5, n_heavy=37, gen=gen)
    arrays = {"in/pos3": pos3, "in/lat3": lat3, "in/edge_index": enc.to(torch.int32)}
    variants = []
    cases = [("attn_cos_cd2", 2, [4, 64, 64, 32], "linear", True, "cosine", "gelu"),
             ("attn_dot_cd2", 2, [4, 64, 64, 16], "linear", True, "dot_product", "gelu"),
             ("attn_cos_cd1", 1, [2, 64, 32], "linear", True, "cosine", "gelu"),
             ("attn_dot_cd2_nonlinear", 2, [4 + 32, 64, 32], "nonlinear", True, "dot_product", "gelu"),
             ("act_tanh", 3, [6, 64, 64, 32], "linear", None, "cosine", "tanh"),
             ("act_leaky_relu", 3, [6, 64, 32], "linear", None, "cosine", "leaky_relu"),
             ("act_elu_cd2", 2, [4, 64, 64, 32], "linear", None, "cosine", "elu"),
             ("act_swish", 3, [6, 64, 32], "linear", None, "cosine", "swish"),
             ("act_none", 3, [6, 64, 32], "linear", None, "cosine", "none"),
             ("act_softplus_attn", 3, [6, 64, 32], "linear", True, "cosine", "softplus"),
             ("act_sigmoid", 3, [6, 48, 32], "linear", None, "cosine", "sigmoid"),
             ("act_selu", 3, [6, 64, 32], "linear", None, "cosine", "selu"),
             ("act_mish", 3, [6, 64, 32], "linear", None, "cosine", "mish"),
             ("act_hardswish", 3, [6, 64, 32], "linear", None, "cosine", "hardswish"),
             ("act_relu6", 3, [6, 64, 32], "linear", None, "cosine", "relu6")]
    for tag, cd, layers, tt, use_attn, atype, act in cases:
        it = IntegralTransform(channel_mlp_layers=layers, channel_mlp_non_linearity=activation_fn(act), transform_type=tt,
                               use_attn=use_attn, coord_dim=cd, attention_type=atype)
        with torch.no_grad():
            for p in it.parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
        y, x = pos3[:, :cd].contiguous(), lat3[:, :cd].contiguous()
        f = torch.randn(pos3.shape[0], 32 if tt == "nonlinear" else layers[-1], generator=gen).requires_grad_(True)
        out = it(y_pos=y, x_pos=x, edge_index=enc, f_y=f)
        w = torch.randn(out.shape, generator=gen)
        (out * w).sum().backward()
        arrays[f"in/{tag}/f_y"] = f.detach()
        arrays[f"in/{tag}/w"] = w
        arrays[f"out/{tag}/out"] = out
        arrays[f"grad/{tag}/f_y"] = f.grad
        for k, v in it.state_dict().items():
            arrays[f"sd/{tag}/{k}"] = v
        for k, g in grads_of(it).items():
            arrays[f"grad/{tag}/{k}"] = g
        variants.append(dict(tag=tag, coord_dim=cd, layers=layers, transform_type=tt, use_attn=bool(use_attn),
                             attention_type=atype, act=act))
    save("gno_variants", dict(name="gno_variants", variants=variants), arrays)


def main():
    assert os.path.isdir(REF), f"{REF} not present: goldens can only be regenerated in the authoring container"
    install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "attn_dropout":   # regenerate just this file
        attn_dropout_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cond_norm":
        cond_norm_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "lr_mix":
        lr_schedule_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gno_hidden":
        gno_hidden_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gno_shapes":
        gno_shapes_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gno_variants":
        gno_variants_case()
        return
    ops_case()
    gno_shapes_case()
    gno_hidden_case()
    gno_variants_case()
    attn_dropout_case()
    lr_schedule_case()
    cond_norm_case()
    model_case("model_knn_abs", seed=1, n_per_graph=[200], latent_tokens=(4, 4, 4),
               magno_kw=dict(use_geoembed=[True, False], mlp_type="linear", neighbor_strategy="knn", k_neighbors=4),
               tr_kw=dict(patch_size=2, hidden_size=64, num_layers=2, positional_embedding="absolute"),
               attn_kw=dict(hidden_size=64, num_heads=2, num_kv_heads=2), ffn_kw=dict(hidden_size=128),
               graph="knn", out_ch=1, feats={"pos": 3, "c": 3})
    model_case("model_radius_rope", seed=2, n_per_graph=[150, 170], latent_tokens=(4, 4, 2),
               magno_kw=dict(use_geoembed=[True, True], mlp_type="linear", neighbor_strategy=["radius", "bidirectional"]),
               tr_kw=dict(patch_size=1, hidden_size=64, num_layers=3, positional_embedding="rope"),
               attn_kw=dict(hidden_size=64, num_heads=2, num_kv_heads=1), ffn_kw=dict(hidden_size=128),
               graph="radius", out_ch=4, feats={"pos": 3, "c": 2},
               token_box=((-0.9, -0.8, -0.7), (0.9, 0.8, 0.7)))
    model_case("model_channel_multiscale", seed=3, n_per_graph=[120], latent_tokens=(2, 4, 4),
               magno_kw=dict(use_geoembed=False, mlp_type="channel", scales=[1.0, 2.0], use_scale_weights=True,
                             in_gno_channel_mlp_hidden_layers=[64, 64], out_gno_channel_mlp_hidden_layers=[64]),
               tr_kw=dict(patch_size=2, hidden_size=64, num_layers=2, positional_embedding="rope",
                          use_long_range_skip=False),
               attn_kw=dict(hidden_size=64, num_heads=2, num_kv_heads=2), ffn_kw=dict(hidden_size=96),
               graph="knn", out_ch=2, feats={"pos": 3})


if __name__ == "__main__":
    main()
