"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not shipped, not measured as the product.

CPU restatement (plain PyTorch fp32 ops, no PyG / torch_scatter / omegaconf /
rotary_embedding_torch) of the GAOT-3D forward path of Shizheng-Wen/GAOT-3D, written
functionally over a ``state_dict`` so that it shares no code with the product package
``gaot_3d_amd``.  Backward comes from torch autograd over these ops.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module -- and only as the checker / the timed CPU baseline.

Parity pinning: ``tests/test_oracle_golden.py`` checks every function here against golden
vectors captured in the authoring container by importing the reference's own modules
(``oracle/make_goldens.py`` -> ``tests/golden/*.npz``).  One piece is "parity unpinned":
the RoPE arithmetic lives in the third-party ``rotary-embedding-torch`` wheel (version
unpinned, reference ``requirements.txt:13``), absent from /root/reference and from this
image; ``rope_rotate`` below restates its published algorithm and the rope goldens were
generated through that same restatement (labelled ``rope: third-party, unpinned``).

Every function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# scatter  (src/model/layers/utils/scatter_native.py:4-54)
# --------------------------------------------------------------------------------------
def scatter(src: Tensor, index: Tensor, dim_size: int, reduce: str = "sum") -> Tensor:
    """dim=0 segmented reduction; empty segments give 0 for every reduce."""
    index = index.long()
    shape = list(src.shape)
    shape[0] = dim_size
    out = torch.zeros(shape, dtype=src.dtype, device=src.device)
    idx = index.view([-1] + [1] * (src.dim() - 1)).expand_as(src)
    if reduce in ("sum", "add"):
        return out.scatter_add(0, idx, src)  # scatter_native.py:21-22
    if reduce == "mean":  # scatter_native.py:23-31
        s = out.scatter_add(0, idx, src)
        cnt = torch.bincount(index, minlength=dim_size).to(src.dtype)
        cnt = cnt.view([dim_size] + [1] * (src.dim() - 1))
        return s / cnt.clamp(min=1)
    if reduce in ("max", "amax"):  # scatter_native.py:32-38 (zero-initialised, include_self=False)
        return out.scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    if reduce in ("min", "amin"):  # scatter_native.py:44-49
        out = torch.full(shape, float("inf"), dtype=src.dtype, device=src.device)
        out = out.scatter_reduce(0, idx, src, reduce="amin", include_self=False)
        return torch.where(out == float("inf"), torch.zeros_like(out), out)
    raise ValueError(f"Unsupported reduce operation '{reduce}'")


# --------------------------------------------------------------------------------------
# MLPs  (src/model/layers/mlp.py:308-335 LinearChannelMLP, :227-305 ChannelMLP)
# --------------------------------------------------------------------------------------
def _n_fcs(sd: SD, prefix: str) -> int:
    n = 0
    while f"{prefix}fcs.{n}.weight" in sd:
        n += 1
    return n


def channel_mlp(sd: SD, prefix: str, x: Tensor, act=F.gelu) -> Tensor:
    """Row-major [rows, C_in] -> [rows, C_out].  Handles both Linear ([out,in]) and the
    Conv1d(k=1) storage ([out,in,1]) of mlp_type='channel' (mlp.py:266-275; callers
    transpose around it, magno.py:545,575,775,796-797 -- numerically the same affine map)."""
    n = _n_fcs(sd, prefix)
    for i in range(n):
        w = sd[f"{prefix}fcs.{i}.weight"]
        if w.dim() == 3:
            w = w[:, :, 0]
        x = F.linear(x, w, sd[f"{prefix}fcs.{i}.bias"])
        if i < n - 1:
            x = act(x)  # erf-form GELU, mlp.py:330-331
    return x


# --------------------------------------------------------------------------------------
# IntegralTransform  (src/model/layers/integral_transform.py:80-175)
# --------------------------------------------------------------------------------------
def _segment_softmax(scores: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """integral_transform.py:68-78"""
    smax = scatter(scores, index, dim_size, "max")
    scores = scores - smax[index]
    e = torch.exp(scores)
    esum = scatter(e, index, dim_size, "sum")
    esum = torch.clamp(esum, min=torch.finfo(esum.dtype).tiny)
    return e / esum[index]


def activation_fn(name: str):
    """the reference's name -> callable rule, src/model/layers/mlp.py:27-35: "none" -> identity, "swish" -> SiLU, else F.<name>"""
    if name == "none":
        return lambda x: x
    if name == "swish":
        return F.silu
    if hasattr(F, name):
        return getattr(F, name)
    raise ValueError(f"Activation function {name} not found")


def integral_transform(sd: SD, prefix: str, y_pos: Tensor, x_pos: Tensor, edge_index: Tensor,
                       f_y: Optional[Tensor], transform_type: str = "linear",
                       use_attn: Optional[bool] = None, coord_dim: int = 3,
                       attention_type: str = "cosine", act=F.gelu) -> Tensor:
    nq = x_pos.shape[0]
    if edge_index.shape[1] == 0:  # integral_transform.py:106-112
        n = _n_fcs(sd, prefix + "channel_mlp.")
        cout = sd[f"{prefix}channel_mlp.fcs.{n-1}.weight"].shape[0]
        return torch.zeros(nq, cout, dtype=x_pos.dtype)
    q = edge_index[1].long()  # :114
    s = edge_index[0].long()  # :115
    rep = y_pos[s]            # :117
    slf = x_pos[q]            # :118
    feat = f_y[s] if f_y is not None else None  # :120-123
    attw = None
    if use_attn:              # :126-142
        qc, kc = slf[:, :coord_dim], rep[:, :coord_dim]
        if attention_type == "dot_product":
            qq = F.linear(qc, sd[prefix + "query_proj.weight"], sd[prefix + "query_proj.bias"])
            kk = F.linear(kc, sd[prefix + "key_proj.weight"], sd[prefix + "key_proj.bias"])
            sc = (qq * kk).sum(-1) * (1.0 / (64 ** 0.5))
        elif attention_type == "cosine":
            sc = (F.normalize(qc, p=2, dim=-1) * F.normalize(kc, p=2, dim=-1)).sum(-1)
        else:
            raise ValueError(attention_type)
        attw = _segment_softmax(sc, q, nq)
    agg = torch.cat([rep, slf], dim=-1)  # :146  (source coords first)
    if feat is not None and transform_type in ("nonlinear", "nonlinear_kernelonly"):
        agg = torch.cat([agg, feat], dim=-1)  # :148-152
    k = channel_mlp(sd, prefix + "channel_mlp.", agg, act)  # :154 (channel_mlp_non_linearity, :35)
    if feat is not None and transform_type != "nonlinear_kernelonly":
        k = k * feat          # :156-157
    if attw is not None:
        k = k * attw.unsqueeze(-1)  # :159-160
    red = "sum" if attw is not None else "mean"  # :163
    return scatter(k, q, nq, red)  # :165-171


# --------------------------------------------------------------------------------------
# GeometricEmbedding  (src/model/layers/geoembed.py)
# --------------------------------------------------------------------------------------
def geoembed_stat_features(source_pos: Tensor, query_pos: Tensor, edge_index: Tensor) -> Tensor:
    """Normalised 9-feature statistical descriptor, geoembed.py:99-182."""
    nq, nd = query_pos.shape
    nb = edge_index[0].long()
    qi = edge_index[1].long()
    cnt = torch.bincount(qi, minlength=nq)              # :118-119
    n_i = cnt.float()
    has = n_i > 0
    nbr = source_pos[nb]
    qpe = query_pos[qi]
    dist = torch.norm(nbr - qpe, dim=1)                 # :132
    d_avg = scatter(dist, qi, nq, "mean")               # :133
    e_x2 = scatter(dist ** 2, qi, nq, "mean")           # :135-136
    d_var = torch.clamp(e_x2 - d_avg ** 2, min=0.0)     # :137-139
    cen = scatter(nbr, qi, nq, "mean")                  # :142
    delta = cen - query_pos                             # :143
    ctr = nbr - cen[qi]                                 # :146
    covc = ctr.unsqueeze(2) * ctr.unsqueeze(1)          # :147
    cov_sum = scatter(covc, qi, nq, "sum")              # :148
    n_c = n_i.clone()
    n_c[n_c == 0] = 1.0
    cov = cov_sum / n_c.view(-1, 1, 1)                  # :151
    pca = torch.zeros(nq, nd, dtype=query_pos.dtype)
    if has.any():                                       # :155-166
        reg = cov[has] + 1e-6 * torch.eye(nd, dtype=cov.dtype).unsqueeze(0)
        ev = torch.linalg.eigvalsh(reg).flip(dims=[1])  # descending
        pca[has] = ev
    feats = torch.cat([n_i[:, None], d_avg[:, None], d_var[:, None], delta, pca], dim=1)  # :172
    feats[~has] = 0.0                                   # :175
    mean = feats.mean(dim=0, keepdim=True)              # :177
    std = feats.std(dim=0, keepdim=True)                # :178 (unbiased)
    std = torch.where(std < 1e-6, torch.ones_like(std), std)  # :179
    return (feats - mean) / std                         # :180


def geoembed(sd: SD, prefix: str, source_pos: Tensor, query_pos: Tensor, edge_index: Tensor,
             method: str = "statistical", pooling: str = "max") -> Tensor:
    """geoembed.py:57-93"""
    if method == "statistical":
        f = geoembed_stat_features(source_pos, query_pos, edge_index)
        h = F.relu(F.linear(f, sd[prefix + "mlp.0.weight"], sd[prefix + "mlp.0.bias"]))  # :35-41
        return F.linear(h, sd[prefix + "mlp.2.weight"], sd[prefix + "mlp.2.bias"])
    if method == "pointnet":  # geoembed.py:184-222
        nq = query_pos.shape[0]
        cout = sd[prefix + "fc.0.weight"].shape[0]
        out = torch.zeros(nq, cout, dtype=query_pos.dtype)
        if edge_index.numel() == 0:
            return out
        qi = edge_index[1].long()
        si = edge_index[0].long()
        has = torch.bincount(qi, minlength=nq) > 0
        if not torch.any(has):
            return out
        c = source_pos[si] - query_pos[qi]
        h = F.relu(F.linear(c, sd[prefix + "pointnet_mlp.0.weight"], sd[prefix + "pointnet_mlp.0.bias"]))
        h = F.relu(F.linear(h, sd[prefix + "pointnet_mlp.2.weight"], sd[prefix + "pointnet_mlp.2.bias"]))
        pooled = scatter(h, qi, nq, "max" if pooling == "max" else "mean")
        po = F.linear(pooled, sd[prefix + "fc.0.weight"], sd[prefix + "fc.0.bias"])
        return torch.where(has[:, None], po, out)  # :219 (masked assignment)
    raise ValueError(method)


# --------------------------------------------------------------------------------------
# Neighbour sampling (src/model/layers/magno.py:297-371).  The reference draws from torch's generator (randperm /
# dropout_edge: third-party torch_geometric, "parity unpinned"); the product draws from a counter-based hash of
# (seed, position in the query-sorted edge list), restated here bit for bit (integer work).
# --------------------------------------------------------------------------------------
def _sample_hash(seed: int, idx):
    import numpy as np
    seed &= 0xFFFFFFFFFFFFFFFF
    lo, hi = np.uint64(seed & _M32), np.uint64(seed >> 32)
    x = (lo ^ ((idx.astype(np.uint64) * np.uint64(0x9E3779B1)) & _M32)) & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    x = (x + hi) & _M32
    x ^= x >> np.uint64(17); x = (x * np.uint64(0xed5ad4bb)) & _M32
    x ^= x >> np.uint64(11); x = (x * np.uint64(0xac4c1b51)) & _M32
    x ^= x >> np.uint64(15)
    return x


def neighbor_sampling_keep(seed: int, dst_sorted: Tensor, num_query: int, strategy: str, max_neighbors=None,
                           sample_ratio=None) -> Tensor:
    """bool keep flag per edge of a QUERY-SORTED edge list (positions = the product's CSR order)"""
    import numpy as np
    n = dst_sorted.numel()
    h = _sample_hash(seed, np.arange(n, dtype=np.uint64))
    if strategy == "ratio":
        thr = min(int(sample_ratio * 4294967296.0 + 0.5), 4294967295)
        return torch.from_numpy(h < np.uint64(thr))
    keep = np.ones(n, dtype=bool)
    d = dst_sorted.numpy()
    counts = np.bincount(d, minlength=num_query)
    start = np.concatenate([[0], np.cumsum(counts)])
    for r in np.nonzero(counts > max_neighbors)[0]:
        lo, hi = start[r], start[r + 1]
        order = np.lexsort((np.arange(lo, hi), h[lo:hi]))      # by hash, ties by position
        keep[lo:hi] = False
        keep[lo + order[:max_neighbors]] = True
    return torch.from_numpy(keep)


# --------------------------------------------------------------------------------------
# MAGNO encoder / decoder  (src/model/layers/magno.py:468-600, 691-798)
# --------------------------------------------------------------------------------------
def _pair(v, n=2):
    if isinstance(v, (list, tuple)):
        return tuple(v)
    return (v,) * n


def _gather_feats(batch, attr) -> Tensor:
    """magno.py:485-499"""
    if isinstance(attr, (list, tuple)):
        fs = []
        for a in attr:
            f = getattr(batch, a, None)
            if f is None:
                raise AttributeError(f"MAGNOEncoder requires feature attribute '{a}'")
            fs.append(f)
        return torch.cat(fs, dim=-1)
    f = getattr(batch, attr, None)
    if f is None:
        raise AttributeError(f"MAGNOEncoder requires feature attribute '{attr}'")
    return f


def _scale_sum(sd: SD, prefix: str, outs: Sequence[Tensor], pos: Tensor, use_scale_weights: bool) -> Tensor:
    """magno.py:586-596 / 780-790"""
    if len(outs) == 1:
        return outs[0]
    st = torch.stack(list(outs), dim=0)
    if use_scale_weights:
        w = F.relu(F.linear(pos, sd[prefix + "scale_weighting.0.weight"], sd[prefix + "scale_weighting.0.bias"]))
        w = F.linear(w, sd[prefix + "scale_weighting.2.weight"], sd[prefix + "scale_weighting.2.bias"])
        w = torch.softmax(w, dim=-1)
        return (st * w.permute(1, 0).unsqueeze(-1)).sum(dim=0)
    return st.sum(dim=0)


def magno_encoder(sd: SD, cfg, batch, latent_pos: Tensor, prefix: str = "encoder.") -> Tensor:
    """-> [B, M, C].  Edges are always taken from the batch (precompute_edges=True path,
    magno.py:506-516); graph construction is outside the hot path."""
    phys_pos = batch.pos
    nb = batch.num_graphs
    feat = _gather_feats(batch, cfg.encoder_feature_attr)
    use_geo = _pair(cfg.use_geoembed)[0]
    outs = []
    for si, _ in enumerate(cfg.scales):
        ei = getattr(batch, f"encoder_edge_index_s{si}")
        enc = None
        if cfg.use_gno:
            lifted = channel_mlp(sd, prefix + "lifting.", feat)  # :542-545
            enc = integral_transform(sd, prefix + "gno.", phys_pos, latent_pos, ei, lifted,
                                     cfg.in_gno_transform_type, cfg.use_attn, cfg.gno_coord_dim,
                                     cfg.attention_type)  # :546-553
        geo = None
        if use_geo:
            geo = geoembed(sd, prefix + "geoembed.", phys_pos, latent_pos, ei,
                           cfg.embedding_method, cfg.pooling)  # :557-565
        if enc is not None and geo is not None:
            enc = channel_mlp(sd, prefix + "recovery.", torch.cat([enc, geo], dim=-1))  # :570-575
        elif enc is None and geo is not None:
            enc = geo
        elif enc is None:
            raise ValueError("GNO and GeoEmbed are both disabled.")  # :581
        outs.append(enc)
    out = _scale_sum(sd, prefix, outs, latent_pos, cfg.use_scale_weights)
    return out.view(nb, latent_pos.shape[0] // nb, cfg.lifting_channels)  # :598


def magno_decoder(sd: SD, cfg, rndata_flat: Tensor, phys_pos_query: Tensor, latent_pos: Tensor,
                  batch, prefix: str = "decoder.") -> Tensor:
    """-> [N_query, out]"""
    use_geo = _pair(cfg.use_geoembed)[1]
    outs = []
    for si, _ in enumerate(cfg.scales):
        ei = getattr(batch, f"decoder_edge_index_s{si}")
        dec = integral_transform(sd, prefix + "gno.", latent_pos, phys_pos_query, ei, rndata_flat,
                                 cfg.out_gno_transform_type, cfg.use_attn, cfg.gno_coord_dim,
                                 cfg.attention_type)  # :751-758
        if use_geo:  # :761-775
            geo = geoembed(sd, prefix + "geoembed.", latent_pos, phys_pos_query, ei,
                           cfg.embedding_method, cfg.pooling)
            dec = channel_mlp(sd, prefix + "recovery.", torch.cat([dec, geo], dim=-1))
        outs.append(dec)
    out = _scale_sum(sd, prefix, outs, phys_pos_query, cfg.use_scale_weights)
    return channel_mlp(sd, prefix + "projection.", out)  # :793-797


# --------------------------------------------------------------------------------------
# Transformer  (src/model/layers/attn.py)
# --------------------------------------------------------------------------------------
def rmsnorm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    """attn.py:167-178"""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x) * w


def rope_rotate(t: Tensor, freqs: Tensor) -> Tensor:
    """Restatement of rotary_embedding_torch.RotaryEmbedding(dim).rotate_queries_or_keys(t)
    (third-party, unpinned; call sites attn.py:86-87,119-120).  t: [B,h,S,d]; freqs: [d/2]
    (= 1/10000^(arange(0,d,2)/d), stored as the non-trainable parameter ``rotary_emb.freqs``).
    Positions are arange(S) along dim -2; each frequency is repeated twice adjacently; pairs
    (x_{2i}, x_{2i+1}) are rotated: out = t*cos + rotate_half(t)*sin."""
    s = t.shape[-2]
    pos = torch.arange(s, dtype=freqs.dtype)
    ang = pos[:, None] * freqs[None, :]              # [S, d/2]
    ang = ang.repeat_interleave(2, dim=-1)           # (n r), r=2
    x1 = t[..., 0::2]
    x2 = t[..., 1::2]
    rh = torch.stack((-x2, x1), dim=-1).flatten(-2)  # rotate_half on interleaved pairs
    return t * ang.cos() + rh * ang.sin()


# Attention dropout (attn.py:122-127: F.scaled_dot_product_attention(dropout_p=atten_dropout) when training).  The
# mask torch draws comes from its Philox stream and cannot be reproduced outside torch -- "parity unpinned" for the
# mask itself; SDPA's arithmetic GIVEN a mask (P * keep / (1-p), then @ V) is pinned by the reference golden
# ``attn_dropout`` (make_goldens.py replays torch's CPU mask through the same generator state).  The product draws
# its mask from a counter-based integer function (csrc/attn_dropout.h) that is restated here bit for bit.
_M32 = 0xFFFFFFFF


def _mix_a(x):
    import numpy as np
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & _M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & _M32
    x ^= x >> np.uint64(16)
    return x


def _mix_b(x):
    import numpy as np
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(17); x = (x * np.uint64(0xed5ad4bb)) & _M32
    x ^= x >> np.uint64(11); x = (x * np.uint64(0xac4c1b51)) & _M32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x31848bab)) & _M32
    x ^= x >> np.uint64(14)
    return x


def dropout_threshold(p: float) -> int:
    """p is realised to 1/65536: keep iff a 16-bit uniform >= thr"""
    return max(0, min(65535, int(p * 65536.0 + 0.5)))


def dropout_keep_mask(seed: int, b: int, h: int, s: int, p: float) -> Tensor:
    """keep[b, h, q, k] (bool) of the product's attention dropout: W = A(seed, bh, q) xor B(seed, bh, k >> 1),
    keep = halfword(W, k & 1) >= thr.  ``seed`` is the unsigned 64-bit word the kernels read."""
    import numpy as np
    seed &= 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & _M32, seed >> 32
    thr = dropout_threshold(p)
    bh = np.arange(b * h, dtype=np.uint64)
    rk = _mix_a((np.uint64(lo) + np.uint64(0x9E3779B9) * (bh + np.uint64(1))) & _M32)        # [BH]
    ck = _mix_b((np.uint64(hi) + np.uint64(0x85EBCA6B) * (bh + np.uint64(1))) & _M32)
    q = np.arange(s, dtype=np.uint64)
    aw = _mix_a((rk[:, None] + q[None, :]) & _M32)                                            # [BH, S]
    bw = _mix_b((ck[:, None] + (q[None, :] >> np.uint64(1))) & _M32)                          # [BH, S] (pair word of k)
    w = aw[:, :, None] ^ bw[:, None, :]                                                       # [BH, Sq, Sk]
    half = np.where((q & np.uint64(1))[None, None, :] == 1, w >> np.uint64(16), w & np.uint64(0xFFFF))
    return torch.from_numpy((half >= thr).reshape(b, h, s, s))


def dropout_keep_rows(seed: int, bh: int, q_rows, s: int, p: float) -> Tensor:
    """rows ``q_rows`` of dropout_keep_mask(...)[b, h] for the flattened head index bh = b * H + h -> bool [len(q_rows), s]
    (the same integer function, evaluated for a subset of query rows: full-size checks at S = 16 384)"""
    import numpy as np
    seed &= 0xFFFFFFFFFFFFFFFF
    lo, hi = seed & _M32, seed >> 32
    thr = dropout_threshold(p)
    one = np.array([bh + 1], dtype=np.uint64)
    rk = _mix_a((np.uint64(lo) + np.uint64(0x9E3779B9) * one) & _M32)
    ck = _mix_b((np.uint64(hi) + np.uint64(0x85EBCA6B) * one) & _M32)
    q = np.asarray(q_rows, dtype=np.uint64)
    kk = np.arange(s, dtype=np.uint64)
    aw = _mix_a((rk + q) & _M32)                                   # [R]
    bw = _mix_b((ck + (kk >> np.uint64(1))) & _M32)                # [S]
    w = aw[:, None] ^ bw[None, :]
    half = np.where((kk & np.uint64(1))[None, :] == 1, w >> np.uint64(16), w & np.uint64(0xFFFF))
    return torch.from_numpy(half >= thr)


def sdpa(q: Tensor, k: Tensor, v: Tensor, keep: Optional[Tensor] = None, p_drop: float = 0.0) -> Tensor:
    """F.scaled_dot_product_attention(q, k, v, dropout_p) as the reference calls it (attn.py:122-127): non-causal, no mask,
    scale 1/sqrt(head_dim); ``keep`` = the Bernoulli mask of the training path (torch.dropout on the attention weights:
    mask and rescale by 1/(1-p)).  q, k, v: [B, H, S, d] (any float dtype: the full-size checks call it in fp64)."""
    if keep is None:
        # the reference's own call (attn.py:126).  On the CPU this is torch's tiled kernel: no [S, S] weights in memory (8.6 GB per
        # layer at S = 16 384, kept for the backward) and ~20x faster than the explicit form below, which it equals to fp32
        # rounding (1e-7 on the output, 1e-6 on the gradients at S = 4 096)
        # p_drop > 0 without a mask: torch draws the mask itself, exactly the reference's training-mode call (timing runs only:
        # the draw cannot be replayed outside torch, parity runs pass ``keep``)
        return F.scaled_dot_product_attention(q, k, v, dropout_p=p_drop)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1]), dim=-1)  # the same, written out: the mask acts on the weights
    att = att * keep.to(att.dtype) / (1.0 - p_drop)
    return att @ v


def conditioned_norm(sd: SD, prefix: str, c: Tensor, x: Tensor) -> Tensor:
    """mlp.py:74-128 (ConditionedNorm; its MLPs have num_layers=2 -> one Linear each, activation "none")"""
    sc = 1 + c * F.linear(c, sd[prefix + "mlp_scale.layers.0.weight"], sd[prefix + "mlp_scale.layers.0.bias"])
    bi = c * F.linear(c, sd[prefix + "mlp_bias.layers.0.weight"], sd[prefix + "mlp_bias.layers.0.bias"])
    return x * sc[:, None, :] + bi[:, None, :]


def attention(sd: SD, prefix: str, x: Tensor, num_heads: int, num_kv_heads: int, rope: bool,
              keep: Optional[Tensor] = None, p_drop: float = 0.0, condition: Optional[Tensor] = None) -> Tensor:
    """attn.py:89-131; ``keep`` [B,H,S,S] = the dropout mask of the training path (None: eval / atten_dropout=0),
    ``p_drop`` the probability it was drawn with."""
    if (prefix + "correction.mlp_scale.layers.0.weight") in sd:  # attn.py:101-102
        x = conditioned_norm(sd, prefix + "correction.", condition, x)
    q = F.linear(x, sd[prefix + "q_proj.weight"])
    k = F.linear(x, sd[prefix + "k_proj.weight"])
    v = F.linear(x, sd[prefix + "v_proj.weight"])
    b, s, _ = q.shape
    hd = q.shape[-1] // num_heads
    q = q.view(b, s, num_heads, hd).transpose(1, 2)
    k = k.view(b, s, num_kv_heads, hd).transpose(1, 2)
    v = v.view(b, s, num_kv_heads, hd).transpose(1, 2)
    if num_kv_heads != num_heads:  # :114-116
        rep = num_heads // num_kv_heads
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
    if rope:  # :118-120
        fr = sd[prefix + "rotary_emb.freqs"]
        q = rope_rotate(q, fr)
        k = rope_rotate(k, fr)
    o = sdpa(q, k, v, keep, p_drop).transpose(1, 2).contiguous().view(b, s, -1)
    return F.linear(o, sd[prefix + "o_proj.weight"])


def ffn(sd: SD, prefix: str, x: Tensor, condition: Optional[Tensor] = None) -> Tensor:
    """attn.py:155-161"""
    a = F.linear(x, sd[prefix + "w1.weight"])
    g = F.linear(x, sd[prefix + "w3.weight"])
    y = F.linear(F.silu(a) * g, sd[prefix + "w2.weight"])
    if (prefix + "correction.mlp_scale.layers.0.weight") in sd:  # :158-159: on the OUTPUT of the FFN
        y = conditioned_norm(sd, prefix + "correction.", condition, y)
    return y


def transformer_block(sd: SD, prefix: str, x: Tensor, tcfg, rope: bool, skip: Optional[Tensor],
                      keep: Optional[Tensor] = None, p_drop: float = 0.0) -> Tensor:
    """attn.py:205-230 -- note the second residual adds the *normalised* h."""
    ac = tcfg.attn_config
    if skip is not None and (prefix + "skip_proj.weight") in sd:
        x = F.linear(torch.cat([x, skip], dim=-1), sd[prefix + "skip_proj.weight"], sd[prefix + "skip_proj.bias"])
    h = rmsnorm(x, sd[prefix + "attn_norm.weight"], tcfg.norm_eps) if tcfg.use_attn_norm else x
    h = x + attention(sd, prefix + "attn.", h, ac.num_heads, ac.num_kv_heads, rope, keep, p_drop)
    h = rmsnorm(h, sd[prefix + "ffn_norm.weight"], tcfg.norm_eps) if tcfg.use_ffn_norm else h
    return h + ffn(sd, prefix + "ffn.", h)


def transformer(sd: SD, prefix: str, x: Tensor, tcfg, rope: bool, drop=None) -> Tensor:
    """attn.py:298-325.  ``drop``: None, or (list of keep masks in block call order, p) for the training path, or
    ("torch", p): SDPA draws its own mask (the reference's call as it stands; CPU-baseline timing)."""
    masks = iter(drop[0]) if drop is not None and not isinstance(drop[0], str) else None
    pd = drop[1] if drop is not None else 0.0
    nxt = (lambda: next(masks)) if masks is not None else (lambda: None)
    if (prefix + "input_proj.weight") in sd:
        x = F.linear(x, sd[prefix + "input_proj.weight"], sd[prefix + "input_proj.bias"])
    n = tcfg.num_layers
    skips = []
    for i in range(n // 2):
        x = transformer_block(sd, f"{prefix}encoder_layers.{i}.", x, tcfg, rope, None, nxt(), pd)
        skips.append(x)
    if n % 2 == 1:
        x = transformer_block(sd, f"{prefix}middle_layer.", x, tcfg, rope, None, nxt(), pd)
    for i in range(n // 2):
        sk = skips.pop() if tcfg.use_long_range_skip else None
        x = transformer_block(sd, f"{prefix}decoder_layers.{i}.", x, tcfg, rope, sk, nxt(), pd)
    if (prefix + "output_proj.weight") in sd:
        x = F.linear(x, sd[prefix + "output_proj.weight"], sd[prefix + "output_proj.bias"])
    return x


# --------------------------------------------------------------------------------------
# GAOT3D  (src/model/gaot_3d.py)
# --------------------------------------------------------------------------------------
def absolute_pe(positions: Tensor, embed_dim: int) -> Tensor:
    """gaot_3d.py:102-144"""
    freq = 1 / 10000 ** (2 * torch.arange(0, embed_dim // 2, dtype=torch.float32) / embed_dim)
    ang = positions[:, :, None] * freq[None, None, :]
    pe = torch.zeros(positions.shape[0], embed_dim)
    pe[:, 0::2] = torch.sin(ang).sum(dim=1)
    pe[:, 1::2] = torch.cos(ang).sum(dim=1)
    return pe


def patch_positions(latent_tokens, p: int) -> Tensor:
    """gaot_3d.py:85-100"""
    d, h, w = latent_tokens
    return torch.stack(torch.meshgrid(torch.arange(d // p, dtype=torch.float32),
                                      torch.arange(h // p, dtype=torch.float32),
                                      torch.arange(w // p, dtype=torch.float32), indexing="ij"),
                       dim=-1).reshape(-1, 3)


def process(sd: SD, tcfg, latent_tokens, rndata: Tensor, drop=None) -> Tensor:
    """gaot_3d.py:166-222"""
    b, m, c = rndata.shape
    d, h, w = latent_tokens
    p = tcfg.patch_size
    assert m == d * h * w
    assert d % p == 0 and h % p == 0 and w % p == 0
    nd, nh, nw = d // p, h // p, w // p
    x = rndata.view(b, nd, p, nh, p, nw, p, c).permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous()
    x = x.view(b, nd * nh * nw, p * p * p * c)
    x = F.linear(x, sd["patch_linear.weight"], sd["patch_linear.bias"])
    rope = False
    if tcfg.positional_embedding == "absolute":
        x = x + absolute_pe(patch_positions(latent_tokens, p), p * p * p * c)
    elif tcfg.positional_embedding == "rope":
        rope = True
    x = transformer(sd, "processor.", x, tcfg, rope, drop)
    x = x.view(b, nd, nh, nw, p, p, p, c).permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous()
    return x.view(b, d * h * w, c)


def gaot3d_forward(sd: SD, mcfg, batch, tokens_pos: Optional[Tensor] = None,
                   query_coord_pos: Optional[Tensor] = None, drop=None) -> Tensor:
    """gaot_3d.py:248-332.  ``mcfg`` has .magno, .transformer, .latent_tokens."""
    nb = batch.num_graphs
    lt = sd["latent_tokens"] if tokens_pos is None else tokens_pos
    latent = lt.repeat(nb, 1)  # :283-285
    q_pos = batch.pos if query_coord_pos is None else query_coord_pos
    rn = magno_encoder(sd, mcfg.magno, batch, latent)
    rn = process(sd, mcfg.transformer, tuple(mcfg.latent_tokens), rn, drop)
    flat = rn.reshape(-1, mcfg.magno.lifting_channels)
    return magno_decoder(sd, mcfg.magno, flat, q_pos, latent, batch)


def mse_loss(pred: Tensor, target: Tensor) -> Tensor:
    """nn.MSELoss() (src/trainer/base.py:56, used stat.py:550)"""
    return ((pred - target) ** 2).mean()


def train_step_grads(sd: SD, mcfg, batch, tokens_pos: Optional[Tensor] = None, drop=None):
    """One zero_grad -> forward -> MSE -> backward (optimizers.py:272-274) over a
    state_dict; returns (pred, loss, {name: grad}).  Non-float / frozen entries
    (``latent_tokens`` buffer, ``rotary_emb.freqs``) get no grad, as in the reference."""
    leaf = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point() and k != "latent_tokens" and not k.endswith("rotary_emb.freqs"):
            t.requires_grad_(True)
        leaf[k] = t
    pred = gaot3d_forward(leaf, mcfg, batch, tokens_pos, drop=drop)
    loss = mse_loss(pred, batch.x)
    loss.backward()
    grads = {k: v.grad for k, v in leaf.items() if v.requires_grad and v.grad is not None}
    return pred.detach(), loss.detach(), grads


def element_dropout_keep(seed: int, n: int, p: float) -> Tensor:
    """bool keep[n] of the product's element dropout (csrc/graph.hip k_dropout; stands for nn.Dropout in the reference's
    channel MLPs, mlp.py:268-272, 318-322 -- torch's own draw is "parity unpinned"): hash(seed, i) >= round(p 2^32)"""
    import numpy as np
    thr = min(int(p * 4294967296.0 + 0.5), 4294967295)
    return torch.from_numpy(_sample_hash(seed, np.arange(n, dtype=np.uint64)) >= np.uint64(thr))
