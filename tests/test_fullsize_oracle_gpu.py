"""Oracle parity AT BASELINE configs[1] sizes (VERDICT r1 "full-size parity is self-comparison only"):

* attention at S = 16 384 tokens, head_dim 32: one head against the oracle's SDPA restatement (oracle.sdpa, reference
  attn.py:110-127) evaluated in fp64 on the host, query-chunked -- forward output and dQ / dK / dV, for the exact-fp32
  kernels and for the bf16 matrix-core kernels, without and with the training-mode dropout mask (the mask the kernels
  regenerate is checked bit for bit against the oracle's integer restatement on sampled rows of the 2.7 x 10^8-element
  mask, then taken from the device for the fp64 reference);
* the fused GNO integral transform on the full 4.0 M-edge graphs of the 500 000-point sample, both directions
  (points -> 131 072 latent tokens: variable degree with empty rows and rows of hundreds of edges; tokens -> points):
  forward against oracle.integral_transform (reference integral_transform.py:114-171) on the sub-graph of >= 1 000 sampled
  query rows that include the heaviest row and empty rows; backward: grad f_y on >= 1 000 sampled source rows (all of
  their edges) and ALL weight / bias gradients against the oracle's kernel MLP evaluated in fp64 over all 4 M edges.

Tolerances: fp32 kernels rtol 1e-4 / atol 1e-5-of-peak on outputs, rtol 1e-3 on gradients; bf16 kernels
max-abs <= 1e-2 of the reference's peak on outputs, cosine >= 0.999 and max-abs <= 1e-2 / 2e-2 of peak on gradients (<= 3 x the
achieved errors of profiles/archive/r4_ad_parity.txt: a 5 x regression fails)."""
import math
import os
import sys

import pytest
import torch

import parity as PAR

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import gaot_oracle as orc  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S_FULL = 16384
N_PTS, LATENT, KNN = 500_000, (64, 64, 32), 8


def _attention_fp64(qkv, w, freqs, keep, p_eff, chunk=2048):
    """oracle.sdpa in fp64 on [1, 1, S, 32] operands, one query chunk at a time (S x S in fp64 is 2 GB): returns the
    output and the gradients of <out, w> with respect to the fused q | k | v projection"""
    s = qkv.shape[0]
    # thread count as for the whole-step oracle below: all 256 hardware threads of the pool's hosts run torch's CPU kernels several
    # times slower than 32-64 of them (bench.py's cpu_baseline measured 9x)
    torch.set_num_threads(int(os.environ.get("GAOT_ORACLE_THREADS", min(os.cpu_count() or 1, 32))))
    x = qkv.double().requires_grad_(True)
    q, k, v = (t.view(1, s, 1, 32).transpose(1, 2) for t in x.split(32, dim=1))
    if freqs is not None:
        # fp32 frequencies: the reference forms the rotation angles position * frequency in fp32 (rotary_embedding_torch on
        # fp32 tensors), and at position 16 383 that rounding (~1e-3 rad) is part of the operator's definition
        q, k = orc.rope_rotate(q, freqs.float()), orc.rope_rotate(k, freqs.float())
    outs = []
    for lo in range(0, s, chunk):
        hi = min(lo + chunk, s)
        kp = None if keep is None else keep[None, None, lo:hi]
        o = orc.sdpa(q[:, :, lo:hi], k, v, kp, p_eff)                     # [1, 1, c, 32]
        (o[0, 0] * w[lo:hi].double()).sum().backward(retain_graph=True)
        outs.append(o.detach()[0, 0])
    return torch.cat(outs), x.grad


@pytest.mark.parametrize("p", [0.0, 0.1])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_attention_full_sequence_vs_fp64_oracle(precision, p):
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd import ops
    s = S_FULL
    g = torch.Generator().manual_seed(1234)
    qkv = torch.randn(s, 96, generator=g)
    qkv[:, :64] *= 1.5                         # logits of spread ~2.2: a softmax that is neither flat nor one-hot
    w = torch.randn(s, 32, generator=g)
    freqs = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    seed0 = 0x5EED_F011 + int(p * 100)
    keep, p_eff = None, 0.0
    gaot_3d_amd.set_precision(precision)
    try:
        GF.set_dropout_seed(seed0, DEV)
        qd = qkv.to(DEV).requires_grad_(True)
        out = GF.AttentionFn.apply(qd, freqs.to(DEV), 1, s, 1, 1, p)
        (out * w.to(DEV)).sum().backward()
        torch.cuda.synchronize()
    finally:
        gaot_3d_amd.set_precision("fp32")
    if p > 0.0:
        p_eff = orc.dropout_threshold(p) / 65536.0
        word = GF.dropout_seed_sequence(seed0, 1)[0]
        st = torch.tensor([word - (1 << 64) if word >= (1 << 63) else word], dtype=torch.int64, device=DEV)
        keep = ops.attn_dropout_mask(st, p, 1, 1, s).cpu()[0, 0].bool()     # what the kernels regenerate
        rows = [0, 1, 31, 32, 4095, 4096, 8191, 12345, s - 2, s - 1] + torch.randint(0, s, (54,), generator=g).tolist()
        assert torch.equal(keep[rows], orc.dropout_keep_rows(word, 0, rows, s, p))   # bit-exact against the oracle's draw
        rate = keep.float().mean().item()
        print(f"[parity] attn_full_{precision}_p{p}: mask rows bit-exact on {len(rows)} sampled rows, keep rate {rate:.5f}")
        assert abs(rate - (1.0 - p_eff)) < 1e-3
    ref, gref = _attention_fp64(qkv, w, freqs, keep, p_eff)
    tag = f"attn_full_S{s}_{precision}_p{p}"
    names = (("dq", 0, 32), ("dk", 32, 64), ("dv", 64, 96))
    if precision == "fp32":
        PAR.close(f"{tag}/out", out, ref, 1e-4, 1e-5 * float(ref.abs().max()))
        for nm, lo, hi in names:
            PAR.close(f"{tag}/{nm}", qd.grad[:, lo:hi], gref[:, lo:hi], 1e-3, 2e-5 * float(gref[:, lo:hi].abs().max()))
    else:
        PAR.close_peak(f"{tag}/out", out, ref, 2e-2, rel_l2=1e-2)
        for nm, lo, hi in names:
            PAR.cosine(f"{tag}/{nm}", qd.grad[:, lo:hi], gref[:, lo:hi], 0.999)
            PAR.close_peak(f"{tag}/{nm}", qd.grad[:, lo:hi], gref[:, lo:hi], 2e-2)   # achieved ~8e-3 (2e-2 of peak)


@pytest.mark.parametrize("h", [4, 8])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_fused_backward_vs_fp64_oracle(p, h):
    """The FUSED backward (dK, dV and dQ from one pass, bf16 slab partials, fixed-order reduction) at S = 16 384, against the
    oracle's SDPA in fp64 for ONE of the heads (head 2: mask keyed by the head index, per-head slices of lse / delta / partials),
    and against the two-pass kernels.  4 heads: the compiled kernel k_attn_bwd_fused with the queries in two parts (128 slabs x
    heads < 256 workgroups); 8 heads: a whole-sequence launch = the hand-scheduled k_attn_bwd_asm (round 5)."""
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd import ops
    s, hd = S_FULL, 2
    g = torch.Generator().manual_seed(4321)
    qkv = torch.randn(s, 3 * h * 32, generator=g)
    qkv[:, :2 * h * 32] *= 1.5
    w = torch.randn(s, h * 32, generator=g)
    freqs = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    seed0 = 0xF05ED + int(p * 100)
    scale = 32 ** -0.5
    word = GF.dropout_seed_sequence(seed0, 1)[0]
    st = torch.tensor([word - (1 << 64) if word >= (1 << 63) else word], dtype=torch.int64, device=DEV) if p > 0 else None
    assert ops.get_precision() == "fp32"
    qd, wd, fd = qkv.to(DEV), w.to(DEV), freqs.to(DEV)
    o, lse, img = ops.attn_fwd_bf16(qd, fd, 1, s, h, h, scale, p, st)
    g_fused = ops.attn_bwd_bf16(img, o, wd, lse, 1, s, h, h, scale, p, st, freqs=fd, fused=True)
    g_fused2 = ops.attn_bwd_bf16(img, o, wd, lse, 1, s, h, h, scale, p, st, freqs=fd, fused=True)
    g_two = ops.attn_bwd_bf16(img, o, wd, lse, 1, s, h, h, scale, p, st, freqs=fd, fused=False)
    torch.cuda.synchronize()
    assert torch.equal(g_fused, g_fused2)                       # fixed summation order: bit-reproducible
    tag = f"attn_fused_S{s}_h{h}_p{p}"
    for nm, lo, hi in (("dq", 0, h * 32), ("dk", h * 32, 2 * h * 32), ("dv", 2 * h * 32, 3 * h * 32)):
        PAR.cosine(f"{tag}/{nm} fused vs two-pass", g_fused[:, lo:hi], g_two[:, lo:hi], 0.99999)
        PAR.close_peak(f"{tag}/{nm} fused vs two-pass", g_fused[:, lo:hi], g_two[:, lo:hi], 1e-2)
    keep, p_eff = None, 0.0
    if p > 0.0:
        p_eff = orc.dropout_threshold(p) / 65536.0
        keep = ops.attn_dropout_mask(st, p, 1, h, s)[0, hd].cpu().bool()
        rows = [0, 31, 4096, s - 1] + torch.randint(0, s, (28,), generator=g).tolist()
        assert torch.equal(keep[rows], orc.dropout_keep_rows(word, hd, rows, s, p))
    cols = lambda blk: slice((blk * h + hd) * 32, (blk * h + hd + 1) * 32)
    one = torch.cat([qkv[:, cols(0)], qkv[:, cols(1)], qkv[:, cols(2)]], dim=1)
    ref, gref = _attention_fp64(one, w[:, hd * 32:(hd + 1) * 32], freqs, keep, p_eff)
    PAR.close_peak(f"{tag}/out", o[:, hd * 32:(hd + 1) * 32], ref, 2e-2, rel_l2=1e-2)
    # the kernels return dq / dk w.r.t. the UNrotated projection (freqs given), as the fp64 oracle does
    for blk, nm in ((0, "dq"), (1, "dk"), (2, "dv")):
        got, want = g_fused[:, cols(blk)], gref[:, 32 * blk:32 * (blk + 1)]
        PAR.cosine(f"{tag}/{nm} vs fp64", got, want, 0.999)
        PAR.close_peak(f"{tag}/{nm} vs fp64", got, want, 2e-2)   # achieved ~8e-3 (2e-2 of peak)


@pytest.mark.parametrize("b,s,h,hkv,p", [(8, 2085, 8, 4, 0.1), (2, 4133, 8, 8, 0.0), (1, 16384 - 37, 8, 4, 0.1),
                                         (1, 16384, 1, 1, 0.1), (1, 16384, 2, 1, 0.0), (1, 9000 + 13, 2, 2, 0.1), (1, 16384, 4, 4, 0.1),
                                         # >= 256 workgroups: whole-sequence launches = the hand-scheduled k_attn_bwd_asm (round 5) --
                                         # ragged last stage / slab, batches, grouped-query heads (two q heads per kv head), no dropout
                                         (2, 8192 + 77, 8, 8, 0.1), (4, 8192, 8, 4, 0.1), (1, 16384 - 37, 8, 8, 0.0), (2, 16384 - 130, 8, 4, 0.1)])
def test_attention_fused_backward_ragged_gqa_batches(b, s, h, hkv, p):
    """ragged sequence lengths (keys and queries past S inside the last slab / stage), grouped-query heads (a workgroup loops
    over the heads of its kv head), batches, and FEW heads per launch (the heads one rank of a sharded step owns: the queries
    are split into up to eight parts whose dK / dV partials are summed in part order): fused against the two-pass kernels on
    the same image"""
    from gaot_3d_amd import ops
    g = torch.Generator().manual_seed(s)
    qkv = (torch.randn(b * s, (h + 2 * hkv) * 32, generator=g) * 0.7).to(DEV)
    d_o = torch.randn(b * s, h * 32, generator=g).to(DEV)
    freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(DEV)
    st = torch.tensor([77 + s], dtype=torch.int64, device=DEV) if p > 0 else None
    scale = 32 ** -0.5
    o, lse, img = ops.attn_fwd_bf16(qkv, freqs, b, s, h, hkv, scale, p, st)
    a = ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, hkv, scale, p, st, freqs=freqs, fused=True)
    a2 = ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, hkv, scale, p, st, freqs=freqs, fused=True)
    r = ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, hkv, scale, p, st, freqs=freqs, fused=False)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert torch.equal(a, a2)          # fixed summation orders (slabs, waves, query parts): bit-identical reruns
    for nm, lo, hi in (("dq", 0, h * 32), ("dk", h * 32, (h + hkv) * 32), ("dv", (h + hkv) * 32, (h + 2 * hkv) * 32)):
        PAR.cosine(f"attn_fused_ragged_b{b}_s{s}/{nm}", a[:, lo:hi], r[:, lo:hi], 0.99999)
        PAR.close_peak(f"attn_fused_ragged_b{b}_s{s}/{nm}", a[:, lo:hi], r[:, lo:hi], 1e-2)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sample():
    from gaot_3d_amd.data import make_synthetic_sample
    batch, tokens = make_synthetic_sample(N_PTS, LATENT, k=KNN, seed=0, device=DEV)
    return batch, tokens.to(DEV)


def _mlp_sd(nh, seed):
    g = torch.Generator().manual_seed(seed)
    dims = [6] + [64] * nh + [32]
    sd = {}
    for i in range(len(dims) - 1):
        bound = 1.0 / math.sqrt(dims[i])
        sd[f"channel_mlp.fcs.{i}.weight"] = (torch.rand(dims[i + 1], dims[i], generator=g) * 2 - 1) * bound * 1.7
        sd[f"channel_mlp.fcs.{i}.bias"] = (torch.rand(dims[i + 1], generator=g) * 2 - 1) * bound
    return sd


def _sample_rows(deg, n, gen):
    """>= n rows: the heaviest, the lightest non-empty, some empty ones, the first / last row and random ones"""
    rows = {int(deg.argmax()), 0, deg.numel() - 1}
    empty = (deg == 0).nonzero().flatten()
    if empty.numel():
        rows.update(empty[torch.randint(0, empty.numel(), (50,), generator=gen)].tolist())
    nz = (deg > 0).nonzero().flatten()
    rows.add(int(nz[deg[nz].argmin()]))
    rows.update(nz[torch.randint(0, nz.numel(), (n,), generator=gen)].tolist())
    return torch.tensor(sorted(rows), dtype=torch.long)


_CFG3_GRAPHS = {}


def _cfg3_graph(batch, tokens, which):
    """BASELINE configs[3] graphs on the 500 000-point sample, built by the device kernels (checked against the graph oracle
    in tests/test_graph_gpu.py): radius-graph encoder (r = 0.033, at most 32 points per token) and bidirectional decoder
    (knn k = 1 united with the radius graph around the points)"""
    if which not in _CFG3_GRAPHS:
        from gaot_3d_amd.model.layers.magno import get_neighbor_strategy
        lat_b = torch.zeros(tokens.shape[0], dtype=torch.long, device=DEV)
        strat, dec = ("radius", False) if which == "encoder_radius" else ("bidirectional", True)
        _CFG3_GRAPHS[which] = get_neighbor_strategy(strat, batch.pos, batch.batch, tokens, lat_b, 0.033, 1, dec,
                                                    latent_dims=(64, 64, 32)).to(torch.int32).contiguous()
    return _CFG3_GRAPHS[which]


@pytest.mark.parametrize("side,nh", [("encoder", 3), ("decoder", 2), ("encoder_radius", 3), ("decoder_bidirectional", 2)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_gno_full_graph_vs_oracle(sample, precision, side, nh):
    """the fused GNO forward / backward on the FULL graphs of configs[1] (knn, flipped) and of configs[3] (radius-capped
    encoder, bidirectional decoder: variable degree on both sides) against the oracle on sampled rows"""
    from gaot_3d_amd import ops
    batch, tokens = sample
    m = tokens.shape[0]
    if side == "encoder":      # points -> tokens: variable degree, empty tokens, tokens with hundreds of points
        ei, y_pos, x_pos, n_src, n_dst = batch.encoder_edge_index_s0, batch.pos, tokens, N_PTS, m
    elif side == "encoder_radius":
        ei, y_pos, x_pos, n_src, n_dst = _cfg3_graph(batch, tokens, side), batch.pos, tokens, N_PTS, m
    elif side == "decoder_bidirectional":
        ei, y_pos, x_pos, n_src, n_dst = _cfg3_graph(batch, tokens, side), tokens, batch.pos, m, N_PTS
    else:                      # tokens -> points: degree k everywhere
        ei, y_pos, x_pos, n_src, n_dst = batch.decoder_edge_index_s0, tokens, batch.pos, m, N_PTS
    gen = torch.Generator().manual_seed(77)
    sd = _mlp_sd(nh, 5 + nh)
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(nh + 1)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(nh + 1)]
    f_y = torch.randn(n_src, 32, generator=gen)
    dout = torch.randn(n_dst, 32, generator=gen)
    graph = ops.build_graph(ei, n_src, n_dst)
    prec = 0 if precision == "fp32" else 1
    out = ops.gno_forward(ws, bs, y_pos, x_pos, f_y.to(DEV), graph, precision=prec)
    gf, gw, gb = ops.gno_backward(ws, bs, y_pos, x_pos, f_y.to(DEV), dout.to(DEV), graph, precision=prec)
    torch.cuda.synchronize()
    ei_c = ei.cpu().long()
    src, dst = ei_c[0], ei_c[1]
    deg_dst = torch.bincount(dst, minlength=n_dst)
    deg_src = torch.bincount(src, minlength=n_src)
    y_c, x_c = y_pos.cpu(), x_pos.cpu()
    tag = f"gno_full_{side}_nh{nh}_{precision}"
    print(f"[parity] {tag}: E={ei.shape[1]} max query degree {int(deg_dst.max())}, empty query rows {int((deg_dst == 0).sum())}, "
          f"max source degree {int(deg_src.max())}")

    # ---- forward: the oracle's IntegralTransform on the sub-graph of the sampled query rows --------------------------------
    qs = _sample_rows(deg_dst, 1000, gen)
    local = torch.full((n_dst,), -1, dtype=torch.long)
    local[qs] = torch.arange(qs.numel())
    sel = local[dst] >= 0
    sub = torch.stack([src[sel], local[dst[sel]]])
    ref = orc.integral_transform(sd, "", y_c, x_c[qs], sub, f_y)
    assert int((deg_dst[qs] == 0).sum()) > 0 or side.startswith("decoder")
    if precision == "fp32":
        PAR.close(f"{tag}/out[{qs.numel()} rows, {int(sel.sum())} edges]", out[qs.to(DEV)], ref, 1e-4, 1e-5 * float(ref.abs().max()))
    else:
        PAR.close_peak(f"{tag}/out[{qs.numel()} rows, {int(sel.sum())} edges]", out[qs.to(DEV)], ref, 1e-2, rel_l2=1e-2)   # achieved 2.9e-3

    # ---- backward, grad f_y: every edge of >= 1 000 sampled source rows; mean reduction = sum / degree of the query -------
    ss = _sample_rows(deg_src, 1000, gen)
    local = torch.full((n_src,), -1, dtype=torch.long)
    local[ss] = torch.arange(ss.numel())
    sel = local[src] >= 0
    es, ed = src[sel], dst[sel]
    sd64 = {k: v.double() for k, v in sd.items()}
    fsub = f_y[ss].double().requires_grad_(True)
    kern = orc.channel_mlp(sd64, "channel_mlp.", torch.cat([y_c[es], x_c[ed]], dim=1).double())       # :146-154
    msg = kern * fsub[local[es]]                                                                          # :156-157
    (msg * (dout[ed].double() / deg_dst[ed].clamp(min=1)[:, None])).sum().backward()                      # :165-171 (mean)
    if precision == "fp32":
        PAR.close(f"{tag}/grad_f_y[{ss.numel()} rows]", gf[ss.to(DEV)], fsub.grad, 1e-3, 1e-5 * float(fsub.grad.abs().max()))
    else:
        PAR.cosine(f"{tag}/grad_f_y[{ss.numel()} rows]", gf[ss.to(DEV)], fsub.grad, 0.999)
        PAR.close_peak(f"{tag}/grad_f_y[{ss.numel()} rows]", gf[ss.to(DEV)], fsub.grad, 1e-2)     # achieved 2.8e-3 (r4_ad_parity)

    # ---- backward, weight / bias gradients: all E edges through the oracle's kernel MLP in fp64, chunked -------------------
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd64.items()}
    f64, d64 = f_y.double(), dout.double() / deg_dst.clamp(min=1).double()[:, None]
    for lo in range(0, src.numel(), 500_000):
        e_s, e_d = src[lo:lo + 500_000], dst[lo:lo + 500_000]
        kern = orc.channel_mlp(leaves, "channel_mlp.", torch.cat([y_c[e_s], x_c[e_d]], dim=1).double())
        (kern * f64[e_s] * d64[e_d]).sum().backward()
    for i in range(nh + 1):
        for nm, got in (("weight", gw[i]), ("bias", gb[i])):
            r = leaves[f"channel_mlp.fcs.{i}.{nm}"].grad
            if precision == "fp32":
                PAR.close(f"{tag}/grad_{nm}{i}", got, r, 1e-3, 2e-5 * float(r.abs().max()))
            else:
                PAR.cosine(f"{tag}/grad_{nm}{i}", got, r, 0.999)
                PAR.close_peak(f"{tag}/grad_{nm}{i}", got, r, 2e-2)     # achieved <= 6.3e-3 (bias), 5.3e-3 (weights)


@pytest.fixture(scope="module")
def sample8m():
    """BASELINE configs[4]-sized geometry: 8 000 000 points, 64 M edges per direction (knn k = 8 and its flip)"""
    from gaot_3d_amd.data import make_synthetic_sample
    batch, tokens = make_synthetic_sample(8_000_000, LATENT, k=KNN, seed=1, device=DEV, out_channels=4)
    yield batch, tokens.to(DEV)
    del batch
    torch.cuda.empty_cache()


@pytest.mark.parametrize("side,nh", [("encoder", 3), ("decoder", 2)])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_gno_8m_point_graph_vs_oracle(sample8m, precision, side, nh):
    """the fused GNO forward / backward on the 64 M-edge graphs of an 8 M-point sample (BASELINE configs[4]; VERDICT r4: that
    size was exercised only through properties and sharded == unsharded) against the oracle on SAMPLED rows: forward on the
    sub-graph of >= 1 000 (encoder: 400) query rows; grad f_y on every edge of >= 1 000 source rows (full dout); weight / bias gradients
    from a second backward whose dout is zero outside >= 4 000 (encoder: 1 500) sampled query rows -- the kernels still walk all 64 M edges
    (row pointers and gathers beyond 2^31 bytes), the oracle needs the sampled rows' edges only (fp64)."""
    from gaot_3d_amd import ops
    batch, tokens = sample8m
    n, m = batch.pos.shape[0], tokens.shape[0]
    if side == "encoder":
        ei, y_pos, x_pos, n_src, n_dst = batch.encoder_edge_index_s0, batch.pos, tokens, n, m
    else:
        ei, y_pos, x_pos, n_src, n_dst = batch.decoder_edge_index_s0, tokens, batch.pos, m, n
    gen = torch.Generator().manual_seed(78)
    sd = _mlp_sd(nh, 15 + nh)
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(nh + 1)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(nh + 1)]
    f_y = torch.randn(n_src, 32, generator=gen)
    dout = torch.randn(n_dst, 32, generator=gen)
    graph = ops.build_graph(ei, n_src, n_dst)
    prec = 0 if precision == "fp32" else 1
    f_d = f_y.to(DEV)
    out = ops.gno_forward(ws, bs, y_pos, x_pos, f_d, graph, precision=prec)
    gf, _, _ = ops.gno_backward(ws, bs, y_pos, x_pos, f_d, dout.to(DEV), graph, precision=prec)
    ei_c = ei.cpu().long()
    src, dst = ei_c[0], ei_c[1]
    deg_dst = torch.bincount(dst, minlength=n_dst)
    deg_src = torch.bincount(src, minlength=n_src)
    # sampled row counts by degree: the encoder's query rows (latent tokens) have ~490 edges each, the decoder's 8 -- the oracle's cost
    # is the EDGES of the sampled rows (round 6: 49 s -> ~20 s per encoder case; same assertions)
    heavy = float(deg_dst.float().mean()) > 64
    n_w, n_s = (1500, 400) if heavy else (4000, 1000)
    qw = _sample_rows(deg_dst, n_w, gen)                    # query rows that carry the masked dout
    dmask = torch.zeros_like(dout)
    dmask[qw] = dout[qw]
    _, gw, gb = ops.gno_backward(ws, bs, y_pos, x_pos, f_d, dmask.to(DEV), graph, precision=prec)
    torch.cuda.synchronize()
    y_c, x_c = y_pos.cpu(), x_pos.cpu()
    tag = f"gno_8m_{side}_nh{nh}_{precision}"
    print(f"[parity] {tag}: E={ei.shape[1]} max query degree {int(deg_dst.max())}, empty query rows {int((deg_dst == 0).sum())}, "
          f"max source degree {int(deg_src.max())}")
    assert ei.shape[1] == 64_000_000

    qs = _sample_rows(deg_dst, n_s, gen)
    local = torch.full((n_dst,), -1, dtype=torch.long)
    local[qs] = torch.arange(qs.numel())
    sel = local[dst] >= 0
    ref = orc.integral_transform(sd, "", y_c, x_c[qs], torch.stack([src[sel], local[dst[sel]]]), f_y)
    if precision == "fp32":
        PAR.close(f"{tag}/out[{qs.numel()} rows]", out[qs.to(DEV)], ref, 1e-4, 1e-5 * float(ref.abs().max()))
    else:
        PAR.close_peak(f"{tag}/out[{qs.numel()} rows]", out[qs.to(DEV)], ref, 1e-2, rel_l2=1e-2)

    sd64 = {k: v.double() for k, v in sd.items()}
    ss = _sample_rows(deg_src, 1000, gen)
    local = torch.full((n_src,), -1, dtype=torch.long)
    local[ss] = torch.arange(ss.numel())
    sel = local[src] >= 0
    es, ed = src[sel], dst[sel]
    fsub = f_y[ss].double().requires_grad_(True)
    kern = orc.channel_mlp(sd64, "channel_mlp.", torch.cat([y_c[es], x_c[ed]], dim=1).double())
    (kern * fsub[local[es]] * (dout[ed].double() / deg_dst[ed].clamp(min=1)[:, None])).sum().backward()
    if precision == "fp32":
        PAR.close(f"{tag}/grad_f_y[{ss.numel()} rows]", gf[ss.to(DEV)], fsub.grad, 1e-3, 1e-5 * float(fsub.grad.abs().max()))
    else:
        PAR.cosine(f"{tag}/grad_f_y[{ss.numel()} rows]", gf[ss.to(DEV)], fsub.grad, 0.999)
        PAR.close_peak(f"{tag}/grad_f_y[{ss.numel()} rows]", gf[ss.to(DEV)], fsub.grad, 1e-2)

    local = torch.full((n_dst,), -1, dtype=torch.long)
    local[qw] = torch.arange(qw.numel())
    sel = local[dst] >= 0
    es, ed = src[sel], dst[sel]
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd64.items()}
    kern = orc.channel_mlp(leaves, "channel_mlp.", torch.cat([y_c[es], x_c[ed]], dim=1).double())
    (kern * f_y[es].double() * (dout[ed].double() / deg_dst[ed].clamp(min=1).double()[:, None])).sum().backward()
    for i in range(nh + 1):
        for nm, got in (("weight", gw[i]), ("bias", gb[i])):
            r = leaves[f"channel_mlp.fcs.{i}.{nm}"].grad
            if precision == "fp32":
                PAR.close(f"{tag}/grad_{nm}{i}[{qw.numel()} query rows]", got, r, 1e-3, 2e-5 * float(r.abs().max()))
            else:
                PAR.cosine(f"{tag}/grad_{nm}{i}[{qw.numel()} query rows]", got, r, 0.999)
                PAR.close_peak(f"{tag}/grad_{nm}{i}[{qw.numel()} query rows]", got, r, 2e-2)


@pytest.mark.parametrize("side", ["encoder_knn", "decoder_knn", "encoder_bidirectional"])
def test_geoembed_statistics_full_size_vs_oracle(sample, side):
    """Statistical GeoEmbed features at configs[1] size against the oracle's restatement of geoembed.py:99-182 on the WHOLE
    graph (z-score over all rows included): token rows of the knn encoder graph (Q = 131 072, ~30 points each), the decoder
    side (Q = N = 500 000 points with 8 tokens each) and the reference yaml's bidirectional encoder graph (token rows with
    hundreds of points next to empty ones -- where the one-sweep kernel's covariance E[uu^T] - E[u]E[u]^T from fp64 moments
    would show cancellation if it had any).  Both device forms are checked: the one-sweep moment form the model uses and the
    two-sweep form of the point-sharded decoder side."""
    from gaot_3d_amd import ops
    batch, tokens = sample
    pos, lat = batch.pos.cpu(), tokens.cpu()
    if side == "encoder_knn":
        src, qry, ei = pos, lat, batch.encoder_edge_index_s0.cpu().long()
    elif side == "decoder_knn":
        src, qry, ei = lat, pos, batch.decoder_edge_index_s0.cpu().long()
    else:
        from gaot_3d_amd.model.layers.magno import get_neighbor_strategy
        lat_b = torch.zeros(tokens.shape[0], dtype=torch.long, device=DEV)
        ei = get_neighbor_strategy("bidirectional", batch.pos, batch.batch, tokens, lat_b, 0.033, 1, False,
                                   latent_dims=(64, 64, 32)).cpu().long()
        src, qry = pos, lat
    deg = torch.bincount(ei[1], minlength=qry.shape[0])
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref = orc.geoembed_stat_features(src, qry, ei)                               # the reference's own fp32 arithmetic
    ref64 = orc.geoembed_stat_features(src.double(), qry.double(), ei).float()     # the same restatement evaluated in fp64
    g = ops.build_graph(ei.to(DEV), src.shape[0], qry.shape[0])
    one = ops.geoembed_from_moments(ops.geoembed_moments(src.to(DEV), qry.to(DEV), g))
    two = ops.geoembed_stats_sharded_queries(src.to(DEV), qry.to(DEV), g, None, qry.shape[0])
    torch.cuda.synchronize()
    print(f"[parity] geoembed_full/{side}: {qry.shape[0]} rows, {ei.shape[1]} edges, degree max {int(deg.max())}, "
          f"empty rows {int((deg == 0).sum())}")
    heavy = int(deg.argmax())
    # the reference sums thousands of fp32 terms per row and forms E[d^2] - E[d]^2 in fp32; the kernels accumulate in fp64.
    # Bar: within 5e-4 of the feature peak of the fp32 oracle, and closer to the fp64 evaluation than the fp32 oracle itself is
    noise = float((ref - ref64).abs().max())
    print(f"[parity] geoembed_full/{side}: fp32 oracle vs its fp64 evaluation max_abs={noise:.3e}")
    for name, got in (("one_sweep", one), ("two_sweep", two)):
        PAR.close_peak(f"geoembed_full/{side}/{name} vs fp32 oracle", got, ref, 5e-4)
        PAR.close_peak(f"geoembed_full/{side}/{name} vs fp64 oracle", got, ref64, 5e-5)
        assert float((got.cpu() - ref64).abs().max()) <= noise + 1e-4
        PAR.close_peak(f"geoembed_full/{side}/{name}/heaviest_row", got[heavy], ref64[heavy], 5e-5)
    PAR.close(f"geoembed_full/{side}/one_vs_two_sweep", one, two, 1e-4, 1e-4)


def _fullsize_oracle_skip_reason():
    """one oracle step at 500 000 points takes ~2 minutes of 64 host threads and ~90 GB of host memory: run by default on a
    host that has them (GAOT_FULLSIZE_ORACLE=1 forces, =0 skips)"""
    flag = os.environ.get("GAOT_FULLSIZE_ORACLE")
    if flag == "1":
        return None
    if flag == "0":
        return "GAOT_FULLSIZE_ORACLE=0"
    try:
        import psutil
        ram = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        return "host memory unknown (psutil missing)"
    cores = os.cpu_count() or 1
    if ram < 110 or cores < 24:
        return (f"whole-step oracle needs >= 110 GB of free host memory and >= 24 cores, this host has {ram:.0f} GB / {cores}; "
                f"recorded result: profiles/archive/r2_l_parity_fullsize_model_vs_oracle.txt")
    return None


def _skip_or_fail_fullsize(case, why):
    """the two whole-step oracle tests are the most valuable parity tests of the suite: a host too small for them must not drop
    them silently -- the skip is printed as a `[parity] ... skipped` line (tests/conftest.py copies those into the parity log and
    the terminal summary), and GAOT_REQUIRE_FULLSIZE=1 turns it into a failure"""
    print(f"[parity] {case}: skipped -- {why}")
    if os.environ.get("GAOT_REQUIRE_FULLSIZE") == "1":
        pytest.fail(f"GAOT_REQUIRE_FULLSIZE=1 and {case} cannot run: {why}")
    pytest.skip(why)


def test_model_full_size_vs_oracle():
    """The WHOLE configs[1] step -- 500 000 points, 4 M edges per direction, 16 384 tokens, L = 10, RoPE, attention dropout
    off -- against the oracle (CPU restatement of the reference, fp32) on the same sample and weights: predictions, loss and
    every parameter gradient, for the fp32 kernels (tight) and the bf16 kernels (bf16 bar).  The last projection is scaled
    so that predictions are O(1) and the loss depends on them."""
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model import init_model
    import bench
    import parity as PAR
    why = _fullsize_oracle_skip_reason()
    if why is not None:
        _skip_or_fail_fullsize("test_model_full_size_vs_oracle", why)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    latent, n, k, layers = (64, 64, 32), 500000, 8, 10
    cfg = bench.model_config(latent, layers, k, 0.0)
    torch.manual_seed(0)
    model = init_model(6, 1, "gaot_3d", cfg)
    batch, tokens = make_synthetic_sample(n, latent, k=k, seed=0)
    sd = {kk: v.clone() for kk, v in model.state_dict().items()}
    with torch.no_grad():
        p0 = orc.gaot3d_forward(sd, cfg, batch, tokens)
    last = model.decoder.projection.fcs[-1]
    PAR.unit_scale_last_layer(last.weight, last.bias, float(p0.std()))
    sd = {kk: v.clone() for kk, v in model.state_dict().items()}
    pred_r, loss_r, grads_r = orc.train_step_grads(sd, cfg, batch, tokens)
    print(f"[parity] full-size oracle: loss={float(loss_r):.6f} pred std={float(pred_r.std()):.4f}")
    bd, tk = batch.to(DEV), tokens.to(DEV)
    for precision in ("fp32", "bf16"):
        gaot_3d_amd.set_precision(precision)
        try:
            m = init_model(6, 1, "gaot_3d", cfg)
            m.load_state_dict(sd)
            m = m.to(DEV).eval().train()
            gaot_3d_amd.clear_graph_cache(bd)
            pred = m(batch=bd, tokens_pos=tk)
            loss = GF.mse_loss(pred, bd.x)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            gaot_3d_amd.set_precision("fp32")
        grads = {kk: p.grad for kk, p in m.named_parameters() if p.requires_grad and p.grad is not None}
        if precision == "fp32":
            PAR.close_peak("fullsize_vs_oracle_fp32/pred", pred, pred_r, 2e-4, rel_l2=1e-4)
            PAR.close("fullsize_vs_oracle_fp32/loss", loss, loss_r, 1e-5, 0.0)
            PAR.grads_cosine("fullsize_vs_oracle_fp32/grads", grads, grads_r, 0.99999, per_tensor=0.9999)
        else:
            # achieved (profiles/archive/r4_ad_parity.txt): pred 3.2e-3 of peak, rel_l2 1.4e-3, loss 2.3e-3, gradient cosine 0.999991
            PAR.close_peak("fullsize_vs_oracle_bf16/pred", pred, pred_r, 1e-2, rel_l2=5e-3)
            PAR.close("fullsize_vs_oracle_bf16/loss", loss, loss_r, 7e-3, 0.0)
            PAR.grads_cosine("fullsize_vs_oracle_bf16/grads", grads, grads_r, 0.9999, per_tensor=0.99)
        del m, pred, loss, grads
        torch.cuda.empty_cache()


def test_configs3_full_size_vs_oracle():
    """The WHOLE configs[3] step (only operator-by-operator before) -- 500 000 points, radius-graph encoder capped
    at 32 points per token, bidirectional decoder (variable degree, empty rows), statistical GeoEmbed on BOTH sides, 5 input
    channels (pos + [Mach, AOA]), L = 10, RoPE, attention dropout off -- against the oracle (CPU restatement of the reference,
    fp32; geoembed.py:99-182, magno.py:468-600, 691-798) on the same sample, graphs and weights: predictions, loss and every
    parameter gradient, fp32 kernels 2e-4 of peak / gradient cosine 0.99999, bf16 kernels at the bf16 bar."""
    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd.model import init_model
    import bench
    import parity as PAR
    why = _fullsize_oracle_skip_reason()
    if why is not None:
        _skip_or_fail_fullsize("test_configs3_full_size_vs_oracle", why)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    latent, n, k, layers = (64, 64, 32), 500000, 8, 10
    cfg = bench.model_config(latent, layers, k, 0.0, "cfg3")
    torch.manual_seed(0)
    model = init_model(5, 1, "gaot_3d", cfg)
    bd, tk = bench.make_workload_sample("cfg3", n, latent, k, 0, DEV)       # graphs built by the device kernels
    batch, tokens = bd.to("cpu"), tk.cpu()
    print(f"[parity] configs[3] graphs: encoder {batch.encoder_edge_index_s0.shape[1]} edges, decoder "
          f"{batch.decoder_edge_index_s0.shape[1]} edges")
    sd = {kk: v.clone() for kk, v in model.state_dict().items()}
    with torch.no_grad():
        p0 = orc.gaot3d_forward(sd, cfg, batch, tokens)
    last = model.decoder.projection.fcs[-1]
    PAR.unit_scale_last_layer(last.weight, last.bias, float(p0.std()))
    sd = {kk: v.clone() for kk, v in model.state_dict().items()}
    pred_r, loss_r, grads_r = orc.train_step_grads(sd, cfg, batch, tokens)
    print(f"[parity] configs[3] full-size oracle: loss={float(loss_r):.6f} pred std={float(pred_r.std()):.4f}")
    for precision in ("fp32", "bf16"):
        gaot_3d_amd.set_precision(precision)
        try:
            m = init_model(5, 1, "gaot_3d", cfg)
            m.load_state_dict(sd)
            m = m.to(DEV).eval().train()
            gaot_3d_amd.clear_graph_cache(bd)
            pred = m(batch=bd, tokens_pos=tk)
            loss = GF.mse_loss(pred, bd.x)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            gaot_3d_amd.set_precision("fp32")
        grads = {kk: p.grad for kk, p in m.named_parameters() if p.requires_grad and p.grad is not None}
        if precision == "fp32":
            PAR.close_peak("configs3_fullsize_vs_oracle_fp32/pred", pred, pred_r, 2e-4, rel_l2=1e-4)
            PAR.close("configs3_fullsize_vs_oracle_fp32/loss", loss, loss_r, 1e-5, 0.0)
            PAR.grads_cosine("configs3_fullsize_vs_oracle_fp32/grads", grads, grads_r, 0.99999, per_tensor=0.9999)
        else:
            # achieved: pred 3.5e-3 of peak, rel_l2 7.2e-4, loss 3.7e-4, gradient cosine 0.999999
            PAR.close_peak("configs3_fullsize_vs_oracle_bf16/pred", pred, pred_r, 1e-2, rel_l2=3e-3)
            PAR.close("configs3_fullsize_vs_oracle_bf16/loss", loss, loss_r, 2e-3, 0.0)
            PAR.grads_cosine("configs3_fullsize_vs_oracle_bf16/grads", grads, grads_r, 0.9999, per_tensor=0.99)
        del m, pred, loss, grads
        torch.cuda.empty_cache()
