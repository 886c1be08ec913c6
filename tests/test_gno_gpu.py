"""GPU parity: CSR builder and the fused GNO integral transform (fwd + bwd) through the C ABI,
against the CPU oracle (pinned to the reference by tests/test_oracle_golden.py) and the committed
golden vectors.  Tolerances: outputs rtol 1e-4 / atol 1e-5, grads rtol 1e-3 / atol 1e-5."""
import os
import sys

import pytest
import torch

import golden_io as gio

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import gaot_oracle as orc  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def report(name, a, b):
    err = (a - b).abs().max().item() if a.numel() else 0.0
    rel = err / (b.abs().max().item() + 1e-30) if a.numel() else 0.0
    print(f"[parity] {name}: max_abs={err:.3e} max_rel_to_peak={rel:.3e}")
    return err


def close(name, a, b, rtol, atol):
    a = a.detach().cpu()
    b = b.detach().cpu()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    report(name, a, b)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{name}: max abs err {(a - b).abs().max().item():.3e}"


def rand_graph(n_src, n_dst, e, seed, empty_rows=True, heavy=True, dtype=torch.int64):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n_src, (e,), generator=g)
    dst = torch.randint(0, n_dst, (e,), generator=g)
    if empty_rows and e > 0:
        m = dst % 7 == 3
        dst[m] = (dst[m] + 1) % n_dst                          # some rows never hit
    if heavy and e > 0:
        dst[: e // 10] = n_dst // 2                            # one row with >> 32 edges
    return torch.stack([src, dst]).to(dtype)


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("e,nr", [(0, 5), (1, 1), (1000, 37), (50000, 4096), (200000, 70000), (5000, 20_000_000)])
def test_csr_build(dtype, e, nr):
    from gaot_3d_amd import ops
    ei = rand_graph(max(nr, 1) * 2, nr, e, seed=e + nr, dtype=dtype)
    for sort_row, rows in ((1, nr), (0, max(nr, 1) * 2)):
        s = ops.csr_build(ei.to(DEV), sort_row, rows)
        torch.cuda.synchronize()
        key = ei[sort_row].long()
        order = torch.sort(key, stable=True).indices
        cnt = torch.bincount(key, minlength=rows)
        rowptr = torch.zeros(rows + 1, dtype=torch.long)
        rowptr[1:] = torch.cumsum(cnt, 0)
        assert torch.equal(s.rowptr.cpu().long(), rowptr)
        assert torch.equal(s.perm.cpu().long(), order)              # stable: bit-exact permutation
        assert torch.equal(s.key.cpu().long(), key[order])
        assert torch.equal(s.other.cpu().long(), ei[1 - sort_row].long()[order])


@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("e,nr", [(1, 4), (1000, 37), (50000, 4096), (200000, 70000)])
def test_csr_build_sorted_input(dtype, e, nr):
    """neighbour lists that arrive grouped by query (what a neighbour search emits) take the streaming path:
    same bit-exact result as the counting sort, including empty rows before, between and after the keys"""
    from gaot_3d_amd import ops
    ei = rand_graph(max(nr, 1) * 2, nr, e, seed=e + nr + 1, dtype=dtype)
    order0 = torch.sort(ei[1], stable=True).indices
    ei = ei[:, order0].contiguous()                                  # sorted by row 1 (query)
    if nr > 8:
        ei[1] = torch.clamp(ei[1], 3, nr - 3)                         # empty rows at both ends
    for sort_row, rows in ((1, nr), (0, max(nr, 1) * 2)):            # sort_row 0 stays unsorted: general path
        s = ops.csr_build(ei.to(DEV), sort_row, rows)
        torch.cuda.synchronize()
        key = ei[sort_row].long()
        order = torch.sort(key, stable=True).indices
        cnt = torch.bincount(key, minlength=rows)
        rowptr = torch.zeros(rows + 1, dtype=torch.long)
        rowptr[1:] = torch.cumsum(cnt, 0)
        assert torch.equal(s.rowptr.cpu().long(), rowptr)
        assert torch.equal(s.perm.cpu().long(), order)
        assert torch.equal(s.key.cpu().long(), key[order])
        assert torch.equal(s.other.cpu().long(), ei[1 - sort_row].long()[order])


def _mlp_sd(layers, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i in range(len(layers) - 1):
        bound = 1.0 / layers[i] ** 0.5
        sd[f"channel_mlp.fcs.{i}.weight"] = (torch.rand(layers[i + 1], layers[i], generator=g) * 2 - 1) * bound
        sd[f"channel_mlp.fcs.{i}.bias"] = (torch.rand(layers[i + 1], generator=g) * 2 - 1) * bound
    return sd


def _run_gno(sd, y_pos, x_pos, ei, f_y, w_out):
    from gaot_3d_amd import ops
    n = len([k for k in sd if k.endswith("weight")])
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(n)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(n)]
    g = ops.build_graph(ei.to(DEV), y_pos.shape[0], x_pos.shape[0])
    yd, xd, fd = y_pos.to(DEV), x_pos.to(DEV), f_y.to(DEV)
    out = ops.gno_forward(ws, bs, yd, xd, fd, g)
    gf, gw, gb = ops.gno_backward(ws, bs, yd, xd, fd, w_out.to(DEV), g)
    torch.cuda.synchronize()
    return out, gf, gw, gb


def _oracle_gno(sd, y_pos, x_pos, ei, f_y, w_out):
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    f = f_y.clone().requires_grad_(True)
    out = orc.integral_transform(sdr, "", y_pos, x_pos, ei, f)
    (out * w_out).sum().backward()
    return out.detach(), f.grad, sdr


def _compare(sd, y_pos, x_pos, ei, f_y, w_out, tag):
    out, gf, gw, gb = _run_gno(sd, y_pos, x_pos, ei, f_y, w_out)
    ro, rgf, sdr = _oracle_gno(sd, y_pos, x_pos, ei, f_y, w_out)
    close(f"{tag}/out", out, ro, 1e-4, 1e-5)
    close(f"{tag}/grad_f", gf, rgf, 1e-3, 1e-5)
    for i in range(len(gw)):
        rw = sdr[f"channel_mlp.fcs.{i}.weight"].grad
        rb = sdr[f"channel_mlp.fcs.{i}.bias"].grad
        scale = max(1.0, rw.abs().max().item())
        close(f"{tag}/grad_w{i}", gw[i] / scale, rw / scale, 1e-3, 1e-5)
        close(f"{tag}/grad_b{i}", gb[i] / scale, rb / scale, 1e-3, 1e-5)


def test_gno_golden_variable_degree():
    """IntegralTransform 'linear' golden captured from the reference (empty rows + a >32-degree row)."""
    meta, g = gio.load("ops")
    tag = "it_linear_noattn"
    sd = gio.sub(g["sd"], tag)
    pos, lat, ei = g["in"]["pos"], g["in"]["lat"], g["in"]["edge_index"]
    w = g["in"][f"{tag}/w"]
    out, gf, gw, gb = _run_gno(sd, pos, lat, ei, g["in"]["f_y"], w)
    close("golden/out", out, g["out"][f"{tag}/out"], 1e-4, 1e-5)
    close("golden/grad_f", gf, g["grad"][f"{tag}/f_y"], 1e-3, 1e-5)
    gg = gio.sub(g["grad"], tag)
    for i in range(len(gw)):
        close(f"golden/grad_w{i}", gw[i], gg[f"channel_mlp.fcs.{i}.weight"], 1e-3, 1e-5)
        close(f"golden/grad_b{i}", gb[i], gg[f"channel_mlp.fcs.{i}.bias"], 1e-3, 1e-5)


def test_gno_four_hidden_layers_forward_only():
    """NH=4: forward is supported; the exact-fp32 backward would need 172 KB of LDS per workgroup and must refuse loudly (the
    bf16 backward takes four hidden layers: test_gno_bf16_backward[4])"""
    from gaot_3d_amd import ops
    from gaot_3d_amd._lib import GaotError
    gen = torch.Generator().manual_seed(4)
    ei = rand_graph(500, 90, 3000, seed=14, dtype=torch.int32)
    y, x = torch.rand(500, 3, generator=gen), torch.rand(90, 3, generator=gen)
    f, w = torch.randn(500, 32, generator=gen), torch.randn(90, 32, generator=gen)
    sd = _mlp_sd([6, 64, 64, 64, 64, 32], seed=4)
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(5)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(5)]
    g = ops.build_graph(ei.to(DEV), 500, 90)
    out = ops.gno_forward(ws, bs, y.to(DEV), x.to(DEV), f.to(DEV), g)
    close("nh4/out", out, orc.integral_transform(sd, "", y, x, ei, f), 1e-4, 1e-5)
    with pytest.raises(GaotError):
        ops.gno_backward(ws, bs, y.to(DEV), x.to(DEV), f.to(DEV), w.to(DEV), g, precision=0)


@pytest.mark.parametrize("nh", [1, 2, 3])
def test_gno_random_graph(nh):
    gen = torch.Generator().manual_seed(nh)
    n_src, n_dst, e = 3000, 700, 20011
    ei = rand_graph(n_src, n_dst, e, seed=10 + nh, dtype=torch.int32)
    y = torch.rand(n_src, 3, generator=gen) * 2 - 1
    x = torch.rand(n_dst, 3, generator=gen) * 2 - 1
    f = torch.randn(n_src, 32, generator=gen)
    w = torch.randn(n_dst, 32, generator=gen)
    sd = _mlp_sd([6] + [64] * nh + [32], seed=nh)
    _compare(sd, y, x, ei, f, w, f"rand_nh{nh}")


def test_gno_knn_fixed_degree_and_flip():
    """encoder-style knn graph (phys-major) and its flip (decoder): fixed degree on one side."""
    from gaot_3d_amd.data import knn_edges_bruteforce, latent_grid
    gen = torch.Generator().manual_seed(5)
    lat = latent_grid((6, 6, 5))
    pos = torch.rand(2000, 3, generator=gen) * 2 - 1
    enc = knn_edges_bruteforce(pos, lat, 8).to(torch.int32)
    f = torch.randn(2000, 32, generator=gen)
    _compare(_mlp_sd([6, 64, 64, 64, 32], 1), pos, lat, enc, f, torch.randn(lat.shape[0], 32, generator=gen), "enc_knn")
    fl = torch.randn(lat.shape[0], 32, generator=gen)
    _compare(_mlp_sd([6, 64, 64, 32], 2), lat, pos, enc.flip(0).contiguous(), fl,
             torch.randn(2000, 32, generator=gen), "dec_flip")


def test_gno_empty_and_tiny():
    sd = _mlp_sd([6, 64, 32], 3)
    y = torch.rand(5, 3)
    x = torch.rand(4, 3)
    f = torch.randn(5, 32)
    w = torch.randn(4, 32)
    out, gf, gw, gb = _run_gno(sd, y, x, torch.zeros(2, 0, dtype=torch.int64), f, w)
    assert out.abs().max().item() == 0.0 and gf.abs().max().item() == 0.0
    assert all(t.abs().max().item() == 0.0 for t in gw + gb)
    _compare(sd, y, x, torch.tensor([[0, 4, 2], [1, 1, 3]]), f, w, "tiny")


def test_gno_deterministic():
    """no float atomics anywhere: two runs are bit-identical"""
    gen = torch.Generator().manual_seed(9)
    ei = rand_graph(5000, 900, 60000, seed=3, dtype=torch.int32)
    y = torch.rand(5000, 3, generator=gen)
    x = torch.rand(900, 3, generator=gen)
    f = torch.randn(5000, 32, generator=gen)
    w = torch.randn(900, 32, generator=gen)
    sd = _mlp_sd([6, 64, 64, 64, 32], 4)
    a = _run_gno(sd, y, x, ei, f, w)
    b = _run_gno(sd, y, x, ei, f, w)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for u, v in zip(a[2] + a[3], b[2] + b[3]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("nh", [1, 2, 3, 4])
def test_gno_bf16_forward(nh):
    """bf16 matrix-core variant (hidden/last layers on v_mfma_f32_32x32x16_bf16, layer 0 exact fp32): tolerance of
    bf16 operands -- rtol 2e-2 on the output scale (SURVEY §8d bf16 mode)"""
    from gaot_3d_amd import ops
    gen = torch.Generator().manual_seed(20 + nh)
    n_src, n_dst, e = 3000, 700, 20011
    ei = rand_graph(n_src, n_dst, e, seed=30 + nh, dtype=torch.int32)
    y = torch.rand(n_src, 3, generator=gen) * 2 - 1
    x = torch.rand(n_dst, 3, generator=gen) * 2 - 1
    f = torch.randn(n_src, 32, generator=gen)
    sd = _mlp_sd([6] + [64] * nh + [32], seed=nh)
    n = nh + 1
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(n)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(n)]
    g = ops.build_graph(ei.to(DEV), n_src, n_dst)
    out = ops.gno_forward(ws, bs, y.to(DEV), x.to(DEV), f.to(DEV), g, precision=1).cpu()
    ref = orc.integral_transform(sd, "", y, x, ei, f)
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    print(f"[parity] gno_bf16_fwd_nh{nh}: max_abs={err:.3e} ref_peak={scale:.3e}")
    assert err <= 2e-2 * scale
    out32 = ops.gno_forward(ws, bs, y.to(DEV), x.to(DEV), f.to(DEV), g, precision=0).cpu()
    assert torch.allclose(out32, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("nh", [1, 2, 3, 4])
def test_gno_bf16_backward(nh):
    """bf16 matrix-core backward: gradients against the fp32 oracle -- cosine >= 0.999 and max error <= 3e-2 of the
    gradient's peak (bf16 operand tolerance); graph with empty rows and a >32-degree row on both sides"""
    from gaot_3d_amd import ops
    gen = torch.Generator().manual_seed(40 + nh)
    n_src, n_dst, e = 3000, 700, 20011
    ei = rand_graph(n_src, n_dst, e, seed=50 + nh, dtype=torch.int32)
    ei[0, 100:400] = 17          # a heavy SOURCE row as well (segmented grad_f over > 32 edges)
    y = torch.rand(n_src, 3, generator=gen) * 2 - 1
    x = torch.rand(n_dst, 3, generator=gen) * 2 - 1
    f = torch.randn(n_src, 32, generator=gen)
    w = torch.randn(n_dst, 32, generator=gen)
    sd = _mlp_sd([6] + [64] * nh + [32], seed=nh)
    n = nh + 1
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(n)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(n)]
    g = ops.build_graph(ei.to(DEV), n_src, n_dst)
    gf, gw, gb = ops.gno_backward(ws, bs, y.to(DEV), x.to(DEV), f.to(DEV), w.to(DEV), g, precision=1)
    torch.cuda.synchronize()
    _, rgf, sdr = _oracle_gno(sd, y, x, ei, f, w)

    def check(name, a, b):
        a, b = a.detach().cpu().double().flatten(), b.double().flatten()
        cos = (a * b).sum() / (a.norm() * b.norm() + 1e-30)
        err = (a - b).abs().max().item()
        print(f"[parity] gno_bf16_bwd_nh{nh}/{name}: cosine={cos.item():.6f} max_abs={err:.3e} ref_peak={b.abs().max().item():.3e}")
        assert cos.item() >= 0.999, name
        assert err <= 3e-2 * b.abs().max().item() + 1e-7, name

    check("grad_f", gf, rgf)
    for i in range(n):
        check(f"grad_w{i}", gw[i], sdr[f"channel_mlp.fcs.{i}.weight"].grad)
        check(f"grad_b{i}", gb[i], sdr[f"channel_mlp.fcs.{i}.bias"].grad)


@pytest.mark.parametrize("e", [1, 15, 16, 17, 127, 129, 2049])
def test_gno_bf16_backward_tile_tails(e):
    """the bf16 backward's waves own 16-edge tiles, eight per workgroup iteration: edge counts around those boundaries (one
    edge, a tile minus / plus one, an iteration plus one), a source row that runs through every tile (open on both sides
    of the inner tiles: the partial slots + k_segment_fixup<32, 4>), and two runs bit-identical (no atomics)"""
    from gaot_3d_amd import ops
    gen = torch.Generator().manual_seed(70 + e)
    n_src, n_dst = 40, 23
    ei = rand_graph(n_src, n_dst, e, seed=80 + e, dtype=torch.int32)
    ei[0, : max(1, (2 * e) // 3)] = 7       # one long source row
    y = torch.rand(n_src, 3, generator=gen) * 2 - 1
    x = torch.rand(n_dst, 3, generator=gen) * 2 - 1
    f = torch.randn(n_src, 32, generator=gen)
    w = torch.randn(n_dst, 32, generator=gen)
    nh = 3
    sd = _mlp_sd([6] + [64] * nh + [32], seed=5)
    ws = [sd[f"channel_mlp.fcs.{i}.weight"].to(DEV) for i in range(nh + 1)]
    bs = [sd[f"channel_mlp.fcs.{i}.bias"].to(DEV) for i in range(nh + 1)]
    g = ops.build_graph(ei.to(DEV), n_src, n_dst)
    runs = []
    for _ in range(2):
        gf, gw, gb = ops.gno_backward(ws, bs, y.to(DEV), x.to(DEV), f.to(DEV), w.to(DEV), g, precision=1)
        torch.cuda.synchronize()
        runs.append([gf] + list(gw) + list(gb))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    _, rgf, sdr = _oracle_gno(sd, y, x, ei, f, w)
    refs = [rgf] + [sdr[f"channel_mlp.fcs.{i}.weight"].grad for i in range(nh + 1)] + [sdr[f"channel_mlp.fcs.{i}.bias"].grad for i in range(nh + 1)]
    for i, (a, b) in enumerate(zip(runs[0], refs)):
        a, b = a.detach().cpu().double(), b.double()
        err = (a - b).abs().max().item()
        print(f"[parity] gno_bf16_bwd_tails_e{e}/{i}: max_abs={err:.3e} ref_peak={b.abs().max().item():.3e}")
        assert err <= 3e-2 * b.abs().max().item() + 1e-6, (e, i)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_integral_transform_other_shapes_golden(precision):
    """The reference's default shapes -- lifting_channels 16, gno_coord_dim 2 (magno.py:25,28) -- 64 and 48 channels, coord
    dim 1 and a four-hidden-layer kernel MLP, against goldens captured from the reference's IntegralTransform
    (tests/golden/gno_shapes.npz).  All but the four-hidden-layer case in fp32 mode (general path when gradients are wanted) run
    the FUSED kernels through exact zero padding / 32-channel passes (IntegralTransform._fused_plan)."""
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    meta, g = gio.load("gno_shapes")
    ei = g["in"]["edge_index"].to(DEV)
    gaot_3d_amd.set_precision(precision)
    try:
        for v in meta["variants"]:
            tag, cd, layers = v["tag"], v["coord_dim"], v["layers"]
            it = IntegralTransform(channel_mlp_layers=layers, transform_type="linear", coord_dim=cd)
            it.load_state_dict(gio.sub(g["sd"], tag), strict=True)
            it = it.to(DEV).train()
            y, x = g["in"]["pos3"][:, :cd].contiguous().to(DEV), g["in"]["lat3"][:, :cd].contiguous().to(DEV)
            f = g["in"][f"{tag}/f_y"].to(DEV).requires_grad_(True)
            plan = it._fused_plan(list(it.channel_mlp.fcs), f, y)
            # four hidden layers with gradients: general per-edge path in fp32 mode, fused in bf16 mode (fragments from L2)
            assert (plan is None) == (len(layers) == 6 and precision == "fp32"), (tag, plan)
            out = it(y_pos=y, x_pos=x, edge_index=ei, f_y=f)
            (out * g["in"][f"{tag}/w"].to(DEV)).sum().backward()
            torch.cuda.synchronize()
            ref = g["out"][f"{tag}/out"]
            if precision == "fp32":
                close(f"gno_shapes/{tag}/out", out, ref, 1e-4, 1e-5)
                close(f"gno_shapes/{tag}/grad_f", f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-5)
                for k, p in it.named_parameters():
                    close(f"gno_shapes/{tag}/grad_{k}", p.grad, g["grad"][f"{tag}/{k}"], 1e-3, 2e-5)
            else:
                peak = float(ref.abs().max())
                assert report(f"gno_shapes_bf16/{tag}/out", out.detach().cpu(), ref) <= 3e-2 * peak
                a = torch.cat([p.grad.flatten().cpu().double() for _, p in it.named_parameters()] + [f.grad.flatten().cpu().double()])
                r = torch.cat([g["grad"][f"{tag}/{k}"].flatten().double() for k, _ in it.named_parameters()] +
                              [g["grad"][f"{tag}/f_y"].flatten().double()])
                cos = float(a @ r / (a.norm() * r.norm()))
                print(f"[parity] gno_shapes_bf16/{tag}/grads cosine={cos:.6f}")
                assert cos >= 0.999
    finally:
        gaot_3d_amd.set_precision("fp32")


@pytest.mark.gpu
def test_flipped_decoder_graph_shares_the_encoder_lists():
    """the shipped configuration's decoder edge list is the encoder's with its rows swapped (a separate tensor): sorted by query
    it IS the encoder's list sorted by source and vice versa, so `graph_for` hands the decoder the encoder's two lists instead of
    two more radix-sort builds -- identical arrays to a build of its own, and a list that is NOT the flip still gets its own"""
    from gaot_3d_amd import ops
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model.layers.integral_transform import graph_for
    batch, tokens = make_synthetic_sample(20000, (8, 8, 4), k=5, seed=3, device=DEV)
    n, m = batch.pos.shape[0], tokens.shape[0]
    enc, dec = batch.encoder_edge_index_s0, batch.decoder_edge_index_s0
    assert dec.data_ptr() != enc.data_ptr() and torch.equal(dec, enc.flip(0))
    ge = graph_for(enc, n, m, batch, ("enc", 0))
    gd = graph_for(dec, m, n, batch, ("dec", 0))
    assert gd.by_dst is ge.by_src and gd.by_src is ge.by_dst and (gd.num_src, gd.num_dst) == (m, n)
    own = ops.build_graph(dec, m, n)
    for a, b in ((gd.by_dst, own.by_dst), (gd.by_src, own.by_src)):
        assert a.num_rows == b.num_rows
        for f in ("rowptr", "perm", "key", "other"):
            assert torch.equal(getattr(a, f), getattr(b, f)), f
    import gaot_3d_amd
    gaot_3d_amd.clear_graph_cache(batch)            # the lists go, the fact about the inputs stays
    assert "_gaot_graphs" not in batch.__dict__ and batch.__dict__["_gaot_flip"][0][4] is True
    other = dec.clone()
    other[:, [0, 7]] = other[:, [7, 0]]             # same edge set, another order: not the flip -> its own build
    batch.decoder_edge_index_s0 = other
    ge = graph_for(enc, n, m, batch, ("enc", 0))
    g2 = graph_for(other, m, n, batch, ("dec", 0))
    assert g2.by_dst is not ge.by_src and torch.equal(g2.by_dst.rowptr, own.by_dst.rowptr)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_integral_transform_hidden_widths_golden(precision):
    """kernel-MLP hidden widths other than 64 (`in_/out_gno_channel_mlp_hidden_layers` are free lists, magno.py:32,36) against
    goldens captured from the reference's IntegralTransform (tests/golden/gno_hidden.npz): widths <= 64 -- 32 / 32, 48 / 64 / 16,
    a single layer of 8 -- run the FUSED kernels through exact zero padding to 64 (gelu(0) = 0, zero rows / columns in the
    neighbouring weights), 128 takes the general per-edge path"""
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    meta, g = gio.load("gno_hidden")
    ei = g["in"]["edge_index"].to(DEV)
    gaot_3d_amd.set_precision(precision)
    try:
        for v in meta["variants"]:
            tag, cd, layers = v["tag"], v["coord_dim"], v["layers"]
            it = IntegralTransform(channel_mlp_layers=layers, transform_type="linear", coord_dim=cd)
            it.load_state_dict(gio.sub(g["sd"], tag), strict=True)
            it = it.to(DEV).train()
            y, x = g["in"]["pos3"][:, :cd].contiguous().to(DEV), g["in"]["lat3"][:, :cd].contiguous().to(DEV)
            f = g["in"][f"{tag}/f_y"].to(DEV).requires_grad_(True)
            plan = it._fused_plan(list(it.channel_mlp.fcs), f, y)
            assert (plan is None) == (max(layers[1:-1]) > 64), (tag, plan)
            out = it(y_pos=y, x_pos=x, edge_index=ei, f_y=f)
            (out * g["in"][f"{tag}/w"].to(DEV)).sum().backward()
            torch.cuda.synchronize()
            ref = g["out"][f"{tag}/out"]
            if precision == "fp32":
                close(f"gno_hidden/{tag}/out", out, ref, 1e-4, 1e-5)
                close(f"gno_hidden/{tag}/grad_f", f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-5)
                for k, p in it.named_parameters():
                    close(f"gno_hidden/{tag}/grad_{k}", p.grad, g["grad"][f"{tag}/{k}"], 1e-3, 2e-5)
            else:
                peak = float(ref.abs().max())
                assert report(f"gno_hidden_bf16/{tag}/out", out.detach().cpu(), ref) <= 3e-2 * peak
                a = torch.cat([p.grad.flatten().cpu().double() for _, p in it.named_parameters()] + [f.grad.flatten().cpu().double()])
                r = torch.cat([g["grad"][f"{tag}/{k}"].flatten().double() for k, _ in it.named_parameters()] +
                              [g["grad"][f"{tag}/f_y"].flatten().double()])
                cos = float(a @ r / (a.norm() * r.norm()))
                print(f"[parity] gno_hidden_bf16/{tag}/grads cosine={cos:.6f}")
                assert cos >= 0.999
    finally:
        gaot_3d_amd.set_precision("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_integral_transform_variants_golden(precision):
    """Variant holes of round 3: `use_attn` with coord_dim 2 / 1 (the reference's default is
    gno_coord_dim 2, magno.py:28; scores on `[:, :coord_dim]`, integral_transform.py:126-142) and kernel MLPs with the other
    activations of the reference's `activation_fn` (mlp.py:27-35) -- goldens captured from the reference's IntegralTransform
    (tests/golden/gno_variants.npz).  All run the general per-edge HIP path."""
    import torch.nn.functional as F
    import gaot_3d_amd
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    meta, g = gio.load("gno_variants")
    ei = g["in"]["edge_index"].to(DEV)
    gaot_3d_amd.set_precision(precision)
    try:
        for v in meta["variants"]:
            tag, cd, layers = v["tag"], v["coord_dim"], v["layers"]
            # the reference hands callables around (activation_fn(name)); strings are accepted as well
            act = {"tanh": F.tanh, "leaky_relu": F.leaky_relu, "elu": F.elu, "swish": torch.nn.SiLU(), "none": "none",
                   "gelu": F.gelu}.get(v["act"], v["act"])
            it = IntegralTransform(channel_mlp_layers=layers, channel_mlp_non_linearity=act, transform_type=v["transform_type"],
                                   use_attn=v["use_attn"] or None, coord_dim=cd, attention_type=v["attention_type"])
            it.load_state_dict(gio.sub(g["sd"], tag), strict=True)
            it = it.to(DEV).train()
            y, x = g["in"]["pos3"][:, :cd].contiguous().to(DEV), g["in"]["lat3"][:, :cd].contiguous().to(DEV)
            f = g["in"][f"{tag}/f_y"].to(DEV).requires_grad_(True)
            out = it(y_pos=y, x_pos=x, edge_index=ei, f_y=f)
            (out * g["in"][f"{tag}/w"].to(DEV)).sum().backward()
            torch.cuda.synchronize()
            ref = g["out"][f"{tag}/out"]
            if precision == "fp32":
                close(f"gno_variants/{tag}/out", out, ref, 1e-4, 1e-5)
                close(f"gno_variants/{tag}/grad_f", f.grad, g["grad"][f"{tag}/f_y"], 1e-3, 1e-5)
                for k, p in it.named_parameters():
                    close(f"gno_variants/{tag}/grad_{k}", p.grad, g["grad"][f"{tag}/{k}"], 1e-3, 2e-5)
            else:
                peak = float(ref.abs().max())
                assert report(f"gno_variants_bf16/{tag}/out", out.detach().cpu(), ref) <= 3e-2 * peak
                a = torch.cat([p.grad.flatten().cpu().double() for _, p in it.named_parameters()] + [f.grad.flatten().cpu().double()])
                r = torch.cat([g["grad"][f"{tag}/{k}"].flatten().double() for k, _ in it.named_parameters()] +
                              [g["grad"][f"{tag}/f_y"].flatten().double()])
                cos = float(a @ r / (a.norm() * r.norm()))
                print(f"[parity] gno_variants_bf16/{tag}/grads cosine={cos:.6f}")
                assert cos >= 0.999
    finally:
        gaot_3d_amd.set_precision("fp32")
