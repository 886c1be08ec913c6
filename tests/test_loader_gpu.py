"""GPU checks of the callers' side: the trainer's edge pre-computation pass (`io.enrich_sample`, reference
stat.py:163-214) on the device graph kernels against the host restatement -- at the full 500 000-point size for the knn
configuration, at 20 000 points for the variable-degree strategies (the host restatement is O(N M) memory) -- and the
loader's pinned / overlapped upload and device-resident cache."""
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pairs(e):
    e = e.long()
    return set((e[0] * (1 << 32) + e[1]).tolist())


def test_enrich_sample_full_size_knn_matches_host():
    from gaot_3d_amd import io
    from gaot_3d_amd.data import MeshBatch, knn_edges_grid, latent_grid, rescale, superellipsoid_surface
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    n, dims = 500_000, (64, 64, 32)
    pos, nrm = superellipsoid_surface(n, generator=torch.Generator().manual_seed(1))
    raw = MeshBatch(pos=pos * 1.7 + 0.3, x=torch.zeros(n, 1), c=nrm)          # un-normalised coordinates, as on disk
    cfg = MAGNOConfig(gno_coord_dim=3, neighbor_strategy="knn", k_neighbors=8, gno_radius=0.033, scales=[1.0])
    lat = latent_grid(dims)
    out = io.enrich_sample(raw, lat, cfg, latent_dims=dims, device=DEV)
    enc, dec = out.encoder_edge_index_s0, out.decoder_edge_index_s0
    assert enc.dtype == torch.int32 and enc.shape == (2, 8 * n) and not enc.is_cuda and out.num_latent_nodes == lat.shape[0]
    assert torch.equal(dec, enc.flip(0))
    host = knn_edges_grid(rescale(raw.pos), dims, (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), 8)     # host restatement, O(N 125)
    # ties between equidistant tokens may be broken differently: compare as sets and allow a handful of boundary cases
    a, b = _pairs(enc), _pairs(host)
    assert len(a ^ b) <= 2e-5 * len(b), len(a ^ b)
    assert torch.equal(out.encoder_query_counts_s0, torch.bincount(enc[1].long(), minlength=lat.shape[0]).to(torch.int32))
    assert torch.equal(out.decoder_query_counts_s0, torch.full((n,), 8, dtype=torch.int32))


@pytest.mark.parametrize("strategy", [["radius", "bidirectional"], "bidirectional", ["bidirectional", "reverse"]])
def test_enrich_sample_variable_degree_matches_host(strategy):
    from gaot_3d_amd import io
    from gaot_3d_amd.data import MeshBatch, latent_grid, rescale
    from gaot_3d_amd.model.layers.magno import MAGNOConfig, get_neighbor_strategy, parse_neighbor_strategy
    n, dims = 20_000, (16, 16, 8)
    g = torch.Generator().manual_seed(4)
    raw = MeshBatch(pos=torch.rand(n, 3, generator=g) * 5 - 2, x=torch.zeros(n, 1))
    cfg = MAGNOConfig(gno_coord_dim=3, neighbor_strategy=strategy, k_neighbors=2, gno_radius=0.11, scales=[1.0, 1.5])
    lat = latent_grid(dims)
    out = io.enrich_sample(raw, lat, cfg, latent_dims=dims, device=DEV)
    p = rescale(raw.pos)
    zp, zl = torch.zeros(n, dtype=torch.long), torch.zeros(lat.shape[0], dtype=torch.long)
    es, ds_ = parse_neighbor_strategy(strategy)
    for si, scale in enumerate(cfg.scales):
        for name, strat, is_dec, nq in (("encoder", es, False, lat.shape[0]), ("decoder", ds_, True, n)):
            want = get_neighbor_strategy(strat, p, zp, lat, zl, cfg.gno_radius * scale, 2, is_dec)      # CPU restatement
            got = getattr(out, f"{name}_edge_index_s{si}")
            assert got.dtype == torch.int32 and torch.equal(got.long(), want), (name, si)
            cnt = getattr(out, f"{name}_query_counts_s{si}")
            assert torch.equal(cnt, torch.bincount(want[1], minlength=nq).to(torch.int32))


def test_enrich_sample_morton_reorder_is_a_pure_permutation():
    """io.enrich_sample(reorder="morton") stores the points along the Z-order curve (gather locality for the GNO kernels): every
    per-point attribute moves with its point, the edge set is the same up to the renumbering, and the model's prediction put
    back in file order equals the prediction on the file-order sample (fp32 mode: only summation orders inside the segments
    change)"""
    import gaot_3d_amd
    import test_model_gpu as TM
    from gaot_3d_amd import io
    from gaot_3d_amd.data import MeshBatch, latent_grid, superellipsoid_surface
    from gaot_3d_amd.model import init_model
    n, dims = 20_000, (8, 8, 4)
    g = torch.Generator().manual_seed(9)
    pos, nrm = superellipsoid_surface(n, generator=g)
    raw = MeshBatch(pos=pos, x=torch.randn(n, 1, generator=g), c=nrm)
    cfg = TM.small_config(False, 2, latent=dims, k=4)
    lat = latent_grid(dims)
    a = io.enrich_sample(raw, lat, cfg.magno, latent_dims=dims, device=DEV)
    b = io.enrich_sample(raw, lat, cfg.magno, latent_dims=dims, device=DEV, reorder="morton")
    perm = b.point_perm
    assert sorted(perm.tolist()) == list(range(n))
    assert torch.equal(b.x, raw.x[perm]) and torch.equal(b.c, raw.c[perm]) and torch.equal(b.pos, a.pos[perm])
    inv = torch.argsort(perm)
    ea, eb = a.encoder_edge_index_s0.long(), b.encoder_edge_index_s0.long()
    assert _pairs(torch.stack([perm[eb[0]], eb[1]])) == _pairs(ea)              # same graph, points renumbered
    d = ((b.pos[1:] - b.pos[:-1]).norm(dim=1).mean() / (a.pos[1:] - a.pos[:-1]).norm(dim=1).mean()).item()
    assert d < 0.2, d                                                            # consecutive points are neighbours in space now
    gaot_3d_amd.set_precision("fp32")
    torch.manual_seed(0)
    model = init_model(6, 1, "gaot_3d", cfg).to(DEV).eval()
    preds = []
    for smp in (a, b):
        batch = MeshBatch.from_data_list([smp], num_latent_nodes=lat.shape[0]).to(DEV)
        with torch.no_grad():
            preds.append(model(batch=batch, tokens_pos=lat.to(DEV)).cpu())
    err = float((preds[1][inv] - preds[0]).abs().max()) / float(preds[0].abs().max())
    print(f"[parity] morton reorder: max|pred(reordered)[inv] - pred| = {err:.2e} of peak")
    assert err <= 1e-5


def test_loader_stages_through_pinned_memory_and_keeps_batches_on_the_device(tmp_path):
    import gaot_3d_amd
    from gaot_3d_amd import dataset as D
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.io import save_sample
    from gaot_3d_amd.model import init_model
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    root = tmp_path / "d"
    (root / "processed").mkdir(parents=True)
    dims = (8, 8, 4)
    names = []
    for i in range(3):
        b, tokens = make_synthetic_sample(3000 + 100 * i, dims, k=4, seed=i)
        b.num_latent_nodes = tokens.shape[0]
        save_sample(b, str(root / "processed" / f"m{i}.pt"))
        names.append(f"m{i}")
    (tmp_path / "order.txt").write_text("\n".join(names) + "\n")
    cfg = types.SimpleNamespace(name="toy", base_path=str(root), processed_folder="processed", train_size=3, val_size=0,
                                test_size=1, rand_dataset=False, active_variables=None)
    ds = D.VTKMeshDataset(str(root), str(tmp_path / "order.txt"), cfg, "train")
    loader = D.SampleLoader(ds, batch_size=1, device=DEV, num_latent_nodes=tokens.shape[0], device_cache=8)
    gaot_3d_amd.set_precision("fp32")
    mcfg = types.SimpleNamespace(
        magno=MAGNOConfig(gno_coord_dim=3, lifting_channels=32, encoder_feature_attr=["pos", "c"], mlp_type="linear",
                          use_geoembed=[True, False], neighbor_strategy="knn", k_neighbors=4, precompute_edges=True),
        transformer=TransformerConfig(patch_size=2, hidden_size=256, num_layers=2, positional_embedding="rope",
                                      attn_config=AttentionConfig(atten_dropout=0.0), ffn_config=FFNConfig()),
        latent_tokens=dims)
    torch.manual_seed(0)
    model = init_model(6, 1, "gaot_3d", mcfg).to(DEV).eval()
    tok = tokens.to(DEV)
    first, outs = [], []
    with torch.no_grad():
        for b in loader:
            assert b.pos.is_cuda and b.encoder_edge_index_s0.is_cuda and b.num_graphs == 1
            outs.append(model(batch=b, tokens_pos=tok).clone())
            assert "_gaot_graphs" in b.__dict__
            first.append(b)
        graphs = [id(b.__dict__["_gaot_graphs"][("enc", 0), b.pos.shape[0], tok.shape[0]][2]) for b in first]
        for i, b in enumerate(loader):      # second epoch: the same device-resident objects, neighbour lists and all
            assert b is first[i]
            assert id(b.__dict__["_gaot_graphs"][("enc", 0), b.pos.shape[0], tok.shape[0]][2]) == graphs[i]
            assert torch.equal(model(batch=b, tokens_pos=tok), outs[i])
    # B = 2 through the loader: EnrichedData.__inc__ offsets applied, the model runs the batch and equals the ORACLE on the
    # same batch (not the per-sample results: the GeoEmbed z-score runs over the query rows of the whole batch,
    # geoembed.py:177-180)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import gaot_oracle as orc  # checker only
    loader2 = D.SampleLoader(ds, batch_size=2, device=DEV, num_latent_nodes=tok.shape[0], drop_last=True)
    with torch.no_grad():
        (b2,) = list(loader2)
        o2 = model(batch=b2, tokens_pos=tok)
    assert b2.num_graphs == 2 and o2.shape[0] == outs[0].shape[0] + outs[1].shape[0]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = orc.gaot3d_forward(sd, mcfg, b2.to("cpu"), tokens)
    err = (o2.cpu() - ref).abs().max().item()
    print(f"[parity] loader B=2 batch vs oracle: max_abs={err:.3e} ref_peak={ref.abs().max().item():.3e}")
    assert torch.allclose(o2.cpu(), ref, rtol=1e-4, atol=1e-5)
