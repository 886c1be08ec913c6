"""CPU-only checks: the C-ABI library loads and exports every symbol include/*.h declares (no compute),
the drop-in model has the reference's state_dict layout, the host-side data helpers follow the reference's
conventions, and the product refuses to run without the GPU path."""
import glob
import sys
import os
import re
import types

import pytest
import torch

import golden_io as gio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gaot_3d_amd import _lib
    lib = _lib.load()
    declared = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        declared |= set(re.findall(r"\b(gaot_\w+)\s*\(", src))
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by libgaot3d_hip.so"
    assert declared == set(_lib.SIGNATURES.keys()), declared ^ set(_lib.SIGNATURES.keys())
    assert lib.gaot_abi_version() == 11



def test_attn_bwd_asm_include_is_current_and_owns_its_agprs(tmp_path):
    """k_attn_bwd_asm keeps dK^T / dV^T accumulators and the K / V fragments in a0-a223 ACROSS its asm statements: the
    generated include must be what the generator writes, and the compiler-generated part of the kernel must not touch those
    registers (checked on the device assembly: everything outside the ;;#ASMSTART / ;;#ASMEND blocks)."""
    import re, subprocess
    csrc = os.path.join(ROOT, "gaot_3d_amd", "csrc")
    gen = subprocess.run([sys.executable, os.path.join(csrc, "gen_attn_bwd_asm.py")], capture_output=True, text=True, check=True,
                         env={k: v for k, v in os.environ.items() if k != "GEN_NT"}).stdout
    assert gen == open(os.path.join(csrc, "attn_bwd_asm.inc")).read(), "attn_bwd_asm.inc is stale: python3 gen_attn_bwd_asm.py > attn_bwd_asm.inc"
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    asm = tmp_path / "attn.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-ffp-contract=fast", "-S",
                    "--cuda-device-only", os.path.join(csrc, "attn_bf16.hip"), "-o", str(asm)], check=True, capture_output=True)
    txt = asm.read_text()
    kernels = re.findall(r"^(_ZN\S*k_attn_bwd_asm\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M)
    assert len(kernels) >= 2, "k_attn_bwd_asm<true> / <false> not found in the device assembly"
    for name, body in kernels:
        inside, bad = False, []
        for ln in body.split("\n"):
            if "#ASMSTART" in ln:
                inside = True
            elif "#ASMEND" in ln:
                inside = False
            elif not inside and not ln.strip().startswith(";"):
                for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", ln):
                    if int(m.group(1)) < 224:
                        bad.append(ln.strip())
        assert not bad, f"{name}: compiler-generated code touches the asm's AGPRs: {bad[:5]}"
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body) or "LAB" in name, f"{name} needs scratch"
    # the forward's generated tile loop (k_attn_fwd_asm): same contract for its include and its a0-a95 (Q fragments, O^T accumulators)
    gen = subprocess.run([sys.executable, os.path.join(csrc, "gen_attn_fwd_asm.py")], capture_output=True, text=True, check=True,
                         env={k: v for k, v in os.environ.items() if not k.startswith("GEN_")}).stdout
    assert gen == open(os.path.join(csrc, "attn_fwd_asm.inc")).read(), "attn_fwd_asm.inc is stale: python3 gen_attn_fwd_asm.py > attn_fwd_asm.inc"
    kernels = re.findall(r"^(_ZN\S*k_attn_fwd_asm\w*):[^\n]*\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M)
    assert len(kernels) >= 2, "k_attn_fwd_asm<true> / <false> not found in the device assembly"
    for name, body in kernels:
        inside, bad = False, []
        for ln in body.split("\n"):
            if "#ASMSTART" in ln:
                inside = True
            elif "#ASMEND" in ln:
                inside = False
            elif not inside and not ln.strip().startswith(";"):
                for m in re.finditer(r"\ba\[?(\d+)(?::(\d+))?\]?", ln):
                    if int(m.group(1)) < 160:
                        bad.append(ln.strip())
        assert not bad, f"{name}: compiler-generated code touches the asm's AGPRs: {bad[:5]}"
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body), f"{name} needs scratch"


def test_hot_path_kernels_need_no_scratch(tmp_path):
    """No kernel of the shipped path may need scratch memory (spilled registers / private arrays): the code objects' metadata is
    read from the built library.  Known exceptions are variants off the default path (lab / fallback instantiations)."""
    import re, shutil, subprocess
    from gaot_3d_amd import _lib
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not installed")
    _lib.load()
    so = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "lib", "libgaot3d_hip.so")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(so, work / "lib.so")
    subprocess.run([objdump, "--offloading", str(work / "lib.so")], check=True, capture_output=True, cwd=work)
    objs = sorted(f for f in os.listdir(work) if "gfx950" in f)
    assert objs, "no gfx950 code object found in the library"
    allowed = ("k_attn_bwd_fusedILb1ELi4ELi4E",          # one-wave-per-SIMD lab variant (GAOT_ATTN_BWD_VARIANT=1)
               "k_attn_bwd_dkv_bf16ILi4E", "k_attn_bwd_dq_bf16ILi4E",   # two-pass fallback, four workgroups per CU
               "k_gno_bwdILi3ELi64E",                      # fp32-mode GNO backward, three hidden layers
               "k_gno_bwd3_bf16ILi4E")                      # four hidden layers: fragments from L2, 85 spilled registers
    seen, offenders = 0, []
    for f in objs:
        notes = subprocess.run([readelf, "--notes", str(work / f)], check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.private_segment_fixed_size:\s+(\d+)", notes, re.S):
            name, between, scratch = m.group(1), m.group(2), int(m.group(3))
            if ".name:" in between:      # the size belongs to another entry
                continue
            seen += 1
            if scratch and not any(a in name for a in allowed):
                offenders.append((name, scratch))
    assert seen > 100, seen
    assert not offenders, offenders


@pytest.mark.parametrize("case", ["model_knn_abs", "model_radius_rope", "model_channel_multiscale"])
def test_state_dict_layout_matches_reference(case):
    from test_model_gpu import product_config
    from gaot_3d_amd.model import init_model
    meta, g = gio.load(case)
    model = init_model(meta["in_size"], meta["out_size"], "gaot_3d", product_config(meta))
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["sd"].keys())           # same names in the same registration order
    for k, v in g["sd"].items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    model.load_state_dict(g["sd"], strict=True)
    assert sum(p.numel() for p in model.parameters()) == meta["nparams"]
    trainable = {k for k, p in model.named_parameters() if p.requires_grad}
    assert not any(k.endswith("rotary_emb.freqs") for k in trainable)


def test_yaml_config_parameter_count():
    """model section of the reference's pressure.yaml: 11 244 737 parameters"""
    from gaot_3d_amd.model import init_model
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    cfg = types.SimpleNamespace(
        magno=MAGNOConfig(gno_coord_dim=3, lifting_channels=32, mlp_type="linear", use_geoembed=[True, False],
                          encoder_feature_attr=["pos", "c"], neighbor_strategy="bidirectional"),
        transformer=TransformerConfig(patch_size=2, positional_embedding="rope", num_layers=10,
                                      attn_config=AttentionConfig(), ffn_config=FFNConfig()),
        latent_tokens=(64, 64, 32))
    with torch.device("meta"):
        m = init_model(6, 1, "gaot_3d", cfg)
    assert sum(p.numel() for p in m.parameters()) == 11244737
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 11244577


def test_init_model_rejects_unknown():
    from gaot_3d_amd.model import init_model
    with pytest.raises(ValueError):
        init_model(3, 1, "gino", types.SimpleNamespace())


def test_variants_have_no_cpu_fallback_and_bad_options_raise():
    """every IntegralTransform / GeoEmbed variant runs on the HIP path only: CPU tensors fail loudly (no fallback);
    invalid option strings raise ValueError like the reference"""
    from gaot_3d_amd._lib import GaotError
    from gaot_3d_amd.graph import apply_neighbor_sampling
    from gaot_3d_amd.model.layers.geoembed import GeometricEmbedding
    from gaot_3d_amd.model.layers.integral_transform import IntegralTransform
    ei = torch.tensor([[0, 1, 2], [0, 1, 1]])
    it = IntegralTransform(channel_mlp_layers=[38, 64, 32], transform_type="nonlinear")
    with pytest.raises((GaotError, RuntimeError)):
        it(torch.zeros(4, 3), torch.zeros(2, 3), ei, torch.zeros(4, 32))
    with pytest.raises((GaotError, RuntimeError)):
        GeometricEmbedding(3, 32, method="pointnet")(torch.zeros(4, 3), torch.zeros(2, 3), ei)
    # empty edge list: zeros without touching the device (integral_transform.py:106-112)
    out = it(torch.zeros(4, 3), torch.zeros(2, 3), torch.zeros(2, 0, dtype=torch.long), torch.zeros(4, 32))
    assert out.shape == (2, 32) and not out.any()
    with pytest.raises(ValueError):
        GeometricEmbedding(3, 32, method="nope")
    with pytest.raises(ValueError):
        IntegralTransform(channel_mlp_layers=[6, 64, 32], transform_type="nope")(torch.zeros(4, 3), torch.zeros(2, 3), ei,
                                                                               torch.zeros(4, 32))
    with pytest.raises(ValueError):
        apply_neighbor_sampling(ei, 2, None, "nope")
    with pytest.raises(ValueError):
        apply_neighbor_sampling(ei, 2, None, "max_neighbors")
    assert apply_neighbor_sampling(ei, 2, None, None) is ei
    assert apply_neighbor_sampling(ei, 2, None, "ratio", sample_ratio=0.5, training=False) is ei


def test_no_cpu_fallback():
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd._lib import GaotError
    with pytest.raises((GaotError, RuntimeError)):
        GF.linear(torch.zeros(4, 8), torch.zeros(3, 8), None, precision=0)


def test_batching_increments_and_graph_helpers():
    from gaot_3d_amd.data import (MeshBatch, coalesce_edges, knn_edges_bruteforce, knn_edges_grid, latent_grid,
                                  radius_edges_bruteforce)
    g = torch.Generator().manual_seed(0)
    lat = latent_grid((6, 5, 4))
    assert torch.allclose(lat[1], torch.tensor([-1.0, -1.0, -1.0 + 2 / 3]))   # w fastest, "ij" indexing
    pos = torch.rand(500, 3, generator=g) * 2 - 1
    bf = knn_edges_bruteforce(pos, lat, 8)
    gr = knn_edges_grid(pos, (6, 5, 4), (-1.0,) * 3, (1.0,) * 3, 8)
    assert torch.equal(bf[0], gr[0])
    assert torch.equal(bf[1].view(-1, 8).sort(1).values, gr[1].view(-1, 8).sort(1).values)
    # radius: centres = latent, cap 32, rows [phys, latent] sorted by latent
    rad = radius_edges_bruteforce(pos, lat, 0.6, 32, "latent")
    assert (rad[1][1:] >= rad[1][:-1]).all() and torch.bincount(rad[1]).max() <= 32
    assert ((pos[rad[0]] - lat[rad[1]]).norm(dim=1) <= 0.6 + 1e-6).all()
    co = coalesce_edges(torch.cat([bf, bf], 1), lat.shape[0])
    assert co.shape[1] == bf.shape[1]
    # batching offsets follow EnrichedData.__inc__ (reference pyg_datasets.py:18-26)
    s1 = MeshBatch(pos=pos[:10], x=pos[:10, :1], encoder_edge_index_s0=torch.tensor([[0, 9], [1, 2]]),
                   decoder_edge_index_s0=torch.tensor([[1, 2], [0, 9]]))
    s2 = MeshBatch(pos=pos[:7], x=pos[:7, :1], encoder_edge_index_s0=torch.tensor([[3], [5]]),
                   decoder_edge_index_s0=torch.tensor([[5], [3]]))
    b = MeshBatch.from_data_list([s1, s2], num_latent_nodes=120)
    assert b.num_graphs == 2 and b.pos.shape[0] == 17
    assert torch.equal(b.encoder_edge_index_s0, torch.tensor([[0, 9, 13], [1, 2, 125]]))
    assert torch.equal(b.decoder_edge_index_s0, torch.tensor([[1, 2, 125], [0, 9, 13]]))
    assert torch.equal(b.batch, torch.tensor([0] * 10 + [1] * 7))


def test_get_neighbor_strategy_conventions():
    from gaot_3d_amd.data import latent_grid
    from gaot_3d_amd.model.layers.magno import get_neighbor_strategy
    g = torch.Generator().manual_seed(1)
    lat = latent_grid((3, 3, 3))
    pos = torch.rand(40, 3, generator=g) * 2 - 1
    bp, bl = torch.zeros(40, dtype=torch.long), torch.zeros(27, dtype=torch.long)
    enc = get_neighbor_strategy("knn", pos, bp, lat, bl, 0.5, 2, False)
    dec = get_neighbor_strategy("knn", pos, bp, lat, bl, 0.5, 2, True)
    assert enc.shape == (2, 80) and enc[0].max() < 40 and enc[1].max() < 27     # [phys, latent]
    assert torch.equal(dec, enc.flip(0))                                          # [latent, phys]
    bi = get_neighbor_strategy("bidirectional", pos, bp, lat, bl, 0.9, 2, False)
    rev = get_neighbor_strategy("reverse", pos, bp, lat, bl, 0.9, 2, True)
    assert torch.equal(rev, bi.flip(0))        # 'reverse' = flip of the *bidirectional* encoder graph
    with pytest.raises(ValueError):
        get_neighbor_strategy("nope", pos, bp, lat, bl, 0.5, 1, False)


def test_colocate_keeps_parameters_and_values():
    """co-locating q/k/v (or w1/w3) weights in one buffer must not change parameter identity, shapes, values or the
    state_dict -- only where the storage lives (reference keeps them as separate nn.Linear weights, attn.py:76-79)"""
    import torch
    from gaot_3d_amd import functional as GF
    lin = [torch.nn.Linear(16, n, bias=False) for n in (16, 8, 8)]
    ps = [m.weight for m in lin]
    before = [p.detach().clone() for p in ps]
    ids = [id(p) for p in ps]
    assert not GF._adjacent([p.data for p in ps])
    GF.colocate(ps)
    assert [id(p) for p in ps] == ids
    assert GF._adjacent([p.data for p in ps])
    for p, b in zip(ps, before):
        assert p.shape == b.shape and torch.equal(p.detach(), b) and p.requires_grad and p.is_leaf
    # in-place updates (optimizer, load_state_dict) keep the co-location; .to()-style replacement undoes it harmlessly
    with torch.no_grad():
        ps[1].add_(1.0)
    lin[0].load_state_dict({"weight": torch.zeros(16, 16)})
    assert GF._adjacent([p.data for p in ps]) and float(ps[0].abs().sum()) == 0.0
    assert torch.equal(ps[1].detach(), before[1] + 1.0)
    GF.colocate(ps)   # idempotent
    assert GF._adjacent([p.data for p in ps])
    # neighbours in memory that are NOT slices of one buffer must not be taken for one matrix
    a, b = torch.zeros(4, 4), torch.zeros(4, 4)
    assert not GF._adjacent([a, b])


def test_mix_lr_schedule_matches_reference_golden():
    """gaot_3d_amd.schedule.MixLRScheduler against the per-epoch learning rates of the reference's CustomLRScheduler +
    AdamWOptimizer phase split (golden lr_mix.npz, oracle/make_goldens.py: lr_schedule_case)"""
    from gaot_3d_amd.schedule import MixLRScheduler
    meta, g = gio.load("lr_mix")
    for total in meta["totals"]:
        want = g["out"][f"lr_{total}"].double()
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=meta["lr"])
        sch = MixLRScheduler(opt, total, meta["lr"], meta["max_lr"], meta["min_lr"], meta["final_lr"])
        got = []
        for _ in range(total + 2):
            got.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        assert torch.allclose(torch.tensor(got, dtype=torch.float64), want, rtol=1e-12, atol=0), total


def test_neural_field_sampling_shapes_and_semantics():
    from gaot_3d_amd.data import MeshBatch
    from gaot_3d_amd.schedule import sample_nodes_neural_field
    g = torch.Generator().manual_seed(0)
    s1 = MeshBatch(pos=torch.rand(50, 3, generator=g), x=torch.arange(50.0)[:, None], c=torch.rand(50, 2, generator=g))
    s2 = MeshBatch(pos=torch.rand(20, 3, generator=g), x=100 + torch.arange(20.0)[:, None], c=torch.rand(20, 2, generator=g))
    b = MeshBatch.from_data_list([s1, s2], num_latent_nodes=8)
    sb, qp, qb, tgt = sample_nodes_neural_field(b, 30, 30, generator=torch.Generator().manual_seed(1))
    assert sb.pos.shape == (50, 3) and sb.num_graphs == 2                 # 30 of 50, all 20 of 20
    assert torch.equal(sb.batch, torch.tensor([0] * 30 + [1] * 20)) and torch.equal(sb.ptr, torch.tensor([0, 30, 50]))
    assert torch.equal(qp, sb.pos) and torch.equal(tgt, sb.x)             # same counts -> same subset
    assert len(set(sb.x[:30, 0].tolist())) == 30 and sb.x[:30].max() < 50 and sb.x[30:].min() >= 100
    sb, qp, qb, tgt = sample_nodes_neural_field(b, 10, 25, generator=torch.Generator().manual_seed(2))
    assert sb.pos.shape[0] == 20 and qp.shape[0] == 45 and torch.equal(qb, torch.tensor([0] * 25 + [1] * 20))
    assert tgt.shape == (45, 1) and tgt[:25].max() < 50 and tgt[25:].min() >= 100


def test_sample_reader_without_pyg(tmp_path):
    """gaot_3d_amd.io.load_sample on a pickle shaped like torch_geometric's Data (attribute bags: Data -> _store ->
    _mapping), written through stand-in modules that are removed again before reading -- the reader must not need
    torch_geometric -- plus the plain-dict round trip and the CPU path of the edge pre-computation pass"""
    import sys
    import types as _t
    from gaot_3d_amd import io as gio2
    from gaot_3d_amd.data import latent_grid
    mods = {}
    for name in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data", "torch_geometric.data.storage"):
        mods[name] = _t.ModuleType(name)
    GlobalStorage = type("GlobalStorage", (), {"__module__": "torch_geometric.data.storage"})
    Data = type("Data", (), {"__module__": "torch_geometric.data.data"})
    mods["torch_geometric.data.storage"].GlobalStorage = GlobalStorage
    mods["torch_geometric.data.data"].Data = Data
    sys.modules.update(mods)
    try:
        g = torch.Generator().manual_seed(0)
        st = GlobalStorage()
        d = Data()
        st._mapping = {"pos": torch.rand(40, 3, generator=g), "x": torch.rand(40, 3, 1, generator=g), "c": torch.rand(40, 2, generator=g),
                       "filename": "run_7", "num_latent_nodes": 27,
                       "encoder_edge_index_s0": torch.randint(0, 27, (2, 90), generator=g).to(torch.int32)}
        st._parent = d
        d._store = st
        d._edge_attr_cls, d._tensor_attr_cls = None, None
        path = tmp_path / "run_7.pt"
        torch.save(d, path)
    finally:
        for name in mods:
            sys.modules.pop(name, None)
    assert "torch_geometric" not in sys.modules
    s = gio2.load_sample(str(path), active_variables=[0, 2])
    assert s.pos.shape == (40, 3) and s.x.shape == (40, 2) and s.c.shape == (40, 2)
    assert s.filename == "run_7" and s.num_latent_nodes == 27 and s.encoder_edge_index_s0.dtype == torch.int32
    assert torch.equal(s.x, st._mapping["x"][:, [0, 2], 0]) and torch.equal(s.ptr, torch.tensor([0, 40]))
    p2 = tmp_path / "plain.pt"
    gio2.save_sample(s, str(p2))
    s2 = gio2.load_sample(str(p2))
    assert torch.equal(s2.pos, s.pos) and torch.equal(s2.encoder_edge_index_s0, s.encoder_edge_index_s0)
    cfg = types.SimpleNamespace(neighbor_strategy=["knn", "knn"], scales=[1.0], gno_radius=0.3, k_neighbors=2)
    e = gio2.enrich_sample(s, latent_grid((3, 3, 3)), cfg)
    assert e.num_latent_nodes == 27 and e.encoder_edge_index_s0.dtype == torch.int32 and e.encoder_edge_index_s0.shape == (2, 80)
    assert e.decoder_edge_index_s0.shape == (2, 80) and int(e.decoder_query_counts_s0.sum()) == 80
    assert torch.equal(e.encoder_query_counts_s0.long(), torch.bincount(e.encoder_edge_index_s0[1].long(), minlength=27))
    assert float(e.pos.min()) == float(s.pos.min())          # coordinates stay as stored; edges use the rescaled copy


def test_dataset_transforms_stats_and_loader(tmp_path):
    """reference data layer on plain files (no PyG): order-file split (incl. rand_dataset's default_rng(42) shuffle), the
    three transforms, the normalisation-statistics file (keys, unbiased std, reload), and the loader's collation with the
    EnrichedData.__inc__ offsets -- all against direct torch / numpy restatements of the reference lines"""
    import types
    import numpy as np
    from gaot_3d_amd import dataset as D
    from gaot_3d_amd.data import MeshBatch, knn_edges_bruteforce, latent_grid
    from gaot_3d_amd.io import save_sample
    root = tmp_path / "data"
    (root / "processed").mkdir(parents=True)
    g = torch.Generator().manual_seed(0)
    lat = latent_grid((3, 3, 2))
    names, samples = [], []
    for i in range(7):
        n = 40 + 5 * i
        pos = torch.rand(n, 3, generator=g) * 3 - 1
        enc = knn_edges_bruteforce(pos, lat, 2)
        s = MeshBatch(pos=pos, x=torch.randn(n, 2, 1, generator=g) * (i + 1), c=torch.randn(n, 3, generator=g) + i,
                      encoder_edge_index_s0=enc.to(torch.int32), decoder_edge_index_s0=enc.flip(0).to(torch.int32),
                      num_latent_nodes=lat.shape[0], filename=f"s{i}")
        save_sample(s, str(root / "processed" / f"s{i}.pt"))
        names.append(f"s{i}")
        samples.append(s)
    order = tmp_path / "order.txt"
    order.write_text("\n".join(names) + "\n")
    cfg = types.SimpleNamespace(name="toy", base_path=str(root), processed_folder="processed", train_size=4, val_size=2,
                                test_size=1, rand_dataset=False, active_variables=[1])
    tr = D.VTKMeshDataset(str(root), str(order), cfg, "train")
    va = D.VTKMeshDataset(str(root), str(order), cfg, "val")
    te = D.VTKMeshDataset(str(root), str(order), cfg, "test")
    assert tr.split_filenames == ["s0.pt", "s1.pt", "s2.pt", "s3.pt"] and va.split_filenames == ["s4.pt", "s5.pt"]
    assert te.split_filenames == ["s6.pt"] and len(tr) == 4
    s0 = tr[0]
    assert s0.x.shape == (40, 1) and torch.equal(s0.x[:, 0], samples[0].x[:, 1, 0])      # active_variables + squeeze(-1)
    cfg_r = types.SimpleNamespace(**{**cfg.__dict__, "rand_dataset": True})
    idx = np.arange(7)
    np.random.default_rng(seed=42).shuffle(idx)
    assert D.VTKMeshDataset(str(root), str(order), cfg_r, "train").split_filenames == [f"s{i}.pt" for i in idx[:4]]
    with pytest.raises(ValueError):
        D.VTKMeshDataset(str(root), str(order), cfg, "bogus")
    # transforms
    p = samples[2].pos.clone()
    want = (p - p.min()) / (p.max() - p.min()) * 2 - 1
    assert torch.allclose(D.RescalePosition()(MeshBatch(pos=p.clone())).pos, want)
    dom = ([-1.16, -1.2, 0.0], [4.21, 1.19, 1.77])
    want = (p - (-1.2)) / (4.21 - (-1.2)) * 2 - 1
    assert torch.allclose(D.RescalePositionNew(phy_domain=dom)(MeshBatch(pos=p.clone())).pos, want)
    # statistics: all training points, unbiased std, positions irrelevant
    stats = D.calculate_or_load_stats(cfg, str(order), str(root))
    allx = torch.cat([samples[i].x[:, [1], 0] for i in range(4)])
    allc = torch.cat([samples[i].c for i in range(4)])
    assert torch.allclose(stats["mean"], allx.mean(0), atol=1e-6) and torch.allclose(stats["std"], allx.std(0), rtol=1e-5)
    assert torch.allclose(stats["c_mean"], allc.mean(0), atol=1e-6) and torch.allclose(stats["c_std"], allc.std(0), rtol=1e-5)
    assert os.path.exists(root / "toy_norm_stats.pt")
    again = D.calculate_or_load_stats(cfg, str(order), str(root))            # loaded from the file
    assert all(torch.equal(stats[k], again[k]) for k in stats)
    norm = D.NormalizeFeatures(stats["mean"], stats["std"], stats["c_mean"], stats["c_std"])
    b = norm(MeshBatch(x=allx.clone(), c=allc.clone()))
    assert torch.allclose(b.x, (allx - stats["mean"]) / (stats["std"] + 1e-10)) and abs(float(b.c.mean())) < 1e-5
    # loader: B = 2 batches carry the __inc__ offsets (encoder rows +[nodes, latent], decoder rows +[latent, nodes])
    tf = D.Compose([D.RescalePosition(), norm])
    ds = D.VTKMeshDataset(str(root), str(order), cfg, "train", transform=tf)
    batches = list(D.SampleLoader(ds, batch_size=2, device="cpu", num_latent_nodes=lat.shape[0]))
    assert len(batches) == 2 and batches[0].num_graphs == 2
    b0 = batches[0]
    n0 = samples[0].pos.shape[0]
    e0, e1 = samples[0].encoder_edge_index_s0, samples[1].encoder_edge_index_s0
    want = torch.cat([e0, e1 + torch.tensor([[n0], [lat.shape[0]]], dtype=torch.int32)], dim=1)
    assert torch.equal(b0.encoder_edge_index_s0, want) and torch.equal(b0.decoder_edge_index_s0, want.flip(0))
    assert torch.equal(b0.batch, torch.cat([torch.zeros(n0, dtype=torch.long), torch.ones(45, dtype=torch.long)]))
    assert b0.ptr.tolist() == [0, 40, 85]
    shuffled = D.SampleLoader(ds, batch_size=1, device="cpu", shuffle=True, seed=3)
    a = [bb.pos.shape[0] for bb in shuffled]
    shuffled.set_epoch(1)
    c = [bb.pos.shape[0] for bb in shuffled]
    assert sorted(a) == sorted(c) == [40, 45, 50, 55] and a != c


@pytest.mark.parametrize("strategy,is_decoder", [("knn", False), ("radius", False), ("bidirectional", False), ("knn", True),
                                                 ("radius", True), ("bidirectional", True), ("reverse", True)])
def test_host_graph_helper_matches_graph_oracle(strategy, is_decoder):
    """the product's CPU graph helper (data preparation on the host: gaot_3d_amd/data.py + magno.get_neighbor_strategy for CPU
    tensors) against the oracle's independent restatement of the reference conventions (oracle/graph_oracle.py): two graphs
    per batch, 32-per-centre cap reached"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import graph_oracle as gorc
    from gaot_3d_amd.data import latent_grid
    from gaot_3d_amd.model.layers.magno import get_neighbor_strategy
    g = torch.Generator().manual_seed(5)
    lat1 = latent_grid((5, 4, 3))
    lat = lat1.repeat(2, 1)
    pos = torch.rand(900, 3, generator=g) * 2 - 1
    bp = torch.cat([torch.zeros(500, dtype=torch.long), torch.ones(400, dtype=torch.long)])
    bl = torch.arange(2).repeat_interleave(lat1.shape[0])
    got = get_neighbor_strategy(strategy, pos, bp, lat, bl, 0.5, 3, is_decoder)
    ref = gorc.get_neighbor_strategy(strategy, pos, bp, lat, bl, 0.5, 3, is_decoder)
    assert got.shape == ref.shape
    if strategy == "knn":
        assert set(map(tuple, got.t().tolist())) == set(map(tuple, ref.t().tolist()))
    else:
        assert torch.equal(got.long(), ref)
