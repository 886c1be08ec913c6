"""CPU, world_size 2 over gloo: the point-shard partition and its two exchange steps (forward all-reduce of the
encoder's per-token sums/counts, backward all-reduce of the decoder's latent gradient, flat all-reduce of the
per-point parameter gradients) reproduce the unsharded result.  The per-rank compute in this test is the CPU
oracle (test infrastructure); on the GPU the same host logic wraps the HIP kernels (tests/test_model_gpu.py)."""
import os

import pytest
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def _problem():
    from gaot_3d_amd.data import MeshBatch, knn_edges_bruteforce, latent_grid
    g = torch.Generator().manual_seed(0)
    lat = latent_grid((4, 4, 3))
    pos = torch.rand(301, 3, generator=g) * 2 - 1
    enc = knn_edges_bruteforce(pos, lat, 3)
    batch = MeshBatch(pos=pos, x=torch.randn(301, 2, generator=g), c=torch.randn(301, 3, generator=g),
                      batch=torch.zeros(301, dtype=torch.long), encoder_edge_index_s0=enc,
                      decoder_edge_index_s0=enc.flip(0))
    sd = {}
    for pre, layers in (("enc.", [6, 64, 32]), ("dec.", [6, 64, 32])):
        for i in range(2):
            sd[f"{pre}channel_mlp.fcs.{i}.weight"] = torch.randn(layers[i + 1], layers[i], generator=g) * 0.3
            sd[f"{pre}channel_mlp.fcs.{i}.bias"] = torch.randn(layers[i + 1], generator=g) * 0.1
    sd["lift.weight"] = torch.randn(32, 3, generator=g) * 0.3
    sd["mix.weight"] = torch.randn(32, 32, generator=g) * 0.2      # stands for the replicated latent processor
    sd["proj.weight"] = torch.randn(2, 32, generator=g) * 0.3
    return batch, lat, sd


def _pipeline(sd, batch, lat, n_total, group=None):
    """lift -> encoder GNO -> (replicated) mixer -> decoder GNO -> projection -> MSE; sharded when group is given"""
    import gaot_oracle as orc
    from gaot_3d_amd.sharding import AllReduceGradFn, GlobalSegmentMeanFn
    f = batch.c @ sd["lift.weight"].t()
    enc_sd = {k[4:]: v for k, v in sd.items() if k.startswith("enc.")}
    dec_sd = {k[4:]: v for k, v in sd.items() if k.startswith("dec.")}
    z = orc.integral_transform(enc_sd, "", batch.pos, lat, batch.encoder_edge_index_s0, f)
    if group is not None:
        deg = torch.bincount(batch.encoder_edge_index_s0[1], minlength=lat.shape[0]).float()
        z = GlobalSegmentMeanFn.apply(z, deg, group)
    z = torch.tanh(z @ sd["mix.weight"].t())
    if group is not None:
        z = AllReduceGradFn.apply(z, group)
    y = orc.integral_transform(dec_sd, "", lat, batch.pos, batch.decoder_edge_index_s0, z)
    pred = y @ sd["proj.weight"].t()
    loss = ((pred - batch.x) ** 2).sum() / (n_total * pred.shape[1])
    return pred, loss


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaot_3d_amd.sharding import allreduce_partial_grads, shard_batch
        torch.set_num_threads(1)
        batch, lat, sd = _problem()
        n = batch.pos.shape[0]
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        local = shard_batch(batch, rank, world, lat.shape[0])
        pred, loss = _pipeline(leaves, local, lat, n, dist.group.WORLD)
        loss.backward()
        # per-point / per-edge parameters carry partial sums; the replicated mixer already has the full gradient
        allreduce_partial_grads([v for k, v in leaves.items() if not k.startswith("mix.")], dist.group.WORLD)
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        ret[rank] = dict(pred=pred.detach(), loss=tot, lo_hi=local.shard[2:4],
                         grads={k: v.grad.clone() for k, v in leaves.items()})
    finally:
        dist.destroy_process_group()


def test_point_shard_two_ranks_matches_unsharded():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    batch, lat, sd = _problem()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pred, loss = _pipeline(leaves, batch, lat, batch.pos.shape[0])
    loss.backward()
    got = torch.cat([ret[r]["pred"] for r in range(world)])
    assert ret[0]["lo_hi"] == (0, 150) and ret[1]["lo_hi"] == (150, 301)
    assert torch.allclose(got, pred.detach(), rtol=1e-5, atol=1e-6)
    for r in range(world):
        assert torch.allclose(ret[r]["loss"], loss.detach(), rtol=1e-5, atol=1e-7)
        for k, v in leaves.items():
            assert torch.allclose(ret[r]["grads"][k], v.grad, rtol=1e-4, atol=1e-6), (r, k)


def test_shard_batch_partition():
    from gaot_3d_amd.sharding import shard_batch, shard_range
    batch, lat, _ = _problem()
    n = batch.pos.shape[0]
    seen_enc, seen_dec = 0, 0
    for world in (1, 2, 3, 8):
        seen_enc = seen_dec = 0
        cover = []
        for r in range(world):
            s = shard_batch(batch, r, world, lat.shape[0])
            lo, hi = shard_range(n, r, world)
            cover.append((lo, hi))
            assert s.pos.shape[0] == hi - lo and torch.equal(s.pos, batch.pos[lo:hi]) and torch.equal(s.x, batch.x[lo:hi])
            e, d = s.encoder_edge_index_s0, s.decoder_edge_index_s0
            assert e.shape[1] == 3 * (hi - lo) and int(e[0].min()) >= 0 and int(e[0].max()) < hi - lo
            assert torch.equal(d, e.flip(0))
            assert not hasattr(s, "geo_pos")    # no rank keeps the full geometry: GeoEmbed statistics travel as moments
            seen_enc += e.shape[1]
            seen_dec += d.shape[1]
        assert cover[0][0] == 0 and cover[-1][1] == n and all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
        assert seen_enc == batch.encoder_edge_index_s0.shape[1] and seen_dec == batch.decoder_edge_index_s0.shape[1]


# ---- head-parallel attention exchange (gaot_3d_amd/sharding.py helpers used by functional.AttentionFn) ------------
def _attn_math(qkv, s, h, hkv):
    """plain softmax attention on a fused [rows, (h + 2 hkv) * 32] projection (stands in for the HIP kernels on CPU)"""
    rows = qkv.shape[0]
    b = rows // s
    q, k, v = qkv.split([h * 32, hkv * 32, hkv * 32], dim=1)
    q = q.view(b, s, h, 32).transpose(1, 2)
    k = k.view(b, s, hkv, 32).transpose(1, 2).repeat_interleave(h // hkv, dim=1)
    v = v.view(b, s, hkv, 32).transpose(1, 2).repeat_interleave(h // hkv, dim=1)
    att = torch.softmax(q @ k.transpose(-1, -2) / 32 ** 0.5, dim=-1)
    return (att @ v).transpose(1, 2).reshape(rows, h * 32)


class _HeadParallelAttn(torch.autograd.Function):
    """the exchange of functional.AttentionFn(head_group=...) around a stand-in attention"""

    @staticmethod
    def forward(ctx, qkv, s, h, hkv, group, rank, world):
        from gaot_3d_amd import sharding as sh
        local = sh.local_qkv(qkv, rank, world, h, hkv).detach().requires_grad_(True)
        with torch.enable_grad():
            o_local = _attn_math(local, s, h // world, hkv // world)
        ctx.local, ctx.o_local, ctx.meta = local, o_local, (h, hkv, group, rank, world)
        return sh.gather_head_outputs(o_local.detach(), group, world)

    @staticmethod
    def backward(ctx, d_o):
        from gaot_3d_amd import sharding as sh
        h, hkv, group, rank, world = ctx.meta
        hl = h // world
        (dl,) = torch.autograd.grad(ctx.o_local, ctx.local, d_o[:, rank * hl * 32:(rank + 1) * hl * 32])
        return sh.gather_qkv_grads(dl, group, world, h, hkv), None, None, None, None, None, None


def _hp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(3)
        s, h, hkv = 24, 4, (2 if world == 2 else 4)
        qkv = torch.randn(2 * s, (h + 2 * hkv) * 32, generator=g).requires_grad_(True)
        w = torch.randn(2 * s, h * 32, generator=g)
        out = _HeadParallelAttn.apply(qkv, s, h, hkv, dist.group.WORLD, rank, world)
        (out * w).sum().backward()
        ret[rank] = (out.detach(), qkv.grad.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_head_parallel_exchange_two_ranks_matches_full_attention(world):
    """world 2 and 4 over gloo: every rank computes its share of the (grouped-query) heads; gathered outputs and gathered
    q|k|v gradients equal full attention on every rank (four ranks: a wrong chunk ORDER in a gather is invisible with two)"""
    from gaot_3d_amd.sharding import head_slices
    assert head_slices(1, 2, 4, 2) == ((64, 128), (160, 192), (224, 256))
    port = 31500 + (os.getpid() % 2000) + world
    ret = mp.Manager().dict()
    mp.spawn(_hp_worker, args=(world, port, ret), nprocs=world, join=True)
    g = torch.Generator().manual_seed(3)
    s, h, hkv = 24, 4, (2 if world == 2 else 4)
    qkv = torch.randn(2 * s, (h + 2 * hkv) * 32, generator=g).requires_grad_(True)
    w = torch.randn(2 * s, h * 32, generator=g)
    ref = _attn_math(qkv, s, h, hkv)
    (ref * w).sum().backward()
    for r in range(world):
        out, grad = ret[r]
        assert torch.allclose(out, ref.detach(), rtol=1e-5, atol=1e-6)
        assert torch.allclose(grad, qkv.grad, rtol=1e-4, atol=1e-6)


# ---- sequence-parallel Transformer exchange (sharding.SliceRowsFn / SeqToHeadsFn / HeadsToSeqFn / AllGatherRowsFn) -----
def _seq_layer(x, wqkv, wo, s_total, h, hkv, group=None):
    """one stand-in block on token rows: q|k|v projection, attention (all rows of the rank's heads), output projection"""
    from gaot_3d_amd import sharding as sh
    qkv = x @ wqkv.t()
    if group is None:
        o = _attn_math(qkv, s_total, h, hkv)
    else:
        world = dist.get_world_size(group)
        loc = sh.SeqToHeadsFn.apply(qkv, group, h, hkv)
        o = sh.HeadsToSeqFn.apply(_attn_math(loc, s_total, h // world, hkv // world), group)
    return x + o @ wo.t()


def _seq_problem(world=2):
    g = torch.Generator().manual_seed(11)
    s, h, hkv, d = 24, 4, (2 if world == 2 else 4), 48
    x = torch.randn(s, d, generator=g)
    wqkv = torch.randn((h + 2 * hkv) * 32, d, generator=g) * 0.2
    wo = torch.randn(d, h * 32, generator=g) * 0.2
    wdec = torch.randn(s, d, generator=g)       # stands for the decoder: every rank consumes the full latent grid
    return s, h, hkv, x, wqkv, wo, wdec


def _seq_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaot_3d_amd import sharding as sh
        torch.set_num_threads(1)
        s, h, hkv, x, wqkv, wo, wdec = _seq_problem(world)
        x = x.requires_grad_(True)
        wqkv, wo = wqkv.clone().requires_grad_(True), wo.clone().requires_grad_(True)
        grp = dist.group.WORLD
        rows = sh.SliceRowsFn.apply(x, grp)                      # replicated latent -> my token rows
        y = _seq_layer(_seq_layer(rows, wqkv, wo, s, h, hkv, grp), wqkv, wo, s, h, hkv, grp)
        full = sh.AllGatherRowsFn.apply(y, grp)                  # my rows -> replicated latent for the decoder
        # every rank's "decoder" sees only its share of the loss (its physical points): partial gradients of `full`
        share = (full * wdec)[rank::world].sum()
        share.backward()
        for p in (wqkv, wo):                                     # row-partial weight gradients -> flat SUM all-reduce
            dist.all_reduce(p.grad)
        ret[rank] = (full.detach(), x.grad.clone(), wqkv.grad.clone(), wo.grad.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sequence_parallel_exchange_two_ranks_matches_unsharded(world):
    """world 2 and 4 over gloo: token rows split over the ranks, heads split inside attention (all-to-all both ways), rows
    all-gathered for the replicated consumer; output, input gradient and weight gradients equal the unsharded layer"""
    from gaot_3d_amd import sharding as sh
    q = torch.arange(5 * 8 * 32, dtype=torch.float32).view(5, 8 * 32)
    assert torch.equal(sh._unpack_heads(sh._pack_heads(q, 2, 4, 2), 2, 4, 2), q)
    port = 33500 + (os.getpid() % 2000) + world
    ret = mp.Manager().dict()
    mp.spawn(_seq_worker, args=(world, port, ret), nprocs=world, join=True)
    s, h, hkv, x, wqkv, wo, wdec = _seq_problem(world)
    x = x.requires_grad_(True)
    wqkv, wo = wqkv.requires_grad_(True), wo.requires_grad_(True)
    y = _seq_layer(_seq_layer(x, wqkv, wo, s, h, hkv), wqkv, wo, s, h, hkv)
    (y * wdec).sum().backward()
    for r in range(world):
        full, gx, gq, go = ret[r]
        assert torch.allclose(full, y.detach(), rtol=1e-5, atol=1e-5)
        for got, ref in ((gx, x.grad), (gq, wqkv.grad), (go, wo.grad)):   # fp32 sums in another order: bound relative to the peak
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6, float((got - ref).abs().max())


# ---- bucketed gradient all-reduce launched from gradient hooks (sharding.GradBuckets) ----------------------------------
def _bucket_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaot_3d_amd import comm
        from gaot_3d_amd.sharding import GradBuckets
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(5)
        ps = [torch.nn.Parameter(torch.randn(n, 7, generator=g)) for n in (3, 50, 11, 200, 5)]
        unused = torch.nn.Parameter(torch.randn(4, 4, generator=g))            # never receives a gradient on ANY rank
        lonely = torch.nn.Parameter(torch.randn(6, 7, generator=g))            # a gradient on rank 1 only (rank 0's shard is empty)
        # backward finishes parameters in reverse order: buckets [p4, p3] | [p2, lonely, p1] | [p0, unused].  Rank 1 completes
        # bucket 1 from its hooks, rank 0 is one gradient short there and launches it (and bucket 2) in finish(): the order of
        # the collectives on the group must be the same on both ranks all the same
        allp = [unused, ps[0], ps[1], lonely, ps[2], ps[3], ps[4]]
        gb = GradBuckets(allp, dist.new_group(backend="gloo"), bucket_bytes=1000)
        assert [len(b["params"]) for b in gb.buckets] == [2, 3, 2] and any(q is lonely for q in gb.buckets[1]["params"])
        x = torch.randn(7, generator=torch.Generator().manual_seed(100 + rank))
        launched = []
        for it in range(3):                                                     # later passes: counters were re-armed
            for p in ps + [unused, lonely]:
                p.grad = None
            n0 = comm.COUNTS["collectives"]
            loss = sum(((p * (i + 1)) @ x).sum() for i, p in enumerate(ps))
            if rank == 1:
                loss = loss + (lonely @ x).sum()
            loss.backward()
            launched_in_backward = comm.COUNTS["collectives"] - n0
            launched.append(launched_in_backward)
            gb.finish()
        grads_out = [p.grad.clone() for p in ps]
        lonely_out = None if lonely.grad is None else lonely.grad.clone()
        unused_none = unused.grad is None
        # a second backward while a bucket is in flight must be refused, not raced
        refused = False
        for p in ps + [unused, lonely]:
            p.grad = None
        loss = sum(((p * (i + 1)) @ x).sum() for i, p in enumerate(ps)) + (lonely @ x).sum() * 0.0
        loss.backward()
        try:
            sum((p @ x).sum() for p in ps).backward()
        except RuntimeError as e:
            refused = "already launched" in str(e)
        gb.reset()
        # `unused` starts to receive a gradient.  Step 0: on rank 0 only, in time for its bucket (which leaves in finish()
        # there).  Then the set is re-learnt without it (two steps with no gradient).  Step 3: on both ranks, and on rank 1 -- which has
        # `lonely` and therefore launches every bucket from its hooks -- it arrives BEHIND its bucket's launch: no error, the same sum
        # on both ranks in that very step; step 4: an ordinary bucket member
        late = []
        for it in range(5):
            for p in ps + [unused, lonely]:
                p.grad = None
            loss = 0.0
            if it == 0 and rank == 0:
                loss = loss + (unused * 3.0).sum()
            if it >= 3:
                loss = loss + (unused * (3.0 if rank == 0 else 5.0)).sum()      # created first: its gradient arrives last
            loss = loss + sum(((p * (i + 1)) @ x).sum() for i, p in enumerate(ps))
            if rank == 1:
                loss = loss + (lonely @ x).sum()
            loss.backward()
            gb.finish()
            late.append(None if unused.grad is None else unused.grad.clone())
        ret[rank] = (grads_out, unused_none, launched, lonely_out, refused, late)
    finally:
        dist.destroy_process_group()


def test_grad_buckets_allreduce_from_hooks():
    world = 2
    port = 35500 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_bucket_worker, args=(world, port, ret), nprocs=world, join=True)
    xs = [torch.randn(7, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    for r in range(world):
        grads, unused_none, in_bwd, lonely, refused, late = ret[r]
        # complete buckets were launched while backward was still running, in index order: rank 1 has every gradient of
        # buckets 0 and 1, rank 0 only of bucket 0; once `unused` is known to be unused everywhere (after the first step) it
        # no longer holds the last bucket back on rank 1
        assert unused_none and in_bwd == ([1, 1, 1] if r == 0 else [2, 3, 3]), in_bwd
        # the rank whose shard produced no gradient for `lonely` holds the other rank's gradient afterwards
        assert lonely is not None and torch.allclose(lonely, xs[1][None, :].expand(6, 7), rtol=1e-6, atol=1e-6)
        assert refused
        assert torch.equal(late[0], torch.full((4, 4), 3.0)) and late[1] is None and late[2] is None, late
        assert torch.equal(late[3], torch.full((4, 4), 8.0)) and torch.equal(late[4], torch.full((4, 4), 8.0)), late
        for i, gsum in enumerate(grads):
            ref = sum((i + 1) * x for x in xs)[None, :].expand_as(gsum)
            assert torch.allclose(gsum, ref, rtol=1e-6, atol=1e-6), i


def test_bench_spawns_its_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` outside torchrun must start the ranks as a fresh child (before any GPU call), relay the
    child's JSON line and exit 0; --dry-run keeps the ranks off the GPU (gloo rendezvous only)"""
    import json
    import subprocess
    import bench
    args = bench.parse_args(["--gpus", "4", "--steps", "3"])
    cmd = bench.spawn_command(args, ["--gpus", "4", "--steps", "3"], 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert args.scaling == "strong"                     # BASELINE configs[2]: ONE 500K-point sample over the N GPUs
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["scaling"] == "strong"
