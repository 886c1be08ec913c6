#!/usr/bin/env python3
"""bench.py -- mesh-points/sec of one GAOT-3D training step (zero_grad, forward, MSE, backward, AdamW step)
on ONE synthetic DrivAerNet++-shaped sample (BASELINE.json configs[1]: 500K points, latent 64x64x32, knn k=8
encoder + flipped decoder, model section of the reference's pressure.yaml: C=32, P=2, d=256, h=8, F=1024, L=10, rope).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

N>1 shards the PHYSICAL POINTS of the one sample across ranks (latent tokens replicated, RCCL all-reduce of the
encoder's per-token sums/counts and of the decoder's latent gradient); value = points of the whole sample / step time.
Rank 0 prints ONE JSON line.  The timed region starts with all inputs resident in HBM; the neighbour-list (CSR) build
and the geometric-embedding statistics are recomputed inside every timed step (nothing is cached across steps).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def model_config(latent, layers, k, atten_dropout=0.1):
    from gaot_3d_amd.model.layers.attn import AttentionConfig, FFNConfig, TransformerConfig
    from gaot_3d_amd.model.layers.magno import MAGNOConfig
    return types.SimpleNamespace(
        magno=MAGNOConfig(use_gno=True, gno_coord_dim=3, neighbor_strategy="knn", k_neighbors=k, projection_channels=256,
                          in_gno_channel_mlp_hidden_layers=[64, 64, 64], out_gno_channel_mlp_hidden_layers=[64, 64],
                          lifting_channels=32, gno_radius=0.033, use_geoembed=[True, False],
                          embedding_method="statistical", encoder_feature_attr=["pos", "c"], mlp_type="linear",
                          precompute_edges=True),
        transformer=TransformerConfig(patch_size=2, hidden_size=256, use_attn_norm=True, use_ffn_norm=True, norm_eps=1e-6,
                                      num_layers=layers, positional_embedding="rope", use_long_range_skip=True,
                                      attn_config=AttentionConfig(hidden_size=256, num_heads=8, num_kv_heads=8,
                                                                  atten_dropout=atten_dropout),
                                      ffn_config=FFNConfig(hidden_size=1024)),
        latent_tokens=tuple(latent))


def algorithmic_work(n_pts, m_lat, e_enc, e_dec, s_tok, layers, heads=8, dh=32, c=32, b=4):
    """Per-launch algorithmic flops / bytes of the instrumented kernels (SURVEY §8d per-unit figures)."""
    att = 2 * s_tok * s_tok * dh * heads  # one S x S x dh product over all heads
    return {
        "attn_fwd": dict(flops=2 * att, bytes=None, bound="mfma"),
        "attn_bwd_dkv": dict(flops=3 * att, bytes=None, bound="mfma"),   # dP, dV, dK (S recompute not counted)
        "attn_bwd_dq": dict(flops=1 * att, bytes=None, bound="mfma"),    # dQ (S, dP recompute not counted)
        "gno_fwd_nh3": dict(flops=e_enc * 21280, bytes=e_enc * (32 + c * b) + m_lat * c * b, bound="hbm"),
        "gno_fwd_nh2": dict(flops=e_dec * 13088, bytes=e_dec * (32 + c * b) + n_pts * c * b, bound="hbm"),
        "gno_bwd_nh3": dict(flops=3 * e_enc * 21280, bytes=e_enc * (32 + 2 * c * b) + n_pts * c * b, bound="hbm"),
        "gno_bwd_nh2": dict(flops=3 * e_dec * 13088, bytes=e_dec * (32 + 2 * c * b) + m_lat * c * b, bound="hbm"),
    }


def cpu_baseline(layers, k, seed, atten_dropout):
    """The oracle (CPU restatement of the reference, pure PyTorch fp32) timed on the host cores on a bounded sample
    of the same workload: 1/16 of the points and 1/8 of the latent grid, same widths and depth."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gaot_oracle as orc  # timed CPU baseline only
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model import init_model
    n, latent = 62500, (32, 32, 16)
    cores = min(os.cpu_count() or 1, 16)   # more threads only add fork/join overhead on these op sizes
    torch.set_num_threads(cores)
    cfg = model_config(latent, layers, k)
    torch.manual_seed(seed)
    model = init_model(6, 1, "gaot_3d", cfg)
    sd = {kk: v.clone() for kk, v in model.state_dict().items()}
    batch, tokens = make_synthetic_sample(n, latent, k=k, seed=seed)
    s_tok = latent[0] * latent[1] * latent[2] // 8

    def one_step():
        drop = None
        if atten_dropout > 0.0:   # the training path draws one Bernoulli keep mask per attention call, as SDPA does
            masks = [torch.empty(1, 8, s_tok, s_tok).bernoulli_(1.0 - atten_dropout) for _ in range(layers)]
            drop = (masks, atten_dropout)
        orc.train_step_grads(sd, cfg, batch, tokens, drop=drop)

    one_step()  # warm-up
    times = []
    t_begin = time.perf_counter()
    while len(times) < 3 and time.perf_counter() - t_begin < 25.0:
        t0 = time.perf_counter()
        one_step()
        times.append(time.perf_counter() - t0)
    t = sorted(times)[0]
    return dict(value=n / t, unit="points/s", cores=cores, kind="port",
                sample=f"oracle fwd+MSE+bwd on N={n} points (1/8 of the sample), latent {latent[0]}x{latent[1]}x{latent[2]} "
                       f"(1/8), k={k}, L={layers}, d=256, attention dropout {atten_dropout}, fp32, best of {len(times)} "
                       f"after 1 warm-up ({t:.2f} s/step)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=500000)
    ap.add_argument("--latent", type=str, default="64,64,32")
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--knn", type=int, default=8)
    ap.add_argument("--precision", type=str, default=os.environ.get("GAOT_PRECISION", "bf16"), choices=["fp32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph", action="store_true", help="also use the hipGraph replay for N>1 (default: N=1 only)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--atten-dropout", type=float, default=0.1,
                    help="attention dropout of the training step (reference default AttentionConfig.atten_dropout = 0.1, "
                         "attn.py:22; no shipped config overrides it)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra dropout-free timing at N=1")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = every GPU owns --points points of one N x --points sample (default); strong = one "
                         "--points sample split over the N GPUs (BASELINE configs[2])")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    import torch.distributed as dist
    # GAOT_BENCH_ONE_DEVICE=1 (testing only): every rank on cuda:0 over gloo, to exercise the N>1 code path -- sharding,
    # exchange steps, max-over-ranks timing -- on a single-GPU box.  The number it prints is not a scaling result.
    one_device = os.environ.get("GAOT_BENCH_ONE_DEVICE", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if one_device else "nccl", init_method="env://")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import gaot_3d_amd
    from gaot_3d_amd import functional as GF
    from gaot_3d_amd import ops
    from gaot_3d_amd.data import make_synthetic_sample
    from gaot_3d_amd.model import init_model
    gaot_3d_amd.set_precision(args.precision)

    latent = tuple(int(v) for v in args.latent.split(","))
    cfg = model_config(latent, args.layers, args.knn, args.atten_dropout)
    torch.manual_seed(args.seed)
    model = init_model(6, 1, "gaot_3d", cfg).to(dev).train()
    use_graph = (not args.no_graph) and (world == 1 or args.graph)
    from gaot_3d_amd.optim import AdamW   # fused multi-tensor HIP step, same semantics as torch.optim.AdamW (tests)
    opt = AdamW(model.parameters(), lr=3e-4, weight_decay=1e-5)

    # Weak scaling (the path shards by mesh points): every GPU owns --points points of ONE sample of
    # world x --points points (N = 1: configs[1], 500 K points; N = 8: a 4 M-point sample, between configs[1] and the
    # 8-10 M-point configs[4]); latent grid, Transformer and parameters are replicated, as the north star prescribes.
    n_total = args.points * world if args.scaling == "weak" else args.points
    batch, tokens = make_synthetic_sample(n_total, latent, k=args.knn, seed=args.seed, device=str(dev))
    tokens = tokens.to(dev)
    if world > 1:
        from gaot_3d_amd import sharding
        batch = sharding.shard_batch(batch, rank, world, num_latent=tokens.shape[0])
        step_ctx = sharding.ShardedStep(model, dist.group.WORLD, n_total)
    else:
        step_ctx = None

    def step():
        gaot_3d_amd.clear_graph_cache(batch)
        opt.zero_grad(set_to_none=True)
        if step_ctx is None:
            pred = model(batch=batch, tokens_pos=tokens)
            loss = GF.mse_loss(pred, batch.x)
            loss.backward()
        else:
            loss = step_ctx.forward_backward(batch, tokens)
        opt.step()
        return loss

    # The step is ~600 short kernels; launched eagerly from Python the host becomes the bottleneck.  Capture ONE
    # whole step (CSR build, forward, loss, backward, gradient exchange, AdamW) into a hipGraph after the warm-up and
    # replay it: the timed region then measures the device work.  --no-graph times the eager launches instead.
    def measure():
        graph = None
        loss = None
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(max(args.warmup, 1)):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    loss = step()
            except Exception as ex:  # capture is a launch optimisation only; fall back to eager launches
                print(f"[bench] hipGraph capture failed ({type(ex).__name__}: {ex}); timing eager launches", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        if graph is None:
            for _ in range(args.warmup):
                step()
        else:
            for _ in range(args.warmup):
                graph.replay()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if graph is None:
                loss = step()
            else:
                graph.replay()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = tt.item()
        return elapsed, graph, loss

    elapsed, graph, loss = measure()
    secondary = None
    if world == 1 and args.atten_dropout > 0.0 and not args.no_secondary:
        # the same step with the dropout switched off (what the kernels do in eval mode / atten_dropout = 0)
        for mod in model.modules():
            if hasattr(mod, "atten_dropout"):
                mod.atten_dropout = 0.0
        e2, g2, _ = measure()
        secondary = dict(ms_per_step=e2 / args.steps * 1e3, value=n_total / (e2 / args.steps), atten_dropout=0.0)
        del g2
        for mod in model.modules():
            if hasattr(mod, "atten_dropout"):
                mod.atten_dropout = args.atten_dropout
    # per-kernel durations: HIP events around the instrumented launches of two more (eager) steps on the same stream
    ops.timing_reset(True)
    t_e = time.perf_counter()
    for _ in range(2):
        step()
    t_host = (time.perf_counter() - t_e) / 2
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t_e) / 2
    n_timed_steps = 2
    timing = ops.timing_summary()
    ops.timing_reset(False)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        m_lat = latent[0] * latent[1] * latent[2]
        s_tok = m_lat // 8
        e = n_total * args.knn
        work = algorithmic_work(n_total // world, m_lat, e // world, e // world, s_tok, args.layers)
        peaks = {"mfma": (157.3 if args.precision == "fp32" else 2500.0, "TFLOP/s"), "hbm": (8000.0, "GB/s")}
        per_kernel = {}
        for name, (calls, tot_ms) in timing.items():
            if name not in work or calls == 0:
                continue
            avg_s = tot_ms / calls * 1e-3
            w = work[name]
            ent = dict(calls_per_step=calls / n_timed_steps, avg_ms=tot_ms / calls, total_ms_per_step=tot_ms / n_timed_steps,
                       tflops=w["flops"] / avg_s / 1e12)
            if w["bytes"]:
                ent["gbps"] = w["bytes"] / avg_s / 1e9
            ent["bound"] = w["bound"]
            per_kernel[name] = ent
        dom = max(per_kernel, key=lambda kname: per_kernel[kname]["total_ms_per_step"]) if per_kernel else None
        roof = None
        if dom:
            d = per_kernel[dom]
            # GNO kernels: layer 0 always runs on exact-fp32 MFMA, the other layers on the precision's matrix rate;
            # report against whichever roof (matrix rate of the arithmetic used, or HBM) binds tighter
            if d["bound"] == "hbm" and d["tflops"] / peaks["mfma"][0] > d.get("gbps", 0) / 8000.0:
                ach, (peak, unit), bound = d["tflops"], peaks["mfma"], "mfma"
            elif d["bound"] == "hbm":
                ach, (peak, unit), bound = d["gbps"], peaks["hbm"], "hbm"
            else:
                ach, (peak, unit), bound = d["tflops"], peaks["mfma"], "mfma"
            roof = dict(kernel=dom, bound=bound, achieved=round(ach, 3), peak=peak, unit=unit, frac=round(ach / peak, 4),
                        traffic=None, avg_ms=round(d["avg_ms"], 4))
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    roof["traffic"] = json.load(open(pmc)).get(dom, {}).get("bytes_per_launch")
                except Exception:
                    pass
        out = {
            "metric": "mesh-points/sec fwd+bwd, 500K-pt DrivAerNet++ sample",
            "value": n_total / (elapsed / args.steps),
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{'configs[1]' if world == 1 else ('configs[1] x ' + str(world) + ' points' if args.scaling == 'weak' else 'configs[2] (one sample split over ' + str(world) + ' GPUs)')}: one {n_total}-point car-like surface sample (pos+normals), latent "
                                   f"{latent[0]}x{latent[1]}x{latent[2]}, knn k={args.knn} encoder + flipped decoder, "
                                   f"C=32 P=2 d=256 h=8 F=1024 L={args.layers} rope, attention dropout {args.atten_dropout} "
                                   f"(training mode), MSE + AdamW step; CSR build and geoembed stats inside the step",
                       "points": n_total, "latent_tokens": list(latent), "edges": e, "layers": args.layers,
                       "precision": args.precision, "points_per_gpu": n_total // world, "atten_dropout": args.atten_dropout,
                       "sharding": f"point-shard x{world}; latent grid / Transformer replicated, attention heads split over the "
                                   f"ranks (all-gather of head outputs)" if world > 1 else "none"},
            "loss": float(loss.detach()),
            "launch": "hipGraph replay of one captured step" if graph is not None else "eager",
            # wall time of the two instrumented eager steps (only meaningful without a captured graph: right after a
            # capture the eager allocations go back to hipMalloc)
            "instrumented_eager_ms_per_step": round(t_eager * 1e3, 3) if graph is None else None,
            "instrumented_eager_host_ms_per_step": round(t_host * 1e3, 3) if graph is None else None,
            "without_attention_dropout": secondary,
            "roofline": roof,
            "kernels": {kname: {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in ent.items()}
                        for kname, ent in per_kernel.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.layers, args.knn, args.seed, args.atten_dropout)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
