#!/usr/bin/env python3
"""Kernel microbenchmarks at the configs[1] shapes (for rocprofv3 / quick A-B): attention fwd+bwd, GEMM shapes, GNO."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gaot_3d_amd
from gaot_3d_amd import ops, functional as GF

what = sys.argv[1] if len(sys.argv) > 1 else "attn"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda:0"
gaot_3d_amd.set_precision(os.environ.get("GAOT_PRECISION", "bf16"))
torch.manual_seed(0)


def timeit(fn, name, flops=None):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name}: {dt*1e3:.3f} ms" + (f"  {flops/dt/1e12:.1f} TF/s" if flops else ""))


if what == "attn":
    b, s, h = 1, int(os.environ.get('MB_S', 16384)), int(os.environ.get('MB_H', 8))
    qkv = torch.randn(b * s, 3 * h * 32, device=dev)
    freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(dev)
    d_o = torch.randn(b * s, h * 32, device=dev)
    att = 2 * s * s * 32 * h
    pd = float(os.environ.get("MB_DROP", 0.0))     # attention dropout probability
    sd = torch.tensor([12345], dtype=torch.int64, device=dev) if pd > 0 else None
    fused = {"0": False, "1": True}.get(os.environ.get("MB_FUSED"))     # None: the library's choice
    if ops.get_precision() == "bf16":
        o, lse, img = ops.attn_fwd_bf16(qkv, freqs, b, s, h, h, 32 ** -0.5, pd, sd)
        timeit(lambda: ops.attn_fwd_bf16(qkv, freqs, b, s, h, h, 32 ** -0.5, pd, sd), "attn_fwd_bf16(+prep)", 2 * att)
        timeit(lambda: ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, h, 32 ** -0.5, pd, sd, fused=fused), "attn_bwd_bf16(all)", 4 * att)
        ops.timing_reset(True)
        for _ in range(reps):
            ops.attn_fwd_bf16(qkv, freqs, b, s, h, h, 32 ** -0.5, pd, sd)
            ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, h, 32 ** -0.5, pd, sd, fused=fused)
        torch.cuda.synchronize()
        for name, (calls, tot) in ops.timing_summary().items():
            print(f"  {name}: {tot / calls:.4f} ms")
        ops.timing_reset(False)
    else:
        o, lse = ops.attn_fwd(qkv, b, s, h, h, 32 ** -0.5)
        timeit(lambda: ops.attn_fwd(qkv, b, s, h, h, 32 ** -0.5), "attn_fwd_f32", 2 * att)
        timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, b, s, h, h, 32 ** -0.5), "attn_bwd_f32", 4 * att)
elif what == "gemm":
    m = 16384
    for (n, k) in ((768, 256), (256, 256), (2048, 256), (256, 1024), (256, 512)):
        x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev); dy = torch.randn(m, n, device=dev)
        fl = 2 * m * n * k
        timeit(lambda: ops.gemm(x, w, m, n, k, k, k, False, True), f"fwd  x[{m},{k}] W[{n},{k}]^T", fl)
        timeit(lambda: ops.gemm(dy, w, m, k, n, n, k, False, False), f"dx   dy[{m},{n}] W[{n},{k}]", fl)
        timeit(lambda: ops.gemm(dy, x, n, k, m, n, k, True, False), f"dW   dy^T[{n},{m}] x[{m},{k}]", fl)
        wb = w.bfloat16()
        timeit(lambda: ops.gemm(x, wb, m, n, k, k, k, False, True), f"fwd  (bf16 W)", fl)
        timeit(lambda: ops.gemm(dy, wb, m, k, n, n, k, False, False), f"dx   (bf16 W)", fl)
        timeit(lambda: ops.gemm(x, wb, m, n, k, k, k, False, True, out_dtype=torch.bfloat16), f"fwd  (bf16 W, bf16 C)", fl)
        xb = x.bfloat16()
        timeit(lambda: ops.gemm(xb, wb, m, n, k, k, k, False, True), f"fwd  (bf16 x, bf16 W)", fl)
        mb = (m * k + m * n + n * k) * 4 / 1e6
        print(f"     (operand + result bytes {mb:.0f} MB -> {mb / 5e3 * 1e3:.1f} us at 5 TB/s)")
elif what == "gemm16":
    # the bf16-in-memory GEMMs of one block at configs[1] (A and B bf16): dx of w1|w3, w2 forward
    m = 16384
    for (n, k, bt, name) in ((256, 2048, False, "dx_w13  dag[16384,2048] W13[2048,256]"), (256, 1024, True, "w2 fwd  u[16384,1024] W2[256,1024]^T")):
        a = torch.randn(m, k, device=dev).bfloat16()
        w = (torch.randn(n, k, device=dev) if bt else torch.randn(k, n, device=dev)).bfloat16()
        fl = 2 * m * n * k
        timeit(lambda: ops.gemm(a, w, m, n, k, k, w.shape[1], False, bt), name, fl)
elif what == "wgrad":
    # weight gradients of one block at configs[1]: dW = dy^T x over 16384 tokens, both operands bf16 in memory.  MB_SETS > 1
    # rotates through that many operand sets (6 x 75 MB do not fit the 256-MB Infinity Cache: operands come from HBM, as in the step)
    rows = 16384
    nsets = int(os.environ.get("MB_SETS", 1))
    for (m, n, name) in ((2048, 256, "w1|w3"), (256, 1024, "w2"), (768, 256, "q|k|v"), (256, 256, "o_proj"), (256, 512, "skip_proj")):
        sets = [((torch.randn(rows, m, device=dev) * 0.5).bfloat16(), (torch.randn(rows, n, device=dev) * 0.5).bfloat16()) for _ in range(nsets)]
        a, b = sets[0]
        c = ops.gemm(a, b, m, n, rows, m, n, True, False)
        ref = a.float().t() @ b.float()
        err = (c - ref).abs().max().item() / ref.abs().max().item()
        it = [0]
        def run():
            x, y = sets[it[0] % nsets]
            it[0] += 1
            return ops.gemm(x, y, m, n, rows, m, n, True, False)
        timeit(run, f"dW {name:9s} [{m} x {n}] rel err {err:.1e}", 2 * rows * m * n)
elif what == "graph":
    # device graph construction at configs[1]: 500 000 surface points against the 64x64x32 token grid
    from gaot_3d_amd import graph
    from gaot_3d_amd.data import make_synthetic_sample
    batch, tokens = make_synthetic_sample(500000, (64, 64, 32), k=8, seed=0, device=dev)
    tokens = tokens.to(dev)
    pos = batch.pos
    g = graph.as_latent_grid(tokens, (64, 64, 32))
    timeit(lambda: graph.knn_to_grid(pos, g, 8), "knn_to_grid k=8 (E = 4.0 M)")
    timeit(lambda: graph.knn_to_grid(pos, g, 1), "knn_to_grid k=1")
    for r in (0.033, 0.05):
        e = graph._decoder_edges("radius", pos, g, r, 1)
        timeit(lambda: graph._decoder_edges("radius", pos, g, r, 1), f"radius r={r} centres=phys (E = {e.shape[1]})")
        e = graph._encoder_edges("radius", pos, g, r, 1)
        timeit(lambda: graph._encoder_edges("radius", pos, g, r, 1), f"radius r={r} centres=latent, cap 32 per token (E = {e.shape[1]})")
        e = graph._encoder_edges("bidirectional", pos, g, r, 1)
        timeit(lambda: graph._encoder_edges("bidirectional", pos, g, r, 1), f"bidirectional k=1 r={r} encoder (E = {e.shape[1]})")
elif what == "gno":
    from gaot_3d_amd.data import make_synthetic_sample
    batch, tokens = make_synthetic_sample(500000, (64, 64, 32), k=8, seed=0, device=dev)
    n, m = 500000, tokens.shape[0]
    tokens = tokens.to(dev)
    for nh, ei, ns, nd, yp, xp in ((3, batch.encoder_edge_index_s0, n, m, batch.pos, tokens),
                                   (2, batch.decoder_edge_index_s0, m, n, tokens, batch.pos),
                                   (4, batch.encoder_edge_index_s0, n, m, batch.pos, tokens)):
        ws = [torch.randn(64, 6, device=dev) * 0.3] + [torch.randn(64, 64, device=dev) * 0.1 for _ in range(nh - 1)] + [torch.randn(32, 64, device=dev) * 0.1]
        bs = [torch.zeros(w.shape[0], device=dev) for w in ws]
        timeit(lambda: ops.build_graph(ei, ns, nd), f"csr x2 (E={ei.shape[1]})")
        g = ops.build_graph(ei, ns, nd)
        f = torch.randn(ns, 32, device=dev); go = torch.randn(nd, 32, device=dev)
        e = ei.shape[1]
        fl = e * {2: 13088, 3: 21280, 4: 29472}[nh]
        timeit(lambda: ops.gno_forward(ws, bs, yp, xp, f, g), f"gno_fwd nh={nh}", fl)
        timeit(lambda: ops.gno_backward(ws, bs, yp, xp, f, go, g), f"gno_bwd nh={nh}", 3 * fl)
