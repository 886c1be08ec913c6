#!/usr/bin/env python3
"""Run the bf16 attention backward on fixed inputs and save d(q|k|v): two runs with different GAOT_ATTN_BWD_VARIANT values are
compared with `cmp` mode.  usage: attn_bwd_variant_check.py run <out.pt> | cmp <a.pt> <b.pt>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

CASES = [(1, 16384, 8, 8, 0.1), (1, 16384, 8, 8, 0.0), (1, 15935, 8, 8, 0.1), (2, 8192, 8, 4, 0.1), (1, 16384, 4, 4, 0.1)]

if sys.argv[1] == "run":
    import gaot_3d_amd
    from gaot_3d_amd import ops
    gaot_3d_amd.set_precision("bf16")
    out = {}
    for (b, s, h, hkv, p) in CASES:
        g = torch.Generator().manual_seed(s + h)
        qkv = torch.randn(b * s, (h + 2 * hkv) * 32, generator=g).cuda()
        d_o = torch.randn(b * s, h * 32, generator=g).cuda()
        freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).cuda()
        sd = torch.tensor([777], dtype=torch.int64, device="cuda") if p > 0 else None
        o, lse, img = ops.attn_fwd_bf16(qkv, freqs, b, s, h, hkv, 32 ** -0.5, p, sd)
        d1 = ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, hkv, 32 ** -0.5, p, sd, freqs, fused=True)
        d2 = ops.attn_bwd_bf16(img, o, d_o, lse, b, s, h, hkv, 32 ** -0.5, p, sd, freqs, fused=True)
        torch.cuda.synchronize()
        out[(b, s, h, hkv, p)] = (d1.cpu(), bool(torch.equal(d1, d2)), bool(torch.isfinite(d1).all()))
    torch.save(out, sys.argv[2])
else:
    a, b_ = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        (x, rep_x, fin_x), (y, rep_y, fin_y) = a[k], b_[k]
        h, hkv = k[2], k[3]
        cols = {"dq": (0, h * 32), "dk": (h * 32, (h + hkv) * 32), "dv": ((h + hkv) * 32, (h + 2 * hkv) * 32)}
        msg = []
        for nm, (lo, hi) in cols.items():
            u, v = x[:, lo:hi].double().flatten(), y[:, lo:hi].double().flatten()
            cos = float(u @ v / (u.norm() * v.norm()))
            msg.append(f"{nm} cos {cos:.7f} max|d| {float((u - v).abs().max()):.3e} of peak {float(u.abs().max()):.3e}")
        print(k, f"bit-identical rerun {rep_x}/{rep_y} finite {fin_x}/{fin_y}", "; ".join(msg))
