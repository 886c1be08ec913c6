import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from gaot_3d_amd import ops
from test_ops_gpu import gen
DEV = "cuda:0"
b, s, h, hkv, p = 4, 2048, 8, 8, 0.0
qkv = (gen(b * s, (h + 2 * hkv) * 32, seed=s + h) * 0.7).to(DEV)
freqs = (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))).to(DEV)
big = qkv.clone()
big[5, h * 32:h * 32 + 32] = 200.0
big[:, :32] *= 4.0
outs = {}
for mode in ("1", "0"):
    os.environ["GAOT_ATTN_FWD_ASM"] = mode
    o, lse, img = ops.attn_fwd_bf16(big, freqs, b, s, h, hkv, 32 ** -0.5, p, None)
    torch.cuda.synchronize()
    outs[mode] = (o.clone(), lse.clone())
o1, o0 = outs["1"][0].view(b, s, h, 32), outs["0"][0].view(b, s, h, 32)
err = (o1 - o0).abs().amax(dim=3)      # [b, s, h]
for bb in range(b):
    for hh in range(h):
        e = err[bb, :, hh]
        if float(e.max()) > 1e-3:
            bad = (e > 1e-3).nonzero().flatten()
            print(f"batch {bb} head {hh}: max err {float(e.max()):.3e}, {len(bad)} bad rows, first {bad[:6].tolist()} last {bad[-3:].tolist()}")
print("lse err", float((outs['1'][1] - outs['0'][1]).abs().max()))
