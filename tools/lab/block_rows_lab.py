"""round 6 lab: one TransformerBlock (forward + backward, bf16 mode, graph replay) at the row counts a rank of a sequence-parallel step
sees (S / G rows for the row-wise operators; attention excluded by timing the block with the attention kernels' share measured apart):
the row-block kernels (csrc/ffn_fused.hip) against the per-operator launches, to place the switch-over row count"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaot_3d_amd
from gaot_3d_amd import functional as GF
from gaot_3d_amd.model.layers import attn as A

dev = "cuda:0"
gaot_3d_amd.set_precision("bf16")
SW = ("_FFN_FUSED", "_FFN_BWD_DX", "_NORM_FFN", "_BLOCK_TAIL", "_OPROJ_BWD_IMAGE", "_NORM_QKV", "_NORM_BWD_FUSED", "_CAT_QKV", "_CAT_BWD_DX")


def run(rows, fused, skip):
    for k in SW:
        setattr(GF, k, fused)
    torch.manual_seed(0)
    blk = A.TransformerBlock(256, 256, attn_config=A.AttentionConfig(hidden_size=256, num_heads=8, num_kv_heads=8, atten_dropout=0.0,
                                                                      positional_embedding="rope"),
                             ffn_config=A.FFNConfig(hidden_size=1024), skip_connection=skip).to(dev).train()
    x = torch.randn(1, rows, 256, device=dev, requires_grad=True)
    sk = torch.randn(1, rows, 256, device=dev, requires_grad=True) if skip else None

    def step():
        for p in blk.parameters():
            p.grad = None
        x.grad = None
        y = blk(x, relative_positions=True, skip=sk) if skip else blk(x, relative_positions=True)
        y.square().mean().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            step()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 20 * 1e3


for skip in (False, True):
    for rows in (1024, 2048, 4096, 8192, 16384):
        t0, t1 = run(rows, False, skip), run(rows, True, skip)
        print(f"rows={rows:6d} skip_proj={int(skip)}: per-operator launches {t0:8.1f} us   row-block kernels {t1:8.1f} us   ({t1 - t0:+.1f})")
