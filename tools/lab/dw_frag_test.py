"""(lab, run through `python tools/lab/dw_frag_lab.py test`) Weight gradients from operands in MFMA-fragment order (tools/lab/dw_frag.hip; reference attn.py:110-157: the .grad of the nn.Linear weights
of a Transformer block) against the same product in fp64 on the bf16-rounded operands, and against the generic weight-gradient GEMM."""
import pytest
import torch

DEV = "cuda:0"


def _timg_reference(x: torch.Tensor) -> torch.Tensor:
    """the T-image of include/gaot3d_hip.h restated with index arithmetic: [groups, KS, 4, 64 lanes, 8] bf16"""
    rows, cols = x.shape
    rp = (rows + 63) // 64 * 64
    xp = torch.zeros(rp, cols, dtype=torch.bfloat16, device=x.device)
    xp[:rows] = x.to(torch.bfloat16)
    ks = rp // 16
    lane = torch.arange(64, device=x.device)
    j = torch.arange(8, device=x.device)
    row_in = 8 * (j[None, :] // 4) + 4 * (lane[:, None] // 32) + (j[None, :] % 4)           # [64, 8]
    col_in = (lane % 32)[:, None].expand(64, 8)
    t = xp.view(ks, 16, cols // 128, 4, 32)                                                   # [ks, r, g, c, l31]
    out = t[:, row_in, :, :, col_in]                                                          # [64, 8, ks, g, c] (advanced indices first)
    return out.permute(3, 2, 4, 0, 1).contiguous()                                            # [g, ks, c, 64, 8]


@pytest.mark.parametrize("rows,cols,dtype", [(64, 128, torch.float32), (1000, 256, torch.bfloat16), (4096, 768, torch.float32)])
def test_timg_pack_layout(rows, cols, dtype):
    import __main__ as lab
    from gaot_3d_amd import ops
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device=DEV).to(dtype)
    img = lab.timg_pack(x)
    ref = _timg_reference(x)
    assert torch.equal(img.view(torch.bfloat16).view(ref.shape), ref)


@pytest.mark.parametrize("rows,n1,n2", [(16384, 2048, 256), (16384, 256, 1024), (16384, 768, 256), (16384, 256, 256), (1000, 256, 128), (64, 128, 128),
                                        (4136, 384, 256)])
def test_dw_frag_against_fp64(rows, n1, n2):
    import __main__ as lab
    from gaot_3d_amd import ops
    torch.manual_seed(1)
    a = torch.randn(rows, n1, device=DEV).bfloat16()
    b = torch.randn(rows, n2, device=DEV).bfloat16()
    dw = lab.dw_frag(lab.timg_pack(a), lab.timg_pack(b), rows, n1, n2)
    torch.cuda.synchronize()
    ref = a.double().t() @ b.double()
    err = (dw.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[parity] dw_frag rows={rows} [{n1} x {n2}]: max diff / peak vs fp64 on the same bf16 operands {err:.2e}")
    assert err < 2e-6      # fp32 accumulation of exact bf16 products
    again = lab.dw_frag(lab.timg_pack(a), lab.timg_pack(b), rows, n1, n2)
    assert torch.equal(dw, again)
    gen = ops.gemm_dw(a, b, n1, n2, rows, n1, n2, 1)
    err2 = (dw - gen).abs().max().item() / ref.abs().max().item()
    assert err2 < 2e-6
