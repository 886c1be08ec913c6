"""Shared comparison helpers of the GPU parity tests.  Every helper prints one `[parity] ...` line with the ACHIEVED
errors (max-abs, max-abs relative to the reference's peak, relative L2); tests/conftest.py collects those lines into the
parity log that tools/gpu_pass.sh copies to profiles/parity_<tag>.txt."""
import torch


def _f(t):
    return t.detach().double().cpu()


def stats(a, b):
    a, b = _f(a), _f(b)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    peak = b.abs().max().item() if b.numel() else 0.0
    l2 = ((a - b).norm() / b.norm().clamp(min=1e-300)).item() if b.numel() else 0.0
    return err, peak, l2


def close(name, a, b, rtol, atol):
    """torch.allclose(a, b, rtol, atol) -- the fp32 bar of SURVEY §8d"""
    assert tuple(a.shape) == tuple(b.shape), (name, a.shape, b.shape)
    err, peak, l2 = stats(a, b)
    print(f"[parity] {name}: max_abs={err:.3e} ref_peak={peak:.3e} max_rel_to_peak={err / max(peak, 1e-300):.3e} rel_l2={l2:.3e}")
    assert torch.allclose(_f(a), _f(b), rtol=rtol, atol=atol), f"{name}: max abs err {err:.3e} (peak {peak:.3e})"


def close_peak(name, a, b, rel, rel_l2=None):
    """max|a - b| <= rel * max|b| (and optionally ||a - b|| <= rel_l2 * ||b||): a bound that scales with the signal, so it
    keeps its teeth when the reference values are small (an untrained model predicts O(1e-2))"""
    assert tuple(a.shape) == tuple(b.shape), (name, a.shape, b.shape)
    err, peak, l2 = stats(a, b)
    print(f"[parity] {name}: max_abs={err:.3e} ref_peak={peak:.3e} max_rel_to_peak={err / max(peak, 1e-300):.3e} rel_l2={l2:.3e} "
          f"(bound {rel:.1e} of peak{'' if rel_l2 is None else f', rel_l2 {rel_l2:.1e}'})")
    assert torch.isfinite(_f(a)).all(), f"{name}: non-finite values"
    assert err <= rel * peak + 1e-30, f"{name}: max abs err {err:.3e} > {rel:.1e} x peak {peak:.3e}"
    if rel_l2 is not None:
        assert l2 <= rel_l2, f"{name}: relative L2 error {l2:.3e} > {rel_l2:.1e}"


def cosine(name, a, b, bound):
    a, b = _f(a).flatten(), _f(b).flatten()
    c = float(a @ b / (a.norm() * b.norm()).clamp(min=1e-300))
    err, peak, l2 = stats(a, b)
    print(f"[parity] {name}: cosine={c:.6f} max_abs={err:.3e} ref_peak={peak:.3e} rel_l2={l2:.3e}")
    assert torch.isfinite(a).all(), f"{name}: non-finite values"
    assert c >= bound, f"{name}: cosine {c:.6f} < {bound}"
    return c


def grads_cosine(name, got: dict, ref: dict, bound, per_tensor=None, energy=1e-6):
    """overall cosine of the concatenated gradients, and per-tensor cosine for every tensor that carries more than
    ``energy`` of the gradient energy"""
    keys = [k for k in ref if k in got]
    a = torch.cat([_f(got[k]).flatten() for k in keys])
    r = torch.cat([_f(ref[k]).flatten() for k in keys])
    c = cosine(name, a, r, bound)
    if per_tensor is not None:
        total = float(r.norm() ** 2)
        for k in keys:
            x, y = _f(got[k]).flatten(), _f(ref[k]).flatten()
            if float(y.norm() ** 2) > energy * total:
                ck = float(x @ y / (x.norm() * y.norm() + 1e-300))
                assert ck >= per_tensor, (name, k, ck)
    return c


def unit_scale_last_layer(weight, bias, pred_std):
    """divide the model's last affine map by the spread of its current predictions: an untrained model then predicts O(1)
    values, so loss and prediction comparisons are sensitive to the forward arithmetic (same map on both sides)"""
    with torch.no_grad():
        weight.div_(pred_std)
        if bias is not None:
            bias.div_(pred_std)
