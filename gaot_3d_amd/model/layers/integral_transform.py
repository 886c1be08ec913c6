"""IntegralTransform with the reference's constructor/forward signature
(src/model/layers/integral_transform.py:31-40, 80-87).

Default configuration -- transform_type='linear', no attention weights, kernel MLP 6 -> 64 (x1..4) -> 32 -- runs the
fused HIP kernels (csrc/gno.hip, gno_bf16.hip: no per-edge tensor ever reaches HBM).  Every other variant the
reference offers ('nonlinear', 'nonlinear_kernelonly', use_attn cosine / dot_product, other MLP shapes, f_y=None)
runs the general path: per-edge tensors in dst-sorted order, the MLP on the GEMM kernels, gather / segment /
element-wise glue and its autograd in csrc/edgeops.hip (gaot_3d_amd/edgeops.py).  HIP only, no CPU fallback."""
from typing import Optional

import torch
import torch.nn as nn

from ... import edgeops as EO
from ... import functional as GF
from ... import ops
from .mlp import LinearChannelMLP, activation_name


def _is_flip_of_encoder(cache_owner, slot, dec_ei: torch.Tensor, enc_ei: torch.Tensor) -> bool:
    """Is this decoder edge list the encoder's with its two rows swapped (the shipped configuration: knn encoder, "flipped"
    decoder)?  A fact about the SAMPLE's inputs, found once per pair of tensors (object identity + in-place version) with two
    device comparisons and kept on the batch beside -- not inside -- the neighbour-list cache, so `clear_graph_cache` (a step that
    rebuilds its lists) does not repeat the host synchronisation.  Never evaluated while a hipGraph is being captured."""
    facts = cache_owner.__dict__.setdefault("_gaot_flip", {})
    ent = facts.get(slot)
    if ent is not None and ent[0] is dec_ei and ent[1] == dec_ei._version and ent[2] is enc_ei and ent[3] == enc_ei._version:
        return ent[4]
    if dec_ei.is_cuda and torch.cuda.is_current_stream_capturing():
        return False
    flip = bool(dec_ei.shape == enc_ei.shape and dec_ei.dtype == enc_ei.dtype and dec_ei.device == enc_ei.device
                and torch.equal(dec_ei[0], enc_ei[1]) and torch.equal(dec_ei[1], enc_ei[0]))
    facts[slot] = (dec_ei, dec_ei._version, enc_ei, enc_ei._version, flip)
    return flip


def graph_for(edge_index: torch.Tensor, num_src: int, num_dst: int, cache_owner=None, key=None):
    """Row-sorted neighbour lists for one edge_index; cached on the batch object so that encoder GNO, GeoEmbed and the
    backward pass share one build per sample and scale.  An entry is valid only for the very tensor it was built from
    (object identity and in-place version counter; the entry keeps the tensor alive, so its address cannot be handed to
    another edge list meanwhile): edges rebuilt or re-uploaded per forward replace the entry of their slot, they never
    hit a stale one, and the cache holds one entry per (side, scale).
    A decoder list that is the encoder's with the rows swapped needs no build of its own: sorted by query it IS the encoder's
    list sorted by source and vice versa (same edge order, same stable sort: identical arrays), so the decoder's graph shares the
    encoder's two lists -- two radix-sort builds per step instead of four (0.36 ms at E = 4 M, ~5 ms at E = 64 M)."""
    if cache_owner is not None:
        cache = cache_owner.__dict__.setdefault("_gaot_graphs", {})
        k = (key, num_src, num_dst)
        ent = cache.get(k)
        if ent is not None and ent[0] is edge_index and ent[1] == edge_index._version:
            return ent[2]
        g = None
        if isinstance(key, tuple) and len(key) == 2 and key[0] == "dec":
            enc = cache.get((("enc", key[1]), num_dst, num_src))
            if (enc is not None and enc[1] == enc[0]._version
                    and _is_flip_of_encoder(cache_owner, key[1], edge_index, enc[0])):
                ge = enc[2]
                g = ops.BipartiteGraph(ge.by_src, ge.by_dst, num_src, num_dst)
        if g is None:
            g = ops.build_graph(edge_index, num_src, num_dst)
        cache[k] = (edge_index, edge_index._version, g)
        return g
    return ops.build_graph(edge_index, num_src, num_dst)


class IntegralTransform(nn.Module):
    def __init__(self, channel_mlp=None, channel_mlp_layers=None, channel_mlp_non_linearity="gelu",
                 transform_type="linear", use_attn=None, coord_dim=None, attention_type="cosine"):
        """``channel_mlp_non_linearity``: the reference's callable (F.gelu, integral_transform.py:35) or its name"""
        super().__init__()
        self.transform_type = transform_type
        self.use_attn = use_attn
        self.coord_dim = coord_dim
        self.attention_type = attention_type
        if channel_mlp is None:
            if channel_mlp_layers is None:
                raise ValueError("Need channel_mlp or layers")
            self.channel_mlp = LinearChannelMLP(layers=channel_mlp_layers, non_linearity=channel_mlp_non_linearity)
        else:
            self.channel_mlp = channel_mlp
        if self.use_attn:
            if coord_dim is None:
                raise ValueError("coord_dim must be specified when use_attn is True")
            if attention_type == "dot_product":
                self.query_proj = nn.Linear(coord_dim, 64)
                self.key_proj = nn.Linear(coord_dim, 64)
                self.scaling_factor = 1.0 / (64 ** 0.5)
            elif attention_type != "cosine":
                raise ValueError(f"Invalid attention_type: {attention_type}. Must be 'cosine' or 'dot_product'.")

    def forward(self, y_pos, x_pos, edge_index, f_y: Optional[torch.Tensor] = None, weights=None, batch_y=None,
                batch_x=None, graph=None):
        """y_pos [N_y,3] source coords, x_pos [N_x,3] query coords, edge_index [2,E] (row 0 -> y, row 1 -> x),
        f_y [N_y,C].  ``graph`` (optional) = prebuilt neighbour lists for edge_index."""
        if self.transform_type not in ("linear", "nonlinear", "nonlinear_kernelonly"):
            raise ValueError(f"Invalid transform_type: {self.transform_type}")
        fcs = list(self.channel_mlp.fcs)
        if edge_index is not None and edge_index.shape[1] == 0 and graph is None:   # integral_transform.py:106-112
            return torch.zeros(x_pos.shape[0], fcs[-1].weight.shape[0], dtype=x_pos.dtype, device=x_pos.device)
        if graph is None:
            graph = graph_for(edge_index.to(x_pos.device), y_pos.shape[0], x_pos.shape[0])
        plan = self._fused_plan(fcs, f_y, y_pos)
        if plan is not None:
            return self._forward_fused(fcs, y_pos, x_pos, f_y, graph, *plan)
        return self._forward_general(fcs, y_pos, x_pos, f_y, graph)

    def _fused_plan(self, fcs, f_y, y_pos):
        """(coord_dim, channels) when the fused kernels (csrc/gno*.hip: coordinates of dimension 3, hidden width 64, 32
        channels per pass) can evaluate this transform EXACTLY, possibly through zero padding; None -> general path.
        Kernel MLP coord-pair -> 64 (x 1..4; with gradients in fp32 mode: 1..3) -> C with GELU, 'linear' transform, mean reduction:
          * coord_dim 1 / 2 (the reference's default is 2, magno.py:28): coordinates padded with zeros to 3-D and the first
            layer's weight given zero columns for them -- the products that are added are exactly 0;
          * C != 32 (the reference's default is 16, magno.py:25): the last layer / f_y / the output are cut into blocks of 32
            channels (the last one zero-padded); every block is one pass of the 32-channel kernels, so C = 16 costs what
            C = 32 costs and C = 64 two passes (the hidden layers are recomputed), still without a per-edge tensor in HBM."""
        if self.use_attn or self.transform_type != "linear" or f_y is None:
            return None
        # the exact-fp32 backward keeps the hidden activations of 128 edges in LDS: three hidden layers at most when gradients
        # are needed in fp32 mode; the bf16 backward takes four (its operand fragments then come from L2, gno_bwd3_bf16.hip)
        need_grad = torch.is_grad_enabled() and (f_y.requires_grad or any(fc.weight.requires_grad for fc in fcs))
        from ... import ops as _ops
        max_fcs = 4 if (need_grad and _ops.get_precision() != "bf16") else 5
        if activation_name(getattr(self.channel_mlp, "non_linearity", "gelu")) != "gelu" or not (2 <= len(fcs) <= max_fcs):
            return None
        if self.training and getattr(self.channel_mlp, "dropout_p", 0.0) > 0.0:
            return None
        dims = [(fc.weight.shape[0], fc.weight.shape[1]) for fc in fcs]
        cd = y_pos.shape[1]
        c = dims[-1][0]
        # hidden widths: the kernels are written for 64; any width <= 64 (the lists in_/out_gno_channel_mlp_hidden_layers are free,
        # magno.py:32,36) is zero-padded to it -- gelu(0) = 0 and the padded rows / columns of the neighbouring weights are 0,
        # so every added product is exactly 0 (a width of 32 then costs what 64 costs; wider layers take the general path)
        chain = all(dims[i][1] == dims[i - 1][0] for i in range(1, len(dims)))
        ok = (cd in (1, 2, 3) and dims[0][1] == 2 * cd and chain and all(1 <= d[0] <= 64 for d in dims[:-1])
              and 1 <= c <= 256 and f_y.shape[1] == c and all(fc.bias is not None for fc in fcs))
        return (cd, c) if ok else None

    def _fused_eligible(self, fcs, f_y) -> bool:
        """the transform runs on the fused kernels as it is (the shipped configuration): no padding, one pass"""
        dims = [(fc.weight.shape[0], fc.weight.shape[1]) for fc in fcs]
        return (not self.use_attn and self.transform_type == "linear" and f_y is not None and dims[0] == (64, 6)
                and dims[-1] == (32, 64) and f_y.shape[1] == 32)

    def _forward_fused(self, fcs, y_pos, x_pos, f_y, graph, cd: int, c: int):
        w2 = lambda fc: fc.weight[:, :, 0] if fc.weight.dim() == 3 else fc.weight    # Conv1d(k=1) storage of mlp_type='channel'
        pad_hidden = any(fc.weight.shape[0] != 64 for fc in fcs[:-1])
        if cd == 3 and c == 32 and not pad_hidden:                                    # the shipped shape: straight through
            params = []
            for fc in fcs:
                params += [fc.weight, fc.bias]
            return GF.GnoFn.apply(f_y, y_pos, x_pos, graph, *params)
        y3, x3, w0 = y_pos, x_pos, w2(fcs[0])
        if cd < 3:
            y3 = torch.nn.functional.pad(y_pos, (0, 3 - cd))
            x3 = torch.nn.functional.pad(x_pos, (0, 3 - cd))
            # [src coords | 0.. | query coords | 0..] by concatenation: no index tensor built on the host (a pageable
            # host-to-device copy per forward is a synchronisation and is illegal inside a hipGraph capture)
            z = w0.new_zeros(w0.shape[0], 3 - cd)
            w0 = torch.cat([w0[:, :cd], z, w0[:, cd:2 * cd], z], dim=1)
        pad2 = lambda w, rows, cols: w if (w.shape[0] == rows and w.shape[1] == cols) else torch.nn.functional.pad(
            w, (0, cols - w.shape[1], 0, rows - w.shape[0]))
        pad1 = lambda v, n: v if v.shape[0] == n else torch.nn.functional.pad(v, (0, n - v.shape[0]))
        w0, b0 = pad2(w0, 64, 6), pad1(fcs[0].bias, 64)
        mid = []
        for fc in fcs[1:-1]:
            mid += [pad2(w2(fc), 64, 64), pad1(fc.bias, 64)]
        wl, bl = w2(fcs[-1]), fcs[-1].bias
        wl = pad2(wl, wl.shape[0], 64)
        outs = []
        for c0 in range(0, c, 32):
            n = min(32, c - c0)
            wb, bb, fb = wl[c0:c0 + n], bl[c0:c0 + n], f_y[:, c0:c0 + n]
            if n < 32:
                wb = torch.nn.functional.pad(wb, (0, 0, 0, 32 - n))
                bb = torch.nn.functional.pad(bb, (0, 32 - n))
                fb = torch.nn.functional.pad(fb, (0, 32 - n))
            o = GF.GnoFn.apply(fb.contiguous(), y3, x3, graph, w0, b0, *mid, wb.contiguous(), bb.contiguous())
            outs.append(o[:, :n] if n < 32 else o)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    def _forward_general(self, fcs, y_pos, x_pos, f_y, g):
        """integral_transform.py:114-171 on per-edge tensors (dst-sorted order).  Coordinates of dimension 1 / 2 (the
        reference's default is ``gno_coord_dim: 2``, magno.py:28) are padded with zeros to the kernels' 3-D rows and the first
        layer's weight gets zero columns for the padding: every added product is exactly 0, as in ``_fused_plan``."""
        tt = self.transform_type
        act = activation_name(getattr(self.channel_mlp, "non_linearity", "gelu"))
        pdrop = getattr(self.channel_mlp, "dropout_p", 0.0)
        cdp = y_pos.shape[1]
        if cdp > 3 or x_pos.shape[1] != cdp:
            raise NotImplementedError(f"coordinates of dimension {tuple(y_pos.shape[1:])} / {tuple(x_pos.shape[1:])}: 1..3 supported")
        y3, x3 = y_pos, x_pos
        if cdp < 3:
            y3 = torch.nn.functional.pad(y_pos, (0, 3 - cdp))
            x3 = torch.nn.functional.pad(x_pos, (0, 3 - cdp))
        with_f = f_y is not None and tt != "linear"
        k = EO.EdgeInputFn.apply(y3, x3, f_y if with_f else None, g)                                        # :146-152
        for i, fc in enumerate(fcs):                                                                        # :154
            w = fc.weight[:, :, 0] if fc.weight.dim() == 3 else fc.weight
            if i == 0 and cdp < 3:      # [src coords | query coords | features] -> [src, 0.. | query, 0.. | features]
                z = w.new_zeros(w.shape[0], 3 - cdp)
                w = torch.cat([w[:, :cdp], z, w[:, cdp:2 * cdp], z, w[:, 2 * cdp:]], dim=1)
            k = GF.linear(k, w, fc.bias, act=act if i < len(fcs) - 1 else None)
            k = GF.dropout(k, pdrop, self.training)
        if f_y is not None and tt != "nonlinear_kernelonly":                                                # :156-157
            k = EO.MulFn.apply(k, EO.GatherFn.apply(f_y, g, 0))
        mode = EO.MEAN
        if self.use_attn:                                                                                   # :126-142
            # the scores see the first coord_dim coordinates only (integral_transform.py:128-129: `[:, :self.coord_dim]`)
            cd = int(self.coord_dim) if self.coord_dim is not None else cdp
            if not 1 <= cd <= min(3, cdp):
                raise ValueError(f"coord_dim {cd} with {cdp}-dimensional coordinates")
            xa, ya = x3, y3
            if cd < cdp:             # drop the coordinates the scores do not see, keep the kernels' 3-D rows
                xa = torch.nn.functional.pad(x_pos[:, :cd], (0, 3 - cd))
                ya = torch.nn.functional.pad(y_pos[:, :cd], (0, 3 - cd))
            if self.attention_type == "dot_product":
                wq, wk = self.query_proj.weight, self.key_proj.weight
                if cd < 3:           # Linear(coord_dim -> 64) on zero-padded coordinates: zero columns, exact
                    wq = torch.nn.functional.pad(wq, (0, 3 - cd))
                    wk = torch.nn.functional.pad(wk, (0, 3 - cd))
                qn = GF.linear(xa.contiguous(), wq, self.query_proj.bias, precision=0)
                kn = GF.linear(ya.contiguous(), wk, self.key_proj.bias, precision=0)
                sc = EO.RowDotFn.apply(EO.GatherFn.apply(qn, g, 1), EO.GatherFn.apply(kn, g, 0), self.scaling_factor)
            else:
                sc = EO.edge_coords(ya.contiguous(), xa.contiguous(), g, 2)   # cosine of the raw coordinates: no parameters
            k = EO.RowScaleFn.apply(k, EO.SegmentSoftmaxFn.apply(sc, g))                                    # :159-160
            mode = EO.SUM                                                                                   # :163
        return EO.SegmentReduceFn.apply(k, g, mode)
