"""IntegralTransform with the reference's constructor/forward signature
(src/model/layers/integral_transform.py:31-40, 80-87) running the fused HIP kernel (csrc/gno.hip).

Supported on the HIP path: transform_type='linear', use_attn falsy, coord_dim 3, kernel MLP
6 -> 64 (x1..4) -> 32.  Everything else raises NotImplementedError (no silent fallback)."""
from typing import Optional

import torch
import torch.nn as nn

from ... import functional as GF
from ... import ops
from .mlp import LinearChannelMLP


def graph_for(edge_index: torch.Tensor, num_src: int, num_dst: int, cache_owner=None, key=None):
    """Row-sorted neighbour lists for one edge_index; cached on the batch object so that encoder GNO,
    GeoEmbed and the backward pass share one build per sample and scale."""
    if cache_owner is not None:
        cache = cache_owner.__dict__.setdefault("_gaot_graphs", {})
        k = (key, edge_index.data_ptr(), tuple(edge_index.shape), num_src, num_dst)
        g = cache.get(k)
        if g is None:
            g = ops.build_graph(edge_index, num_src, num_dst)
            cache[k] = g
        return g
    return ops.build_graph(edge_index, num_src, num_dst)


class IntegralTransform(nn.Module):
    def __init__(self, channel_mlp=None, channel_mlp_layers=None, channel_mlp_non_linearity="gelu",
                 transform_type="linear", use_attn=None, coord_dim=None, attention_type="cosine"):
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
        if self.use_attn:
            raise NotImplementedError("IntegralTransform(use_attn=True) is not implemented on the HIP path")
        if self.transform_type != "linear":
            raise NotImplementedError(f"transform_type='{self.transform_type}' is not implemented on the HIP path "
                                      f"(only 'linear')")
        if f_y is None:
            raise NotImplementedError("IntegralTransform without f_y is not implemented on the HIP path")
        if graph is None:
            graph = graph_for(edge_index.to(x_pos.device), y_pos.shape[0], x_pos.shape[0])
        params = []
        for fc in self.channel_mlp.fcs:
            params += [fc.weight, fc.bias]
        return GF.GnoFn.apply(f_y, y_pos, x_pos, graph, *params)
