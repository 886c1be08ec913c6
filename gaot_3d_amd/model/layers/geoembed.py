"""GeometricEmbedding with the reference's signature (src/model/layers/geoembed.py:24, 57-64).
method='statistical' runs csrc/geoembed.hip (one neighbour-list sweep + in-register 3x3 eigen-solve)
followed by the 9 -> 64 -> C MLP on the HIP GEMM.  'pointnet' raises NotImplementedError."""
from typing import Optional

import torch
import torch.nn as nn

from ... import functional as GF
from ... import ops
from .integral_transform import graph_for


class GeometricEmbedding(nn.Module):
    def __init__(self, input_dim, output_dim, method="statistical", pooling="max", **kwargs):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.method = method.lower()
        self.pooling = pooling.lower()
        self.kwargs = kwargs
        if self.pooling not in ["max", "mean"]:
            raise ValueError(f"Unsupported pooling method: {self.pooling}. Supported methods: 'max', 'mean'.")
        if self.method == "statistical":
            self.mlp = nn.Sequential(nn.Linear(3 + 2 * input_dim, 64), nn.ReLU(), nn.Linear(64, output_dim))
        elif self.method == "pointnet":
            self.pointnet_mlp = nn.Sequential(nn.Linear(input_dim, 32), nn.ReLU(), nn.Linear(32, 32), nn.ReLU())
            self.fc = nn.Sequential(nn.Linear(32, output_dim))
        else:
            raise ValueError(f"Unknown method: {self.method}")

    def forward(self, source_pos, query_pos, edge_index, batch_source: Optional[torch.Tensor] = None,
                batch_query: Optional[torch.Tensor] = None, neighbors_counts: Optional[torch.Tensor] = None,
                graph=None, shard_group=None):
        """``shard_group`` (extension, gaot_3d_amd/sharding.py): the edges of this sample are spread over the ranks of
        the group; the per-row statistics are assembled from additive fp64 moments with one SUM all-reduce."""
        if self.method != "statistical":
            raise NotImplementedError("GeometricEmbedding(method='pointnet') is not implemented on the HIP path")
        if graph is None:
            graph = graph_for(edge_index.to(query_pos.device), source_pos.shape[0], query_pos.shape[0])
        if shard_group is not None:
            import torch.distributed as dist
            mom = ops.geoembed_moments(source_pos, query_pos, graph)
            dist.all_reduce(mom, op=dist.ReduceOp.SUM, group=shard_group)
            feats = ops.geoembed_from_moments(mom)
        else:
            feats = ops.geoembed_stats(source_pos, query_pos, graph)  # geometry only: no autograd through it
        h = GF.linear(feats, self.mlp[0].weight, self.mlp[0].bias, act="relu", precision=0)
        return GF.linear(h, self.mlp[2].weight, self.mlp[2].bias, precision=0)
