"""GeometricEmbedding with the reference's signature (src/model/layers/geoembed.py:24, 57-64).
method='statistical' runs csrc/geoembed.hip (one neighbour-list sweep + in-register 3x3 eigen-solve)
followed by the 9 -> 64 -> C MLP on the HIP GEMM.  method='pointnet' (geoembed.py:184-222) runs the general per-edge
path: edge offsets, the 3 -> 32 -> 32 ReLU MLP on the GEMM kernels, segment max / mean and the output linear
(csrc/edgeops.hip)."""
from typing import Optional

import torch
import torch.nn as nn

from ... import edgeops as EO
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
                graph=None, shard_group=None, sharded_queries_total: Optional[int] = None):
        """``shard_group`` (extension, gaot_3d_amd/sharding.py): the edges of this sample are spread over the ranks of
        the group; the per-row statistics are assembled from additive fp64 moments with one SUM all-reduce."""
        if graph is None:
            graph = graph_for(edge_index.to(query_pos.device), source_pos.shape[0], query_pos.shape[0])
        if self.method == "pointnet":
            if shard_group is not None and sharded_queries_total is None:
                # encoder side of a point-sharded sample: a token's edges are spread over the ranks
                return self._forward_pointnet_sharded(source_pos, query_pos, graph, shard_group)
            return self._forward_pointnet(source_pos, query_pos, graph)   # decoder side: every query's edges are local
        if shard_group is not None and sharded_queries_total is not None:
            # decoder side of a point-sharded sample: the query rows are spread over the ranks, only the z-score is global
            feats = ops.geoembed_stats_sharded_queries(source_pos, query_pos, graph, shard_group, sharded_queries_total)
        elif shard_group is not None:
            import torch.distributed as dist
            mom = ops.geoembed_moments(source_pos, query_pos, graph)
            from ... import comm
            comm.run(lambda: dist.all_reduce(mom, op=dist.ReduceOp.SUM, group=shard_group), (mom,), "all_reduce")
            feats = ops.geoembed_from_moments(mom)
        else:
            # geometry only: no autograd through it.  One sweep over the neighbour lists (additive fp64 moments about the
            # query position, then centroid / covariance / eigenvalues per row): 0.22 ms at configs[1] against 0.37 ms
            # for the two-sweep kernel (gaot_geoembed_raw: centroid first, then centred second moments); same features.
            # A per-sample constant: kept on the neighbour-list object (which the batch caches per edge tensor) for as long
            # as the very same coordinate tensors are passed again (identity + in-place version, references held)
            ent = graph.__dict__.get("_geo_feats")
            if (ent is not None and ent[0] is source_pos and ent[1] == source_pos._version and ent[2] is query_pos
                    and ent[3] == query_pos._version):
                feats = ent[4]
            else:
                feats = ops.geoembed_from_moments(ops.geoembed_moments(source_pos, query_pos, graph))
                graph.__dict__["_geo_feats"] = (source_pos, source_pos._version, query_pos, query_pos._version, feats)
        h = GF.linear(feats, self.mlp[0].weight, self.mlp[0].bias, act="relu", precision=0)
        return GF.linear(h, self.mlp[2].weight, self.mlp[2].bias, precision=0)

    def _forward_pointnet_sharded(self, source_pos, query_pos, g, group):
        """the same when the edges of a query row are spread over the ranks of ``group`` (sharding.py): per-edge MLP and the
        segment pooling on the local edges, then max -> all-reduce(MAX) with the gradient routed to the owning rank(s), mean ->
        all-reduce of sums and counts; fc and the empty-row mask on the global result (identical on every rank)"""
        from ...sharding import GlobalSegmentMaxFn, GlobalSegmentMeanFn
        nq = query_pos.shape[0]
        rp = g.by_dst.rowptr
        deg = (rp[1:] - rp[:-1]).to(torch.float32)
        if g.by_dst.num_edges == 0:            # this rank holds no edge: it still takes part in the exchanges
            local = torch.zeros(nq, 32, dtype=query_pos.dtype, device=query_pos.device)
        else:
            c = EO.edge_coords(source_pos, query_pos, g, 1)
            h = GF.linear(c, self.pointnet_mlp[0].weight, self.pointnet_mlp[0].bias, act="relu", precision=0)
            h = GF.linear(h, self.pointnet_mlp[2].weight, self.pointnet_mlp[2].bias, act="relu", precision=0)
            local = EO.SegmentReduceFn.apply(h, g, EO.MAX if self.pooling == "max" else EO.MEAN)
        if self.pooling == "max":
            pooled, gdeg = GlobalSegmentMaxFn.apply(local, deg, group)
        else:
            pooled = GlobalSegmentMeanFn.apply(local, deg, group)
            gdeg = deg.clone()
            import torch.distributed as dist
            from ... import comm
            comm.run(lambda: dist.all_reduce(gdeg, op=dist.ReduceOp.SUM, group=group), (gdeg,), "all_reduce")
        po = GF.linear(pooled, self.fc[0].weight, self.fc[0].bias, precision=0)
        return EO.RowScaleFn.apply(po, (gdeg > 0).to(torch.float32))

    def _forward_pointnet(self, source_pos, query_pos, g):
        """geoembed.py:184-222: MLP(nbr - query) per edge, segment max | mean, fc; rows without neighbours = 0"""
        nq = query_pos.shape[0]
        if g.by_dst.num_edges == 0:
            return torch.zeros(nq, self.output_dim, dtype=query_pos.dtype, device=query_pos.device)
        c = EO.edge_coords(source_pos, query_pos, g, 1)                                    # :196-198
        h = GF.linear(c, self.pointnet_mlp[0].weight, self.pointnet_mlp[0].bias, act="relu", precision=0)
        h = GF.linear(h, self.pointnet_mlp[2].weight, self.pointnet_mlp[2].bias, act="relu", precision=0)
        pooled = EO.SegmentReduceFn.apply(h, g, EO.MAX if self.pooling == "max" else EO.MEAN)   # :211-216
        po = GF.linear(pooled, self.fc[0].weight, self.fc[0].bias, precision=0)
        rp = g.by_dst.rowptr
        has = (rp[1:] > rp[:-1]).to(torch.float32)                                          # :219 masked assignment
        return EO.RowScaleFn.apply(po, has)
