"""MAGNO encoder / decoder with the reference's class names, constructor signatures, config dataclass
and state_dict layout (src/model/layers/magno.py: MAGNOConfig :21-66, MAGNOEncoder :377-600,
MAGNODecoder :605-798), computing through the HIP kernels.

Edges are INPUTS of the hot path (precompute_edges=True, magno.py:506-511 / 715-720).  Building them
(get_neighbor_strategy, :116-295) is the next row of SURVEY §8f; ``get_neighbor_strategy`` here covers
the strategies through plain torch helpers so the data layer keeps its import, and is not the measured
path."""
from dataclasses import dataclass, field
from typing import Any, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ... import functional as GF
from ...data import coalesce_edges, knn_edges_bruteforce, radius_edges_bruteforce
from .geoembed import GeometricEmbedding
from ...graph import apply_neighbor_sampling  # noqa: F401  (the reference exports it from this module, magno.py:297-371)
from .integral_transform import IntegralTransform, graph_for
from .mlp import ChannelMLP, LinearChannelMLP


@dataclass
class MAGNOConfig:
    # GNO parameters
    use_gno: bool = True
    gno_coord_dim: int = 2
    gno_radius: float = 0.033
    # MAGNOEncoder
    lifting_channels: int = 16
    encoder_feature_attr: Any = "x"
    in_gno_channel_mlp_hidden_layers: list = field(default_factory=lambda: [64, 64, 64])
    in_gno_transform_type: str = "linear"
    # MAGNODecoder
    projection_channels: int = 256
    out_gno_channel_mlp_hidden_layers: list = field(default_factory=lambda: [64, 64])
    out_gno_transform_type: str = "linear"
    mlp_type: str = "channel"
    # multiscale aggregation
    scales: list = field(default_factory=lambda: [1.0])
    use_scale_weights: bool = False
    use_graph_cache: bool = True
    gno_use_torch_cluster: bool = False
    gno_use_torch_scatter: str = True
    node_embedding: bool = False
    use_attn: Optional[bool] = None
    attention_type: str = "cosine"
    # Geometric embedding
    use_geoembed: Any = field(default_factory=lambda: [True, True])
    embedding_method: str = "statistical"
    pooling: str = "max"
    # Sampling
    sampling_strategy: Optional[str] = None
    max_neighbors: Optional[int] = None
    sample_ratio: Optional[float] = None
    # neighbor finding strategy
    neighbor_strategy: Any = "radius"
    k_neighbors: int = 1
    # Dataset
    precompute_edges: bool = True
    asynchronous_graph_building: bool = False


def parse_neighbor_strategy(neighbor_strategy: Union[str, List[str]]) -> Tuple[str, str]:
    if isinstance(neighbor_strategy, str):
        return neighbor_strategy, neighbor_strategy
    if isinstance(neighbor_strategy, (list, tuple)) and len(neighbor_strategy) == 2:
        return neighbor_strategy[0], neighbor_strategy[1]
    raise ValueError(f"neighbor_strategy must be str or list of length 2, got {neighbor_strategy}")


def parse_geoembed_strategy(use_geoembed: Union[bool, List[bool]]) -> Tuple[bool, bool]:
    if isinstance(use_geoembed, bool):
        return use_geoembed, use_geoembed
    if isinstance(use_geoembed, (list, tuple)) and len(use_geoembed) == 2:
        return use_geoembed[0], use_geoembed[1]
    raise ValueError(f"use_geoembed must be bool or list of length 2, got {use_geoembed}")


def _per_graph(fn, phys_pos, batch_idx_phys, latent_pos, batch_idx_latent):
    """apply a single-graph edge builder per batch element and offset the indices"""
    nb = int(batch_idx_phys.max().item()) + 1 if batch_idx_phys.numel() else 1
    out = []
    for b in range(nb):
        pm = (batch_idx_phys == b).nonzero(as_tuple=True)[0]
        lm = (batch_idx_latent == b).nonzero(as_tuple=True)[0]
        e = fn(phys_pos[pm], latent_pos[lm])  # rows index into the per-graph subsets
        out.append((e, pm, lm))
    return out


def get_neighbor_strategy(neighbor_strategy: str, phys_pos, batch_idx_phys, latent_tokens_pos, batch_idx_latent,
                          radius: float, k_neighbors: int = 1, is_decoder: bool = False, latent_dims=None):
    """Edge construction with the reference's conventions (magno.py:116-295): encoder edges are
    [phys_idx, latent_idx], decoder edges [latent_idx, phys_idx]; 'bidirectional' = coalesce(knn U radius);
    'reverse' (decoder only) = flip of the *bidirectional* encoder graph; PyG radius keeps at most 32
    neighbours per centre.  With the points on the HIP device the graph is built by the device kernels
    (gaot_3d_amd/graph.py, csrc/graph.hip): closed-form cell lookups when the tokens are the model's regular
    D x H x W grid (``latent_dims``), brute-force scans of all tokens for any other token set; for CPU tensors by
    the brute-force torch restatement below (host-side data preparation, also the checker of the device path in
    tests/test_graph_gpu.py)."""
    if phys_pos.is_cuda:   # device kernels: cell lookups on the regular token grid, brute-force scans for any other token set
        from ... import graph as device_graph
        return device_graph.get_neighbor_strategy(neighbor_strategy, phys_pos, batch_idx_phys, latent_tokens_pos,
                                                  batch_idx_latent, radius, k_neighbors, is_decoder, latent_dims=latent_dims)

    def enc(strategy, p, l):
        knn = rad = None
        if strategy in ("knn", "bidirectional"):
            knn = knn_edges_bruteforce(p, l, k_neighbors)
        if strategy in ("radius", "bidirectional"):
            rad = radius_edges_bruteforce(p, l, radius, 32, centers="latent")
        if strategy == "knn":
            return knn
        if strategy == "radius":
            return rad
        if strategy == "bidirectional":
            return coalesce_edges(torch.cat([knn, rad], dim=1), l.shape[0])
        raise ValueError(f"Unknown encoder strategy: {strategy}")

    def dec(strategy, p, l):
        if strategy == "reverse":
            return enc("bidirectional", p, l).flip(0)
        knn = rad = None
        if strategy in ("knn", "bidirectional"):
            knn = knn_edges_bruteforce(p, l, k_neighbors).flip(0)
        if strategy in ("radius", "bidirectional"):
            rad = radius_edges_bruteforce(p, l, radius, 32, centers="phys")
        if strategy == "knn":
            return knn
        if strategy == "radius":
            return rad
        if strategy == "bidirectional":
            return coalesce_edges(torch.cat([knn, rad], dim=1), p.shape[0])
        raise ValueError(f"Unknown decoder strategy: {strategy}")

    parts = _per_graph((lambda p, l: dec(neighbor_strategy, p, l)) if is_decoder else
                       (lambda p, l: enc(neighbor_strategy, p, l)),
                       phys_pos, batch_idx_phys, latent_tokens_pos, batch_idx_latent)
    outs = []
    for e, pm, lm in parts:
        if is_decoder:
            outs.append(torch.stack([lm[e[0]], pm[e[1]]]))
        else:
            outs.append(torch.stack([pm[e[0]], lm[e[1]]]))
    return torch.cat(outs, dim=1) if outs else torch.empty((2, 0), dtype=torch.long, device=phys_pos.device)


def _make_mlp(mlp_type, layers):
    if mlp_type == "linear":
        return LinearChannelMLP(layers=layers)
    if len(layers) == 2:
        return ChannelMLP(in_channels=layers[0], out_channels=layers[1], n_layers=1)
    return ChannelMLP(in_channels=layers[0], out_channels=layers[-1], hidden_channels=layers[1], n_layers=len(layers) - 1,
                      n_dim=1)


def _check_unsupported(cfg: MAGNOConfig):
    if cfg.sampling_strategy not in (None, "max_neighbors", "ratio"):
        raise ValueError(f"Invalid sampling strategy: {cfg.sampling_strategy}")


def _sample(module, edge_index, num_query):
    """reference magno.py:529-538 / 739-748"""
    if module.sampling_strategy is None:
        return edge_index
    from ...graph import apply_neighbor_sampling
    return apply_neighbor_sampling(edge_index, num_query, edge_index.device, module.sampling_strategy, module.max_neighbors,
                                   module.sample_ratio, module.training)


def _sum_scales(outs):
    acc = outs[0]
    for o in outs[1:]:
        acc = GF.add(acc, o)
    return acc


def _make_scale_weighting(coord_dim, num_scales):
    return nn.Sequential(nn.Linear(coord_dim, 16), nn.ReLU(), nn.Linear(16, num_scales))


def _mix_scales(module, outs, pos):
    """sum over scales, or softmax-weighted by an MLP of the query coordinates (reference magno.py:586-596)"""
    if len(outs) == 1:
        return outs[0]
    if not module.use_scale_weights:
        return _sum_scales(outs)
    sw = module.scale_weighting
    h = GF.linear(pos, sw[0].weight, sw[0].bias, act="relu", precision=0)
    logits = GF.linear(h, sw[2].weight, sw[2].bias, precision=0)
    return GF.ScaleMixFn.apply(logits, *outs)


class MAGNOEncoder(nn.Module):
    def __init__(self, in_channels, out_channels, gno_config: MAGNOConfig):
        super().__init__()
        _check_unsupported(gno_config)
        self.sampling_strategy = gno_config.sampling_strategy
        self.max_neighbors = gno_config.max_neighbors
        self.sample_ratio = gno_config.sample_ratio
        self.gno_radius = gno_config.gno_radius
        self.scales = gno_config.scales
        self.lifting_channels = gno_config.lifting_channels
        self.coord_dim = gno_config.gno_coord_dim
        self.feature_attr_name = gno_config.encoder_feature_attr
        self.precompute_edges = gno_config.precompute_edges
        self.mlp_type = gno_config.mlp_type
        self.encoder_strategy, self.decoder_strategy = parse_neighbor_strategy(gno_config.neighbor_strategy)
        self.k_neighbors = gno_config.k_neighbors
        self.use_gno = gno_config.use_gno
        if self.use_gno:
            kin = self.coord_dim * 2
            if gno_config.in_gno_transform_type in ("nonlinear", "nonlinear_kernelonly"):
                kin += in_channels
            layers = [kin] + list(gno_config.in_gno_channel_mlp_hidden_layers) + [self.lifting_channels]
            self.gno = IntegralTransform(channel_mlp_layers=layers, transform_type=gno_config.in_gno_transform_type,
                                         use_attn=gno_config.use_attn, coord_dim=self.coord_dim,
                                         attention_type=gno_config.attention_type)
            self.lifting = _make_mlp(gno_config.mlp_type, [in_channels, self.lifting_channels])
        else:
            self.gno = None
            self.lifting = None
        self.use_geoembed = parse_geoembed_strategy(gno_config.use_geoembed)[0]
        if self.use_geoembed:
            self.geoembed = GeometricEmbedding(input_dim=self.coord_dim, output_dim=self.lifting_channels,
                                               method=gno_config.embedding_method, pooling=gno_config.pooling)
            self.recovery = _make_mlp(gno_config.mlp_type, [2 * self.lifting_channels, self.lifting_channels])
        self.use_scale_weights = gno_config.use_scale_weights
        if self.use_scale_weights:
            self.num_scales = len(self.scales)
            self.scale_weighting = _make_scale_weighting(self.coord_dim, self.num_scales)
            self.scale_weight_activation = nn.Softmax(dim=-1)

    def _features(self, batch):
        names = self.feature_attr_name if isinstance(self.feature_attr_name, (list, tuple)) else [self.feature_attr_name]
        feats = []
        for a in names:
            f = getattr(batch, a, None)
            if f is None:
                if self.use_gno:
                    raise AttributeError(f"MAGNOEncoder requires feature attribute '{a}' but it was not found in the batch.")
            else:
                feats.append(f)
        return feats

    def forward(self, batch, latent_tokens_pos: torch.Tensor, latent_tokens_batch_idx: torch.Tensor) -> torch.Tensor:
        phys_pos = batch.pos
        device = phys_pos.device
        num_graphs = batch.num_graphs
        m_per_graph = latent_tokens_pos.shape[0] // num_graphs
        phys_feat = self._features(batch)
        lifted = None
        if self.use_gno:  # scale-independent: lift once; [f0|f1|..] W^T is evaluated without materialising the cat
            if len(self.lifting.fcs) == 1 and not (self.training and getattr(self.lifting, "dropout_p", 0.0) > 0.0):
                lifted = GF.cat_linear(phys_feat, self.lifting.fcs[0].weight, self.lifting.fcs[0].bias, precision=0)
            else:   # a caller-supplied deeper lifting MLP (the reference's constructor always builds one layer)
                lifted = self.lifting.forward_rows(phys_feat[0] if len(phys_feat) == 1 else torch.cat(phys_feat, dim=1))
        outs = []
        for si, scale in enumerate(self.scales):
            if self.precompute_edges:
                attr = f"encoder_edge_index_s{si}"
                if not hasattr(batch, attr):
                    raise AttributeError(f"Batch object missing pre-computed '{attr}'")
                edge_index = getattr(batch, attr).to(device)
            else:
                edge_index = get_neighbor_strategy(self.encoder_strategy, phys_pos, batch.batch, latent_tokens_pos,
                                                   latent_tokens_batch_idx, self.gno_radius * scale, self.k_neighbors,
                                                   False, latent_dims=getattr(self, "latent_dims", None)).to(device)
            edge_index = _sample(self, edge_index, latent_tokens_pos.shape[0])
            g = graph_for(edge_index, phys_pos.shape[0], latent_tokens_pos.shape[0],
                          batch if self.sampling_strategy is None else None, ("enc", si))
            enc = self.gno(y_pos=phys_pos, x_pos=latent_tokens_pos, edge_index=edge_index, f_y=lifted,
                           graph=g) if self.use_gno else None
            shard_group = getattr(self, "_shard_group", None)
            if shard_group is not None and enc is not None:
                # point-sharded sample (gaot_3d_amd/sharding.py): the local mean becomes the mean over every rank's edges
                from ...sharding import GlobalSegmentMeanFn
                deg = (g.by_dst.rowptr[1:] - g.by_dst.rowptr[:-1]).to(enc.dtype)
                enc = GlobalSegmentMeanFn.apply(enc, deg, shard_group)
            # GeoEmbed statistics of a sharded sample: additive moments of the local edges, summed over the ranks
            geo = self.geoembed(phys_pos, latent_tokens_pos, edge_index, graph=g, shard_group=shard_group) \
                if self.use_geoembed else None
            if enc is not None and geo is not None:
                enc = GF.cat_linear([enc, geo], self.recovery.fcs[0].weight, self.recovery.fcs[0].bias, precision=0)
            elif enc is None and geo is not None:
                enc = geo
            elif enc is None:
                raise ValueError("GNO and GeoEmbed are both disabled. No encoding will be performed.")
            outs.append(enc)
        out = _mix_scales(self, outs, latent_tokens_pos)
        return out.view(num_graphs, m_per_graph, self.lifting_channels)


class MAGNODecoder(nn.Module):
    def __init__(self, in_channels, out_channels, gno_config: MAGNOConfig):
        super().__init__()
        _check_unsupported(gno_config)
        self.sampling_strategy = gno_config.sampling_strategy
        self.max_neighbors = gno_config.max_neighbors
        self.sample_ratio = gno_config.sample_ratio
        self.gno_radius = gno_config.gno_radius
        self.scales = gno_config.scales
        self.coord_dim = gno_config.gno_coord_dim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.use_geoembed = parse_geoembed_strategy(gno_config.use_geoembed)[1]
        self.use_scale_weights = gno_config.use_scale_weights
        self.precompute_edges = gno_config.precompute_edges
        self.mlp_type = gno_config.mlp_type
        self.encoder_strategy, self.decoder_strategy = parse_neighbor_strategy(gno_config.neighbor_strategy)
        self.k_neighbors = gno_config.k_neighbors
        kin = self.coord_dim * 2
        if gno_config.out_gno_transform_type in ("nonlinear", "nonlinear_kernelonly"):
            kin += in_channels
        layers = [kin] + list(gno_config.out_gno_channel_mlp_hidden_layers) + [in_channels]
        self.gno = IntegralTransform(channel_mlp_layers=layers, transform_type=gno_config.out_gno_transform_type,
                                     use_attn=gno_config.use_attn, coord_dim=self.coord_dim,
                                     attention_type=gno_config.attention_type)
        self.projection = _make_mlp(gno_config.mlp_type, [in_channels, gno_config.projection_channels, out_channels])
        if self.use_geoembed:
            self.geoembed = GeometricEmbedding(input_dim=self.coord_dim, output_dim=in_channels,
                                               method=gno_config.embedding_method, pooling=gno_config.pooling)
            self.recovery = _make_mlp(gno_config.mlp_type, [2 * in_channels, in_channels])
        if self.use_scale_weights:
            self.num_scales = len(self.scales)
            self.scale_weighting = _make_scale_weighting(self.coord_dim, self.num_scales)
            self.scale_weight_activation = nn.Softmax(dim=-1)

    def forward(self, rndata_flat, phys_pos_query, batch_idx_phys_query, latent_tokens_pos, latent_tokens_batch_idx,
                batch=None) -> torch.Tensor:
        device = rndata_flat.device
        outs = []
        for si, scale in enumerate(self.scales):
            if self.precompute_edges:
                attr = f"decoder_edge_index_s{si}"
                if not hasattr(batch, attr):
                    raise AttributeError(f"Batch object missing pre-computed '{attr}'")
                edge_index = getattr(batch, attr).to(device)
            else:
                edge_index = get_neighbor_strategy(self.decoder_strategy, phys_pos_query, batch_idx_phys_query,
                                                   latent_tokens_pos, latent_tokens_batch_idx, self.gno_radius * scale,
                                                   self.k_neighbors, True,
                                                   latent_dims=getattr(self, "latent_dims", None)).to(device)
            edge_index = _sample(self, edge_index, phys_pos_query.shape[0])
            g = graph_for(edge_index, latent_tokens_pos.shape[0], phys_pos_query.shape[0],
                          batch if self.sampling_strategy is None else None, ("dec", si))
            dec = self.gno(y_pos=latent_tokens_pos, x_pos=phys_pos_query, edge_index=edge_index, f_y=rndata_flat, graph=g)
            if self.use_geoembed:
                sg = getattr(self, "_shard_group", None)
                if sg is not None:   # point-sharded sample: the z-score (geoembed.py:177-180) runs over ALL query points
                    total = getattr(batch, "shard", (0, 0, 0, 0, phys_pos_query.shape[0]))[4]
                    geo = self.geoembed(latent_tokens_pos, phys_pos_query, edge_index, graph=g, shard_group=sg,
                                        sharded_queries_total=total)
                else:
                    geo = self.geoembed(latent_tokens_pos, phys_pos_query, edge_index, graph=g)
                dec = GF.cat_linear([dec, geo], self.recovery.fcs[0].weight, self.recovery.fcs[0].bias, precision=0)
            outs.append(dec)
        return self.projection.forward_rows(_mix_scales(self, outs, phys_pos_query))
