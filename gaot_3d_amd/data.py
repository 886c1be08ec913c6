"""
Host-side sample container and synthetic-input helpers for the GAOT-3D hot path.

``MeshBatch`` is a PyG-free, duck-typed stand-in for the ``torch_geometric.data.Batch`` the
reference trainer hands to ``model(batch, tokens_pos)`` (attributes consumed:
reference src/model/layers/magno.py:480-516,715-725; src/trainer/stat.py:543-550).  Batching
follows ``EnrichedData.__inc__`` (reference src/data/pyg_datasets.py:12-31): encoder edges
are offset by (num_nodes, num_latent_nodes), decoder edges by (num_latent_nodes, num_nodes).

The graph helpers below only *produce inputs* (edges are inputs to the hot path);
they are plain torch and run on either device.  They are not the measured path.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor


class MeshBatch:
    """Attribute bag with ``.to(device)``; mirrors the Batch attributes the model reads."""

    def __init__(self, **kw):
        self.num_graphs = 1
        for k, v in kw.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def to(self, device, non_blocking: bool = False) -> "MeshBatch":
        out = MeshBatch()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v)
        return out

    @staticmethod
    def from_data_list(samples: Sequence["MeshBatch"], num_latent_nodes: int) -> "MeshBatch":
        """Concatenate single-graph samples with the reference's increment rules."""
        out = MeshBatch()
        node_off, lat_off = 0, 0
        cat: dict = {}
        bidx: List[Tensor] = []
        ptr = [0]
        for gi, s in enumerate(samples):
            n = s.pos.shape[0]
            for k, v in s.__dict__.items():
                if not torch.is_tensor(v):
                    continue
                if k.startswith("encoder_edge_index"):
                    inc = torch.tensor([[node_off], [lat_off]], dtype=v.dtype)
                    v = v + inc
                    cat.setdefault(k, []).append((v, 1))
                elif k.startswith("decoder_edge_index"):
                    inc = torch.tensor([[lat_off], [node_off]], dtype=v.dtype)
                    v = v + inc
                    cat.setdefault(k, []).append((v, 1))
                elif k in ("batch", "ptr"):
                    continue
                else:
                    cat.setdefault(k, []).append((v, 0))
            bidx.append(torch.full((n,), gi, dtype=torch.long))
            node_off += n
            lat_off += num_latent_nodes
            ptr.append(node_off)
        for k, lst in cat.items():
            setattr(out, k, torch.cat([t for t, _ in lst], dim=lst[0][1]))
        out.batch = torch.cat(bidx)
        out.ptr = torch.tensor(ptr, dtype=torch.long)
        out.num_graphs = len(samples)
        return out


def latent_grid(latent_tokens: Sequence[int], lo=(-1.0, -1.0, -1.0), hi=(1.0, 1.0, 1.0)) -> Tensor:
    """Regular D x H x W grid, ``indexing='ij'`` row-major (d slowest), as the reference builds
    it (src/model/gaot_3d.py:35-46; trainer copy src/trainer/stat.py:238-252)."""
    d, h, w = latent_tokens
    mg = torch.meshgrid(torch.linspace(lo[0], hi[0], d), torch.linspace(lo[1], hi[1], h),
                        torch.linspace(lo[2], hi[2], w), indexing="ij")
    return torch.stack(mg, dim=-1).reshape(-1, 3)


def knn_edges_bruteforce(phys_pos: Tensor, latent_pos: Tensor, k: int) -> Tensor:
    """[2, N*k] rows [phys_idx, latent_idx], phys-major -- the layout ``pyg_knn(x=latent,
    y=phys, k)`` returns (reference magno.py:183-189)."""
    d = torch.cdist(phys_pos, latent_pos)
    nn_idx = d.topk(k, dim=1, largest=False).indices  # [N,k]
    n = phys_pos.shape[0]
    src = torch.arange(n, device=phys_pos.device).repeat_interleave(k)
    return torch.stack([src, nn_idx.reshape(-1)], dim=0)


def knn_edges_grid(phys_pos: Tensor, latent_tokens: Sequence[int], lo, hi, k: int, halo: int = 2,
                   chunk: int = 131072) -> Tensor:
    """Same result as ``knn_edges_bruteforce`` against ``latent_grid(latent_tokens, lo, hi)`` but
    O(N * (2*halo+1)^3): the latent tokens are a regular grid, so the k nearest tokens of a point
    lie in a small block of cells around its nearest node."""
    dev = phys_pos.device
    dims = torch.tensor(list(latent_tokens), device=dev)
    lo_t = torch.tensor(lo, dtype=phys_pos.dtype, device=dev)
    hi_t = torch.tensor(hi, dtype=phys_pos.dtype, device=dev)
    step = (hi_t - lo_t) / (dims - 1).clamp(min=1).to(phys_pos.dtype)
    r = torch.arange(-halo, halo + 1, device=dev)
    off = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), dim=-1).reshape(-1, 3)  # [R,3]
    out = []
    for s in range(0, phys_pos.shape[0], chunk):
        p = phys_pos[s:s + chunk]
        base = torch.round((p - lo_t) / step).long()
        base = torch.minimum(torch.maximum(base, torch.zeros_like(base) + halo), dims - 1 - halo)
        cell = base[:, None, :] + off[None, :, :]                      # [n,R,3]
        cell = torch.minimum(torch.maximum(cell, torch.zeros_like(cell)), (dims - 1).expand_as(cell))
        cpos = lo_t + cell.to(phys_pos.dtype) * step
        d2 = ((cpos - p[:, None, :]) ** 2).sum(-1)
        lin = (cell[..., 0] * dims[1] + cell[..., 1]) * dims[2] + cell[..., 2]
        # duplicates can appear after clamping at the boundary: push them away
        srt, order = lin.sort(dim=1)
        dup = torch.zeros_like(srt, dtype=torch.bool)
        dup[:, 1:] = srt[:, 1:] == srt[:, :-1]
        d2s = d2.gather(1, order).masked_fill(dup, float("inf"))
        sel = d2s.topk(k, dim=1, largest=False).indices
        out.append(srt.gather(1, sel))
    nn_idx = torch.cat(out, dim=0)
    n = phys_pos.shape[0]
    src = torch.arange(n, device=dev).repeat_interleave(k)
    return torch.stack([src, nn_idx.reshape(-1)], dim=0)


def radius_edges_bruteforce(phys_pos: Tensor, latent_pos: Tensor, radius: float,
                            max_num_neighbors: Optional[int] = 32, centers: str = "latent") -> Tensor:
    """Encoder-style radius graph: latent tokens are centres, every phys point within ``radius``
    connects; rows [phys_idx, latent_idx] sorted by latent (reference magno.py:193-201, PyG
    ``radius`` default cap 32 neighbours per centre).  ``centers='phys'`` builds the decoder
    variant (magno.py:253-261) and returns rows [latent_idx, phys_idx] sorted by phys."""
    if centers == "latent":
        c, o = latent_pos, phys_pos
    else:
        c, o = phys_pos, latent_pos
    d = torch.cdist(c, o)
    mask = d <= radius
    ci, oi = mask.nonzero(as_tuple=True)  # sorted by centre, then other index
    if max_num_neighbors is not None and ci.numel() > 0:
        cnt = torch.bincount(ci, minlength=c.shape[0])
        start = torch.cumsum(cnt, 0) - cnt
        rank = torch.arange(ci.numel(), device=ci.device) - start[ci]
        keep = rank < max_num_neighbors
        ci, oi = ci[keep], oi[keep]
    return torch.stack([oi, ci], dim=0)


def coalesce_edges(edge_index: Tensor, num_cols: int) -> Tensor:
    """Sort by (row0,row1) and drop duplicates -- ``torch_geometric.utils.coalesce`` semantics
    used for the 'bidirectional' strategy (reference magno.py:219-220)."""
    key = edge_index[0].long() * num_cols + edge_index[1].long()
    key = torch.unique(key, sorted=True)
    return torch.stack([key // num_cols, key % num_cols], dim=0)


def superellipsoid_surface(n: int, semi_axes=(2.685, 1.195, 0.885), power: float = 4.0,
                           generator: Optional[torch.Generator] = None) -> Tuple[Tensor, Tensor]:
    """n points + unit normals on a car-like closed surface |x/a|^p+|y/b|^p+|z/c|^p = 1
    (semi-axes ~ half the DrivAerNet++ domain extents, reference src/data/metadata.py:32)."""
    g = generator
    u = torch.randn(n, 3, generator=g)
    u = u / u.norm(dim=1, keepdim=True)
    s = (u.abs() ** power).sum(dim=1, keepdim=True) ** (-1.0 / power)
    ax = torch.tensor(semi_axes)
    p_unit = u * s                                  # on the unit superellipsoid
    pos = p_unit * ax
    nrm = torch.sign(p_unit) * p_unit.abs() ** (power - 1) / ax
    nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp(min=1e-12)
    return pos, nrm


def rescale(x: Tensor, lims=(-1.0, 1.0)) -> Tensor:
    """Global min/max rescale (reference src/utils/scale.py:13-25)."""
    return (x - x.min()) / (x.max() - x.min()) * (lims[1] - lims[0]) + lims[0]


def morton_order(pos: Tensor, bits: int = 10) -> Tensor:
    """permutation that sorts points along the Z-order (Morton) curve of their bounding box, `bits` bits per axis (stable:
    points of one cell keep their order).  A per-sample constant like the neighbour lists: points that are close in space end
    up close in memory, so the coordinate / feature rows the GNO kernels gather for one token share cache lines (no reference
    counterpart: the reference gathers through torch indexing in whatever order the mesh file has)."""
    p = pos.detach().to(torch.float64)
    lo, hi = p.min(dim=0).values, p.max(dim=0).values
    q = ((p - lo) / (hi - lo).clamp(min=1e-30) * ((1 << bits) - 1)).round().clamp(0, (1 << bits) - 1).to(torch.int64)
    code = torch.zeros(p.shape[0], dtype=torch.int64, device=pos.device)
    for b in range(bits):
        for a in range(p.shape[1]):
            code |= ((q[:, a] >> b) & 1) << (b * p.shape[1] + a)
    return torch.sort(code, stable=True).indices


def _synthetic_fields(n_points: int, out_channels: int, seed: int, surface: bool, order: str = "random"):
    g = torch.Generator().manual_seed(seed)
    if surface:
        pos, nrm = superellipsoid_surface(n_points, generator=g)
        pos = rescale(pos)
    else:
        pos = torch.rand(n_points, 3, generator=g) * 2 - 1
        nrm = torch.randn(n_points, 3, generator=g)
        nrm = nrm / nrm.norm(dim=1, keepdim=True)
    x = torch.randn(n_points, out_channels, generator=g)
    if order == "morton":      # the same point set, stored along the Z-order curve (what a mesh file's locality looks like)
        perm = morton_order(pos)
        pos, nrm, x = pos[perm].contiguous(), nrm[perm].contiguous(), x[perm].contiguous()
    elif order != "random":
        raise ValueError(f"order must be 'random' or 'morton', got {order}")
    return pos, nrm, x


def make_synthetic_shard(n_points: int, latent_tokens: Sequence[int], rank: int, world: int, k: int = 8,
                         in_normals: bool = True, out_channels: int = 1, seed: int = 0, surface: bool = True,
                         device: str = "cpu", order: str = "random") -> Tuple[MeshBatch, Tensor]:
    """Rank ``rank``'s share of ``make_synthetic_sample(n_points, ...)`` -- identical values to
    ``sharding.shard_batch`` of the whole sample -- without ever holding the whole sample on the device: the host
    draws the per-point fields (7 floats a point; the global rescale needs all of them), the rank's contiguous point
    range goes to the device and only ITS edges are built there."""
    pos, nrm, x = _synthetic_fields(n_points, out_channels, seed, surface, order)
    lo, hi = (n_points * rank) // world, (n_points * (rank + 1)) // world
    pos_d = pos[lo:hi].contiguous().to(device)
    enc = knn_edges_grid(pos_d, latent_tokens, (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), k)
    b = MeshBatch(pos=pos_d, x=x[lo:hi].contiguous().to(device), batch=torch.zeros(hi - lo, dtype=torch.long, device=device),
                  encoder_edge_index_s0=enc.to(torch.int32), decoder_edge_index_s0=enc.flip(0).to(torch.int32))
    if in_normals:
        b.c = nrm[lo:hi].contiguous().to(device)
    b.num_graphs = 1
    b.shard = (rank, world, lo, hi, n_points)
    b.ptr = torch.tensor([0, hi - lo], dtype=torch.long, device=device)
    return b, latent_grid(latent_tokens).to(device)


def make_synthetic_sample(n_points: int, latent_tokens: Sequence[int], k: int = 8, in_normals: bool = True,
                          out_channels: int = 1, seed: int = 0, surface: bool = True,
                          device: str = "cpu", order: str = "random") -> Tuple[MeshBatch, Tensor]:
    """Seeded synthetic sample of the BASELINE shapes: points on a car-like surface
    (or uniform in the cube), rescaled to [-1,1]; ``c`` = unit normals; knn(k) encoder edges
    (phys-major) and the flipped list as decoder edges; N(0,1) target.  Returns (batch, tokens_pos).
    ``order``: "random" (the draw's order: no locality at all, the worst case for the gathers) or "morton"."""
    pos, nrm, x = _synthetic_fields(n_points, out_channels, seed, surface, order)
    pos_d = pos.to(device)
    enc = knn_edges_grid(pos_d, latent_tokens, (-1.0, -1.0, -1.0), (1.0, 1.0, 1.0), k)
    b = MeshBatch(pos=pos_d, x=x.to(device), batch=torch.zeros(n_points, dtype=torch.long, device=device),
                  encoder_edge_index_s0=enc.to(torch.int32), decoder_edge_index_s0=enc.flip(0).to(torch.int32))
    if in_normals:
        b.c = nrm.to(device)
    b.num_graphs = 1
    b.ptr = torch.tensor([0, n_points], dtype=torch.long, device=device)
    return b, latent_grid(latent_tokens).to(device)
