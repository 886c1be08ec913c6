"""Neighbour-graph construction on the device: the reference's ``get_neighbor_strategy`` and friends
(src/model/layers/magno.py:116-295), which run torch_cluster knn / radius on the CPU, restated for latent tokens that
form a regular D x H x W grid (gaot_3d.py:35-46; the trainer's rescaled copy stat.py:238-252) and executed by the HIP
kernels of csrc/graph.hip through the C ABI.  Conventions kept: encoder edges are ``[phys_idx, latent_idx]``, decoder
edges ``[latent_idx, phys_idx]``; 'bidirectional' = coalesce(cat(knn, radius)); 'reverse' = flip of the *bidirectional*
encoder graph whatever the encoder strategy is (magno.py:263-273); pyg_radius keeps at most 32 neighbours per centre
(magno.py:199, 259).  Outputs are int32 ``[2, E]`` tensors on the device (the on-disk dtype, stat.py:191, 208).

The only host synchronisations are the reads of list lengths (edge counts size the outputs); graph construction sits in
the data path, outside the captured training step.  No CPU fallback: a CPU tensor or an irregular token set raises."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch

from . import _lib, ops
from ._lib import GaotError, check

Tensor = torch.Tensor
RADIUS_CAP = 32   # torch_cluster / PyG default max_num_neighbors (reference leaves it at the default)


class _Grid(C.Structure):   # gaot_grid_t
    _fields_ = [("dims", C.c_int32 * 3), ("lo", C.c_float * 3), ("hi", C.c_float * 3)]


@dataclass
class LatentGrid:
    dims: Tuple[int, int, int]
    lo: Tuple[float, float, float]
    hi: Tuple[float, float, float]
    pos: Tensor   # [D*H*W, 3] fp32 on the device: the coordinates distances are measured to

    @property
    def num_tokens(self) -> int:
        return self.dims[0] * self.dims[1] * self.dims[2]

    def c_struct(self) -> _Grid:
        return _Grid((C.c_int32 * 3)(*self.dims), (C.c_float * 3)(*self.lo), (C.c_float * 3)(*self.hi))


def as_latent_grid(latent_pos: Tensor, dims: Sequence[int], rtol: float = 1e-4) -> LatentGrid:
    """Check that ``latent_pos [D*H*W, 3]`` is the reference's ``meshgrid(linspace, indexing='ij')`` grid (possibly
    rescaled per axis) and describe it.  Raises GaotError otherwise (no tree search on this path)."""
    d, h, w = (int(v) for v in dims)
    if latent_pos.dim() != 2 or latent_pos.shape[1] != 3 or latent_pos.shape[0] != d * h * w:
        raise GaotError(f"latent tokens {tuple(latent_pos.shape)} do not form a {d}x{h}x{w} grid")
    p = latent_pos.detach().to(torch.float32).contiguous()
    g = p.view(d, h, w, 3)
    lo = [float(g[0, 0, 0, a]) for a in range(3)]
    hi = [float(g[-1, -1, -1, a]) for a in range(3)]
    span = max(max(abs(hi[a] - lo[a]) for a in range(3)), 1e-30)
    ref = torch.stack(torch.meshgrid(*[torch.linspace(lo[a], hi[a], n, device=p.device) for a, n in enumerate((d, h, w))],
                                     indexing="ij"), dim=-1)
    if float((g - ref).abs().max()) > rtol * span or any((n > 1 and not hi[a] > lo[a]) for a, n in enumerate((d, h, w))):
        raise GaotError("latent tokens are not a regular ascending meshgrid(linspace) grid: device graph construction "
                        "needs the reference's regular token grid")
    return LatentGrid((d, h, w), tuple(lo), tuple(hi), p)


@dataclass
class TokenSet:
    """latent tokens that are NOT a regular grid: searched by the brute-force device kernels (every point scans all tokens)"""
    pos: Tensor   # [M, 3] fp32 on the device

    @property
    def num_tokens(self) -> int:
        return int(self.pos.shape[0])


def _need_cuda(t: Tensor, what: str):
    if not t.is_cuda:
        raise GaotError(f"{what}: device graph construction runs on the HIP device only (got a {t.device} tensor)")


def exclusive_scan(x: Tensor) -> Tensor:
    """int32 [n] -> int32 [n+1] exclusive prefix sums (last element = total)"""
    lib = _lib.load()
    n = x.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=x.device)
    ws = torch.empty(max(lib.gaot_exclusive_scan_workspace_bytes(n), 16), dtype=torch.uint8, device=x.device)
    check(lib.gaot_exclusive_scan_i32(ops._ptr(x), n, ops._ptr(out), ops._ptr(ws), ws.numel(), ops._stream()),
          "gaot_exclusive_scan_i32")
    return out


def knn_to_grid(phys_pos: Tensor, grid: LatentGrid, k: int) -> Tensor:
    """int32 [N, k]: the k nearest tokens of every point, by (distance, token index)"""
    _need_cuda(phys_pos, "knn_to_grid")
    lib = _lib.load()
    p = phys_pos.detach().to(torch.float32).contiguous()
    out = torch.empty(p.shape[0], k, dtype=torch.int32, device=p.device)
    if isinstance(grid, TokenSet):
        check(lib.gaot_knn_brute(ops._ptr(p), p.shape[0], ops._ptr(grid.pos), grid.num_tokens, k, ops._ptr(out), ops._stream()),
              "gaot_knn_brute")
        return out
    gs = grid.c_struct()
    check(lib.gaot_knn_grid(ops._ptr(p), p.shape[0], C.byref(gs), ops._ptr(grid.pos), k, ops._ptr(out), ops._stream()),
          "gaot_knn_grid")
    return out


def radius_pairs(phys_pos: Tensor, grid: LatentGrid, radius: float, cap: Optional[int]) -> Tuple[Tensor, Tensor]:
    """(point_idx, token_idx) int32 lists of all pairs with d <= radius, grouped by point (ascending), tokens ascending
    inside a point, at most ``cap`` tokens per point (None: no cap)"""
    _need_cuda(phys_pos, "radius_pairs")
    lib = _lib.load()
    p = phys_pos.detach().to(torch.float32).contiguous()
    n = p.shape[0]
    brute = isinstance(grid, TokenSet)
    gs = None if brute else grid.c_struct()
    capv = int(cap) if cap is not None else 0x7fffffff
    counts = torch.empty(n, dtype=torch.int32, device=p.device)
    if brute:
        check(lib.gaot_radius_brute_count(ops._ptr(p), n, ops._ptr(grid.pos), grid.num_tokens, float(radius), capv,
                                          ops._ptr(counts), ops._stream()), "gaot_radius_brute_count")
    else:
        check(lib.gaot_radius_grid_count(ops._ptr(p), n, C.byref(gs), ops._ptr(grid.pos), float(radius), capv, ops._ptr(counts),
                                         ops._stream()), "gaot_radius_grid_count")
    offs = exclusive_scan(counts)
    total = int(offs[-1])     # host sync: sizes the output
    pt = torch.empty(total, dtype=torch.int32, device=p.device)
    tk = torch.empty(total, dtype=torch.int32, device=p.device)
    if total and brute:
        check(lib.gaot_radius_brute_fill(ops._ptr(p), n, ops._ptr(grid.pos), grid.num_tokens, float(radius), capv, ops._ptr(offs),
                                         ops._ptr(pt), ops._ptr(tk), ops._stream()), "gaot_radius_brute_fill")
    elif total:
        check(lib.gaot_radius_grid_fill(ops._ptr(p), n, C.byref(gs), ops._ptr(grid.pos), float(radius), capv, ops._ptr(offs),
                                        ops._ptr(pt), ops._ptr(tk), ops._stream()), "gaot_radius_grid_fill")
    return pt, tk


def _compact(a: Tensor, b: Tensor, flags: Tensor) -> Tuple[Tensor, Tensor]:
    lib = _lib.load()
    offs = exclusive_scan(flags)
    total = int(offs[-1])
    oa = torch.empty(total, dtype=torch.int32, device=a.device)
    ob = torch.empty(total, dtype=torch.int32, device=a.device)
    if total:
        check(lib.gaot_compact_pairs(ops._ptr(a), ops._ptr(b), ops._ptr(flags), ops._ptr(offs), a.numel(), ops._ptr(oa),
                                     ops._ptr(ob), ops._stream()), "gaot_compact_pairs")
    return oa, ob


def cap_per_key(key: Tensor, other: Tensor, num_keys: int, cap: int) -> Tuple[Tensor, Tensor]:
    """stable-sort (key, other) by key and keep the first ``cap`` entries of every key (pyg_radius' per-centre cap)"""
    lib = _lib.load()
    if key.numel() == 0:
        return key, other
    s = ops.csr_build(torch.stack([other, key]), 1, num_keys)
    flags = torch.empty(key.numel(), dtype=torch.int32, device=key.device)
    check(lib.gaot_segment_cap_flags(ops._ptr(s.rowptr), ops._ptr(s.key), key.numel(), int(cap), ops._ptr(flags), ops._stream()),
          "gaot_segment_cap_flags")
    return _compact(s.key, s.other, flags)


def apply_neighbor_sampling(edge_index: Tensor, num_query_nodes: int, device=None, sampling_strategy: Optional[str] = None,
                            max_neighbors: Optional[int] = None, sample_ratio: Optional[float] = None,
                            training: bool = True, seed: Optional[Tensor] = None) -> Tensor:
    """reference magno.py:297-371, same arguments (+ ``seed``: one-element int64 device tensor; default = the next
    word of the device seed stream, functional.next_dropout_seed).  'max_neighbors': every query keeps at most
    ``max_neighbors`` uniformly random edges (applied in training and eval alike, as in the reference); 'ratio': every
    edge is kept with probability ``sample_ratio`` in training mode.  The result is sorted by query (the reference keeps
    the input order; the operators downstream do not depend on it)."""
    if sampling_strategy is None:
        return edge_index
    if num_query_nodes == 0 or edge_index.shape[1] == 0:
        return edge_index
    if sampling_strategy == "max_neighbors":
        if max_neighbors is None:
            raise ValueError("max_neighbors must be provided when using 'max_neighbors' sampling strategy")
    elif sampling_strategy == "ratio":
        if sample_ratio is None:
            raise ValueError("sample_ratio must be provided when using 'ratio' sampling strategy")
        if sample_ratio >= 1.0 or not training:
            return edge_index
    else:
        raise ValueError(f"Invalid sampling strategy: {sampling_strategy}")
    _need_cuda(edge_index, "edge_index")
    lib = _lib.load()
    if seed is None:
        from .functional import next_dropout_seed
        seed = next_dropout_seed(edge_index.device)
    e = edge_index.shape[1]
    flags = torch.empty(e, dtype=torch.int32, device=edge_index.device)
    s = ops.csr_build(edge_index, 1, num_query_nodes)          # key = query, other = source
    if sampling_strategy == "max_neighbors":
        check(lib.gaot_segment_random_cap_flags(ops._ptr(seed), ops._ptr(s.rowptr), ops._ptr(s.key), e, int(max_neighbors),
                                                ops._ptr(flags), ops._stream()), "gaot_segment_random_cap_flags")
    else:
        check(lib.gaot_random_keep_flags(ops._ptr(seed), e, float(sample_ratio), ops._ptr(flags), ops._stream()),
              "gaot_random_keep_flags")
    q, src = _compact(s.key, s.other, flags)
    return torch.stack([src, q]).to(edge_index.dtype)


def coalesce(row0: Tensor, row1: Tensor, num_row0: int, num_row1: int) -> Tensor:
    """sort by (row0, row1) and drop duplicates -- torch_geometric.utils.coalesce as the reference uses it for
    'bidirectional' (magno.py:219-220): stable sort by row1, stable sort by row0, adjacent-unique, compact"""
    lib = _lib.load()
    if row0.numel() == 0:
        return torch.stack([row0, row1])
    s1 = ops.csr_build(torch.stack([row0, row1]), 1, num_row1)       # key = row1, other = row0
    s0 = ops.csr_build(torch.stack([s1.other, s1.key]), 0, num_row0)  # key = row0 (stable: row1 stays ascending)
    flags = torch.empty(row0.numel(), dtype=torch.int32, device=row0.device)
    check(lib.gaot_unique_pair_flags(ops._ptr(s0.key), ops._ptr(s0.other), row0.numel(), ops._ptr(flags), ops._stream()),
          "gaot_unique_pair_flags")
    a, b = _compact(s0.key, s0.other, flags)
    return torch.stack([a, b])


def _encoder_edges(strategy: str, p: Tensor, grid: LatentGrid, radius: float, k: int) -> Tensor:
    """[phys_idx, latent_idx] for one graph (magno.py:166-222)"""
    n, m = p.shape[0], grid.num_tokens
    knn = rad = None
    if strategy in ("knn", "bidirectional"):
        idx = knn_to_grid(p, grid, k)
        src = torch.arange(n, dtype=torch.int32, device=p.device).repeat_interleave(k)
        knn = torch.stack([src, idx.reshape(-1)])
    if strategy in ("radius", "bidirectional"):
        pt, tk = radius_pairs(p, grid, radius, None)          # every pair; the cap applies per TOKEN here
        tk2, pt2 = cap_per_key(tk, pt, m, RADIUS_CAP)          # grouped by token, points ascending, <= 32 per token
        rad = torch.stack([pt2, tk2])
    if strategy == "knn":
        return knn
    if strategy == "radius":
        return rad
    if strategy == "bidirectional":
        both = torch.cat([knn, rad], dim=1)
        return coalesce(both[0].contiguous(), both[1].contiguous(), n, m)
    raise ValueError(f"Unknown encoder strategy: {strategy}")


def _decoder_edges(strategy: str, p: Tensor, grid: LatentGrid, radius: float, k: int) -> Tensor:
    """[latent_idx, phys_idx] for one graph (magno.py:224-295)"""
    n, m = p.shape[0], grid.num_tokens
    if strategy == "reverse":
        return _encoder_edges("bidirectional", p, grid, radius, k).flip(0).contiguous()
    knn = rad = None
    if strategy in ("knn", "bidirectional"):
        idx = knn_to_grid(p, grid, k)
        dst = torch.arange(n, dtype=torch.int32, device=p.device).repeat_interleave(k)
        knn = torch.stack([idx.reshape(-1), dst])
    if strategy in ("radius", "bidirectional"):
        pt, tk = radius_pairs(p, grid, radius, RADIUS_CAP)    # points are the centres: <= 32 tokens per point
        rad = torch.stack([tk, pt])
    if strategy == "knn":
        return knn
    if strategy == "radius":
        return rad
    if strategy == "bidirectional":
        both = torch.cat([knn, rad], dim=1)
        return coalesce(both[0].contiguous(), both[1].contiguous(), m, n)
    raise ValueError(f"Unknown decoder strategy: {strategy}")


def get_neighbor_strategy(neighbor_strategy: str, phys_pos: Tensor, batch_idx_phys: Optional[Tensor], latent_tokens_pos: Tensor,
                          batch_idx_latent: Optional[Tensor], radius: float, k_neighbors: int = 1, is_decoder: bool = False,
                          latent_dims: Optional[Sequence[int]] = None) -> Tensor:
    """Device version of the reference function of the same name (magno.py:116-164; same arguments plus the grid shape).
    ``latent_tokens_pos`` holds the tokens of every graph of the batch back to back (gaot_3d.py:283-285: the same grid
    tiled ``num_graphs`` times); ``batch_idx_*`` are the sorted PyG batch vectors (None = one graph)."""
    _need_cuda(phys_pos, "get_neighbor_strategy")
    lat = latent_tokens_pos.detach().to(phys_pos.device, torch.float32).contiguous()
    m = int(latent_dims[0]) * int(latent_dims[1]) * int(latent_dims[2]) if latent_dims is not None else 0
    regular = m > 0 and lat.shape[0] % m == 0 and lat.shape[0] > 0
    if regular:
        nb = lat.shape[0] // m
        lptr = [b * m for b in range(nb + 1)]
    else:   # any token set: the graphs of the batch are the runs of the (sorted) token batch vector
        if batch_idx_latent is None or batch_idx_latent.numel() == 0:
            nb, lptr = 1, [0, lat.shape[0]]
        else:
            nb = int(batch_idx_latent.max().item()) + 1
            lptr = [0] + torch.cumsum(torch.bincount(batch_idx_latent, minlength=nb), 0).tolist()
    if batch_idx_phys is None or nb == 1:
        ptr = [0, phys_pos.shape[0]]
    else:
        cnt = torch.bincount(batch_idx_phys, minlength=nb)
        ptr = [0] + torch.cumsum(cnt, 0).tolist()
    outs = []
    for b in range(nb):
        toks = lat[lptr[b]:lptr[b + 1]]
        grid = None
        if regular:
            try:
                grid = as_latent_grid(toks, latent_dims)
            except GaotError:
                grid = None
        if grid is None:   # not the reference's regular grid: brute-force kernels (csrc/graph.hip: k_knn_brute, k_radius_brute)
            grid = TokenSet(toks)
        p = phys_pos[ptr[b]:ptr[b + 1]]
        e = (_decoder_edges if is_decoder else _encoder_edges)(neighbor_strategy, p, grid, radius, k_neighbors)
        if b:
            inc = torch.tensor([[lptr[b]], [ptr[b]]] if is_decoder else [[ptr[b]], [lptr[b]]], dtype=torch.int32, device=e.device)
            e = e + inc
        outs.append(e)
    return outs[0] if nb == 1 else torch.cat(outs, dim=1)
