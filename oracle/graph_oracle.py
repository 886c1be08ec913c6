"""TEST INFRASTRUCTURE -- CPU restatement of the reference's neighbour-graph construction
(src/model/layers/magno.py:116-295, ``get_neighbor_strategy`` and friends).  Only tests/ may import this file; the
product (gaot_3d_amd/graph.py, csrc/graph.hip; host helpers in gaot_3d_amd/data.py) never does.

Parity: "third-party, unpinned" -- the reference delegates the searches to ``torch_geometric.nn.knn`` / ``radius``
(torch_cluster) and ``torch_geometric.utils.coalesce``, none of which is present here (requirements.txt:7 leaves their
versions open and the reference ships no graph fixture).  What is restated is their published contract as the reference
uses it:
  * knn(x=latent, y=phys, k)            -> for every phys point its k nearest latent tokens, rows [phys, latent]
                                           (magno.py:183-189); ties -> lowest token index (stable order)
  * radius(x=phys, y=latent, r)         -> for every latent CENTRE the phys points with distance <= r, at most
                                           max_num_neighbors = 32 of them (PyG default; the override is commented out,
                                           magno.py:199), taken in ascending point index; rows [phys, latent]
                                           (magno.py:193-201)
  * decoder: the same searches with the roles swapped, rows [latent, phys] (magno.py:242-261)
  * 'bidirectional' = coalesce(cat(knn, radius)): sorted by (row 0, row 1), duplicates dropped (magno.py:219-220, 292-293)
  * 'reverse' (decoder only) = flip of a freshly built BIDIRECTIONAL encoder graph, whatever the encoder strategy
    (magno.py:263-273)
  * batches: every graph is searched on its own; indices are global (offset by the batch masks) (magno.py:176-181)
All distances in float64 so that the checker itself has no rounding ties the fp32 kernels could disagree with; the tests
compare edge SETS where ties in fp32 are possible (knn) and exact lists elsewhere.
"""
from typing import Optional

import torch

Tensor = torch.Tensor


def _dist(a: Tensor, b: Tensor) -> Tensor:
    return torch.cdist(a.double(), b.double())


def knn(phys: Tensor, latent: Tensor, k: int) -> Tensor:
    """[2, N*k] rows [phys, latent], grouped by phys point, nearest first (magno.py:183-189)"""
    d = _dist(phys, latent)
    idx = torch.argsort(d, dim=1, stable=True)[:, :k]
    n = phys.shape[0]
    return torch.stack([torch.arange(n).repeat_interleave(k), idx.reshape(-1)])


def radius(centres: Tensor, others: Tensor, r: float, cap: Optional[int] = 32):
    """for every centre the indices of `others` within r (ascending index, at most `cap`): (centre idx, other idx) lists"""
    d = _dist(centres, others)
    ci, oi = [], []
    for c in range(centres.shape[0]):
        hit = torch.nonzero(d[c] <= r, as_tuple=True)[0]
        if cap is not None:
            hit = hit[:cap]
        ci.append(torch.full((hit.numel(),), c, dtype=torch.long))
        oi.append(hit)
    if not ci:
        return torch.zeros(0, dtype=torch.long), torch.zeros(0, dtype=torch.long)
    return torch.cat(ci), torch.cat(oi)


def coalesce(e: Tensor) -> Tensor:
    """sort by (row 0, row 1), drop duplicates (torch_geometric.utils.coalesce as used at magno.py:220, 293)"""
    if e.shape[1] == 0:
        return e
    pairs = sorted(set(map(tuple, e.t().tolist())))
    return torch.tensor(pairs, dtype=torch.long).t().contiguous()


def encoder_edges(strategy: str, phys: Tensor, latent: Tensor, r: float, k: int) -> Tensor:
    """rows [phys, latent] (magno.py:176-221)"""
    if strategy == "knn":
        return knn(phys, latent, k)
    if strategy == "radius":
        c, o = radius(latent, phys, r)          # centres = latent tokens
        return torch.stack([o, c])
    if strategy == "bidirectional":
        c, o = radius(latent, phys, r)
        return coalesce(torch.cat([knn(phys, latent, k), torch.stack([o, c])], dim=1))
    raise ValueError(f"Unknown encoder strategy: {strategy}")


def decoder_edges(strategy: str, phys: Tensor, latent: Tensor, r: float, k: int) -> Tensor:
    """rows [latent, phys] (magno.py:235-295)"""
    if strategy == "reverse":
        return encoder_edges("bidirectional", phys, latent, r, k).flip(0)
    if strategy == "knn":
        return knn(phys, latent, k).flip(0)
    if strategy == "radius":
        c, o = radius(phys, latent, r)          # centres = phys points
        return torch.stack([o, c])
    if strategy == "bidirectional":
        c, o = radius(phys, latent, r)
        return coalesce(torch.cat([knn(phys, latent, k).flip(0), torch.stack([o, c])], dim=1))
    raise ValueError(f"Unknown decoder strategy: {strategy}")


def get_neighbor_strategy(neighbor_strategy: str, phys_pos: Tensor, batch_idx_phys: Optional[Tensor], latent_tokens_pos: Tensor,
                          batch_idx_latent: Optional[Tensor], radius_: float, k_neighbors: int = 1,
                          is_decoder: bool = False) -> Tensor:
    """the reference function's result for CPU tensors: per graph of the batch, global indices (magno.py:116-124)"""
    if batch_idx_phys is None:
        batch_idx_phys = torch.zeros(phys_pos.shape[0], dtype=torch.long)
    if batch_idx_latent is None:
        batch_idx_latent = torch.zeros(latent_tokens_pos.shape[0], dtype=torch.long)
    nb = int(batch_idx_phys.max()) + 1 if batch_idx_phys.numel() else 1
    outs = []
    for b in range(nb):
        pm = torch.nonzero(batch_idx_phys == b, as_tuple=True)[0]
        lm = torch.nonzero(batch_idx_latent == b, as_tuple=True)[0]
        p, l = phys_pos[pm], latent_tokens_pos[lm]
        if is_decoder:
            e = decoder_edges(neighbor_strategy, p, l, radius_, k_neighbors)
            outs.append(torch.stack([lm[e[0]], pm[e[1]]]))
        else:
            e = encoder_edges(neighbor_strategy, p, l, radius_, k_neighbors)
            outs.append(torch.stack([pm[e[0]], lm[e[1]]]))
    return torch.cat(outs, dim=1) if outs else torch.zeros(2, 0, dtype=torch.long)
