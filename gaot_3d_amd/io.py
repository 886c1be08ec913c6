"""On-disk samples of the reference's data layer without torch_geometric.

The reference stores one ``torch_geometric.data.Data`` / ``EnrichedData`` per ``<name>.pt`` (src/data/pyg_datasets.py:
125-137, written by src/trainer/stat.py:163-214) with ``pos [N,3]``, ``x [N,out]``, optional ``c``, ``filename``,
``num_latent_nodes`` and -- when the edges are precomputed -- ``encoder_edge_index_s{i}`` / ``decoder_edge_index_s{i}``
(int32 ``[2,E]``) and ``{encoder,decoder}_query_counts_s{i}`` (int32).  PyG is a third-party dependency that is absent
here ("third-party, unpinned": no sample file or fixture of the reference is available to pin the pickle layout), so the
reader makes no assumption beyond "every torch_geometric class is an attribute bag": classes of ``torch_geometric.*`` and
the reference's own ``EnrichedData`` are unpickled as inert stand-ins and the tensors / scalars are collected from the
object's storage mapping.  ``enrich_sample`` is the device version of the trainer's edge pre-computation pass."""
from __future__ import annotations

import pickle
from typing import Any, Dict, Optional, Sequence

import torch

from .data import MeshBatch, rescale

Tensor = torch.Tensor


class _Bag:
    """inert stand-in for any torch_geometric / reference data class met in a pickle"""

    def __init__(self, *args, **kwargs):
        self._args, self._kwargs = args, kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple):      # (dict, slots-dict)
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        else:
            self.__dict__["_state"] = state


# Globals a tensor pickle legitimately needs.  Everything else met in a sample file is either a data-container class
# (torch_geometric.*, the reference's EnrichedData, any other class) -> inert attribute bag, or refused: a crafted .pt
# cannot import and call arbitrary functions (os.system, builtins.eval, ...) through this reader.
_ALLOWED = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
    ("builtins", "int"), ("builtins", "float"), ("builtins", "str"), ("builtins", "bool"), ("builtins", "bytes"),
    ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_tensor_v3"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.serialization", "_get_layout"), ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Tensor"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
}
_STORAGE_SUFFIX = "Storage"


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        if module in ("torch", "torch.storage") and name.endswith(_STORAGE_SUFFIX):   # FloatStorage, UntypedStorage, ...
            return super().find_class(module, name)
        if module == "gaot_3d_amd.data" and name == "MeshBatch":
            return MeshBatch
        root = module.split(".")[0]
        if root in ("os", "posix", "nt", "subprocess", "sys", "builtins", "importlib", "runpy", "shutil", "socket", "pickle",
                    "ctypes", "operator", "functools", "types", "code", "codecs", "pty", "commands", "webbrowser"):
            raise pickle.UnpicklingError(f"refusing to load global {module}.{name} from a sample file")
        # torch_geometric.*, the reference's src.data.pyg_datasets.EnrichedData and any other data-container class
        return type(name, (_Bag,), {"__module__": module})


class _PickleModule:
    """what torch.load expects of ``pickle_module``"""
    __name__ = "gaot_3d_amd.io"
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError, UnpicklingError, PicklingError = pickle.PickleError, pickle.UnpicklingError, pickle.PicklingError


def _collect(obj: Any, out: Dict[str, Any], depth: int = 0) -> None:
    """tensors and plain values reachable through the attribute bags' dicts (Data -> _store -> _mapping)"""
    if depth > 4:
        return
    items = obj.items() if isinstance(obj, dict) else (obj.__dict__.items() if isinstance(obj, _Bag) else ())
    for k, v in items:
        if isinstance(v, (_Bag, dict)) and (k.startswith("_") or isinstance(v, _Bag)):
            if k != "_parent":
                _collect(v, out, depth + 1)
        elif isinstance(k, str) and not k.startswith("_") and k not in out:
            if torch.is_tensor(v) or isinstance(v, (int, float, str, list, tuple)):
                out[k] = v


def load_sample(path: str, active_variables: Optional[Sequence[int]] = None, map_location="cpu") -> MeshBatch:
    """one reference ``.pt`` sample as a single-graph MeshBatch (dataset ``get``: pyg_datasets.py:125-137, including
    the ``active_variables`` column selection and the squeeze of a trailing singleton axis of ``x``)"""
    obj = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)
    fields: Dict[str, Any] = {}
    if isinstance(obj, MeshBatch):
        fields = dict(obj.__dict__)
    elif isinstance(obj, dict):
        fields = {k: v for k, v in obj.items() if isinstance(k, str)}
    else:
        _collect(obj, fields)
    if "pos" not in fields:
        raise ValueError(f"{path}: no 'pos' attribute found")
    x = fields.get("x")
    if torch.is_tensor(x):
        if active_variables is not None:
            x = x[:, list(active_variables)]
        if x.dim() == 3:
            x = x.squeeze(-1)
        fields["x"] = x
    b = MeshBatch(**fields)
    n = b.pos.shape[0]
    b.num_graphs = 1
    b.batch = torch.zeros(n, dtype=torch.long, device=b.pos.device)
    b.ptr = torch.tensor([0, n], dtype=torch.long, device=b.pos.device)
    return b


def save_sample(sample: MeshBatch, path: str) -> None:
    """plain-dict form of a sample (tensors on the CPU); ``load_sample`` reads it back"""
    d = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in sample.__dict__.items()
         if not k.startswith("_") and k not in ("batch", "ptr", "num_graphs")}
    torch.save(d, path)


def enrich_sample(sample: MeshBatch, latent_tokens_pos: Tensor, gno_config, latent_dims=None, device=None,
                  rescale_pos: bool = True, reorder: Optional[str] = None) -> MeshBatch:
    """the trainer's pre-computation pass (stat.py:163-214): rescale the coordinates to [-1, 1], build the encoder /
    decoder edge lists of every scale with ``get_neighbor_strategy`` -- on ``device`` through the graph kernels when the
    tokens are a regular grid (``latent_dims``) -- and store them as int32 together with the per-query counts and
    ``num_latent_nodes``.  Returns a new sample on the CPU (what the reference writes back to the .pt file).
    ``reorder="morton"`` (extension): the points -- ``pos`` and every per-point attribute -- are stored along the Z-order curve
    before the edges are built (``data.morton_order``); ``point_perm`` keeps the permutation (row i of the enriched sample is row
    ``point_perm[i]`` of the file), so predictions can be put back in file order with ``pred[torch.argsort(point_perm)]``."""
    from .data import morton_order
    from .model.layers.magno import get_neighbor_strategy, parse_neighbor_strategy
    dev = torch.device(device) if device is not None else sample.pos.device
    out = MeshBatch(**{k: v for k, v in sample.__dict__.items() if not k.startswith("_")})
    if reorder == "morton":
        perm = morton_order(sample.pos).cpu()
        n0 = sample.pos.shape[0]
        for k, v in list(out.__dict__.items()):
            if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n0 and not k.startswith(("encoder_", "decoder_")):
                setattr(out, k, v[perm.to(v.device)].contiguous())
        out.point_perm = perm
        sample = out
    elif reorder is not None:
        raise ValueError(f"reorder must be None or 'morton', got {reorder}")
    pos = sample.pos.to(torch.float32)
    pos = rescale(pos, (-1, 1)) if rescale_pos else pos
    lat = latent_tokens_pos.to(dev, torch.float32)
    p = pos.to(dev)
    n, m = p.shape[0], lat.shape[0]
    bp = torch.zeros(n, dtype=torch.long, device=dev)
    bl = torch.zeros(m, dtype=torch.long, device=dev)
    enc_s, dec_s = parse_neighbor_strategy(gno_config.neighbor_strategy)
    for si, scale in enumerate(gno_config.scales):
        r = gno_config.gno_radius * scale
        for name, strat, is_dec, nq in (("encoder", enc_s, False, m), ("decoder", dec_s, True, n)):
            ei = get_neighbor_strategy(strat, p, bp, lat, bl, r, gno_config.k_neighbors, is_dec, latent_dims=latent_dims)
            ei = ei.to(torch.int32)
            cnt = torch.bincount(ei[1].long(), minlength=nq).to(torch.int32) if ei.numel() else torch.zeros(nq, dtype=torch.int32, device=dev)
            setattr(out, f"{name}_edge_index_s{si}", ei.cpu())
            setattr(out, f"{name}_query_counts_s{si}", cnt.cpu())
    out.num_latent_nodes = m
    return out
