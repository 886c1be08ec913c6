"""The callers' side of the hot path: the reference's per-sample dataset, its transforms, the normalisation
statistics file and a loader that hands the model batches already resident in HBM -- without torch_geometric.

  * ``VTKMeshDataset``      reference src/data/pyg_datasets.py:33-142: ``<processed>/<name>.pt`` per sample, split by the
                            order file and the config's train / val / test sizes (``rand_dataset``: numpy default_rng(42)).
  * ``RescalePosition`` / ``RescalePositionNew`` / ``NormalizeFeatures``   src/data/pyg_transforms.py:16-105.
  * ``calculate_or_load_stats``   src/trainer/stat.py:56-124: ``{name}_norm_stats.pt`` with keys mean / std [/ c_mean / c_std],
                            unbiased std over all training points.
  * ``SampleLoader``        batches samples with the ``EnrichedData.__inc__`` offsets (pyg_datasets.py:9-31, via
                            ``MeshBatch.from_data_list``), stages them through PINNED host memory and a side HIP stream so the
                            copy of batch i+1 overlaps the step on batch i, and keeps the most recently used batches on the
                            device together with what the model cached on them (the row-sorted neighbour lists and the
                            geometry-only GeoEmbed statistics are per-sample constants).

Pickle layout of PyG ``Data`` objects: see ``gaot_3d_amd.io`` ("third-party, unpinned")."""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Callable, Iterator, List, Optional, Sequence

import numpy as np
import torch

from .data import MeshBatch, rescale
from .io import load_sample

Tensor = torch.Tensor
EPSILON = 1e-10   # pyg_transforms.py:9


# ---- transforms (pyg_transforms.py:16-105) -----------------------------------------------------------------------------
class RescalePosition:
    """pos -> (pos - min) / (max - min) * (hi - lo) + lo with the GLOBAL min / max over all coordinates (scale.py:13-25)"""

    def __init__(self, lims=(-1., 1.)):
        self.lims = lims

    def __call__(self, data):
        if getattr(data, "pos", None) is not None:
            data.pos = rescale(data.pos, lims=self.lims)
        return data

    forward = __call__

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(lims={self.lims})"


def rescale_new(x: Tensor, lims=(-1, 1), phys_domain=([-1, -1, -1], [1, 1, 1])) -> Tensor:
    """scale.py:5-11: one min / max over ALL entries of the fixed physical domain"""
    dom = torch.tensor(phys_domain)
    lo, hi = dom.min(), dom.max()
    return ((x - lo) / (hi - lo)) * (lims[1] - lims[0]) + lims[0]


class RescalePositionNew:
    def __init__(self, lims=(-1., 1.), phy_domain=([-1.16, -1.2, 0.0], [4.21, 1.19, 1.77])):
        self.lims = lims
        self.phy_domain = phy_domain

    def __call__(self, data):
        if getattr(data, "pos", None) is not None:
            data.pos = rescale_new(data.pos, lims=self.lims, phys_domain=self.phy_domain)
        return data

    forward = __call__

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(lims={self.lims})"


class NormalizeFeatures:
    """x -> (x - mean) / (std + 1e-10), and c likewise when its statistics are given (pyg_transforms.py:67-105)"""

    def __init__(self, mean: Tensor, std: Tensor, c_mean: Optional[Tensor] = None, c_std: Optional[Tensor] = None):
        self.mean, self.std = mean.detach(), std.detach()
        self.c_mean = c_mean.detach() if c_mean is not None else None
        self.c_std = c_std.detach() if c_std is not None else None

    def __call__(self, data):
        if getattr(data, "x", None) is not None:
            data.x = (data.x - self.mean.to(data.x.device)) / (self.std.to(data.x.device) + EPSILON)
        if getattr(data, "c", None) is not None and self.c_mean is not None and self.c_std is not None:
            data.c = (data.c - self.c_mean.to(data.c.device)) / (self.c_std.to(data.c.device) + EPSILON)
        return data

    forward = __call__

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(mean=..., std=..., has_c_norm={self.c_mean is not None and self.c_std is not None})"


class Compose:
    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = list(transforms)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


# ---- dataset (pyg_datasets.py:33-142) -----------------------------------------------------------------------------------
class VTKMeshDataset:
    def __init__(self, root, order_file, dataset_config, split="train", transform=None, pre_transform=None, pre_filter=None):
        self.root, self.order_file, self.dataset_config, self.split = root, order_file, dataset_config, split
        self.transform = transform
        self.active_variables = getattr(dataset_config, "active_variables", None)
        self._load_split_indices()

    @property
    def processed_dir(self) -> str:
        return os.path.join(self.root, self.dataset_config.processed_folder)

    @property
    def processed_file_names(self) -> List[str]:
        with open(self.order_file, "r") as f:
            return [f"{line.strip()}.pt" for line in f if line.strip()]

    def _load_split_indices(self):
        with open(self.order_file, "r") as f:
            names = [line.strip() for line in f if line.strip()]
        cfg = self.dataset_config
        idx = np.arange(len(names))
        if getattr(cfg, "rand_dataset", False):
            np.random.default_rng(seed=42).shuffle(idx)
        if self.split == "train":
            sel = idx[:cfg.train_size]
        elif self.split == "val":
            sel = idx[cfg.train_size:cfg.train_size + cfg.val_size]
        elif self.split == "test":
            sel = idx[-cfg.test_size:]
        else:
            raise ValueError(f"Invalid split: {self.split}")
        self.split_filenames = [f"{names[i]}.pt" for i in sel]

    def __len__(self) -> int:
        return len(self.split_filenames)

    len = __len__

    def get(self, idx: int) -> MeshBatch:
        path = os.path.join(self.processed_dir, self.split_filenames[idx])
        if not os.path.exists(path):
            raise FileNotFoundError(f"Processed file not found: {path}. Ensure preprocessing script was run.")
        return load_sample(path, active_variables=self.active_variables)

    def __getitem__(self, idx: int) -> MeshBatch:
        data = self.get(idx)
        return self.transform(data) if self.transform is not None else data

    def __iter__(self) -> Iterator[MeshBatch]:
        return (self[i] for i in range(len(self)))


# ---- normalisation statistics (stat.py:56-124) ---------------------------------------------------------------------------
def calculate_or_load_stats(dataset_config, order_file_path: str, data_root: str, dtype=torch.float32) -> dict:
    """-> {"mean", "std"[, "c_mean", "c_std"]}: loaded from ``{data_root}/{name}_norm_stats.pt`` when present (and
    ``force_recompute_stats`` is not set), otherwise the per-column mean and UNBIASED standard deviation over every point of
    the training split (positions rescaled first, as the reference's temporary dataset does), saved to that file.  The
    reference concatenates all samples before torch.mean / torch.std; here the sums run sample by sample in fp64."""
    stats_file = os.path.join(data_root, f"{dataset_config.name}_norm_stats.pt")
    if os.path.exists(stats_file) and not getattr(dataset_config, "force_recompute_stats", False):
        stats = torch.load(stats_file, weights_only=True)   # the file holds tensors only
        out = {"mean": stats["mean"].to(dtype), "std": stats["std"].to(dtype)}
        if "c_mean" in stats and "c_std" in stats:
            out["c_mean"], out["c_std"] = stats["c_mean"].to(dtype), stats["c_std"].to(dtype)
        return out
    ds = VTKMeshDataset(root=dataset_config.base_path, order_file=order_file_path, dataset_config=dataset_config,
                        split="train", transform=RescalePosition())
    acc = {}
    for sample in ds:
        for key in ("x", "c"):
            v = getattr(sample, key, None)
            if v is None:
                continue
            v = v.double().reshape(v.shape[0], -1)
            a = acc.setdefault(key, [0, torch.zeros(v.shape[1], dtype=torch.float64), torch.zeros(v.shape[1], dtype=torch.float64)])
            a[0] += v.shape[0]
            a[1] += v.sum(0)
            a[2] += (v * v).sum(0)
    if "x" not in acc:
        raise ValueError("No data found in training set to calculate statistics.")

    def finish(a):
        n, s, s2 = a
        mean = s / n
        var = (s2 - n * mean * mean) / max(n - 1, 1)
        return mean.to(dtype), var.clamp(min=0).sqrt().to(dtype)
    out = {}
    out["mean"], out["std"] = finish(acc["x"])
    if "c" in acc:
        out["c_mean"], out["c_std"] = finish(acc["c"])
    os.makedirs(os.path.dirname(stats_file) or ".", exist_ok=True)
    torch.save(dict(out), stats_file)
    return out


# ---- loader: pinned staging, overlapped upload, device-resident cache ----------------------------------------------------
def _pin(batch: MeshBatch) -> MeshBatch:
    out = MeshBatch()
    for k, v in batch.__dict__.items():
        setattr(out, k, v.pin_memory() if (torch.is_tensor(v) and not v.is_cuda and torch.cuda.is_available()) else v)
    return out


class SampleLoader:
    """for batch in SampleLoader(dataset, batch_size, device): ...   (the reference uses a PyG DataLoader with
    ``Batch.from_data_list``; batch_size 1 in every shipped config, stat.py:366-419)

    * collation = ``MeshBatch.from_data_list`` with ``num_latent_nodes`` (the EnrichedData.__inc__ offsets);
    * every batch goes through pinned host memory and is uploaded on a side stream while the previous batch computes; the
      consumer's stream waits on the upload's event before the batch is handed over;
    * ``device_cache`` most recently used batches STAY on the device, with whatever the model cached on the batch object
      (``_gaot_graphs``: the row-sorted neighbour lists, and the GeoEmbed statistics attached to them) -- per-sample
      constants that later epochs then do not rebuild.  0 disables the cache."""

    def __init__(self, dataset, batch_size: int = 1, device="cuda", shuffle: bool = False, seed: int = 0, num_latent_nodes: int = 0,
                 device_cache: int = 0, drop_last: bool = False):
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.seed, self.epoch = shuffle, seed, 0
        self.num_latent_nodes = int(num_latent_nodes)
        self.device_cache = int(device_cache)
        self.drop_last = drop_last
        self._cache: "OrderedDict[tuple, MeshBatch]" = OrderedDict()
        self._stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _order(self) -> List[int]:
        idx = list(range(len(self.dataset)))
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(len(idx), generator=g).tolist()
        return idx

    def _host_batch(self, ids: Sequence[int]) -> MeshBatch:
        samples = [self.dataset[i] for i in ids]
        nl = self.num_latent_nodes or int(getattr(samples[0], "num_latent_nodes", 0) or 0)
        if len(samples) == 1:
            b = samples[0]
            b.num_graphs = 1
        else:
            b = MeshBatch.from_data_list(samples, nl)
        return _pin(b)

    def _upload(self, host: MeshBatch):
        if self._stream is None:
            return host.to(self.device), None
        with torch.cuda.stream(self._stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return dev, ev

    def __iter__(self) -> Iterator[MeshBatch]:
        order = self._order()
        groups = [tuple(order[i:i + self.batch_size]) for i in range(0, len(order), self.batch_size)]
        if self.drop_last and groups and len(groups[-1]) < self.batch_size:
            groups.pop()

        def fetch(key):
            if key in self._cache:
                self._cache.move_to_end(key)
                return self._cache[key], None, None
            host = self._host_batch(key)
            dev, ev = self._upload(host)
            return dev, ev, host          # the pinned source must outlive the asynchronous copy

        pending = fetch(groups[0]) if groups else None
        for gi, key in enumerate(groups):
            dev, ev, host = pending
            pending = fetch(groups[gi + 1]) if gi + 1 < len(groups) else None    # overlaps the consumer's step on `dev`
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                # the tensors were allocated on the upload stream's pool: tell the allocator that the consumer stream uses
                # them, or a dropped batch's blocks could be handed to the next upload while this step's kernels still run
                for v in dev.__dict__.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
            if self.device_cache > 0:
                self._cache[key] = dev
                self._cache.move_to_end(key)
                while len(self._cache) > self.device_cache:
                    self._cache.popitem(last=False)
            yield dev
            del host
