"""gaot_3d_amd -- MI355X (gfx950) native forward/backward hot path of GAOT-3D behind the reference's
src/model operator API.  ``gaot_3d_amd.model.init_model`` is the drop-in factory; the kernels live in
``gaot_3d_amd/csrc`` and are reached through the C ABI declared in ``include/gaot3d_hip.h``."""
from .ops import get_precision, set_precision  # noqa: F401


def clear_graph_cache(batch):
    """Drop the per-batch neighbour lists (row-sorted edge lists) cached by the model on ``batch``: the next forward rebuilds
    them.  (The per-sample FACT that a decoder edge list is the encoder's with its rows swapped -- `_gaot_flip`, see
    `model/layers/integral_transform.graph_for` -- is about the inputs, not a list, and stays.)"""
    batch.__dict__.pop("_gaot_graphs", None)
