"""``scatter`` at the reference's import path (src/model/layers/utils/scatter_native.py:4-54; the reference selects it
when torch_scatter is absent, integral_transform.py:9-24, geoembed.py:6-20).  Same signature, argument meaning and
errors; computed by the HIP segment-reduction kernels (gaot_3d_amd/edgeops.py: stable sort of the index, fixed-order
reduction, no atomics)."""
from ....edgeops import scatter

scatter_native = scatter

__all__ = ["scatter", "scatter_native"]
