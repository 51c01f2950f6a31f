# -*- coding: utf-8 -*-
"""
xcontour_amd -- MI355X (gfx950) implementation of the contour-coordinate hot path of
miniufo/xcontour behind the reference's own API (reference xcontour/__init__.py:2-6).

The cell-touching work (min/max, weighted multi-channel histogram with in-kernel
|grad q|^2, A(Yeq) row sums, local wave activity, the fused Keff pipeline) runs in
hand-written HIP kernels reached through the C ABI of include/xcontour_hip.h.
There is no CPU fallback.
"""
from .core import Contour2D, Table
from .utils import equivalent_latitudes, latitude_lengths_at, cell_area, grad_metrics, \
    cartesian_metrics, Rearth
from .labeled import DataArray, Dataset
from .ncio import open_dataset
from .pipeline import KeffPlan, shard_slabs
from ._native import Context, default_context, XContourHipError

__version__ = "0.1.0"
