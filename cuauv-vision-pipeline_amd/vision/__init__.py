"""`vision` — host-side mirror of the reference package of the same name (ayf7/cuauv-vision-pipeline),
restricted to the per-frame detection hot path and its drop-in boundary.

Reference modules import `vision.core.base`, `vision.core.tuners`, `vision.utils.color`,
`vision.utils.transform`, `vision.utils.feature` (modules/red_buoy.py:3-8, modules/bins.py:5-7); the
same imports resolve here, and the arithmetic behind them runs as hand-written HIP kernels on an
MI355X through libvp.so (include/vp.h).  There is no CPU fallback: calling an operator without a
usable GPU raises `vision._vp.VpError`.
"""
__all__ = ["utils", "core"]
