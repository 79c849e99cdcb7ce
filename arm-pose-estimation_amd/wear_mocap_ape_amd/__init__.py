"""wear_mocap_ape_amd -- MI355X (gfx950) implementation of the per-frame arm-pose inference
path of ``wear_mocap_ape``, behind the reference's own ``Estimator`` call surface.

Drop-in: replace ``from wear_mocap_ape.estimate.watch_phone_pocket_nn import WatchPhonePocketNN``
with ``from wear_mocap_ape_amd.estimate.watch_phone_pocket_nn import WatchPhonePocketNN`` (same
constructor kwargs, methods, array shapes and joint layout).  All arithmetic of the hot path
runs in hand-written HIP kernels behind the C ABI of ``include/ape_hip.h``; there is no CPU
fallback -- without ``libape_hip.so`` and a gfx950 device the compute entry points raise.
"""
__version__ = "0.1.0"
