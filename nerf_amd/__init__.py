"""nerf_amd -- MI355X (gfx950) native NeRF ray-march hot path behind the call surface of
Enigmatisms/NeRF (``render_image`` / ``NeRF.render`` / ``ProposalNetwork`` / ``MipNeRF`` ...).

Compute lives in hand-written HIP kernels (``nerf_amd/csrc`` -> ``libnerf_amd.so``, C-ABI in
``include/nerf_amd.h``); this package is the thin host-side mirror of the reference's Python
interface.  Importing it without the built library raises: there is no CPU or torch fallback.
"""
from . import _lib                                   # noqa: F401  (fails loudly when the .so is missing)
from .ops import set_precision, current_precision, set_train_dumps    # noqa: F401

__all__ = ["set_precision", "current_precision", "set_train_dumps"]
