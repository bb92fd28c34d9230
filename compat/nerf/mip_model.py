"""`nerf.mip_model` served by the MI355X-native package: every name of nerf_amd.mip_model."""
from nerf_amd.mip_model import *          # noqa: F401,F403
import nerf_amd.mip_model as _impl


def __getattr__(name):              # names a star import does not bind (leading underscore, late additions)
    return getattr(_impl, name)
