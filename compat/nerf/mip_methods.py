"""`nerf.mip_methods` served by the MI355X-native package: every name of nerf_amd.mip_methods."""
from nerf_amd.mip_methods import *          # noqa: F401,F403
import nerf_amd.mip_methods as _impl


def __getattr__(name):              # names a star import does not bind (leading underscore, late additions)
    return getattr(_impl, name)
