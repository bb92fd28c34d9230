"""`nerf.nerf_helper` served by the MI355X-native package: every name of nerf_amd.nerf_helper."""
from nerf_amd.nerf_helper import *          # noqa: F401,F403
import nerf_amd.nerf_helper as _impl


def __getattr__(name):              # names a star import does not bind (leading underscore, late additions)
    return getattr(_impl, name)
