"""`nerf.utils` served by the MI355X-native package: every name of nerf_amd.utils."""
from nerf_amd.utils import *          # noqa: F401,F403
import nerf_amd.utils as _impl


def __getattr__(name):              # names a star import does not bind (leading underscore, late additions)
    return getattr(_impl, name)
