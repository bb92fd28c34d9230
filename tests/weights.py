"""The closed-form parity weights live in the package (nerf_amd/synthetic_weights.py: bench.py and smoke() use them too and must not import
from the test tree); the tests and tests/golden/make_golden.py keep importing them under this name."""
from nerf_amd.synthetic_weights import *          # noqa: F401,F403
from nerf_amd.synthetic_weights import _hash_uniform, _layer, _state   # noqa: F401
