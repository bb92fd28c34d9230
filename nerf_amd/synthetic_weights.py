"""Closed-form (integer-hash, libm-free) network weights: the synthetic, random-init-like parameters of bench.py, smoke() and the
parity tests (tests/weights.py re-exports this module).

Both the golden generator (which loads them into the *reference* modules) and the tests (which load
them into the oracle / the HIP path) call these, so the weights themselves never need storing.

Two flavours per network:
  * ``small`` -- |w| <= ~0.035 (std 0.02, like the reference's trunc-normal init, nerf_base.py:15-19),
                 zero-mean biases of std ~0.01;
  * ``he``    -- std sqrt(2/fan_in): activations stay O(1) through all ReLU layers, which makes the
                 high-frequency PE terms matter and stresses error amplification.
"""
import math

import numpy as np
import torch


def _hash_uniform(rows: int, cols: int, seed: int) -> np.ndarray:
    """Deterministic pseudo-uniform in [-0.5, 0.5): pure uint64 arithmetic (splitmix-style)."""
    i = np.arange(rows, dtype=np.uint64)[:, None]
    j = np.arange(cols, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        x = i * np.uint64(0x9E3779B97F4A7C15) + j * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(seed) * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(40)).astype(np.float64) / float(1 << 24) - 0.5)


def _layer(out_f: int, in_f: int, seed: int, flavour: str):
    u = _hash_uniform(out_f, in_f, seed)
    ub = _hash_uniform(1, out_f, seed + 7919)[0]
    if flavour == "small":
        w = u * (0.02 * math.sqrt(12.0))
        b = ub * (0.01 * math.sqrt(12.0))
    elif flavour == "he":
        w = u * (math.sqrt(2.0 / in_f) * math.sqrt(12.0))
        b = ub * 0.2
    else:
        raise ValueError(flavour)
    return torch.from_numpy(w.astype(np.float32)), torch.from_numpy(b.astype(np.float32))


def _state(shapes, flavour: str, base_seed: int):
    sd = {}
    for n, (name, out_f, in_f) in enumerate(shapes):
        w, b = _layer(out_f, in_f, base_seed + 101 * n, flavour)
        sd[name + ".weight"] = w
        sd[name + ".bias"] = b
    return sd


def proposal_state(flavour: str = "small", hidden: int = 256, L: int = 10):
    i = 6 * L + 3
    shapes = [("layers.0", hidden, i), ("layers.2", hidden, hidden), ("layers.4", hidden, hidden),
              ("layers.6", hidden, hidden), ("layers.8", 1, hidden)]
    return _state(shapes, flavour, 1000)


def mip_state(flavour: str = "small", hidden: int = 256, Lp: int = 10, Ld: int = 4):
    i = 6 * Lp + 3
    shapes = [("lin_block1.0", hidden, i), ("lin_block1.2", hidden, hidden), ("lin_block1.4", hidden, hidden),
              ("lin_block1.6", hidden, hidden), ("lin_block2.0", hidden, hidden + i), ("lin_block2.2", hidden, hidden),
              ("lin_block2.4", 256, hidden), ("bottle_neck.0", 256, 256), ("opacity_head.0", 1, 256),
              ("rgb_layer.0", 128, 256 + 6 * Ld + 3), ("rgb_layer.2", 3, 128)]
    return _state(shapes, flavour, 5000)


def ref_state(flavour: str = "small"):
    """RefNeRF(10, 4) in the reference's state_dict order (ref_model.py:31-62)."""
    i, hidden, out_dim, bottle = 63, 256, 256, 128
    din = 1 + bottle + 38
    shapes = [("spa_block1.0", hidden, i), ("spa_block1.2", hidden, hidden), ("spa_block1.4", hidden, hidden), ("spa_block1.6", hidden, hidden),
              ("spa_block2.0", hidden, hidden + i), ("spa_block2.2", hidden, hidden), ("spa_block2.4", hidden, hidden), ("spa_block2.6", out_dim, hidden),
              ("rho_tau_head", 2, out_dim), ("norm_col_tint_head", 9, out_dim), ("bottle_neck", bottle, out_dim), ("spec_rgb_head.0", 3, out_dim),
              ("dir_block1.0", hidden, din), ("dir_block1.2", hidden, hidden), ("dir_block1.4", hidden, hidden), ("dir_block1.6", hidden, hidden),
              ("dir_block2.0", hidden, hidden + din), ("dir_block2.2", hidden, hidden), ("dir_block2.4", out_dim, hidden), ("dir_block2.6", out_dim, hidden)]
    return _state(shapes, flavour, 9000)
