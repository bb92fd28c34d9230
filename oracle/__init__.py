"""CPU oracle for the NeRF ray-march hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product package (``nerf_amd``) never imports it and has no CPU fallback.
"""
