#!/usr/bin/env python3
"""Where does the HOST time of an eager training step go?  At the reference's default batch (1 024 rays) the eager step is launch-bound
(1.13 ms of kernels in a 1.49 ms iteration, profiles/r06_timeline_1024/mip_1024_eager.md): cProfile of N eager iterations of
scripts/gpu_train_rate.py's step, top functions by own time and by cumulative time.
    python scripts/gpu_host_profile.py [n_rays] [iters] [ref]"""
import cProfile
import importlib.util
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gpu_train_rate", os.path.join(ROOT, "scripts", "gpu_train_rate.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
is_ref = len(sys.argv) > 3
step = (mod.make_ref_step if is_ref else mod.make_step)(n_rays, 64, 128, "bf16")
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
print("unprofiled: %.3f ms / iteration (%d rays, %s)" % ((time.perf_counter() - t0) / iters * 1e3, n_rays, "Ref-NeRF" if is_ref else "MipNeRF"))
# the backward runs on autograd's device thread, which cProfile (main thread) only sees as time inside run_backward: single-threaded engine
# for the profiled pass, so that HipOp.backward and everything under it is attributed
ctx = torch.autograd.set_multithreading_enabled(False) if hasattr(torch.autograd, "set_multithreading_enabled") else None
pr = cProfile.Profile()
pr.enable()
for _ in range(iters):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    txt = s.getvalue()
    print("\n".join(txt.splitlines()[:60]))
print("(per-call figures are inflated by the profiler; the ranking is what matters; %d iterations)" % iters)
