"""The teacher-forcing harness of scripts/psnr_windows.py, on the CPU at toy sizes: a window restarted from a state the oracle kept in the
middle of its run (parameters, both Adam moments, step counts, the generator position) reproduces the oracle's own continuation BIT FOR
BIT -- losses of every iteration and the held-out render -- so a difference a HIP window shows against it is the HIP path's, not the
harness's.  (The GPU twin of this check ran at full size: profiles/r06_psnr/selfcheck_null_1thread_seed1_at250.log.)"""
import os

import torch

import test_gpu_training_psnr as T


def test_window_from_a_kept_state_is_the_oracles_own_continuation(tmp_path):
    saved = (T.H, T.C_N, T.F_N, T.RAYS, T.ITERS, T.N_HELD, T.LR, T.SCHED, T.CHECKPOINTS, T.SAVE_EVERY)
    threads = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        T.H, T.C_N, T.F_N, T.RAYS, T.ITERS, T.N_HELD, T.SAVE_EVERY = 8, 8, 8, 16, 12, 1, 4
        T.LR = 1e-3
        T.SCHED = T.long_schedule(T.LR, T.ITERS, hold=0.5)
        T.CHECKPOINTS = (T.ITERS,)
        views = T.analytic_scene(4)
        path = str(tmp_path / "cpu_seed3.state")
        hist, held = T.run_oracle(views, 3, path, keep_all=True)
        assert len(hist) == 12 and sorted(f for f in os.listdir(tmp_path) if ".it" in f) == ["cpu_seed3.state.it%05d" % k for k in (0, 4, 8, 12)]
        for k in (0, 4, 8):
            st = torch.load(path + ".it%05d" % k, weights_only=False)
            nx = torch.load(path + ".it%05d" % (k + 4), weights_only=False)
            assert st["it"] == k and (k == 0) == (len(st["opt"]["state"]) == 0)            # (iteration 0 has no Adam moments yet)
            h, he = T.run_oracle(views, 3, init=st, stop=k + 4)
            assert h == hist[k:k + 4] == nx["hist"][k:k + 4]                                # the same losses, to the last bit
            assert he[0] == nx["held_at"]
        assert held == [torch.load(path + ".it00012", weights_only=False)["held_at"]]
        # a state of ANOTHER recipe is refused
        T.RAYS = 32
        try:
            T.run_oracle(views, 3, init=st, stop=12)
            raise SystemExit("a state of another recipe was accepted")
        except AssertionError:
            pass
    finally:
        T.H, T.C_N, T.F_N, T.RAYS, T.ITERS, T.N_HELD, T.LR, T.SCHED, T.CHECKPOINTS, T.SAVE_EVERY = saved
        torch.set_num_threads(threads)
