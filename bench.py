#!/usr/bin/env python3
"""Headline benchmark: rays/s of the NeRF ray-march hot path on MI355X (BASELINE.json metric).

One "step" = one full 800x800 image (640 000 rays, 64 proposal + 128 fine samples per ray) through
ray generation -> stratified sampling -> proposal MLP -> sigma->weights -> max-blur -> inverse-transform
sampling (sorted) -> fine MLP -> alpha compositing (rgb, depth, weights written), i.e. SURVEY.md section 8a
rows 1-10 -- BASELINE configs[1] ("Original NeRF, Lego 800x800, 64+128 samples, bf16, 1xMI355X").
Synthetic inputs (no dataset on the box): orbit pose pose_spherical(theta,-30,4), Lego's camera_angle_x,
near/far 2/6, white background, closed-form deterministic network weights; every uniform (stratified jitter, inverse-CDF
draws) is drawn INSIDE the kernels with Philox4x32-10 (the default, what the drop-in render_image does), or -- `--rng resident` --
pre-generated and resident in HBM before the timed region.

N GPUs: one process per GPU (torch.distributed, backend nccl = RCCL); every rank renders its own image
per step (weak scaling, rays are independent: no data-path collective); value = total rays / max-over-ranks time.
`python bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset) re-executes itself under torch.distributed.run
with N ranks on 127.0.0.1 (ddp_train.py:307-323 does the same with mp.spawn); started BY a launcher it insists that
WORLD_SIZE == N, and with the nccl backend that the box has N devices -- it never silently runs fewer ranks than asked.

--mode train-ddp: one step = every rank's 16 384-ray training step (train.py:164-199 body) + ONE flat all_reduce of the
744 069 gradient elements of both networks (ddp_train.py:98) + Adam; the collective is timed separately (bytes, us).

An N > 1 line is self-contained: besides the max-over-ranks `value` it carries every rank's own ms_per_step (min / max / list), what an
MFMA-only stream sustains on EVERY rank's GPU while all of them run it at once (`roofline.mfma_stream_ref`: eight GPUs share a chassis
power budget, so the clocks differ from a one-GPU box), and -- measured by rank 0 after the timed region while the other ranks wait at a
host-side (gloo) barrier -- `cpu_baseline`, `train_step` and `roofline.gemm_ref` exactly as in the N = 1 line.

Prints ONE JSON line on rank 0.
"""
import argparse
import datetime
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 800
C_COARSE, N_FINE = 64, 128
NEAR, FAR = 2.0, 6.0
MAC_PROP, MAC_FINE = 212_992, 527_872                      # per sample
FLOP_PER_RAY = 2 * (C_COARSE * MAC_PROP + N_FINE * MAC_FINE)   # 162.4e6
PEAK_BF16_DENSE = 2.5e15                                   # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_F32_MFMA = 157.3e12
# cpu_baseline.kind: the contract's two values are "reference" (oracle/_ref: the reference's own code built here) and "port" (the oracle).
# The reference is pure Python and cannot travel to the GPU box, so what is timed is the ORACLE -- said in full beside the code word:
KIND_DETAIL = ("oracle/nerf_oracle.py: this repo's torch-CPU fp32 restatement of the reference's render path (pinned bit-exact against goldens "
               "generated from the real reference, tests/golden/make_golden.py); NOT the reference's own source")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-train-rate", action="store_true", help="skip the training-step rate (train_step)")
    p.add_argument("--no-gemm-ref", action="store_true", help="skip the hipBLASLt GEMM reference measurement (roofline.gemm_ref)")
    p.add_argument("--fused", action="store_true",
                   help="fine MLP with the fused compositing epilogue (one launch for rows 8-10) instead of two launches; A/B switch")
    p.add_argument("--model", default="mip", choices=["mip", "ref"],
                   help="mip = BASELINE configs[1] (the headline); ref = Ref-NeRF (configs[3] shape: 64 + 192 merged samples): the render path, or with "
                        "--mode train-ddp the data-parallel training step with prop_normal")
    p.add_argument("--weights", default="small", choices=["small", "he", "zero"],
                   help="closed-form test weights; 'zero' is a power/DVFS diagnostic, never a reported number")
    p.add_argument("--cpu-rays", type=int, default=20000, help="rays of the same workload timed on the host cores (~14 s of CPU work)")
    p.add_argument("--rng", default="philox", choices=["philox", "resident"],
                   help="philox = every uniform drawn inside the kernels (no uniform tensor exists: what the drop-in render_image does by "
                        "default); resident = pre-drawn (N,64) + (N,129) uniform tensors resident in HBM before the timed region (round 1)")
    p.add_argument("--mode", default="render", choices=["render", "render-strong", "train-ddp"],
                   help="render = the headline (weak scaling: one image per rank and step); render-strong = ONE image per step split over the "
                        "ranks by rays + a final all_gather (strong scaling); train-ddp = per-rank training steps with the flat gradient "
                        "all_reduce timed separately")
    p.add_argument("--ipe", action="store_true", help="train-ddp: integrated PE in the fine pass (BASELINE configs[2])")
    p.add_argument("--contract", action="store_true", help="train-ddp: Mip-NeRF 360 scene contraction, unbounded near/far (BASELINE configs[4])")
    p.add_argument("--hipgraph", action="store_true", help="train-ddp: replay the step (collective included) from a hipGraph")
    p.add_argument("--train-dumps", default="bf16", choices=["bf16", "fp8"],
                   help="storage of the training dumps in bf16 mode: bf16, or e4m3 with per-sample-and-K-group scales (nerf_amd.set_train_dumps)")
    p.add_argument("--train-rays", type=int, default=16384, help="rays per rank and step in --mode train-ddp")
    p.add_argument("--prop-width", type=int, default=256, choices=[256, 128],
                   help="hidden width of the proposal network: 256 = BASELINE configs[1] (the headline); 128 = the reference's class default "
                        "(addtional.py:61, --prop_net_width 128) on the narrow-tile kernel -- a different, cheaper workload, labelled as such")
    p.add_argument("--fine-width", type=int, default=256, choices=[256, 128],
                   help="hidden width of the fine MipNeRF: 256 = the headline; 128 = --nerf_net_width 128 on the narrow-tile kernel (labelled)")
    p.add_argument("--dump-image", default=None, help="render-strong: rank 0 saves the last timed image (N,4) to this path (N-independence tests)")
    p.add_argument("--launch-check", action="store_true",
                   help="control-flow check of the N-rank launch path only (rendezvous, world size, barrier, max-over-ranks): no GPU work")
    return p.parse_args()


def committed_traffic(key: str):
    """-> (bytes per launch / step or None, {"round", "measured_by", "stale", ...}): the counter-measured HBM traffic committed in
    profiles/pmc_traffic.json (PMC collection needs the profiler around the process, so it is not measured in this run) together with the
    kernel-source hashes it was measured on -- `stale: true` when the tree's sources are not those any more (a kernel change must not
    silently keep an old traffic figure: VERDICT r5 item 7)."""
    import hashlib
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tfile):
        return None, None
    rec = json.load(open(tfile)).get(key)
    if not isinstance(rec, dict):
        return None, None
    changed = []
    for f, want in rec.get("sources", {}).items():
        path = os.path.join(ROOT, "nerf_amd", "csrc", f)
        have = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None
        if have != want:
            changed.append(f)
    return rec["bytes"], {"round": rec.get("round"), "measured_by": rec.get("measured_by"), "stale": bool(changed), "sources_changed_since": changed}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


METRIC_NAME = "rays/sec (64+128 samples), Lego 800x800"
AFTER_TIMED = "after the timed region (max over ranks, per-rank statistics, rank 0's solo measurements)"


def partial_record(msg: str) -> None:
    """One parsable line on stdout saying that no measurement exists and why (`value: null`).  EVERY rank may print one (tagged with its
    rank): the launcher tears the whole job down as soon as ONE rank exits, so a record that only rank 0 could write is lost whenever
    another rank's watchdog wins the race (tests/test_multiprocess.py failed exactly there in the full-suite order)."""
    line = json.dumps({"metric": METRIC_NAME, "value": None, "unit": "rays/s", "n_gpus": Watchdog.world, "rank": Watchdog.rank,
                       "error": msg[:1500], "phase": Watchdog.phase}) + "\n"
    # ONE write of the whole line (< PIPE_BUF: atomic on a pipe): several ranks write their records into the launcher's stdout at the same
    # moment, and print()'s separate writes of text and newline interleaved two records on one line (seen under load in the CPU suite)
    try:
        sys.stdout.flush()
        os.write(sys.stdout.fileno(), line.encode())
    except (OSError, ValueError, AttributeError):
        sys.stdout.write(line)
        sys.stdout.flush()
    Watchdog.done = True                                     # one record per rank: the termination watcher does not add a second one


class Watchdog:
    """A stuck N > 1 run must fail LOUDLY inside the driver's slot instead of hanging until it is killed: when the
    deadline passes, every rank writes one line saying where it was (`PHASE`), dumps every thread's Python stack to stderr and prints a
    rank-tagged partial JSON line (`value: null`, `error`, `phase`, `rank`) so that the driver's log holds a parsable record whichever rank
    fires first; ranks other than 0 then wait GRACE seconds before exiting (rank 0's own watchdog, armed at the same deadline, gets its
    line out before the launcher's tear-down), and the process exits non-zero.  Armed by default when N > 1: 900 s up to the end of the
    timed region, re-armed for the phase in which rank 0 measures alone so that the whole run stays inside 1 620 s of the driver's 1 800 s
    slot (BENCH_DUMP_STACKS_AFTER overrides the first limit; 0 = off); the preflight arms its own short one."""
    phase = "start"
    rank = 0
    world = 1
    done = False                 # the final record is out: a termination after this point needs no partial line
    run = None                   # the whole-run watchdog (re-armed per phase)
    t0 = time.monotonic()
    GRACE = float(os.environ.get("BENCH_WATCHDOG_GRACE", "15"))
    TOTAL = 1620.0

    def __init__(self, seconds: float, what: str, code: int = 124):
        import threading
        self.what, self.code = what, code
        self.timer = threading.Timer(seconds, self.fire, args=(seconds,))
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        self.timer.cancel()

    def fire(self, seconds):
        import faulthandler
        msg = "bench.py: %s did not finish within %.0f s (rank %d of %d, phase: %s)" % (self.what, seconds, Watchdog.rank, Watchdog.world, Watchdog.phase)
        sys.stderr.write(msg + "\n")
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        sys.stderr.flush()
        if not Watchdog.done:
            partial_record(msg)
        if Watchdog.rank != 0:
            time.sleep(Watchdog.GRACE)
        os._exit(self.code)

    @classmethod
    def arm_run(cls, seconds: float, what: str = "the run"):
        if cls.run is not None:
            cls.run.cancel()
        cls.run = cls(seconds, what) if seconds > 0 else None

    @classmethod
    def rearm_after_timed_region(cls):
        """The phase in which the other ranks idle at the host-side barrier while rank 0 measures cpu_baseline / train_step alone (the 25-minute
        gloo group): a healthy run may spend longer here than the first limit allows, so the deadline becomes "the whole run inside
        TOTAL seconds" (the 900 s default used to pre-empt the 25-minute control-group timeout)."""
        if cls.run is not None:
            left = max(60.0, cls.TOTAL - (time.monotonic() - cls.t0))
            cls.arm_run(left, "the run (deadline of the whole run: %.0f s)" % cls.TOTAL)


def set_phase(name: str) -> None:
    Watchdog.phase = name
    if name == AFTER_TIMED:
        Watchdog.rearm_after_timed_region()


def install_termination_record() -> None:
    """A rank the launcher terminates (SIGTERM: another rank died first -- rank 0 included) still leaves a parsable record.  The C-level
    handler feeds a wake-up pipe at once, even while the main thread sits inside a collective or a device synchronisation (a Python-level
    handler alone would wait for that call to return); a watcher thread prints the rank-tagged partial line and exits 143."""
    import signal
    import threading
    r, w = os.pipe()
    os.set_blocking(w, False)
    signal.signal(signal.SIGTERM, lambda *_: None)
    signal.set_wakeup_fd(w, warn_on_full_buffer=False)

    def watch():
        while True:
            b = os.read(r, 1)
            if b and b[0] == signal.SIGTERM:
                break
        if not Watchdog.done:
            partial_record("bench.py: rank %d of %d was terminated by the launcher (SIGTERM: another rank exited first), phase: %s"
                           % (Watchdog.rank, Watchdog.world, Watchdog.phase))
        sys.stderr.flush()
        os._exit(143)
    threading.Thread(target=watch, daemon=True, name="termination-record").start()


def device_identity(dev):
    """what this rank runs on: (index, name, PCI bus id or uuid) -- two ranks reporting the same identity share a GPU"""
    if dev.type != "cuda":                                   # (--launch-check: the control flow on host tensors)
        return {"index": None, "name": "cpu", "bus": None, "cus": 0, "gb": 0.0}
    pr = torch.cuda.get_device_properties(dev)
    bus = None
    if hasattr(pr, "pci_bus_id"):
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    elif hasattr(pr, "uuid"):
        bus = str(pr.uuid)
    return {"index": dev.index, "name": pr.name, "bus": bus, "cus": pr.multi_processor_count, "gb": round(pr.total_memory / 2 ** 30, 1)}


def preflight(dist, comm, world, rank, backend, dev, limit_s: float = 120.0):
    """Before anything else of an N > 1 run: prove in seconds that the fabric works, or exit non-zero with ONE line
    saying what does not.  (i) every rank's device identity, gathered over the host-side group: with RCCL two ranks on one device is an
    error; (ii) one 3 MB all_reduce (the gradient buffer of --mode train-ddp: 744 069 fp32) and (iii) one 10 MB all_gather (rgb + depth of
    an 800 x 800 image, the gather of --mode render-strong) over the DATA-PATH backend, results checked element-exact, each timed after
    one untimed warm-up call; every rank prints its line to stderr.  The whole preflight runs under its own watchdog (`limit_s`)."""
    set_phase("preflight")
    wd = Watchdog(limit_s, "the preflight (process-group collectives across %d ranks)" % world, code=125)
    ident = device_identity(dev)
    idents = comm.gather(ident)
    if backend == "nccl":
        seen = {}
        for r, d in enumerate(idents):
            key = d["bus"] if d["bus"] is not None else d["index"]
            if key in seen:
                sys.exit("bench.py preflight: ranks %d and %d both run on device %s (%s) -- one process per GPU" % (seen[key], r, key, d["name"]))
            seen[key] = r
    on = dev if backend == "nccl" else torch.device("cpu")                  # (gloo: the one-GPU-box smoke path, host tensors)
    if os.environ.get("BENCH_TEST_HANG") == "preflight" and rank == world - 1:
        time.sleep(3600)                                                      # (tests: a rank that never reaches the collective)
    if os.environ.get("BENCH_TEST_HANG") == "mismatch" and rank == world - 1:
        rank_value_bug = 1.0                                                  # (tests: a fabric that returns wrong sums)
    else:
        rank_value_bug = 0.0
    n_red = 744_069
    x = torch.full((n_red,), float(rank + 1), dtype=torch.float32, device=on)
    n_gat = (640_000 * 4 + world - 1) // world                                # floats per rank: 16 B/ray of the image, split over the ranks
    y = torch.full((n_gat,), float(rank), dtype=torch.float32, device=on)
    outs = [torch.empty_like(y) for _ in range(world)]

    def timed(fn):
        fn()                                                                  # warm-up: communicator / ring set-up
        if on.type == "cuda":
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        fn()
        if on.type == "cuda":
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e6

    def red():
        x.fill_(float(rank + 1) + rank_value_bug)
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
    us_red = timed(red)
    want = world * (world + 1) / 2.0
    if not bool((x == want).all()):
        sys.exit("bench.py preflight: rank %d: all_reduce(SUM) of %d floats over %s returned %r, expected %r" % (rank, n_red, backend, float(x[0]), want))
    us_gat = timed(lambda: dist.all_gather(outs, y))
    for r, o in enumerate(outs):
        if not bool((o == float(r)).all()):
            sys.exit("bench.py preflight: rank %d: all_gather over %s: the chunk of rank %d holds %r" % (rank, backend, r, float(o[0])))
    sys.stderr.write("bench.py preflight: rank %d/%d on cuda:%s %s [%s] %d CUs %.0f GB | %s all_reduce %.2f MB %.0f us | all_gather %.2f MB %.0f us\n"
                     % (rank, world, ident["index"], ident["name"], ident["bus"], ident["cus"], ident["gb"], backend, n_red * 4 / 1e6, us_red,
                        n_gat * 4 * world / 1e6, us_gat))
    sys.stderr.flush()
    rows = comm.gather({"rank": rank, "device": ident, "allreduce_us": us_red, "allgather_us": us_gat})
    wd.cancel()
    set_phase("setup")
    return {"backend": backend, "allreduce_bytes": n_red * 4, "allgather_bytes": n_gat * 4 * world, "ranks": rows}


def launch_or_verify(a):
    """`--gpus N` is a promise about the number of ranks.  Under a launcher: WORLD_SIZE must equal N.  Stand-alone with N > 1:
    become the launcher (one process per GPU, rendezvous on 127.0.0.1) and exit with the job's status."""
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != a.gpus:
            sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    if a.gpus > 1 and backend == "nccl" and not a.launch_check and torch.cuda.device_count() < a.gpus:
        sys.exit("bench.py: --gpus %d needs %d visible devices, this box has %d (one process per GPU; refusing to run fewer ranks)"
                 % (a.gpus, a.gpus, torch.cuda.device_count()))
    if "WORLD_SIZE" in os.environ or a.gpus == 1:
        return
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout passes through this process: if the job dies without ANY record (a rank killed outright, the launcher itself
    # failing) the launcher writes the partial line, so that `python bench.py --gpus N` never ends non-zero without a parsable line
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1)
    seen = False
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{"):
            try:
                rec = json.loads(line)
                seen = seen or "value" in rec or "launch_check" in rec
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0 and not seen:
        Watchdog.world, Watchdog.phase = a.gpus, "unknown (no rank left a record)"
        partial_record("bench.py: the %d-rank job exited with status %d and no rank printed a record (see stderr)" % (a.gpus, rc))
    sys.exit(rc)


def launch_check(a, world, rank):
    """Everything the N-rank path does around the GPU work, on CPU tensors (the world-size-2 gloo test of tests/test_multiprocess.py):
    process group, the Comm helper of the real modes -- barriers, max-over-ranks, the per-rank gathers of a self-contained N > 1 line,
    the wait of the other ranks while rank 0 measures alone -- and the one JSON line from rank 0."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
    assert dist.get_world_size() == a.gpus
    comm = Comm(dist, world, rank, "gloo", torch.device("cpu"))
    pre = preflight(dist, comm, world, rank, "gloo", torch.device("cpu"), limit_s=float(os.environ.get("BENCH_PREFLIGHT_LIMIT", "120")))
    set_phase("timed region")
    if os.environ.get("BENCH_TEST_HANG") == "run" and rank == world - 1:
        time.sleep(3600)                                     # (tests: a rank stuck in the run -- the default watchdog must end it loudly)
    if os.environ.get("BENCH_TEST_HANG") == "rank0-dies" and rank == 0:
        import signal
        os.kill(os.getpid(), signal.SIGKILL)                 # (tests: rank 0 dies FIRST and silently -- the survivors must leave the record)
    if os.environ.get("BENCH_TEST_HANG") == "all-die":
        import signal
        os.kill(os.getpid(), signal.SIGKILL)                 # (tests: nobody is left to write a record -- the launching bench.py does)
    comm.dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))                            # "the timed region": rank r takes (r + 1) x 10 ms
    local_dt = time.perf_counter() - t0
    set_phase(AFTER_TIMED)
    dt = comm.max_over_ranks(local_dt)
    spread = rank_spread(comm, local_dt, 1)
    streams = comm.gather({"rank": rank, "tflops": 1000.0 + rank})
    if rank == 0:
        time.sleep(0.2)                                      # rank 0's extras (cpu_baseline, train_step): the others wait in comm.finish()
        print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "max_over_ranks": float(world),
                          "ms_per_step": dt * 1e3, "ms_per_step_ranks": spread, "per_rank": streams, "preflight": pre}), flush=True)
        Watchdog.done = True
    comm.finish()


class Comm:
    """The bench's view of the process group: the data-path backend (nccl = RCCL; gloo only to smoke-test the N > 1 control flow on a
    one-GPU box) for the timed region's barriers, and a host-side gloo group for everything after it -- per-rank statistics, and the
    barrier the other ranks wait at while rank 0 measures the CPU baseline (a wait on the host: their GPUs stay idle instead of spinning
    in a collective kernel)."""

    def __init__(self, dist, world, rank, backend, dev):
        self.dist, self.world, self.rank, self.backend, self.dev = dist, world, rank, backend, dev
        self.ctl = None
        self.preflight = None
        if dist is not None and backend != "gloo":
            # the ONLY long wait of an N > 1 run: the ranks idle here while rank 0 measures cpu_baseline / train_step alone after the timed
            # region (a few minutes); 25 minutes keeps even that inside the driver's 1 800 s slot
            self.ctl = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=25))

    def sync(self):
        """the contract's bracket of the timed region: barrier over the data-path backend + device synchronisation"""
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.ctl)

    def max_over_ranks(self, dt: float) -> float:
        if self.dist is None:
            return dt
        t = torch.tensor([dt], device=self.dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """-> [rank 0's obj, rank 1's obj, ...] on every rank"""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.ctl)
        return out

    def finish(self):
        if self.dist is not None:
            self.barrier()
            Watchdog.done = True                             # rank 0's record is out (it printed before this barrier)
            self.dist.destroy_process_group()


def rank_spread(comm, local_dt, steps):
    ms = comm.gather(local_dt / steps * 1e3)
    return {"min": min(ms), "max": max(ms), "all": ms,
            "note": "every rank's own wall time per step between the barriers of the timed region; `ms_per_step` / `value` use the max"}


def cpu_baseline_train(n_rays: int = 512, steps: int = 3):
    """The reference's training step on the host cores (the oracle's restatement of train.py:164-199 under torch autograd + torch.optim.Adam,
    fp32), `steps` steps of `n_rays` rays at 64 + 128 samples after one warm-up step."""
    import torch.nn.functional as F
    from nerf_amd import synthetic_weights as Wt
    from oracle import nerf_oracle as O
    torch.manual_seed(0)
    prop = {k: v.clone().requires_grad_(True) for k, v in Wt.proposal_state("small").items()}
    mip = {k: v.clone().requires_grad_(True) for k, v in Wt.mip_state("small").items()}
    opt = torch.optim.Adam(list(mip.values()) + list(prop.values()), lr=1e-4)
    d = F.normalize(torch.randn(n_rays, 3) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)
    rays = torch.cat((torch.tensor([0.0, 0.0, 4.0]).expand(n_rays, 3), d), -1)
    tgt = torch.rand(n_rays, 3)
    res = (FAR - NEAR) / C_COARSE
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)

    def one():
        z_c = torch.linspace(NEAR, FAR - res, C_COARSE) + torch.rand((n_rays, C_COARSE)) * res
        pts = rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]
        pw = O.max_blur(O.sigma_to_weights(F.softplus(O.proposal_forward(prop, pts)), z_c, rays[:, 3:]), 0.01)
        z_f, below = O.inverse_sample(pw, z_c, torch.rand((n_rays, N_FINE + 1)), sort=True)
        z_f = z_f[..., :-1]
        rend, wts, _ = O.composite(O.mip_forward(mip, O.length2pts(rays, z_f)), z_f, rays[:, 3:], white_bkg=True)
        loss = O.proposal_loss(O.get_bounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step()

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": steps * n_rays / dt, "unit": "rays/s", "cores": cores, "kind": "port", "kind_detail": KIND_DETAIL,
            "sample": "%d training steps of %d rays (64+128 samples; oracle forward, torch autograd backward, torch.optim.Adam), torch CPU fp32, %d threads on a %d-CPU host, %.1f s"
                      % (steps, n_rays, cores, os.cpu_count() or 1, dt)}


def cpu_baseline(n_rays: int):
    """The reference CPU path (= the oracle, proven equal to the reference by tests/test_oracle_golden.py),
    fp32, all host cores, on the first `n_rays` rays of the same image; one 2500-ray tile at a time like
    render_image does."""
    from nerf_amd import synthetic_weights as Wt
    from oracle import nerf_oracle as O
    prop_sd, mip_sd = Wt.proposal_state("small"), Wt.mip_state("small")
    pose = O.pose_spherical(30.0, -30.0, 4.0)[:3]
    dirs = O.ray_dirs_image(pose, H, W, O.fov2focal(0.6911112070083618, (H, W))).reshape(-1, 3)
    g = torch.Generator().manual_seed(0)
    tile = 2500
    n_tiles = max(1, n_rays // tile)

    def one(t):
        rays = torch.cat((pose[:, -1].expand(tile, -1), dirs[t * tile:(t + 1) * tile]), -1)
        u1, u2 = torch.rand(tile, 64, generator=g), torch.rand(tile, N_FINE + 1, generator=g)
        with torch.no_grad():
            O.render_rays(prop_sd, mip_sd, rays, u1, u2, NEAR, FAR, N_FINE, white_bkg=True)

    # torch's CPU GEMMs stop scaling long before 256 threads on these layer sizes: probe a few thread counts on
    # one tile each and keep the fastest (the baseline should be the CPU path at its best, not oversubscribed)
    ncpu = os.cpu_count() or 1
    best = None
    for th in sorted({min(ncpu, c) for c in (16, 32, 64, 128)}):
        torch.set_num_threads(th)
        one(0)
        t0 = time.perf_counter()
        one(0)
        d = time.perf_counter() - t0
        if best is None or d < best[1]:
            best = (th, d)
    cores = best[0]
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    for t in range(n_tiles):
        one(t)
    dt = time.perf_counter() - t0
    return {"value": n_tiles * tile / dt, "unit": "rays/s", "cores": cores, "kind": "port", "kind_detail": KIND_DETAIL,
            "sample": "%d rays (%d tiles of 2500) of the same 800x800, 64+128 workload, torch CPU fp32, %d threads (best of 16/32/64/128 on a %d-CPU host), %.1f s"
                      % (n_tiles * tile, n_tiles, cores, ncpu, dt)}


def gemm_reference(dev):
    """the nominal 2.5 PFLOP/s next to what the library GEMM (hipBLASLt through torch.matmul) sustains on this box on
    random bf16 data, after the timed region: (i) a large square GEMM = the power-limited practical ceiling of the MFMA pipe,
    (ii) one 256-wide NeRF layer applied to a full launch's samples UNFUSED (activations read from / written to HBM) = what a
    layer-by-layer implementation of the hot path gets."""
    out = {}
    for name, (m, k, n) in (("square_8192", (8192, 8192, 8192)), ("layer_256_unfused", (1 << 23, 256, 256))):
        x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
        w = torch.randn(k, n, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            y = x @ w
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        s.record()
        for _ in range(reps):
            y = x @ w
        e.record()
        torch.cuda.synchronize()
        out[name + "_tflops"] = 2.0 * m * k * n * reps / (s.elapsed_time(e) * 1e-3) / 1e12
        del x, w, y
    out["note"] = "torch.matmul (hipBLASLt) bf16 on random data, measured on this box after the timed region"
    return out


MFMA_STREAM_MODES = (("constant_operands", 0), ("random_operands", 1), ("weights_x_relu_activations", 2), ("weights_x_relu_activations_a_from_lds", 3))


def mfma_stream_measure(dev):
    """What streams of nothing but v_mfma_f32_32x32x16_bf16 (one wave per SIMD, every CU) sustain on THIS GPU under its power limit, measured
    after the timed region: with constant operands (the datapath does not toggle: an optimistic ceiling), with
    operands that change on every MFMA (pseudo-random bf16: pessimistic), with network-like data (random weights x post-ReLU-like
    activations, half zeros) and with that data arriving through the weight ring's ds_read_b128 cadence.  TFLOP/s per mode."""
    from nerf_amd import ops
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    out = {}
    for name, mode in MFMA_STREAM_MODES:
        ops.mfma_stream(2000, n_cu, dev, mode)
        torch.cuda.synchronize()
        iters = 40000                                        # ~40 ms: long enough for the power governor to settle
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.mfma_stream(iters, n_cu, dev, mode)
        e.record()
        torch.cuda.synchronize()
        out[name] = n_cu * 4 * iters * 64 * 32768.0 / (s.elapsed_time(e) * 1e-3) / 1e12
    return out


def mfma_stream_reference(comm, dev, achieved_tflops, executed_tflops):
    """roofline.mfma_stream_ref: every rank measures its own GPU AT THE SAME TIME (the GPUs of a node share a chassis power budget), rank 0
    reports its own figures in the keys of the one-GPU line and every rank's in `per_rank`."""
    comm.barrier()
    mine = mfma_stream_measure(dev)
    per_rank = comm.gather(mine)
    if comm.rank != 0:
        return None
    tf, tf_data = mine["constant_operands"], mine["weights_x_relu_activations"]
    return {"tflops": tf, "frac_of_datasheet_peak": tf * 1e12 / PEAK_BF16_DENSE, "kernel_frac_of_this": achieved_tflops / tf,
            "kernel_executed_frac_of_this": executed_tflops / tf,
            "data_realistic_tflops": tf_data, "data_realistic_frac_of_datasheet_peak": tf_data * 1e12 / PEAK_BF16_DENSE,
            "kernel_frac_of_data_realistic": achieved_tflops / tf_data, "kernel_executed_frac_of_data_realistic": executed_tflops / tf_data,
            "streams_tflops": mine, "per_rank": per_rank,
            "note": "MFMA-only streams (nerf_amd_mfma_stream, modes 0-3) on this box after the timed region, all ranks at once: `tflops` = constant "
                    "operands (optimistic: nothing toggles), `data_realistic_tflops` = pseudo-random weights x post-ReLU-like activations "
                    "(the ceiling a kernel multiplying real data can be held against); 64 MFMAs per iteration and wave, 40 ms per stream"}


def _spread(xs):
    xs = sorted(xs)
    return {"min": xs[0], "median": statistics.median(xs), "max": xs[-1], "n": len(xs)}


def timed_steps(step, iters, warm):
    """Run `step` warm + iters times without a host synchronisation in between and time it three ways:
      wall  total host time / iters between two device synchronisations (what a training loop sees);
      gpu   per-iteration time between HIP events recorded on the compute stream at the iteration boundaries (min / median / max:
            an outlier iteration -- an allocation, a clock dip -- is visible instead of averaged in);
      alloc what torch's caching allocator did meanwhile (hipMalloc'ed segments, retries), and the library's persistent buffers.
    wall - gpu median = host-side cost the device does not hide; gpu max - min = jitter."""
    from nerf_amd import ops
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    ar0 = dict(ops.ARENA_STATS)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(iters):
        step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    st1 = torch.cuda.memory_stats()
    gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(iters)]
    alloc = {"hipMalloc_segments": st1.get("segment.all.allocated", 0) - st0.get("segment.all.allocated", 0),
             "block_allocations": st1.get("allocation.all.allocated", 0) - st0.get("allocation.all.allocated", 0),
             "alloc_retries": st1.get("num_alloc_retries", 0) - st0.get("num_alloc_retries", 0),
             "reserved_gb": st1.get("reserved_bytes.all.current", 0) / 1e9,
             "persistent_buffer_uses": ops.ARENA_STATS["persistent"] - ar0["persistent"], "fresh_buffer_uses": ops.ARENA_STATS["fresh"] - ar0["fresh"]}
    return wall, gpu, alloc


def kernel_sum_ms(step):
    """Sum of the device kernels' own durations over ONE iteration (torch.profiler's kernel records, i.e. roctracer inside the process --
    no rocprof around it): what the step costs with every launch gap removed.  None when the profiler is unavailable on the box."""
    try:
        from torch.profiler import ProfilerActivity, profile
        step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            step()
            torch.cuda.synchronize()
        tot, n = 0.0, 0
        for ev in prof.events():
            if str(getattr(ev, "device_type", "")).endswith("CUDA") and getattr(ev, "device_time_total", 0) > 0:
                tot += ev.device_time_total
                n += 1
        return {"ms": tot / 1e3, "kernels": n} if n else None
    except Exception as ex:                                   # noqa: BLE001  (a measurement aid must not take the bench line down)
        return {"ms": None, "error": repr(ex)[:200]}


def train_rate(precision):
    """Second figure of SURVEY 8d: rays/s of a whole training step (train.py:164-199 body + Adam: HIP training forward with activation
    dump, fused dgrad chain, MFMA weight gradients, one-launch Adam -- no library GEMM anywhere) on synthetic rays, measured after the
    timed region; never part of `value`.  Its roofline: 3 x the forward's algorithmic flops (forward + dgrad + wgrad) against the dense
    bf16 MFMA peak.  Every figure is the MEDIAN of >= 30 iterations with its min / max, host wall clock beside the device time, and the
    allocator's activity -- round 3's line was one mean of ten iterations and hid a 4 ms gap between the driver's box and the kernels' sum."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_train_rate", os.path.join(ROOT, "scripts", "gpu_train_rate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {"unit": "rays/s (fwd + bwd + Adam step, 64+128 samples)", "precision": precision,
           "timing": "median of the per-iteration HIP-event times of a free-running loop (no host synchronisation inside); wall = host clock over the same loop"}
    peak = PEAK_BF16_DENSE if precision == "bf16" else PEAK_F32_MFMA

    def entry(n, step, iters, warm, profile_kernels=False):
        wall, gpu, alloc = timed_steps(step, iters, warm)
        sp = _spread(gpu)
        e = {"rays_per_s": n / (sp["median"] * 1e-3), "ms_per_iter": sp["median"], "ms_min": sp["min"], "ms_max": sp["max"], "iters": iters,
             "wall_ms_per_iter": wall * 1e3, "allocator": alloc}
        if profile_kernels:
            e["kernel_sum"] = kernel_sum_ms(step)
        return e

    for n, iters in ((512, 60), (4096, 40), (16384, 30)):
        out["rays_%d" % n] = entry(n, mod.make_step(n, 64, 128, precision), iters, 5, profile_kernels=(n == 16384))
    out["rays_512_hipgraph"] = entry(512, mod.make_step(512, 64, 128, precision, graph=True), 100, 5)
    # the reference's DEFAULT training batch (--sample_ray_num 1024, procedures.py:170), eager and replayed from a hipGraph; `kernel_sum`
    # = the kernels' own durations with every launch gap removed + the number of launches of one step
    out["rays_1024"] = entry(1024, mod.make_step(1024, 64, 128, precision), 60, 5, profile_kernels=True)
    out["rays_1024_hipgraph"] = entry(1024, mod.make_step(1024, 64, 128, precision, graph=True), 100, 5)
    for k in ("rays_1024", "rays_1024_hipgraph"):
        out[k]["roofline_frac"] = out[k]["rays_per_s"] * 3 * FLOP_PER_RAY / peak
    best = out["rays_16384"]["rays_per_s"]
    out["roofline"] = {"bound": "mfma", "kernel": "whole training step, 16384 rays (3 x 162.4 MFLOP/ray)", "achieved": best * 3 * FLOP_PER_RAY / 1e12,
                       "peak": peak / 1e12, "unit": "TFLOP/s", "frac": best * 3 * FLOP_PER_RAY / peak, "traffic": None}
    # measured traffic of one step from the committed PMC passes, and the bandwidth it implies at this run's step time, next to the
    # algorithmic floor of the dump design (2 KiB per sample and hidden layer)
    step_bytes, step_src = committed_traffic("train_step_16384_%s" % precision)
    floor_bytes = 16384 * (N_FINE * 8 + C_COARSE * 4) * 2048.0
    out["roofline"]["traffic"] = step_bytes
    out["roofline"]["traffic_source"] = step_src
    out["roofline"]["traffic_stale"] = step_src["stale"] if step_src else None
    out["roofline"]["hbm"] = {"algorithmic_bytes": floor_bytes, "measured_bytes": step_bytes, "peak_tbps": 8.0,
                              "achieved_tbps": (step_bytes * best / 16384 / 1e12) if step_bytes else None,
                              "algorithmic_tbps": floor_bytes * best / 16384 / 1e12}
    out["iteration_512"] = iteration_rate(precision)
    out["iteration_1024"] = iteration_rate(precision, n_rays=1024)      # the reference's default --sample_ray_num through TrainStep (fused loss kernels)
    # the product's own iteration at the large batch: device-side sampler, fused loss kernels, one-launch Adam (the `rays_16384` /
    # `refnerf_rays_16384` figures above run the reference-style body of scripts/gpu_train_rate.py with its torch expressions)
    out["iteration_16384"] = iteration_rate(precision, n_rays=16384, iters=30)
    out["refnerf_iteration_16384"] = iteration_rate(precision, n_rays=16384, iters=10, ref=True)
    for k, fl in (("iteration_16384", 3 * FLOP_PER_RAY), ("refnerf_iteration_16384", 3 * 2 * (C_COARSE * MAC_PROP + (N_FINE + C_COARSE) * 1_071_616))):
        for m in ("eager", "hipgraph"):
            out[k][m]["roofline_frac"] = out[k][m]["rays_per_s"] * fl / peak
    # Ref-NeRF (BASELINE configs[3]) with prop_normal: the reference's batch and the paper's 2^14-ray batch, 64 + (128 + 64 merged) samples
    ref_flop_per_ray = 2 * (C_COARSE * MAC_PROP + (N_FINE + C_COARSE) * 1_071_616)
    for n, iters in ((512, 40), (16384, 10)):
        e = entry(n, mod.make_ref_step(n, 64, 128, precision), iters, 3)
        e["roofline_frac"] = e["rays_per_s"] * 3 * ref_flop_per_ray / peak
        e["note"] = "Ref-NeRF step with prop_normal (train.py:176-187); roofline_frac = 3 x %.1f MFLOP/ray (forward + dgrad + wgrad; the two density-gradient chains not counted) / MFMA peak" % (ref_flop_per_ray / 1e6)
        out["refnerf_rays_%d" % n] = e
    for graph in (False, True):                            # the reference's default batch on the Ref-NeRF branch
        step = mod.make_ref_step(1024, 64, 128, precision, graph=graph)
        e = entry(1024, step, 60 if graph else 30, 3, profile_kernels=not graph)
        e["roofline_frac"] = e["rays_per_s"] * 3 * ref_flop_per_ray / peak
        out["refnerf_rays_1024" + ("_hipgraph" if graph else "")] = e
    import nerf_amd
    nerf_amd.set_precision(precision)
    return out


def iteration_rate(precision, n_rays=512, iters=200, ref=False):
    """The reference's whole training ITERATION (train.py:151-218: ray sampling from an 800x800 image included) through
    nerf_amd.training.TrainStep -- pose, pixel table, seed, Adam step count and learning rate in device memory, every random number drawn
    in kernels -- eager and replayed from a hipGraph, with the learning rate rewritten on the host every iteration like DecayLrScheduler."""
    import nerf_amd
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.optim import Adam
    from nerf_amd.training import TrainStep
    from oracle import nerf_oracle as O                    # (pose / focal helpers only; nothing timed)
    nerf_amd.set_precision(precision)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)
    out = {}
    img = torch.rand(3, H, W, device=dev)
    pose = O.pose_spherical(30.0, -30.0, 4.0)[:3].contiguous().to(dev)
    focal = O.fov2focal(0.6911112070083618, (H, W))
    for mode in ("eager", "hipgraph"):
        if ref:                                              # BASELINE configs[3]: the Ref-NeRF branch with prop_normal (train.py:176-187)
            from nerf_amd.ref_model import RefNeRF
            prop, mip = ProposalNetwork(10, 256).to(dev).train(), RefNeRF(10, 4).to(dev).train()
        else:
            prop, mip = ProposalNetwork(10, 256).to(dev).train(), MipNeRF(10, 4, 256).to(dev).train()
        opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=5e-4, lr_on_device=True)
        # (gradients in TrainStep's own flat buffer, WITHOUT the data-parallel all_reduce: in an N > 1 run only rank 0 measures this, and a
        #  collective that one rank enters alone never returns -- the two-rank GPU test of round 4 found exactly that hang)
        step = TrainStep(prop, mip, opt, (H, W), focal, 2.0, 6.0, ray_num=n_rays, coarse_pnum=C_COARSE, fine_pnum=N_FINE, seed=11, prop_normal=ref)
        step.set_image(img, pose)
        if mode == "hipgraph":
            step.capture(warmup=3)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            for gr in opt.param_groups:
                gr["lr"] = 5e-4 * (0.999 ** i)
            step(img, pose)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[mode] = {"rays_per_s": n_rays / dt, "ms_per_iter": dt * 1e3}
    out["note"] = "nerf_amd.training.TrainStep: sampler + forward + backward + Adam, image/pose copied in and lr rewritten every iteration"
    return out


def train_ddp(a, comm):
    """--mode train-ddp: what ddp_train.py's inner loop does per iteration (ddp_train.py:66-68,98: forward, backward, gradient
    all-reduce, optimizer step), one rank per GPU, `--train-rays` rays per rank (weak scaling).  The gradients of BOTH networks live in
    one persistent flat buffer the weight-gradient kernels write into (nerf_amd.parallel.FlatGradients); the ONE all_reduce over it is
    bracketed by its own events and reported separately.  --ipe / --contract: BASELINE configs[2] / configs[4] (integrated PE in the
    fine pass / scene contraction with unbounded depths); --hipgraph: the step incl. the RCCL collective replayed from a hipGraph."""
    import math
    import torch.nn.functional as F
    import nerf_amd
    from nerf_amd import ops, parallel
    from nerf_amd.addtional import ProposalLoss, ProposalNetwork, getBounds
    dist, world, rank, dev, backend = comm.dist, comm.world, comm.rank, comm.dev, comm.backend
    from nerf_amd.mip_methods import maxBlurFilter
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.nerf_base import NeRF
    from nerf_amd.utils import inverseSample
    nerf_amd.set_precision(a.precision)
    nerf_amd.set_train_dumps(a.train_dumps)
    n_rays, c_n, f_n = a.train_rays, C_COARSE, N_FINE
    near, far = (0.2, 30.0) if a.contract else (NEAR, FAR)
    is_ref = a.model == "ref"                                         # BASELINE configs[3]: Ref-NeRF with prop_normal (train.py:164-199, ref branch)
    if is_ref and (a.ipe or a.contract or a.train_dumps != "bf16"):
        sys.exit("bench.py: --mode train-ddp --model ref takes no --ipe / --contract / fp8 dumps")
    torch.manual_seed(0)                                              # same initial weights on every rank ...
    if is_ref:
        from nerf_amd.ref_model import BackFaceLoss, RefNeRF, WeightedNormalLoss
        prop, mip = ProposalNetwork(10, 256).to(dev).train(), RefNeRF(10, 4).to(dev).train()
        wnl, bfl = WeightedNormalLoss(), BackFaceLoss()
    else:
        prop, mip = ProposalNetwork(10, 256).to(dev).train(), MipNeRF(10, 4, 256).to(dev).train()
    if dist is not None:
        parallel.broadcast_parameters([mip, prop], src=0)             # ... and made sure of (ddp_train.py:98 DDP does this at wrap time)
    from nerf_amd.optim import Adam                                   # torch.optim.Adam's update as one HIP launch
    opt = Adam(list(mip.parameters()) + list(prop.parameters()), lr=1e-4, lr_on_device=a.hipgraph)
    flat = parallel.FlatGradients([mip, prop], opt)
    g = torch.Generator(device=dev).manual_seed(1000 + rank)          # every rank draws its own rays
    o = torch.tensor([0.0, 0.0, 0.5 if a.contract else 4.0], device=dev).expand(n_rays, 3)
    d = F.normalize(torch.randn(n_rays, 3, device=dev, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -1.0], device=dev), dim=-1)
    rays = torch.cat((o, d), -1).contiguous()
    tgt = torch.rand(n_rays, 3, device=dev, generator=g)
    res = (far - near) / c_n
    base = torch.linspace(near, far - res, c_n).to(dev)
    radius = 2.0 / math.sqrt(12.0) / 1111.0 if a.ipe else None        # pixel footprint of an 800x800 Lego camera (focal 1111)
    dir_norm = ops.dirs_norm(rays) if a.ipe else None
    seed_dev = torch.full((1,), 1234 + rank, dtype=torch.int64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    n_grad = flat.flat.numel()
    ploss = ProposalLoss()

    def ref_loss(z_c):
        """the Ref-NeRF branch of train.py:164-199 with prop_normal: density-gradient normals of both networks (RefNeRF.get_grad), merged
        coarse + fine depths, normal / back-face / coarse-normal losses"""
        pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous().requires_grad_(True)
        dens = prop.forward(pts)
        coarse_grad = -RefNeRF.get_grad(dens, pts)
        pw = maxBlurFilter(ProposalNetwork.get_weights(F.softplus(dens), z_c, rays[:, 3:]), 0.01)
        ops.advance_seed(seed_dev)
        fl, below = inverseSample(pw, z_c, f_n + 1, sort=True, u=ops.philox_uniforms((n_rays, f_n + 1), seed_dev=seed_dev))
        samples, fl, below, sort_ids = NeRF.coarseFineMerge(rays, z_c, fl, below)
        pos, dd = samples.split((3, 3), dim=-1)
        pos.requires_grad_(True)
        rgbo, nrm = mip.forward(pos, dd)
        dgrad = -RefNeRF.get_grad(rgbo[..., -1], pos)
        rgbo[..., -1] = F.softplus(rgbo[..., -1] + 0.5)
        rend, wts, _ = NeRF.render(rgbo, fl, rays[:, 3:], mip.density_act)             # (the reference's positional quirk, train.py:182)
        cnl = wnl(pw, RefNeRF.coarse_grad_select(dgrad, sort_ids, c_n).detach(), coarse_grad)
        return ploss(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2) + 4e-4 * (wnl(wts, dgrad, nrm) + 0.1 * cnl) + 0.1 * bfl(wts, nrm, dd)

    def step(timed_idx=None):
        u_c = ops.philox_uniforms((n_rays, c_n), seed_dev=seed_dev)   # (device-resident seed: nothing in the step reads host state)
        z_c = base + u_c * res
        if is_ref:
            loss = ref_loss(z_c)
            flat.begin_step()
            loss.backward()
            if timed_idx is not None:
                ev[timed_idx][0].record()
            flat.all_reduce()
            if timed_idx is not None:
                ev[timed_idx][1].record()
            opt.step()
            ops.advance_seed(seed_dev)
            return
        pts = (rays[:, None, :3] + rays[:, None, 3:] * z_c[:, :, None]).contiguous()
        dens = F.softplus(prop.forward(pts, contract=a.contract))
        pw = maxBlurFilter(ProposalNetwork.get_weights(dens, z_c, rays[:, 3:]), 0.01)
        ops.advance_seed(seed_dev)
        z_all, below = inverseSample(pw, z_c, f_n + 1, sort=True, u=ops.philox_uniforms((n_rays, f_n + 1), seed_dev=seed_dev))
        z_f = z_all[..., :-1].contiguous()
        if a.ipe:
            rgbo = mip.forward_rays(rays, z_all, f_n, ipe_radius=radius, ipe_dir_norm=dir_norm, contract=a.contract)
        elif a.contract:
            rgbo = mip.forward_rays(rays, z_f, f_n, contract=True)
        else:
            rgbo = mip.forward(NeRF.length2pts(rays, z_f))
        rend, wts, _ = NeRF.render(rgbo, z_f, rays[:, 3:], white_bkg=True)
        loss = ploss(getBounds(pw, below), wts.detach()) + torch.mean((rend - tgt) ** 2)
        flat.begin_step()                                             # (the gradient kernels overwrite the flat buffer: no zeroing pass)
        loss.backward()
        if timed_idx is not None:
            ev[timed_idx][0].record()
        flat.all_reduce()                                             # ONE collective (no-op without a process group)
        if timed_idx is not None:
            ev[timed_idx][1].record()
        opt.step()
        ops.advance_seed(seed_dev)

    graph = None
    for _ in range(max(a.warmup, 2 if a.hipgraph else 0)):
        step()
    if a.hipgraph:
        if dist is not None and backend != "nccl":
            sys.exit("bench.py: --hipgraph needs the RCCL backend (a gloo collective cannot be captured)")
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if dist is not None else {})):
            step()
    comm.sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if graph is not None:
            graph.replay()
        else:
            step(i)
    comm.sync()
    local_dt = time.perf_counter() - t0
    set_phase(AFTER_TIMED)
    dt = comm.max_over_ranks(local_dt)
    spread = rank_spread(comm, local_dt, a.steps)
    ar_ms = [s_.elapsed_time(e_) for s_, e_ in ev] if graph is None else None
    ar_ranks = comm.gather((sum(ar_ms) / a.steps * 1e3) if ar_ms else None)
    flop_per_ray = 3 * FLOP_PER_RAY                                      # forward + dgrad + wgrad
    if is_ref:                                                           # 64 proposal + (128 + 64 merged) Ref-NeRF samples; the two density-gradient chains not counted
        flop_per_ray = 3 * 2 * (C_COARSE * MAC_PROP + (N_FINE + C_COARSE) * 1_071_616)
    achieved = a.steps * n_rays * flop_per_ray / dt / 1e12               # per GPU
    rec = None
    if rank == 0:
        variant = ("Mip-NeRF + integrated PE (BASELINE configs[2])" if a.ipe else "NeRF / Mip-NeRF point PE") + (", scene contraction, near/far 0.2/30 (configs[4])" if a.contract else "")
        if is_ref:
            variant = "Ref-NeRF with prop_normal (BASELINE configs[3]): 64 proposal + 192 merged samples, IDE, density-gradient normals of both networks"
        rec = {"metric": "training rays/s (64+128 samples, fwd + bwd + gradient all_reduce + Adam)", "value": world * a.steps * n_rays / dt,
               "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
               "ms_per_step_ranks": spread,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f32",
               "data": "synthetic",
               "config": {"workload": "train.py:164-199 body + Adam on %d synthetic rays per rank, 64+128 samples, %s + ProposalNetwork(10,256); %s%s"
                                      % (n_rays, "RefNeRF(10,4)" if is_ref else "MipNeRF(10,4,256)", variant, ("; step replayed from a hipGraph (collective inside)" if graph is not None else "") +
                                         ("; training dumps in scaled e4m3" if a.train_dumps == "fp8" else "")),
                          "rays_per_step_per_gpu": n_rays, "parallelism": "ray-sharded replicas (dp%d), one flat gradient all_reduce per step" % world},
               "allreduce": {"elements": n_grad, "bytes": 4 * n_grad,
                             "us_per_step": ar_ranks[0] if dist is not None else None, "us_per_step_ranks": ar_ranks if dist is not None else None,
                             "backend": backend if dist is not None else None,
                             "note": "ONE all_reduce(AVG) on the persistent flat gradient buffer the weight-gradient kernels write into "
                                     "(nerf_amd/parallel.py FlatGradients; no cat / copy-back); HIP events on the compute stream around the call: it includes "
                                     "the wait for the slowest rank's backward (null when the step is replayed from a hipGraph or runs without a process group)"},
               "roofline": {"bound": "mfma", "kernel": "whole training step (3 x forward flops)", "achieved": achieved,
                            "peak": PEAK_BF16_DENSE / 1e12, "unit": "TFLOP/s", "frac": achieved * 1e12 / PEAK_BF16_DENSE, "traffic": None}}
    if a.precision == "bf16" and not a.no_gemm_ref:
        ref = mfma_stream_reference(comm, dev, achieved, achieved)
        if rank == 0:
            rec["roofline"]["mfma_stream_ref"] = ref
    if rank == 0:
        if not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_train()
        if comm.preflight is not None:
            rec["preflight"] = comm.preflight
        print(json.dumps(rec), flush=True)
        Watchdog.done = True
    set_phase("finish (host-side barrier)")
    comm.finish()


def render_strong(a, comm):
    """--mode render-strong: ONE 800x800 image per step, its 640 000 rays split into `world` contiguous shards (no collective on
    the data path), every uniform drawn in-kernel as a function of the GLOBAL ray index -- so the image does not depend on N -- and one
    all_gather of rgb + depth (16 B/ray) at the end of the step.  value = image rays / max-over-ranks time; "scaling": "strong"."""
    from nerf_amd import synthetic_weights as Wt
    from nerf_amd import ops, parallel
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.utils import fov2Focal, pose_spherical
    dist, world, rank, dev, backend = comm.dist, comm.world, comm.rank, comm.dev, comm.backend
    prec = ops.BF16 if a.precision == "bf16" else ops.F32
    prop, mip = ProposalNetwork(10, 256), MipNeRF(10, 4, 256)
    prop.load_state_dict(Wt.proposal_state("small")); mip.load_state_dict(Wt.mip_state("small"))
    prop, mip = prop.to(dev).eval(), mip.to(dev).eval()
    pk_prop, pk_mip = prop.packed(prec), mip.packed(prec)
    n = H * W
    focal = fov2Focal(0.6911112070083618, (H, W))
    fx, fy = float(focal[1]), float(focal[0])
    start, end = parallel.shard_range(n, rank, world, align=256)
    cnt = end - start
    z_base = torch.linspace(NEAR, FAR, C_COARSE).to(dev)
    poses = [pose_spherical(float(th), -30.0, 4.0)[:3] for th in torch.linspace(-180, 180, 41)[:-1]]
    ws = None

    def step(i):
        nonlocal ws
        rays = ops.generate_rays(poses[i % len(poses)], H, W, fx, fy, dev, start, cnt)
        rgb, depth, _, ws = ops.render_rays(pk_prop, pk_mip, prec, rays, z_base, None, None, N_FINE, NEAR, FAR, True, want_depth=True, workspace=ws,
                                            seed=0x5EED0000 + i, rng_ray_offset=start)
        out = torch.cat((rgb, depth[:, None]), -1)
        return parallel.gather_shards(out, n, 256) if dist is not None else out

    set_phase("warm-up")
    for i in range(a.warmup):
        step(i)
    set_phase("timed region")
    comm.sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        img = step(a.warmup + i)
    comm.sync()
    local_dt = time.perf_counter() - t0
    set_phase(AFTER_TIMED)
    dt = comm.max_over_ranks(local_dt)
    spread = rank_spread(comm, local_dt, a.steps)
    assert img.shape == (n, 4) and bool(torch.isfinite(img).all())
    if a.dump_image and rank == 0:                          # the LAST timed image (every rank holds the same one): N-independence checks
        torch.save(img.cpu(), a.dump_image)
    achieved = a.steps * n * FLOP_PER_RAY / dt / 1e12 / world          # per GPU
    rec = None
    if rank == 0:
        rec = {"metric": "rays/s (64+128 samples), 800x800, one image split over the ranks", "value": a.steps * n / dt, "unit": "rays/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "ms_per_step_ranks": spread, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "bf16" if prec == ops.BF16 else "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1] image (800x800, 64+128 samples, rows 1-10) rendered ONCE per step by all ranks together: "
                                      "contiguous 256-aligned ray shards, in-kernel Philox uniforms keyed by the global ray index, one all_gather of "
                                      "rgb + depth (16 B/ray) per image", "rays_per_step": n, "parallelism": "ray-sharded (dp%d), strong scaling" % world},
               "gather": {"bytes_per_image": 16 * n, "backend": backend if dist is not None else None},
               "roofline": {"bound": "mfma", "kernel": "whole render path (proposal + fine MLP flops of this rank's shard) over the step's wall time, gather included",
                            "achieved": achieved, "peak": (PEAK_BF16_DENSE if prec == ops.BF16 else PEAK_F32_MFMA) / 1e12, "unit": "TFLOP/s",
                            "frac": achieved * 1e12 / (PEAK_BF16_DENSE if prec == ops.BF16 else PEAK_F32_MFMA), "traffic": None},
               "whole_path_tflops": a.steps * n * FLOP_PER_RAY / dt / 1e12}
    if prec == ops.BF16 and not a.no_gemm_ref:
        ref = mfma_stream_reference(comm, dev, achieved, achieved)
        if rank == 0:
            rec["roofline"]["mfma_stream_ref"] = ref
    if rank == 0:
        if not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(a.cpu_rays)
        if comm.preflight is not None:
            rec["preflight"] = comm.preflight
        print(json.dumps(rec), flush=True)
        Watchdog.done = True
    set_phase("finish (host-side barrier)")
    comm.finish()


def main():
    # multi-process GPU work on these hosts needs dmabuf IPC (without it RCCL fails with `hipIpcGetMemHandle: invalid argument`); the variable is
    # read when the HIP runtime initialises, i.e. at the first device call below -- and inherited by the ranks this process may launch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    a = parse()
    launch_or_verify(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus
    Watchdog.rank, Watchdog.world = rank, world
    # every thread's Python stack + a partial JSON line + non-zero exit after N seconds: armed by DEFAULT at 900 s when N > 1 (a stuck
    # collective must show up inside the driver's 1 800 s slot as a message, not as a killed job); BENCH_DUMP_STACKS_AFTER=N overrides, 0 = off
    limit = int(os.environ.get("BENCH_DUMP_STACKS_AFTER", "900" if world > 1 else "0"))
    Watchdog.arm_run(limit)
    if world > 1:
        install_termination_record()
    if a.launch_check:
        return launch_check(a, world, rank)
    dist = None
    backend = os.environ.get("BENCH_BACKEND", "nccl")          # "gloo" only to smoke-test the N>1 control flow on a 1-GPU box
    n_dev = max(1, torch.cuda.device_count())
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local % n_dev)
        # DATA-PATH collectives (barriers of the timed region, max-over-ranks, the gradient all_reduce, the image all_gather): 5 minutes.
        # Nothing on the data path waits for another rank's solo measurement -- that wait is Comm.ctl's (gloo, host side).
        data_wait = datetime.timedelta(minutes=5)
        set_phase("init_process_group(%s)" % backend)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local % n_dev), timeout=data_wait)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=data_wait)
        assert dist.get_world_size() == a.gpus
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    comm = Comm(dist, world, rank, backend, dev)
    comm.preflight = preflight(dist, comm, world, rank, backend, dev) if world > 1 else None
    set_phase("mode %s" % a.mode)
    if a.mode == "train-ddp":
        return train_ddp(a, comm)
    if a.mode == "render-strong":
        return render_strong(a, comm)

    from nerf_amd import synthetic_weights as Wt
    from nerf_amd import ops
    from nerf_amd.addtional import ProposalNetwork
    from nerf_amd.mip_model import MipNeRF
    from nerf_amd.utils import fov2Focal, pose_spherical

    prec = ops.BF16 if a.precision == "bf16" else ops.F32
    is_ref = a.model == "ref"
    if is_ref:
        from nerf_amd.ref_model import RefNeRF
        prop, mip = ProposalNetwork(10, 256), RefNeRF(10, 4)
    else:
        prop, mip = ProposalNetwork(10, a.prop_width), MipNeRF(10, 4, a.fine_width)
    wtag = "small" if a.weights == "zero" else a.weights
    prop.load_state_dict(Wt.proposal_state(wtag, hidden=a.prop_width))
    mip.load_state_dict(Wt.ref_state(wtag) if is_ref else Wt.mip_state(wtag, hidden=a.fine_width))
    if a.weights == "zero":
        for q in list(prop.parameters()) + list(mip.parameters()):
            q.data.zero_()
    prop, mip = prop.to(dev).eval(), mip.to(dev).eval()
    pk_prop, pk_mip = prop.packed(prec), mip.packed(prec)               # weight upload + pack: outside the timed region

    n_rays = H * W
    focal = fov2Focal(0.6911112070083618, (H, W))
    fx, fy = float(focal[1]), float(focal[0])
    g = torch.Generator(device=dev).manual_seed(1000 + rank)
    u_strat = u_inv = None
    if a.rng == "resident":
        u_strat = torch.rand((n_rays, C_COARSE), device=dev, generator=g)    # resident in HBM before timing
        u_inv = torch.rand((n_rays, N_FINE + 1), device=dev, generator=g)
    z_base = torch.linspace(NEAR, FAR, C_COARSE).to(dev)
    jitter = (FAR - NEAR) / N_FINE
    poses = [pose_spherical(float(th), -30.0, 4.0)[:3] for th in torch.linspace(-180, 180, 41)[:-1]]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    ev_p = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]      # the proposal launch

    philox = a.rng == "philox"

    def step(i, timed_idx=None):
        pose = poses[(i + 7 * rank) % len(poses)]
        rays = ops.generate_rays(pose, H, W, fx, fy, dev)                                            # row 1
        seed = 0x5EED0000 + 1000003 * rank + i                                                       # a fresh stream per image
        us, ui = (None, None) if philox else (u_strat, u_inv)
        sc = ops.samples_rays(rays, C_COARSE, z_base=z_base, u=us, z_jitter=jitter, seed=seed)      # rows 2-4
        if timed_idx is not None:
            ev_p[timed_idx][0].record()
        dens = ops.proposal_forward_samples(pk_prop, prec, sc, (n_rays, C_COARSE), dev)
        if timed_idx is not None:
            ev_p[timed_idx][1].record()
        z_fine, _, _, z_c = ops.resample(dens, None, z_base, us, jitter, rays, ui, N_FINE + 1, want_zc=is_ref, seed=seed)   # rows 5-7
        if is_ref:                                                                                    # procedures.py:71-74
            z_all = ops.merge_depths(z_fine, z_c)
            if timed_idx is not None:
                ev[timed_idx][0].record()
            rgbo, _ = ops.ref_forward_samples(pk_mip, prec, ops.samples_rays(rays, z_all.shape[-1], z=z_all), (n_rays, z_all.shape[-1]), dev,
                                              want_normal=False)
            if timed_idx is not None:
                ev[timed_idx][1].record()
            rgb, w, depth, _ = ops.composite(rgbo, z_all, rays, True, True, ops.ACT_SOFTPLUS, (NEAR, FAR), sigma_shift=0.5)
            return rgb, depth, w
        if not a.fused:
            if timed_idx is not None:
                ev[timed_idx][0].record()
            rgbo = ops.mip_forward_samples(pk_mip, prec, ops.samples_rays(rays, N_FINE, z=z_fine), (n_rays, N_FINE), dev)
            if timed_idx is not None:
                ev[timed_idx][1].record()
            if os.environ.get("NERF_AMD_CLOCKPROBE") and timed_idx is not None:       # diagnostic lib variants only (MLP_CLOCKPROBE)
                torch.cuda.synchronize()
                cyc, ticks = rgbo.view(-1)[:4].view(torch.int64).tolist()
                sys.stderr.write("clockprobe: fine kernel %.2f ms at %.0f MHz\n" % (ticks / 1e5, cyc / max(ticks, 1) * 100.0))
            rgb, w, depth, _ = ops.composite(rgbo, z_fine, rays, True, True, ops.ACT_RELU, (NEAR, FAR))
            return rgb, depth, w
        if timed_idx is not None:                                                                     # rows 8-10, one launch:
            ev[timed_idx][0].record()                                                                 # fine MLP + fused compositing
        rgb, depth, w = ops.mip_forward_composite(pk_mip, prec, rays, z_fine, N_FINE, True, NEAR, FAR, want_depth=True,
                                                  want_weights=not os.environ.get("BENCH_NO_WEIGHTS"))
        if timed_idx is not None:
            ev[timed_idx][1].record()
        return rgb, depth, w

    set_phase("warm-up")
    for i in range(a.warmup):
        step(i)
    set_phase("timed region")
    comm.sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        out = step(a.warmup + i, i)
    comm.sync()
    local_dt = time.perf_counter() - t0
    set_phase(AFTER_TIMED)
    dt = comm.max_over_ranks(local_dt)
    spread = rank_spread(comm, local_dt, a.steps)
    assert os.environ.get("NERF_AMD_LIB") or bool(torch.isfinite(out[0]).all())     # (ablation builds compute garbage)

    fine_ms = sum(s.elapsed_time(e) for s, e in ev) / a.steps
    fine_ms_ranks = comm.gather(fine_ms)
    prop_ms = sum(s.elapsed_time(e) for s, e in ev_p) / a.steps
    wd = a.prop_width
    mac_prop = 63 * wd + 3 * wd * wd + wd                          # = MAC_PROP at width 256
    fw = a.fine_width
    mac_fine = 63 * fw + 3 * fw * fw + (fw + 63) * fw + fw * fw + 256 * fw + 256 * 256 + 256 + 283 * 128 + 128 * 3      # = MAC_FINE at width 256
    fine_flops = n_rays * N_FINE * 2 * mac_fine
    kernel_name = ("mip_kernel (fine MLP, %d MAC/sample; bottle_neck folded into rgb_layer.0 at pack time)" % mac_fine) + (" + fused compositing epilogue" if a.fused else "")
    flop_per_ray = FLOP_PER_RAY
    if is_ref:
        fine_flops = n_rays * (N_FINE + C_COARSE) * 2 * 1_071_616
        kernel_name = "ref_kernel (Ref-NeRF spatial + directional MLP, 1071616 MAC/sample, 192 merged samples/ray)"
        flop_per_ray = 2 * (C_COARSE * MAC_PROP + (N_FINE + C_COARSE) * 1_071_616)
    peak = PEAK_BF16_DENSE if prec == ops.BF16 else PEAK_F32_MFMA
    executed_flops = n_rays * ((N_FINE + C_COARSE) * 2 * 2128 * 512 if is_ref else N_FINE * 2 * (928 if fw == 256 else 352) * 512)   # mlp_layout.h N_FRAGS
    achieved = fine_flops / (fine_ms * 1e-3)
    rec = None
    if rank == 0:
        # HBM traffic of the dominant kernel per launch: rocprofv3 PMC passes of this same command (profiles/*pmc*), FETCH_SIZE
        # doubled per the gfx950 note of MI355X_MICROARCH.md; null when no profile for this configuration is committed
        traffic, traffic_src = committed_traffic("%s_%s" % (a.model, a.precision))
        rec = {
            "metric": "rays/s (64+128 samples), 800x800" if not is_ref else "rays/s (64+192 samples, Ref-NeRF), 800x800", "value": world * a.steps * n_rays / dt, "unit": "rays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "ms_per_step_ranks": spread,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if prec == ops.BF16 else "f32", "data": "synthetic" if a.weights == "small" else "synthetic (DIAGNOSTIC weights=%s)" % a.weights,
            "config": {"workload": ("BASELINE configs[1]: NeRF render 800x800 (640000 rays/step/GPU), 64 proposal + 128 fine samples, "
                                    "proposal MLP 63->256x4->1 + MipNeRF 8x256 MLP, rows 1-10 of SURVEY 8a, " +
                                    ("uniforms drawn in-kernel (Philox4x32-10)" if a.rng == "philox" else "uniforms resident in HBM")) if not is_ref else
                                   ("Ref-NeRF render 800x800 (BASELINE configs[3] shape), 64 proposal + 192 merged samples, rows 1-8,10,13"),
                       "rays_per_step_per_gpu": n_rays, "samples": [C_COARSE, N_FINE], "mlp_arith": "bf16 MFMA, fp32 accumulate"
                       if prec == ops.BF16 else "fp32 MFMA", "parallelism": "ray-sharded replicas (dp%d)" % world},
            "roofline": {"bound": "mfma", "kernel": kernel_name,
                         "achieved": achieved / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": achieved / peak,
                         "ms_per_launch": fine_ms, "ms_per_launch_ranks": fine_ms_ranks, "flop_per_launch": fine_flops, "traffic": traffic,
                         "traffic_source": dict(traffic_src, note="not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same "
                                                "command, committed in profiles/pmc_traffic.json with the kernel-source hashes") if traffic is not None else None,
                         "traffic_stale": traffic_src["stale"] if traffic_src else None,
                         # MFMA work actually issued (512 MAC per 32x16 fragment and sample; padded K, bottle_neck folded away)
                         "executed_tflops": executed_flops / (fine_ms * 1e-3) / 1e12, "executed_frac": executed_flops / (fine_ms * 1e-3) / peak},
            "roofline_proposal": {"bound": "mfma", "kernel": "proposal_kernel (63->%dx4->1, %d MAC/sample, stratified sample fetch + encoding in the prologue)" % (wd, mac_prop),
                                  "ms_per_launch": prop_ms, "flop_per_launch": n_rays * C_COARSE * 2 * mac_prop,
                                  "achieved": n_rays * C_COARSE * 2 * mac_prop / (prop_ms * 1e-3) / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                                  "frac": n_rays * C_COARSE * 2 * mac_prop / (prop_ms * 1e-3) / peak},
            "whole_path_tflops": world * a.steps * n_rays * flop_per_ray / dt / 1e12,
        }
        if wd != 256 or fw != 256:
            rec["config"]["workload"] += "; PROPOSAL WIDTH %d, FINE WIDTH %d (not the headline configuration)" % (wd, fw)
            rec["whole_path_tflops"] = world * a.steps * n_rays * 2 * (C_COARSE * mac_prop + N_FINE * mac_fine) / dt / 1e12
    if prec == ops.BF16 and not a.no_gemm_ref:
        ref = mfma_stream_reference(comm, dev, achieved / 1e12, executed_flops / (fine_ms * 1e-3) / 1e12)       # every rank, at the same time
        if rank == 0:
            rec["roofline"]["mfma_stream_ref"] = ref
    if rank == 0:                                            # (the other ranks wait at the host-side barrier of comm.finish())
        if not a.no_gemm_ref and prec == ops.BF16:
            rec["roofline"]["gemm_ref"] = gemm_reference(dev)
        if not a.no_train_rate and not is_ref:
            rec["train_step"] = train_rate(a.precision)
        if not a.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(a.cpu_rays)
        if comm.preflight is not None:
            rec["preflight"] = comm.preflight
        print(json.dumps(rec), flush=True)
        Watchdog.done = True
    set_phase("finish (host-side barrier)")
    comm.finish()


if __name__ == "__main__":
    try:
        main()
    except BaseException as e:                               # (SystemExit included: the preflight's one-line refusals)
        # an N > 1 rank that dies of an exception (a collective whose peer vanished, a refused preflight) leaves the record as well
        failed = not isinstance(e, SystemExit) or e.code not in (None, 0)
        if failed and Watchdog.world > 1 and "WORLD_SIZE" in os.environ and not Watchdog.done:
            partial_record("bench.py: rank %d of %d ended with %s: %s" % (Watchdog.rank, Watchdog.world, type(e).__name__, str(e)[:300]))
        raise
