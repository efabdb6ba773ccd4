#!/usr/bin/env python3
"""bench.py -- inner-step utterances/sec of the MI355X-native meta-ASR path (BASELINE.json metric).

A "step" is ONE inner-loop step of FOMetaASRInterface.run_task (src/fo_meta_interface.py:238-250):
run_batch(train=True) [forward + label-smoothed CE + backward] -> clip_grad_norm_(5) -> SGD(momentum .9,
nesterov) step, on one synthetic batch of 10 s x 80-dim fbank utterances that is already resident in
HBM when the timed region starts.  Model = config/transformer/pretrain/fometa-hkust.yaml geometry
(d_model 512, 8 heads, d_inner 2048, 2 enc / 4 dec, odim 367, dropout 0.1), random-init weights.

    python bench.py [--gpus N --steps K --warmup W]          # N>1 via torch.distributed.run, one rank per GPU

N ranks = N groups of independent accent-tasks (FOMAML shards tasks over GPUs; the inner step has no exchange),
so scaling is "weak": value = N * tasks_per_gpu * B * K / max-over-ranks time.  The meta-gradient all-reduce of
the OUTER step is measured separately (field "meta_step") because the metric counts inner steps.

--tasks-per-gpu (default 4): tasks of one meta-step are independent, so each GPU runs several of them concurrently
(one model replica + HIP stream + host thread per task; `--tasks_per_gpu` of pretrain.py).  Every task still performs
full B-utterance inner steps; "single_task_fomaml" / "single_task_train" in the output are the same measurement with one task per GPU under
the launch schedule of pretrain.py --algo fomaml / of train.py (masr_set_ksplit off / on).
GPU_MAX_HW_QUEUES=8 (ROCm runtime setting, set below unless the caller chose a value): four task streams plus the
copy/side streams need more than the default four hardware queues, otherwise two tasks serialise on one queue
(measured: 4 tasks 5490 utt/s with 4 queues, 6790 with 8; 3 tasks 6440 either way).
"""
import argparse
import json
import os
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")            # read by the HIP runtime when it initialises (first GPU call)

import numpy as np
import torch

HKUST = {
    "idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.1, "pos_dropout": 0.1,
    "tgt_share_weight": 1, "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4},
    "inner_optimizer_cls": "SGD", "inner_optimizer_opt": {"momentum": 0.9, "nesterov": True},
    "meta_opt_cls": "noam", "meta": {"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}},
}
ODIM = 367
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


_T0 = time.perf_counter()


def log(msg):
    """progress to stderr (stdout carries only the one JSON line)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def synth_batch(B, T, D, seed):
    """feat ~ N(0,1), ilens = T, olens ~ U{10..40}, labels ~ U{1..365}."""
    rng = np.random.RandomState(seed)
    xs = torch.from_numpy(rng.randn(B, T, D).astype(np.float32))
    olens = rng.randint(10, 41, size=B)
    ys = [torch.from_numpy(rng.randint(1, 366, size=int(n)).astype(np.int64)) for n in olens]
    return xs, torch.full((B,), T, dtype=torch.int64), ys, torch.from_numpy(olens.astype(np.int64))


def fwd_flops_by_class(T, D, L=31, E=512, F=2048, NE=2, ND=4, C=367):
    """SURVEY 8(d) algorithmic forward FLOPs per utterance (multiply-add = 2), split by kernel class."""
    H2, W2 = T // 2, D // 2
    Tp, Dp = H2 // 2, W2 // 2
    conv = 2 * 9 * (1 * 64 * T * D + 64 * 64 * T * D + 64 * 128 * H2 * W2 + 128 * 128 * H2 * W2)
    gemm = 2 * Tp * (128 * Dp) * E                                                            # vgg2enc
    gemm += NE * (2 * Tp * E * 3 * E + 2 * Tp * E * E + 4 * Tp * E * F)                        # encoder Linears
    gemm += ND * ((2 * L * E * 3 * E + 2 * L * E * E) + (2 * L * E * E + 2 * Tp * E * 2 * E + 2 * L * E * E) + 4 * L * E * F)
    gemm += 2 * L * E * C                                                                     # char_trans
    attn = NE * 4 * Tp * Tp * E + ND * (4 * L * L * E + 4 * L * Tp * E)                        # QK^T and PV
    return {"conv": conv, "gemm": gemm, "attn": attn}


def fwd_flops_per_utt(T, D, L=31, **kw):
    return sum(fwd_flops_by_class(T, D, L, **kw).values())


def host_cores():
    """cores this job may actually use: affinity mask capped by the cgroup CPU quota (a GPU box hands one GPU's job
    a 16-core share of a much larger host; spawning one thread per visible core thrashes)"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("MASR_CPU_THREADS", "16"))))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


class ClockSampler:
    """shader / memory clock and board power of the local GPU, read from sysfs by a background thread while a leg runs (no child
    process, no HIP call): pp_dpm_sclk marks the current DPM level with '*', hwmon power1_average is in microwatts.  What the driver's
    record needs to carry for "MFMA busy 0.5 vs 0.42 of nominal peak = clock": the reading UNDER LOAD.  (MI355X_MICROARCH.md: the
    in-kernel clock runs up to ~10 % below this reading; it is an upper bound.)"""

    def __init__(self, local=0, period=0.25):
        import glob
        self.period, self.samples, self._stop, self._th = period, [], threading.Event(), None
        # the box may hold 8 GPUs of which this job sees one: find OUR card by its PCI address; failing that, sample every card and
        # report the one that draws the most power (the busy one)
        self.dev, self.cands = None, [os.path.dirname(f) for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))]
        try:
            pr = torch.cuda.get_device_properties(local)
            addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            if os.path.exists(f"/sys/bus/pci/devices/{addr}/pp_dpm_sclk"):
                self.dev, self.how = f"/sys/bus/pci/devices/{addr}", f"PCI {addr}"
        except Exception:
            pass

    @staticmethod
    def _cur(path):
        try:
            for line in open(path):
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().split("Mhz")[0].split("MHz")[0])
        except Exception:
            return None
        return None

    @staticmethod
    def _power(dev):
        import glob
        for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_input")):
            try:
                return float(open(f).read()) / 1e6
            except Exception:
                pass
        return None

    @staticmethod
    def _cap(dev):
        """board power cap in W (hwmon power1_cap, microwatts): the four-slot headline runs AT it, so the line carries it beside the reading"""
        import glob
        for f in glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_cap")):
            try:
                return float(open(f).read()) / 1e6
            except Exception:
                pass
        return None

    def _read(self, dev):
        return (self._cur(os.path.join(dev, "pp_dpm_sclk")), self._cur(os.path.join(dev, "pp_dpm_mclk")), self._power(dev))

    def _run(self):
        while not self._stop.wait(self.period):
            if self.dev:
                self.samples.append(self._read(self.dev))
            else:                                                # no PCI match: the card under the highest power is ours
                rows = [self._read(d) for d in self.cands]
                rows = [r for r in rows if r[0] is not None]
                if rows:
                    self.samples.append(max(rows, key=lambda r: (r[2] or 0.0, r[0])))

    def __enter__(self):
        if self.dev or self.cands:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._th:
            self._th.join()

    def summary(self):
        col = lambda i: [x[i] for x in self.samples if x[i] is not None]
        s_, m_, p_ = col(0), col(1), col(2)
        if not s_:
            return {"available": False, "note": "pp_dpm_sclk not readable on this box"}
        return {"available": True, "samples": len(s_), "sclk_mhz_min": min(s_), "sclk_mhz_median": float(np.median(s_)), "sclk_mhz_max": max(s_),
                "mclk_mhz_median": float(np.median(m_)) if m_ else None, "power_w_median": float(np.median(p_)) if p_ else None,
                "power_cap_w": self._cap(self.dev or (self.cands[0] if self.cands else "")),
                "card": getattr(self, "how", "busiest of %d cards (by power)" % len(self.cands)),
                "source": "sysfs pp_dpm_sclk / pp_dpm_mclk (current DPM level) and hwmon power1_average, sampled every %.2f s during the leg" % self.period}


def cpu_baseline(cfg, B, T, D, steps=6):
    """The oracle's inner step (fp32 torch on the host cores) on a bounded sample of the same workload."""
    from oracle import ref_cpu
    n = host_cores()
    torch.set_num_threads(n)
    c = dict(cfg)
    c["dropout"] = c["pos_dropout"] = 0.0            # the oracle is the dropout-free restatement
    sd = ref_cpu.deterministic_state_dict(c, ODIM, seed=1)
    p = ref_cpu.leafify(sd, c)
    bufs = {}
    batch = synth_batch(B, T, D, 0)
    lr = ref_cpu.inner_lr(c)
    ref_cpu.inner_step(p, c, (batch[0], batch[1], batch[2], batch[3].clone()), 0.2, bufs, lr)      # warm-up
    log(f"cpu warm-up step done ({n} threads)")
    t0 = time.perf_counter()
    for _ in range(steps):
        ref_cpu.inner_step(p, c, (batch[0], batch[1], batch[2], batch[3].clone()), 0.2, bufs, lr)
        log("cpu step done")
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "utt/s", "cores": n, "cpu_model": cpu_model(), "kind": "port",
            "sample": f"{steps} timed + 1 warm-up inner steps, B={B} x T={T} x D={D}, same model/config, fp32 torch CPU oracle, dropout 0"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps per task slot (120 x 6.5 ms = 0.8 s: a 30-step region read 3 % below the 6 s long_run leg of the same process)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=16, help="utterances per inner step per GPU (16 = what the shipped half_batch_ilen rule yields at 1000 frames)")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--idim", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16, help="utterances per step of the CPU baseline leg (same B as the GPU leg by default)")
    ap.add_argument("--cpu-steps", type=int, default=6)
    ap.add_argument("--long-seconds", type=float, default=6.0, help="also time a region of at least this many seconds (\"long_run\"); 0 = skip")
    ap.add_argument("--mixed", action="store_true", help="(default now) time the mixed-length leg (ilens ~ U{200..1500}), reported as \"mixed_lengths\"")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-length leg")
    ap.add_argument("--no-matrix", action="store_true", help="skip the SURVEY 8(d) matrix legs (idim 83, B 32, the 4e2d / E256 geometry), reported as \"matrix\"")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (\"e2e_pretrain\": the pretrain.py --algo fomaml loop WITH its data path, "
                    "shards on disk -> BucketSampler -> pinned collate -> upload -> tasks -> meta update; tools/bench_pretrain.py in a child process)")
    ap.add_argument("--single-seconds", type=float, default=1.0, help="length of the one-task-per-GPU regions (\"single_task_fomaml\", \"single_task_train\")")
    ap.add_argument("--no-stagger", action="store_true", help="start all concurrent tasks at the same instant (lock-step)")
    ap.add_argument("--no-meta-step", action="store_true", help="skip the whole-meta-step leg (\"meta_step\" in the output)")
    ap.add_argument("--meta-rounds", type=int, default=2, help="task rounds per rank in the meta-step leg (>= 2 shows the all-reduce overlap)")
    ap.add_argument("--meta-steps", type=int, default=5)
    ap.add_argument("--tasks-per-gpu", type=int, default=4, help="concurrent independent accent-tasks per GPU (1 = reference order)")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This process has not touched the
    # GPU (no HIP call, no torch.cuda.* so far), it only waits for the child launcher and returns its exit code; the ranks
    # are fresh child processes (a GPU-initialised process is never re-exec'd).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        log(f"launching {args.gpus} ranks: {' '.join(cmd)}")
        sys.exit(subprocess.run(cmd, env=env).returncode)

    # stdout carries exactly ONE JSON line: native libraries (gloo, RCCL with NCCL_DEBUG) print to fd 1 on their own, so
    # fd 1 is pointed at stderr for the whole run and the JSON line goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                 f"`python bench.py --gpus N` or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("MASR_BENCH_BACKEND", "nccl")       # "gloo": rehearsal of the N>1 path on a 1-GPU box
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local = local % max(ndev, 1)
    # MASR_FORCE_COLLECTIVE=1 with one rank: the process group, barriers, max-over-ranks reductions and the meta-step exchange legs all
    # go through the collective backend (RCCL) although an all-reduce over one rank is the identity -- the N-GPU code path, executed
    # on a 1-GPU box (tests/test_hip_rccl_world1.py)
    if world > 1 or os.environ.get("MASR_FORCE_COLLECTIVE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    def all_reduce_(t, op=None):
        """in-place all-reduce of a device tensor (RCCL), or through host memory for the gloo rehearsal"""
        if backend == "nccl":
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
        else:
            c = t.cpu()
            dist.all_reduce(c, op=op or dist.ReduceOp.SUM)
            t.copy_(c)

    import masr_amd
    from masr_amd.engine import MasrEngine
    from masr_amd.model import reference_init_state_dict        # the product's own seed-exact init (no oracle in the GPU leg)

    cfg = dict(HKUST)
    cfg["idim"] = args.idim
    B, T, D = args.batch, args.frames, args.idim
    K = max(1, args.tasks_per_gpu)
    o = cfg["meta"]["optimizer_opt"]
    lr = cfg["d_model"] ** (-0.5) * o["k"] * o["warmup_steps"] ** (-0.5)      # inner lr, fo_meta_interface.py:41-45
    torch.manual_seed(531)                                       # CLI seed of the reference (pretrain.py:29)
    sd0 = reference_init_state_dict(cfg, ODIM)                   # random-init weights exactly as the reference draws them

    class Task:                                                  # one accent-task slot: replica + stream + resident batch
        def __init__(self, k, cfg=cfg, sd0=sd0, B=B, T=T, D=D):
            self.eng = MasrEngine(cfg, ODIM, label_smoothing=0.2, device=dev)
            self.eng.load_state_dict(sd0)
            self.eng.set_seed(531 + rank * 64 + k)
            self.eng.set_concurrency(K)
            xs, self.il, self.ys, self.ol = synth_batch(B, T, D, seed=rank * 64 + k)   # numpy seed 0 + task index
            self.xs = xs.to(dev)
            self.mom = torch.zeros_like(self.eng.params)
            # slot 0 on the default stream, as in FOMetaASRInterface: never more than K streams with work queued
            self.stream = torch.cuda.current_stream(dev) if k == 0 else torch.cuda.Stream(device=dev)
            self.i = 0

        def step(self):
            self.eng.run_batch(self.xs, self.il, self.ys, self.ol, train=True)
            self.eng.clip_sgd_step(self.mom, 5.0, lr, 0.9, True, first_step=(self.i == 0))
            self.i += 1

    tasks = [Task(k) for k in range(K)]
    eng = tasks[0].eng
    log(f"rank {rank}: {K} task slot(s) ready, workspace {eng._l.masr_workspace_bytes(eng.h, B, T, 42) / 1e9:.2f} GB each")

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(active, nsteps, nwarm, stagger_s=0.0):
        """nwarm untimed + nsteps timed inner steps on every task of `active`, one host thread + HIP stream per task.
        stagger_s: task k starts k * stagger_s late (inside the timed region): identical synthetic tasks started together
        stay in lock-step, so all of them want the chip-filling conv kernels at the same moment and the latency-bound
        decoder kernels at the same moment; real accent-tasks have different batch shapes and de-phase by themselves."""
        gate = threading.Barrier(len(active) + 1)

        def body(t, k):
            with torch.cuda.stream(t.stream):
                for _ in range(nwarm):
                    t.step()
                t.stream.synchronize()
                gate.wait()                                      # (1) warm-up done everywhere
                gate.wait()                                      # (2) timed region starts
                if stagger_s > 0 and k > 0:
                    time.sleep(k * stagger_s)
                for _ in range(nsteps):
                    t.step()
                t.stream.synchronize()
        ths = [threading.Thread(target=body, args=(t, k)) for k, t in enumerate(active)]
        for th in ths:
            th.start()
        gate.wait()
        barrier()
        t0 = time.perf_counter()
        gate.wait()
        for th in ths:
            th.join()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    F_step = 3 * fwd_flops_per_utt(T, D)

    SINGLE_STANDS_FOR = {
        False: "pretrain.py --algo fomaml with one task per GPU (BASELINE configs[3] per rank): FOMetaASRInterface keeps the decoder's k-split GEMM "
               "schedule OFF for every --tasks_per_gpu (K slots == the sequential run == N ranks, bit for bit)",
        True: "train.py mono-accent / multi-task (BASELINE configs[1]): the trainer turns the decoder's k-split GEMM schedule ON (masr_set_ksplit)"}

    def time_single(task_list, Bs, flops_per_utt, seconds, ksplit=False):
        """one task per GPU over a region of at least `seconds` (a first short region sizes it): value, ms per step, model FLOPs / peak.
        ksplit: the schedule of the CLI the figure stands for (SINGLE_STANDS_FOR); the slots' own settings are put back afterwards."""
        e0 = task_list[0].eng
        e0.set_concurrency(1)                                    # (the engine has the GPU to itself: the launch-geometry hint of a lone task)
        e0.set_ksplit(ksplit)
        dtp = timed(task_list[:1], 8, args.warmup)
        n1 = max(8, int(seconds / (dtp / 8)) + 1)
        dt1 = timed(task_list[:1], n1, 2)
        assert (e0.step_counters()["ksplit_gemms"] > 0) == bool(ksplit)
        e0.set_concurrency(K)
        e0.set_ksplit(False)
        v = world * Bs * n1 / dt1
        return {"value": v, "unit": "utt/s", "ms_per_step": dt1 / n1 * 1e3, "steps": n1, "seconds": dt1, "ksplit": bool(ksplit),
                "stands_for": SINGLE_STANDS_FOR[bool(ksplit)],
                "model_frac_of_bf16_peak": v * flops_per_utt / 1e12 / (PEAK_BF16_TFLOPS * world)}
    single = single_first = single_train = None
    if K > 1 and args.single_seconds <= 0:
        # (profiling runs: a short one-task leg only, for the stagger -- 10 + warm-up steps in the trace, as tools/save_profiles.py assumes)
        n1 = max(5, args.steps // 3)
        tasks[0].eng.set_concurrency(1)
        dt1 = timed(tasks[:1], n1, args.warmup)
        tasks[0].eng.set_concurrency(K)
        single_first = {"value": world * B * n1 / dt1, "unit": "utt/s", "ms_per_step": dt1 / n1 * 1e3, "steps": n1, "seconds": dt1,
                        "model_frac_of_bf16_peak": world * B * n1 / dt1 * F_step / 1e12 / (PEAK_BF16_TFLOPS * world)}
        single = single_first
    elif K > 1:
        # >= 1 s, BEFORE the headline region: it also brings the clocks up, and the stagger of the K-task legs is derived from a
        # number that one ramp hiccup cannot halve (round 3: 6 steps = 16 ms as the first timed region of the process read 2 800 utt/s
        # on the driver's box against 5 900 here).  The figure reported as "single_task" is measured again AFTER the long run.
        single_first = time_single(tasks, B, F_step, args.single_seconds)
        log(f"single task per GPU (first, {single_first['seconds']:.2f} s): {single_first['value']:.1f} utt/s")
    stagger = 0.0
    if K > 1 and not args.no_stagger and single_first is not None:
        stagger = single_first["ms_per_step"] * 1e-3 / K         # spread the K task phases over one single-task step
    dt = timed(tasks, args.steps, args.warmup, stagger)
    st = eng.read_stats()
    assert np.isfinite(st["loss"]) and np.isfinite(st["grad_norm"]), st
    log(f"timed region: {dt:.3f} s for {args.steps} steps x {K} task(s); loss {st['loss']:.4f}")
    long_run = None
    if args.long_seconds > 0:
        # the driver's K-step region is ~0.2 s (clock ramp and jitter are a visible part of it): the same measurement over >= 2 s
        nlong = max(args.steps, int(1.1 * args.long_seconds / (dt / args.steps)) + 1)      # (+10 %: the short region includes the ramp)
        with ClockSampler(local) as clk:
            dtl = timed(tasks, nlong, 2, stagger)
        long_run = {"steps": nlong, "seconds": dtl, "value": world * K * B * nlong / dtl, "ms_per_step": dtl / nlong * 1e3,
                    "clocks_under_load": clk.summary()}
        log(f"long run: {dtl:.2f} s for {nlong} steps -> {long_run['value']:.1f} utt/s; clocks {long_run['clocks_under_load']}")
    if K > 1 and args.single_seconds > 0:
        with ClockSampler(local) as clk1:
            single = time_single(tasks, B, F_step, args.single_seconds)      # warm clocks: right behind the long run
        single["clocks_under_load"] = clk1.summary()
        single["first_measurement"] = single_first
        log(f"single task per GPU, pretrain.py schedule (after the long run, {single['seconds']:.2f} s): {single['value']:.1f} utt/s = {single['model_frac_of_bf16_peak']:.3f} of the bf16 peak")
        single_train = time_single(tasks, B, F_step, args.single_seconds, ksplit=True)
        log(f"single task per GPU, train.py schedule (k-split on, {single_train['seconds']:.2f} s): {single_train['value']:.1f} utt/s = {single_train['model_frac_of_bf16_peak']:.3f} of the bf16 peak")

    # ---- mixed-length leg (ilens ~ U{200..1500}): every task cycles through 8 batches of its own; all
    # utterances of a batch share one length (what the reference's BucketSampler yields) and the half-batch rule applies
    # above 512 frames (B/1 or B*2 utterances, batch_size 32 of the shipped configs = 2 * B here)
    mixed = None
    if not args.no_mixed:
        rng = np.random.RandomState(1234 + rank)
        pools = []
        for t in tasks:
            pool = []
            for j in range(8):
                Tm = int(rng.randint(200, 1501))
                Bm = B if Tm > 512 else 2 * B
                xs, il, ys, ol = synth_batch(Bm, Tm, D, seed=rank * 1000 + len(pools) * 8 + j)
                pool.append((xs.to(dev), il, ys, ol))
            pools.append(pool)
        for t, pool in zip(tasks, pools):
            t.pool, t.j = pool, 0

            def mstep(t=t):
                xs, il, ys, ol = t.pool[t.j % 8]
                t.j += 1
                t.eng.run_batch(xs, il, ys, ol, train=True)
                t.eng.clip_sgd_step(t.mom, 5.0, lr, 0.9, True, first_step=False)
            t.step = mstep
        nm = max(8, args.steps // 2 // 8 * 8)                    # whole cycles of the pool
        dtm = timed(tasks, nm, 8, stagger)
        utt = sum(sum(p[j % 8][1].numel() for j in range(nm)) for p in pools)
        frames = sum(sum(int(p[j % 8][1].sum()) for j in range(nm)) for p in pools)
        mixed = {"value": world * utt / dtm, "unit": "utt/s", "frames_per_s": world * frames / dtm, "steps": nm,
                 "ilens": "U{200..1500}, one length per batch; B utterances above 512 frames, 2B below (half-batch rule)"}
        log(f"mixed lengths: {mixed['value']:.1f} utt/s, {mixed['frames_per_s'] / 1e6:.2f} M frames/s")
        for t in tasks:
            del t.step                                           # back to the fixed-shape step of the class

    # ---- SURVEY 8(d) matrix: the same K-task measurement at the shipped idim 83, at B = 32 (the reference's full-batch rule below 512
    # frames; not the headline shape) and on the 4e2d / E256 / H4 geometry -- one number each, short regions
    matrix = None
    if not args.no_matrix:
        matrix = {}
        nmx = max(6, args.steps // 3)
        for name, over, Bm, Dm in (("idim83", {}, B, 83), ("batch32", {}, 2 * B, D),
                                   ("geometry_4e2d_E256", {"d_model": 256, "nheads": 4, "encoder": {"nlayers": 4}, "decoder": {"nlayers": 2}}, B, D)):
            c2 = dict(cfg, idim=Dm, **over)
            torch.manual_seed(531)
            sdm = reference_init_state_dict(c2, ODIM)
            tm = [Task(k, c2, sdm, Bm, T, Dm) for k in range(K)]
            for t_, t0_ in zip(tm, tasks):
                t_.stream = t0_.stream
            dtm_ = timed(tm, nmx, 3, stagger)
            matrix[name] = {"value": world * K * Bm * nmx / dtm_, "unit": "utt/s", "ms_per_step": dtm_ / nmx * 1e3, "steps": nmx, "batch_per_task": Bm,
                            "idim": Dm, "tasks_per_gpu": K, **({"model": over} if over else {})}
            log(f"matrix {name}: {matrix[name]['value']:.1f} utt/s")
            if name == "idim83" and K > 1:                       # the shipped feature width with one task per GPU (what configs[3] runs per GPU)
                matrix[name]["single_task_fomaml"] = time_single(tm, Bm, 3 * fwd_flops_per_utt(T, Dm), 0.5 * args.single_seconds)
                log(f"matrix {name}, one task per GPU: {matrix[name]['single_task_fomaml']['value']:.1f} utt/s")
            del tm
            torch.cuda.empty_cache()

    def step(i):
        tasks[0].step()

    # ---- one whole FOMAML meta-step (fo_meta_interface.py:136-158,200-221) with its exchange: every rank runs R task
    # rounds {copy meta->task, k=1 inner step, val-batch gradient, clip 5}; each round's clipped gradient is all-reduced
    # on the side stream (parallel.TaskSharder.reduce_async, RCCL) WHILE the next round's inner step runs; then
    # sum / n_tasks and the replicated Noam-Adam step.  Timed with and without the exchange: the difference is the part
    # of the all-reduce that is NOT hidden.  (The metric itself counts inner steps, which have no exchange.)
    meta = None
    if not args.no_meta_step:
      try:
          from masr_amd.parallel import TaskSharder
          sharder = TaskSharder.from_env() if dist is not None else TaskSharder()
          t0_ = tasks[0]
          R, nmeta = args.meta_rounds, args.meta_steps
          orig = eng.params.clone()
          ea, eas = torch.zeros_like(orig), torch.zeros_like(orig)
          contribs = [torch.zeros_like(orig) for _ in range(R)]
          total = torch.zeros_like(orig)

          def meta_step(i, exchange):
              for r in range(R):
                  eng.copy(eng.params, orig)
                  eng.mark_dirty()
                  eng.run_batch(t0_.xs, t0_.il, t0_.ys, t0_.ol, train=True)
                  eng.clip_sgd_step(t0_.mom, 5.0, lr, 0.9, True, first_step=True)
                  eng.run_batch(t0_.xs, t0_.il, t0_.ys, t0_.ol, train=True)        # val batch at the adapted weights
                  contribs[r].zero_()
                  eng.clip_accumulate(contribs[r], 5.0)
                  if exchange and dist is not None:
                      if backend == "nccl":
                          sharder.reduce_async(contribs[r])
                      else:
                          all_reduce_(contribs[r])
              if exchange:
                  sharder.wait_all()
              total.zero_()
              for c in contribs:
                  eng.axpy(total, c, 1.0)
              eng.scale(total, 1.0 / (R * world))
              eng.adam_step(orig, total, ea, eas, 1e-7, 0.9, 0.98, 1e-9, i + 1)

          def time_meta(exchange):
              meta_step(0, exchange)
              barrier()
              t1 = time.perf_counter()
              for i in range(nmeta):
                  meta_step(1 + i, exchange)
              barrier()
              d = time.perf_counter() - t1
              if dist is not None:
                  tt = torch.tensor([d], device=coll_dev, dtype=torch.float64)
                  dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                  d = float(tt.item())
              return d / nmeta * 1e3
          ms_no = time_meta(False)
          ms_ex = time_meta(True) if dist is not None else ms_no
          ar_ms = None
          if dist is not None:
              barrier()
              t1 = time.perf_counter()
              for _ in range(5):
                  all_reduce_(contribs[0])
              torch.cuda.synchronize(dev)
              ar_ms = (time.perf_counter() - t1) / 5 * 1e3
          meta = {"rounds_per_rank": R, "inner_steps": 1, "tasks": R * world, "ms": ms_ex, "ms_without_exchange": ms_no,
                  "exposed_exchange_ms": ms_ex - ms_no, "allreduce_ms_isolated": ar_ms, "allreduces_per_meta_step": R if dist is not None else 0,
                  "payload_mb": orig.numel() * 4 / 1e6, "backend": ("rccl" if backend == "nccl" else backend) if dist is not None else None,
                  "utt_per_s": world * R * 2 * B / (ms_ex * 1e-3)}
          # ---- what a first run on several GPUs must say about itself: which transport carried the meta-gradient and why,
          # the task-slot cap the CLI would apply, and what the (quirk-kept) evaluate() costs the ranks that do not evaluate
          from masr_amd.fo_meta_interface import slot_cap
          meta["transport"] = sharder.transport if dist is not None and backend == "nccl" else (f"pg_{backend}" if dist is not None else "none")
          meta["transport_note"] = (sharder.transport_note or ("native exchange (masr_allreduce) is opt-in: MASR_NATIVE_ALLREDUCE=1"
                                                                 if not sharder.native else "MASR_NATIVE_ALLREDUCE=1")) if dist is not None else "one rank, no collective"
          meta["slot_cap"] = {"tasks_per_gpu_asked": K, "tasks_per_gpu_run_by_pretrain_cli": slot_cap(K, 8, world, dist is not None), "meta_batch_size": 8,
                              "rule": "collective and tasks_per_gpu > 3 and ceil(meta_batch_size / world) > tasks_per_gpu -> 3 (--no_slot_cap keeps the setting)"}
          n_dev = 8
          barrier()
          t_e = time.perf_counter()
          if rank == 0:                                           # evaluate() as shipped (reference quirk Q2): rank 0 alone, on ITS last task's adapted weights
              for _ in range(n_dev):
                  eng.run_batch(t0_.xs, t0_.il, t0_.ys, t0_.ol, train=False)
              torch.cuda.synchronize(dev)
          eval_ms = (time.perf_counter() - t_e) * 1e3
          barrier()
          wait_ms = (time.perf_counter() - t_e) * 1e3
          idle = wait_ms if rank != 0 else 0.0
          if dist is not None:
              tt = torch.tensor([idle, eval_ms if rank == 0 else 0.0], device=coll_dev, dtype=torch.float64)
              dist.all_reduce(tt, op=dist.ReduceOp.MAX)
              idle, eval_ms = float(tt[0].item()), float(tt[1].item())
          meta["evaluate"] = {"dev_batches": n_dev, "eval_ms_rank0": eval_ms, "idle_ms_other_ranks": idle if world > 1 else 0.0,
                              "mode": "rank 0 alone while the other ranks wait at a barrier (reference quirk Q2: evaluate() runs on the last task's adapted "
                                      "weights, which differ per rank); pretrain.py --fix_snapshot_meta_weights evaluates the META weights and splits the dev "
                                      "accents over the ranks"}
          log(f"meta-step: {ms_ex:.2f} ms with exchange, {ms_no:.2f} ms without, isolated all-reduce {ar_ms}; transport {meta['transport']}")
          # ---- the same meta-step as pretrain.py --tasks_per_gpu K runs it: the K tasks of a rank concurrently on their slots (host
          # thread + stream each, slot 0 on the default stream), the meta update reading the K gradient buffers in one pass
          # (one rank) or their local sum all-reduced first (several ranks); no host sync between meta-steps
          if K > 1:
              main = torch.cuda.current_stream(dev)

              def task_body(t):
                  with torch.cuda.stream(t.stream):
                      e = t.eng
                      e.copy(e.params, orig)
                      e.mark_dirty()
                      e.run_batch(t.xs, t.il, t.ys, t.ol, train=True)
                      e.clip_sgd_step(t.mom, 5.0, lr, 0.9, True, first_step=True)
                      e.run_batch(t.xs, t.il, t.ys, t.ol, train=True)
                      e.clip_grads(5.0)

              def meta_step_slots(i):
                  ths = []
                  for t in tasks:
                      if t.stream != main:
                          t.stream.wait_stream(main)
                      th = threading.Thread(target=task_body, args=(t,))
                      th.start(); ths.append(th)
                  for th in ths:
                      th.join()
                  for t in tasks:
                      if t.stream != main:
                          main.wait_stream(t.stream)
                  if dist is None:
                      eng.adam_sum_step(orig, [t.eng.grads for t in tasks], 1.0 / K, ea, eas, 1e-7, 0.9, 0.98, 1e-9, i + 1)
                  else:
                      total.zero_()
                      for t in tasks:
                          eng.axpy(total, t.eng.grads, 1.0)
                      if backend == "nccl":
                          sharder.reduce_async(total)
                          sharder.wait_all()
                      else:
                          all_reduce_(total)
                      eng.scale(total, 1.0 / (K * world))
                      eng.adam_step(orig, total, ea, eas, 1e-7, 0.9, 0.98, 1e-9, i + 1)
              for i in range(2):
                  meta_step_slots(i)
              barrier()
              t1 = time.perf_counter()
              for i in range(nmeta):
                  meta_step_slots(2 + i)
              barrier()
              d = time.perf_counter() - t1
              if dist is not None:
                  tt = torch.tensor([d], device=coll_dev, dtype=torch.float64)
                  dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                  d = float(tt.item())
              ms_k = d / nmeta * 1e3
              meta["concurrent_slots"] = {"tasks_per_gpu": K, "tasks": K * world, "ms": ms_k, "utt_per_s": world * K * 2 * B / (ms_k * 1e-3),
                                          "allreduces_per_meta_step": 1 if dist is not None else 0}
              log(f"meta-step, {K} concurrent task slots: {ms_k:.2f} ms = {meta['concurrent_slots']['utt_per_s']:.0f} utt/s")
              for t in tasks:
                  t.eng.copy(t.eng.params, orig)
                  t.eng.mark_dirty()
          eng.copy(eng.params, orig)
          eng.mark_dirty()
          del contribs, total, ea, eas
      except Exception as ex:                                     # (the metric counts inner steps: an exchange leg that fails must not take the line down)
        import traceback
        log("meta-step leg FAILED: " + traceback.format_exc())
        meta = {"error": f"{type(ex).__name__}: {ex}"}

    # ---- roofline: every conv launch of the step timed live with HIP events on the launch stream (one engine slot per
    # launch), the LONGEST launch is the "dominant kernel"; plus a per-class table (algorithmic GFLOP per step / measured ms)
    roof = None
    prof_all = None
    nprof = 5
    if rank == 0 and not args.no_profile:
        eng.set_concurrency(1)
        eng.profile(True)
        for i in range(nprof):
            step(i)
        prof_all = eng.profile_read()
        eng.profile(False)
        H2, W2 = T // 2, D // 2
        c2, c3, c4 = 2.0 * 9 * 64 * 64 * B * T * D, 2.0 * 9 * 64 * 128 * B * H2 * W2, 2.0 * 9 * 128 * 128 * B * H2 * W2
        # name -> (kernel, algorithmic FLOPs per launch, algorithmic HBM bytes per launch: bf16 in + out; the pooling forwards write the
        # pooled map (2 B) and one code byte per pooled element instead of the full-resolution map)
        conv_launch = {
            "conv2_fwd": ("conv3x3_resw_kernel<16,16> (64->64 forward + fused 2x2 max-pool, pooled map + codes out)", c2, B * T * D * 64 * 2 + B * H2 * W2 * 64 * 3),
            "conv3_fwd": ("conv3x3_stream_kernel<64,128> (64->128 forward; + the ReLU mask of its output as 16 bytes of sign bits per pixel)", c3, B * H2 * W2 * ((64 + 128) * 2 + 16)),
            "conv4_fwd": ("conv3x3_stream_kernel<128,128> (128->128 forward + fused 2x2 max-pool, pooled map + codes out)", c4, B * H2 * W2 * 128 * 2 + B * (H2 // 2) * (W2 // 2) * 128 * 3),
            # (reads d(conv2 out), the ReLU mask of conv1's output as one 64-bit word per pixel, the fp32 network input; writes 640 sums per workgroup)
            "conv2_dgrad": ("conv3x3_resw_w1x_kernel<unpool> (64<-64 dgrad on 16x16 tiles, patches expanded from the pooled gradient + pool codes, + fused conv1 weight gradient)", c2, B * H2 * W2 * 64 * 3 + B * T * D * (8 + 4)),
            "conv3_dgrad": ("conv3x3_stream_kernel<128,64> (64<-128 dgrad)", c3, B * H2 * W2 * (128 + 64) * 2),
            "conv4_dgrad": ("conv3x3_stream_kernel<128,128,32-row tiles,sign-bit mask,unpool> (128<-128 dgrad through the ReLU mask, patches expanded from the pooled gradient + pool codes)", c4, B * (H2 // 2) * (W2 // 2) * 128 * 3 + B * H2 * W2 * (128 * 2 + 16)),
            # (the wgrad slots time the partial-slab kernel alone; the slab reduces are under conv1_wgrad, "weight-gradient folds")
            # (the two in front of a max-pool read their dy as pooled gradient (2 B) + pool code (1 B) per pooled element; every launch
            # writes 256 partial slabs of 9*CIN*COUT fp32 per 64x64 channel block: 37.7 MB)
            "conv2_wgrad": ("conv3x3_wgrad2_kernel<64,64,pooled dy> (one workgroup per CU)", c2, B * T * D * 64 * 2 + B * H2 * W2 * 64 * 3 + 256 * 9 * 64 * 64 * 4),
            "conv3_wgrad": ("conv3x3_wgrad2_kernel<64,128>", c3, B * H2 * W2 * (64 + 128) * 2 + 128 * 2 * 9 * 64 * 64 * 4),
            "conv4_wgrad": ("conv3x3_wgrad2_kernel<128,128,pooled dy>", c4, B * H2 * W2 * 128 * 2 + B * (H2 // 2) * (W2 // 2) * 128 * 3 + 64 * 4 * 9 * 64 * 64 * 4),
        }
        pmc, pmc_src = {}, None
        try:                                                     # HBM bytes per launch from this round's PMC passes (tools/pmc_traffic.py)
            pj = json.load(open(ROOT / "profiles" / "pmc_traffic.json"))
            if pj["workload"] == {"batch": B, "frames": T, "idim": D}:
                pmc, pmc_src = pj["kernels"], f"profiles/pmc_traffic.json ({pj['collected']}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{pj['command']}`)"
        except Exception:
            pass
        launches = {}
        for name, (kern, flops, abytes) in conv_launch.items():
            ms, n = prof_all[name]
            if n == 0:
                continue
            per = ms / n
            launches[name] = {"kernel": kern, "avg_launch_ms": per, "gflop_per_launch": flops / 1e9, "tflops": flops / (per * 1e-3) / 1e12,
                              "frac_of_bf16_peak": flops / (per * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "algorithmic_bytes": abytes,
                              "traffic_bytes": pmc.get(name)}
        # the encoder-row Linear weight gradients are ONE launch since round 3 (round 4: gemm_wgrad_grouped16_kernel): a candidate too
        ms_w, n_w = prof_all["wgrad_enc"]
        if n_w == nprof:                                         # (one launch per step: the grouped path is on)
            E_, F_, Tp_ = cfg["d_model"], cfg["d_inner"], T // 4
            shapes = [(3 * E_, E_), (E_, E_), (F_, E_), (E_, F_)] * cfg["encoder"]["nlayers"] + [(E_, 128 * (D // 4))] + [(2 * E_, E_)] * cfg["decoder"]["nlayers"]
            rows = B * Tp_
            fl = sum(2.0 * rows * n_ * k_ for n_, k_ in shapes)
            by = sum(rows * (n_ + k_) * 2 + n_ * k_ * 4 for n_, k_ in shapes)
            merged = prof_all["wgrad_dec"][1] == 0               # the decoder-row weight gradients ride in the same launch (engine.hip flush_enc_wgrads)
            if merged:
                rows_dd = B * (int(max(tasks[0].ol)) + 1)
                dshapes = [(3 * E_, E_), (E_, E_), (E_, E_), (E_, E_), (F_, E_), (E_, F_)] * cfg["decoder"]["nlayers"] + [(ODIM, E_)]
                fl += sum(2.0 * rows_dd * n_ * k_ for n_, k_ in dshapes)
                by += sum(rows_dd * (n_ + k_) * 2 + n_ * k_ * 4 for n_, k_ in dshapes)
            per = ms_w / n_w
            launches["wgrad_enc"] = {"kernel": ("gemm_wgrad_grouped16_kernel (ALL Linear weight gradients of the step: %d GEMMs dW = dY^T X over the %d encoder rows and %d over the %d decoder rows, one grid of 256x256 tiles on eight waves, LDS-DMA ring; the long tiles are dispatched first)" % (len(shapes), rows, len(dshapes), rows_dd)) if merged else
                                               ("gemm_wgrad_grouped16_kernel (ALL encoder-row Linear weight gradients of the step, %d GEMMs dW = dY^T X over %d rows, one grid of 256x256 tiles on eight waves, LDS-DMA ring)" % (len(shapes), rows)),
                                     "avg_launch_ms": per, "gflop_per_launch": fl / 1e9, "tflops": fl / (per * 1e-3) / 1e12,
                                     "frac_of_bf16_peak": fl / (per * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "algorithmic_bytes": by, "traffic_bytes": pmc.get("wgrad_enc")}
        dom = max(launches, key=lambda k: launches[k]["avg_launch_ms"])
        d = launches[dom]
        roof = {"bound": "mfma", "kernel": d["kernel"], "slot": dom, "achieved": d["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": d["frac_of_bf16_peak"], "traffic": d["traffic_bytes"], "traffic_source": pmc_src if d["traffic_bytes"] else None,
                "algorithmic_bytes": d["algorithmic_bytes"], "avg_launch_ms": d["avg_launch_ms"], "flops_per_launch": d["gflop_per_launch"] * 1e9,
                "selection": "longest single launch of the step at this commit (every chip-filling launch that is timed alone is listed under \"launches\": the nine conv launches and the grouped weight-gradient GEMM)",
                "launches": launches}
        # per-class table of ONE single-task step: algorithmic FLOPs (forward x 3; conv1 has no dgrad) vs measured kernel time
        Lmax = int(max(tasks[0].ol)) + 1
        fc = fwd_flops_by_class(T, D, Lmax)
        gf = {"conv": (3 * fc["conv"] - 2 * 9 * 64 * T * D) * B / 1e9, "gemm": 3 * fc["gemm"] * B / 1e9, "attn": 3 * fc["attn"] * B / 1e9}
        cls_of = lambda k: "conv" if k.startswith("conv") else "gemm" if k.startswith(("gemm", "wgrad")) else "attn" if k.startswith("attn") else "other"
        per_class = {c: {"ms_per_step": 0.0, "launches_per_step": 0.0} for c in ("conv", "gemm", "attn", "other")}
        for k, (ms, n) in prof_all.items():
            per_class[cls_of(k)]["ms_per_step"] += ms / nprof
            per_class[cls_of(k)]["launches_per_step"] += n / nprof               # (profiling scopes; a scope may hold 2-3 tiny launches)
        for c_, g_ in gf.items():
            t_ = per_class[c_]["ms_per_step"] * 1e-3
            per_class[c_].update({"algorithmic_gflop_per_step": g_, "tflops": g_ / 1e3 / t_, "frac_of_bf16_peak": g_ / 1e3 / t_ / PEAK_BF16_TFLOPS})
        per_class["other"]["note"] = "HBM-bound passes: LayerNorm, loss/embedding, grad-norm + clip + SGD (24 B/param), operand shadows"
        # the HBM-bound class against the HBM roofline: ALGORITHMIC bytes of each pass (what it must read + write once) over its measured time
        NP = eng.numel
        E_, Tp_ = cfg["d_model"], T // 4
        rows_e, rows_d = B * Tp_, B * Lmax
        n_ln_e, n_ln_d = 2 * cfg["encoder"]["nlayers"] + 1, 3 * cfg["decoder"]["nlayers"] + 1
        ln_elems = (n_ln_e * rows_e + n_ln_d * rows_d) * E_
        hbm_bytes = {
            "optim": NP * (4 + 20),              # grad-norm reads g; clip + SGD reads p, g, momentum and writes p, momentum
            "shadows": NP * (4 + 2 + 2),         # fp32 weights in, bf16 operand + its transpose out
            "layernorm": ln_elems * (10 + 14),   # fwd: x in, y fp32 + bf16 out; bwd: dy + x in, dx fp32 + bf16 out
            "conv1_fwd": B * T * D * (4 + 64 * 2 + 8),   # fp32 input in, 64-channel bf16 map + one 64-bit word of ReLU sign bits per pixel out
        }
        # (no `pool` pass since round 5: the dgrad and weight-gradient kernels behind a max-pool expand pooled gradient + codes themselves)
        hbm_bytes = {k: v for k, v in hbm_bytes.items() if prof_all[k][0] > 0}
        if prof_all["shadows"][0] == 0:
            # the SGD step runs INSIDE the shadow launch (MASR_FUSED_SGD=1): p is read once; one pass, timed under `optim`
            hbm_bytes["optim"] = NP * (4 + 20 + 4)
            hbm_bytes.pop("shadows", None)
        t_hbm = sum(prof_all[k][0] for k in hbm_bytes) / nprof * 1e-3
        b_hbm = float(sum(hbm_bytes.values()))
        per_class["other"].update({
            "bytes_per_step": b_hbm, "achieved_TBps": b_hbm / t_hbm / 1e12, "frac_of_8TBps": b_hbm / t_hbm / 8e12,
            "passes": {k: {"algorithmic_bytes": v, "ms_per_step": prof_all[k][0] / nprof, "TBps": v / (prof_all[k][0] / nprof * 1e-3) / 1e12}
                       for k, v in hbm_bytes.items()},
            "bytes_note": "algorithmic bytes per step of the passes listed under `passes` (each tensor read / written once) over their event-timed "
                          "duration; conv1_fwd (HBM-bound, listed with the conv class above) is included here, the `misc` slot (loss, embedding, "
                          "casts, split-K combine: ~0.1 ms of small launches) is not"})
        tot_ms = sum(v["ms_per_step"] for v in per_class.values())
        roof["per_class"] = per_class
        roof["single_task_step"] = {"kernel_ms": tot_ms, "algorithmic_gflop": sum(gf.values()),
                                    "frac_of_bf16_peak": sum(gf.values()) / 1e3 / (tot_ms * 1e-3) / PEAK_BF16_TFLOPS}

    if rank == 0:
        utt = world * K * B * args.steps
        value = utt / dt
        F = fwd_flops_per_utt(T, D)
        out = {
            "metric": "inner-step utterances/sec (10s x 80-dim fbank)", "value": value, "unit": "utt/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "pretrain.py --algo fomaml inner step (run_batch + clip 5 + SGD nesterov), "
                                   "config/transformer/pretrain/fometa-hkust.yaml geometry, synthetic 10s x %d-dim fbank" % D,
                       "batch_per_task": B, "tasks_per_gpu": K, "frames": T, "idim": D, "dropout": cfg["dropout"],
                       "tasks": world * K, "parallelism": f"{K} concurrent task(s) per GPU x {world} GPU(s)"},
            "algorithmic_gflop_per_utt_fwd_bwd": 3 * F / 1e9,
            "model_tflops": value * 3 * F / 1e12, "model_frac_of_bf16_peak": value * 3 * F / 1e12 / (PEAK_BF16_TFLOPS * world),
            # one task per GPU, each figure under the schedule of the CLI it stands for ("stands_for"): the FOMAML interface runs whole
            # reductions in the decoder's few-row GEMMs, the mono / multi trainers the k-split (+3 % alone on the GPU, a different fp32 order)
            "single_task_fomaml": single,
            "single_task_train": single_train,
            "long_run": long_run,
            "rccl_ranks": world if (dist is not None and backend == "nccl") else 0,
            "mixed_lengths": mixed,
            "matrix": matrix,
            "loss": st["loss"], "grad_norm": st["grad_norm"],
        }
        if roof:
            out["roofline"] = roof
            out["kernel_ms_per_step"] = {k: v[0] / nprof for k, v in prof_all.items()}
        if meta:
            out["meta_step"] = meta
        if not args.no_e2e and world == 1 and dist is None:
            # the product loop end to end (not part of `value`, whose inputs are resident in HBM): a child process, because this one
            # holds K engines and the loop wants the same streams; synthetic shards of the bench shape on local disk
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory(prefix="masr_e2e_") as td:
                log("end-to-end pretrain loop (child process) ...")
                cmd = [sys.executable, str(ROOT / "tools" / "bench_pretrain.py"), "--utts", "512", "--frames", str(T), "--idim", str(D), "--meta-steps", "40",
                       "--warm", "5", "--configs", f"host:{K}", "--root", td]
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    e2e = json.loads(r.stdout.strip().splitlines()[-1])
                    res = e2e["results"][0]
                    out["e2e_pretrain"] = {"value": res["utt_per_s"], "unit": "utt/s", "ms_per_meta_step": res["ms_per_meta_step"], "meta_steps": res["meta_steps"],
                                           "tasks_per_gpu": res["tasks_per_gpu"], "shards": "host memmap (local disk), pinned collate, one DMA per batch",
                                           "workload": e2e["workload"], "command": " ".join(cmd[1:-2])}
                    log(f"end-to-end: {res['utt_per_s']:.0f} utt/s")
                except Exception as ex:                              # never fail the bench line over the extra leg
                    out["e2e_pretrain"] = {"error": f"{type(ex).__name__}: {ex}"}
            # BASELINE configs[1]: train.py mono-accent on one GPU, WITH its data path (tools/bench_train.py: one 2048-utterance shard on disk ->
            # BucketSampler -> collate -> one step per batch, SGD as in the shipped adapt configs and Noam-Adam), a child process as above
            out["e2e_train"] = {}
            for opt_name in ("SGD", "noam"):
                with tempfile.TemporaryDirectory(prefix="masr_e2e_train_") as td:
                    cmd = [sys.executable, str(ROOT / "tools" / "bench_train.py"), "--utts", "2048", "--steps", "200", "--warm", "20", "--optimizer", opt_name, "--root", td]
                    try:
                        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                        res = json.loads(r.stdout.strip().splitlines()[-1])
                        out["e2e_train"][opt_name] = {"value": res["utt_per_s"], "unit": "utt/s", "ms_per_step": res["ms_per_step"], "steps": res["steps"],
                                                      "workload": res["workload"], "command": " ".join(cmd[1:-2])}
                        log(f"end-to-end train.py ({opt_name}): {res['utt_per_s']:.0f} utt/s")
                    except Exception as ex:
                        out["e2e_train"][opt_name] = {"error": f"{type(ex).__name__}: {ex}"}
        if not args.no_cpu_baseline and world == 1:
            log("cpu baseline (oracle on host cores) ...")
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, T, D, args.cpu_steps)
            log("cpu baseline done")
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # a rank that fails must END: with several ranks the others sit in a collective or a barrier, and worker threads or a half-issued
        # exchange can keep this interpreter alive for ever -- the launcher then never learns that the job is dead (seen: a 2-rank run silent
        # for its whole time limit after an error on rank 0 only)
        import traceback
        traceback.print_exc()
        sys.stderr.flush(); sys.stdout.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            os._exit(1)
        raise
