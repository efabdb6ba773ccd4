"""RCCL executes on the MI355X at least once before the first multi-GPU run: the exchange of parallel.TaskSharder and one
`pretrain.py --algo fomaml` run through the nccl backend with world_size 1 (MASR_FORCE_COLLECTIVE=1 routes a single rank through
the collective path: an all-reduce over one rank is the identity, so the meta weights must equal the plain run's).  What this
exercises that the gloo rehearsals cannot: init_process_group("nccl", device_id=...), the side-stream wait_stream ordering,
work.wait() + current_stream().wait_stream(side), RCCL's own stream next to the task streams.
Reference loop being sharded: src/fo_meta_interface.py:136-158,200-221."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from oracle.make_goldens import cfg3_workspace  # noqa: E402

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**kw):
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASR_DIST_BACKEND", "MASR_FORCE_COLLECTIVE"):
        env.pop(k, None)
    env.update(kw)
    return env


@pytest.mark.parametrize("native", [True, False])
def test_tasksharder_exchange_through_rccl_with_one_rank(tmp_path, native):
    """native: the exchange through the C ABI (masr_allreduce: librccl bound directly, own side stream, clip pipelined with the
    collective; opt-in: MASR_NATIVE_ALLREDUCE=1); otherwise ProcessGroupNCCL's all_reduce, the default transport"""
    env = _env(MASR_NATIVE_ALLREDUCE="1") if native else _env()
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "_rccl_world1_worker.py")], cwd=tmp_path, env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0 and "rccl-world1-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("tasks_per_gpu", [1, 2])
def test_pretrain_cli_meta_steps_through_rccl_with_one_rank(golden_dir, tmp_path, tasks_per_gpu):
    cfg = cfg3_workspace(tmp_path, golden_dir)
    yaml.safe_dump(cfg, open(tmp_path / "cfg3.yaml", "w"))
    worker = str(ROOT / "tests" / "_dist_pretrain_worker.py")
    extra = ["--tasks_per_gpu", str(tasks_per_gpu)]
    plain = subprocess.run([sys.executable, worker, str(tmp_path), "plain"] + extra, cwd=tmp_path, env=_env(), capture_output=True, text=True, timeout=420)
    assert plain.returncode == 0, plain.stderr[-3000:]
    rccl = subprocess.run([sys.executable, worker, str(tmp_path), "rccl"] + extra, cwd=tmp_path, env=_env(MASR_FORCE_COLLECTIVE="1", MASR_NATIVE_ALLREDUCE="1"),
                          capture_output=True, text=True, timeout=420)
    assert rccl.returncode == 0, rccl.stderr[-3000:]
    a, b = torch.load(tmp_path / "plain_r0.pt"), torch.load(tmp_path / "rccl_r0.pt")
    assert not a["collective"] and b["collective"] and b["backend"] == "nccl"
    assert b["native"] and not a["native"]                     # the exchange went through the C ABI (masr_allreduce), clip on the wire at K = 1
    if tasks_per_gpu == 1:
        # the communicator cannot be made (forced): every rank agrees to fall back to ProcessGroupNCCL, the clip that was to ride on
        # the wire runs as one pass first -- same meta weights, bit for bit
        fb = subprocess.run([sys.executable, worker, str(tmp_path), "fallback"] + extra, cwd=tmp_path,
                            env=_env(MASR_FORCE_COLLECTIVE="1", MASR_NATIVE_ALLREDUCE="1", MASR_TEST_FAIL_ALLREDUCE_INIT="1"), capture_output=True, text=True, timeout=420)
        assert fb.returncode == 0, fb.stderr[-3000:]
        c = torch.load(tmp_path / "fallback_r0.pt")
        assert c["collective"] and not c["native"] and "ProcessGroupNCCL instead" in fb.stderr
        assert torch.equal(c["meta"], a["meta"])
    assert a["step"] == b["step"] == 5
    # the same four task gradients; summed as (((0 + g0) + g1) + g2) + g3 either way at one task per GPU, as (g0 + g1) + (g2 + g3)
    # through the per-wave sums at two: fp32 rounding of the gradient only, Adam's step is <= lr = 3.2e-8 per element and meta-step
    d = float((a["meta"] - b["meta"]).abs().max())
    assert d <= (0.0 if tasks_per_gpu == 1 else 4 * 2.5 * 3.2e-8 + 1.2e-7), d
    log_dir = tmp_path / "testing-logs" / "pretrain" / "cfg3" / "fomaml" / "rccl" / "canada" / "0"
    assert (log_dir / "snapshot.step.4").exists() and len((log_dir / "dev_avg_wer").read_text().splitlines()) == 2


def test_bench_collective_legs_through_rccl_with_one_rank(tmp_path):
    """bench.py with MASR_FORCE_COLLECTIVE=1: init_process_group("nccl", device_id=...), the barriers, the max-over-ranks reduction
    of the timed region and the meta-step legs with their all-reduces (per round on the side stream; one per wave of task slots) run
    through RCCL with one rank -- what the driver's multi-GPU bench will execute first, minus the other ranks."""
    import json
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "2", "--tasks-per-gpu", "2", "--no-cpu-baseline",
                        "--long-seconds", "0", "--meta-steps", "2", "--no-matrix", "--no-mixed", "--no-e2e"], cwd=tmp_path, env=_env(MASR_FORCE_COLLECTIVE="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["value"] > 0
    ms = d["meta_step"]
    assert ms["backend"] == "rccl" and ms["allreduces_per_meta_step"] == 2 and ms["allreduce_ms_isolated"] > 0 and ms["ms"] > 0
    assert ms["concurrent_slots"]["allreduces_per_meta_step"] == 1 and ms["concurrent_slots"]["ms"] > 0
    # a first multi-GPU run describes itself: the transport (default: ProcessGroupNCCL; the C-ABI exchange is opt-in), the slot cap the CLI
    # would apply, and what the quirk-kept evaluate() costs
    assert ms["transport"] == "pg_nccl" and "MASR_NATIVE_ALLREDUCE" in ms["transport_note"]
    assert ms["slot_cap"]["tasks_per_gpu_asked"] == 2 and ms["slot_cap"]["tasks_per_gpu_run_by_pretrain_cli"] == 2
    ev = ms["evaluate"]
    assert ev["dev_batches"] == 8 and ev["eval_ms_rank0"] > 0 and ev["idle_ms_other_ranks"] == 0.0 and "fix_snapshot_meta_weights" in ev["mode"]
