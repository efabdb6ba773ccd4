"""The multi-rank PRODUCT path with the real HIP engine: `pretrain.py --algo fomaml` (config-3 toy workspace, 4 accents) as 2 ranks
started by torch.distributed.run, both on the one GPU of the box, meta-gradient exchange over gloo (MASR_DIST_BACKEND=gloo:
RCCL refuses two ranks on one device).  Everything but the transport is the 8-GPU code path: rank-consistent index draws,
task-per-rank sharding, padding rounds, replicated Noam-Adam, rank-0 evaluation + barrier, rank-0 checkpoints.  The two ranks must
end with identical meta weights, equal to the single-process CLI run up to the summation order of the four task gradients."""
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


def test_two_rank_cli_run_matches_single_process(golden_dir, tmp_path):
    cfg = cfg3_workspace(tmp_path, golden_dir)
    yaml.safe_dump(cfg, open(tmp_path / "cfg3.yaml", "w"))
    worker = str(ROOT / "tests" / "_dist_pretrain_worker.py")
    env = dict(os.environ, PYTHONPATH=str(ROOT), MASR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r1 = subprocess.run([sys.executable, worker, str(tmp_path), "w1"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port()), worker, str(tmp_path), "w2"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr[-2000:]
    one = torch.load(tmp_path / "w1_r0.pt")
    a, b = torch.load(tmp_path / "w2_r0.pt"), torch.load(tmp_path / "w2_r1.pt")
    assert one["step"] == a["step"] == b["step"] == 5
    assert torch.equal(a["meta"], b["meta"]), "replicated Noam-Adam must leave identical meta weights on both ranks"
    # 4 task gradients summed in task order (one process) vs (t0 + t2) + (t1 + t3) after two all-reduces: fp32 rounding of the
    # gradient only; Adam's step is <= lr = 3.2e-8 per element and meta-step, 4 meta-steps
    assert float((a["meta"] - one["meta"]).abs().max()) <= 4 * 2.5 * 3.2e-8 + 1.2e-7
    log_dir = tmp_path / "testing-logs" / "pretrain" / "cfg3" / "fomaml" / "w2" / "canada" / "0"
    for f in ("snapshot.step.4", "model.wer.best", "best_wer", "dev_avg_wer", "train_loss", "global_step"):
        assert (log_dir / f).exists(), f
    assert len((log_dir / "dev_avg_wer").read_text().splitlines()) == 2           # rank 0 evaluated after meta-steps 2 and 4


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) starts two ranks itself and prints exactly ONE JSON line on stdout
    with n_gpus = 2, the weak-scaling value, the long-run leg and the whole-meta-step leg (exchange over gloo here, both ranks on
    the one GPU of the box); a --gpus that disagrees with WORLD_SIZE is refused."""
    import json
    env = dict(os.environ, MASR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--tasks-per-gpu", "2",
                        "--no-cpu-baseline", "--long-seconds", "0.3", "--meta-steps", "2", "--no-matrix", "--no-mixed", "--no-e2e"], cwd=tmp_path, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "utt/s" and d["value"] > 0
    assert d["config"]["tasks"] == 4 and d["long_run"]["seconds"] >= 0.2
    ms = d["meta_step"]
    assert ms["tasks"] == 4 and ms["allreduces_per_meta_step"] == 2 and ms["ms"] >= ms["ms_without_exchange"] > 0 and ms["allreduce_ms_isolated"] > 0
    cs = ms["concurrent_slots"]                                                       # 2 ranks x 2 task slots, one all-reduce of the local sum
    assert cs["tasks_per_gpu"] == 2 and cs["tasks"] == 4 and cs["allreduces_per_meta_step"] == 1 and cs["ms"] > 0
    assert ms["transport"] == "pg_gloo" and ms["slot_cap"]["tasks_per_gpu_run_by_pretrain_cli"] == 2
    assert ms["evaluate"]["eval_ms_rank0"] > 0 and ms["evaluate"]["idle_ms_other_ranks"] >= ms["evaluate"]["eval_ms_rank0"] * 0.5     # rank 1 waited while rank 0 evaluated
    assert "roofline" in d and d["roofline"]["slot"] in d["roofline"]["launches"]
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], cwd=tmp_path, env=dict(env, WORLD_SIZE="1"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
