#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU: training steps of the BLSTM-CTC model (config/blstm geometry: VGG front-end, 3 x BLSTM-P(360), Linear,
CTC) on one synthetic batch -- run_batch(train) + clip 5 + SGD(momentum .9, nesterov), as MonoASRInterface.train drives BLSTMTrainer.
Prints ms per step; under `rocprofv3 --kernel-trace --stats` it yields the per-kernel times of the CTC lattice and LSTM step kernels
(profiles/rNN_blstm_kernel_stats.txt).

    python tools/bench_blstm.py [--batch 8] [--frames 400] [--steps 20] [--out gpurun_out/blstm.json]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import masr_amd  # noqa: F401
    from masr_amd.blstm_engine import BlstmEngine, reference_init_state_dict
    cfg = {"encoder": {"idim": 83, "enc_dim": 360, "proj_dim": 360, "odim": 360, "sample_rate": "1_1_1", "dropout": "0_0_0"}}      # config/blstm/mono-test.yaml
    torch.manual_seed(531)
    eng = BlstmEngine(cfg, 367)
    eng.load_state_dict(reference_init_state_dict(cfg, 367))
    rng = np.random.RandomState(0)
    B, T = a.batch, a.frames
    xs = torch.from_numpy(rng.randn(B, T, 83).astype(np.float32)).cuda()
    il = torch.full((B,), T, dtype=torch.int64)
    ol = torch.from_numpy(rng.randint(10, 41, size=B).astype(np.int64))
    ys = [torch.from_numpy(rng.randint(1, 366, size=int(n)).astype(np.int64)) for n in ol]
    mom = torch.zeros_like(eng.params)

    def step(i):
        eng.run_batch(xs, il, ys, ol, train=True)
        eng.clip_sgd_step(mom, 5.0, 0.01, 0.9, True, first_step=(i == 0))
    for i in range(a.warm):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warm + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.read_stats()
    res = {"workload": f"BLSTM-CTC training step, B={B} x {T} frames x 83 dims, 3 x BLSTM-P(360), odim 367, SGD", "steps": a.steps,
           "ms_per_step": dt / a.steps * 1e3, "utt_per_s": B * a.steps / dt, "loss": st["loss"], "grad_norm": st["grad_norm"]}
    assert np.isfinite(st["loss"])
    print(json.dumps(res))
    if a.out:
        Path(ROOT / a.out).write_text(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
