#!/usr/bin/env python3
"""Per-launch timing of the engine's profiling slots (HIP events on the launch stream) for one single-task inner step at the bench shape:
    python tools/conv_slots.py [--steps 20] [--batch 16] [--frames 1000] [--idim 80] [--all]
A/B runs of a kernel variant: set its environment switch and compare (the switches are read once per process)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import masr_amd  # noqa: E402,F401
from masr_amd.engine import MasrEngine  # noqa: E402
from masr_amd.model import reference_init_state_dict  # noqa: E402
from bench import HKUST, ODIM, synth_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--idim", type=int, default=80)
    ap.add_argument("--all", action="store_true", help="every slot, not only the conv launches")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    cfg = dict(HKUST, idim=a.idim)
    torch.manual_seed(531)
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
    eng.load_state_dict(reference_init_state_dict(cfg, ODIM))
    xs, il, ys, ol = synth_batch(a.batch, a.frames, a.idim, 0)
    xs = xs.cuda()
    mom = torch.zeros_like(eng.params)
    for i in range(5):
        eng.run_batch(xs, il, ys, ol, train=True)
        eng.clip_sgd_step(mom, 5.0, 2.8e-4, 0.9, True, i == 0)
    eng.profile(True)
    for _ in range(a.steps):
        eng.run_batch(xs, il, ys, ol, train=True)
        eng.clip_sgd_step(mom, 5.0, 2.8e-4, 0.9, True, False)
    prof = eng.profile_read()
    eng.profile(False)
    st = eng.read_stats()
    rows = [(k, ms / a.steps * 1e3, n / a.steps) for k, (ms, n) in prof.items() if n and (a.all or k.startswith("conv"))]
    print(f"{a.tag} loss {st['loss']:.5f} | " + "  ".join(f"{k} {us:.1f}" + (f"/{n:.0f}" if n > 1 else "") for k, us, n in rows)
          + f" | total {sum(ms for ms, _ in prof.values()) / a.steps:.3f} ms")


if __name__ == "__main__":
    main()
