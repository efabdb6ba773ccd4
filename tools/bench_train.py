#!/usr/bin/env python3
"""End-to-end `train.py` (mono-accent, BASELINE configs[1]) throughput WITH the data path: one synthetic accent shard on disk
(N utterances x 1000 frames x 80 dims), hkust geometry, Noam-Adam, through get_trainer(MonoASRInterface...) as train.py
drives it; wall-clock utterances/s of the epoch loop (evaluation off).

    python tools/bench_train.py [--utts 2048] [--steps 100] [--out gpurun_out/e2e_train.json]"""
import argparse
import json
import os
import random
import shutil
import sys
import time
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from bench_pretrain import write_shard


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warm", type=int, default=10)
    ap.add_argument("--njobs", type=int, default=8)
    ap.add_argument("--optimizer", default="noam", choices=["noam", "SGD"])
    ap.add_argument("--root", default="/tmp/masr_e2e_train")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import masr_amd  # noqa: F401
    from masr_amd.mono_interface import MonoASRInterface
    from masr_amd.transformer_torch_trainer import get_trainer
    root = Path(args.root)
    if not (root / "data" / "us" / "dev" / "label.npy").exists():
        shutil.rmtree(root, ignore_errors=True)
        write_shard(root / "data" / "us" / "train", args.utts, 1000, 80, 0)
        write_shard(root / "data" / "us" / "dev", 16, 1000, 80, 100)
        for f in ("toy_spm.model", "toy_spm_units.txt"):
            shutil.copy(ROOT / "tests" / "golden" / f, root / "data" / f)
    os.chdir(root)
    model = {"idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.1, "pos_dropout": 0.1, "tgt_share_weight": 1,
             "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}}
    if args.optimizer == "noam":
        model.update({"optimizer_cls": "noam", "optimizer_opt": {"k": 1.0, "warmup_steps": 25000}})
    else:
        model.update({"optimizer_cls": "SGD", "optimizer_opt": {"lr": 0.01, "momentum": 0.9, "nesterov": True}})
    n_steps = args.warm + args.steps
    cfg = {"asr_model": model,
           "solver": {"setting": "e2e", "data_root": "data", "total_epochs": 1000, "spm_mapping": "data/toy_spm_units.txt", "spm_model": "data/toy_spm.model",
                      "label_smoothing": 0.2, "eval_ival": 10 ** 9, "log_ival": 10 ** 9, "save_ival": 10 ** 9, "batch_size": 32, "dev_batch_size": 16,
                      "min_ilen": 10, "max_ilen": 1500, "dev_max_ilen": 3000, "half_batch_ilen": 512}}
    paras = SimpleNamespace(accent="us", algo="no", model_name="transformer", eval_suffix="e", runs=0, overwrite=True, seed=531, resume=False,
                            use_tensorboard=False, save_verbose=False, split_rate=1.0, freeze_layer=None, pretrain=False, pretrain_suffix=None,
                            pretrain_setting=None, pretrain_runs=0, pretrain_step=0, pretrain_tgt_accent=None, pretrain_model_path=None,
                            njobs=args.njobs, is_bucket=True, is_memmap=True, device="cuda:0", eval_every_epoch=False)
    random.seed(531); np.random.seed(531); torch.manual_seed(531)
    s = get_trainer(MonoASRInterface, cfg, paras, {"us": "us"})
    s.load_data(); s.set_model()
    s.evaluate = lambda: None
    s.save_per_epoch = lambda: None
    stamps, utts = [], [0]
    orig = s.run_batch

    class Done(Exception):
        pass

    def rb(idx, x, ilens, ys, olens, train, **kw):
        if len(stamps) in (args.warm, n_steps):
            torch.cuda.synchronize()
        stamps.append((time.perf_counter(), utts[0]))
        if len(stamps) > n_steps:
            raise Done
        utts[0] += len(ys)
        return orig(idx, x, ilens, ys, olens, train=train, **kw)
    from functools import partial
    s._train = partial(rb, train=True)
    try:
        s.exec()
    except Done:
        pass
    (ta, ua), (tb, ub) = stamps[args.warm], stamps[n_steps]
    res = {"workload": f"train.py mono-accent, 1 accent x {args.utts} utt x 1000 frames x 80 dims, B=16 (half-batch rule), hkust geometry, dropout 0.1, {args.optimizer}",
           "steps": args.steps, "utt": ub - ua, "seconds": tb - ta, "utt_per_s": (ub - ua) / (tb - ta), "ms_per_step": (tb - ta) / args.steps * 1e3}
    line = json.dumps(res)
    print(line)
    if args.out:
        Path(ROOT / args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
