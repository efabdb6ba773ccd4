#!/usr/bin/env python3
"""End-to-end `pretrain.py --algo fomaml` throughput WITH the real data path: synthetic shards on disk in the
reference's layout (N utterances x 1000 frames x 80 dims per accent, NPY-format feat.dat opened by memmap) ->
BucketSampler -> collate -> upload -> run_task / val batch / meta-update, through get_trainer(FOMetaASRInterface...) exactly as
pretrain.py drives it.  bench.py times batches that are already resident in HBM; this tool reports the wall-clock utterances/s of
the whole loop next to it, with the shards on the host (memmap + pinned upload) and resident in HBM (--hbm_shards).

    python tools/bench_pretrain.py [--utts 4096] [--meta-steps 40] [--out gpurun_out/e2e.json]

Reference data path: src/io/dataset.py:21-33,116-198,248-277; loop: src/fo_meta_interface.py:128-177."""
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ACC = [("af", "african"), ("au", "australia"), ("en", "england"), ("us", "us")]


def write_shard(d, n, T, D, seed):
    from numpy.lib.format import open_memmap
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.RandomState(seed)
    gen = np.random.default_rng(seed)                             # float32 normals without a float64 detour
    ilens = np.full(n, T, dtype=np.int64)
    olens = rng.randint(10, 41, size=n).astype(np.int64)
    feat = open_memmap(d / "feat.dat", mode="w+", dtype=np.float32, shape=(n * T, D))
    for i in range(0, n, 256):                                    # bounded host memory
        m = min(256, n - i)
        feat[i * T:(i + m) * T] = gen.standard_normal((m * T, D), dtype=np.float32)
    feat.flush()
    del feat
    np.save(d / "ilens.npy", ilens)
    np.save(d / "olens.npy", olens)
    np.save(d / "label.npy", rng.randint(1, 366, size=int(olens.sum())).astype(np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--idim", type=int, default=80)
    ap.add_argument("--meta-steps", type=int, default=40)
    ap.add_argument("--warm", type=int, default=5)
    ap.add_argument("--njobs", type=int, default=8, help="collate threads (pretrain.py --njobs; its default is the usable core count)")
    ap.add_argument("--root", default="/tmp/masr_e2e")
    ap.add_argument("--out", default=None)
    ap.add_argument("--configs", default="host:1,host:4,hbm:1,hbm:4", help="comma list of <host|hbm>:<tasks_per_gpu>")
    ap.add_argument("--sync-stats", action="store_true", help="pretrain.py --sync_stats: one host sync per task (the host cannot run ahead of the GPU)")
    args = ap.parse_args()

    import masr_amd  # noqa: F401
    from masr_amd.fo_meta_interface import FOMetaASRInterface
    from masr_amd.transformer_torch_trainer import get_trainer

    root = Path(args.root)
    t0 = time.perf_counter()
    if not (root / "data" / ACC[-1][1] / "dev" / "label.npy").exists():
        shutil.rmtree(root, ignore_errors=True)
        for ai, (_, a) in enumerate(ACC):
            write_shard(root / "data" / a / "train", args.utts, args.frames, args.idim, ai)        # numpy seed 0 + accent index
            write_shard(root / "data" / a / "dev", 16, args.frames, args.idim, 100 + ai)
        for f in ("toy_spm.model", "toy_spm_units.txt"):
            shutil.copy(ROOT / "tests" / "golden" / f, root / "data" / f)
    print(f"shards ready in {time.perf_counter() - t0:.1f} s ({args.utts} utt x {args.frames} frames x {args.idim} dims per accent)", file=sys.stderr, flush=True)
    os.chdir(root)
    model = {"idim": args.idim, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.1, "pos_dropout": 0.1, "tgt_share_weight": 1,
             "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}, "inner_optimizer_cls": "SGD",
             "inner_optimizer_opt": {"momentum": 0.9, "nesterov": True}, "meta_opt_cls": "noam",
             "meta": {"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}}}
    n_steps = args.warm + args.meta_steps
    cfg = {"asr_model": model,
           "solver": {"setting": "e2e", "data_root": "data", "total_steps": n_steps + 1, "spm_mapping": "data/toy_spm_units.txt",
                      "spm_model": "data/toy_spm.model", "label_smoothing": 0.2, "eval_ival": n_steps, "log_ival": 10 ** 9, "save_ival": 10 ** 9,
                      "batch_size": 32, "dev_batch_size": 16, "min_ilen": 10, "max_ilen": 1500, "dev_max_ilen": 3000, "half_batch_ilen": 512}}
    results = []
    for spec in args.configs.split(","):
        where, k = spec.split(":")
        paras = SimpleNamespace(config="x", pretrain_suffix=f"{where}{k}", pretrain_accents=[c for c, _ in ACC], num_pretrain=4, tgt_accent="ca",
                                runs=0, overwrite=True, seed=531, meta_k=1, meta_batch_size=4, sample_strategy="normal", max_step=n_steps + 1,
                                resume=False, model_name="transformer", algo="fomaml", njobs=args.njobs, is_bucket=True, is_memmap=True,
                                use_tensorboard=False, device="cuda:0", tasks_per_gpu=int(k),
                                hbm_shards_device="cuda:0" if where == "hbm" else None, sync_stats=args.sync_stats)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(FOMetaASRInterface, cfg, paras, dict(ACC + [("ca", "canada")]))
        tl = time.perf_counter()
        solver.load_data()
        t_load = time.perf_counter() - tl
        solver.set_model()
        solver.evaluate = lambda: None                            # (the dev pass is not part of the training throughput)
        stamps, utts = [], [0]
        orig_rb, orig_final = solver.run_batch, solver._final_meta_update

        def rb(idx, x, ilens, ys, olens, train, accent_idx=None, **kw):
            utts[0] += len(ys)
            return orig_rb(idx, x, ilens, ys, olens, train=train, accent_idx=accent_idx, **kw)

        def final(n_tasks=None):
            orig_final(n_tasks)
            if len(stamps) + 1 in (args.warm, n_steps):              # the two ends of the timed region only: in between the host
                torch.cuda.synchronize()                             # runs ahead of the GPU as it does in a real run
            stamps.append((time.perf_counter(), utts[0]))
        from functools import partial
        solver._train = partial(rb, train=True)
        solver._final_meta_update = final
        hs0 = dict(torch.cuda.host_memory_stats()) if hasattr(torch.cuda, "host_memory_stats") else {}
        solver.exec()
        hs1 = dict(torch.cuda.host_memory_stats()) if hs0 else {}
        (ta, ua), (tb, ub) = stamps[args.warm - 1], stamps[-1]
        res = {"shards": where, "tasks_per_gpu": int(k), "meta_steps": len(stamps) - args.warm, "utt": ub - ua, "seconds": tb - ta,
               "utt_per_s": (ub - ua) / (tb - ta), "ms_per_meta_step": (tb - ta) / (len(stamps) - args.warm) * 1e3, "load_data_s": t_load,
               "pinned_allocs_during_run": {k: hs1[k] - hs0.get(k, 0) for k in hs1 if ("num_host_alloc" in k or "host_alloc_time.total" in k or k == "allocated_bytes.allocated")} if hs0 else None}
        print(json.dumps(res), file=sys.stderr, flush=True)
        results.append(res)
        del solver
        torch.cuda.empty_cache()
    out = {"workload": f"pretrain.py --algo fomaml, 4 accents x {args.utts} utt x {args.frames} frames x {args.idim} dims, meta_k 1, B=16 (half-batch rule), "
                       "hkust geometry, dropout 0.1; per meta-step 4 tasks x (1 inner step + 1 val batch)", "results": results}
    line = json.dumps(out)
    print(line)
    if args.out:
        Path(ROOT / args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
