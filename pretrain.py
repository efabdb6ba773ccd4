#!/usr/bin/env python3
"""pretrain.py -- drop-in CLI of the reference's pretrain.py:19-88 on the MI355X-native path.

Same flags, same YAML schema, same data/ and testing-logs/ layout.  Extra: launch under
`python -m torch.distributed.run --nproc-per-node N pretrain.py ...` to shard the accent-tasks of each meta-step
one-per-GPU (RCCL all-reduce of the meta-gradient); `--hbm_shards` keeps the feature shards resident in HBM."""
import argparse
import datetime
import json
import os
import random
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")            # --tasks_per_gpu > 3 needs more than the default 4 HIP hardware queues

import numpy as np
import torch
import yaml

import masr_amd  # noqa: F401
from masr_amd.marcos import AVAIL_ACCENTS
from masr_amd.utils import setup_host_threads, usable_cpus
from masr_amd.parallel import TaskSharder


def build_parser():
    p = argparse.ArgumentParser(description='Accent-Adaptative ASR pretraining scripts (MI355X-native)')
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--pretrain_suffix', type=str, required=True)
    p.add_argument('--pretrain_accents', type=str, nargs='+', choices=AVAIL_ACCENTS)
    p.add_argument('--num_pretrain', type=int, required=True)
    p.add_argument('--tgt_accent', type=str, choices=AVAIL_ACCENTS)
    p.add_argument('--runs', default=0, type=int)
    p.add_argument('--overwrite', action='store_true')
    p.add_argument('--seed', default=531, type=int)
    p.add_argument('--no_cuda', action='store_true')
    p.add_argument('--no_memmap', action='store_true')
    p.add_argument('--no_bucket', action='store_true')
    p.add_argument('--meta_k', default=None, type=int)
    p.add_argument('--meta_batch_size', default=None, type=int)
    p.add_argument('--sample_strategy', default='normal', choices=['normal', 'meta-split', 'meta-split-dev'])
    p.add_argument('--max_step', default=0, type=int)
    p.add_argument('--resume', action='store_true')
    p.add_argument('--resume_step', default=-1, type=int)
    p.add_argument('--use_tensorboard', action='store_true')
    p.add_argument('--model_name', default='transformer', choices=['blstm', 'transformer'])
    p.add_argument('--algo', choices=['reptile', 'fomaml', 'multi', 'maml'], required=True)
    p.add_argument('--njobs', default=-1, type=int)
    # MI355X-path extras (all default to the reference's behaviour)
    p.add_argument('--hbm_shards', action='store_true', help='keep the fbank shards resident in HBM (GPU gather+pad)')
    p.add_argument('--tasks_per_gpu', type=int, default=1, help='FOMAML: accent-tasks of a meta-step run concurrently on one GPU '
                   '(replica + HIP stream + host thread each); results are identical to 1 (tasks are independent)')
    p.add_argument('--no_slot_cap', action='store_true', help='several ranks: keep --tasks_per_gpu above 3 even when a rank runs several waves of '
                   'tasks per meta-step (their all-reduce then competes with the next wave for the four hardware queues)')
    p.add_argument('--sync_stats', action='store_true', help='FOMAML: read every task\'s loss / accuracy / gradient norm back before the next '
                   'task is queued (the reference\'s timing of its log lines); default: the host runs one meta-step ahead of the GPU and books '
                   'them then -- same numbers, same order')
    p.add_argument('--fix_snapshot_meta_weights', action='store_true', help='save the META weights in snapshots (reference saves the last task\'s adapted weights)')
    p.add_argument('--fix_nan_meta_grad', action='store_true', help='a val-batch gradient with a NaN norm contributes zeros to the meta update (the reference accumulates the NaNs)')
    p.add_argument('--fix_reptile', action='store_true', help='run --algo reptile with the published pseudo-gradient (the reference raises ValueError for it)')
    return p


def main(argv=None):
    paras = build_parser().parse_args(argv)
    paras.pretrain_suffix = paras.pretrain_suffix or "{:%B%d-%H%M%S}".format(datetime.datetime.now())
    paras.cuda, paras.is_bucket, paras.is_memmap = not paras.no_cuda, not paras.no_bucket, not paras.no_memmap
    mbs = paras.num_pretrain if paras.meta_batch_size is None else paras.meta_batch_size
    assert mbs <= paras.num_pretrain, f"Meta batch size {mbs} > Number of pretraining accents {paras.num_pretrain}"
    paras.meta_batch_size = mbs
    paras.njobs = paras.njobs if paras.njobs > 0 else usable_cpus()
    setup_host_threads(paras.njobs)
    config = yaml.safe_load(open(paras.config))

    TaskSharder.init_process_group()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MASR_DIST_BACKEND") == "gloo" and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()                      # gloo rehearsal: more ranks than GPUs share the cards
    paras.device = f"cuda:{local}"
    paras.hbm_shards_device = paras.device if paras.hbm_shards else None

    random.seed(paras.seed)
    np.random.seed(paras.seed)
    torch.manual_seed(paras.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(paras.seed)

    with open(Path('data', 'accent-code.json')) as fin:
        id2accent = json.load(fin)
    if paras.algo == 'multi':
        from masr_amd.multi_interface import MultiASRInterface as ASRInterface
    elif paras.algo in ('fomaml', 'reptile'):
        from masr_amd.fo_meta_interface import FOMetaASRInterface as ASRInterface
    else:
        raise NotImplementedError
    if paras.model_name != 'transformer':
        raise NotImplementedError
    from masr_amd.transformer_torch_trainer import get_trainer

    solver = get_trainer(ASRInterface, config, paras, id2accent)
    solver.load_data()
    solver.set_model()
    solver.exec()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # with several ranks a failed rank must end at once (the others wait for it in a collective; worker threads or a half-issued exchange
        # could otherwise keep this interpreter alive and the launcher would never tear the job down)
        import os
        import sys
        import traceback
        traceback.print_exc()
        sys.stderr.flush(); sys.stdout.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            os._exit(1)
        raise
