#!/usr/bin/env python3
"""train.py -- drop-in CLI of the reference's train.py:22-127 (mono-accent training / fine-tuning) on the
MI355X-native path.  `--test` runs the greedy decoder (masr_recog) and writes best-hyp like src/tester.py."""
import argparse
import json
import os
import random
from pathlib import Path

import numpy as np
import torch
import yaml

import masr_amd  # noqa: F401
from masr_amd.marcos import AVAIL_ACCENTS
from masr_amd.utils import setup_host_threads, usable_cpus


def build_parser():
    p = argparse.ArgumentParser(description='Accent-Adaptative ASR training scripts (MI355X-native)')
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--accent', choices=AVAIL_ACCENTS, required=True)
    p.add_argument('--algo', choices=['reptile', 'fomaml', 'multi', 'maml', 'no'], required=True)
    p.add_argument('--model_name', default='transformer', choices=['blstm', 'las', 'transformer'])
    p.add_argument('--eval_suffix', type=str, default=None)
    p.add_argument('--runs', type=int, default=0)
    p.add_argument('--overwrite', action='store_true')
    p.add_argument('--seed', default=531, type=int)
    p.add_argument('--no_cuda', action='store_true')
    p.add_argument('--no_memmap', action='store_true')
    p.add_argument('--no_bucket', action='store_true')
    p.add_argument('--resume', action='store_true')
    p.add_argument('--use_tensorboard', action='store_true')
    p.add_argument('--save_verbose', action='store_true')
    p.add_argument('--split_rate', type=float, default=1.0)
    p.add_argument('--freeze_layer', type=str, default=None, choices=['VGG', 'VGG_BLSTM'])
    p.add_argument('--pretrain', action='store_true')
    p.add_argument('--pretrain_suffix', type=str, default=None)
    p.add_argument('--pretrain_setting', type=str, default=None)
    p.add_argument('--pretrain_runs', type=int, default=0)
    p.add_argument('--pretrain_step', type=int, default=0)
    p.add_argument('--pretrain_tgt_accent', choices=AVAIL_ACCENTS, default='wa')
    p.add_argument('--pretrain_model_path', type=str, default=None)
    p.add_argument('--test', action='store_true')
    p.add_argument('--test_model', type=str, default='model.wer.best')
    p.add_argument('--decode_batch_size', type=int, default=1)
    p.add_argument('--decode_mode', choices=['greedy', 'beam', 'lm_beam'], default='greedy')
    p.add_argument('--decode_suffix', default=None, type=str)
    p.add_argument('--lm_model_path', default=None, type=str)
    p.add_argument('--njobs', default=-1, type=int)
    p.add_argument('--hbm_shards', action='store_true')
    p.add_argument('--sync_stats', action='store_true', help='read loss / accuracy / gradient norm back every step (default: the NaN test runs on the device, the host '
                   'queues the next step meanwhile and books them a step later -- same weights, same numbers)')
    return p


def main(argv=None):
    paras = build_parser().parse_args(argv)
    paras.cuda, paras.is_bucket, paras.is_memmap = not paras.no_cuda, not paras.no_bucket, not paras.no_memmap
    paras.njobs = paras.njobs if paras.njobs > 0 else usable_cpus()
    setup_host_threads(paras.njobs)
    paras.eval_suffix = paras.eval_suffix or "default"
    paras.device = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
    paras.hbm_shards_device = paras.device if paras.hbm_shards else None
    config = yaml.safe_load(open(paras.config))
    random.seed(paras.seed)
    np.random.seed(paras.seed)
    torch.manual_seed(paras.seed)
    with open(Path('data', 'accent-code.json')) as fin:
        id2accent = json.load(fin)
    if paras.test:
        from masr_amd.tester import Tester
        paras.decode_suffix = f"{paras.decode_mode}_decode" + (f"_{paras.decode_suffix}" if paras.decode_suffix else "")   # train.py:83
        solver = Tester(config, paras, id2accent)
        solver.load_data(); solver.set_model(); solver.exec()
        return
    from masr_amd.mono_interface import MonoASRInterface
    if paras.model_name == 'blstm':                               # train.py:112-113 of the reference
        from masr_amd.blstm_trainer import get_trainer
    elif paras.model_name == 'transformer':
        from masr_amd.transformer_torch_trainer import get_trainer
    else:
        raise NotImplementedError(f"model_name {paras.model_name} (LAS is not on the MI355X path)")
    solver = get_trainer(MonoASRInterface, config, paras, id2accent)
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
