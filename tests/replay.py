"""Shared by the CPU and GPU parity tests: the reference's FOMAML outer loop (src/fo_meta_interface.py:128-177,253-298)
driven with the ORACLE's arithmetic (oracle/ref_cpu.py) over the product's DataContainer, so that one function yields,
for the literal BASELINE config-3 run (4 accents, meta_k inner steps, shipped Noam schedule), every run_batch info, every
meta-gradient and every meta-weight tensor in full.  TEST INFRASTRUCTURE (imports oracle/)."""
import random
from collections import OrderedDict

import torch

from oracle import ref_cpu


def oracle_fomaml_run(cfg, data_container, meta_k, meta_batch_size, max_step, eps, dev_max_ilen=3000, on_batch=None, init_sd=None,
                      algo="fomaml"):
    """-> dict(calls=[(accent, train, info)], steps=[(meta_grad, meta_after_adam)], lr, evals=[[(logit, gold)]])
    Call order = the reference's: per meta-step and task the k inner-step batches, then the val batch; after the meta-step
    of every eval_ival-th global step one eval call per dev batch per accent (on the LAST task's adapted weights, Q2).
    `init_sd`: the initial meta weights (default: ref_cpu.deterministic_state_dict seed 7); `algo`: "fomaml" or "reptile"
    (the latter = ref_cpu.reptile_meta_step, parity unpinned: the reference raises ValueError for it)."""
    mcfg = cfg["asr_model"]
    eval_ival = cfg["solver"]["eval_ival"]
    odim = 367
    init_sd = init_sd if init_sd is not None else ref_cpu.deterministic_state_dict(mcfg, odim, 7)
    meta = OrderedDict((n, t.clone()) for n, t in init_sd.items())
    meta_step_fn = {"fomaml": ref_cpu.fomaml_meta_step, "reptile": ref_cpu.reptile_meta_step}[algo]
    if mcfg["tgt_share_weight"]:
        meta["pre_embed.weight"] = meta["char_trans.weight"]
    adam_state, calls, steps, evals = {}, [], [], []
    task_ids = list(range(data_container.num_datasets))
    global_step, meta_step, lr = 1, 0, None

    def note(accent, train, batch, info):
        calls.append((int(accent), train, info))
        if on_batch is not None:
            on_batch(len(calls) - 1, int(accent), train, batch)

    while global_step < max_step:
        for _ in range(eval_ival):
            random.shuffle(task_ids)
            tasks, metas = [], []
            for a in task_ids[:meta_batch_size]:
                tr = [b for _, b in data_container.get_item(a, meta_k)]
                val = data_container.get_item(a)[0][1]
                tasks.append(([(x, il, ys, ol.clone()) for x, il, ys, ol in tr], (val[0], val[1], val[2], val[3].clone())))
                metas.append((a, tr, val))
            keep = {}
            meta_step += 1
            infos, lr = meta_step_fn(meta, mcfg, tasks, eps, adam_state, meta_step, keep=keep)
            for (a, tr, val), inner, vinfo in zip(metas, keep["inner_infos"], infos):
                for b, info in zip(tr, inner):
                    note(a, True, b, info)
                note(a, True, val, vinfo)
            steps.append((keep["meta_grad"], OrderedDict((n, t.detach().clone()) for n, t in meta.items())))
            if global_step % eval_ival == 0:
                outs = []
                for a, loader in enumerate(data_container.dev_loaders):
                    for x, il, ys, ol in loader:
                        if il.max() > dev_max_ilen:
                            continue
                        info, logit, gold = ref_cpu.run_batch_eval(keep["last_adapted"], mcfg, (x, il, ys, ol.clone()), eps)
                        note(a, False, (x, il, ys, ol), info)
                        outs.append((logit, gold))
                evals.append(outs)
            global_step += 1
    return {"calls": calls, "steps": steps, "lr": lr, "evals": evals, "global_step": global_step}


def eight_accent_setup(tmp_path, golden_dir):
    """the 8-accent workspace + the product's DataContainer + the initial weights `pretrain.py` starts from at seed 531 (the
    loaders' base-seed draws of load_data() come first, then the seed-exact init replay -- checked against the golden's
    fingerprints of the REFERENCE's own initial weights)"""
    import random
    import numpy as np
    import masr_amd  # noqa: F401
    from masr_amd.io.dataset import DataContainer
    from masr_amd.model import reference_init_state_dict
    from oracle.make_goldens import EIGHT_ACCENTS, eight_workspace
    cfg, ft = eight_workspace(tmp_path, golden_dir)
    sv = cfg["solver"]
    random.seed(531); np.random.seed(531); torch.manual_seed(531)
    dc = DataContainer([tmp_path / "data" / a for _, a in EIGHT_ACCENTS], batch_size=sv["batch_size"], dev_batch_size=sv["dev_batch_size"],
                       is_memmap=True, is_bucket=True, min_ilen=sv["min_ilen"], max_ilen=sv["max_ilen"], half_batch_ilen=sv["half_batch_ilen"])
    init = reference_init_state_dict(cfg["asr_model"], 367)
    init["pos_encoder.pe"] = ref_cpu.sinusoid_pe(3000, cfg["asr_model"]["d_model"])
    return cfg, ft, dc, init
