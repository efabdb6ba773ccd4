"""Diagnostic (not collected): the noam/k=1/warmup-6 fine-tune of test_finetune_with_the_host_running_ahead..., printing loss, gradient norm
and the largest weight per step -- does the run diverge (loss and weights grow) before a NaN shows, or does the NaN come out of the blue?"""
import os, random, sys
from types import SimpleNamespace
from functools import partial
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tempfile, pathlib
import masr_amd  # noqa
from test_hip_misc import _common, ODIM
from oracle import ref_cpu
from masr_amd.mono_interface import MonoASRInterface
from masr_amd.transformer_torch_trainer import get_trainer
tmp = pathlib.Path(tempfile.mkdtemp()); os.chdir(tmp)
cfg, id2accent = _common(tmp, {"optimizer_cls": "noam", "optimizer_opt": {"k": float(os.environ.get("K", "1.0")), "warmup_steps": 6}})
cfg["solver"]["eval_ival"] = 5
if os.environ.get("NODROP"): cfg["asr_model"]["dropout"] = 0.0
snap = tmp / "pre.snapshot"
torch.save(ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7), snap)
paras = SimpleNamespace(accent="af", algo="fomaml", model_name="transformer", eval_suffix="e", runs=0, overwrite=True, seed=531,
                        resume=False, use_tensorboard=False, save_verbose=False, split_rate=1.0, freeze_layer=None, pretrain=True,
                        pretrain_suffix="p", pretrain_setting=None, pretrain_runs=0, pretrain_step=0, pretrain_tgt_accent="ca",
                        pretrain_model_path=str(snap), njobs=2, is_bucket=True, is_memmap=True, device="cuda:0", sync_stats=True)
random.seed(531); np.random.seed(531); torch.manual_seed(531)
s = get_trainer(MonoASRInterface, cfg, paras, id2accent)
s.load_data(); s.set_model()
orig = s.run_batch
def rb(*a, **k):
    r = orig(*a, **k)
    if k.get('train'): print('   batch', tuple(a[1].shape) if hasattr(a[1], 'shape') else None, 'ilens', [int(v) for v in a[2]], 'olens', [int(v) for v in a[4]] if len(a) > 4 else None)
    if k.get("train"):
        e = s.asr_model.engine
        g = e.grads
        print(f"step {s.global_step}: loss {r['loss'] if isinstance(r, dict) else r} |g| {float(g.norm()):.4g} finite-g {bool(torch.isfinite(g).all())} max|w| {float(e.params.abs().max()):.3f} lr {s.asr_opt.lr:.4f}")
        if not torch.isfinite(g).all() or float(g.norm()) > 1e3:
            for n, (off, shape) in e.table.items():
                k_ = int(np.prod(shape)); t = g[off:off + k_]
                if not torch.isfinite(t).all() or float(t.abs().max()) > 1e2: print("   bad grad:", n, int((~torch.isfinite(t)).sum()), "non-finite of", k_, "max", float(t.abs().max()))
    return r
s._train = partial(rb, train=True)
s.evaluate = lambda: None
s.exec()
