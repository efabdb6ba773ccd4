#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference on CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference,
which never travels to the GPU box).  Nothing from the reference is copied: this
script imports it, feeds it seeded inputs and stores inputs' recipes + outputs.

    python oracle/make_goldens.py            # rewrites tests/golden/

Harness = SURVEY.md Appendix C: sys.modules stubs for the missing third-party
logging/dashboard modules, `.cuda()` -> identity.
"""
import json
import os
import random
import shutil
import sys
import tempfile
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
REF = Path(os.environ.get("MASR_REFERENCE", "/root/reference"))
OUT = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))

from oracle import ref_cpu  # noqa: E402  (shared deterministic init + synthetic data recipes)


class RunningAvgDict(dict):
    """Stand-in (true torchexp semantics unknown -> smoothed logs are 'parity unpinned')."""
    def __init__(self, decay_rate=1.0):
        super().__init__()
        self.decay_rate = decay_rate
        self._n = {}
    def add(self, info, n=1):
        for k, v in info.items():
            v = float(v)
            if k not in self:
                self[k] = v
                self._n[k] = n
            elif self.decay_rate >= 1.0:
                tot = self._n[k] + n
                self[k] = (self[k] * self._n[k] + v * n) / tot
                self._n[k] = tot
            else:
                self[k] = self.decay_rate * self[k] + (1 - self.decay_rate) * v

# --------------------------------------------------------------------------- #
# stubs (Appendix C)
# --------------------------------------------------------------------------- #
def install_stubs():
    tl = types.ModuleType("tqdmlogger")
    tl.log = lambda *a, **k: None
    tl.seclog = lambda *a, **k: None
    tl.flush = lambda *a, **k: None
    tl.logger = SimpleNamespace(info=lambda *a, **k: None)
    an = types.ModuleType("tqdmlogger.ansistyle")
    an.stylize = lambda s, *a, **k: s
    an.fg = an.bg = an.attr = lambda *a, **k: ""
    an.RESET = ""
    tl.ansistyle = an
    sys.modules["tqdmlogger"] = tl
    sys.modules["tqdmlogger.ansistyle"] = an

    te = types.ModuleType("torchexp")
    tes = types.ModuleType("torchexp.stat")
    tes.RunningAvgDict = RunningAvgDict
    te.stat = tes
    sys.modules["torchexp"] = te
    sys.modules["torchexp.stat"] = tes

    sys.modules["torch_optimizer"] = types.ModuleType("torch_optimizer")

    cm = types.ModuleType("comet_ml")
    class _Exp:
        def __init__(self, *a, **k): self.alive = True
        def get_key(self): return "stub"
        def __getattr__(self, name): return lambda *a, **k: None
    cm.Experiment = cm.ExistingExperiment = _Exp
    sys.modules["comet_ml"] = cm

    ed = types.ModuleType("editdistance")
    def _lev(a, b):
        a, b = list(a), list(b)
        prev = list(range(len(b) + 1))
        for i, x in enumerate(a, 1):
            cur = [i]
            for j, y in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
            prev = cur
        return prev[-1]
    ed.eval = _lev
    sys.modules["editdistance"] = ed
    ip = types.ModuleType("IPython")
    ip.embed = lambda *a, **k: None
    sys.modules["IPython"] = ip

    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self


TINY = {  # tiny transformer used for all model goldens (dropout 0 for parity)
    "idim": 83, "nheads": 4, "d_model": 64, "d_inner": 128, "dropout": 0.0, "pos_dropout": 0.0,
    "tgt_share_weight": 1, "encoder": {"nlayers": 2}, "decoder": {"nlayers": 2},
    "inner_optimizer_cls": "SGD", "inner_optimizer_opt": {"momentum": 0.9, "nesterov": True},
    "meta_opt_cls": "noam", "meta": {"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}},
}
ODIM = 367


def synth_batch(seed, ilens, olens, idim=83):
    """Seeded synthetic batch (shared recipe with tests): feat ~ N(0,1), labels U{1..365}."""
    g = torch.Generator().manual_seed(seed)
    B, T = len(ilens), max(ilens)
    xs = torch.zeros(B, T, idim)
    for b, n in enumerate(ilens):
        xs[b, :n] = torch.randn(n, idim, generator=g)
    ys = [torch.randint(1, 366, (n,), generator=g) for n in olens]
    return xs, torch.tensor(ilens), ys, torch.tensor(olens)


def flat_checks(t):
    """small fingerprint of a tensor: sum, abs-sum, l2, first/last 4 values."""
    t = t.detach().double().reshape(-1)
    return np.array([t.sum(), t.abs().sum(), t.norm(), *t[:4].tolist(), *t[-4:].tolist()])


def build_ref_model(cfg, sd):
    from src.model.transformer_pytorch.mono_transformer_torch import MyTransformer
    m = MyTransformer(["x"] * ODIM, cfg)
    missing = m.load_state_dict(sd)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def ref_run_batch(model, batch, eps):
    """The reference's own run_batch arithmetic, by calling its trainer class body:
    we instantiate the mixin on a minimal Interface so the REAL run_batch code runs."""
    from src.transformer_torch_trainer import get_trainer

    class _Iface:
        def __init__(self, config, paras, id2accent):
            self.global_step = 1          # %500 != 0 -> no probe
        def load_model(self): pass
    solver = get_trainer(_Iface, None, None, None)
    solver.asr_model = model
    solver.label_smooth_rate = eps
    solver.asr_opt = SimpleNamespace(zero_grad=lambda: [setattr(p, "grad", None) for p in model.parameters()])
    xs, il, ys, ol = batch
    info = solver.run_batch(0, xs, il.clone(), [y.clone() for y in ys], ol.clone(), train=True)
    return info


def gen_model_goldens():
    cfg = TINY
    sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=7)
    out = {}
    cases = {
        "ragged": ([64, 52, 40, 33], [9, 7, 5, 3]),
        "same": ([48, 48, 48], [6, 6, 4]),
        "single": ([37], [5]),
    }
    for cname, (ilens, olens) in cases.items():
        for eps in (0.2, 0.0):
            model = build_ref_model(cfg, sd)
            model.train()
            batch = synth_batch(11, ilens, olens)
            xs, il, ys, ol = batch
            with torch.no_grad():
                logit, gold = model(xs, il.clone(), [y.clone() for y in ys], ol.clone())
            info = ref_run_batch(model, batch, eps)
            key = f"{cname}_eps{eps}"
            out[f"{key}/logit"] = logit.numpy()
            out[f"{key}/gold"] = gold.numpy()
            out[f"{key}/loss"] = np.float64(info["loss"])
            out[f"{key}/acc"] = np.float64(info["acc"])
            named = dict(model.named_parameters())
            for n, p in named.items():
                out[f"{key}/gradfp/{n}"] = flat_checks(p.grad)
            # a few full gradients (small tensors) for element-wise checks
            for n in ("feat_extractor.0.weight", "feat_extractor.0.bias", "char_trans.bias",
                      "encoder.layers.0.norm1.weight", "decoder.layers.1.multihead_attn.in_proj_bias",
                      "encoder.norm.bias", "decoder.layers.0.self_attn.out_proj.weight"):
                out[f"{key}/grad/{n}"] = named[n].grad.numpy().copy()
            if cname == "ragged" and eps == 0.2:
                # inner step: clip + SGD(nesterov) x2 with the reference's optimizer objects
                opt = torch.optim.SGD(model.parameters(), lr=ref_cpu.inner_lr(cfg) * 1000, momentum=0.9, nesterov=True)
                gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
                opt.step()
                out["inner/gradnorm0"] = np.float64(gn)
                batch2 = synth_batch(12, ilens, olens)
                solver_info = ref_run_batch(model, batch2, eps)
                gn2 = torch.nn.utils.clip_grad_norm_(model.parameters(), 5)
                opt.step()
                out["inner/loss1"] = np.float64(solver_info["loss"])
                out["inner/gradnorm1"] = np.float64(gn2)
                for n, p in model.named_parameters():
                    out[f"inner/paramfp/{n}"] = flat_checks(p)
                out["inner/param/vgg2enc.bias"] = named["vgg2enc.bias"].detach().numpy().copy()
                out["inner/param/char_trans.weight"] = named["char_trans.weight"].detach().numpy().copy()
            if cname == "ragged" and eps == 0.2:
                model = build_ref_model(cfg, sd)          # fresh weights (the inner steps above moved them)
                model.eval()
                with torch.no_grad():
                    hyp = model.recog(xs, il.clone())
                out["recog/hyp"] = hyp.numpy()
                with torch.no_grad():
                    logit_eval, _ = model(xs, il.clone(), [y.clone() for y in ys], ol.clone())
                out["eval/logit"] = logit_eval.numpy()
    np.savez_compressed(OUT / "model_tiny.npz", **out)
    print("model_tiny.npz", len(out), "arrays")


def gen_masks_noam():
    from src.nets_utils import make_bool_pad_mask, generate_square_subsequent_mask
    from src.model.transformer_pytorch.optimizer import TransformerOptimizer
    out = {}
    out["pad_mask"] = make_bool_pad_mask(torch.tensor([7, 3, 5])).numpy()
    out["causal5"] = generate_square_subsequent_mask(5).numpy()
    w = torch.nn.Parameter(torch.zeros(3))
    opt = TransformerOptimizer(torch.optim.Adam([w], betas=(0.9, 0.98), eps=1e-9), 1.0, 512, 25000)
    lrs = []
    for _ in range(12):
        w.grad = torch.ones(3)
        opt.step()
        lrs.append(opt.lr)
    out["noam_lr_512_25000"] = np.array(lrs)
    opt = TransformerOptimizer(torch.optim.Adam([w], betas=(0.9, 0.98), eps=1e-9), 0.7, 64, 4)
    lrs = []
    for _ in range(12):
        w.grad = torch.ones(3)
        opt.step()
        lrs.append(opt.lr)
    out["noam_lr_64_4_k0.7"] = np.array(lrs)
    # Adam trajectory on a fixed gradient sequence (pins beta/eps/bias-correction semantics)
    g = torch.Generator().manual_seed(3)
    w = torch.nn.Parameter(torch.randn(16, generator=g))
    out["adam_w0"] = w.detach().numpy().copy()
    opt = TransformerOptimizer(torch.optim.Adam([w], betas=(0.9, 0.98), eps=1e-9), 1.0, 64, 4)
    gs = []
    for _ in range(5):
        gr = torch.randn(16, generator=g)
        gs.append(gr.numpy().copy())
        w.grad = gr.clone()
        opt.step()
    out["adam_grads"] = np.stack(gs)
    out["adam_w5"] = w.detach().numpy().copy()
    np.savez_compressed(OUT / "masks_noam.npz", **out)
    print("masks_noam.npz")


def write_toy_shard(root, accent, split, n_utt, seed, idim=83, lo=12, hi=40):
    """Appendix D layout: feat.dat (NPY format), ilens/olens/label .npy."""
    from numpy.lib.format import open_memmap
    rng = np.random.RandomState(seed)
    d = Path(root) / accent / split
    d.mkdir(parents=True, exist_ok=True)
    ilens = rng.randint(lo, hi, size=n_utt).astype(np.int64)
    olens = rng.randint(2, 6, size=n_utt).astype(np.int64)
    feat = open_memmap(d / "feat.dat", mode="w+", dtype=np.float32, shape=(int(ilens.sum()), idim))
    feat[:] = rng.randn(int(ilens.sum()), idim).astype(np.float32)
    feat.flush()
    del feat
    np.save(d / "ilens.npy", ilens)
    np.save(d / "olens.npy", olens)
    np.save(d / "label.npy", rng.randint(1, 366, size=int(olens.sum())).astype(np.int64))
    return ilens, olens


def write_learnable_shard(root, accent, split, n_utt, seed, idim=83, n_class=8):
    """A toy shard a tiny model can actually learn within a few hundred steps (peaked logits -> arg-max decisions that
    survive bf16 rounding): every utterance belongs to one of `n_class` classes; a class fixes the label sequence (2..5
    tokens from U{1..365}, a seeded table shared by all shards) and the feature pattern (every frame = the class's
    codebook vector + N(0, 0.1) noise); lengths are random (16..45 frames).  Layout: SURVEY Appendix D."""
    from numpy.lib.format import open_memmap
    tab = np.random.RandomState(4242)
    code = (tab.randn(n_class, idim) * 1.5).astype(np.float32)
    seqs = [tab.randint(1, 366, size=tab.randint(2, 6)).astype(np.int64) for _ in range(n_class)]
    rng = np.random.RandomState(seed)
    d = Path(root) / accent / split
    d.mkdir(parents=True, exist_ok=True)
    cls = rng.randint(0, n_class, size=n_utt)
    ilens = rng.randint(16, 46, size=n_utt).astype(np.int64)
    olens = np.array([len(seqs[c]) for c in cls], dtype=np.int64)
    feat = open_memmap(d / "feat.dat", mode="w+", dtype=np.float32, shape=(int(ilens.sum()), idim))
    row = 0
    for c, n in zip(cls, ilens):
        feat[row:row + n] = code[c] + 0.1 * rng.randn(n, idim).astype(np.float32)
        row += n
    feat.flush()
    del feat
    np.save(d / "ilens.npy", ilens)
    np.save(d / "olens.npy", olens)
    np.save(d / "label.npy", np.concatenate([seqs[c] for c in cls]))
    return ilens, olens


def gen_sampler_goldens():
    from src.io.dataset import BucketSampler
    out = {}
    rng = np.random.RandomState(5)
    ilens = rng.randint(8, 60, size=200)
    out["ilens"] = ilens
    random.seed(531)
    np.random.seed(531)
    s = BucketSampler(ilens, min_ilen=10, max_ilen=50, half_batch_ilen=30, batch_size=4,
                      bucket_size=1, bucket_reverse=False, drop_last=False)
    for ep in range(2):
        batches = list(iter(s))
        out[f"epoch{ep}_flat"] = np.array([i for b in batches for i in b])
        out[f"epoch{ep}_sizes"] = np.array([len(b) for b in batches])
    out["len"] = np.int64(len(s))
    np.savez_compressed(OUT / "bucket_sampler.npz", **out)
    print("bucket_sampler.npz")


def gen_fomaml_goldens():
    """Full reference FOMAML run through get_trainer(FOMetaASRInterface...) on toy shards."""
    import yaml
    from src.fo_meta_interface import FOMetaASRInterface
    from src.transformer_torch_trainer import get_trainer
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        (tmp / "data").mkdir()
        for f in ("accent-code.json", "valid_train_en_unigram150.model", "valid_train_en_unigram150_units.txt"):
            shutil.copy(REF / "data" / f, tmp / "data" / f)     # runtime copy only, temp dir
        accents = ["african", "australia"]
        for ai, a in enumerate(accents):
            write_toy_shard(tmp / "data", a, "train", 16, seed=100 + ai)
            write_toy_shard(tmp / "data", a, "dev", 4, seed=200 + ai)
        cfg = {
            "asr_model": dict(TINY),
            "solver": {"setting": "gold", "data_root": "data", "total_steps": 10,
                       "spm_mapping": "data/valid_train_en_unigram150_units.txt",
                       "spm_model": "data/valid_train_en_unigram150.model",
                       "label_smoothing": 0.2, "eval_ival": 2, "log_ival": 1, "save_ival": 2,
                       "batch_size": 4, "dev_batch_size": 4, "min_ilen": 10, "max_ilen": 50,
                       "dev_max_ilen": 3000, "half_batch_ilen": 30},
        }
        cfg["asr_model"]["meta"]["optimizer_opt"]["warmup_steps"] = 4     # visible meta updates
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        paras = SimpleNamespace(
            config="x", pretrain_suffix="g", pretrain_accents=["af", "au"], num_pretrain=2, tgt_accent="ca",
            runs=0, overwrite=True, seed=531, no_cuda=True, no_memmap=False, no_bucket=False, meta_k=2,
            meta_batch_size=2, sample_strategy="normal", max_step=3, resume=False, resume_step=-1,
            use_tensorboard=False, model_name="transformer", algo="fomaml", njobs=0, cuda=False,
            is_bucket=True, is_memmap=True)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(FOMetaASRInterface, cfg, paras, id2accent)
        solver.load_data()
        solver.set_model()
        # replace the RNG-order-dependent init by the shared deterministic one
        sd = ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7)
        solver.asr_model.load_state_dict(sd)
        solver.load_model()                       # re-clone _original + fresh meta optimizer
        # record every batch the reference feeds to run_batch, and the info it returns
        rec = []
        orig_run_batch = solver.run_batch
        def spy(idx, x, ilens, ys, olens, train, accent_idx=None):
            info = orig_run_batch(idx, x, ilens.clone(), [y.clone() for y in ys], olens.clone(),
                                  train=train, accent_idx=accent_idx)
            rec.append((int(idx), x.numpy().copy(), ilens.numpy().copy(),
                        [y.numpy().copy() for y in ys], olens.numpy().copy(), dict(info)))
            return info
        from functools import partial
        solver._train = partial(spy, train=True)
        solver.exec()
        out = {}
        out["n_calls"] = np.int64(len(rec))
        for i, (idx, x, il, ys, ol, info) in enumerate(rec):
            out[f"call{i}/accent"] = np.int64(idx)
            out[f"call{i}/ilens"] = il
            out[f"call{i}/olens"] = ol
            out[f"call{i}/x_fp"] = flat_checks(torch.from_numpy(x))
            out[f"call{i}/ys"] = np.concatenate(ys)
            out[f"call{i}/loss"] = np.float64(info["loss"])
            out[f"call{i}/acc"] = np.float64(info["acc"])
        for n, p in solver._original.items():
            out[f"meta/fp/{n}"] = flat_checks(p)
        out["meta/param/vgg2enc.bias"] = solver._original["vgg2enc.bias"].detach().numpy().copy()
        out["meta/param/decoder.norm.weight"] = solver._original["decoder.norm.weight"].detach().numpy().copy()
        out["meta/param/char_trans.bias"] = solver._original["char_trans.bias"].detach().numpy().copy()
        out["meta/step_num"] = np.int64(solver.meta_opt.step_num)
        out["meta/lr"] = np.float64(solver.meta_opt.lr)
        out["global_step"] = np.int64(solver.global_step)
        out["inner_lr"] = np.float64(solver.inner_lr)
        snap = torch.load(solver.log_dir / "snapshot.latest") if (solver.log_dir / "snapshot.latest").exists() else None
        out["files"] = np.array(sorted(p.name for p in solver.log_dir.iterdir()))
        for f in solver.log_dir.iterdir():
            if f.name.startswith(("dev_", "train_", "best_")) :
                out[f"log/{f.name}"] = np.array(open(f).read())
        for n, t in snap.items():
            out[f"snap/fp/{n}"] = flat_checks(t)
        out["state_dict_keys"] = np.array(list(solver.asr_model.state_dict().keys()))
        np.savez_compressed(OUT / "fomaml_toy.npz", **out)
        print("fomaml_toy.npz calls:", len(rec), "files:", out["files"])
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


CFG3_ACCENTS = [("af", "african"), ("au", "australia"), ("en", "england"), ("us", "us")]


def cfg3_workspace(root, spm_dir):
    """4 accents x (16 train, 4 dev) toy utterances + the toy SentencePiece model (tests/golden/toy_spm.*).  Shared by the
    golden generator below and by the tests that replay it."""
    root = Path(root)
    (root / "data").mkdir(parents=True, exist_ok=True)
    json.dump(dict(CFG3_ACCENTS + [("ca", "canada")]), open(root / "data" / "accent-code.json", "w"))
    shutil.copy(Path(spm_dir) / "toy_spm.model", root / "data" / "toy_spm.model")
    shutil.copy(Path(spm_dir) / "toy_spm_units.txt", root / "data" / "toy_spm_units.txt")
    for ai, (_, a) in enumerate(CFG3_ACCENTS):
        write_toy_shard(root / "data", a, "train", 16, seed=400 + ai)
        write_toy_shard(root / "data", a, "dev", 4, seed=500 + ai)
    model = dict(TINY)
    model["meta"] = {"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}}       # the shipped fometa-hkust.yaml schedule
    solver = {"setting": "cfg3", "data_root": "data", "total_steps": 10, "spm_mapping": "data/toy_spm_units.txt",
              "spm_model": "data/toy_spm.model", "label_smoothing": 0.2, "eval_ival": 2, "log_ival": 1, "save_ival": 2,
              "batch_size": 4, "dev_batch_size": 4, "min_ilen": 10, "max_ilen": 50, "dev_max_ilen": 3000, "half_batch_ilen": 30}
    return {"asr_model": model, "solver": solver}


def cfg3_paras(meta_k, **extra):
    p = dict(config="x", pretrain_suffix=f"k{meta_k}", pretrain_accents=[c for c, _ in CFG3_ACCENTS], num_pretrain=4, tgt_accent="ca",
             runs=0, overwrite=True, seed=531, no_cuda=True, no_memmap=False, no_bucket=False, meta_k=meta_k, meta_batch_size=4,
             sample_strategy="normal", max_step=5, resume=False, resume_step=-1, use_tensorboard=False, model_name="transformer",
             algo="fomaml", njobs=0, cuda=False, is_bucket=True, is_memmap=True)
    p.update(extra)
    return SimpleNamespace(**p)


def gen_fomaml_cfg3_goldens():
    """BASELINE configs[2] literally: `pretrain.py --algo fomaml`, 4 accents, inner_steps (meta_k) = 1, the shipped Noam
    schedule (warmup_steps 25000 -> inner lr 7.9e-4 at d_model 64), plus a meta_k = 2 variant.  The reference's
    get_trainer(FOMetaASRInterface...).exec() runs 4 meta-steps with its own evaluate() after the 2nd and 4th (toy
    SentencePiece model, so CER/WER are reproducible on the GPU box).  Captured: EVERY run_batch call, train and eval
    (batch identity, loss, acc, cer, wer); per meta-step the meta-gradient (_updates / _counter, before Adam) and the
    meta weights after Adam, per tensor; every log file; the snapshot."""
    from src.fo_meta_interface import FOMetaASRInterface
    from src.transformer_torch_trainer import get_trainer
    from functools import partial
    out = {}
    for meta_k in (1, 2):
        tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
        cwd = os.getcwd()
        try:
            cfg = cfg3_workspace(tmp, OUT)
            os.chdir(tmp)
            id2accent = json.load(open("data/accent-code.json"))
            paras = cfg3_paras(meta_k)
            random.seed(531); np.random.seed(531); torch.manual_seed(531)
            solver = get_trainer(FOMetaASRInterface, cfg, paras, id2accent)
            solver.load_data()
            solver.set_model()
            solver.asr_model.load_state_dict(ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7))
            solver.load_model()
            rec = []
            orig_run_batch = solver.run_batch

            def spy(idx, x, ilens, ys, olens, train, accent_idx=None):
                info = orig_run_batch(idx, x, ilens.clone(), [y.clone() for y in ys], olens.clone(), train=train, accent_idx=accent_idx)
                rec.append((int(idx), bool(train), x.numpy().copy(), ilens.numpy().copy(), [y.numpy().copy() for y in ys], dict(info)))
                return info
            solver._train = partial(spy, train=True)
            solver._eval = partial(spy, train=False)
            steps = []
            orig_final = solver._final_meta_update

            def final_spy():
                mg = {n: (u / solver._counter).clone() for n, u in solver._updates.items()}
                orig_final()
                steps.append((mg, {n: t.detach().clone() for n, t in solver._original.items()}))
            solver._final_meta_update = final_spy
            solver.exec()
            pre = f"k{meta_k}/"
            out[pre + "n_calls"] = np.int64(len(rec))
            for i, (idx, train, x, il, ys, info) in enumerate(rec):
                out[f"{pre}call{i}/accent"] = np.int64(idx)
                out[f"{pre}call{i}/train"] = np.int64(train)
                out[f"{pre}call{i}/ilens"] = il
                out[f"{pre}call{i}/x_fp"] = flat_checks(torch.from_numpy(x))
                out[f"{pre}call{i}/ys"] = np.concatenate(ys)
                for k, v in info.items():
                    out[f"{pre}call{i}/{k}"] = np.float64(v)
            out[pre + "n_meta_steps"] = np.int64(len(steps))
            for si, (mg, meta) in enumerate(steps):
                for n in mg:
                    out[f"{pre}step{si}/metagrad/fp/{n}"] = flat_checks(mg[n])
                    if n != "pos_encoder.pe" and n in meta:
                        out[f"{pre}step{si}/meta/fp/{n}"] = flat_checks(meta[n])
                for n in ("vgg2enc.bias", "decoder.norm.weight", "char_trans.bias", "feat_extractor.0.weight", "encoder.layers.1.linear2.bias"):
                    out[f"{pre}step{si}/metagrad/full/{n}"] = mg[n].numpy().copy()
            out[pre + "meta/step_num"] = np.int64(solver.meta_opt.step_num)
            out[pre + "meta/lr"] = np.float64(solver.meta_opt.lr)
            out[pre + "global_step"] = np.int64(solver.global_step)
            out[pre + "inner_lr"] = np.float64(solver.inner_lr)
            out[pre + "files"] = np.array(sorted(p.name for p in solver.log_dir.iterdir()))
            for f in solver.log_dir.iterdir():
                if f.name.startswith(("dev_", "train_", "best_")) or f.name == "global_step":
                    out[f"{pre}log/{f.name}"] = np.array(open(f).read())
            snap = torch.load(solver.log_dir / "snapshot.latest")
            for n, t in snap.items():
                out[f"{pre}snap/fp/{n}"] = flat_checks(t)
            print(f"fomaml_cfg3 k={meta_k}: {len(rec)} calls ({sum(1 for r in rec if not r[1])} eval), {len(steps)} meta-steps, files {list(out[pre + 'files'])}")
            print("   dev_avg_wer:", repr(str(out[pre + 'log/dev_avg_wer'])), "best_wer:", repr(str(out[pre + 'log/best_wer'])))
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(OUT / "fomaml_cfg3.npz", **out)
    print("fomaml_cfg3.npz", len(out), "arrays")



def _chain_cfgs():
    """the chain's two YAML-shaped configs (pretrain: cfg3's model; fine-tune: the adapt/*.yaml shape)"""
    base = {"data_root": "data", "spm_mapping": "data/toy_spm_units.txt", "spm_model": "data/toy_spm.model", "dev_max_ilen": 3000,
            "min_ilen": 10, "max_ilen": 60, "half_batch_ilen": 100}
    pre_model = dict(TINY)
    pre_model["meta"] = {"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}}
    pre = {"asr_model": pre_model,
           "solver": dict(base, setting="chain", total_steps=10, label_smoothing=0.1, eval_ival=2, log_ival=1, save_ival=2, batch_size=4,
                          dev_batch_size=4)}
    ft_model = {k: v for k, v in TINY.items() if k not in ("inner_optimizer_cls", "inner_optimizer_opt", "meta_opt_cls", "meta")}
    ft_model.update({"optimizer_cls": "noam", "optimizer_opt": {"k": 0.2, "warmup_steps": 100}})
    ft = {"asr_model": ft_model,
          "solver": dict(base, setting="chain-ft", total_epochs=20, label_smoothing=0.1, eval_ival=50, log_ival=1000, batch_size=8,
                         dev_batch_size=8, pretrain_module=["feat_extractor", "vgg2enc", "char_trans", "pre_embed", "encoder", "decoder"],
                         freeze_module=["feat_extractor"], beam_decode={"beam_size": 1})}
    return pre, ft


def chain_workspace(root, spm_dir):
    """BASELINE configs[4] in miniature: learnable toy shards for 4 pretraining accents and the target accent (canada:
    train / dev / test), the toy SentencePiece model, and the two YAML-shaped configs (pretrain: cfg3's; fine-tune: the
    adapt/*.yaml shape with pretrain_module / freeze_module)."""
    root = Path(root)
    (root / "data").mkdir(parents=True, exist_ok=True)
    json.dump(dict(CFG3_ACCENTS + [("ca", "canada")]), open(root / "data" / "accent-code.json", "w"))
    for f in ("toy_spm.model", "toy_spm_units.txt"):
        shutil.copy(Path(spm_dir) / f, root / "data" / f)
    for ai, (_, a) in enumerate(CFG3_ACCENTS):
        write_learnable_shard(root / "data", a, "train", 32, seed=600 + ai)
        write_learnable_shard(root / "data", a, "dev", 4, seed=700 + ai)
    write_learnable_shard(root / "data", "canada", "train", 96, seed=801, n_class=2)
    write_learnable_shard(root / "data", "canada", "dev", 8, seed=802, n_class=2)
    write_learnable_shard(root / "data", "canada", "test", 12, seed=803, n_class=2)
    return _chain_cfgs()


EIGHT_ACCENTS = [("af", "african"), ("au", "australia"), ("en", "england"), ("us", "us"), ("hk", "hongkong"), ("in", "indian"),
                 ("ir", "ireland"), ("nz", "newzealand")]


def eight_workspace(root, spm_dir):
    """BASELINE configs[3] / configs[4] at toy size: EIGHT pretraining accents (learnable shards, 32 train / 4 dev utterances
    each) + the target accent canada (train / dev / test), the toy SentencePiece model, the pretrain YAML (cfg3's model, the
    shipped Noam schedule) and the chain's fine-tune YAML.  Shared by gen_fomaml_8acc_goldens and tests/test_hip_cfg45.py."""
    root = Path(root)
    (root / "data").mkdir(parents=True, exist_ok=True)
    json.dump(dict(EIGHT_ACCENTS + [("ca", "canada")]), open(root / "data" / "accent-code.json", "w"))
    for f in ("toy_spm.model", "toy_spm_units.txt"):
        shutil.copy(Path(spm_dir) / f, root / "data" / f)
    for ai, (_, a) in enumerate(EIGHT_ACCENTS):
        write_learnable_shard(root / "data", a, "train", 32, seed=900 + ai)
        write_learnable_shard(root / "data", a, "dev", 4, seed=950 + ai)
    write_learnable_shard(root / "data", "canada", "train", 96, seed=801, n_class=2)
    write_learnable_shard(root / "data", "canada", "dev", 8, seed=802, n_class=2)
    write_learnable_shard(root / "data", "canada", "test", 12, seed=803, n_class=2)
    pre, ft = _chain_cfgs()
    pre["solver"]["setting"], ft["solver"]["setting"] = "eight", "eight-ft"
    return pre, ft


def eight_paras(algo="fomaml", meta_k=1, **extra):
    p = vars(cfg3_paras(meta_k, pretrain_suffix="eight", pretrain_accents=[c for c, _ in EIGHT_ACCENTS], num_pretrain=8, meta_batch_size=8,
                        max_step=3, algo=algo))
    p.update(extra)
    return SimpleNamespace(**p)


def gen_fomaml_8acc_goldens():
    """BASELINE configs[3] as far as one process goes: `pretrain.py --algo fomaml --num_pretrain 8 --meta_batch_size 8 --meta_k 1`
    run by the REFERENCE (its own seed-531 initialisation after load_data(), as its CLI does): two meta-steps of eight tasks,
    evaluate() on the eight dev sets after the second.  Captured: every run_batch call (batch identity, loss, acc, cer, wer), per
    meta-step the fingerprints of the meta-gradient and of the post-Adam meta weights (+ a few small tensors in full), the
    fingerprint of every initial tensor, every log file, the snapshot."""
    from src.fo_meta_interface import FOMetaASRInterface
    from src.transformer_torch_trainer import get_trainer
    from functools import partial
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    out = {}
    try:
        cfg, _ = eight_workspace(tmp, OUT)
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(FOMetaASRInterface, cfg, eight_paras(), id2accent)
        solver.load_data(); solver.set_model()
        for n, t in solver._original.items():
            out[f"init/fp/{n}"] = flat_checks(t)
        rec, steps = [], []
        orig = solver.run_batch

        def spy(idx, x, ilens, ys, olens, train, accent_idx=None):
            info = orig(idx, x, ilens.clone(), [y.clone() for y in ys], olens.clone(), train=train, accent_idx=accent_idx)
            rec.append((int(idx), bool(train), ilens.numpy().copy(), np.concatenate([y.numpy() for y in ys]), dict(info)))
            return info
        solver._train, solver._eval = partial(spy, train=True), partial(spy, train=False)
        orig_final = solver._final_meta_update

        def final_spy():
            mg = {n: (u / solver._counter).clone() for n, u in solver._updates.items()}
            orig_final()
            steps.append((mg, {n: t.detach().clone() for n, t in solver._original.items()}))
        solver._final_meta_update = final_spy
        solver.exec()
        out["n_calls"] = np.int64(len(rec))
        for i, (idx, train, il, ys, info) in enumerate(rec):
            out[f"call{i}/accent"], out[f"call{i}/train"], out[f"call{i}/ilens"], out[f"call{i}/ys"] = np.int64(idx), np.int64(train), il, ys
            for k, v in info.items():
                out[f"call{i}/{k}"] = np.float64(v)
        out["n_meta_steps"] = np.int64(len(steps))
        for si, (mg, meta) in enumerate(steps):
            for n in mg:
                out[f"step{si}/metagrad/fp/{n}"] = flat_checks(mg[n])
                out[f"step{si}/meta/fp/{n}"] = flat_checks(meta[n])
            for n in ("vgg2enc.bias", "decoder.norm.weight", "char_trans.bias", "feat_extractor.0.weight", "encoder.layers.1.linear2.bias"):
                out[f"step{si}/metagrad/full/{n}"] = mg[n].numpy().copy()
        out["meta/step_num"], out["meta/lr"] = np.int64(solver.meta_opt.step_num), np.float64(solver.meta_opt.lr)
        out["global_step"], out["inner_lr"] = np.int64(solver.global_step), np.float64(solver.inner_lr)
        out["files"] = np.array(sorted(p.name for p in solver.log_dir.iterdir()))
        for f in solver.log_dir.iterdir():
            if f.name.startswith(("dev_", "train_", "best_")) or f.name == "global_step":
                out[f"log/{f.name}"] = np.array(open(f).read())
        for n, t in torch.load(solver.log_dir / "snapshot.latest").items():
            out[f"snap/fp/{n}"] = flat_checks(t)
        np.savez_compressed(OUT / "fomaml_8acc.npz", **out)
        print(f"fomaml_8acc.npz: {len(rec)} calls ({sum(1 for r in rec if not r[1])} eval), {len(steps)} meta-steps, {len(out)} arrays")
        print("   task order:", [r[0] for r in rec if r[1]][::2], "dev_avg_wer", repr(str(out["log/dev_avg_wer"])))
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def chain_paras(stage, **extra):
    """argparse namespaces of the three CLI calls of the chain: pretrain.py --algo X / train.py --pretrain ... / train.py --test"""
    if stage == "pretrain":
        p = vars(cfg3_paras(1, pretrain_suffix="chain"))
    else:
        p = dict(config="x", accent="ca", algo="fomaml", model_name="transformer", eval_suffix="ft", runs=0, overwrite=True, seed=531,
                 resume=False, use_tensorboard=False, save_verbose=False, split_rate=1.0, freeze_layer=None, pretrain=True,
                 pretrain_suffix="chain", pretrain_setting="chain", pretrain_runs=0, pretrain_step=4, pretrain_tgt_accent="ca",
                 pretrain_model_path=None, njobs=0, is_bucket=True, is_memmap=True, no_cuda=True, cuda=False, test=False,
                 eval_every_epoch=False)
        if stage == "test":
            p.update(test=True, test_model="model.wer.best", decode_suffix="greedy_decode", decode_mode="greedy", decode_batch_size=4,
                     njobs=1, lm_model_path=None)
    p.update(extra)
    return SimpleNamespace(**p)


def gen_chain_goldens():
    """BASELINE configs[4] chain on toy data, run by the REFERENCE: pretrain.py --algo fomaml (4 accents) -> snapshot.step.4
    -> train.py fine-tune on the target accent (pretrain_module / freeze_module, Noam-Adam, 20 epochs, evaluate() every 50
    steps) -> train.py --test (Tester, greedy, batch 4) -> best-hyp.  Captured: the fine-tune's per-call losses, its dev logs
    and file set, the best-hyp lines."""
    from functools import partial
    from src.fo_meta_interface import FOMetaASRInterface
    from src.mono_interface import MonoASRInterface
    from src.tester import Tester
    from src.transformer_torch_trainer import get_trainer
    from src.io.dataset import get_loader
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        pre_cfg, ft_cfg = chain_workspace(tmp, OUT)
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        out = {}
        # ---- 1. pretrain
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(FOMetaASRInterface, pre_cfg, chain_paras("pretrain"), id2accent)
        solver.load_data(); solver.set_model()                  # the reference's own seed-531 initialisation, as its CLI does
        out["pre/init/fp/vgg2enc.weight"] = flat_checks(solver._original["vgg2enc.weight"])
        rec = []
        orig = solver.run_batch

        def spy(idx, x, ilens, ys, olens, train, accent_idx=None):
            info = orig(idx, x, ilens.clone(), [y.clone() for y in ys], olens.clone(), train=train, accent_idx=accent_idx)
            rec.append((bool(train), dict(info)))
            return info
        solver._train, solver._eval = partial(spy, train=True), partial(spy, train=False)
        solver.exec()
        out["pre/loss"] = np.array([r[1]["loss"] for r in rec])
        out["pre/train"] = np.array([r[0] for r in rec])
        snap = torch.load(solver.log_dir / "snapshot.step.4")
        for n, t in snap.items():
            out[f"pre/snap/fp/{n}"] = flat_checks(t)
        for f in solver.log_dir.iterdir():
            if f.name.startswith(("dev_", "train_", "best_")):
                out[f"pre/log/{f.name}"] = np.array(open(f).read())
        # ---- 2. fine-tune from the snapshot (path built by TrainInterface from the pretrain_* flags, train_interface.py:64-68)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        ft = get_trainer(MonoASRInterface, ft_cfg, chain_paras("finetune"), id2accent)
        ft.load_data(); ft.set_model()
        rec2 = []
        orig2 = ft.run_batch

        def spy2(idx, x, ilens, ys, olens, train, accent_idx=None):
            info = orig2(idx, x, ilens.clone(), [y.clone() for y in ys], olens.clone(), train=train, accent_idx=accent_idx)
            rec2.append((bool(train), ilens.numpy().copy(), np.concatenate([y.numpy() for y in ys]), dict(info)))
            return info
        ft._train, ft._eval = partial(spy2, train=True), partial(spy2, train=False)
        ft.exec()
        out["ft/n_calls"] = np.int64(len(rec2))
        out["ft/train"] = np.array([r[0] for r in rec2])
        out["ft/loss"] = np.array([r[3]["loss"] for r in rec2])
        out["ft/acc"] = np.array([r[3]["acc"] for r in rec2])
        for i in range(40):                                   # batch identity of the first calls (the order is RNG-driven)
            out[f"ft/call{i}/ilens"], out[f"ft/call{i}/ys"] = rec2[i][1], rec2[i][2]
        out["ft/global_step"], out["ft/ep"] = np.int64(ft.global_step), np.int64(ft.ep)
        out["ft/files"] = np.array(sorted(p.name for p in ft.log_dir.iterdir()))
        for f in ft.log_dir.iterdir():
            if f.name.startswith(("dev_", "train_", "best_")):
                out[f"ft/log/{f.name}"] = np.array(open(f).read())
        frozen = torch.load(ft.log_dir / "snapshot.latest")["feat_extractor.0.weight"]
        assert torch.equal(frozen, snap["feat_extractor.0.weight"])
        # ---- 3. decode the test shard with model.wer.best
        t = Tester(ft_cfg, chain_paras("test"), id2accent)
        t.load_data()
        t.eval_set = get_loader(t.data_dir.joinpath("test"), batch_size=4, half_batch_ilen=512, is_memmap=True, is_bucket=False,
                                shuffle=False, num_workers=0)
        t.set_model()
        t.exec()
        lines = (t.decode_dir / "best-hyp").read_text().splitlines()
        out["test/lines"] = np.array(lines)
        np.savez_compressed(OUT / "chain_toy.npz", **out)
        print("chain_toy.npz: pretrain calls", len(rec), "fine-tune calls", len(rec2), "global_step", ft.global_step)
        print("   dev_acc:", str(out["ft/log/dev_acc"]).replace("\n", " | ")[-300:])
        print("   best_wer:", str(out["ft/log/best_wer"]).strip(), "| best-hyp:", sum(l.split("\t")[0] == l.split("\t")[1] for l in lines), "of", len(lines), "exact")
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)



def _toy_workspace(tmp):
    (tmp / "data").mkdir()
    for f in ("accent-code.json", "valid_train_en_unigram150.model", "valid_train_en_unigram150_units.txt"):
        shutil.copy(REF / "data" / f, tmp / "data" / f)     # runtime copy only, temp dir
    for ai, a in enumerate(["african", "australia"]):
        write_toy_shard(tmp / "data", a, "train", 16, seed=100 + ai)
        write_toy_shard(tmp / "data", a, "dev", 4, seed=200 + ai)
    return {"setting": "gold", "data_root": "data", "total_steps": 10, "total_epochs": 2,
            "spm_mapping": "data/valid_train_en_unigram150_units.txt", "spm_model": "data/valid_train_en_unigram150.model",
            "label_smoothing": 0.2, "eval_ival": 3, "log_ival": 1, "save_ival": 3, "batch_size": 4, "dev_batch_size": 4,
            "min_ilen": 10, "max_ilen": 50, "dev_max_ilen": 3000, "half_batch_ilen": 30}


def _record_calls(solver):
    rec = []
    orig = solver.run_batch
    def spy(idx, x, ilens, ys, olens, train, accent_idx=None):
        info = orig(idx, x, ilens.clone(), [y.clone() for y in ys], olens.clone(), train=train, accent_idx=accent_idx)
        rec.append((int(idx), x.numpy().copy(), ilens.numpy().copy(), [y.numpy().copy() for y in ys], olens.numpy().copy(), dict(info)))
        return info
    from functools import partial
    solver._train = partial(spy, train=True)
    return rec


def _dump_calls(out, rec):
    out["n_calls"] = np.int64(len(rec))
    for i, (idx, x, il, ys, ol, info) in enumerate(rec):
        out[f"call{i}/accent"] = np.int64(idx)
        out[f"call{i}/ilens"] = il
        out[f"call{i}/x_fp"] = flat_checks(torch.from_numpy(x))
        out[f"call{i}/ys"] = np.concatenate(ys)
        out[f"call{i}/loss"] = np.float64(info["loss"])
        out[f"call{i}/acc"] = np.float64(info["acc"])


def gen_multi_goldens():
    """Reference multi-task pretraining (src/multi_interface.py:94-140) through get_trainer(MultiASRInterface...):
    7 steps of random-accent batches, clip 5, Noam-Adam on the model itself."""
    from src.multi_interface import MultiASRInterface
    from src.transformer_torch_trainer import get_trainer
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        solver_cfg = _toy_workspace(tmp)
        model = {k: v for k, v in TINY.items() if k not in ("inner_optimizer_cls", "inner_optimizer_opt", "meta_opt_cls", "meta")}
        model.update({"optimizer_cls": "noam", "optimizer_opt": {"k": 1.0, "warmup_steps": 20}})
        cfg = {"asr_model": model, "solver": solver_cfg}
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        paras = SimpleNamespace(config="x", pretrain_suffix="m", pretrain_accents=["af", "au"], num_pretrain=2, tgt_accent="ca",
                                runs=0, overwrite=True, seed=531, no_cuda=True, no_memmap=False, no_bucket=False, meta_k=None,
                                meta_batch_size=None, sample_strategy="normal", max_step=7, resume=False, resume_step=-1,
                                use_tensorboard=False, model_name="transformer", algo="multi", njobs=0, cuda=False,
                                is_bucket=True, is_memmap=True)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(MultiASRInterface, cfg, paras, id2accent)
        solver.load_data()
        solver.set_model()
        sd = ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7)
        solver.asr_model.load_state_dict(sd)
        solver.evaluate = lambda: None                      # (the dev pass is covered by the FOMAML golden run)
        rec = _record_calls(solver)
        solver.exec()
        out = {}
        _dump_calls(out, rec)
        out["global_step"] = np.int64(solver.global_step)
        out["step_num"] = np.int64(solver.asr_opt.step_num)
        out["lr"] = np.float64(solver.asr_opt.lr)
        for n in ("vgg2enc.bias", "decoder.norm.weight", "char_trans.bias", "encoder.layers.0.linear1.bias"):
            out[f"param/{n}"] = solver.asr_model.state_dict()[n].detach().numpy().copy()
        out["files"] = np.array(sorted(p.name for p in solver.log_dir.iterdir()))
        np.savez_compressed(OUT / "multi_toy.npz", **out)
        print("multi_toy.npz calls:", len(rec), "global_step", solver.global_step, "files:", out["files"])
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_mono_goldens():
    """Reference fine-tuning (train.py path: src/train_interface.py + src/mono_interface.py) through
    get_trainer(MonoASRInterface...): initialised from a pretraining snapshot, feat_extractor frozen, SGD(nesterov),
    two epochs over the toy shard."""
    from src.mono_interface import MonoASRInterface
    from src.transformer_torch_trainer import get_trainer
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        solver_cfg = _toy_workspace(tmp)
        solver_cfg.update({"eval_ival": 1000, "log_ival": 1000, "save_ival": 3, "freeze_module": ["feat_extractor"],
                           "pretrain_module": ["encoder", "decoder", "feat_extractor", "vgg2enc", "char_trans", "pre_embed"]})
        model = {k: v for k, v in TINY.items() if k not in ("inner_optimizer_cls", "inner_optimizer_opt", "meta_opt_cls", "meta")}
        model.update({"optimizer_cls": "SGD", "optimizer_opt": {"lr": 0.05, "momentum": 0.9, "nesterov": True}})
        cfg = {"asr_model": model, "solver": solver_cfg}
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        sd = ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7)
        torch.save(sd, tmp / "pre.snapshot")
        paras = SimpleNamespace(config="x", accent="af", algo="fomaml", model_name="transformer", eval_suffix="e", runs=0, overwrite=True,
                                seed=531, resume=False, use_tensorboard=False, save_verbose=False, split_rate=1.0, freeze_layer=None,
                                pretrain=True, pretrain_suffix="p", pretrain_setting=None, pretrain_runs=0, pretrain_step=0,
                                pretrain_tgt_accent="ca", pretrain_model_path=str(tmp / "pre.snapshot"), njobs=0, is_bucket=True,
                                is_memmap=True, no_cuda=True, cuda=False, test=False, eval_every_epoch=False)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(MonoASRInterface, cfg, paras, id2accent)
        solver.load_data()
        solver.set_model()
        solver.evaluate = lambda: None
        rec = _record_calls(solver)
        solver.exec()
        out = {}
        _dump_calls(out, rec)
        out["global_step"] = np.int64(solver.global_step)
        out["ep"] = np.int64(solver.ep)
        st = solver.asr_model.state_dict()
        for n in ("vgg2enc.bias", "decoder.norm.weight", "char_trans.bias", "encoder.layers.0.linear1.bias", "feat_extractor.2.bias"):
            out[f"param/{n}"] = st[n].detach().numpy().copy()
        out["files"] = np.array(sorted(p.name for p in solver.log_dir.iterdir()))
        np.savez_compressed(OUT / "mono_toy.npz", **out)
        print("mono_toy.npz calls:", len(rec), "global_step", solver.global_step, "ep", solver.ep, "files:", out["files"])
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_tester_goldens():
    """Reference decode path (train.py --test: src/tester.py Tester.load_data/set_model/exec, greedy, batch 4) on a toy
    test shard with the deterministic tiny model: the best-hyp file, line by line."""
    from src.tester import Tester
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        solver_cfg = _toy_workspace(tmp)
        solver_cfg["beam_decode"] = {"beam_size": 1}
        write_toy_shard(tmp / "data", "african", "test", 6, seed=300)
        model = dict(TINY)
        cfg = {"asr_model": model, "solver": solver_cfg}
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        log_dir = tmp / "testing-logs" / "evaluation" / "gold" / "no" / "ev" / "ev" / "african" / "0"
        log_dir.mkdir(parents=True)
        (log_dir / "exp_key").write_text("stub\n")
        sd = ref_cpu.deterministic_state_dict(model, ODIM, seed=7)
        torch.save(sd, log_dir / "model.wer.best")
        paras = SimpleNamespace(accent="af", algo="no", pretrain_suffix=None, eval_suffix="ev", runs=0, model_name="transformer",
                                test_model="model.wer.best", decode_suffix="greedy_decode", decode_mode="greedy", decode_batch_size=4,
                                cuda=False, njobs=1, resume=False, overwrite=True, is_memmap=True, lm_model_path=None)
        t = Tester(cfg, paras, id2accent)
        t.load_data()
        # in-process loading (the reference hard-codes num_workers=1; a worker process adds nothing to the result)
        from src.io.dataset import get_loader
        t.eval_set = get_loader(t.data_dir.joinpath("test"), batch_size=4, half_batch_ilen=512, is_memmap=True, is_bucket=False,
                                shuffle=False, num_workers=0)
        t.set_model()
        t.exec()
        lines = (log_dir / "greedy_decode" / "best-hyp").read_text().splitlines()
        np.savez_compressed(OUT / "tester_toy.npz", lines=np.array(lines))
        print("tester_toy.npz", len(lines), "lines; first:", lines[0][:80])
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


BLSTM_TINY = {"optimizer": {"type": "SGD"}, "optimizer_opt": {"lr": 0.01, "momentum": 0.9, "nesterov": True},
              "encoder": {"idim": 83, "enc_dim": 40, "proj_dim": 40, "odim": 40, "sample_rate": "1_1", "dropout": "0_0"}}


def gen_blstm_goldens():
    """Reference MonoBLSTM (config/blstm geometry, small widths) + the loss of BLSTMTrainer.run_batch on a ragged batch."""
    from oracle import blstm_cpu
    from src.model.blstm.mono_blstm import MonoBLSTM
    id2char = ["<blank>"] + [f"u{i}" for i in range(1, 366)] + ["</s>"]
    model = MonoBLSTM(id2char, BLSTM_TINY)
    sd = blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11)
    assert list(model.state_dict().keys()) == list(sd.keys()), (list(model.state_dict().keys()), list(sd.keys()))
    model.load_state_dict(sd)
    model.train()
    out = {"state_dict_keys": np.array(list(sd.keys()))}
    for tag, (ilens, olens) in {"ragged": ([61, 50, 38, 30], [7, 5, 4, 3]), "single": ([45], [6])}.items():
        xs, il, ys, ol = synth_batch(21, ilens, olens)
        # BLSTMTrainer.run_batch (src/blstm_trainer.py:55-70), executed verbatim on the reference model
        sos = ys[0].new([model.sos_id]); eos = ys[0].new([model.eos_id])
        y_true = torch.cat([torch.cat([sos, y, eos], dim=0) for y in ys])
        pred, enc_lens = model(xs, il)
        logp = torch.nn.functional.log_softmax(pred, dim=-1)
        loss = torch.nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True)(logp.transpose(0, 1).contiguous(), y_true, enc_lens.cpu().long(), (ol + 2).long())
        model.zero_grad()
        loss.backward()
        out[f"{tag}/loss"] = np.float64(loss.item())
        out[f"{tag}/logits"] = pred.detach().numpy().copy()
        out[f"{tag}/enc_lens"] = enc_lens.numpy().copy()
        for n, prm in model.named_parameters():
            out[f"{tag}/grad/{n}"] = flat_checks(prm.grad)
        out[f"{tag}/gradfull/head.bias"] = model.head.bias.grad.numpy().copy()
        out[f"{tag}/gradfull/encoder.vgg.0.weight"] = model.encoder.vgg[0].weight.grad.numpy().copy()
        out[f"{tag}/gradfull/encoder.blstm.rnn0.weight_hh_l0_reverse"] = model.encoder.blstm.rnn0.weight_hh_l0_reverse.grad.numpy().copy()
        gn = torch.sqrt(sum((prm.grad ** 2).sum() for prm in model.parameters()))
        out[f"{tag}/grad_norm"] = np.float64(gn.item())
    # the reference's own initialisation (seed 531 of the CLIs), fingerprinted per tensor, for the init replay of the build
    torch.manual_seed(531)
    m2 = MonoBLSTM(id2char, BLSTM_TINY)
    for n, t in m2.state_dict().items():
        out[f"init/{n}"] = flat_checks(t)
    np.savez_compressed(OUT / "blstm_tiny.npz", **out)
    print("blstm_tiny.npz loss", out["ragged/loss"], "grad norm", out["ragged/grad_norm"])


BLSTM_SUB = {"optimizer": {"type": "SGD"}, "optimizer_opt": {"lr": 0.01, "momentum": 0.9, "nesterov": True},
             "encoder": {"idim": 83, "enc_dim": 40, "proj_dim": 40, "odim": 40, "sample_rate": "1_2_2", "dropout": "0_0_0"}}


def gen_blstm_subsample_goldens():
    """The reference's RNNP with time sub-sampling between the BLSTM layers (sample_rate 1_2_2: ys_pad[:, ::sub], enc_lens -> (len + 1) // sub,
    src/modules/encoder.py:118-121) + the loss of BLSTMTrainer.run_batch; lengths chosen so that odd and even frame counts both occur."""
    from oracle import blstm_cpu
    from src.model.blstm.mono_blstm import MonoBLSTM
    id2char = ["<blank>"] + [f"u{i}" for i in range(1, 366)] + ["</s>"]
    model = MonoBLSTM(id2char, BLSTM_SUB)
    sd = blstm_cpu.deterministic_state_dict(BLSTM_SUB, ODIM, seed=12)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model.train()
    out = {"state_dict_keys": np.array(list(sd.keys()))}
    for tag, (ilens, olens) in {"ragged": ([118, 101, 77, 60], [4, 3, 2, 2]), "single": ([90], [3])}.items():
        xs, il, ys, ol = synth_batch(23, ilens, olens)
        sos = ys[0].new([model.sos_id]); eos = ys[0].new([model.eos_id])
        y_true = torch.cat([torch.cat([sos, y, eos], dim=0) for y in ys])
        pred, enc_lens = model(xs, il)
        logp = torch.nn.functional.log_softmax(pred, dim=-1)
        loss = torch.nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True)(logp.transpose(0, 1).contiguous(), y_true, enc_lens.cpu().long(), (ol + 2).long())
        model.zero_grad()
        loss.backward()
        out[f"{tag}/loss"] = np.float64(loss.item())
        out[f"{tag}/logits"] = pred.detach().numpy().copy()
        out[f"{tag}/enc_lens"] = enc_lens.numpy().copy()
        for n, prm in model.named_parameters():
            out[f"{tag}/grad/{n}"] = flat_checks(prm.grad)
        out[f"{tag}/gradfull/head.bias"] = model.head.bias.grad.numpy().copy()
        out[f"{tag}/gradfull/encoder.blstm.bt1.bias"] = model.encoder.blstm.bt1.bias.grad.numpy().copy()
        out[f"{tag}/gradfull/encoder.blstm.rnn2.weight_hh_l0_reverse"] = model.encoder.blstm.rnn2.weight_hh_l0_reverse.grad.numpy().copy()
        gn = torch.sqrt(sum((prm.grad ** 2).sum() for prm in model.parameters()))
        out[f"{tag}/grad_norm"] = np.float64(gn.item())
    np.savez_compressed(OUT / "blstm_sub.npz", **out)
    print("blstm_sub.npz loss", out["ragged/loss"], "enc_lens", out["ragged/enc_lens"], "logits", out["ragged/logits"].shape)


def gen_blstm_mono_goldens():
    """BASELINE configs[0]: train.py mono-accent, config/blstm CTC on a toy memmap shard -- the reference's
    get_trainer(MonoASRInterface...) from src/blstm_trainer.py, two epochs of clip-5 + SGD(nesterov) steps."""
    import torch.optim.lr_scheduler as lrs
    _orig = lrs.ReduceLROnPlateau
    class _RLROP(_orig):                                    # torch >= 2.7 dropped the `verbose` kwarg the reference passes
        def __init__(self, *a, verbose=None, **k): super().__init__(*a, **k)
    lrs.ReduceLROnPlateau = _RLROP
    from oracle import blstm_cpu
    from src.mono_interface import MonoASRInterface
    from src.blstm_trainer import get_trainer
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        solver_cfg = _toy_workspace(tmp)
        solver_cfg.update({"eval_ival": 1000, "log_ival": 1000})
        cfg = {"asr_model": dict(BLSTM_TINY), "solver": solver_cfg}
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        paras = SimpleNamespace(config="x", accent="af", algo="no", model_name="blstm", eval_suffix="e", runs=0, overwrite=True,
                                seed=531, resume=False, use_tensorboard=False, save_verbose=False, split_rate=1.0, freeze_layer=None,
                                pretrain=False, pretrain_suffix=None, pretrain_setting=None, pretrain_runs=0, pretrain_step=0,
                                pretrain_tgt_accent="ca", pretrain_model_path=None, njobs=0, is_bucket=True, is_memmap=True,
                                no_cuda=True, cuda=False, test=False, eval_every_epoch=False)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(MonoASRInterface, cfg, paras, id2accent)
        solver.load_data()
        solver.set_model()
        sd = blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11)
        solver.asr_model.load_state_dict(sd)
        solver.evaluate = lambda: None
        rec = []
        orig = solver.run_batch
        def spy(cur_b, x, ilens, ys, olens, train):
            info = orig(cur_b, x, ilens.clone(), [y.clone() for y in ys], olens.clone(), train=train)
            rec.append((int(cur_b), x.numpy().copy(), ilens.numpy().copy(), [y.numpy().copy() for y in ys], olens.numpy().copy(), dict(info)))
            return info
        from functools import partial
        solver._train = partial(spy, train=True)
        solver.exec()
        out = {}
        out["n_calls"] = np.int64(len(rec))
        for i, (idx, x, il, ys, ol, info) in enumerate(rec):
            out[f"call{i}/accent"] = np.int64(idx)
            out[f"call{i}/ilens"] = il
            out[f"call{i}/ys"] = np.concatenate(ys)
            out[f"call{i}/loss"] = np.float64(info["loss"])
        out["global_step"] = np.int64(solver.global_step)
        out["ep"] = np.int64(solver.ep)
        st = solver.asr_model.state_dict()
        for n in ("head.bias", "encoder.blstm.bt1.bias", "encoder.blstm.rnn0.bias_hh_l0_reverse", "encoder.vgg.7.bias"):
            out[f"param/{n}"] = st[n].detach().numpy().copy()
        out["files"] = np.array(sorted(p.name for p in solver.log_dir.iterdir()))
        np.savez_compressed(OUT / "blstm_mono_toy.npz", **out)
        print("blstm_mono_toy.npz calls:", len(rec), "losses", [round(r[5]["loss"], 4) for r in rec[:3]], "...", round(rec[-1][5]["loss"], 4),
              "files:", out["files"])
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
        lrs.ReduceLROnPlateau = _orig


def gen_blstm_tester_goldens():
    """Reference decode path for the BLSTM model (src/tester.py batch_greedy_decode, blstm branch): best-hyp lines."""
    from oracle import blstm_cpu
    from src.tester import Tester
    tmp = Path(tempfile.mkdtemp(prefix="masr_gold_"))
    cwd = os.getcwd()
    try:
        solver_cfg = _toy_workspace(tmp)
        solver_cfg["beam_decode"] = {"beam_size": 1}
        write_toy_shard(tmp / "data", "african", "test", 6, seed=300)
        cfg = {"asr_model": dict(BLSTM_TINY), "solver": solver_cfg}
        os.chdir(tmp)
        id2accent = json.load(open("data/accent-code.json"))
        log_dir = tmp / "testing-logs" / "evaluation" / "gold" / "no" / "ev" / "ev" / "african" / "0"
        log_dir.mkdir(parents=True)
        (log_dir / "exp_key").write_text("stub\n")
        torch.save(blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11), log_dir / "model.wer.best")
        paras = SimpleNamespace(accent="af", algo="no", pretrain_suffix=None, eval_suffix="ev", runs=0, model_name="blstm",
                                test_model="model.wer.best", decode_suffix="greedy_decode", decode_mode="greedy", decode_batch_size=4,
                                cuda=False, njobs=1, resume=False, overwrite=True, is_memmap=True, lm_model_path=None)
        t = Tester(cfg, paras, id2accent)
        t.load_data()
        from src.io.dataset import get_loader
        t.eval_set = get_loader(t.data_dir.joinpath("test"), batch_size=4, half_batch_ilen=512, is_memmap=True, is_bucket=False,
                                shuffle=False, num_workers=0)
        t.set_model()
        t.exec()
        lines = (log_dir / "greedy_decode" / "best-hyp").read_text().splitlines()
        np.savez_compressed(OUT / "blstm_tester_toy.npz", lines=np.array(lines))
        print("blstm_tester_toy.npz", len(lines), "lines; first:", lines[0][:100])
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


def gen_ctc_goldens():
    """nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True) as called at blstm_trainer.py:22,65-70."""
    out = {}
    g = torch.Generator().manual_seed(9)
    T, B, C = 24, 4, 11
    logits = torch.randn(T, B, C, generator=g, requires_grad=True)
    tl = torch.tensor([5, 3, 7, 1])
    il = torch.tensor([24, 20, 15, 9])
    tgt = torch.randint(1, C, (int(tl.sum()),), generator=g)
    tgt[1] = tgt[0]                                   # repeated label case
    lp = torch.log_softmax(logits, dim=-1)
    loss = torch.nn.CTCLoss(blank=0, reduction="mean", zero_infinity=True)(lp, tgt, il, tl)
    loss.backward()
    out.update(logits=logits.detach().numpy(), targets=tgt.numpy(), il=il.numpy(), tl=tl.numpy(),
               loss=np.float64(loss.item()), grad_logits=logits.grad.numpy())
    # infeasible sample (target longer than input) -> zero_infinity
    logits2 = torch.randn(4, 2, C, generator=g, requires_grad=True)
    tl2 = torch.tensor([6, 2]); il2 = torch.tensor([4, 4])
    tgt2 = torch.randint(1, C, (8,), generator=g)
    loss2 = torch.nn.CTCLoss(blank=0, reduction="mean", zero_infinity=True)(torch.log_softmax(logits2, -1), tgt2, il2, tl2)
    loss2.backward()
    out.update(inf_logits=logits2.detach().numpy(), inf_targets=tgt2.numpy(), inf_il=il2.numpy(), inf_tl=tl2.numpy(),
               inf_loss=np.float64(loss2.item()), inf_grad_logits=logits2.grad.numpy())
    np.savez_compressed(OUT / "ctc.npz", **out)
    print("ctc.npz")


def gen_init_goldens():
    """Reference init under torch.manual_seed(531) (RNG-order sensitive; pins our init replay)."""
    import yaml
    cfg = yaml.safe_load(open(REF / "config/transformer/pretrain/fometa-hkust.yaml"))["asr_model"]
    from src.model.transformer_pytorch.mono_transformer_torch import MyTransformer
    out = {}
    for tag, c in (("hkust", cfg), ("tiny", TINY)):
        torch.manual_seed(531)
        m = MyTransformer(["x"] * ODIM, c)
        sd = m.state_dict()
        out[f"{tag}/keys"] = np.array(list(sd.keys()))
        out[f"{tag}/nparams"] = np.int64(sum(p.numel() for p in m.parameters()))
        for n, t in sd.items():
            if n != "pos_encoder.pe":
                out[f"{tag}/fp/{n}"] = flat_checks(t)
    np.savez_compressed(OUT / "init.npz", **out)
    print("init.npz")


def fullsize_batch(idim, B=16, T=1000, seed=77):
    """the headline bench shape (BASELINE.json configs[1]: B = 16 utterances x 1000 frames x idim, what the shipped half-batch
    rule yields at 10 s): seeded features ~ N(0,1), 10..40 labels per utterance.  Shared by the golden generator and the GPU test."""
    g = torch.Generator().manual_seed(seed + idim)
    xs = torch.randn(B, T, idim, generator=g)
    ilens = torch.full((B,), T, dtype=torch.int64)
    olens = torch.tensor([10 + (7 * i) % 31 for i in range(B)])
    ys = [torch.randint(1, 366, (int(n),), generator=g) for n in olens]
    return xs, ilens, ys, olens


def gen_hkust_fullsize_goldens():
    """The reference model at the HEADLINE shape: config/transformer/pretrain/fometa-hkust.yaml geometry (E512/H8/F2048/2e4d, 24.88 M
    parameters), its own seed-531 initialisation, dropout 0, one B = 16 x T = 1000 batch at idim 80 (BASELINE's wording) and 83 (the
    shipped idim): run_batch(train) -> loss, acc, global gradient norm (clip_grad_norm_'s return), the fingerprint of EVERY parameter
    gradient and a few small gradients in full, then clip 5 + the shipped inner SGD step and the loss of a second run_batch on the
    same batch (the quantity an inner step is judged by).  ~1 minute of CPU."""
    import yaml
    base = yaml.safe_load(open(REF / "config/transformer/pretrain/fometa-hkust.yaml"))["asr_model"]
    from src.model.transformer_pytorch.mono_transformer_torch import MyTransformer
    out = {}
    for idim in (80, 83):
        cfg = dict(base, idim=idim, dropout=0.0, pos_dropout=0.0)
        torch.manual_seed(531)
        model = MyTransformer(["x"] * ODIM, cfg)
        model.train()
        batch = fullsize_batch(idim)
        pre = f"d{idim}/"
        info = ref_run_batch(model, batch, 0.2)
        out[pre + "loss"], out[pre + "acc"] = np.float64(info["loss"]), np.float64(info["acc"])
        out[pre + "n_total"] = np.int64(int((batch[3] + 1).sum()))
        named = dict(model.named_parameters())
        for n, p in named.items():
            out[pre + f"gradfp/{n}"] = flat_checks(p.grad)
        for n in ("feat_extractor.0.weight", "feat_extractor.0.bias", "feat_extractor.2.bias", "feat_extractor.7.bias", "vgg2enc.bias",
                  "char_trans.bias", "encoder.norm.weight", "decoder.norm.bias", "decoder.layers.3.multihead_attn.in_proj_bias"):
            out[pre + f"grad/{n}"] = named[n].grad.numpy().copy()
        opt = torch.optim.SGD(model.parameters(), lr=ref_cpu.inner_lr(cfg), momentum=0.9, nesterov=True)
        out[pre + "grad_norm"] = np.float64(torch.nn.utils.clip_grad_norm_(model.parameters(), 5))
        opt.step()
        info1 = ref_run_batch(model, batch, 0.2)
        out[pre + "loss_after_inner_step"] = np.float64(info1["loss"])
        out[pre + "grad_norm_after_inner_step"] = np.float64(torch.nn.utils.clip_grad_norm_(model.parameters(), 5))
        # a larger step on the same gradient direction makes the loss change large against the 1e-3 tolerance
        print(f"fullsize idim {idim}: loss {info['loss']:.6f} acc {info['acc']:.4f} |g| {out[pre + 'grad_norm']:.4f} -> loss {info1['loss']:.6f}")
    np.savez_compressed(OUT / "hkust_fullsize.npz", **out)
    print("hkust_fullsize.npz", len(out), "arrays")


def gen_metric_goldens():
    """Metric.batch_cal_er (src/monitor/metric.py:36-87) on seeded logits with the reference's sentencepiece model."""
    from src.monitor.metric import Metric
    units = ['<s>'] + [l.rstrip().split(' ')[0] for l in open(REF / "data" / "valid_train_en_unigram150_units.txt")] + ['</s>']
    m = Metric(str(REF / "data" / "valid_train_en_unigram150.model"), units, 0, len(units) - 1)
    g = torch.Generator().manual_seed(21)
    B, L = 6, 12
    gold = torch.full((B, L), -1, dtype=torch.int64)
    pred_ids = torch.randint(1, 366, (B, L), generator=g)
    for b in range(B):
        n = int(torch.randint(3, L - 1, (1,), generator=g))
        gold[b, :n] = torch.randint(1, 366, (n,), generator=g)
        gold[b, n] = 366
        keep = torch.rand(n, generator=g) < 0.6                     # hypotheses share ~60 % of the reference tokens
        pred_ids[b, :n] = torch.where(keep, gold[b, :n], pred_ids[b, :n])
        pred_ids[b, n + int(torch.randint(0, 2, (1,), generator=g))] = 366
    pred_ids[5, 0] = 366                                            # leading </s>: not a stop (discard_ch_after_eos quirk)
    logits = torch.nn.functional.one_hot(pred_ids, 367).float()
    out = {"pred_ids": pred_ids.numpy(), "gold": gold.numpy()}
    out["cer"] = np.float64(m.batch_cal_er(logits, gold, ['att'], ['cer'])['att_cer'])
    out["wer"] = np.float64(m.batch_cal_er(logits, gold, ['att'], ['wer'])['att_wer'])
    out["per_cer"] = np.array([m.cal_att_cer(pred_ids[b], gold[b]) for b in range(B)])
    out["per_wer"] = np.array([m.cal_att_wer(pred_ids[b], gold[b]) for b in range(B)])
    np.savez_compressed(OUT / "metric.npz", **out)
    print("metric.npz", out["cer"], out["wer"])


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    install_stubs()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(4)
    gens = [gen_masks_noam, gen_sampler_goldens, gen_ctc_goldens, gen_init_goldens, gen_metric_goldens, gen_model_goldens,
            gen_fomaml_goldens, gen_fomaml_cfg3_goldens, gen_fomaml_8acc_goldens, gen_chain_goldens, gen_multi_goldens, gen_mono_goldens, gen_tester_goldens, gen_blstm_goldens, gen_blstm_subsample_goldens,
            gen_blstm_mono_goldens, gen_blstm_tester_goldens, gen_hkust_fullsize_goldens]
    only = set(sys.argv[1:])                       # e.g.  python oracle/make_goldens.py gen_fomaml_cfg3_goldens
    for g in gens:
        if not only or g.__name__ in only:
            g()


if __name__ == "__main__":
    main()
