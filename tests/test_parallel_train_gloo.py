"""World-2 (gloo, CPU) runs of the PRODUCT loops -- FOMetaASRInterface.train() and MultiASRInterface.train() -- on real toy
shards through the real DataContainer / BucketSampler, with a CPU engine double in place of the HIP engine (the double
exists only here; it has MasrEngine's call surface and a gradient that is a deterministic function of the batch CONTENT
and the current weights).  Checked:
  * FOMAML: 2 ranks end with the same meta weights as the single-process run (same task order, every accent served the
    batches of the single-process stream whichever rank it lands on, all-reduce(sum)/n_tasks, replicated Adam);
  * the batches the two ranks materialise are, together, exactly the single-process batch list (each once);
  * evaluate()/snapshots run inside the multi-rank loop (rank-0 evaluation + barrier; meta-weight evaluation split over
    ranks with --fix_snapshot_meta_weights);
  * multi-task: the N ranks of a step draw N different batches of the one shared stream.
Reference loops: src/fo_meta_interface.py:128-177, src/multi_interface.py:94-140."""
import json
import math
import os
import random
import socket
import tempfile
from functools import partial
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import masr_amd  # noqa: F401
from masr_amd.optimizer import FlatAdam, TransformerOptimizer
from oracle.make_goldens import write_toy_shard

N = 193
ACCENTS = {"af": "african", "au": "australia", "ca": "canada", "en": "england", "us": "us", "hk": "hongkong", "in": "indian",
           "ir": "ireland", "nz": "newzealand"}
EIGHT = ["af", "au", "en", "us", "hk", "in", "ir", "nz"]


def batch_fingerprint(x, ys):
    return round(float(x.double().sum()), 4) + 1000.0 * int(sum(int(y.sum()) for y in ys))


class CpuEngine:
    def __init__(self, log):
        self.params = torch.linspace(-1, 1, N, dtype=torch.float64)
        self.grads = torch.zeros(N, dtype=torch.float64)
        self._norm, self._loss, self.log = 0.0, 0.0, log
        self.interrupt_at = None
        self.device = torch.device("cpu")

    def copy(self, dst, src): dst.copy_(src)
    def mark_dirty(self): pass
    def set_seed(self, s): self._drop = [int(s), 0]                    # the dropout stream: (seed, position), as masr_dropout_state
    def dropout_state(self): return tuple(getattr(self, "_drop", [0, 0]))
    def set_dropout_state(self, d): self._drop = list(d)
    def axpy(self, y, x, a): y.add_(x, alpha=a)
    def scale(self, x, a): x.mul_(a)

    def state_dict(self, flat=None, clone=True):
        t = self.params if flat is None else flat
        return {"w": t.clone() if clone else t}

    def load_state_dict(self, sd): self.params.copy_(sd["w"])

    def run_batch(self, x, ilens, ys, olens, train):
        fp = batch_fingerprint(x, ys)
        if train and self.interrupt_at is not None and sum(1 for t, _ in self.log if t) == self.interrupt_at:
            self.interrupt_at = None
            raise KeyboardInterrupt
        self.log.append((bool(train), fp))
        if train:
            self._drop = [getattr(self, "_drop", [0, 0])[0], getattr(self, "_drop", [0, 0])[1] + 1]
        ph = (fp % 7.0) + 0.5
        self.grads = self.params * (0.1 + 0.01 * (fp % 3.0)) + 0.4 * torch.sin(torch.arange(N, dtype=torch.float64) * ph)
        self._loss = float(self.grads.abs().mean())

    def read_stats(self):
        return {"loss": self._loss, "n_correct": 1.0, "n_total": 2.0, "grad_norm": self._norm}

    def clip_grads(self, max_norm):
        self._norm = float(self.grads.norm())
        self.grads.mul_(min(1.0, max_norm / (self._norm + 1e-6)))

    def clip_sgd_step(self, buf, max_norm, lr, momentum, nesterov, first):
        self.clip_grads(max_norm)
        g = self.grads
        b = g.clone() if int(first) & 1 else buf * momentum + g          # bit 0: first step; bit 1: last step (buffer not kept)
        if buf is not None and not int(first) & 2:
            buf.copy_(b)
        self.params.sub_(lr * (g + momentum * b if nesterov else b))

    def adam_step(self, p, g, m, v, lr, b1, b2, eps, t, weight_decay=0.0, decoupled=False):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(m, (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps), value=-lr / (1 - b1 ** t))


def get_cpu_trainer(cls, config, paras, id2accent, log):
    """the Trainer mixin contract (transformer_torch_trainer.get_trainer) over the CPU double"""

    class CpuTrainer(cls):
        def set_model(self):
            eng = CpuEngine(log)
            eng.set_seed(531 + 7919 * 64 * self.sharder.rank)
            self.asr_model = SimpleNamespace(engine=eng, train=lambda: None, eval=lambda: None,
                                             load_state_dict=eng.load_state_dict)
            self.label_smooth_rate = 0.0
            mp_ = self.config['asr_model']
            if 'inner_optimizer_cls' not in mp_:
                self.asr_opt = TransformerOptimizer(FlatAdam(eng, eng.params, betas=(0.9, 0.98), eps=1e-9), 1.0, 64, 4)
            super().load_model()

        def _make_slots(self):
            """task slots as the product makes them (replica + engine per slot), without HIP streams"""
            models = [self.asr_model]
            for _ in range(self.tasks_per_gpu - 1):
                e = CpuEngine(log)
                models.append(SimpleNamespace(engine=e, train=lambda: None, eval=lambda: None, load_state_dict=e.load_state_dict))
            self._slots = [{'model': m, 'engine': m.engine, 'stream': None} for m in models]
            for i, sl in enumerate(self._slots):                         # one dropout stream per (rank, slot), as the product seeds them
                sl['engine'].set_seed(531 + 7919 * (self.sharder.rank * self.tasks_per_gpu + i))

        def exec(self):
            self.train()

        def run_batch(self, cur_b, x, ilens, ys, olens, train, accent_idx=None, engine=None, want_info=True):
            eng = engine if engine is not None else self.asr_model.engine
            eng.run_batch(x, ilens, ys, olens, train)
            olens += 1
            st = eng.read_stats()
            info = {'loss': st['loss'], 'acc': 0.5}
            if not train:
                info['cer'] = 50.0 + st['loss']
                info['wer'] = 60.0 + st['loss']
            return info

        def opt_step(self):
            self.asr_opt.optimizer.grad = self.asr_model.engine.grads
            self.asr_opt.step()

        def clip_grad_norm_(self, max_norm, engine=None):
            eng = engine if engine is not None else self.asr_model.engine
            eng.clip_grads(max_norm)
            return eng.read_stats()['grad_norm']

    return CpuTrainer(config, paras, id2accent)


def make_workspace(root):
    root = Path(root)
    (root / "data").mkdir(parents=True)
    json.dump(ACCENTS, open(root / "data" / "accent-code.json", "w"))
    with open(root / "data" / "units.txt", "w") as f:
        for i in range(365):
            f.write(f"u{i} {i + 1}\n")
    for ai, a in enumerate(ACCENTS.values()):
        write_toy_shard(root / "data", a, "train", 14, seed=100 + ai)
        write_toy_shard(root / "data", a, "dev", 3, seed=200 + ai)


def run(root, algo, world, rank, fix_snapshot=False, meta_batch=3, steps=5, accents=("af", "au", "en", "us"), meta_k=2, fix_reptile=False,
        deferred=False, log_ival=1, eval_ival=2, save_ival=2, tasks_per_gpu=1, is_bucket=True, resume=False, suffix=None, njobs=None,
        interrupt_at=None):
    os.chdir(root)
    model = {"d_model": 64}
    if algo in ("fomaml", "reptile"):
        model.update({"inner_optimizer_cls": "SGD", "inner_optimizer_opt": {"momentum": 0.9, "nesterov": True},
                      "meta_opt_cls": "noam", "meta": {"optimizer_opt": {"k": 1.0, "warmup_steps": 4}}})
    else:
        model.update({"optimizer_cls": "noam", "optimizer_opt": {"k": 1.0, "warmup_steps": 4}})
    cfg = {"asr_model": model,
           "solver": {"setting": "t", "data_root": "data", "total_steps": 100, "spm_mapping": "data/units.txt", "spm_model": "none",
                      "label_smoothing": 0.0, "eval_ival": eval_ival, "log_ival": log_ival, "save_ival": save_ival, "batch_size": 4, "dev_batch_size": 4,
                      "min_ilen": 10, "max_ilen": 50, "dev_max_ilen": 3000, "half_batch_ilen": 30}}
    paras = SimpleNamespace(pretrain_suffix=suffix or f"w{world}", pretrain_accents=list(accents), num_pretrain=len(accents), tgt_accent="ca", runs=0,
                            overwrite=True, seed=531, meta_k=meta_k, meta_batch_size=meta_batch, sample_strategy="normal", max_step=steps,
                            resume=resume, model_name="transformer", algo=algo, njobs=njobs if njobs is not None else (2 if world > 1 else 0), is_bucket=is_bucket, is_memmap=True,
                            use_tensorboard=False, fix_snapshot_meta_weights=fix_snapshot, tasks_per_gpu=tasks_per_gpu, fix_reptile=fix_reptile)
    random.seed(531); np.random.seed(531); torch.manual_seed(531)
    if algo in ("fomaml", "reptile"):
        from masr_amd.fo_meta_interface import FOMetaASRInterface as Iface
    else:
        from masr_amd.multi_interface import MultiASRInterface as Iface
    log = []
    solver = get_cpu_trainer(Iface, cfg, paras, ACCENTS, log)
    booked = []
    if deferred:
        # a trainer WITH the asynchronous stats read (as the HIP one): run_batch(want_info=False) returns nothing, the numbers
        # come out of a handle whenever the loop asks for them -- the handle notes at which meta-step that happened
        eng_of = lambda: solver.asr_model.engine
        plain_rb = solver.run_batch
        solver.run_batch = lambda *a, want_info=True, **k: (plain_rb(*a, **k) if want_info else (plain_rb(*a, **k), None)[1])
        solver._train = partial(solver.run_batch, train=True)

        def clip_stats_async(max_norm, engine=None):
            eng_of().clip_grads(max_norm)
            st, made = dict(eng_of().read_stats()), solver.global_step

            class H:
                def get(self_):
                    booked.append((made, solver.global_step))
                    return {'loss': st['loss'], 'acc': 0.5}, st['grad_norm']
            return H()
        solver.clip_stats_async = clip_stats_async
    n_reduces = []
    plain_reduce = solver.sharder.reduce_async
    solver.sharder.reduce_async = lambda buf, **k: (n_reduces.append(solver.global_step), plain_reduce(buf, **k))[1]
    solver.load_data()
    solver.set_model()
    solver.asr_model.engine.interrupt_at = interrupt_at
    solver.exec()
    slots = getattr(solver, "_slots", None) or [{"engine": solver.asr_model.engine}]
    dropout = [sl["engine"].dropout_state() for sl in slots]
    weights = solver._original.clone() if algo in ("fomaml", "reptile") else solver.asr_model.engine.params.clone()
    files = sorted(p.name for p in solver.log_dir.iterdir()) if rank == 0 else []
    dev_log = (solver.log_dir / "dev_avg_wer").read_text() if rank == 0 and (solver.log_dir / "dev_avg_wer").exists() else ""
    return {"dropout": dropout, "n_reduces": n_reduces, "tasks_per_gpu": getattr(solver, "tasks_per_gpu", 1), "weights": weights, "train_fps": [fp for tr, fp in log if tr], "files": files, "dev_avg_wer": dev_log,
            "global_step": solver.global_step, "train_info": {k: float(v) for k, v in solver.train_info.items()}, "booked": booked,
            "train_loss_log": (solver.log_dir / "train_loss").read_text() if rank == 0 and (solver.log_dir / "train_loss").exists() else ""}


def _worker(rank, world, port, root, algo, fix_snapshot, out_dir, kw):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = run(root, algo, world, rank, fix_snapshot, **kw)
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ranks(root, algo, fix_snapshot=False, world=2, **kw):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), root, algo, fix_snapshot, d, kw), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=False) for r in range(world)]


def _two_ranks(root, algo, fix_snapshot=False):
    return _ranks(root, algo, fix_snapshot, 2)


@pytest.fixture()
def workspace():
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        make_workspace(d)
        yield d
        os.chdir(cwd)


def test_deferred_task_stats_are_booked_one_meta_step_late_with_the_same_numbers(workspace):
    """the host may run one meta-step ahead of the GPU: a task's {loss, acc, norm} are then read back while the NEXT meta-step is
    queued.  Same weights, same running averages, same train_* log lines as with a read-back per task; every handle is consumed
    at the latest one meta-step after it was made, and before anything reads train_info (log line, evaluate, snapshot)."""
    ref = run(workspace, "fomaml", 1, 0, steps=7, log_ival=3, eval_ival=4, save_ival=5)
    got = run(workspace, "fomaml", 1, 0, steps=7, log_ival=3, eval_ival=4, save_ival=5, deferred=True)
    assert torch.equal(ref["weights"], got["weights"]) and ref["train_fps"] == got["train_fps"]
    assert ref["train_info"] == got["train_info"] and ref["train_loss_log"] == got["train_loss_log"] and ref["dev_avg_wer"] == got["dev_avg_wer"]
    steps_run = sorted({made for made, _ in got["booked"]})
    assert not ref["booked"] and len(steps_run) >= 7 and len(got["booked"]) == 3 * len(steps_run)    # 3 tasks per meta-step, each booked once
    lag = [at - made for made, at in got["booked"]]
    assert set(lag) == {0, 1} and lag.count(1) >= 6                                 # mostly a step late, never more
    for made, at in got["booked"]:
        if made % 3 == 0 or made % 4 == 0 or (made + 1) % 5 == 0:                  # log / evaluate / snapshot at that step: booked before
            assert at == made


@pytest.mark.parametrize("fix_snapshot", [False, True])
def test_fomaml_train_two_ranks_match_single_process(workspace, fix_snapshot):
    single = run(workspace, "fomaml", 1, 0, fix_snapshot)
    r0, r1 = _two_ranks(workspace, "fomaml", fix_snapshot)
    assert single["global_step"] == r0["global_step"] == r1["global_step"] == 5
    assert torch.equal(r0["weights"], r1["weights"]), "replicated Adam must leave identical meta weights on every rank"
    torch.testing.assert_close(r0["weights"], single["weights"], rtol=1e-9, atol=1e-11)
    # together the ranks materialised exactly the single-process batch list, each batch once (as a multiset: the toy shards
    # are small, so an utterance comes back in a later epoch, possibly on the other rank)
    assert sorted(r0["train_fps"] + r1["train_fps"]) == sorted(single["train_fps"])
    assert len(r0["train_fps"]) > 0 and len(r1["train_fps"]) > 0
    # evaluation + checkpoints ran inside the multi-rank loop and left the reference's files (rank 0 writes)
    for f in ("snapshot.latest", "snapshot.step.2", "snapshot.step.4", "model.wer.best", "best_wer", "best_cer", "global_step",
              "dev_avg_wer", "dev_african_loss", "train_loss"):
        assert f in r0["files"], (f, r0["files"])
    assert sorted(single["files"]) == sorted(r0["files"])
    assert len(r0["dev_avg_wer"].splitlines()) == len(single["dev_avg_wer"].splitlines()) == 2
    if fix_snapshot:                                   # meta weights are rank-independent -> same dev numbers as one process
        a = [float(l.split()[1]) for l in r0["dev_avg_wer"].splitlines()]
        b = [float(l.split()[1]) for l in single["dev_avg_wer"].splitlines()]
        assert np.allclose(a, b, rtol=1e-9)


def test_multi_train_two_ranks_draw_different_batches(workspace):
    single = run(workspace, "multi", 1, 0, steps=9)
    r0, r1 = _two_ranks(workspace, "multi")
    assert torch.equal(r0["weights"], r1["weights"])
    # rank r of step s took draw 2 s + r of the one shared stream: the two ranks never share a batch within a step, and the
    # interleaved sequence is the single-process stream
    n = min(len(r0["train_fps"]), len(r1["train_fps"]))
    assert n >= 4
    assert all(a != b for a, b in zip(r0["train_fps"], r1["train_fps"]))
    inter = [fp for pair in zip(r0["train_fps"], r1["train_fps"]) for fp in pair]
    assert inter[:len(single["train_fps"])] == single["train_fps"][:len(inter)]


@pytest.mark.parametrize("algo,meta_k", [("fomaml", 1), ("reptile", 5)])
def test_eight_accents_one_task_per_rank_on_eight_ranks(workspace, algo, meta_k):
    """BASELINE configs[3] / [4] as far as a CPU can take them: `pretrain.py --algo fomaml` with 8 accents sharded one task per
    rank on 8 ranks (all-reduce of the meta-gradient, replicated Noam-Adam), and `--algo reptile --fix_reptile` with 8 accents and
    inner_steps = 5.  Eight gloo ranks through the real train() loop end with the single-process meta weights; every rank ran
    exactly one task per meta-step; together they consumed the single-process batch list."""
    kw = dict(meta_batch=8, steps=4, accents=EIGHT, meta_k=meta_k, fix_reptile=(algo == "reptile"))
    single = run(workspace, algo, 1, 0, **kw)
    ranks = _ranks(workspace, algo, world=8, **kw)
    for r in ranks[1:]:
        assert torch.equal(ranks[0]["weights"], r["weights"])
    torch.testing.assert_close(ranks[0]["weights"], single["weights"], rtol=1e-9, atol=1e-11)
    per_step = meta_k + 1                                            # k inner batches + the val batch of the rank's one task
    n_meta = len(single["train_fps"]) // (8 * per_step)              # (the loop runs whole eval_ival rounds: 4 meta-steps)
    assert n_meta == 4 and all(len(r["train_fps"]) == n_meta * per_step for r in ranks)
    assert sorted(fp for r in ranks for fp in r["train_fps"]) == sorted(single["train_fps"])


def test_meta_batch_five_on_eight_ranks_three_ranks_pad(workspace):
    """--meta_batch_size 5 of 8 accents on 8 ranks: task position p of a meta-step runs on rank p % 8, so ranks 5..7 own nothing in
    any meta-step -- they still draw every task's batch INDICES (shared RNG streams), contribute a zero buffer to the one all-reduce
    per meta-step and apply the replicated Noam-Adam step.  All eight end on the single-process meta weights; ranks 0..4 together
    consumed the single-process batch list, ranks 5..7 materialised no batch at all."""
    kw = dict(meta_batch=5, steps=4, accents=EIGHT, meta_k=1)
    single = run(workspace, "fomaml", 1, 0, **kw)
    ranks = _ranks(workspace, "fomaml", world=8, **kw)
    for r in ranks[1:]:
        assert torch.equal(ranks[0]["weights"], r["weights"])
    torch.testing.assert_close(ranks[0]["weights"], single["weights"], rtol=1e-9, atol=1e-11)
    n_meta = len(single["train_fps"]) // (5 * 2)
    assert n_meta == 4
    assert all(len(r["train_fps"]) == n_meta * 2 for r in ranks[:5]) and all(not r["train_fps"] for r in ranks[5:])
    assert sorted(fp for r in ranks for fp in r["train_fps"]) == sorted(single["train_fps"])
    for r in ranks:                                                    # one all-reduce per meta-step on EVERY rank (the idle ones pad)
        assert [r["n_reduces"].count(st) for st in sorted(set(r["n_reduces"]))] == [1] * n_meta


@pytest.mark.parametrize("algo,meta_batch,K,waves", [("fomaml", 4, 2, 1), ("fomaml", 7, 2, 2), ("fomaml", 7, 3, 2), ("reptile", 7, 3, 2)])
def test_task_slots_on_two_ranks_issue_one_allreduce_per_wave(workspace, algo, meta_batch, K, waves):
    """--tasks_per_gpu K on several ranks: the K task gradients of a wave are summed on the rank and go out as ONE all-reduce
    (K times less traffic than one per task; what bench.py's meta_step.concurrent_slots leg times).  Every rank issues the same
    number of all-reduces per meta-step -- the waves of the busiest rank, ranks with fewer waves pad with zeros -- and the ranks
    end with the single-process, one-task-at-a-time meta weights.  meta_batch 7 on 2 ranks: 4 + 3 tasks, i.e. 2 + 2 waves at
    K = 2 and 2 + 1 (one padded) at K = 3."""
    kw = dict(meta_batch=meta_batch, steps=4, accents=EIGHT, meta_k=2, fix_reptile=(algo == "reptile"))
    single = run(workspace, algo, 1, 0, **kw)
    slots1 = run(workspace, algo, 1, 0, tasks_per_gpu=K, **kw)
    torch.testing.assert_close(slots1["weights"], single["weights"], rtol=1e-9, atol=1e-11)
    r0, r1 = _ranks(workspace, algo, world=2, tasks_per_gpu=K, **kw)
    assert r0["tasks_per_gpu"] == K
    assert torch.equal(r0["weights"], r1["weights"])
    torch.testing.assert_close(r0["weights"], single["weights"], rtol=1e-9, atol=1e-11)
    assert sorted(r0["train_fps"] + r1["train_fps"]) == sorted(single["train_fps"])
    for r in (r0, r1):
        per_step = [r["n_reduces"].count(st) for st in sorted(set(r["n_reduces"]))]
        assert per_step == [waves] * 4, per_step



@pytest.mark.parametrize("fix_snapshot", [False, True])
def test_no_bucket_loaders_stay_rank_consistent_across_evaluations(workspace, fix_snapshot):
    """--no_bucket: the train loaders' RandomSampler reads the TORCH default stream, and so does every dev iterator evaluate()
    creates (base seed).  Ranks that skip an accent's evaluation take the same draws, otherwise an accent that lands on another rank
    after an evaluation would be served a different permutation there: with 6 meta-steps and an evaluation every 2, the two ranks
    together must still consume exactly the single-process batch list and end on its meta weights."""
    kw = dict(steps=7, is_bucket=False, meta_batch=3)
    single = run(workspace, "fomaml", 1, 0, fix_snapshot, **kw)
    r0, r1 = _ranks(workspace, "fomaml", fix_snapshot, 2, **kw)
    assert torch.equal(r0["weights"], r1["weights"])
    assert sorted(r0["train_fps"] + r1["train_fps"]) == sorted(single["train_fps"])
    torch.testing.assert_close(r0["weights"], single["weights"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("algo,is_bucket,eval_ival,save_ival,stop,total", [
    ("fomaml", True, 2, 3, 3, 9), ("fomaml", True, 3, 5, 7, 13), ("fomaml", False, 2, 3, 3, 9), ("multi", True, 2, 3, 3, 9), ("reptile", True, 2, 3, 3, 7)])
def test_resume_continues_the_data_and_task_streams_exactly(workspace, algo, is_bucket, eval_ival, save_ival, stop, total):
    """--resume from snapshot.latest + meta_state.latest (CPU double, real loops / DataContainer / samplers): the resumed run draws
    the batches the uninterrupted run drew after the checkpoint -- same task order, same sampler cursors through epoch roll-overs,
    same RNG streams across evaluate() -- and ends on identical weights and running averages.  With a checkpoint in the middle of an
    eval chunk (5 of 3-step chunks) the resumed run re-enters that chunk."""
    kw = dict(eval_ival=eval_ival, save_ival=save_ival, is_bucket=is_bucket, fix_reptile=(algo == "reptile"))
    full = run(workspace, algo, 1, 0, steps=total, suffix="full", **kw)
    part = run(workspace, algo, 1, 0, steps=stop, suffix="part", **kw)
    n_part = len(part["train_fps"])
    res = run(workspace, algo, 1, 0, steps=total, suffix="part", resume=True, **kw)
    assert res["global_step"] == full["global_step"]
    saved_at = (stop // save_ival) * save_ival
    per_step = len(full["train_fps"]) // (full["global_step"] - 1)
    assert res["train_fps"] == full["train_fps"][(saved_at - 1) * per_step:], "the resumed run drew other batches than the uninterrupted one"
    assert torch.equal(res["weights"], full["weights"])
    assert res["train_info"] == full["train_info"]
    assert n_part == (stop - 1) * per_step


@pytest.mark.parametrize("algo,K", [("fomaml", 1), ("fomaml", 2), ("multi", 1)])
def test_two_rank_resume_restores_each_ranks_own_dropout_streams(workspace, algo, K):
    """the dropout streams are per (rank, slot).  A checkpoint written by rank 0 holds EVERY rank's stream positions and a resumed
    rank restores its own: after stop + resume both ranks sit where the uninterrupted two-rank run left them (seed and position),
    the two ranks' streams differ from each other, and the meta weights are those of the uninterrupted run."""
    kw = dict(meta_batch=4, eval_ival=2, save_ival=3, tasks_per_gpu=K)
    full = _ranks(workspace, algo, world=2, steps=9, suffix="full", **kw)
    _ranks(workspace, algo, world=2, steps=4, suffix="part", **kw)               # saved at global_step 3
    res = _ranks(workspace, algo, world=2, steps=9, suffix="part", resume=True, **kw)
    for r in (0, 1):
        assert res[r]["dropout"] == full[r]["dropout"], (r, res[r]["dropout"], full[r]["dropout"])
        assert torch.equal(res[r]["weights"], full[r]["weights"])
    assert full[0]["dropout"] != full[1]["dropout"]
    assert {seed for seed, _ in full[0]["dropout"]}.isdisjoint({seed for seed, _ in full[1]["dropout"]})


@pytest.mark.parametrize("algo", ["fomaml", "multi"])
def test_sigint_checkpoint_does_not_lose_the_batches_drawn_ahead(workspace, algo):
    """Ctrl-C while the NEXT step's batches are already drawn (look-ahead, collate pool): the checkpoint holds the RNG / sampler /
    task-order state as it stood BEFORE that draw, so the resumed run draws those batches again -- exactly what a loop without
    look-ahead (the reference's) leaves behind: the interrupted step's own batches are spent, nothing after them is skipped."""
    per_step = 3 * 3 if algo == "fomaml" else 1                        # meta_batch 3 x (2 inner + val) | one batch per step
    kw = dict(eval_ival=5, save_ival=50, njobs=2)                      # (the loops run whole eval_ival chunks, as the reference's do)
    full = run(workspace, algo, 1, 0, steps=16, suffix="full", **kw)
    # a train call in the middle of (meta-)step 4; FOMAML: in its SECOND task (the look-ahead runs once the first task is queued)
    hit = 3 * per_step + (4 if algo == "fomaml" else 0)
    part = run(workspace, algo, 1, 0, steps=9, suffix="part", interrupt_at=hit, **kw)
    assert part["global_step"] == 4 and len(part["train_fps"]) == hit
    res = run(workspace, algo, 1, 0, steps=9, suffix="part", resume=True, **kw)
    assert res["train_fps"][:per_step] == full["train_fps"][4 * per_step:5 * per_step], "the step drawn ahead of the SIGINT was skipped"
    assert res["train_fps"] == full["train_fps"][4 * per_step:4 * per_step + len(res["train_fps"])]
