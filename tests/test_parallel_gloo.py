"""N > 1 path on CPU: two gloo ranks shard the tasks of each meta-step and must reproduce the single-process
meta weights exactly (same task order, same per-task gradients, all-reduce(sum) / n_tasks, replicated Adam).
The engine here is a CPU test double with the MasrEngine call surface -- it exists only in this test."""
import math
import os
import random
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import masr_amd
from masr_amd.fo_meta_interface import FOMetaASRInterface
from masr_amd.optimizer import FlatAdam, TransformerOptimizer
from masr_amd.parallel import TaskSharder

N = 257


class FakeEngine:
    """params/grads on CPU; 'gradient' of a batch = params * a_task + b_batch (deterministic, task dependent)"""

    def __init__(self):
        self.params = torch.linspace(-1, 1, N)
        self.grads = torch.zeros(N)
        self._norm = 0.0

    def copy(self, dst, src): dst.copy_(src)
    def mark_dirty(self): pass
    def axpy(self, y, x, a): y.add_(x, alpha=a)
    def scale(self, x, a): x.mul_(a)

    def run_batch(self, x, ilens, ys, olens, train):
        task, k = x
        g = torch.Generator().manual_seed(1000 * task + k)
        self.grads = self.params * (0.1 + 0.05 * task) + torch.randn(N, generator=g) * (3.0 if k == 7 else 0.3)

    def read_stats(self):
        return {"loss": float(self.params.sum()), "n_correct": 1.0, "n_total": 2.0, "grad_norm": self._norm}

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
        assert weight_decay == 0.0
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(m, (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps), value=-lr / (1 - b1 ** t))


class FakeData:
    def __init__(self): self.cnt = {}
    def get_item(self, accent, num=1):
        out = []
        for _ in range(num):
            k = self.cnt.get(accent, 0)
            self.cnt[accent] = k + 1
            out.append((accent, ((accent, k), torch.tensor([8]), [torch.tensor([1])], torch.tensor([1]))))
        return out


def make_solver(sharder, num_tasks, meta_batch):
    s = FOMetaASRInterface.__new__(FOMetaASRInterface)
    eng = FakeEngine()
    s.asr_model = SimpleNamespace(engine=eng, train=lambda: None, eval=lambda: None)
    s.sharder = sharder
    s.paras = SimpleNamespace(algo="fomaml", resume=False)
    s.config = {"asr_model": {"inner_optimizer_cls": "SGD", "inner_optimizer_opt": {"momentum": 0.9, "nesterov": True},
                              "d_model": 64}}
    s.meta_k, s.meta_batch_size, s.num_pretrain = 2, meta_batch, num_tasks
    s.inner_lr = 0.05
    s._updates, s._counter = None, 0
    s._original = eng.params.clone()
    s.meta_opt = TransformerOptimizer(FlatAdam(eng, s._original, betas=(0.9, 0.98), eps=1e-9), 1.0, 64, 4)
    s.data_container = FakeData()
    s.train_info = SimpleNamespace(add=lambda *a, **k: None)
    s.accents = [str(i) for i in range(num_tasks)]
    s.global_step = 1
    s._task_rng = random.Random(531)
    from functools import partial

    def run_batch(idx, x, ilens, ys, olens, train, accent_idx=None, **kw):
        eng.run_batch(x, ilens, ys, olens, train)
        return {"loss": 0.0, "acc": 0.0}
    s._train = partial(run_batch, train=True)
    s.clip_grad_norm_ = lambda mx: (eng.clip_grads(mx), eng.read_stats()["grad_norm"])[1]
    return s


def meta_steps(s, n_steps):
    from masr_amd.marcos import GRAD_CLIP
    task_ids = list(range(s.num_pretrain))
    for _ in range(n_steps):
        s._task_rng.shuffle(task_ids)
        mb = task_ids[:s.meta_batch_size]
        # every rank must see the same per-accent batch counters: advance the fake data of non-owned tasks too
        n_local = 0
        for accent in mb:
            mine = accent in s.sharder.my_tasks(mb)
            tr = s.data_container.get_item(accent, s.meta_k)
            val = s.data_container.get_item(accent)[0]
            if not mine:
                continue
            s.run_task(tr)
            s._train(val[0], *val[1], accent_idx=val[0])
            s.clip_grad_norm_(GRAD_CLIP)
            s._partial_meta_update()
            n_local += 1
        s._pad_rounds(len(mb), n_local)
        s._final_meta_update(len(mb))
    return s._original.clone()


def _worker(rank, world, port, num_tasks, meta_batch, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = make_solver(TaskSharder.from_env(), num_tasks, meta_batch)
    assert s.sharder.world == world
    res = meta_steps(s, 3)
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("num_tasks,meta_batch", [(4, 4), (5, 3), (2, 1)])
def test_two_ranks_match_single_process(num_tasks, meta_batch):
    single = meta_steps(make_solver(TaskSharder(), num_tasks, meta_batch), 3)
    with tempfile.TemporaryDirectory() as d:
        port = 29500 + random.randint(0, 2000)
        mp.spawn(_worker, args=(2, port, num_tasks, meta_batch, d), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
    assert torch.equal(r0, r1), "replicated Adam must leave identical meta weights on every rank"
    torch.testing.assert_close(r0, single, rtol=1e-6, atol=1e-7)


def test_task_partition():
    a, b = TaskSharder(0, 2), TaskSharder(1, 2)
    mb = [3, 0, 2, 1, 4]
    assert a.my_tasks(mb) == [3, 2, 4] and b.my_tasks(mb) == [0, 1]
    assert sorted(a.my_tasks(mb) + b.my_tasks(mb)) == sorted(mb)
    assert TaskSharder().my_tasks(mb) == mb
