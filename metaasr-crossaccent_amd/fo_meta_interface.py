"""FOMetaASRInterface: the first-order meta-learning loop of the reference (src/fo_meta_interface.py:18-302) on the
flat HBM buffers of the HIP engine.

  _original : flat fp32 copy of the meta weights            (reference: OrderedDict of cloned tensors, :103)
  _updates  : flat fp32 meta-gradient accumulator            (reference: dict of per-tensor accumulators, :184-192)
  run_task  : copy meta -> model (one 99.5 MB D2D copy), fresh SGD, k x {run_batch, clip 5, NaN ? skip : step}
  train     : per task run_task + val-batch gradient + clip + accumulate; then /= counter and Noam-Adam
Multi-GPU: tasks of a meta-step are split round-robin over ranks and `_updates` is all-reduced (parallel.py).
Reference quirks kept by default: snapshots hold the LAST TASK's adapted weights (Q1/F8),
evaluation runs on them too (Q2), a NaN val gradient is still accumulated (Q5); --fix_snapshot_meta_weights / --fix_nan_meta_grad opt out."""
import contextlib
import math
import pickle
import random
import threading
from functools import partial

import torch

from .marcos import *  # noqa: F401,F403
from .monitor import logger
from .monitor.stat import RunningAvgDict
from .optimizer import FlatAdam, FlatSGD, TransformerOptimizer
from .pretrain_interface import PretrainInterface


_TASK_STREAMS = {}


def _task_streams(dev, n):
    lst = _TASK_STREAMS.setdefault(str(dev), [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=dev))
    return lst[:n]


class _Resolved:
    """stats handle of a task whose numbers are already on the host"""

    def __init__(self, info, grad_norm):
        self.info, self.grad_norm = info, grad_norm

    def get(self):
        return self.info, self.grad_norm


def slot_cap(tasks_per_gpu, meta_batch_size, world, collective, no_slot_cap=False):
    """task slots a rank actually runs: several ranks AND several waves per meta-step means the all-reduce of wave w (RCCL's own stream)
    runs beside the tasks of wave w+1, i.e. a fifth stream with work queued -- three task slots keep the total at the four that run side
    by side on the hardware queues (DESIGN: concurrent task slots).  Also reported by bench.py (meta_step.slot_cap)."""
    rounds = -(-meta_batch_size // world)
    if collective and tasks_per_gpu > 3 and rounds > tasks_per_gpu and not no_slot_cap:
        return 3
    return tasks_per_gpu


class FOMetaASRInterface(PretrainInterface):
    def __init__(self, config, paras, id2accent):
        super().__init__(config, paras, id2accent)
        assert paras.meta_k is not None
        self.meta_k = paras.meta_k
        self.asr_model = self.asr_opt = None
        self.dashboard.set_status('pretraining')
        self._train = partial(self.run_batch, train=True)
        self._eval = partial(self.run_batch, train=False)
        self.meta_batch_size = paras.meta_batch_size if paras.meta_batch_size is not None else self.num_pretrain
        self._updates = None
        self._counter = 0
        self._pending = []                                           # (meta-step, accent, stats handle, batch size) not booked yet
        self._task_grads = []                                        # gradient buffers of this meta-step's tasks (fused meta update)
        mp = config['asr_model']
        o = mp['meta']['optimizer_opt']
        self.inner_lr = mp['d_model'] ** (-0.5) * o['k'] * (o['warmup_steps'] ** (-0.5))         # :41-45
        self.fix_snapshot = bool(getattr(paras, 'fix_snapshot_meta_weights', False))
        # --fix_reptile: the reference's `--algo reptile` dies with ValueError in _partial_meta_update (:197-198);
        # with the flag the published Reptile pseudo-gradient (theta_meta - theta_k) is used instead (parity unpinned)
        self.fix_reptile = bool(getattr(paras, 'fix_reptile', False))
        # --fix_nan_meta_grad: the reference warns about a NaN val-batch gradient and accumulates it all the same (:151-154, quirk Q5): one bad
        # batch and the meta weights are NaN for good.  With the flag the val-batch clip turns such a gradient into zeros on the device
        # (include/masr.h masr_set_drop_nan_grads): the task still counts in the mean, the warning is still logged.
        self.fix_nan = bool(getattr(paras, 'fix_nan_meta_grad', False))
        # MI355X extension: tasks of one meta-step are independent, so several of them can run CONCURRENTLY on one GPU
        # (one model replica + HIP stream + host thread per slot).  A B=16 inner step leaves ~40 % of the 256 CUs idle
        # (small decoder GEMMs, kernel tails); three concurrent tasks raise the throughput ~1.6x.  1 = reference order.
        self.tasks_per_gpu = max(1, int(getattr(paras, 'tasks_per_gpu', 1) or 1))
        if self.tasks_per_gpu > 8:
            # the one-pass meta update / wave sum read at most 8 gradient buffers (include/masr.h masr_adam_sum_step, masr_sum_n),
            # and more than four streams with work queued take turns on the hardware queues anyway
            raise ValueError(f"--tasks_per_gpu {self.tasks_per_gpu}: at most 8 task slots per GPU (4 is the most that pays)")
        rounds = -(-self.meta_batch_size // self.sharder.world)
        capped = slot_cap(self.tasks_per_gpu, self.meta_batch_size, self.sharder.world, self.sharder.collective, getattr(paras, 'no_slot_cap', False))
        if capped != self.tasks_per_gpu:
            logger.notice(f"tasks_per_gpu {self.tasks_per_gpu} -> {capped}: {rounds} tasks per rank and meta-step run as several waves whose "
                          f"all-reduce overlaps the next wave (four-queue rule, DESIGN 6.2; --no_slot_cap keeps the setting)")
            self.tasks_per_gpu = capped
        self._slots = None
        self._n_reduces = 0                                          # all-reduces this rank has issued in the current meta-step
        # task order: the reference's global `random` stream (:136).  Every rank consumes that stream identically (data
        # draws of non-owned tasks are replayed index-only, see train()), so the order is the same on all ranks and the
        # same as in the single-process run.
        self._task_rng = random
        logger.notice(f"Meta batch size {self.meta_batch_size}, {self.meta_k} inner steps, inner lr {self.inner_lr:.3e}, "
                      f"rank {self.sharder.rank}/{self.sharder.world}")

    # ------------------------------------------------------------------ checkpoints (file names / contents: Appendix D)
    def _snapshot_sd(self):
        eng = self.asr_model.engine
        flat = self._original if self.fix_snapshot else None            # default: adapted weights of the last task (Q1)
        return eng.state_dict(flat=flat, clone=False)

    def save_best_model(self, tpe='wer', only_stat=False):
        if self.sharder.rank != 0:
            return
        if not only_stat:
            torch.save({k: v.cpu() for k, v in self._snapshot_sd().items()}, self.log_dir.joinpath(f'model.{tpe}.best'))
        with open(self.log_dir.joinpath(f'best_{tpe}'), 'w') as fout:
            print('{} {}'.format(self.global_step, getattr(self, f'best_{tpe}')), file=fout)

    def _stream_state(self):
        """what decides the NEXT batches and task order: the three RNG streams, every sampler's arrangement + cursor, the task list"""
        from .io.dataset import capture_rng
        return {'rng': capture_rng(), 'data': self.data_container.state_dict(),
                'task_ids': list(getattr(self, '_task_ids', None) or range(self.num_pretrain))}

    def save_per_steps(self, gather=True):
        self._drain_stats()
        # the dropout streams are per (rank, slot): every rank's positions go into rank 0's file (a resumed rank restores ITS OWN;
        # gather=False -- the SIGINT path, where the other ranks may not be in this call -- keeps this rank's only)
        mine = [sl['engine'].dropout_state() for sl in (self._slots or []) if hasattr(sl['engine'], 'dropout_state')]
        dropout = dict(enumerate(self.sharder.all_gather_object(mine))) if gather else {self.sharder.rank: mine}
        if self.sharder.rank != 0:
            return
        sd = {k: v.cpu() for k, v in self._snapshot_sd().items()}
        torch.save(sd, self.log_dir.joinpath("snapshot.latest"))
        with open(self.log_dir.joinpath("info_dict.latest"), 'wb') as f:
            pickle.dump(self.train_info, f)
        with open(self.log_dir.joinpath("global_step"), 'w') as f:
            print(self.global_step, file=f)
        torch.save(sd, self.log_dir.joinpath(f"snapshot.step.{self.global_step}"))
        # extension (not in the reference, which cannot resume pretraining): meta weights + meta-Adam state
        # ... and everything else an exact continuation needs: the three RNG streams and the per-accent sampler state that decide
        # which batches come next, the best-so-far error rates, the dropout streams' positions.  (The look-ahead never crosses a
        # save boundary, see train(), so no drawn-but-unused batch is lost here.  A SIGINT can land while a look-ahead is
        # outstanding: the streams are then saved as they stood BEFORE that draw, `_ahead_state`, so the resumed run draws it again.)
        stream = getattr(self, '_ahead_state', None) or self._stream_state()
        torch.save({'original': self._original.cpu(), 'adam': self.meta_opt.optimizer.state_dict(),
                    'step_num': self.meta_opt.step_num, 'rng': stream['rng'], 'data': stream['data'],
                    'best': (self.best_wer, self.best_cer), 'task_ids': stream['task_ids'], 'dropout': dropout},
                   self.log_dir.joinpath("meta_state.latest"))
        self.dashboard.log_step()

    def load_model(self):
        eng = self.asr_model.engine
        if self.paras.resume:
            eng.load_state_dict(torch.load(self.resume_model_path))
            self.dashboard.set_step(self.global_step)
        self._original = eng.params.clone()                              # clone_state_dict(state_dict(keep_vars=True)) (:103)
        if self.config['asr_model']['meta_opt_cls'] != 'noam':
            raise NotImplementedError("Should use noam optimizer in outer loop transformer learning")
        o = self.config['asr_model']['meta']['optimizer_opt']
        self.meta_opt = TransformerOptimizer(FlatAdam(eng, self._original, betas=(0.9, 0.98), eps=1e-09),
                                             o['k'], self.config['asr_model']['d_model'], o['warmup_steps'])
        self._make_slots()
        ms = self.log_dir.joinpath("meta_state.latest")
        if self.paras.resume and ms.exists():
            st = torch.load(ms, weights_only=False)
            self._original.copy_(st['original'])
            self.meta_opt.optimizer.load_state_dict(st['adam'])
            self.meta_opt.step_num = st['step_num']
            self.meta_opt.lr = st['adam']['lr']
            if 'rng' in st:                                              # exact continuation of the data / task-order streams
                from .io.dataset import restore_rng
                self.data_container.load_state_dict(st['data'])
                self.best_wer, self.best_cer = st['best']
                self._task_ids = list(st['task_ids'])
                dropout = st['dropout']
                if isinstance(dropout, dict):                            # per rank; a rank the file does not know keeps its own seeds
                    dropout = dropout.get(self.sharder.rank, [])
                elif self.sharder.rank != 0:                             # (files written before the streams were saved per rank: rank 0's)
                    dropout = []
                for sl, d in zip(self._slots or [], dropout):
                    sl['engine'].set_dropout_state(d)
                restore_rng(st['rng'])                                   # last: everything above (and set_model) consumed the streams

    def _make_slots(self):
        """slot 0 is the trainer's own model; further slots are replicas of the same architecture"""
        import torch as _t
        from .model import MyTransformer
        dev = self.asr_model.engine.device
        models = [self.asr_model]
        for _ in range(self.tasks_per_gpu - 1):
            models.append(MyTransformer(self.id2ch, self.config['asr_model'], self.label_smooth_rate, device=dev, init=False))
        # slot 0 runs on the caller's (main) stream, see _run_tasks_concurrently; the others on streams made ONCE per process:
        # torch hands out its 32 pooled streams round-robin and HIP spreads them over the hardware queues, so a second
        # interface in the same process (tests, tools/bench_pretrain.py) would otherwise land on queues that share a
        # hardware pipe with the first one's -- measured: 8 040 utt/s on the first three pooled streams, 5 850 on the next three
        streams = [None] + _task_streams(dev, self.tasks_per_gpu - 1) if self.tasks_per_gpu > 1 else [None]
        self._slots = [{'model': m, 'engine': m.engine, 'stream': st} for m, st in zip(models, streams)]
        for i, sl in enumerate(self._slots):                                 # one dropout stream per (rank, slot)
            sl['engine'].set_seed(getattr(self.paras, 'seed', 531) + 7919 * (self.sharder.rank * self.tasks_per_gpu + i))
            if hasattr(sl['engine'], 'set_concurrency'):
                sl['engine'].set_concurrency(self.tasks_per_gpu)
            # "K task slots == the sequential run, bit for bit" must hold for every K: the k-split of the decoder's few-row GEMMs (a different
            # fp32 summation order that pays for a lone task only; the engine's default is off) stays off in this interface whatever K is
            if hasattr(sl['engine'], 'set_ksplit'):
                sl['engine'].set_ksplit(False)
            if self.fix_nan:
                sl['engine'].set_drop_nan_grads(True)

    def write_tr_logs(self):
        for k, v in self.train_info.items():
            self.write_log(f"train_{k}", float(v))

    def write_dev_logs(self, prefix, info):
        for k, v in info.items():
            self.write_log(f"{prefix}_{k}", float(v))

    def check_evaluate(self):
        if self.global_step % self.eval_ival == 0:
            self.evaluate()

    def _task_on_slot(self, slot, tr_batches, val_batch, out, i):
        """one task = run_task + val-batch gradient + clip, entirely on the slot's stream (host thread body)"""
        try:
            with (torch.cuda.stream(slot['stream']) if slot['stream'] is not None else contextlib.nullcontext()):
                self.run_task(tr_batches, engine=slot['engine'])
                info = self._train(val_batch[0], *val_batch[1], accent_idx=val_batch[0], engine=slot['engine'], want_info=False)
                out[i] = self._clip_and_stats(info, engine=slot['engine'])
        except BaseException as e:                                # re-raised on the main thread after the join
            out[i] = e

    def _clip_and_stats(self, info, engine=None):
        """clip_grad_norm_(GRAD_CLIP) of the val-batch gradient + the task's {loss, acc} and norm, as a handle: with a trainer that
        can copy them to the host asynchronously (`clip_stats_async`) nothing waits here -- the reference uses these numbers for
        log lines only (:147-151) -- and `_drain_stats` collects them one meta-step later, in task order.  Otherwise (or when
        run_batch already returned an info): the one host sync of the task, as before."""
        self._wire_clip = False
        if info is None and hasattr(self, 'clip_stats_async') and not getattr(self.paras, 'sync_stats', False):
            # native exchange, one task at a time: only the NORM is formed here -- the scale by the clip coefficient rides on the
            # all-reduce of _partial_meta_update, chunk by chunk beside the collective (include/masr.h masr_allreduce)
            if self._clip_on_the_wire(engine):
                self._wire_clip = True
                return self.clip_stats_async(GRAD_CLIP, engine=engine, norm_only=True)
            return self.clip_stats_async(GRAD_CLIP, engine=engine)
        grad_norm = self.clip_grad_norm_(GRAD_CLIP, engine=engine) if engine is not None else self.clip_grad_norm_(GRAD_CLIP)
        if info is None:
            info = self.info_from_stats(engine) if engine is not None else self.info_from_stats()
        return _Resolved(info, grad_norm)

    def _clip_on_the_wire(self, engine=None):
        sh = self.sharder
        # (--fix_nan_meta_grad: the clip stays in masr_clip_grads, which is where a NaN-norm gradient is zeroed)
        return (engine is None and getattr(sh, 'native', False) and self.tasks_per_gpu == 1 and self.paras.algo == 'fomaml' and not self.fix_nan
                and hasattr(self.asr_model.engine, 'grad_norm_device_ptr'))

    def _drain_stats(self, keep_steps=0):
        """book the pending task stats (all but those of the newest `keep_steps` meta-steps): NaN warning + train_info.add, in
        the order the tasks ran.  Readers of train_info (log line, evaluate, snapshot) drain everything first."""
        while self._pending and (keep_steps == 0 or self._pending[0][0] <= self.global_step - keep_steps):
            step, accent_id, handle, batch_size = self._pending.pop(0)
            info, grad_norm = handle.get()
            if math.isnan(grad_norm):
                logger.warning(f"grad norm NaN @ step {step} on {self.accents[accent_id]}, ignore...")
            self.train_info.add(info, batch_size)

    def _run_tasks_concurrently(self, drawn, while_running=None):
        """tasks_per_gpu > 1.  `drawn` = this rank's tasks of the meta-step as _draw_meta_batch returns them (indices drawn on
        the main thread in the reference's order, batches being assembled by the collate pool).  Waves of K tasks run
        concurrently, one replica + stream + host thread each; gradients are accumulated in task order -> deterministic."""
        fetched = [(a, [(i, f.result()) for i, f in tr], (val[0], val[1].result())) for a, tr, val in drawn]
        cuda = self._slots[0]['engine'].device.type == 'cuda'
        main = torch.cuda.current_stream() if cuda else None       # (CPU engines: the doubles of the gloo tests; no streams)
        # slot 0 runs ON the main stream: K streams in all.  Four streams run side by side without loss (tools/queue_probe.py);
        # with the host running ahead, a fifth queue holding nothing but the meta-update behind its waits on the task streams
        # shares a place with one task stream, which then only gets served once the other three have drained (measured:
        # three tasks done after 14.8 ms, the fourth after 23.3 ms; tools/e2e_gpu_timeline.py)
        self._slots[0]['stream'] = main
        K = self.tasks_per_gpu
        # every task of the meta-step on a slot of its own: the meta update reads the K gradient buffers directly (one pass
        # instead of zero + K accumulations + scale + Adam; same additions in the same order)
        fused = (len(fetched) <= min(K, 8) and not self.sharder.collective and self.paras.algo == 'fomaml' and self._updates is None
                 and hasattr(self._slots[0]['engine'], 'adam_sum_step') and not getattr(self.paras, 'no_fused_meta_update', False))
        for w0 in range(0, len(fetched), K):
            wave = fetched[w0:w0 + K]
            out = [None] * len(wave)
            threads = []
            for i, (accent_id, tr, val) in enumerate(wave):
                sl = self._slots[i]
                if cuda and sl['stream'] != main:
                    sl['stream'].wait_stream(main)                  # meta weights / previous accumulation are ready
                t = threading.Thread(target=self._task_on_slot, args=(sl, tr, val, out, i))
                t.start(); threads.append(t)
            if while_running is not None:                            # host work of the main thread, hidden behind the first wave
                while_running(); while_running = None
            for t in threads:
                t.join()
            for o in out:
                if isinstance(o, BaseException):
                    raise o
            for i, (accent_id, tr, val) in enumerate(wave):
                sl = self._slots[i]
                if cuda and sl['stream'] != main:
                    main.wait_stream(sl['stream'])
                self._counter += 1                                   # counted here (main thread), not in the worker
                if fused:
                    self._task_grads.append(sl['engine'].grads)
                elif not self.sharder.collective:
                    self._partial_meta_update(engine=sl['engine'])
                self._pending.append((self.global_step, accent_id, out[i], len(val[1][2])))
                self.asr_model = sl['model']                         # quirk Q1/Q2: the LAST task's adapted weights are "the model"
            if self.sharder.collective:
                self._exchange_wave([self._slots[i]['engine'] for i in range(len(wave))])
        return len(fetched)

    def _exchange_wave(self, engines):
        """several ranks, K task slots: the wave's task gradients are summed on this rank first (task order, one pass) and the sum
        goes out as ONE all-reduce -- the sum is linear, so this is the `_updates[n] += p.grad` of every task of every rank
        (:190-196) with K times less xGMI traffic than an all-reduce per task.  Issued from the main stream (RCCL runs it on its
        own): with one wave per meta-step nothing is left to overlap with, with several it overlaps the next wave."""
        if self.paras.algo != 'fomaml' and not (self.paras.algo == 'reptile' and self.fix_reptile):
            raise ValueError(f"Not support meta algo {self.paras.algo}")
        eng = engines[0]
        contrib = torch.empty_like(eng.params)
        if self.paras.algo == 'reptile':                               # sum_k (theta_meta - theta_k) = n theta_meta - sum_k theta_k
            self._sum_n(eng, contrib, [e.params for e in engines], -1.0)
            eng.axpy(contrib, self._original, float(len(engines)))
        else:
            self._sum_n(eng, contrib, [e.grads for e in engines], 1.0)
        self.sharder.reduce_async(contrib, side_stream=False)
        self._n_reduces = getattr(self, '_n_reduces', 0) + 1
        if self._updates is None:
            self._updates = []
        self._updates.append(contrib)

    @staticmethod
    def _sum_n(eng, out, bufs, scale):
        if hasattr(eng, 'sum_n'):
            eng.sum_n(out, bufs, scale)                                # one pass (include/masr.h masr_sum_n)
        else:
            out.zero_()
            for b in bufs:
                eng.axpy(out, b, scale)

    def _draw_meta_batch(self, meta_batch):
        """Batch INDICES of every task of the meta-step, drawn in the reference's order (per task: k inner batches, then the
        val batch -- nothing else consumes the RNG streams in between, :139-145); the batches of this rank's tasks are
        assembled by the collate pool while earlier tasks already run on the GPU.  Tasks of other ranks are drawn index-only:
        that advances their samplers and the shared RNG streams exactly as their owner does, so every rank sees the data order
        of the single-process run whichever rank an accent lands on next."""
        out = []
        for pos, accent_id in enumerate(meta_batch):
            mode = 'async' if self.sharder.owns(pos) else False
            tr = self.data_container.get_item(accent_id, self.meta_k, materialize=mode)
            val = self.data_container.get_item(accent_id, materialize=mode)[0]
            if mode:
                out.append((accent_id, tr, val))
        return out

    def _shuffle_and_draw(self, task_ids):
        self._task_rng.shuffle(task_ids)                                          # :136
        meta_batch = list(task_ids[:self.meta_batch_size])
        return meta_batch, self._draw_meta_batch(meta_batch)

    # ------------------------------------------------------------------ the outer loop (:128-177)
    def train(self):
        # (random.shuffle permutes the list IN PLACE, step after step: its current order is part of a checkpoint)
        task_ids = self._task_ids = getattr(self, '_task_ids', None) or list(range(self.num_pretrain))
        nxt = None
        try:
            first_it = (self.global_step - 1) % self.eval_ival          # (a resumed run re-enters the chunk it was saved in)
            while self.global_step < self.max_step:
                for it in range(first_it, self.eval_ival):
                    first_it = 0
                    meta_batch, drawn = nxt if nxt is not None else self._shuffle_and_draw(task_ids)
                    self._ahead_state = None                            # (whatever was drawn ahead is being used now)
                    # look-ahead: the NEXT meta-step's task order and batch indices are drawn now (same RNG order as drawing
                    # them after this step: nothing in between consumes `random` / `np.random`), so that its batches are
                    # assembled in pinned memory by the collate pool while this step runs on the GPU.  Bucketed loaders only:
                    # a RandomSampler also reads the torch stream, which evaluate()'s dev loaders touch in between.
                    more = it + 1 < self.eval_ival or self.global_step + 1 < self.max_step
                    # (not across a checkpoint: save_per_steps records the RNG / sampler state "nothing drawn beyond this step")
                    ahead = more and self.is_bucket and self.data_container.pool is not None and (self.global_step + 1) % self.save_ival != 0
                    nxt = None

                    def look_ahead():                                   # runs once the first task's launches are queued: its host
                        nonlocal nxt                                    # time (index draws, HBM gathers) overlaps GPU work
                        if ahead:
                            self._ahead_state = self._stream_state()    # what a SIGINT checkpoint must hold while `nxt` is unused
                            nxt = self._shuffle_and_draw(task_ids)
                        # ... and the host now WAITS for the previous meta-step's stats, i.e. it stays at most one meta-step
                        # ahead of the GPU, which always has this step's launches queued behind the previous one's
                        self._drain_stats(keep_steps=1)
                    n_local = 0
                    if self.tasks_per_gpu > 1:
                        n_local = self._run_tasks_concurrently(drawn, look_ahead)
                    elif not drawn:
                        look_ahead()
                    for ti, (accent_id, tr, val) in enumerate(drawn if self.tasks_per_gpu == 1 else []):
                        self.run_task([(i, f.result()) for i, f in tr])
                        if ti == 0:
                            look_ahead()
                        val_batch = (val[0], val[1].result())
                        batch_size = len(val_batch[1][2])
                        info = self._train(val_batch[0], *val_batch[1], accent_idx=val_batch[0], want_info=False)
                        handle = self._clip_and_stats(info)
                        self._partial_meta_update()
                        self._pending.append((self.global_step, accent_id, handle, batch_size))
                        n_local += 1
                    self._pad_rounds(len(meta_batch), n_local)
                    self._final_meta_update(len(meta_batch))
                    if self.global_step % self.log_ival == 0 or self.global_step % self.eval_ival == 0 or (self.global_step + 1) % self.save_ival == 0:
                        self._drain_stats()                             # train_info is about to be read: this step's tasks included
                    self.log_msg(self.meta_opt.lr)
                    self.check_evaluate()
                    self.global_step += 1
                    self.dashboard.step()
                    if self.global_step % self.save_ival == 0:
                        self.save_per_steps()
        except KeyboardInterrupt:
            logger.warning("Pretraining stopped")
            self._drain_stats()
            self.save_per_steps(gather=False)
            self.dashboard.set_status('pretrained(SIGINT)')
            self.sharder.close()
        else:
            self._drain_stats()
            logger.notice("Pretraining completed")
            self.dashboard.set_status('pretrained')
            self.sharder.close()

    def _partial_meta_update(self, engine=None):
        """_updates[n] += p.grad for every parameter (:180-198) == one flat axpy.  With several ranks the freshly
        accumulated task gradient is all-reduced on the side stream while the next task runs."""
        eng = engine if engine is not None else self.asr_model.engine
        reptile = self.paras.algo == 'reptile' and self.fix_reptile
        if self.paras.algo != 'fomaml' and not reptile:
            raise ValueError(f"Not support meta algo {self.paras.algo}")    # reptile/maml: no reference implementation
        if not self.sharder.collective:
            if self._updates is None:
                self._updates = torch.zeros_like(eng.params)
            if reptile:                                                    # pseudo-gradient theta_meta - theta_k
                eng.axpy(self._updates, self._original, 1.0)
                eng.axpy(self._updates, eng.params, -1.0)
            else:
                eng.axpy(self._updates, eng.grads, 1.0)
            return
        if self._updates is None:
            self._updates = []
        if reptile:
            contrib = self._original.clone()
            eng.axpy(contrib, eng.params, -1.0)
        else:
            contrib = eng.grads.clone()                                   # per-task buffer handed to the side stream
        if not reptile and getattr(self, '_wire_clip', False):           # (_clip_and_stats left the gradient unscaled, norm on the device)
            self._wire_clip = False
            self.sharder.reduce_async(contrib, clip=(eng.grad_norm_device_ptr(), GRAD_CLIP))
            self._n_reduces = getattr(self, '_n_reduces', 0) + 1
            self._updates.append(contrib)
            return
        self.sharder.reduce_async(contrib)                              # side stream: overlaps this rank's next task
        self._n_reduces = getattr(self, '_n_reduces', 0) + 1
        self._updates.append(contrib)

    def _pad_rounds(self, n_tasks, n_local):
        """every rank must issue the same number of all-reduces per meta-step: ranks that own fewer tasks than
        ceil(n_tasks / world) contribute zero buffers for the missing rounds"""
        if not self.sharder.collective:
            return
        K = getattr(self, 'tasks_per_gpu', 1)
        rounds = (n_tasks + self.sharder.world - 1) // self.sharder.world          # tasks of the busiest rank ...
        rounds = (rounds + K - 1) // K                                               # ... = its waves = its all-reduces
        for _ in range(rounds - getattr(self, '_n_reduces', 0)):
            z = torch.zeros_like(self.asr_model.engine.params)
            self.sharder.reduce_async(z, side_stream=K == 1)
            if self._updates is None:
                self._updates = []
            self._updates.append(z)
        self._n_reduces = 0

    def _final_meta_update(self, n_tasks=None):
        """_updates /= counter; attach as .grad of the meta weights; Noam-Adam step; reset (:200-221).
        counter = number of tasks of the whole meta-step (all ranks)."""
        eng = self.asr_model.engine
        if self.sharder.collective:
            self.sharder.wait_all()
            if len(self._updates) == 1:
                self._updates = self._updates[0]
            elif len(self._updates) <= 8 and hasattr(eng, 'adam_sum_step'):
                self._task_grads = self._updates                        # Adam reads the reduced buffers directly (one pass)
            else:
                total = torch.zeros_like(eng.params)
                for c in self._updates:
                    eng.axpy(total, c, 1.0)
                self._updates = total
            counter = n_tasks
        else:
            counter = self._counter
        if getattr(self, '_task_grads', None):
            self.meta_opt.optimizer.grad_list, self.meta_opt.optimizer.grad_scale = self._task_grads, 1.0 / counter
        else:
            eng.scale(self._updates, 1.0 / counter)
            self.meta_opt.optimizer.grad = self._updates
        self.meta_opt.step()
        self.meta_opt.zero_grad()
        self._counter = 0
        self._updates = None
        self._task_grads = []

    def run_task(self, batches, engine=None):
        """:223-250 -- fresh copy of the meta weights, fresh SGD (momentum state reset per task), k inner steps."""
        if engine is None:
            self._counter += 1
        eng = engine if engine is not None else self.asr_model.engine
        eng.copy(eng.params, self._original)                              # load_state_dict(self._original)
        eng.mark_dirty()
        self.asr_model.train()
        mp = self.config['asr_model']
        if mp['inner_optimizer_cls'] != 'SGD':
            raise NotImplementedError(f"inner optimizer {mp['inner_optimizer_cls']}")
        opt = FlatSGD(eng, self.inner_lr, mp['inner_optimizer_opt']['momentum'], mp['inner_optimizer_opt']['nesterov'],
                      total_steps=len(batches))          # (dropped after these k steps, :228-250)
        if engine is None:
            self.asr_opt = opt
        for idx, (x, ilens, ys, olens) in batches:
            # the reference discards the info of the inner steps (:240): no host sync for it here, the next launches queue
            # behind this batch while it runs
            if engine is None:
                self._train(idx, x, ilens, ys, olens, want_info=False)
            else:
                self._train(idx, x, ilens, ys, olens, engine=engine, want_info=False)
            opt.clip_and_step(GRAD_CLIP)                                  # clip 5; NaN norm -> step skipped on the device

    # ------------------------------------------------------------------ evaluation (:253-298)
    def evaluate(self):
        """(:253-298).  Quirk Q2: runs on whatever `asr_model` holds -- the last task's adapted weights.  With several
        ranks that model differs per rank, so rank 0 alone evaluates (its own last task) while the others wait.
        --fix_snapshot_meta_weights: the META weights are evaluated (and saved); being rank-independent, the dev accents
        are then split over the ranks and the per-accent averages gathered."""
        sh = self.sharder
        self._drain_stats()
        if self.fix_snapshot:
            eng = self.asr_model.engine
            eng.copy(eng.params, self._original)
            eng.mark_dirty()
            mine = range(sh.rank, self.num_pretrain, sh.world)
            self._skip_dev_draws(set(range(self.num_pretrain)) - set(mine))
            self._evaluate_accents(mine, gather=sh.world > 1)
        elif sh.world > 1:
            if sh.rank == 0:
                self._evaluate_accents(range(self.num_pretrain))
            else:
                self._skip_dev_draws(range(self.num_pretrain))
            sh.barrier()
        else:
            self._evaluate_accents(range(self.num_pretrain))

    def _skip_dev_draws(self, accent_ids):
        """every dev iterator draws a base seed from the torch default generator when it is created (Loader.iter_indices, as torch's
        DataLoader does).  A rank that does not evaluate an accent takes the same draws, so that the torch stream -- which the
        non-bucketed train loaders' RandomSampler reads -- stays identical on all ranks (the index-only replay of task sharding
        relies on identical RNG consumption everywhere)."""
        for idx in accent_ids:
            self.data_container.dev_loaders[idx].iter_indices()

    def _evaluate_accents(self, accent_ids, gather=False):
        self.asr_model.eval()
        self.write_tr_logs()
        dev_info_ls = [RunningAvgDict(decay_rate=1.) for _ in range(self.num_pretrain)]
        for idx in accent_ids:
            dev_loader = self.data_container.dev_loaders[idx]
            for cur_b, (x, ilens, ys, olens) in enumerate(dev_loader):
                if ilens.max() > self.dev_max_ilen:
                    continue
                info = self._eval(idx, x, ilens, ys, olens)
                dev_info_ls[idx].add(info, len(ys))
        if gather:                                                         # every rank ends up with every accent's averages
            parts = self.sharder.all_gather_object({i: dict(dev_info_ls[i]) for i in accent_ids})
            for part in parts:
                for i, d in part.items():
                    dev_info_ls[i] = RunningAvgDict(decay_rate=1.)
                    dev_info_ls[i].add(d)
        for idx in range(self.num_pretrain):
            self.dashboard.log_info(f"dev_{self.accents[idx]}", dev_info_ls[idx])
            self.write_dev_logs(f"dev_{self.accents[idx]}", dev_info_ls[idx])
        dev_avg = RunningAvgDict(decay_rate=1.0)
        for d in dev_info_ls:
            dev_avg.add({k: float(v) for k, v in d.items()})
        self.dashboard.log_info("dev", dev_avg)
        self.write_dev_logs("dev_avg", dev_avg)
        cur_cer, cur_wer = float(dev_avg['cer']), float(dev_avg['wer'])
        if cur_wer < self.best_wer:
            self.best_wer = cur_wer
            self.save_best_model()
        if cur_cer < self.best_cer:
            self.best_cer = cur_cer
            self.save_best_model('cer', only_stat=True)
        self.asr_model.train()

    def run_batch(self, idx, x, ilens, ys, olens, train):
        raise NotImplementedError                                          # provided by the Trainer mixin
