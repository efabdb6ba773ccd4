"""MultiASRInterface: multi-task pretraining (reference: src/multi_interface.py:17-192) -- one random-accent batch per
step, clip 5, NaN-skip / optimiser step.  With N ranks this is plain data parallelism over N random-accent batches
with a gradient all-reduce(mean), which changes the effective batch (parity holds at world_size 1 only)."""
import math
import pickle
from functools import partial

import torch

from .marcos import *  # noqa: F401,F403
from .monitor import logger
from .monitor.stat import RunningAvgDict
from .optimizer import FlatAdam, FlatSGD, TransformerOptimizer
from .pretrain_interface import PretrainInterface


class MultiASRInterface(PretrainInterface):
    def __init__(self, config, paras, id2accent):
        super().__init__(config, paras, id2accent)
        self.asr_model = self.asr_opt = None
        assert self.sample_strategy == 'normal', "Multi-task training only support normal sampling strategy"
        self.dashboard.set_status('pretraining')
        self._train = partial(self.run_batch, train=True)
        self._eval = partial(self.run_batch, train=False)

    def _sd_cpu(self):
        return {k: v.cpu() for k, v in self.asr_model.engine.state_dict(clone=False).items()}

    def save_best_model(self, tpe='wer', only_stat=False):
        if self.sharder.rank != 0:
            return
        if not only_stat:
            torch.save(self._sd_cpu(), self.log_dir.joinpath(f'model.{tpe}.best'))
        with open(self.log_dir.joinpath(f'best_{tpe}'), 'w') as fout:
            print('{} {}'.format(self.global_step, getattr(self, f'best_{tpe}')), file=fout)

    def _stream_state(self):
        from .io.dataset import capture_rng
        return {'rng': capture_rng(), 'data': self.data_container.state_dict()}

    def save_per_steps(self, gather=True):
        getattr(self, '_drain_stats', lambda: None)()
        eng = self.asr_model.engine
        # one dropout stream per rank: all of them go into rank 0's file (gather=False: the SIGINT path, this rank's only)
        mine = [eng.dropout_state()] if hasattr(eng, 'dropout_state') else []
        dropout = dict(enumerate(self.sharder.all_gather_object(mine))) if gather else {self.sharder.rank: mine}
        if self.sharder.rank != 0:
            return
        sd = self._sd_cpu()
        torch.save(sd, self.log_dir.joinpath("snapshot.latest"))
        with open(self.log_dir.joinpath("info_dict.latest"), 'wb') as f:
            pickle.dump(self.train_info, f)
        with open(self.log_dir.joinpath("global_step"), 'w') as f:
            print(self.global_step, file=f)
        torch.save(sd, self.log_dir.joinpath(f"snapshot.step.{self.global_step}"))
        # extension (the reference's pretraining cannot resume): optimiser state + the RNG streams and
        # sampler state that decide the next batches, for an exact continuation (same layout as the FOMAML interface's file)
        # (a SIGINT can land while the next step's batch is drawn but unused: `_ahead_state` = the streams before that draw)
        inner = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
        stream = getattr(self, '_ahead_state', None) or self._stream_state()
        torch.save({'opt': inner.state_dict(), 'step_num': getattr(self.asr_opt, 'step_num', None), 'rng': stream['rng'],
                    'data': stream['data'], 'best': (self.best_wer, self.best_cer), 'dropout': dropout},
                   self.log_dir.joinpath("meta_state.latest"))
        self.dashboard.log_step()

    def load_model(self):
        if self.paras.resume:
            self.asr_model.load_state_dict(torch.load(self.resume_model_path))
            self.dashboard.set_step(self.global_step)
            ms = self.log_dir.joinpath("meta_state.latest")
            if ms.exists():
                from .io.dataset import restore_rng
                st = torch.load(ms, weights_only=False)
                inner = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
                inner.load_state_dict(st['opt'])
                if st.get('step_num') is not None:
                    self.asr_opt.step_num, self.asr_opt.lr = st['step_num'], st['opt']['lr']
                self.data_container.load_state_dict(st['data'])
                self.best_wer, self.best_cer = st['best']
                dropout = st['dropout']
                if isinstance(dropout, dict):                            # per rank (older files: rank 0's list)
                    dropout = dropout.get(self.sharder.rank, [])
                elif self.sharder.rank != 0:
                    dropout = []
                for d in dropout:
                    self.asr_model.engine.set_dropout_state(d)
                restore_rng(st['rng'])

    def write_tr_logs(self):
        for k, v in self.train_info.items():
            self.write_log(f"train_{k}", float(v))

    def write_dev_logs(self, prefix, info):
        for k, v in info.items():
            self.write_log(f"{prefix}_{k}", float(v))

    def check_evaluate(self):
        if self.global_step % self.eval_ival == 0:
            self.evaluate()

    def train(self):
        eng = self.asr_model.engine

        def draw():
            # N ranks = N consecutive draws of the ONE shared accent/batch stream per step (rank r keeps draw r, the
            # others are replayed index-only): N different random-accent batches, same stream as a single process
            mine = None
            for r in range(self.sharder.world):
                item = self.data_container.get_item(materialize='async' if r == self.sharder.rank else False)[0]
                if r == self.sharder.rank:
                    mine = item
            return mine
        # the next step's batch is drawn (np.random accent + bucket sampler: nothing else reads that stream in between) and
        # assembled by the collate pool while this step runs; bucketed loaders with a pool only, as in the FOMAML loop
        ahead = self.is_bucket and self.data_container.pool is not None
        nxt = None
        # the NaN test of the gradient norm on the device, stats booked one step later (see MonoASRInterface.train)
        inner = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
        mode = 'sgd' if isinstance(self.asr_opt, FlatSGD) else 'adam' if type(inner) is FlatAdam else None
        run_ahead = mode is not None and hasattr(self, 'stats_async') and not getattr(self.paras, 'sync_stats', False)
        pending = self._pending = []

        def drain(keep=0):
            while len(pending) > keep:
                step, handle, n = pending.pop(0)
                info, grad_norm = handle.get()
                if math.isnan(grad_norm):
                    logger.warning(f"grad norm NaN @ step {step}")
                if mode == 'adam':
                    self.asr_opt.confirm(not math.isnan(grad_norm))
                self.train_info.add(info, n)
        self._drain_stats = drain
        try:
            first_it = (self.global_step - 1) % self.eval_ival          # (a resumed run re-enters the chunk it was saved in)
            while self.global_step < self.max_step:
                for it in range(first_it, self.eval_ival):
                    first_it = 0
                    idx, fut = nxt if nxt is not None else draw()
                    more = it + 1 < self.eval_ival or self.global_step + 1 < self.max_step
                    # (no draw across a checkpoint: save_per_steps records "nothing drawn beyond this step"; a SIGINT checkpoint
                    # taken while `nxt` is unused holds the streams as they stood before its draw)
                    self._ahead_state = nxt = None
                    if ahead and more and (self.global_step + 1) % self.save_ival != 0:
                        self._ahead_state = self._stream_state()
                        nxt = draw()
                    x, ilens, ys, olens = fut.result()
                    one_sync = hasattr(self, 'info_from_stats')            # {loss, acc} ride on the copy that brings the norm back
                    info = self._train(idx, x, ilens, ys, olens, accent_idx=idx, **({'want_info': False} if one_sync else {}))
                    if self.sharder.world > 1:                            # DP: mean gradient over ranks
                        self.sharder.all_reduce(eng.grads)
                        eng.scale(eng.grads, 1.0 / self.sharder.world)
                    if run_ahead and info is None:
                        if mode == 'sgd':
                            self.asr_opt.clip_and_step(GRAD_CLIP)
                        else:
                            eng.clip_grads(GRAD_CLIP)
                            self.opt_step_guarded()
                        pending.append((self.global_step, self.stats_async(), len(ys)))
                        due = self.global_step % self.log_ival == 0 or self.global_step % self.eval_ival == 0 or (self.global_step + 1) % self.save_ival == 0
                        drain(keep=0 if due else 1)
                    else:
                        drain()
                        grad_norm = self.clip_grad_norm_(GRAD_CLIP)
                        if info is None:
                            info = self.info_from_stats()
                        self.train_info.add(info, len(ys))
                        if math.isnan(grad_norm):
                            logger.warning(f"grad norm NaN @ step {self.global_step}")
                        else:
                            self.opt_step()
                    self.log_msg(self.asr_opt.lr if isinstance(self.asr_opt, TransformerOptimizer) else None)
                    self.check_evaluate()
                    self.global_step += 1
                    self.dashboard.step()
                    if self.global_step % self.save_ival == 0:
                        self.save_per_steps()
            drain()
        except KeyboardInterrupt:
            logger.warning("Pretraining stopped")
            drain()
            self.save_per_steps(gather=False)
            self.dashboard.set_status('pretrained(SIGINT)')
        else:
            logger.notice("Pretraining completed")
            self.dashboard.set_status('pretrained')

    def evaluate(self):
        getattr(self, '_drain_stats', lambda: None)()
        self.asr_model.eval()
        self.write_tr_logs()
        dev_info_ls = [RunningAvgDict(decay_rate=1.) for _ in range(self.num_pretrain)]
        for idx, dev_loader in enumerate(self.data_container.dev_loaders):
            for x, ilens, ys, olens in dev_loader:
                if ilens.max() > self.dev_max_ilen:
                    continue
                dev_info_ls[idx].add(self._eval(idx, x, ilens, ys, olens), len(ys))
            self.dashboard.log_info(f"dev_{self.accents[idx]}", dev_info_ls[idx])
            self.write_dev_logs(f"dev_{self.accents[idx]}", dev_info_ls[idx])
        dev_avg = RunningAvgDict(decay_rate=1.0)
        for d in dev_info_ls:
            dev_avg.add({k: float(v) for k, v in d.items()})
        self.dashboard.log_info("dev", dev_avg)
        self.write_dev_logs("dev_avg", dev_avg)
        if float(dev_avg['wer']) < self.best_wer:
            self.best_wer = float(dev_avg['wer'])
            self.save_best_model()
        if float(dev_avg['cer']) < self.best_cer:
            self.best_cer = float(dev_avg['cer'])
            self.save_best_model('cer', only_stat=True)
        self.asr_model.train()

    def run_batch(self, idx, x, ilens, ys, olens, train):
        raise NotImplementedError
