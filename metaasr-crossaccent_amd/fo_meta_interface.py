"""FOMetaASRInterface: the first-order meta-learning loop of the reference (src/fo_meta_interface.py:18-302) on the
flat HBM buffers of the HIP engine.

  _original : flat fp32 copy of the meta weights            (reference: OrderedDict of cloned tensors, :103)
  _updates  : flat fp32 meta-gradient accumulator            (reference: dict of per-tensor accumulators, :184-192)
  run_task  : copy meta -> model (one 99.5 MB D2D copy), fresh SGD, k x {run_batch, clip 5, NaN ? skip : step}
  train     : per task run_task + val-batch gradient + clip + accumulate; then /= counter and Noam-Adam
Multi-GPU: tasks of a meta-step are split round-robin over ranks and `_updates` is all-reduced (parallel.py).
Reference quirks kept by default: snapshots hold the LAST TASK's adapted weights (Q1/F8),
evaluation runs on them too (Q2), a NaN val gradient is still accumulated (Q5); --fix_* flags opt out."""
import math
import pickle
from functools import partial

import torch

from .marcos import *  # noqa: F401,F403
from .monitor import logger
from .monitor.stat import RunningAvgDict
from .optimizer import FlatAdam, FlatSGD, TransformerOptimizer
from .pretrain_interface import PretrainInterface


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
        mp = config['asr_model']
        o = mp['meta']['optimizer_opt']
        self.inner_lr = mp['d_model'] ** (-0.5) * o['k'] * (o['warmup_steps'] ** (-0.5))         # :41-45
        self.fix_snapshot = bool(getattr(paras, 'fix_snapshot_meta_weights', False))
        self._task_rng = self.sharder.task_rng(getattr(paras, 'seed', 531))
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

    def save_per_steps(self):
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
        torch.save({'original': self._original.cpu(), 'adam': self.meta_opt.optimizer.state_dict(),
                    'step_num': self.meta_opt.step_num}, self.log_dir.joinpath("meta_state.latest"))
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
        ms = self.log_dir.joinpath("meta_state.latest")
        if self.paras.resume and ms.exists():
            st = torch.load(ms)
            self._original.copy_(st['original'])
            self.meta_opt.optimizer.load_state_dict(st['adam'])
            self.meta_opt.step_num = st['step_num']

    def write_tr_logs(self):
        for k, v in self.train_info.items():
            self.write_log(f"train_{k}", float(v))

    def write_dev_logs(self, prefix, info):
        for k, v in info.items():
            self.write_log(f"{prefix}_{k}", float(v))

    def check_evaluate(self):
        if self.global_step % self.eval_ival == 0:
            self.evaluate()

    # ------------------------------------------------------------------ the outer loop (:128-177)
    def train(self):
        task_ids = list(range(self.num_pretrain))
        try:
            while self.global_step < self.max_step:
                for _ in range(self.eval_ival):
                    self._task_rng.shuffle(task_ids)
                    n_local = 0
                    meta_batch = task_ids[:self.meta_batch_size]
                    for accent_id in self.sharder.my_tasks(meta_batch):
                        tr_batches = self.data_container.get_item(accent_id, self.meta_k)
                        self.run_task(tr_batches)
                        val_batch = self.data_container.get_item(accent_id)[0]
                        batch_size = len(val_batch[1][2])
                        info = self._train(val_batch[0], *val_batch[1], accent_idx=val_batch[0])
                        grad_norm = self.clip_grad_norm_(GRAD_CLIP)
                        if math.isnan(grad_norm):
                            logger.warning(f"grad norm NaN @ step {self.global_step} on {self.accents[accent_id]}, ignore...")
                        self._partial_meta_update()
                        self.train_info.add(info, batch_size)
                        n_local += 1
                    self._pad_rounds(len(meta_batch), n_local)
                    self._final_meta_update(len(meta_batch))
                    self.log_msg(self.meta_opt.lr)
                    self.check_evaluate()
                    self.global_step += 1
                    self.dashboard.step()
                    if self.global_step % self.save_ival == 0:
                        self.save_per_steps()
        except KeyboardInterrupt:
            logger.warning("Pretraining stopped")
            self.save_per_steps()
            self.dashboard.set_status('pretrained(SIGINT)')
        else:
            logger.notice("Pretraining completed")
            self.dashboard.set_status('pretrained')

    def _partial_meta_update(self):
        """_updates[n] += p.grad for every parameter (:180-198) == one flat axpy.  With several ranks the freshly
        accumulated task gradient is all-reduced on the side stream while the next task runs."""
        eng = self.asr_model.engine
        if self.paras.algo != 'fomaml':
            raise ValueError(f"Not support meta algo {self.paras.algo}")    # reptile/maml: no reference implementation
        if self.sharder.world == 1:
            if self._updates is None:
                self._updates = torch.zeros_like(eng.params)
            eng.axpy(self._updates, eng.grads, 1.0)
            return
        if self._updates is None:
            self._updates = []
        contrib = eng.grads.clone()                                       # per-task buffer handed to the side stream
        self.sharder.reduce_async(contrib)
        self._updates.append(contrib)

    def _pad_rounds(self, n_tasks, n_local):
        """every rank must issue the same number of all-reduces per meta-step: ranks that own fewer tasks than
        ceil(n_tasks / world) contribute zero buffers for the missing rounds"""
        if self.sharder.world == 1:
            return
        rounds = (n_tasks + self.sharder.world - 1) // self.sharder.world
        for _ in range(rounds - n_local):
            z = torch.zeros_like(self.asr_model.engine.params)
            self.sharder.reduce_async(z)
            if self._updates is None:
                self._updates = []
            self._updates.append(z)

    def _final_meta_update(self, n_tasks=None):
        """_updates /= counter; attach as .grad of the meta weights; Noam-Adam step; reset (:200-221).
        counter = number of tasks of the whole meta-step (all ranks)."""
        eng = self.asr_model.engine
        if self.sharder.world > 1:
            self.sharder.wait_all()
            total = torch.zeros_like(eng.params)
            for c in self._updates:
                eng.axpy(total, c, 1.0)
            self._updates = total
            counter = n_tasks
        else:
            counter = self._counter
        eng.scale(self._updates, 1.0 / counter)
        self.meta_opt.optimizer.grad = self._updates
        self.meta_opt.step()
        self.meta_opt.zero_grad()
        self._counter = 0
        self._updates = None

    def run_task(self, batches):
        """:223-250 -- fresh copy of the meta weights, fresh SGD (momentum state reset per task), k inner steps."""
        self._counter += 1
        eng = self.asr_model.engine
        eng.copy(eng.params, self._original)                              # load_state_dict(self._original)
        eng.mark_dirty()
        self.asr_model.train()
        mp = self.config['asr_model']
        if mp['inner_optimizer_cls'] != 'SGD':
            raise NotImplementedError(f"inner optimizer {mp['inner_optimizer_cls']}")
        self.asr_opt = FlatSGD(eng, self.inner_lr, mp['inner_optimizer_opt']['momentum'], mp['inner_optimizer_opt']['nesterov'])
        for idx, (x, ilens, ys, olens) in batches:
            self._train(idx, x, ilens, ys, olens)
            self.asr_opt.clip_and_step(GRAD_CLIP)                         # clip 5; NaN norm -> step skipped on the device

    # ------------------------------------------------------------------ evaluation (:253-298)
    def evaluate(self):
        self.asr_model.eval()
        self.write_tr_logs()
        dev_info_ls = [RunningAvgDict(decay_rate=1.) for _ in range(self.num_pretrain)]
        for idx, dev_loader in enumerate(self.data_container.dev_loaders):
            for cur_b, (x, ilens, ys, olens) in enumerate(dev_loader):
                if ilens.max() > self.dev_max_ilen:
                    continue
                info = self._eval(idx, x, ilens, ys, olens)
                dev_info_ls[idx].add(info, len(ys))
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
