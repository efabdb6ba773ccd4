"""Mono-accent training / fine-tuning (reference: src/train_interface.py:16-179 + src/mono_interface.py:18-227):
epoch loop over one accent's shard, optional initialisation from a pretraining snapshot restricted to
solver.pretrain_module, optional frozen modules, per-epoch checkpoints, best-WER model selection."""
import math
import pickle
from functools import partial
from pathlib import Path
from shutil import rmtree

import torch

from .io.dataset import get_loader
from .marcos import *  # noqa: F401,F403
from .monitor import logger
from .monitor.dashboard import Dashboard
from .monitor.stat import RunningAvgDict
from .optimizer import FlatAdam, FlatSGD, TransformerOptimizer
from .parallel import TaskSharder
from .pretrain_interface import load_units


class TrainInterface:
    """paths, vocabulary, resume bookkeeping (train_interface.py:16-179)"""

    def __init__(self, config, paras, id2accent):
        self.config, self.paras = config, paras
        self.train_type = 'evaluation'
        sv = config['solver']
        self.is_memmap, self.is_bucket, self.model_name = paras.is_memmap, paras.is_bucket, paras.model_name
        self.eval_ival, self.log_ival = sv['eval_ival'], sv['log_ival']
        self.half_batch_ilen = sv['half_batch_ilen']
        self.dev_max_ilen = sv['dev_max_ilen']
        self.best_wer = self.best_cer = INIT_BEST_ER
        self.sharder = TaskSharder.from_env()
        self.id2units = load_units(config, paras.model_name)
        self._metric = None                      # sentencepiece model is loaded on first use (evaluation / probe)
        self.save_verbose = paras.save_verbose
        cwd = Path.cwd()
        if paras.pretrain:
            assert paras.pretrain_suffix or paras.pretrain_model_path, "You should specify pretrain model and the corresponding prefix"
            if paras.pretrain_model_path:
                self.pretrain_model_path = Path(paras.pretrain_model_path)
            else:
                assert paras.pretrain_suffix and paras.pretrain_setting and paras.pretrain_step > 0, "Should specify pretrain_setting"
                self.pretrain_model_path = Path(cwd, LOG_DIR, 'pretrain', paras.pretrain_setting, paras.algo, paras.pretrain_suffix,
                                                id2accent[paras.pretrain_tgt_accent], str(paras.pretrain_runs),
                                                f"snapshot.step.{paras.pretrain_step}")
            assert self.pretrain_model_path.exists(), f"Pretrain model path {self.pretrain_model_path} not exists"
            self.pretrain_module = sv['pretrain_module']
        else:
            assert paras.pretrain_suffix is None and paras.algo == 'no', \
                f"Training from scratch shouldn't have meta-learner {paras.algo} and pretrain_suffix"
            paras.pretrain_suffix = paras.eval_suffix
        self.accent = id2accent[paras.accent]
        self.data_dir = Path(sv['data_root'], self.accent)
        self.log_dir = Path(cwd, LOG_DIR, self.train_type, sv['setting'], paras.algo, paras.pretrain_suffix, paras.eval_suffix,
                            self.accent, str(paras.runs))
        if not paras.resume:
            if self.log_dir.exists():
                assert paras.overwrite, f"Path exists ({self.log_dir}). Use --overwrite or change suffix"
                rmtree(self.log_dir)
            self.log_dir.mkdir(parents=True)
            self.train_info = RunningAvgDict(decay_rate=0.99)
            self.global_step, self.ep = 1, 0
        else:
            self.resume_model_path = self.log_dir.joinpath('snapshot.latest')
            self.optimizer_path = self.log_dir.joinpath('optimizer.latest')
            assert self.optimizer_path.exists(), f"Optimizer state {self.optimizer_path} not exists..."
            self.ep = int(Path(self.log_dir, 'epoch').read_text().strip())
            self.global_step = int(Path(self.log_dir, 'global_step').read_text().strip())
            self.best_wer = float(Path(self.log_dir, 'best_wer').read_text().strip().split(' ')[1])
            self.best_cer = float(Path(self.log_dir, 'best_cer').read_text().strip().split(' ')[1])
            assert self.resume_model_path.exists(), f"{self.resume_model_path} not exists..."
            with open(self.log_dir.joinpath('info_dict.latest'), 'rb') as fin:
                self.train_info = pickle.load(fin)
        self.dashboard = Dashboard(config, paras, self.log_dir, self.train_type, paras.resume)

    @property
    def metric_observer(self):
        if self._metric is None:
            from .monitor.metric import Metric
            sos = 0 if self.paras.model_name == 'transformer' else len(self.id2units) - 1
            self._metric = Metric(self.config['solver']['spm_model'], self.id2units, sos, len(self.id2units) - 1)
        return self._metric

    def load_data(self):
        self.id2ch = self.id2units
        sv = self.config['solver']
        dev = getattr(self.paras, 'hbm_shards_device', None)
        self.train_set = get_loader(self.data_dir.joinpath('train'), batch_size=sv['batch_size'], min_ilen=sv['min_ilen'],
                                    max_ilen=sv['max_ilen'], half_batch_ilen=sv['half_batch_ilen'], bucket_reverse=False,
                                    is_memmap=self.is_memmap, is_bucket=self.is_bucket, num_workers=self.paras.njobs,
                                    split_rate=getattr(self.paras, 'split_rate', 1.0), device=dev)
        self.dev_set = get_loader(self.data_dir.joinpath('dev'), batch_size=sv['dev_batch_size'], is_memmap=self.is_memmap,
                                  is_bucket=False, shuffle=False, num_workers=self.paras.njobs, device=dev)

    def write_log(self, k, v):
        with open(self.log_dir.joinpath(k), 'a') as fout:
            print(f'{self.global_step} {v}', file=fout)

    def log_msg(self, lr=None):
        if self.global_step % self.log_ival == 0:
            logger.log_info(self.train_info, prefix='train')
            self.dashboard.log_info('train', self.train_info)
            if lr is not None:
                self.dashboard.log_other('lr', lr)

    def write_logs(self, dev_info):
        for k, v in dev_info.items():
            self.write_log(f"dev_{k}", float(v))
        for k, v in self.train_info.items():
            self.write_log(f"train_{k}", float(v))


class MonoASRInterface(TrainInterface):
    def __init__(self, config, paras, id2accent):
        super().__init__(config, paras, id2accent)
        self.asr_model = self.asr_opt = None
        self.max_epoch = config['solver']['total_epochs']
        self.dashboard.set_status('training')
        self._train = partial(self.run_batch, train=True)
        self._eval = partial(self.run_batch, train=False)

    def _sd_cpu(self):
        return {k: v.cpu() for k, v in self.asr_model.engine.state_dict(clone=False).items()}

    def save_per_epoch(self):
        """mono_interface.py:34-59: snapshot.latest, optimizer.latest, info_dict.latest, epoch, global_step (+ snapshot.ep.N)."""
        getattr(self, '_drain_stats', lambda: None)()
        torch.save(self._sd_cpu(), self.log_dir.joinpath("snapshot.latest"))
        opt = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
        # beyond the reference's file (optimiser only): the RNG streams + the train sampler's bucket arrangement, which together
        # decide the next epoch's batches, and the dropout stream's position -- `--resume` then continues the run exactly
        from .io.dataset import capture_rng
        bs = getattr(self.train_set, 'batch_sampler', None)
        eng = self.asr_model.engine
        state = {'opt': opt.state_dict(), 'step_num': getattr(self.asr_opt, 'step_num', None), 'rng': capture_rng(),
                 'sampler': bs.state_dict() if bs is not None else None,
                 'dropout': eng.dropout_state() if hasattr(eng, 'dropout_state') else None}
        with open(self.log_dir.joinpath("optimizer.latest"), 'wb') as f:
            pickle.dump(state, f)
        with open(self.log_dir.joinpath("info_dict.latest"), 'wb') as f:
            pickle.dump(self.train_info, f)
        with open(self.log_dir.joinpath("epoch"), 'w') as f:
            print(self.ep, file=f)
        with open(self.log_dir.joinpath("global_step"), 'w') as f:
            print(self.global_step, file=f)
        if self.save_verbose:
            torch.save(self._sd_cpu(), self.log_dir.joinpath(f"snapshot.ep.{self.ep}"))

    def save_best_model(self, tpe='wer', only_stat=False):
        if not only_stat:
            torch.save(self._sd_cpu(), self.log_dir.joinpath(f'model.{tpe}.best'))
        with open(self.log_dir.joinpath(f'best_{tpe}'), 'w') as fout:
            print('{} {}'.format(self.global_step, getattr(self, f'best_{tpe}')), file=fout)

    def filter_model(self, state_dict):
        """keep tensors whose first path component is listed in solver.pretrain_module (mono_interface.py:75-81)"""
        return {k: v for k, v in state_dict.items() if k.split('.')[0] in self.pretrain_module}

    def load_model(self):
        """resume | initialise from a pretraining snapshot | scratch (mono_interface.py:83-115)"""
        eng = self.asr_model.engine
        self.frozen_mask = None
        if self.paras.resume:
            eng.load_state_dict(torch.load(self.resume_model_path))
            with open(self.optimizer_path, 'rb') as f:
                st = pickle.load(f)
            opt = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
            opt.load_state_dict(st['opt'])
            if st.get('step_num') is not None:
                self.asr_opt.step_num, self.asr_opt.lr = st['step_num'], st['opt']['lr']
            if 'rng' in st:
                from .io.dataset import restore_rng
                if st['sampler'] is not None:
                    self.train_set.batch_sampler.load_state_dict(st['sampler'])
                if st['dropout'] is not None:
                    eng.set_dropout_state(st['dropout'])
                restore_rng(st['rng'])
            self.dashboard.set_step(self.global_step)
        elif self.paras.pretrain:
            cur = eng.state_dict()
            cur.update(self.filter_model(torch.load(self.pretrain_model_path)))
            eng.load_state_dict(cur)
            freeze = self.config['solver'].get('freeze_module')
            if freeze:
                # requires_grad = False in the reference == a 0/1 mask on the flat gradient here
                inner = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
                if getattr(inner, 'weight_decay', 0.0):
                    # torch skips parameters without a gradient entirely (no decay either); the flat pass would decay them
                    raise NotImplementedError("frozen modules together with weight decay")
                self.frozen_mask = torch.ones_like(eng.params)
                for n in eng.table:
                    if n.split('.')[0] in freeze:
                        eng.view(n, self.frozen_mask).zero_()

    def check_evaluate(self):
        if self.global_step % self.eval_ival == 0:
            self.evaluate()

    def train(self):
        eng = self.asr_model.engine
        if self.paras.resume:
            # the reference evaluates at every start (:131), also of a resumed run; an uninterrupted run has no evaluation at this
            # point, so the draws its dev iterator takes from the torch stream are given back (exact continuation of the data order)
            from .io.dataset import capture_rng, restore_rng
            rng = capture_rng()
            self.evaluate()
            restore_rng(rng)
        else:
            self.evaluate()
        try:
            if self.save_verbose:                                            # save_init (:124-127)
                torch.save(self._sd_cpu(), self.log_dir.joinpath("snapshot.init"))
            # Nothing has to come back from the GPU before the next batch is queued: the NaN test of the gradient norm runs on the
            # device -- SGD (the shipped adapt configs): clip + test + step are one pass (FlatSGD.clip_and_step); Adam / AdamW / Noam:
            # the step kernel skips itself on a NaN norm and is given the scalars for both "the step before was applied" and "was
            # skipped" (FlatAdam.step_guarded) -- and {loss, acc, norm} are copied asynchronously and booked ONE step later, when the
            # optimiser's counters are confirmed too.  train.py --sync_stats: a read-back per step instead.  Same weights either way.
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
            while self.ep < self.max_epoch:
                for cur_b, (x, ilens, ys, olens) in enumerate(self.train_set):
                    # one host sync per step where the trainer can defer the batch's {loss, acc} to the copy that brings the
                    # gradient norm back (the NaN test below needs that one on the host before the optimiser may step)
                    one_sync = hasattr(self, 'info_from_stats')
                    info = self._train(cur_b, x, ilens, ys, olens, want_info=False) if one_sync else self._train(cur_b, x, ilens, ys, olens)
                    if self.frozen_mask is not None:
                        eng.grads.mul_(self.frozen_mask)
                    if run_ahead and info is None:
                        if mode == 'sgd':
                            self.asr_opt.clip_and_step(GRAD_CLIP)
                        else:
                            eng.clip_grads(GRAD_CLIP)
                            self.opt_step_guarded()
                        pending.append((self.global_step, self.stats_async(), len(ys)))
                        drain(keep=0 if (self.global_step % self.log_ival == 0 or self.global_step % self.eval_ival == 0) else 1)
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
                drain()
                self.ep += 1
                self.save_per_epoch()
                if getattr(self.paras, 'eval_every_epoch', False):           # train.py --eval_every_epoch (:169-170)
                    self.evaluate()
        except KeyboardInterrupt:
            logger.warning("Training stopped")
            self.evaluate()
            self.dashboard.set_status('trained(SIGINT)')
        else:
            logger.notice("Training completed")
            self.dashboard.set_status('trained')

    def evaluate(self):
        getattr(self, '_drain_stats', lambda: None)()                 # (train_info is written out below)
        self.asr_model.eval()
        dev_info = RunningAvgDict(decay_rate=1.)
        for x, ilens, ys, olens in self.dev_set:
            if ilens.max() > self.dev_max_ilen:
                continue
            dev_info.add(self._eval(0, x, ilens, ys, olens), len(ys))
        self.dashboard.log_info('dev', dev_info)
        self.write_logs(dev_info)
        if float(dev_info['wer']) < self.best_wer:
            self.best_wer = float(dev_info['wer'])
            self.save_best_model()
        if float(dev_info['cer']) < self.best_cer:
            self.best_cer = float(dev_info['cer'])
            self.save_best_model('cer', only_stat=True)
        if getattr(self, 'lr_scheduler', None) is not None:                  # mono_interface.py:220-221 (BLSTM: ReduceLROnPlateau)
            self.lr_scheduler.step(float(dev_info['loss']))
        self.asr_model.train()

    def run_batch(self, cur_b, x, ilens, ys, olens, train):
        raise NotImplementedError
