"""get_trainer(cls, config, paras, id2accent): the Trainer mixin of the reference
(src/transformer_torch_trainer.py:13-107) over the HIP engine.  Same contract: builds
`class TransformerTrainer(cls)` at run time, provides set_model / exec / run_batch / probe_model."""
import torch

from .marcos import IGNORE_ID
from .model import MyTransformer
from .optimizer import FlatAdam, FlatRAdam, FlatSGD, TransformerOptimizer
from .monitor import logger


def get_trainer(cls, config, paras, id2accent):
    logger.notice("Transformer Trainer Init...")

    class TransformerTrainer(cls):
        def __init__(self, config, paras, id2accent):
            super().__init__(config, paras, id2accent)

        def set_model(self):
            mp = self.config['asr_model']
            self.label_smooth_rate = self.config['solver']['label_smoothing']
            device = getattr(self.paras, 'device', None) or "cuda:0"
            self.asr_model = MyTransformer(self.id2ch, mp, self.label_smooth_rate, device=device).cuda()
            eng = self.asr_model.engine
            eng.set_seed(getattr(self.paras, 'seed', 531) + 7919 * 64 * self.sharder.rank)      # dropout stream: one per rank
            if 'inner_optimizer_cls' not in mp:                              # multi-task or mono (:27)
                # one task per stream: the decoder's few-row long-reduction GEMMs run k-split (masr_set_ksplit: +3 % alone on the GPU).  The
                # FOMAML interface leaves it off for every --tasks_per_gpu so that K slots == the sequential run == N ranks, bit for bit.
                eng.set_ksplit(True)
                cls_name = mp['optimizer_cls']
                if cls_name == 'noam':
                    self.asr_opt = TransformerOptimizer(FlatAdam(eng, eng.params, betas=(0.9, 0.98), eps=1e-09),
                                                        mp['optimizer_opt']['k'], mp['d_model'], mp['optimizer_opt']['warmup_steps'])
                elif cls_name == 'SGD':
                    o = mp['optimizer_opt']
                    self.asr_opt = FlatSGD(eng, o['lr'], o.get('momentum', 0.0), o.get('nesterov', False))
                elif cls_name in ('Adam', 'AdamW'):
                    o = mp['optimizer_opt']                                  # torch defaults: Adam wd 0 (L2), AdamW wd 1e-2 (decoupled)
                    if o.get('amsgrad', False):
                        raise NotImplementedError("amsgrad")
                    wd = o.get('weight_decay', 1e-2 if cls_name == 'AdamW' else 0.0)
                    self.asr_opt = FlatAdam(eng, eng.params, betas=tuple(o.get('betas', (0.9, 0.999))), eps=o.get('eps', 1e-8),
                                            lr=o.get('lr', 1e-3), weight_decay=wd, decoupled=(cls_name == 'AdamW'))
                elif cls_name == 'RAdam':                                    # reference: torch_optimizer.RAdam(**optimizer_opt) (:36-41)
                    o = mp['optimizer_opt']
                    self.asr_opt = FlatRAdam(eng, eng.params, betas=tuple(o.get('betas', (0.9, 0.999))), eps=o.get('eps', 1e-8),
                                             lr=o.get('lr', 1e-3), weight_decay=o.get('weight_decay', 0.0))
                else:
                    raise NotImplementedError(f"optimizer_cls {cls_name} (reference: getattr(torch.optim, cls))")
            else:
                logger.notice("During meta-training, model optimizer will reset after running each task")
            self.sos_id, self.eos_id = self.asr_model.sos_id, self.asr_model.eos_id
            super().load_model()

        def exec(self):
            self.train()

        def run_batch(self, cur_b, x, ilens, ys, olens, train, accent_idx=None, engine=None, want_info=True):
            """forward + label-smoothed CE + (train) backward, gradients left in engine.grads
            (reference :59-99).  One host sync (read_stats) instead of the reference's three .item() calls.
            `engine` selects a task slot's replica (concurrent tasks per GPU); default = self.asr_model.
            want_info=False (callers that discard the returned dict, i.e. the inner steps of run_task): no host sync at all."""
            eng = engine if engine is not None else self.asr_model.engine
            eng.run_batch(x, ilens, ys, olens, train=train)
            olens += 1                                                        # quirk Q6 (mono_transformer_torch.py:139)
            if not want_info and not (train and self.global_step % 500 == 0 and engine is None):
                return None
            st = eng.read_stats()
            info = {'loss': st['loss'], 'acc': st['n_correct'] / st['n_total']}
            if train:
                if self.global_step % 500 == 0 and engine is None:
                    self.probe_model(accent_idx)
            else:
                pred, gold = eng.last_logits()
                pred, gold = pred.cpu(), gold.cpu().to(torch.int64)
                info['cer'] = self.metric_observer.batch_cal_er(pred, gold, ['att'], ['cer'])['att_cer']
                info['wer'] = self.metric_observer.batch_cal_er(pred, gold, ['att'], ['wer'])['att_wer']
            return info

        def info_from_stats(self, engine=None):
            """{'loss', 'acc'} of the last run_batch from the stats the last host sync brought back (clip_grad_norm_ reads the
            loss, the token counts and the gradient norm in ONE copy): run_batch(want_info=False) + clip + this = one sync"""
            st = (engine if engine is not None else self.asr_model.engine)._last_stats
            return {'loss': st['loss'], 'acc': st['n_correct'] / st['n_total']}

        def opt_step(self):
            """asr_opt.step() with the engine's flat gradient attached (torch optimisers read p.grad implicitly)"""
            inner = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
            if isinstance(inner, FlatAdam):
                inner.grad = self.asr_model.engine.grads
            self.asr_opt.step()

        def opt_step_guarded(self):
            """opt_step() for Adam-family optimisers with the NaN test on the device (FlatAdam.step_guarded); confirm() later"""
            inner = self.asr_opt.optimizer if isinstance(self.asr_opt, TransformerOptimizer) else self.asr_opt
            inner.grad = self.asr_model.engine.grads
            self.asr_opt.step_guarded()

        def clip_grad_norm_(self, max_norm, engine=None):
            """nn.utils.clip_grad_norm_(self.asr_model.parameters(), max_norm) -> python float (host sync)."""
            eng = engine if engine is not None else self.asr_model.engine
            eng.clip_grads(max_norm)
            return eng.read_stats()['grad_norm']

        def stats_async(self, engine=None):
            """{loss, acc} and the gradient norm of what is queued, as a handle: .get() -> (info, grad_norm) once the copy has
            landed (for loops whose clip + step already ran on the device, e.g. FlatSGD.clip_and_step)"""
            pend = (engine if engine is not None else self.asr_model.engine).read_stats_async()

            class _H:
                def get(self_):
                    st = pend.get()
                    return {'loss': st['loss'], 'acc': st['n_correct'] / st['n_total']}, st['grad_norm']
            return _H()

        def clip_stats_async(self, max_norm, engine=None, norm_only=False):
            """clip_grad_norm_ without the host sync: the clip is queued, and so is the copy of {loss, counts, norm} to the host;
            returns a handle, .get() -> (info, grad_norm) once that copy has landed (used by the meta loops to stay one
            meta-step ahead of the GPU).  norm_only: the norm is formed, the gradient is left unscaled (the caller's all-reduce
            applies the coefficient on the way out, include/masr.h masr_allreduce)"""
            eng = engine if engine is not None else self.asr_model.engine
            if norm_only:
                eng.grad_norm()
            else:
                eng.clip_grads(max_norm)
            pend = eng.read_stats_async()

            class _H:
                def get(self_):
                    st = pend.get()
                    return {'loss': st['loss'], 'acc': st['n_correct'] / st['n_total']}, st['grad_norm']
            return _H()

        def probe_model(self, accent_idx):
            pred, gold = self.asr_model.engine.last_logits()
            p0, g0 = torch.argmax(pred[0].cpu(), dim=-1), gold[0].cpu().to(torch.int64)
            logger.log(f"probe cer {self.metric_observer.cal_att_cer(p0, g0):.2f} wer {self.metric_observer.cal_att_wer(p0, g0):.2f}", prefix='debug')

    return TransformerTrainer(config, paras, id2accent)
