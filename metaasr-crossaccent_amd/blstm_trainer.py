"""get_trainer(cls, config, paras, id2accent) for `--model_name blstm`: the Trainer mixin of the reference
(src/blstm_trainer.py:12-92) over libmasr's BLSTM-CTC engine (masr_blstm_*).  Same contract as the reference: builds
`class BLSTMTrainer(cls)` at run time and provides set_model / exec / run_batch / probe_model / freeze_encoder."""
import torch

from .blstm_engine import MonoBLSTM
from .monitor import logger
from .optimizer import FlatSGD


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='min', factor, patience) with torch's defaults
    (threshold 1e-4 relative, cooldown 0, min_lr 0, eps 1e-8) on a flat optimiser's param_groups."""

    def __init__(self, optimizer, factor=0.2, patience=3, threshold=1e-4, min_lr=0.0, eps=1e-8):
        self.optimizer, self.factor, self.patience, self.threshold, self.min_lr, self.eps = optimizer, factor, patience, threshold, min_lr, eps
        self.best, self.num_bad_epochs = float('inf'), 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = metric, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for g in self.optimizer.param_groups:
                new = max(g['lr'] * self.factor, self.min_lr)
                if g['lr'] - new > self.eps:
                    logger.notice(f"reducing learning rate to {new:.4e}")
                    g['lr'] = new
            self.num_bad_epochs = 0


def get_trainer(cls, config, paras, id2accent):
    logger.notice("BLSTM Trainer Init...")

    class BLSTMTrainer(cls):
        def __init__(self, config, paras, id2accent):
            super().__init__(config, paras, id2accent)

        def set_model(self):
            mp = self.config['asr_model']
            device = getattr(self.paras, 'device', None) or "cuda:0"
            self.asr_model = MonoBLSTM(self.id2ch, mp, device=device).cuda()
            self.sos_id, self.eos_id = self.asr_model.sos_id, self.asr_model.eos_id
            if mp['optimizer']['type'] != 'SGD':
                raise NotImplementedError(f"optimizer {mp['optimizer']['type']} (the shipped config/blstm files use SGD)")
            o = mp['optimizer_opt']
            if o.get('weight_decay', 0) or o.get('dampening', 0):
                raise NotImplementedError("SGD weight_decay / dampening")
            self.asr_opt = FlatSGD(self.asr_model.engine, o['lr'], o.get('momentum', 0.0), o.get('nesterov', False))
            self.lr_scheduler = ReduceLROnPlateau(self.asr_opt, factor=0.2, patience=3)        # blstm_trainer.py:32-35
            super().load_model()
            self.freeze_encoder(getattr(self.paras, 'freeze_layer', None))

        def freeze_encoder(self, module):
            """blstm_trainer.py:39-51: requires_grad = False == a 0/1 mask on the flat gradient"""
            if module is None:
                return
            if module not in ('VGG', 'VGG_BLSTM'):
                raise ValueError(f"Unknown freeze layer {module} (VGG, VGG_BLSTM)")
            eng = self.asr_model.engine
            prefix = 'encoder.vgg.' if module == 'VGG' else 'encoder.'
            if self.frozen_mask is None:
                self.frozen_mask = torch.ones_like(eng.params)
            for n in eng.table:
                if n.startswith(prefix):
                    eng.view(n, self.frozen_mask).zero_()

        def exec(self):
            self.train()

        def run_batch(self, cur_b, x, ilens, ys, olens, train, accent_idx=None):
            """forward + CTC loss (+ backward) -- blstm_trainer.py:55-85.  Targets [sos] + y + [eos] are built inside the C call."""
            eng = self.asr_model.engine
            eng.run_batch(x, ilens, ys, olens, train=train)
            olens += 2                                                        # the reference mutates olens (pad <sos> and <eos>)
            st = eng.read_stats()
            if train:
                info = {'loss': st['loss']}
                if self.global_step % 500 == 0:
                    self.probe_model(ys)
            else:
                logits, _ = eng.last_logits()
                pred = logits.cpu()
                info = {'cer': self.metric_observer.batch_cal_er(pred, ys, ['ctc'], ['cer'])['ctc_cer'],
                        'wer': self.metric_observer.batch_cal_er(pred, ys, ['ctc'], ['wer'])['ctc_wer'], 'loss': st['loss']}
            return info

        def probe_model(self, ys):
            try:
                logits, _ = self.asr_model.engine.last_logits()
                hyp = torch.argmax(logits[0].cpu(), dim=-1)
                self.metric_observer.cal_ctc_cer(hyp, ys[0], show=True, show_decode=True)
                self.metric_observer.cal_ctc_wer(hyp, ys[0], show=True)
            except Exception as e:                                            # the sentencepiece model may be absent in dry runs
                logger.warning(f"probe skipped: {e}")

        def opt_step(self):
            self.asr_opt.step()

        def clip_grad_norm_(self, max_norm, engine=None):
            eng = engine if engine is not None else self.asr_model.engine
            eng.clip_grads(max_norm)
            return eng.read_stats()['grad_norm']

    return BLSTMTrainer(config, paras, id2accent)
