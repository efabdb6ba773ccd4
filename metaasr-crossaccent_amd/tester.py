"""Tester: greedy decoding of the test shard into `<log_dir>/<decode_suffix>/best-hyp` (reference: src/tester.py:18-273).
Line format "<ref ids> TAB <hyp ids>" (space separated), `trim` = cut at the first </s> after position 0 -- unchanged, so
translate.py / score.sh of the reference run on these files as they are.  Only the transformer + greedy path exists in
the reference for this model (beam search raises NotImplementedError there, tester.py:121-124)."""
from pathlib import Path
from shutil import rmtree

import torch

from .io.dataset import get_loader
from .marcos import *  # noqa: F401,F403
from .model import MyTransformer
from .monitor import logger
from .pretrain_interface import load_units


class Tester:
    def __init__(self, config, paras, id2accent):
        self.config, self.paras = config, paras
        self.train_type = 'evaluation'
        self.is_memmap, self.model_name = paras.is_memmap, paras.model_name
        if paras.algo == 'no' and paras.pretrain_suffix is None:
            paras.pretrain_suffix = paras.eval_suffix
        self.data_dir = Path(config['solver']['data_root'], id2accent[paras.accent])
        self.log_dir = Path(Path.cwd(), LOG_DIR, self.train_type, config['solver']['setting'], paras.algo, paras.pretrain_suffix,
                            paras.eval_suffix, id2accent[paras.accent], str(paras.runs))
        self.model_path = Path(self.log_dir, paras.test_model)
        assert self.model_path.exists(), f"{self.model_path.as_posix()} not exists..."
        self.decode_dir = Path(self.log_dir, paras.decode_suffix)
        self.decode_mode = paras.decode_mode
        self.batch_size = paras.decode_batch_size
        if not paras.resume:
            if self.decode_dir.exists():
                assert paras.overwrite, f"Path exists ({self.decode_dir}). Use --overwrite or change decode suffix"
                rmtree(self.decode_dir)
            self.decode_dir.mkdir(parents=True)
            self.prev_decode_step = -1
        else:
            with open(Path(self.decode_dir, 'best-hyp')) as f:
                self.prev_decode_step = sum(1 for _ in f)

    def load_data(self):
        if self.model_name not in ('transformer', 'blstm'):
            raise NotImplementedError
        self.id2ch = load_units(self.config, self.model_name)
        self.eval_set = get_loader(self.data_dir.joinpath('test'), batch_size=self.batch_size,
                                   half_batch_ilen=512 if self.batch_size > 1 else None, is_memmap=self.is_memmap,
                                   is_bucket=False, shuffle=False, num_workers=1)

    def set_model(self):
        device = getattr(self.paras, 'device', None) or "cuda:0"
        if self.model_name == 'blstm':
            from .blstm_engine import MonoBLSTM
            self.asr_model = MonoBLSTM(self.id2ch, self.config['asr_model'], device=device, init=False)
            self.asr_model.load_state_dict(torch.load(self.model_path, map_location='cpu'))
            self.asr_model.eval()
            self.sos_id, self.eos_id, self.blank_id = self.asr_model.sos_id, self.asr_model.eos_id, self.asr_model.blank_id
            return
        self.blank_id = None
        self.asr_model = MyTransformer(self.id2ch, self.config['asr_model'], device=device, init=False)
        self.asr_model.load_state_dict(torch.load(self.model_path, map_location='cpu'))
        self.asr_model.eval()
        self.sos_id, self.eos_id = self.asr_model.sos_id, self.asr_model.eos_id

    def trim(self, hyp):
        """tester.py:189-207 (transformer): everything from the first </s> at position >= 1 is dropped; a
        hypothesis of length <= 1 becomes empty."""
        assert isinstance(hyp, list)
        if self.model_name == 'blstm':                         # tester.py:191-193
            return [i for i in hyp if i < self.eos_id]
        if len(hyp) <= 1:
            return []
        for pos in range(1, len(hyp)):
            if hyp[pos] == self.eos_id:
                return hyp[:pos]
        return hyp

    def batch_greedy_decode(self, xs, ilens, ys, olens):
        if self.model_name == 'blstm':
            # tester.py:216-225: arg-max over ALL T' frames of the padded batch (frames past enc_lens included, as the
            # reference does), trim, collapse repeats, drop blanks
            from itertools import groupby
            logits, _ = self.asr_model(xs, ilens)
            preds = torch.argmax(logits, dim=-1).cpu()
            for pred, y in zip(preds, ys):
                hyp = [x[0] for x in groupby(self.trim(pred.tolist()))]
                self.write_hyp(y.tolist(), [x for x in hyp if x != self.blank_id])
            return True
        preds = self.asr_model.recog(xs, ilens).transpose(0, 1).cpu()
        for pred, y in zip(preds, ys):
            self.write_hyp(y.tolist(), self.trim(pred.tolist()))
        return True

    def write_hyp(self, y, hyp):
        if getattr(self, '_skip_lines', 0) > 0:                  # utterance already in best-hyp (resumed inside a batch)
            self._skip_lines -= 1
            return
        with open(Path(self.decode_dir, 'best-hyp'), 'a') as fout:
            fout.write("{}\t{}\n".format(" ".join(str(i) for i in y), " ".join(str(i) for i in hyp)))

    def exec(self):
        if self.decode_mode != 'greedy':
            raise NotImplementedError(f"{self.decode_mode} haven't supported yet")      # as the reference (tester.py:121-124)
        logger.notice(f"Start greedy decoding: {len(self.eval_set)} batches of <= {self.batch_size}")
        # --resume: prev_decode_step counts the LINES (utterances) already in best-hyp.  The reference's batch path does not
        # skip at all (tester.py:149-152: a resumed batch decode appends everything again); its per-utterance path skips
        # by step.  Here whole batches are skipped while all their utterances are already written; a partially written
        # batch (killed between two write_hyp calls) is decoded again and only its missing tail is appended.
        done = max(self.prev_decode_step, 0)
        seen = 0
        for idxs in self.eval_set.iter_indices():
            n = len(idxs)
            if seen + n <= done:
                seen += n
                continue
            self._skip_lines = done - seen if seen < done else 0
            self.batch_greedy_decode(*self.eval_set.materialize(idxs))
            seen += n
        self._skip_lines = 0
