"""Eval metric path (reference: src/monitor/metric.py:24-87): teacher-forced argmax -> trim at the first
</s> -> sentencepiece DecodePieces -> Levenshtein / reference length * 100.  `editdistance` (C extension the
reference imports) is replaced by libmasr's masr_edit_distance (host C++)."""
import ctypes as C

import torch

from .. import _cabi
from ..marcos import BLANK_SYMBOL, IGNORE_ID


def levenshtein(a, b):
    """edit distance of two sequences of hashables (characters, words, ids) through the C ABI"""
    ids = {}
    ia = [ids.setdefault(x, len(ids)) for x in a]
    ib = [ids.setdefault(x, len(ids)) for x in b]
    aa, bb = (C.c_int32 * max(len(ia), 1))(*ia), (C.c_int32 * max(len(ib), 1))(*ib)
    d = _cabi.lib().masr_edit_distance(aa, len(ia), bb, len(ib))
    if d < 0:
        raise _cabi.MasrError("masr_edit_distance: " + _cabi.lib().masr_last_error().decode())
    return int(d)


class Metric:
    def __init__(self, model_path, id2units, sos_id, eos_id, ignore_id=None):
        import sentencepiece as spmlib
        self.spm = spmlib.SentencePieceProcessor()
        self.spm.Load(str(model_path))
        self.id2units, self.sos_id, self.eos_id, self.ignore_id = id2units, sos_id, eos_id, ignore_id
        self.blank_id = id2units.index(BLANK_SYMBOL) if BLANK_SYMBOL in id2units else None

    def discard_ch_after_eos(self, ls):
        """metric.py:24-33 (note: a leading eos at position 0 is not a stop; an eos-free list becomes empty)."""
        if len(ls) == 1:
            return []
        stop = 0
        for pos in range(1, len(ls)):
            if ls[pos] == self.eos_id:
                stop = pos
                break
        return ls[:stop]

    def _texts(self, pred, y):
        hyp = [x for x in self.discard_ch_after_eos(pred.tolist()) if x != self.sos_id]
        hyp_text = self.spm.DecodePieces([self.id2units[x] for x in hyp])
        ref = [self.id2units[x] for x in y.tolist() if x != self.eos_id and x != IGNORE_ID]
        return hyp_text, self.spm.DecodePieces(ref)

    def cal_att_wer(self, pred, y, show=False, show_decode=False):
        h, r = self._texts(pred, y)
        h, r = h.split(' '), r.split(' ')
        return float(levenshtein(h, r)) / len(r) * 100

    def cal_att_cer(self, pred, y, show=False, show_decode=False):
        h, r = self._texts(pred, y)
        return float(levenshtein(h, r)) / len(r) * 100

    def _ctc_texts(self, pred, y):
        """metric.py:89-135: collapse repeats, drop sos/eos/blank, DecodePieces"""
        from itertools import groupby
        assert self.blank_id is not None
        hyp = [x[0] for x in groupby(pred.tolist())]
        hyp = [x for x in hyp if x != self.sos_id and x != self.eos_id and x != self.blank_id]
        return self.spm.DecodePieces([self.id2units[x] for x in hyp]), self.spm.DecodePieces([self.id2units[x] for x in y.tolist()])

    def cal_ctc_wer(self, pred, y, show=False, show_decode=False):
        h, r = self._ctc_texts(pred, y)
        h, r = h.split(' '), r.split(' ')
        return float(levenshtein(h, r)) / len(r) * 100

    def cal_ctc_cer(self, pred, y, show=False, show_decode=False):
        h, r = self._ctc_texts(pred, y)
        return float(levenshtein(h, r)) / len(r) * 100

    def batch_cal_er(self, preds, ys, modes, er_modes):
        pred = torch.argmax(preds, dim=-1)
        out = {}
        for mode in modes:
            for er in er_modes:
                fn = getattr(self, f"cal_{mode}_{er}")
                out[f"{mode}_{er}"] = sum(fn(h, y) for h, y in zip(pred, ys)) / pred.size(0)
        return out
