"""Eval metric path (reference: src/monitor/metric.py:24-87): teacher-forced argmax -> trim at the first
</s> -> sentencepiece DecodePieces -> Levenshtein / reference length * 100.  `editdistance` (C extension the
reference imports) is replaced by the small DP below."""
import torch

from ..marcos import BLANK_SYMBOL, IGNORE_ID


def levenshtein(a, b):
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


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

    def batch_cal_er(self, preds, ys, modes, er_modes):
        pred = torch.argmax(preds, dim=-1)
        out = {}
        for mode in modes:
            for er in er_modes:
                fn = getattr(self, f"cal_{mode}_{er}")
                out[f"{mode}_{er}"] = sum(fn(h, y) for h, y in zip(pred, ys)) / pred.size(0)
        return out
