"""RunningAvgDict: stand-in for the un-vendored `torchexp.stat.RunningAvgDict` the reference imports
(src/pretrain_interface.py:12).  Its exact semantics are unknown (third party, absent from the reference
tree): decay 1.0 = sample-weighted mean, otherwise an exponential moving average.  Smoothed log values are
therefore "parity unpinned"; raw per-batch infos are what the goldens pin."""


class RunningAvgDict(dict):
    def __init__(self, decay_rate=1.0):
        super().__init__()
        self.decay_rate = decay_rate
        self._n = {}

    def add(self, info, n=1):
        for k, v in info.items():
            v = float(v)
            if k not in self:
                self[k], self._n[k] = v, n
            elif self.decay_rate >= 1.0:
                tot = self._n[k] + n
                self[k] = (self[k] * self._n[k] + v * n) / tot
                self._n[k] = tot
            else:
                self[k] = self.decay_rate * self[k] + (1 - self.decay_rate) * v
