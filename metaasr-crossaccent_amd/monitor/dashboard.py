"""Dashboard: the reference logs to comet.ml / tensorboard (src/monitor/dashboard.py, out of scope).
This one appends JSON lines to <log_dir>/dashboard.jsonl and keeps the call surface the interfaces use."""
import json
import time


class Dashboard:
    def __init__(self, config, paras, log_dir, train_type, resume):
        self.path = log_dir.joinpath("dashboard.jsonl")
        self.global_step = 1
        with open(log_dir.joinpath("exp_key"), "w") as f:
            print("local", file=f)

    def _w(self, **kw):
        kw["t"] = time.time()
        kw["step"] = self.global_step
        with open(self.path, "a") as f:
            f.write(json.dumps(kw) + "\n")

    def set_status(self, status): self._w(status=status)
    def set_step(self, step): self.global_step = step
    def step(self, n=1): self.global_step += n
    def log_step(self): self._w(event="snapshot")
    def log_info(self, prefix, info): self._w(prefix=prefix, **{k: float(v) for k, v in info.items()})
    def log_other(self, name, value): self._w(**{name: value})
    def check(self): pass
    def add_tag(self, tag): self._w(tag=tag)
    def log_config(self, cfg): pass
