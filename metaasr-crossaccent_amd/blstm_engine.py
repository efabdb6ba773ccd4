"""BlstmEngine: flat HBM buffers + ctypes driver of libmasr's BLSTM-CTC calls (masr_blstm_*), the second model family of
the reference (config/blstm: MonoBLSTM, src/model/blstm/mono_blstm.py; loss of BLSTMTrainer.run_batch,
src/blstm_trainer.py:55-85).  Same conventions as engine.MasrEngine; PyTorch is device memory and stream plumbing only."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _cabi
from ._cabi import check, lib

MASR_TRAIN, MASR_EVAL = 1, 0


class BlstmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("idim", "odim", "enc_dim", "proj_dim", "enc_odim", "nlayers")] + [("sample_rate", C.c_int32 * 8)]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class BlstmEngine:
    def __init__(self, model_para: dict, odim: int, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("BlstmEngine needs a HIP device (MI355X); there is no CPU path")
        e = model_para["encoder"]
        rates = [int(v) for v in e["sample_rate"].split("_")]
        drops = [float(v) for v in e["dropout"].split("_")]
        # `dropout` goes to nn.LSTM(dropout=..., num_layers=1) in the reference (src/modules/encoder.py:86-90), where torch applies
        # it only BETWEEN stacked layers: with one layer per module it is a no-op (torch warns), so any value gives the same
        # results and is accepted here.  sample_rate[i] > 1: layer i's output keeps every sample_rate[i]-th frame (encoder.py:118-121).
        if len(drops) != len(rates):
            raise ValueError("encoder.sample_rate and encoder.dropout must list one value per BLSTM layer")
        if not 1 <= len(rates) <= 8 or any(not 1 <= r <= 8 for r in rates):
            raise ValueError("BLSTM encoder: 1 .. 8 layers, sample_rate 1 .. 8 per layer")
        self.device = torch.device(device)
        self.odim = odim
        self.cfg = BlstmConfig(idim=e["idim"], odim=odim, enc_dim=e["enc_dim"], proj_dim=e["proj_dim"], enc_odim=e["odim"], nlayers=len(rates),
                               sample_rate=(C.c_int32 * 8)(*(rates + [0] * (8 - len(rates)))))
        self._l = lib()
        self.h = self._l.masr_blstm_create(C.byref(self.cfg))
        if not self.h:
            raise _cabi.MasrError("masr_blstm_create: " + self._l.masr_last_error().decode())
        self.numel = int(self._l.masr_blstm_param_numel(self.h))
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.table = OrderedDict()
        name = C.create_string_buffer(256); shape = (C.c_int64 * 4)(); ndim, off = C.c_int(), C.c_int64()
        for i in range(self._l.masr_blstm_param_count(self.h)):
            check(self._l.masr_blstm_param_info(self.h, i, name, 256, shape, C.byref(ndim), C.byref(off)), "masr_blstm_param_info")
            self.table[name.value.decode()] = (int(off.value), tuple(int(shape[k]) for k in range(ndim.value)))
        self.ws = None
        self._dirty = True
        self._ensure_ws(1, 16, 4)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self._l.masr_blstm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ensure_ws(self, B, T, maxL):
        need = int(self._l.masr_blstm_workspace_bytes(self.h, B, T, maxL))
        if self.ws is None or self.ws.numel() < need:
            torch.cuda.synchronize(self.device)
            self.ws = None
            self.ws = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            check(self._l.masr_blstm_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.ws), self.ws.numel()), "masr_blstm_bind")
            self._dirty = True

    def view(self, name, flat=None):
        off, shape = self.table[name]
        flat = self.params if flat is None else flat
        return flat[off:off + int(np.prod(shape))].view(shape)

    def state_dict(self, flat=None, clone=True):
        return OrderedDict((n, self.view(n, flat).clone() if clone else self.view(n, flat)) for n in self.table)

    def load_state_dict(self, sd, flat=None):
        dst = self.params if flat is None else flat
        for n, (off, shape) in self.table.items():
            dst[off:off + sd[n].numel()].copy_(sd[n].detach().reshape(-1).to(torch.float32))
        if flat is None:
            self._dirty = True

    def mark_dirty(self):
        self._dirty = True

    def refresh(self):
        if self._dirty:
            check(self._l.masr_blstm_refresh(self.h, self.stream()), "masr_blstm_refresh")
            self._dirty = False

    def run_batch(self, xs, ilens, ys, olens, train: bool):
        """forward + CTC loss (+ backward).  ys: list of int64 label tensors WITHOUT sos/eos; olens: their lengths."""
        ready = getattr(xs, "_masr_ready", None)               # uploaded ahead on the loader's copy stream (io/dataset.py Loader._materialize_ahead)
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        if xs.device != self.device:
            xs = xs.to(self.device, non_blocking=True)
        elif xs.is_cuda:
            xs.record_stream(torch.cuda.current_stream(self.device))      # (made on another stream: see MasrEngine.run_batch)
        xs = xs.contiguous().float()
        B, T, D = xs.shape
        assert D == self.cfg.idim
        il = torch.as_tensor(ilens, dtype=torch.int64).cpu().contiguous()
        ol = torch.as_tensor(olens, dtype=torch.int64).cpu().contiguous()
        yf = torch.cat([torch.as_tensor(y, dtype=torch.int64).reshape(-1) for y in ys]).cpu().contiguous()
        self._ensure_ws(B, T, int(ol.max()))
        self.refresh()
        check(self._l.masr_blstm_run_batch(self.h, _ptr(xs), C.c_void_p(il.data_ptr()), C.c_void_p(yf.data_ptr()), C.c_void_p(ol.data_ptr()),
                                           B, T, MASR_TRAIN if train else MASR_EVAL, self.stream()), "masr_blstm_run_batch")
        self._last_x = xs

    def forward(self, xs, ilens):
        """MonoBLSTM.forward: -> (logits fp32 [B, T', odim], enc_lens int32 [B]) (views into the workspace)"""
        if xs.device != self.device:
            xs = xs.to(self.device, non_blocking=True)
        xs = xs.contiguous().float()
        B, T, D = xs.shape
        il = torch.as_tensor(ilens, dtype=torch.int64).cpu().contiguous()
        self._ensure_ws(B, T, 4)
        self.refresh()
        check(self._l.masr_blstm_forward(self.h, _ptr(xs), C.c_void_p(il.data_ptr()), B, T, self.stream()), "masr_blstm_forward")
        self._last_x = xs
        # the forward-only path reads no stats block: a timed-out resident recurrence must not hand out garbage logits silently
        # (its callers copy the logits to the host next, so this stream sync costs them nothing)
        check(self._l.masr_blstm_check(self.h, self.stream()), "masr_blstm_check")
        return self.last_logits()

    def set_resident_recurrence(self, on: bool):
        """the LSTM recurrence as one launch per layer and pass (include/masr.h masr_blstm_set_resident_recurrence); default on"""
        self._l.masr_blstm_set_resident_recurrence(self.h, int(bool(on)))

    def read_stats(self):
        out = (C.c_float * 4)()
        check(self._l.masr_blstm_read_stats(self.h, out, self.stream()), "masr_blstm_read_stats")
        return {"loss": float(out[0]), "grad_norm": float(out[3])}

    def last_logits(self):
        """(logits fp32 [B, T', odim] view into the workspace, enc_lens int32 [B])"""
        lp, ep = C.c_void_p(), C.c_void_p()
        B, Tp, Cc = C.c_int(), C.c_int(), C.c_int()
        check(self._l.masr_blstm_last_logits(self.h, C.byref(lp), C.byref(ep), C.byref(B), C.byref(Tp), C.byref(Cc)), "masr_blstm_last_logits")
        base = self.ws.data_ptr()
        lo, eo = lp.value - base, ep.value - base
        logits = self.ws[lo:lo + B.value * Tp.value * Cc.value * 4].view(torch.float32).view(B.value, Tp.value, Cc.value)
        lens = self.ws[eo:eo + B.value * 4].view(torch.int32)
        return logits, lens

    def clip_grads(self, max_norm):
        check(self._l.masr_blstm_clip_grads(self.h, max_norm, self.stream()), "masr_blstm_clip_grads")

    def sgd_step(self, params, grads, mom, lr, momentum, nesterov, first_step):
        check(self._l.masr_sgd_step(_ptr(params), _ptr(grads), _ptr(mom), params.numel(), lr, momentum, int(nesterov), int(first_step), self.stream()),
              "masr_sgd_step")

    def clip_sgd_step(self, momentum_buf, max_norm, lr, momentum, nesterov, first_step):
        check(self._l.masr_blstm_clip_sgd_step(self.h, _ptr(momentum_buf), max_norm, lr, momentum, int(nesterov), int(first_step), self.stream()),
              "masr_blstm_clip_sgd_step")
        self._dirty = False


def reference_init_state_dict(model_para: dict, odim: int):
    """Initial weights identical to the reference's for the same torch seed: replays MonoBLSTM.__init__'s RNG consumption
    (module construction order of src/modules/encoder.py:230-248 and mono_blstm.py:36-38) with throw-away torch.nn modules
    on the host, then lecun_normal_init_parameters (src/nets_utils.py:293-318: biases 0, matrices N(0, 1/sqrt(fan_in)))
    over encoder.parameters() and head.parameters()."""
    import math
    from torch import nn
    e = model_para["encoder"]
    H, proj, eodim = e["enc_dim"], e["proj_dim"], e["odim"]
    rates = e["sample_rate"].split("_")
    vgg = nn.Sequential(nn.Conv2d(1, 128, 3, 1, 1), nn.ReLU(), nn.Conv2d(128, 128, 3, 1, 1), nn.ReLU(), nn.MaxPool2d(2, 2, ceil_mode=True),
                        nn.Conv2d(128, 256, 3, 1, 1), nn.ReLU(), nn.Conv2d(256, 256, 3, 1, 1), nn.ReLU(), nn.MaxPool2d(2, 2, ceil_mode=True))
    vgg_o = int(np.ceil(np.ceil(np.array(e["idim"], dtype=np.float32) / 2) / 2)) * 256
    mods = OrderedDict()
    mods["encoder.vgg"] = vgg
    nl = len(rates)
    order = []
    rnn0 = nn.LSTM(vgg_o, H, num_layers=1, bidirectional=True, batch_first=True); bt0 = nn.Linear(2 * H, proj)
    order += [("encoder.blstm.rnn0", rnn0), ("encoder.blstm.bt0", bt0)]
    for i in range(1, nl):
        rnn = nn.LSTM(proj, H, num_layers=1, bidirectional=True, batch_first=True)
        bt = nn.Linear(2 * H, eodim if i == nl - 1 else proj)
        order += [(f"encoder.blstm.rnn{i}", rnn), (f"encoder.blstm.bt{i}", bt)]
    head = nn.Linear(eodim, odim)
    allmods = [("encoder.vgg", vgg)] + order + [("head", head)]
    with torch.no_grad():
        for _, mod in allmods:                       # init_encoder() walks encoder.parameters() (vgg, then blstm), then the head
            for prm in mod.parameters():
                if prm.dim() == 1:
                    prm.zero_()
                else:
                    n = prm.size(1)
                    for k in prm.size()[2:]:
                        n *= k
                    prm.normal_(0, 1.0 / math.sqrt(n))
    sd = OrderedDict()
    for name, mod in allmods:
        for n, t in mod.state_dict().items():
            sd[f"{name}.{n}"] = t.detach()
    return sd


class MonoBLSTM:
    """Facade with the surface the Interface / Trainer code touches (reference: src/model/blstm/mono_blstm.py:23-92)."""

    def __init__(self, id2char, model_para, device="cuda:0", init=True):
        from .marcos import BLANK_SYMBOL
        self.idim = model_para["encoder"]["idim"]
        self.odim = len(id2char)
        self.sos_id = self.eos_id = len(id2char) - 1
        self.blank_id = id2char.index(BLANK_SYMBOL)
        self.engine = BlstmEngine(model_para, self.odim, device=device)
        self.training = True
        if init:
            self.engine.load_state_dict(reference_init_state_dict(model_para, self.odim))

    def cuda(self):
        return self

    def __call__(self, xs_pad, ilens):
        """forward(xs_pad, ilens) -> (out [B, T', odim], enc_lens)"""
        logits, lens = self.engine.forward(xs_pad, ilens)
        return logits, lens.to(torch.int64)

    greedy_decode = __call__                                  # mono_blstm.py:63-64

    def train(self):
        self.training = True

    def eval(self):
        self.training = False

    def state_dict(self):
        return self.engine.state_dict()

    def load_state_dict(self, sd):
        self.engine.load_state_dict(sd)

    @property
    def device(self):
        return self.engine.device
