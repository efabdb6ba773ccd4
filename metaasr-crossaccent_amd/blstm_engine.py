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
    _fields_ = [(n, C.c_int32) for n in ("idim", "odim", "enc_dim", "proj_dim", "enc_odim", "nlayers")]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class BlstmEngine:
    def __init__(self, model_para: dict, odim: int, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("BlstmEngine needs a HIP device (MI355X); there is no CPU path")
        e = model_para["encoder"]
        rates = [int(v) for v in e["sample_rate"].split("_")]
        drops = [float(v) for v in e["dropout"].split("_")]
        if any(r != 1 for r in rates) or any(d != 0 for d in drops):
            raise NotImplementedError("BLSTM encoder: sample_rate 1 and dropout 0 per layer (the shipped config/blstm settings)")
        self.device = torch.device(device)
        self.odim = odim
        self.cfg = BlstmConfig(idim=e["idim"], odim=odim, enc_dim=e["enc_dim"], proj_dim=e["proj_dim"], enc_odim=e["odim"], nlayers=len(rates))
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
        if xs.device != self.device:
            xs = xs.to(self.device, non_blocking=True)
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

    def clip_sgd_step(self, momentum_buf, max_norm, lr, momentum, nesterov, first_step):
        check(self._l.masr_blstm_clip_sgd_step(self.h, _ptr(momentum_buf), max_norm, lr, momentum, int(nesterov), int(first_step), self.stream()),
              "masr_blstm_clip_sgd_step")
        self._dirty = False
