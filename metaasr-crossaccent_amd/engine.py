"""MasrEngine: owner of the flat HBM buffers and thin driver of the libmasr C ABI.

PyTorch is used only as plumbing here: device memory (`torch.empty`), the current HIP stream and
(in parallel.py) `torch.distributed`.  All arithmetic of the hot path runs in libmasr's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import time
from collections import OrderedDict

import numpy as np
import torch

from . import _cabi
from ._cabi import MasrConfig, check, lib

MASR_TRAIN, MASR_EVAL = 1, 0


def sinusoid_pe(max_len: int, E: int) -> torch.Tensor:
    """PositionalEncoding buffer, formula of mono_transformer_torch.py:21-28 -> [max_len, 1, E]."""
    pe = torch.zeros(max_len, E)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, E, 2).float() * (-math.log(10000.0) / E))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(1)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class PendingStats:
    """stats block of one batch on its way to the host (MasrEngine.read_stats_async): a ticket of the engine's ring of page-locked
    blocks (include/masr.h masr_stats_post).  The waiting thread first polls the block's words -- they hold MASR_STATS_PENDING until
    the copy lands -- WITHOUT entering the HIP runtime (the task threads are inside its launch path at that moment and the runtime
    serialises callers), then confirms through the event recorded behind the copy (masr_stats_wait: completion + host visibility).
    The block belongs to the engine and is only handed out again once that event has completed, so a dropped handle is harmless."""
    SENTINEL = 0x7FC0DEAD

    def __init__(self, engine, ticket, words):
        self.engine, self.ticket, self.words = engine, ticket, words
        self._result = None

    def ready(self):
        w = self.words
        return not (w[0] == self.SENTINEL or w[1] == self.SENTINEL or w[2] == self.SENTINEL or w[3] == self.SENTINEL)

    def get(self):
        if self._result is not None:                              # (already brought in, e.g. by the engine before its ring wrapped)
            return self._result
        n, t0 = 0, None
        while not self.ready():
            n += 1
            time.sleep(0 if n < 50 else 5e-5)                 # (releases the interpreter to the task threads either way)
            if n % 4096 == 0:
                t0 = t0 or time.monotonic()
                if time.monotonic() - t0 > 300.0:
                    raise RuntimeError("the stats of a queued batch never reached the host (stream wedged, or the batch failed to launch)")
        out = (C.c_float * 4)()
        check(self.engine._l.masr_stats_wait(self.engine.h, self.ticket, out), "masr_stats_wait")
        self._result = {"loss": float(out[0]), "n_correct": float(out[1]), "n_total": float(out[2]), "grad_norm": float(out[3])}
        return self._result


class MasrEngine:
    """One model instance on one GPU: flat fp32 params / grads + activation workspace."""

    def __init__(self, model_para: dict, odim: int, label_smoothing: float = 0.0, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("MasrEngine needs a HIP device (MI355X); there is no CPU path")
        self.device = torch.device(device)
        self.model_para = model_para
        self.odim = odim
        self.cfg = MasrConfig(
            idim=model_para["idim"], odim=odim, d_model=model_para["d_model"], nheads=model_para["nheads"],
            d_inner=model_para["d_inner"], enc_layers=model_para["encoder"]["nlayers"],
            dec_layers=model_para["decoder"]["nlayers"], tie_weights=int(model_para.get("tgt_share_weight", 0) != 0),
            dropout=float(model_para.get("dropout", 0.0)), pos_dropout=float(model_para.get("pos_dropout", 0.0)),
            label_smoothing=float(label_smoothing))
        self._l = lib()
        self.h = self._l.masr_create(C.byref(self.cfg))
        if not self.h:
            raise _cabi.MasrError("masr_create: " + self._l.masr_last_error().decode())
        self.numel = int(self._l.masr_param_numel(self.h))
        self.tied = bool(self.cfg.tie_weights)
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
            self.grads = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
            self.pe = sinusoid_pe(3000, self.cfg.d_model).to(self.device).contiguous()
        self.table = OrderedDict()           # name -> (offset, shape)
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        ndim, off = C.c_int(), C.c_int64()
        for i in range(self._l.masr_param_count(self.h)):
            check(self._l.masr_param_info(self.h, i, name, 256, shape, C.byref(ndim), C.byref(off)), "masr_param_info")
            self.table[name.value.decode()] = (int(off.value), tuple(int(shape[k]) for k in range(ndim.value)))
        self.ws = None
        self._ws_key = (0, 0, 0)
        self._ensure_ws(1, 64, 8)
        self._dirty = True

    # ------------------------------------------------------------------ buffers
    def __del__(self):
        try:
            if getattr(self, "h", None):
                self._l.masr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ensure_ws(self, B, T, L):
        need = int(self._l.masr_workspace_bytes(self.h, B, T, L))
        if self.ws is None or self.ws.numel() < need:
            torch.cuda.synchronize(self.device)
            self.ws = None
            self.ws = torch.empty(int(need * 1.05) + 4096, dtype=torch.uint8, device=self.device)
            check(self._l.masr_bind(self.h, _ptr(self.params), _ptr(self.grads), _ptr(self.pe), _ptr(self.ws), self.ws.numel()), "masr_bind")
            self._dirty = True

    def view(self, name, flat=None):
        off, shape = self.table[name]
        flat = self.params if flat is None else flat
        return flat[off:off + int(np.prod(shape))].view(shape)

    def state_dict(self, flat=None, clone=True) -> "OrderedDict[str, torch.Tensor]":
        """Reference key set and order, incl. pos_encoder.pe and the tied alias."""
        sd = OrderedDict()
        for n in self.table:
            v = self.view(n, flat)
            sd[n] = v.clone() if clone else v
            if n == "vgg2enc.bias":
                sd["pos_encoder.pe"] = self.pe.clone() if clone else self.pe
            if n == "char_trans.bias" and self.tied:
                sd["pre_embed.weight"] = sd["char_trans.weight"]
        return sd

    def load_state_dict(self, sd, flat=None):
        dst = self.params if flat is None else flat
        for n, (off, shape) in self.table.items():
            t = sd[n]
            dst[off:off + t.numel()].copy_(t.detach().reshape(-1).to(torch.float32))
        if flat is None:
            self._dirty = True

    def mark_dirty(self):
        """call after writing self.params outside the C ABI (copy_, all-reduce, ...)"""
        self._dirty = True

    def refresh(self):
        if self._dirty:
            check(self._l.masr_refresh(self.h, self.stream()), "masr_refresh")
            self._dirty = False

    def set_seed(self, seed: int):
        self._l.masr_set_seed(self.h, C.c_uint64(seed & (2 ** 64 - 1)))

    def set_concurrency(self, slots: int):
        """this engine is one of `slots` task slots sharing the GPU (include/masr.h masr_set_concurrency)"""
        self._l.masr_set_concurrency(self.h, int(slots))

    def dropout_state(self):
        """(seed, batches run since set_seed): the position of the dropout mask stream (include/masr.h masr_dropout_state)"""
        st = (C.c_uint64 * 2)()
        self._l.masr_dropout_state(self.h, st, 0)
        return int(st[0]), int(st[1])

    def set_dropout_state(self, state):
        st = (C.c_uint64 * 2)(int(state[0]), int(state[1]))
        self._l.masr_dropout_state(self.h, st, 1)

    # ------------------------------------------------------------------ the operator
    def run_batch(self, xs: torch.Tensor, ilens, ys, olens, train: bool):
        """forward + loss (+ backward).  xs: device fp32 [B,T,idim] (host tensors are uploaded);
        ilens/olens: int64 host tensors; ys: list of int64 host tensors."""
        ready = getattr(xs, "_masr_ready", None)               # uploaded ahead on the loader's copy stream (io/dataset.py Loader._materialize_ahead)
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        if xs.device != self.device:
            xs = xs.to(self.device, non_blocking=True)
        elif xs.is_cuda:
            # made on another stream (HBM-resident shards are gathered on the main stream a meta-step ahead): tell the allocator it
            # is read on this one, or the block could be handed out again while this stream's backward still needs it
            xs.record_stream(torch.cuda.current_stream(self.device))
        xs = xs.contiguous().float()
        B, T, D = xs.shape
        assert D == self.cfg.idim, f"idim mismatch {D} vs {self.cfg.idim}"
        il = torch.as_tensor(ilens, dtype=torch.int64).cpu().contiguous()
        ol = torch.as_tensor(olens, dtype=torch.int64).cpu().contiguous()
        yf = torch.cat([torch.as_tensor(y, dtype=torch.int64).reshape(-1) for y in ys]).cpu().contiguous()
        L = int(ol.max()) + 1
        self._ensure_ws(B, T, L)
        self.refresh()
        check(self._l.masr_run_batch(self.h, _ptr(xs), C.c_void_p(il.data_ptr()), C.c_void_p(yf.data_ptr()),
                                     C.c_void_p(ol.data_ptr()), B, T, MASR_TRAIN if train else MASR_EVAL, self.stream()),
              "masr_run_batch")
        self._last_x = xs          # keep the input alive until the stream has consumed it

    def recog(self, xs: torch.Tensor, ilens, full: bool = False):
        """greedy decode (MyTransformer.recog): returns int64 [Ldec, B] on the device, Ldec = max(ilens // 4).
        Default = KV-cached incremental decode; full=True = the reference's literal whole-prefix re-decode per step."""
        if xs.device != self.device:
            xs = xs.to(self.device, non_blocking=True)
        xs = xs.contiguous().float()
        B, T, D = xs.shape
        il = torch.as_tensor(ilens, dtype=torch.int64).cpu().contiguous()
        Ldec = int(il.max()) // 4
        self._ensure_ws(B, T, Ldec)
        self.refresh()
        out = torch.zeros(Ldec, B, dtype=torch.int32, device=self.device)
        fn = self._l.masr_recog_full if full else self._l.masr_recog
        check(fn(self.h, _ptr(xs), C.c_void_p(il.data_ptr()), B, T, _ptr(out), self.stream()), "masr_recog")
        self._last_x = xs
        return out.to(torch.int64)

    def read_stats(self):
        out = (C.c_float * 4)()
        check(self._l.masr_read_stats(self.h, out, self.stream()), "masr_read_stats")
        self._last_stats = {"loss": float(out[0]), "n_correct": float(out[1]), "n_total": float(out[2]), "grad_norm": float(out[3])}
        return self._last_stats

    def read_stats_async(self):
        """the same block, copied into page-locked memory by the stream WITHOUT waiting for it (include/masr.h masr_stats_post):
        returns a handle whose .get() waits for that copy only.  The host can then queue the next tasks while these run; the stats
        are what the log lines need one meta-step later."""
        ticket = int(self._l.masr_stats_post(self.h, self.stream()))
        if ticket < 0:
            raise _cabi.MasrError("masr_stats_post: " + self._l.masr_last_error().decode())
        h = PendingStats(self, ticket, self._l.masr_stats_peek(self.h, ticket))
        # a ticket expires after 64 newer posts (the ring of include/masr.h).  A loop that keeps a whole meta-step of handles per
        # engine (FOMAML, many tasks on one slot) could get there: handles still unread when the ring is 3/4 around are read NOW
        # (their copies are dozens of batches old) and keep the numbers, so the limit never reaches a caller
        out = self.__dict__.setdefault("_outstanding", [])
        out.append(h)
        while len(out) > 48:
            out.pop(0).get()
        return h

    def set_step_graphs(self, on: bool):
        """opt-in graph replay of repeated batch shapes (include/masr.h masr_set_step_graphs)"""
        self._l.masr_set_step_graphs(self.h, int(bool(on)))

    def set_split_wgrad_launches(self, on: bool):
        """the step's Linear weight gradients as two launches instead of one (include/masr.h masr_set_split_wgrad_launches; A/B, same bits)"""
        self._l.masr_set_split_wgrad_launches(self.h, int(bool(on)))

    def set_ksplit(self, on: bool):
        """k-split of the decoder's long-reduction few-row GEMMs with the combine inside the next LayerNorm (include/masr.h masr_set_ksplit).
        Default OFF: it changes the fp32 summation order of those GEMMs, so it follows only this call, never the slot count.  The one-task-per-stream
        loops (train.py: mono / multi interface) turn it on (+3 %); the FOMAML interface leaves it off for every --tasks_per_gpu."""
        self._l.masr_set_ksplit(self.h, int(bool(on)))

    def set_drop_nan_grads(self, on: bool):
        """clip_grads / clip_accumulate turn a gradient whose norm is NaN into zeros (include/masr.h masr_set_drop_nan_grads; pretrain.py
        --fix_nan_meta_grad).  Default off: the reference accumulates the NaNs (fo_meta_interface.py:151-154)."""
        self._l.masr_set_drop_nan_grads(self.h, int(bool(on)))

    def step_counters(self):
        """{'direct', 'captured', 'replayed'}: how run_batch calls reached the GPU (kernel by kernel / graph capture / graph replay);
        'ksplit_gemms': k-split GEMM launches of the last step launched or captured (0 = whole reductions: masr_set_ksplit off)"""
        out = (C.c_int64 * 4)()
        self._l.masr_step_counters(self.h, out)
        return {"direct": int(out[0]), "captured": int(out[1]), "replayed": int(out[2]), "ksplit_gemms": int(out[3])}

    def last_logits(self):
        """[B, L, odim] fp32 view of the last forward's logits and gold [B, L] (int32, -1 = pad)."""
        lp, gp = C.c_void_p(), C.c_void_p()
        rows, L, ld = C.c_int(), C.c_int(), C.c_int()
        check(self._l.masr_last_logits(self.h, C.byref(lp), C.byref(gp), C.byref(rows), C.byref(L), C.byref(ld)), "masr_last_logits")
        ws_base = self.ws.data_ptr()
        lo = (lp.value - ws_base)
        logits = self.ws[lo:lo + rows.value * ld.value * 4].view(torch.float32).view(rows.value // L.value, L.value, ld.value)[..., :self.odim]
        go = (gp.value - ws_base)
        gold = self.ws[go:go + rows.value * 4].view(torch.int32).view(rows.value // L.value, L.value)
        return logits, gold

    # ------------------------------------------------------------------ optimiser passes
    def clip_sgd_step(self, momentum_buf, max_norm, lr, momentum, nesterov, first_step):
        check(self._l.masr_clip_sgd_step(self.h, _ptr(momentum_buf), max_norm, lr, momentum, int(nesterov), int(first_step), self.stream()),
              "masr_clip_sgd_step")
        self._dirty = False            # the C call refreshes the shadows itself

    def clip_grads(self, max_norm):
        check(self._l.masr_clip_grads(self.h, max_norm, self.stream()), "masr_clip_grads")

    def clip_accumulate(self, updates, max_norm):
        check(self._l.masr_clip_accumulate(self.h, _ptr(updates), max_norm, self.stream()), "masr_clip_accumulate")

    def grad_norm(self):
        check(self._l.masr_grad_norm(self.h, self.stream()), "masr_grad_norm")

    def grad_norm_device_ptr(self):
        """address of the device float that holds the gradient norm after grad_norm() / clip_grads() (include/masr.h masr_stats_device)"""
        return int(self._l.masr_stats_device(self.h)) + 3 * 4

    def adam_step(self, params, grads, m, v, lr, b1, b2, eps, step, weight_decay=0.0, decoupled=False):
        if weight_decay:
            check(self._l.masr_adamw_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, b1, b2, eps, weight_decay,
                                          int(decoupled), step, self.stream()), "masr_adamw_step")
        else:
            check(self._l.masr_adam_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, b1, b2, eps, step, self.stream()), "masr_adam_step")

    def adam_step_guarded(self, params, grads, m, v, lr_a, t_a, lr_b, t_b, b1, b2, eps, weight_decay, decoupled, slot):
        """Adam / AdamW step skipped on the device when the stats block's gradient norm is NaN (include/masr.h masr_adam_step_guarded)"""
        check(self._l.masr_adam_step_guarded(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr_a, t_a, lr_b, t_b, b1, b2, eps,
                                             weight_decay, int(decoupled), slot, self.stream()), "masr_adam_step_guarded")

    def adam_sum_step(self, params, grad_list, gscale, m, v, lr, b1, b2, eps, step):
        """Adam on (sum of grad_list, in order) * gscale in one pass (include/masr.h masr_adam_sum_step)"""
        arr = (C.c_void_p * len(grad_list))(*[g.data_ptr() for g in grad_list])
        check(self._l.masr_adam_sum_step(_ptr(params), arr, len(grad_list), gscale, _ptr(m), _ptr(v), params.numel(), lr, b1, b2, eps, step,
                                         self.stream()), "masr_adam_sum_step")

    def sum_n(self, out, grad_list, scale=1.0):
        """out = (sum of grad_list, in order) * scale in one pass (include/masr.h masr_sum_n)"""
        arr = (C.c_void_p * len(grad_list))(*[g.data_ptr() for g in grad_list])
        check(self._l.masr_sum_n(_ptr(out), arr, len(grad_list), scale, out.numel(), self.stream()), "masr_sum_n")

    def radam_step(self, params, grads, m, v, lr, b1, b2, eps, step, weight_decay=0.0, variant=1):
        """variant 1: torch_optimizer.RAdam's conventions (the reference's import), 0: torch.optim.RAdam's (include/masr.h)"""
        check(self._l.masr_radam_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, b1, b2, eps, weight_decay, step, int(variant),
                                      self.stream()), "masr_radam_step")

    def sgd_step(self, params, grads, mom, lr, momentum, nesterov, first_step):
        check(self._l.masr_sgd_step(_ptr(params), _ptr(grads), _ptr(mom), params.numel(), lr, momentum, int(nesterov), int(first_step), self.stream()), "masr_sgd_step")

    def scale(self, x, a):
        check(self._l.masr_scale(_ptr(x), x.numel(), a, self.stream()), "masr_scale")

    def axpy(self, y, x, a):
        check(self._l.masr_axpy(_ptr(y), _ptr(x), x.numel(), a, self.stream()), "masr_axpy")

    def copy(self, dst, src):
        check(self._l.masr_copy(_ptr(dst), _ptr(src), src.numel(), self.stream()), "masr_copy")

    # ------------------------------------------------------------------ profiling
    def profile(self, on: bool):
        check(self._l.masr_profile_enable(self.h, int(on)), "masr_profile_enable")

    def profile_read(self):
        ms = (C.c_float * len(_cabi.PROF_NAMES))()
        n = (C.c_int * len(_cabi.PROF_NAMES))()
        check(self._l.masr_profile_read(self.h, ms, n), "masr_profile_read")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_cabi.PROF_NAMES)}
