"""CPU restatement of the reference hot path (TEST INFRASTRUCTURE, see oracle/__init__.py).

Plain fp32 torch tensor ops on CPU (matmul / conv2d / softmax / layer_norm
primitives; no nn.Transformer*, no nn.MultiheadAttention).  Backward comes from
autograd over this explicit forward.  Every function cites the reference
file:line (relative to the reference checkout) it restates.

Layouts are the reference's: activations [T, B, E] inside the transformer,
parameters under the reference's state_dict names.
"""
from __future__ import annotations

import math
import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

IGNORE_ID = -1      # src/marcos.py:8
GRAD_CLIP = 5       # src/marcos.py:7


# --------------------------------------------------------------------------- #
# parameters
# --------------------------------------------------------------------------- #
def param_shapes(cfg: dict, odim: int) -> "OrderedDict[str, tuple]":
    """state_dict entries of MyTransformer in registration order.

    Restates src/model/transformer_pytorch/mono_transformer_torch.py:37-104 and
    SURVEY Appendix D.  `pos_encoder.pe` is a buffer (no grad).  When
    tgt_share_weight != 0, `char_trans.weight` and `pre_embed.weight` are the
    same tensor (:66-70).
    """
    E, Fi = cfg["d_model"], cfg["d_inner"]
    vgg_o = 128 * (cfg["idim"] // 4)                     # :61
    s = OrderedDict()
    for idx, (co, ci) in zip((0, 2, 5, 7), ((64, 1), (64, 64), (128, 64), (128, 128))):
        s[f"feat_extractor.{idx}.weight"] = (co, ci, 3, 3)   # :49-60
        s[f"feat_extractor.{idx}.bias"] = (co,)
    s["vgg2enc.weight"] = (E, vgg_o)
    s["vgg2enc.bias"] = (E,)
    s["pos_encoder.pe"] = (3000, 1, E)
    s["char_trans.weight"] = (odim, E)
    s["char_trans.bias"] = (odim,)
    s["pre_embed.weight"] = (odim, E)

    def attn(prefix):
        s[f"{prefix}.in_proj_weight"] = (3 * E, E)
        s[f"{prefix}.in_proj_bias"] = (3 * E,)
        s[f"{prefix}.out_proj.weight"] = (E, E)
        s[f"{prefix}.out_proj.bias"] = (E,)

    def ffn_norms(prefix, n_norm):
        s[f"{prefix}.linear1.weight"] = (Fi, E)
        s[f"{prefix}.linear1.bias"] = (Fi,)
        s[f"{prefix}.linear2.weight"] = (E, Fi)
        s[f"{prefix}.linear2.bias"] = (E,)
        for i in range(1, n_norm + 1):
            s[f"{prefix}.norm{i}.weight"] = (E,)
            s[f"{prefix}.norm{i}.bias"] = (E,)

    for l in range(cfg["encoder"]["nlayers"]):
        attn(f"encoder.layers.{l}.self_attn")
        ffn_norms(f"encoder.layers.{l}", 2)
    s["encoder.norm.weight"] = (E,)
    s["encoder.norm.bias"] = (E,)
    for l in range(cfg["decoder"]["nlayers"]):
        attn(f"decoder.layers.{l}.self_attn")
        attn(f"decoder.layers.{l}.multihead_attn")
        ffn_norms(f"decoder.layers.{l}", 3)
    s["decoder.norm.weight"] = (E,)
    s["decoder.norm.bias"] = (E,)
    return s


def sinusoid_pe(max_len: int, E: int) -> torch.Tensor:
    """PositionalEncoding buffer [max_len, 1, E] (mono_transformer_torch.py:21-28)."""
    pe = torch.zeros(max_len, E)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, E, 2).float() * (-math.log(10000.0) / E))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(1)


def deterministic_state_dict(cfg: dict, odim: int, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """A reproducible, torch-version-independent parameter set for goldens/tests.

    NOT the reference's init (that is RNG-order dependent); it
    is only a shared starting point that `make_goldens.py` loads into the real
    reference model with load_state_dict and that tests load into the HIP engine.
    Scale follows xavier-uniform for matrices; biases/LN get small non-trivial
    values so bias/LN gradients are exercised.
    """
    rng = np.random.RandomState(seed)
    sd = OrderedDict()
    tied = cfg.get("tgt_share_weight", 0) != 0
    for name, shape in param_shapes(cfg, odim).items():
        if name == "pos_encoder.pe":
            sd[name] = sinusoid_pe(3000, cfg["d_model"])
            continue
        if tied and name == "pre_embed.weight":
            sd[name] = sd["char_trans.weight"]
            continue
        if len(shape) > 1:
            rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[1] * rf, shape[0] * rf
            a = math.sqrt(6.0 / (fan_in + fan_out))
            v = rng.uniform(-a, a, size=shape)
        elif name.endswith("weight"):            # LayerNorm gamma
            v = 1.0 + 0.1 * rng.uniform(-1, 1, size=shape)
        else:
            v = 0.05 * rng.uniform(-1, 1, size=shape)
        sd[name] = torch.from_numpy(v.astype(np.float32))
    return sd


# --------------------------------------------------------------------------- #
# masks (src/nets_utils.py)
# --------------------------------------------------------------------------- #
def make_bool_pad_mask(lengths: torch.Tensor) -> torch.Tensor:
    """mask[b, t] = t >= len[b]  (src/nets_utils.py:85-94)."""
    maxlen = int(lengths.max())
    return torch.arange(maxlen).unsqueeze(0) >= lengths.view(-1, 1)


def generate_square_subsequent_mask(sz: int) -> torch.Tensor:
    """0 on/below the diagonal, -inf above (src/nets_utils.py:9-15)."""
    m = torch.full((sz, sz), float("-inf"))
    return torch.triu(m, diagonal=1)


# --------------------------------------------------------------------------- #
# optional bf16 emulation (checker aid, NOT reference behaviour)
# --------------------------------------------------------------------------- #
# The HIP engine feeds bf16 operands to the MFMA units (fp32 accumulate, fp32 residual stream /
# LayerNorm / softmax / loss).  With QUANT[0] = True the oracle rounds the same tensors to bf16
# (straight-through gradient), so that ReLU / max-pool decisions coincide and the comparison is
# tight enough (~1e-2 -> ~1e-3) to expose real kernel bugs.  Default False = the reference's fp32.
QUANT = [False]


def _q(x):
    if not QUANT[0]:
        return x
    return x + (x.detach().bfloat16().float() - x.detach())


class bf16_emulation:
    def __enter__(self):
        self.prev = QUANT[0]
        QUANT[0] = True
        return self

    def __exit__(self, *a):
        QUANT[0] = self.prev


# --------------------------------------------------------------------------- #
# model forward (mono_transformer_torch.py)
# --------------------------------------------------------------------------- #
def extract_feat(p, xs_pad, ilens):
    """VGG front-end + Linear (mono_transformer_torch.py:113-122; Appendix A.1-2)."""
    x = xs_pad.unsqueeze(1)                                        # [B,1,T,D]
    x = _q(F.relu(F.conv2d(x, p["feat_extractor.0.weight"], p["feat_extractor.0.bias"], padding=1)))
    x = _q(F.relu(F.conv2d(x, _q(p["feat_extractor.2.weight"]), p["feat_extractor.2.bias"], padding=1)))
    x = F.max_pool2d(x, 2, stride=2)                                # floor
    x = _q(F.relu(F.conv2d(x, _q(p["feat_extractor.5.weight"]), p["feat_extractor.5.bias"], padding=1)))
    x = _q(F.relu(F.conv2d(x, _q(p["feat_extractor.7.weight"]), p["feat_extractor.7.bias"], padding=1)))
    x = F.max_pool2d(x, 2, stride=2)                                # [B,128,T',D']
    enc_lens = torch.floor(ilens.to(torch.float32) / 4).to(torch.int64)   # :117
    B, C, Tp, Dp = x.shape
    x = x.transpose(1, 2).contiguous().view(B, Tp, C * Dp)          # feature = c*D'+d (:118-119)
    x = x @ _q(p["vgg2enc.weight"]).t() + p["vgg2enc.bias"]
    return x, enc_lens


def _mha(p, prefix, q_in, kv_in, nheads, attn_mask=None, key_padding_mask=None):
    """torch nn.MultiheadAttention math (packed in_proj; Appendix A.4).

    q_in [Tq,B,E], kv_in [Tk,B,E]; attn_mask additive float [Tq,Tk];
    key_padding_mask bool [B,Tk] (True = pad -> -inf).
    """
    Tq, B, E = q_in.shape
    Tk = kv_in.shape[0]
    hd = E // nheads
    W, b = _q(p[f"{prefix}.in_proj_weight"]), p[f"{prefix}.in_proj_bias"]
    q_in, kv_in = _q(q_in), _q(kv_in)
    q = _q(q_in @ W[:E].t() + b[:E])
    k = _q(kv_in @ W[E:2 * E].t() + b[E:2 * E])
    v = _q(kv_in @ W[2 * E:].t() + b[2 * E:])
    q = q.reshape(Tq, B, nheads, hd).permute(1, 2, 0, 3)            # [B,H,Tq,hd]
    k = k.reshape(Tk, B, nheads, hd).permute(1, 2, 0, 3)
    v = v.reshape(Tk, B, nheads, hd).permute(1, 2, 0, 3)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)                    # [B,H,Tq,Tk]
    if attn_mask is not None:
        s = s + attn_mask
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.view(B, 1, 1, Tk), float("-inf"))
    a = torch.softmax(s, dim=-1)
    if QUANT[0]:        # the engine normalises after P.V: o = (bf16(exp(s-m)) @ v) / l
        mx = s.detach().amax(dim=-1, keepdim=True)
        l = torch.exp(s - mx).sum(-1, keepdim=True)
        a = _q(a * l) / l
    o = _q((a @ v).permute(2, 0, 1, 3).reshape(Tq, B, E))
    return o @ _q(p[f"{prefix}.out_proj.weight"]).t() + p[f"{prefix}.out_proj.bias"]


def _ln(p, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), p[f"{prefix}.weight"], p[f"{prefix}.bias"], 1e-5)


def _ffn(p, prefix, x):
    h = _q(F.relu(_q(x) @ _q(p[f"{prefix}.linear1.weight"]).t() + p[f"{prefix}.linear1.bias"]))
    return h @ _q(p[f"{prefix}.linear2.weight"]).t() + p[f"{prefix}.linear2.bias"]


def encoder_forward(p, cfg, x, pad_mask):
    """post-norm nn.TransformerEncoder + final LayerNorm (:74-85; Appendix A.4)."""
    H = cfg["nheads"]
    for l in range(cfg["encoder"]["nlayers"]):
        pre = f"encoder.layers.{l}"
        x = _ln(p, f"{pre}.norm1", x + _mha(p, f"{pre}.self_attn", x, x, H, key_padding_mask=pad_mask))
        x = _ln(p, f"{pre}.norm2", x + _ffn(p, pre, x))
    return _ln(p, "encoder.norm", x)


def decoder_forward(p, cfg, y, memory, causal, mem_pad_mask):
    """post-norm nn.TransformerDecoder + final LayerNorm (:87-98; Appendix A.6).
    No key-padding mask on the targets (tgt_pad_mask is computed but never used, :198-203)."""
    H = cfg["nheads"]
    for l in range(cfg["decoder"]["nlayers"]):
        pre = f"decoder.layers.{l}"
        y = _ln(p, f"{pre}.norm1", y + _mha(p, f"{pre}.self_attn", y, y, H, attn_mask=causal))
        y = _ln(p, f"{pre}.norm2", y + _mha(p, f"{pre}.multihead_attn", y, memory, H,
                                             key_padding_mask=mem_pad_mask))
        y = _ln(p, f"{pre}.norm3", y + _ffn(p, pre, y))
    return _ln(p, "decoder.norm", y)


def preprocess(p, ys, olens, sos_id, eos_id):
    """mono_transformer_torch.py:124-141.  ys_in = [sos]+y padded with eos -> [L,B];
    ys_out = y+[eos] padded with -1 -> [B,L]; olens += 1 IN PLACE (quirk Q6)."""
    B = len(ys)
    L = max(int(y.numel()) for y in ys) + 1
    ys_in = torch.full((L, B), eos_id, dtype=torch.int64)
    ys_out = torch.full((B, L), IGNORE_ID, dtype=torch.int64)
    for b, y in enumerate(ys):
        n = int(y.numel())
        ys_in[0, b] = sos_id
        ys_in[1:n + 1, b] = y
        ys_out[b, :n] = y
        ys_out[b, n] = eos_id
    emb = p["pre_embed.weight"][ys_in]                               # no sqrt(E) scaling (:132)
    emb = emb + p["pos_encoder.pe"][:L]
    olens += 1
    return emb, ys_out, olens


def model_forward(p, cfg, xs_pad, ilens, ys, olens):
    """MyTransformer.forward (mono_transformer_torch.py:178-208), dropout = 0."""
    odim = p["char_trans.weight"].shape[0]
    sos_id, eos_id = 0, odim - 1                                     # :45-46
    enc, enc_lens = extract_feat(p, xs_pad, ilens)
    enc = enc.transpose(0, 1)                                        # [T',B,E]
    enc = enc + p["pos_encoder.pe"][:enc.shape[0]]
    pad_mask = make_bool_pad_mask(enc_lens)
    ys_in, ys_out, olens = preprocess(p, ys, olens, sos_id, eos_id)
    causal = generate_square_subsequent_mask(ys_in.shape[0])
    memory = encoder_forward(p, cfg, enc, pad_mask)
    out = decoder_forward(p, cfg, ys_in, memory, causal, pad_mask)
    out = out.transpose(0, 1)                                        # [B,L,E]
    logit = _q(out) @ _q(p["char_trans.weight"]).t() + p["char_trans.bias"]
    return logit, ys_out


def recog_greedy(p, cfg, xs_pad, ilens, margins=False):
    """MyTransformer.recog (mono_transformer_torch.py:143-176): encoder once, then
    max(enc_lens) full re-decodes, argmax of EVERY position each step -> [Ldec,B].
    margins=True: also the decision margin of every emitted token, [Ldec,B]: (best logit - second best) / std of that
    position's logits -- how well defined the reference's own arg-max is there (the causal mask makes the last re-decode's
    logits at position t the ones that decided token t)."""
    odim = p["char_trans.weight"].shape[0]
    B = xs_pad.shape[0]
    enc, enc_lens = extract_feat(p, xs_pad, ilens)
    enc = enc.transpose(0, 1)
    enc = enc + p["pos_encoder.pe"][:enc.shape[0]]
    pad_mask = make_bool_pad_mask(enc_lens)
    memory = encoder_forward(p, cfg, enc, pad_mask)
    sos = torch.zeros(1, B, dtype=torch.int64)
    out = torch.zeros(0, B, dtype=torch.int64)
    for _ in range(int(enc_lens.max())):
        tok = torch.cat([sos, out])
        y = p["pre_embed.weight"][tok] + p["pos_encoder.pe"][:tok.shape[0]]
        causal = generate_square_subsequent_mask(tok.shape[0])
        h = decoder_forward(p, cfg, y, memory, causal, pad_mask)
        z = h @ p["char_trans.weight"].t() + p["char_trans.bias"]
        out = torch.argmax(z, dim=-1)
    if margins:
        top = z.topk(2, dim=-1).values
        return out, (top[..., 0] - top[..., 1]) / z.std(dim=-1)
    return out


# --------------------------------------------------------------------------- #
# loss (src/transformer_torch_trainer.py:59-99)
# --------------------------------------------------------------------------- #
def label_smoothed_ce(logit, gold, eps):
    """run_batch loss: q = onehot*(1-eps) + (1-onehot)*eps/C (note /C), masked mean.
    eps == 0 -> F.cross_entropy(ignore_index=-1).  Returns (loss, n_correct, n_total)."""
    pred = logit.reshape(-1, logit.shape[-1])
    g = gold.reshape(-1)
    mask = g.ne(IGNORE_ID)
    n_total = int(mask.sum())
    logp = torch.log_softmax(pred, dim=-1)
    if eps > 0.0:
        C = pred.shape[1]
        gs = mask.long() * g
        one_hot = torch.zeros_like(pred).scatter(1, gs.view(-1, 1), 1.0)
        q = one_hot * (1 - eps) + (1 - one_hot) * eps / C
        loss = (-(q * logp).sum(dim=1))[mask].sum() / n_total
    else:
        loss = (-logp[mask, g[mask]]).sum() / n_total
    n_correct = int((pred.detach().argmax(1).eq(g) & mask).sum())
    return loss, n_correct, n_total


def grad_param_names(p, cfg):
    """Unique gradient-carrying tensors in nn.Module.parameters() order (tied weight once,
    under char_trans.weight which is registered first; pe excluded)."""
    tied = cfg.get("tgt_share_weight", 0) != 0
    return [n for n in p if n != "pos_encoder.pe" and not (tied and n == "pre_embed.weight")]


def run_batch_train(p, cfg, batch, eps):
    """forward + loss + backward (transformer_torch_trainer.py:59-93).  `p` values must be
    leaf tensors with requires_grad (tied names pointing at the same leaf).
    Returns info dict and {name: grad}."""
    xs_pad, ilens, ys, olens = batch
    names = grad_param_names(p, cfg)
    for n in names:
        p[n].grad = None
    logit, gold = model_forward(p, cfg, xs_pad, ilens, ys, olens)
    loss, n_correct, n_total = label_smoothed_ce(logit, gold, eps)
    loss.backward()
    grads = {n: p[n].grad for n in names}
    return {"loss": float(loss.detach()), "acc": float(n_correct) / n_total}, grads, logit.detach(), gold


# --------------------------------------------------------------------------- #
# optimizers
# --------------------------------------------------------------------------- #
def clip_grad_norm_(grads: dict, max_norm: float = GRAD_CLIP) -> float:
    """torch.nn.utils.clip_grad_norm_ (L2): coef = max/(norm+1e-6) clamped to 1
    (call sites src/fo_meta_interface.py:148-149,242-243)."""
    total = torch.sqrt(sum((g.detach().double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return float(total)


def sgd_nesterov_step(params: dict, grads: dict, bufs: dict, lr, momentum, nesterov):
    """torch.optim.SGD (dampening 0, wd 0): first step buf = g; then buf = m*buf + g;
    nesterov: g += m*buf; p -= lr*g  (call site fo_meta_interface.py:228-236,248)."""
    with torch.no_grad():
        for n, g in grads.items():
            g = g.clone()
            if momentum != 0:
                if n not in bufs:
                    bufs[n] = g.clone()
                else:
                    bufs[n].mul_(momentum).add_(g)
                g = g + momentum * bufs[n] if nesterov else bufs[n]
            params[n].sub_(lr * g)


def noam_lr(step_num, k, d_model, warmup):
    """TransformerOptimizer._update_lr (optimizer.py:23-28)."""
    return k * d_model ** (-0.5) * min(step_num ** (-0.5), step_num * warmup ** (-1.5))


def adam_step(params: dict, grads: dict, state: dict, lr, b1=0.9, b2=0.98, eps=1e-9):
    """torch.optim.Adam (no amsgrad / wd): denom = sqrt(v)/sqrt(1-b2^t) + eps;
    p -= lr/(1-b1^t) * m/denom  (call site fo_meta_interface.py:105)."""
    with torch.no_grad():
        for n, g in grads.items():
            st = state.setdefault(n, {"t": 0, "m": torch.zeros_like(g), "v": torch.zeros_like(g)})
            st["t"] += 1
            st["m"].mul_(b1).add_(g, alpha=1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** st["t"], 1 - b2 ** st["t"]
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
            params[n].addcdiv_(st["m"], denom, value=-lr / bc1)


def inner_lr(cfg):
    """fo_meta_interface.py:41-45."""
    o = cfg["meta"]["optimizer_opt"]
    return cfg["d_model"] ** (-0.5) * o["k"] * (o["warmup_steps"] ** (-0.5))


def leafify(sd, cfg):
    """state_dict -> dict of leaf tensors requiring grad (tied names share one leaf)."""
    tied = cfg.get("tgt_share_weight", 0) != 0
    p = OrderedDict()
    for n, t in sd.items():
        if n == "pos_encoder.pe":
            p[n] = t.clone()
        elif tied and n == "pre_embed.weight":
            p[n] = p["char_trans.weight"]
        else:
            p[n] = t.detach().clone().requires_grad_(True)
    return p


def inner_step(p, cfg, batch, eps, bufs, lr, momentum=0.9, nesterov=True):
    """One inner-loop step of run_task (fo_meta_interface.py:238-250): run_batch(train) ->
    clip 5 -> NaN ? skip : SGD step.  THE unit the headline metric counts."""
    info, grads, _, _ = run_batch_train(p, cfg, batch, eps)
    gn = clip_grad_norm_(grads)
    if not math.isnan(gn):
        sgd_nesterov_step(p, grads, bufs, lr, momentum, nesterov)
    info["grad_norm"] = gn
    return info


def fomaml_meta_step(meta, cfg, tasks, eps, adam_state, meta_step_num, momentum=0.9, nesterov=True, keep=None):
    """One outer step (fo_meta_interface.py:139-156,180-250).

    meta: name -> tensor (the `_original` weights, updated in place);
    tasks: list of (train_batches, val_batch); each batch = (xs_pad, ilens, ys, olens).
    Per task: copy meta -> model, fresh SGD, k inner steps, val fwd+bwd at adapted
    weights, clip 5 (NaN only warns), accumulate.  Then /= counter, Noam-Adam step.
    Returns (per-task val infos, lr).  `keep` (dict, optional) receives what the checks of the outer loop need:
    'inner_infos' (per task, the infos of its k inner steps), 'meta_grad' (name -> mean clipped val gradient, before
    Adam) and 'last_adapted' (the last task's adapted weights: what the reference evaluates and snapshots, quirks Q1/Q2).
    """
    names = grad_param_names(meta, cfg)
    updates = {n: torch.zeros_like(meta[n]) for n in names}
    infos, inner_infos = [], []
    lr_in = inner_lr(cfg)
    p = None
    for tr_batches, val_batch in tasks:
        p = leafify(meta, cfg)
        bufs = {}
        inner_infos.append([inner_step(p, cfg, b, eps, bufs, lr_in, momentum, nesterov) for b in tr_batches])
        info, grads, _, _ = run_batch_train(p, cfg, val_batch, eps)
        info["grad_norm"] = clip_grad_norm_(grads)
        for n in names:
            updates[n] += grads[n]
        infos.append(info)
    for n in names:
        updates[n] /= len(tasks)
    if keep is not None:
        keep["inner_infos"] = inner_infos
        keep["meta_grad"] = {n: u.clone() for n, u in updates.items()}
        keep["last_adapted"] = p
    o = cfg["meta"]["optimizer_opt"]
    lr = noam_lr(meta_step_num, o["k"], cfg["d_model"], o["warmup_steps"])
    adam_step(meta, updates, adam_state, lr)
    return infos, lr


def run_batch_eval(p, cfg, batch, eps):
    """run_batch(train=False) without the CER/WER text metrics (transformer_torch_trainer.py:59-84,94-97): loss and
    accuracy of the teacher-forced forward, plus the arg-max tokens and gold the metrics are computed from."""
    xs_pad, ilens, ys, olens = batch
    with torch.no_grad():
        logit, gold = model_forward(p, cfg, xs_pad, ilens, ys, olens)
        loss, n_correct, n_total = label_smoothed_ce(logit, gold, eps)
    return {"loss": float(loss), "acc": float(n_correct) / n_total}, logit, gold


def reptile_meta_step(meta, cfg, tasks, eps, adam_state, meta_step_num, momentum=0.9, nesterov=True, keep=None):
    """Reptile outer step.  NOT in the reference: `--algo reptile` reaches `raise ValueError` in
    fo_meta_interface.py:197-198, so this restates the published algorithm (Nichol et al. 2018, eq. 5) inside
    the reference's loop structure and is "parity unpinned": per task the pseudo-gradient is
    theta_meta - theta_k (k inner SGD steps as in run_task); the val batch is still evaluated at theta_k for the logged
    loss/acc but contributes no gradient; mean over tasks; Noam-Adam on the pseudo-gradient.
    """
    names = grad_param_names(meta, cfg)
    updates = {n: torch.zeros_like(meta[n]) for n in names}
    infos, inner_infos = [], []
    lr_in = inner_lr(cfg)
    p = None
    for tr_batches, val_batch in tasks:
        p = leafify(meta, cfg)
        bufs = {}
        inner_infos.append([inner_step(p, cfg, b, eps, bufs, lr_in, momentum, nesterov) for b in tr_batches])
        info, _, _, _ = run_batch_train(p, cfg, val_batch, eps)
        for n in names:
            updates[n] += meta[n].detach() - p[n].detach()
        infos.append(info)
    for n in names:
        updates[n] /= len(tasks)
    if keep is not None:
        keep["inner_infos"] = inner_infos
        keep["meta_grad"] = {n: u.clone() for n, u in updates.items()}          # the pseudo-gradient mean_k (theta_meta - theta_k), before Adam
        keep["last_adapted"] = p
    o = cfg["meta"]["optimizer_opt"]
    lr = noam_lr(meta_step_num, o["k"], cfg["d_model"], o["warmup_steps"])
    adam_step(meta, updates, adam_state, lr)
    return infos, lr


# --------------------------------------------------------------------------- #
# data (src/io/dataset.py)
# --------------------------------------------------------------------------- #
def bucket_sampler_plan(ilens, min_ilen, max_ilen, half_batch_ilen, batch_size):
    """BucketSampler._create_buckets/_get_batch_size (dataset.py:35-110), bucket_size 1,
    bucket_reverse False.  Consumes python `random` (bucket order).  Returns
    [(bin_idx, index array, batch_size)]."""
    lb = min(2, 1) if not min_ilen else min_ilen
    ub = max(10000, int(np.max(ilens))) if not max_ilen else max_ilen
    half = half_batch_ilen if half_batch_ilen else 10000
    bins = np.arange(lb, ub, 1)
    bucket_idx = np.digitize(ilens, bins, right=True)
    half_idx = np.digitize(half, bins, right=True)
    buckets = []
    for bin_idx in range(1, len(bins) - 1):
        b = np.where(bucket_idx == bin_idx)[0]
        if len(b) > 0:
            buckets.append((bin_idx, b))
    random.shuffle(buckets)
    return [(bi, b, max(1, batch_size // 2) if bi > half_idx else batch_size) for bi, b in buckets]


def bucket_sampler_epoch(plan, drop_last=False):
    """BucketSampler.__iter__ (dataset.py:49-63): a GENERATOR -- each bucket is
    np.random.shuffle'd in place only when iteration reaches it (the lazy RNG consumption
    matters when several accents' iterators are interleaved), then cut into consecutive chunks."""
    for _, bucket, bs in plan:
        np.random.shuffle(bucket)
        batch = []
        for idx in bucket:
            batch.append(int(idx))
            if len(batch) == bs:
                yield batch
                batch = []
        if batch and not drop_last:
            yield batch


def collate(feat, iptr, label, optr, ilens, olens, idxs):
    """CommonVoiceDataset.__getitem__ + collate_fn (dataset.py:21-33,147-153): sort by
    ilen desc (stable), zero-pad to the longest."""
    idxs = sorted(idxs, key=lambda i: int(ilens[i]), reverse=True)
    Tmax = max(int(ilens[i]) for i in idxs)
    xs = torch.zeros(len(idxs), Tmax, feat.shape[1])
    for b, i in enumerate(idxs):
        xs[b, : int(ilens[i])] = torch.from_numpy(np.ascontiguousarray(feat[iptr[i]:iptr[i + 1]]))
    il = torch.tensor([int(ilens[i]) for i in idxs])
    ys = [torch.from_numpy(np.asarray(label[optr[i]:optr[i + 1]]).astype(np.int64)) for i in idxs]
    ol = torch.tensor([int(olens[i]) for i in idxs])
    return xs, il, ys, ol


# --------------------------------------------------------------------------- #
# CTC (config 1 only; call site src/blstm_trainer.py:22,55-70)
# --------------------------------------------------------------------------- #
def ctc_loss_np(log_probs, targets, input_lengths, target_lengths, blank=0):
    """Plain alpha/beta CTC in float64 numpy, semantics of
    nn.CTCLoss(blank=0, reduction='mean', zero_infinity=True):
    per-sample nll / target_length, then batch mean; inf -> 0 (and zero grad).

    log_probs [T,B,C] (already log-softmaxed), targets concatenated [sum(tl)].
    Returns (loss, grad wrt log_probs [T,B,C]) -- grad as torch's CTC backward defines it
    (d loss / d log_probs, NOT folded through the softmax).
    """
    lp = np.asarray(log_probs, dtype=np.float64)
    T, B, C = lp.shape
    grad = np.zeros_like(lp)
    NEG = -np.inf
    total = 0.0
    off = 0
    for b in range(B):
        Tb, Lb = int(input_lengths[b]), int(target_lengths[b])
        tgt = np.asarray(targets[off:off + Lb], dtype=np.int64)
        off += Lb
        S = 2 * Lb + 1
        ext = np.full(S, blank, dtype=np.int64)
        ext[1::2] = tgt
        la = np.full((Tb, S), NEG)
        la[0, 0] = lp[0, b, blank]
        if S > 1:
            la[0, 1] = lp[0, b, ext[1]]
        for t in range(1, Tb):
            for s in range(S):
                a = la[t - 1, s]
                if s > 0:
                    a = np.logaddexp(a, la[t - 1, s - 1])
                if s > 1 and ext[s] != blank and ext[s] != ext[s - 2]:
                    a = np.logaddexp(a, la[t - 1, s - 2])
                la[t, s] = a + lp[t, b, ext[s]]
        ll = la[Tb - 1, S - 1]
        if S > 1:
            ll = np.logaddexp(ll, la[Tb - 1, S - 2])
        nll = -ll
        if not np.isfinite(nll):
            continue                                                 # zero_infinity
        lb = np.full((Tb, S), NEG)
        lb[Tb - 1, S - 1] = lp[Tb - 1, b, ext[S - 1]]
        if S > 1:
            lb[Tb - 1, S - 2] = lp[Tb - 1, b, ext[S - 2]]
        for t in range(Tb - 2, -1, -1):
            for s in range(S):
                a = lb[t + 1, s]
                if s + 1 < S:
                    a = np.logaddexp(a, lb[t + 1, s + 1])
                if s + 2 < S and ext[s + 2] != blank and ext[s + 2] != ext[s]:
                    a = np.logaddexp(a, lb[t + 1, s + 2])
                lb[t, s] = a + lp[t, b, ext[s]]
        scale = 1.0 / (max(Lb, 1) * B)
        total += nll * scale
        # d nll / d lp[t,c] = -exp(alpha+beta - lp - ll) summed over s with ext[s]==c
        for t in range(Tb):
            for s in range(S):
                ab = la[t, s] + lb[t, s]
                if np.isfinite(ab):
                    grad[t, b, ext[s]] -= np.exp(ab - lp[t, b, ext[s]] - ll) * scale
    return total, grad


def radam_torch_optimizer_step(p, g, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One step of `torch_optimizer.RAdam` -- the class the reference's `optimizer_cls: RAdam` resolves to
    (src/transformer_torch_trainer.py:36-41).  The package (jettify/pytorch-optimizer, which carries the RAdam authors' published
    implementation) is NOT vendored in /root/reference and not installed here: this restates its published update rule, **parity
    unpinned** against the package itself.  Differences from torch.optim.RAdam: rectification from N_sma >= 5 (not > 5), eps added to
    sqrt(v) WITHOUT the bias correction of v (that factor sits in the step size), weight decay applied to the weight
    (p += -wd * lr * p) instead of to the gradient.  `state` = {'step', 'exp_avg', 'exp_avg_sq'}, updated in place; p in place."""
    b1, b2 = betas
    state['step'] += 1
    t = state['step']
    state['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1 - b2)
    state['exp_avg'].mul_(b1).add_(g, alpha=1 - b1)
    beta2_t = b2 ** t
    n_sma_max = 2 / (1 - b2) - 1
    n_sma = n_sma_max - 2 * t * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma * n_sma_max / (n_sma_max - 2)) / (1 - b1 ** t)
    else:
        step_size = lr / (1 - b1 ** t)
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr)
    if n_sma >= 5:
        p.addcdiv_(state['exp_avg'], state['exp_avg_sq'].sqrt().add_(eps), value=-step_size)
    else:
        p.add_(state['exp_avg'], alpha=-step_size)
    return p
