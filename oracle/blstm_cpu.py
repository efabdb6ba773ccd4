"""CPU restatement of the reference's BLSTM-CTC model (config/blstm) -- TEST INFRASTRUCTURE ONLY.

Follows MonoBLSTM.forward (src/model/blstm/mono_blstm.py:77-92), BlstmEncoder.forward (src/modules/encoder.py:281-298),
RNNP.forward (src/modules/encoder.py:127-157; packed bidirectional nn.LSTM -> Linear -> tanh per layer) and the loss of
BLSTMTrainer.run_batch (src/blstm_trainer.py:55-70).  The LSTM recurrence is written out explicitly (own gate math and
own packed-sequence handling: a sequence takes no part in steps t >= len; the reverse direction starts from the zero state
at each sequence's own last frame); convolutions, pooling and the CTC loss use torch's functional primitives, autograd
gives the backward.  Pinned to the real reference by tests/golden/blstm_tiny.npz (oracle/make_goldens.py).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


def param_shapes(cfg: dict, odim: int) -> "OrderedDict[str, tuple]":
    """state_dict keys / shapes of MonoBLSTM in the reference's order."""
    e = cfg["encoder"]
    H, proj, eodim = e["enc_dim"], e["proj_dim"], e["odim"]
    nl = len(e["sample_rate"].split("_"))
    Dp = int(np.ceil(np.ceil(e["idim"] / 2) / 2))
    sh = OrderedDict()
    for idx, (co, ci) in zip((0, 2, 5, 7), ((128, 1), (128, 128), (256, 128), (256, 256))):
        sh[f"encoder.vgg.{idx}.weight"] = (co, ci, 3, 3)
        sh[f"encoder.vgg.{idx}.bias"] = (co,)
    for i in range(nl):
        K = 256 * Dp if i == 0 else proj
        for suf in ("", "_reverse"):
            sh[f"encoder.blstm.rnn{i}.weight_ih_l0{suf}"] = (4 * H, K)
            sh[f"encoder.blstm.rnn{i}.weight_hh_l0{suf}"] = (4 * H, H)
            sh[f"encoder.blstm.rnn{i}.bias_ih_l0{suf}"] = (4 * H,)
            sh[f"encoder.blstm.rnn{i}.bias_hh_l0{suf}"] = (4 * H,)
        N = eodim if i == nl - 1 else proj
        sh[f"encoder.blstm.bt{i}.weight"] = (N, 2 * H)
        sh[f"encoder.blstm.bt{i}.bias"] = (N,)
    sh["head.weight"] = (odim, eodim)
    sh["head.bias"] = (odim,)
    return sh


def deterministic_state_dict(cfg: dict, odim: int, seed: int) -> "OrderedDict[str, torch.Tensor]":
    """Seeded weights independent of torch's module-construction RNG order: N(0, 1/sqrt(fan_in)) matrices, small biases."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for n, s in param_shapes(cfg, odim).items():
        if len(s) == 1:
            sd[n] = 0.05 * torch.randn(s, generator=g)
        else:
            fan = int(np.prod(s[1:]))
            sd[n] = torch.randn(s, generator=g) / np.sqrt(fan)
    return sd


def enc_lens_of(ilens) -> torch.Tensor:
    il = np.asarray(ilens, dtype=np.float32)
    return torch.from_numpy(np.ceil(np.ceil(il / 2) / 2).astype(np.int64))


def _lstm_dir(x, lens, wih, whh, bih, bhh, reverse):
    """x [B,T,K] -> y [B,T,H]; packed-sequence semantics, gate order i, f, g, o."""
    B, T, _ = x.shape
    H = whh.shape[1]
    gx = x @ wih.t() + bih + bhh
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    ys = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        z = gx[:, t] + h @ whh.t()
        i, f, g, o = z.chunk(4, dim=1)
        cn = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        hn = torch.sigmoid(o) * torch.tanh(cn)
        live = (t < lens).to(x.dtype).unsqueeze(1)
        c = live * cn + (1 - live) * c
        h = live * hn + (1 - live) * h
        ys[t] = live * hn
    return torch.stack(ys, dim=1)


def forward(p, cfg, xs_pad, ilens):
    """-> (logits [B, T', odim], enc_lens [B])"""
    e = cfg["encoder"]
    rates = [int(v) for v in e["sample_rate"].split("_")]
    nl = len(rates)
    assert all(float(v) == 0 for v in e["dropout"].split("_"))
    x = xs_pad.unsqueeze(1)                                                        # [B,1,T,D]
    x = F.relu(F.conv2d(x, p["encoder.vgg.0.weight"], p["encoder.vgg.0.bias"], padding=1))
    x = F.relu(F.conv2d(x, p["encoder.vgg.2.weight"], p["encoder.vgg.2.bias"], padding=1))
    x = F.max_pool2d(x, 2, stride=2, ceil_mode=True)
    x = F.relu(F.conv2d(x, p["encoder.vgg.5.weight"], p["encoder.vgg.5.bias"], padding=1))
    x = F.relu(F.conv2d(x, p["encoder.vgg.7.weight"], p["encoder.vgg.7.bias"], padding=1))
    x = F.max_pool2d(x, 2, stride=2, ceil_mode=True)
    lens = enc_lens_of(ilens)
    x = x.transpose(1, 2).contiguous().view(x.size(0), x.size(2), -1)              # [B,T',C*D'] (feature = c*D' + d)
    for i in range(nl):
        pre = f"encoder.blstm.rnn{i}."
        yf = _lstm_dir(x, lens, p[pre + "weight_ih_l0"], p[pre + "weight_hh_l0"], p[pre + "bias_ih_l0"], p[pre + "bias_hh_l0"], False)
        yb = _lstm_dir(x, lens, p[pre + "weight_ih_l0_reverse"], p[pre + "weight_hh_l0_reverse"], p[pre + "bias_ih_l0_reverse"],
                       p[pre + "bias_hh_l0_reverse"], True)
        y = torch.cat([yf, yb], dim=2)
        if rates[i] > 1:                                                           # RNNP.forward, encoder.py:118-121 (:149-152)
            y = y[:, ::rates[i]]
            lens = (lens + 1) // rates[i]
        x = torch.tanh(y @ p[f"encoder.blstm.bt{i}.weight"].t() + p[f"encoder.blstm.bt{i}.bias"])
    mask = (torch.arange(x.size(1)).unsqueeze(0) >= lens.unsqueeze(1)).unsqueeze(-1)
    x = x.masked_fill(mask, 0.0)
    return x @ p["head.weight"].t() + p["head.bias"], lens


def run_batch(p, cfg, batch, odim):
    """BLSTMTrainer.run_batch: -> (loss tensor, logits, enc_lens); targets [sos] + y + [eos], sos = eos = odim - 1"""
    xs, ilens, ys, olens = batch
    eos = torch.tensor([odim - 1], dtype=torch.int64)
    y_true = torch.cat([torch.cat([eos, y, eos]) for y in ys])
    logits, lens = forward(p, cfg, xs, ilens)
    logp = F.log_softmax(logits, dim=-1)
    loss = F.ctc_loss(logp.transpose(0, 1).contiguous(), y_true, lens, olens + 2, blank=0, reduction="mean", zero_infinity=True)
    return loss, logits, lens
