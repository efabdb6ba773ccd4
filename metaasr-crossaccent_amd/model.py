"""MyTransformer: the reference's model object (src/model/transformer_pytorch/mono_transformer_torch.py:35-208)
as a facade over MasrEngine.  It keeps the surface the interfaces use -- state_dict / load_state_dict /
parameters / train / eval / forward / recog, sos_id / eos_id -- while every FLOP runs in libmasr."""
from collections import OrderedDict

import torch

from .engine import MasrEngine


def reference_init_state_dict(model_para, odim):
    """Initial weights identical to the reference's for the same torch seed: replays the reference's RNG consumption
    (module construction order of mono_transformer_torch.py:49-104, then xavier_uniform_ over parameters() with
    dim > 1, :106-109) with throw-away torch.nn modules on the host.  Pinned by tests/golden/init.npz."""
    import warnings
    from torch import nn
    p = model_para
    E, H = p['d_model'], p['nheads']
    feat = nn.Sequential(nn.Conv2d(1, 64, 3, 1, 1), nn.ReLU(), nn.Conv2d(64, 64, 3, 1, 1), nn.ReLU(), nn.MaxPool2d(2, 2),
                         nn.Conv2d(64, 128, 3, 1, 1), nn.ReLU(), nn.Conv2d(128, 128, 3, 1, 1), nn.ReLU(), nn.MaxPool2d(2, 2))
    vgg2enc = nn.Linear(128 * (p['idim'] // 4), E)
    char_trans = nn.Linear(E, odim)
    pre_embed = nn.Embedding(odim, E)
    if p.get('tgt_share_weight', 0) != 0:
        char_trans.weight = pre_embed.weight
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc_layer = nn.TransformerEncoderLayer(E, H, p['d_inner'], p.get('dropout', 0.0))
        enc = nn.TransformerEncoder(enc_layer, p['encoder']['nlayers'], nn.LayerNorm(E))
        dec_layer = nn.TransformerDecoderLayer(E, H, p['d_inner'], p.get('dropout', 0.0))
        dec = nn.TransformerDecoder(dec_layer, p['decoder']['nlayers'], nn.LayerNorm(E))
    mods = OrderedDict(feat_extractor=feat, vgg2enc=vgg2enc, char_trans=char_trans, pre_embed=pre_embed, encoder=enc, decoder=dec)
    seen = set()
    for mod in mods.values():
        for prm in mod.parameters():
            if id(prm) in seen:
                continue
            seen.add(id(prm))
            if prm.dim() > 1:
                nn.init.xavier_uniform_(prm)
    sd = OrderedDict()
    for mname, mod in mods.items():
        for n, t in mod.state_dict().items():
            sd[f"{mname}.{n}"] = t.detach()
    return sd


class MyTransformer:
    def __init__(self, id2char, model_para, label_smoothing=0.0, device="cuda:0", init=True):
        self.idim = model_para['idim']
        self.odim = len(id2char)
        self.sos_id, self.eos_id = 0, len(id2char) - 1                      # :45-46
        self.d_model, self.nhead = model_para['d_model'], model_para['nheads']
        self.engine = MasrEngine(model_para, self.odim, label_smoothing, device)
        self.training = True
        if init:
            self.init_parameters()

    @property
    def device(self):
        return self.engine.device

    def cuda(self):
        return self                                                          # already there (reference hard-codes .cuda(), F9)

    def init_parameters(self):
        self.engine.load_state_dict(reference_init_state_dict(self.engine.model_para, self.odim))

    # ---- nn.Module-like surface ---------------------------------------------------------------
    def parameters(self):
        return [self.engine.params]

    def state_dict(self, keep_vars=False):
        return self.engine.state_dict(clone=not keep_vars)

    def load_state_dict(self, sd, strict=True):
        self.engine.load_state_dict(sd)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ---- operator ------------------------------------------------------------------------------
    def forward(self, xs_pad, ilens, ys, olens):
        """(:178-208) -> (logit [B,L,odim] on device, ys_out_pad [B,L] int64 on device, -1 padded).
        Inference-only entry (no autograd graph exists on this path); training goes through run_batch.
        Reproduces quirk Q6: olens is incremented in place."""
        assert xs_pad.size(0) == ilens.size(0) == len(ys) == olens.size(0), "Batch size mismatch"
        self.engine.run_batch(xs_pad, ilens, ys, olens, train=False)
        olens += 1
        logit, gold = self.engine.last_logits()
        return logit, gold.to(torch.int64)

    __call__ = forward

    def recog(self, xs_pad, ilens):
        """greedy decode (:143-176): encoder once, max(enc_lens) full re-decodes, arg-max of every position -> [Ldec, B]"""
        assert xs_pad.size(0) == ilens.size(0), "Batch size mismatch"
        return self.engine.recog(xs_pad, ilens)
