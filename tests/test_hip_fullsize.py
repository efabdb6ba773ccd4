"""Full-size (BASELINE.json bench shape: B 16 x 1000 frames x 80 dims, fometa-hkust geometry) checks of the HIP path through
size-independent properties -- the CPU oracle needs ~2 s per utterance here, so instead of element-wise comparison:
determinism, batch-permutation invariance, zero-padding invariance, consistency of the reported gradient norm, and the
eval == train loss identity without dropout.  Plus the edge shapes the reference's data path can produce."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402
from masr_amd.engine import MasrEngine  # noqa: E402
from masr_amd.model import reference_init_state_dict  # noqa: E402

HK = {"idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.0, "pos_dropout": 0.0, "tgt_share_weight": 1,
      "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}}
ODIM = 367


def batch(B, T, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    ilens = torch.tensor(sorted([T - (37 * i) % (T // 3) if ragged else T for i in range(B)], reverse=True))
    xs = torch.randn(B, T, 80, generator=g)
    for b in range(B):
        xs[b, ilens[b]:] = 0
    olens = torch.tensor([10 + (7 * i) % 30 for i in range(B)])
    ys = [torch.randint(1, 366, (int(n),), generator=g) for n in olens]
    return xs, ilens, ys, olens


@pytest.fixture(scope="module")
def eng():
    torch.manual_seed(531)
    e = MasrEngine(HK, ODIM, label_smoothing=0.2)
    e.load_state_dict(reference_init_state_dict(HK, ODIM))
    return e


def step(eng, xs, ilens, ys, olens, train=True):
    eng.run_batch(xs, ilens, ys, olens.clone(), train=train)
    if train:
        eng.grad_norm()
    st = eng.read_stats()
    return st, (eng.grads.clone() if train else None)


def test_fullsize_determinism_and_grad_norm(eng):
    xs, il, ys, ol = batch(16, 1000, 1)
    s1, g1 = step(eng, xs, il, ys, ol)
    s2, g2 = step(eng, xs, il, ys, ol)
    assert s1 == s2 and torch.equal(g1, g2)                          # fixed reduction orders everywhere -> bit-identical
    assert np.isfinite(s1["loss"]) and 5.0 < s1["loss"] < 7.5        # ~ln(367) + smoothing at random init
    assert s1["n_total"] == float((ol + 1).sum())
    assert abs(float(g1.double().norm()) - s1["grad_norm"]) < 1e-4 * s1["grad_norm"]
    ev, _ = step(eng, xs, il, ys, ol, train=False)
    assert ev["loss"] == s1["loss"] and ev["n_correct"] == s1["n_correct"]      # dropout 0: eval forward == train forward


def test_fullsize_batch_permutation_invariance(eng):
    xs, il, ys, ol = batch(16, 1000, 2, ragged=False)                # equal lengths: any order is a valid sorted batch
    s1, g1 = step(eng, xs, il, ys, ol)
    perm = torch.randperm(16, generator=torch.Generator().manual_seed(3))
    s2, g2 = step(eng, xs[perm], il[perm], [ys[i] for i in perm], ol[perm])
    assert abs(s1["loss"] - s2["loss"]) < 2e-6 * s1["loss"] and s1["n_correct"] == s2["n_correct"]
    rel = float((g1 - g2).norm() / g1.norm())
    print("gradient change under batch permutation:", rel)
    assert rel < 1e-4                                                # only the order of fp32 partial sums changes (measured 2e-7)


def test_fullsize_zero_padding_invariance(eng):
    """extra all-zero frames at the end of the batch tensor change nothing as long as every utterance ends >= 16 frames
    before it: the VGG's receptive field is +-6 input frames, and the encoder masks frames >= ilens // 4.  (An utterance
    that fills the tensor is NOT invariant -- conv(0) + bias != the zero padding of the tensor edge -- in the reference too.)"""
    xs, il, ys, ol = batch(16, 1000, 4)
    il = torch.clamp(il, max=976)
    for b in range(16):
        xs[b, il[b]:] = 0
    s1, g1 = step(eng, xs, il, ys, ol)
    xp = torch.cat([xs, torch.zeros(16, 24, 80)], dim=1)
    s2, g2 = step(eng, xp, il, ys, ol)
    assert abs(s1["loss"] - s2["loss"]) < 2e-6 * s1["loss"]
    rel = float((g1 - g2).norm() / g1.norm())
    print("gradient change under 24 frames of zero padding:", rel)
    assert rel < 1e-4                                                # measured 1e-7


@pytest.mark.parametrize("B,T,ilens,olens", [
    (1, 4, [4], [1]),                     # shortest legal utterance: one encoder frame, one label
    (2, 7, [7, 5], [1, 2]),               # T not a multiple of 4, floor pooling drops frames
    (3, 1501, [1501, 1500, 1499], [40, 1, 17]),   # longer than the 1500-frame upper bucket of the shipped configs
    (32, 200, [200] * 32, [12] * 32),     # the B=32 full-batch rule below half_batch_ilen
])
def test_edge_shapes_run_and_are_finite(B, T, ilens, olens):
    small = dict(HK, d_model=128, nheads=4, d_inner=256, encoder={"nlayers": 1}, decoder={"nlayers": 1})
    torch.manual_seed(1)
    e = MasrEngine(small, ODIM, label_smoothing=0.2)
    e.load_state_dict(reference_init_state_dict(small, ODIM))
    g = torch.Generator().manual_seed(B + T)
    xs = torch.randn(B, T, 80, generator=g)
    il, ol = torch.tensor(ilens), torch.tensor(olens)
    for b in range(B):
        xs[b, il[b]:] = 0
    ys = [torch.randint(1, 366, (n,), generator=g) for n in olens]
    st, gr = step(e, xs, il, ys, ol)
    assert np.isfinite(st["loss"]) and np.isfinite(st["grad_norm"]) and st["grad_norm"] > 0
    assert st["n_total"] == float(sum(olens) + B) and bool(torch.isfinite(gr).all())
    hyp = e.recog(xs, il)
    assert hyp.shape == (max(ilens) // 4, B) and int(hyp.min()) >= 0 and int(hyp.max()) < ODIM
