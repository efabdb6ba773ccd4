"""Error behaviour of the C ABI and the NaN guard of the inner step (GPU).  The reference raises Python exceptions /
asserts and skips the optimiser step when the gradient norm is NaN (fo_meta_interface.py:242-248, multi_interface.py:108-112);
libmasr returns non-zero codes with a message in masr_last_error and decides the skip on the device."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402
from masr_amd import _cabi  # noqa: E402
from masr_amd.engine import MasrEngine  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.make_goldens import TINY, ODIM, synth_batch  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    e = MasrEngine(TINY, ODIM, label_smoothing=0.2)
    e.load_state_dict(ref_cpu.deterministic_state_dict(TINY, ODIM, seed=7))
    return e


def test_bad_arguments_are_reported_not_executed(eng):
    L = _cabi.lib()
    xs, il, ys, ol = synth_batch(3, [40, 32], [5, 4])
    # label outside the vocabulary
    bad = [ys[0].clone(), ys[1].clone()]
    bad[0][0] = ODIM + 3
    with pytest.raises(Exception, match="label out of range"):
        eng.run_batch(xs, il, bad, ol, train=True)
    # an utterance longer than the padded batch / shorter than one encoder frame
    with pytest.raises(Exception, match="ilens"):
        eng.run_batch(xs, torch.tensor([41, 32]), ys, ol, train=True)
    with pytest.raises(Exception, match="ilens"):
        eng.run_batch(xs, torch.tensor([40, 3]), ys, ol, train=True)
    # feature width must match the model
    with pytest.raises(AssertionError, match="idim"):
        eng.run_batch(torch.zeros(2, 40, 80), il, ys, ol, train=True)
    # a config the kernels do not cover is refused at creation
    cfg = _cabi.MasrConfig(idim=83, odim=367, d_model=96, nheads=4, d_inner=128, enc_layers=1, dec_layers=1, tie_weights=1,
                           dropout=0.0, pos_dropout=0.0, label_smoothing=0.0)
    assert not L.masr_create(C.byref(cfg)) and b"masr_create" in L.masr_last_error()
    # binding an undersized workspace
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    rc = L.masr_bind(eng.h, C.c_void_p(eng.params.data_ptr()), C.c_void_p(eng.grads.data_ptr()), C.c_void_p(eng.pe.data_ptr()),
                     C.c_void_p(small.data_ptr()), small.numel())
    assert rc != 0 and b"masr_bind" in L.masr_last_error()
    # the engine is still usable afterwards (nothing was launched by the failed calls)
    eng._ensure_ws(4, 64, 16)
    _cabi.check(L.masr_bind(eng.h, C.c_void_p(eng.params.data_ptr()), C.c_void_p(eng.grads.data_ptr()), C.c_void_p(eng.pe.data_ptr()),
                            C.c_void_p(eng.ws.data_ptr()), eng.ws.numel()), "masr_bind")
    eng.mark_dirty()
    eng.run_batch(xs, il, ys, ol, train=True)
    assert np.isfinite(eng.read_stats()["loss"])


def test_nan_gradient_skips_the_inner_step(eng):
    """`if math.isnan(grad_norm): warn else: opt.step()`: a batch with an inf feature gives a NaN norm; the fused
    clip + SGD kernel must leave parameters and momentum untouched and report the NaN norm."""
    xs, il, ys, ol = synth_batch(3, [40, 32], [5, 4])
    xs = xs.clone()
    xs[0, 5, 7] = float("inf")
    before = eng.params.clone()
    mom = torch.full_like(eng.params, 0.25)
    eng.run_batch(xs, il, ys, ol, train=True)
    eng.clip_sgd_step(mom, 5.0, 0.1, 0.9, True, first_step=False)
    st = eng.read_stats()
    assert np.isnan(st["grad_norm"])
    assert torch.equal(eng.params, before) and bool((mom == 0.25).all())
    # a clean batch afterwards trains normally
    xs2, il2, ys2, ol2 = synth_batch(4, [40, 32], [5, 4])
    eng.run_batch(xs2, il2, ys2, ol2, train=True)
    eng.clip_sgd_step(mom, 5.0, 0.1, 0.9, True, first_step=False)
    st = eng.read_stats()
    assert np.isfinite(st["grad_norm"]) and not torch.equal(eng.params, before)
    eng.load_state_dict(ref_cpu.deterministic_state_dict(TINY, ODIM, seed=7))


def test_ctc_lengths_outside_the_call_are_refused_by_the_kernel():
    """masr_ctc_loss takes its lengths as DEVICE arrays (src/blstm_trainer.py:62-70 hands torch tensors to nn.CTCLoss), so the
    kernel vets them: in_len > T, a negative length or a target wider than the lattice the work buffer holds -> that utterance is
    not run (NaN nll -> NaN mean: loud; zero gradient rows; nothing read or written out of bounds) and masr_ctc_status names it;
    in_len == 0 is torch's no-path case (loss 0, zero gradient) -- the rest of the batch is what torch computes."""
    L = _cabi.lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(9)
    T, B, Cc = 24, 4, 23
    logits = torch.randn(T, B, Cc, generator=g)
    tl = torch.tensor([5, 3, 4, 2])
    tgt = torch.randint(1, Cc, (int(tl.sum()),), generator=g)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), tl.cumsum(0)[:-1]])
    d = lambda t: t.to(torch.int32).cuda()
    maxS = int(2 * tl.max() + 1)
    lg = logits.cuda().contiguous()

    def run(il, tl_=tl):
        work = torch.zeros(int(L.masr_ctc_work_floats(T, B, maxS)), device="cuda")
        nll, loss, grad = torch.zeros(B, device="cuda"), torch.zeros(1, device="cuda"), torch.full_like(lg, 3.0)
        tg, of, ild, tld = d(tgt), d(off), d(il), d(tl_)
        _cabi.check(L.masr_ctc_loss(P(lg), P(tg), P(of), P(ild), P(tld), T, B, Cc, 0, P(nll), P(loss), P(grad), P(work), maxS, S()))
        return L.masr_ctc_status(P(work), T, B, maxS, S()), nll.cpu(), float(loss), grad.cpu(), work

    # ---- in_len == 0 on utterance 1: torch says inf -> 0 (zero_infinity), zero gradient; the others as torch computes them
    il = torch.tensor([24, 0, 20, 18])
    st, nll, loss, grad, _ = run(il)
    lr = logits.clone().requires_grad_(True)
    ref = torch.nn.CTCLoss(blank=0, reduction="mean", zero_infinity=True)(torch.log_softmax(lr, -1), tgt, il, tl)
    ref.backward()
    assert st == 0 and float(nll[1]) == 0.0 and abs(loss - float(ref)) <= 2e-5 * float(ref)
    assert float(grad[:, 1].abs().max()) == 0.0
    torch.testing.assert_close(grad, lr.grad, rtol=2e-3, atol=2e-6)
    # ---- in_len > T on utterance 2, negative on 0, a target wider than maxS on 3: refused, named, NaN; utterance 1 is still right
    for il_bad, tl_bad, who in ((torch.tensor([24, 22, 25, 18]), tl, 2), (torch.tensor([-1, 22, 20, 18]), tl, 0),
                                (torch.tensor([24, 22, 20, 18]), torch.tensor([5, 3, 4, 9]), 3)):
        st, nll, loss, grad, work_bad = run(il_bad, tl_bad)
        assert st == who + 1 and b"mk_ctc_loss" in L.masr_last_error() and str(who).encode() in L.masr_last_error()
        assert np.isnan(float(nll[who])) and np.isnan(loss) and float(grad[:, who].abs().max()) == 0.0
        ok = [b for b in range(B) if b != who]
        assert all(np.isfinite(float(nll[b])) and float(nll[b]) > 0 for b in ok) and bool(torch.isfinite(grad).all())
        # the mark lives in the call's own work buffer: a healthy call on another buffer reads 0 while this one still names its utterance
        assert run(il)[0] == 0 and L.masr_ctc_status(P(work_bad), T, B, maxS, S()) == who + 1
    # ---- arguments the host CAN see are refused before anything is launched
    work = torch.zeros(int(L.masr_ctc_work_floats(T, B, maxS)), device="cuda")
    z = torch.zeros(B, device="cuda")
    assert L.masr_ctc_loss(P(lg), P(d(tgt)), P(d(off)), P(d(il)), P(d(tl)), T, B, Cc, Cc, P(z), P(z), P(torch.zeros_like(lg)), P(work), maxS, S()) != 0
    assert b"blank" in L.masr_last_error()
