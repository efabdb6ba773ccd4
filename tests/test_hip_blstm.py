"""BLSTM-CTC path on the GPU through the C ABI (masr_blstm_*): forward logits, CTC loss and every
parameter gradient against the reference's golden (tests/golden/blstm_tiny.npz, real MonoBLSTM) and the CPU oracle with
and without bf16 emulation of the MFMA operands."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402
from masr_amd.blstm_engine import BlstmEngine  # noqa: E402
from oracle import blstm_cpu  # noqa: E402
from oracle.make_goldens import BLSTM_TINY, ODIM, synth_batch  # noqa: E402


@pytest.fixture(scope="module")
def sd():
    return blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11)


def test_param_table_is_the_reference_state_dict(sd, golden_dir):
    eng = BlstmEngine(BLSTM_TINY, ODIM)
    g = np.load(golden_dir / "blstm_tiny.npz")
    assert list(eng.table) == g["state_dict_keys"].tolist() == list(sd)
    for n, (off, shape) in eng.table.items():
        assert tuple(sd[n].shape) == shape and off % 4 == 0
    eng.load_state_dict(sd)
    back = eng.state_dict()
    assert all(torch.equal(back[n].cpu(), sd[n]) for n in sd)


@pytest.mark.parametrize("tag,ilens,olens", [("ragged", [61, 50, 38, 30], [7, 5, 4, 3]), ("single", [45], [6])])
def test_run_batch_vs_reference_golden(sd, golden_dir, tag, ilens, olens):
    g = np.load(golden_dir / "blstm_tiny.npz")
    xs, il, ys, ol = synth_batch(21, ilens, olens)
    eng = BlstmEngine(BLSTM_TINY, ODIM)
    eng.load_state_dict(sd)
    eng.run_batch(xs, il, ys, ol, train=True)
    st = eng.read_stats()
    logits, lens = eng.last_logits()
    ref_logits = torch.from_numpy(g[f"{tag}/logits"])
    assert lens.cpu().tolist() == g[f"{tag}/enc_lens"].tolist()
    rel_logits = float((logits.cpu() - ref_logits).norm() / ref_logits.norm())
    rel_loss = abs(st["loss"] - float(g[f"{tag}/loss"])) / float(g[f"{tag}/loss"])
    print(f"{tag}: loss {st['loss']:.5f} vs reference {float(g[f'{tag}/loss']):.5f} (rel {rel_loss:.1e}); logits rel-L2 {rel_logits:.1e}")
    assert rel_loss < 1e-3 and rel_logits < 2e-2                      # north-star tolerance on the loss; bf16 operands on the logits
    got = eng.state_dict(flat=eng.grads)
    # full-tensor gradient checks where the golden stores them, norms everywhere
    for n in ("head.bias", "encoder.vgg.0.weight", "encoder.blstm.rnn0.weight_hh_l0_reverse"):
        ref = torch.from_numpy(g[f"{tag}/gradfull/{n}"])
        rel = float((got[n].cpu() - ref).norm() / ref.norm())
        print(f"   grad {n}: rel-L2 {rel:.2e}")
        assert rel < 0.15
    worst = 0.0
    for n in sd:
        ref_l2 = float(g[f"{tag}/grad/{n}"][2])
        mine = float(got[n].double().norm())
        if ref_l2 > 1e-6:
            worst = max(worst, abs(mine / ref_l2 - 1))
            assert abs(mine / ref_l2 - 1) < 0.15, (n, mine, ref_l2)
    gn = float(torch.cat([got[n].reshape(-1) for n in sd]).double().norm())
    print(f"   worst per-tensor gradient-norm deviation {worst:.2e}; total grad norm {gn:.4f} vs {float(g[f'{tag}/grad_norm']):.4f}")
    assert abs(gn / float(g[f"{tag}/grad_norm"]) - 1) < 3e-2


@pytest.mark.parametrize("tag,ilens,olens", [("ragged", [118, 101, 77, 60], [4, 3, 2, 2]), ("single", [90], [3])])
def test_time_subsampling_vs_reference_golden(golden_dir, tag, ilens, olens):
    """encoder.sample_rate 1_2_2: every second frame behind BLSTM layers 1 and 2 (RNNP.forward, src/modules/encoder.py:118-121), against the
    reference's own MonoBLSTM (tests/golden/blstm_sub.npz): output lengths exactly, logits, CTC loss, gradients; and every gradient
    tensor against the oracle with autograd."""
    from oracle.make_goldens import BLSTM_SUB
    g = np.load(golden_dir / "blstm_sub.npz")
    sds = blstm_cpu.deterministic_state_dict(BLSTM_SUB, ODIM, seed=12)
    xs, il, ys, ol = synth_batch(23, ilens, olens)
    eng = BlstmEngine(BLSTM_SUB, ODIM)
    eng.load_state_dict(sds)
    eng.run_batch(xs, il, ys, ol, train=True)
    st = eng.read_stats()
    logits, lens = eng.last_logits()
    ref_logits = torch.from_numpy(g[f"{tag}/logits"])
    assert lens.cpu().tolist() == g[f"{tag}/enc_lens"].tolist()
    assert logits.shape[1] >= ref_logits.shape[1]
    got_l = logits.cpu()[:, :ref_logits.shape[1]]
    rel_logits = float((got_l - ref_logits).norm() / ref_logits.norm())
    rel_loss = abs(st["loss"] - float(g[f"{tag}/loss"])) / float(g[f"{tag}/loss"])
    print(f"{tag}: loss {st['loss']:.5f} vs reference {float(g[f'{tag}/loss']):.5f} (rel {rel_loss:.1e}); logits rel-L2 {rel_logits:.1e}")
    assert rel_loss < 1e-3 and rel_logits < 2e-2
    got = eng.state_dict(flat=eng.grads)
    for n in ("head.bias", "encoder.blstm.bt1.bias", "encoder.blstm.rnn2.weight_hh_l0_reverse"):
        ref = torch.from_numpy(g[f"{tag}/gradfull/{n}"])
        rel = float((got[n].cpu() - ref).norm() / ref.norm())
        print(f"   grad {n}: rel-L2 {rel:.2e}")
        assert rel < 0.15
    # oracle autograd on the same batch: every tensor
    p = {k: v.clone().requires_grad_(True) for k, v in sds.items()}
    loss, _, _ = blstm_cpu.run_batch(p, BLSTM_SUB, (xs, il, ys, ol.clone()), ODIM)
    loss.backward()
    for n in sds:
        ref = p[n].grad
        if float(ref.norm()) > 1e-6:
            rel = float((got[n].cpu() - ref).norm() / ref.norm())
            assert rel < 0.15, (n, rel)
    gn = float(torch.cat([got[n].reshape(-1) for n in sds]).double().norm())
    assert abs(gn / float(g[f"{tag}/grad_norm"]) - 1) < 3e-2


def test_gradients_vs_oracle_per_tensor(sd):
    """every parameter gradient against the CPU oracle (fp32): cosine and norm per tensor"""
    xs, il, ys, ol = synth_batch(21, [61, 50, 38, 30], [7, 5, 4, 3])
    eng = BlstmEngine(BLSTM_TINY, ODIM)
    eng.load_state_dict(sd)
    eng.run_batch(xs, il, ys, ol, train=True)
    got = eng.state_dict(flat=eng.grads)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss, _, _ = blstm_cpu.run_batch(p, BLSTM_TINY, (xs, il, ys, ol), ODIM)
    loss.backward()
    a = torch.cat([got[n].cpu().reshape(-1) for n in sd]).double(); b = torch.cat([p[n].grad.reshape(-1) for n in sd]).double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print(f"global gradient cosine vs oracle {cos:.5f}, norm ratio {float(a.norm() / b.norm()):.4f}")
    assert cos > 0.995
    for n in sd:
        x, y = got[n].cpu().double().reshape(-1), p[n].grad.double().reshape(-1)
        c = float((x * y).sum() / (x.norm() * y.norm() + 1e-30))
        assert c > 0.97, (n, c)


def test_sgd_steps_reduce_the_loss(sd):
    xs, il, ys, ol = synth_batch(21, [61, 50, 38, 30], [7, 5, 4, 3])
    eng = BlstmEngine(BLSTM_TINY, ODIM)
    eng.load_state_dict(sd)
    mom = torch.zeros_like(eng.params)
    losses = []
    for i in range(6):
        eng.run_batch(xs, il, ys, ol, train=True)
        eng.clip_sgd_step(mom, 5.0, 0.01, 0.9, True, first_step=(i == 0))
        losses.append(eng.read_stats()["loss"])
    print("CTC loss over 6 SGD steps on one batch:", [round(l, 4) for l in losses])
    assert losses[-1] < losses[0] and all(np.isfinite(l) for l in losses)


def test_mono_blstm_training_matches_reference_golden(golden_dir, tmp_path, monkeypatch):
    """BASELINE configs[0] (train.py mono-accent, config/blstm CTC, toy memmap shard) through
    get_trainer(MonoASRInterface, ...) of blstm_trainer: the 26 batches of two epochs in the reference's order, per-step
    CTC losses (incl. the batches whose targets do not fit the 4x-subsampled input: zero_infinity -> 0), files."""
    import random
    from functools import partial
    from types import SimpleNamespace
    from masr_amd.blstm_trainer import get_trainer
    from masr_amd.mono_interface import MonoASRInterface
    from oracle.make_goldens import write_toy_shard
    g = np.load(golden_dir / "blstm_mono_toy.npz")
    monkeypatch.chdir(tmp_path)
    data = tmp_path / "data"
    data.mkdir()
    for ai, a in enumerate(["african", "australia"]):
        write_toy_shard(data, a, "train", 16, seed=100 + ai)
        write_toy_shard(data, a, "dev", 4, seed=200 + ai)
    (data / "units.txt").write_text("".join(f"u{i} {i}\n" for i in range(1, 366)))
    cfg = {"asr_model": dict(BLSTM_TINY),
           "solver": {"setting": "gold", "data_root": str(data), "total_steps": 10, "total_epochs": 2, "spm_mapping": str(data / "units.txt"),
                      "spm_model": "unused", "label_smoothing": 0.2, "eval_ival": 1000, "log_ival": 1000, "save_ival": 3, "batch_size": 4,
                      "dev_batch_size": 4, "min_ilen": 10, "max_ilen": 50, "dev_max_ilen": 3000, "half_batch_ilen": 30}}
    paras = SimpleNamespace(accent="af", algo="no", model_name="blstm", eval_suffix="e", runs=0, overwrite=True, seed=531, resume=False,
                            use_tensorboard=False, save_verbose=False, split_rate=1.0, freeze_layer=None, pretrain=False,
                            pretrain_suffix=None, pretrain_setting=None, pretrain_runs=0, pretrain_step=0, pretrain_tgt_accent="ca",
                            pretrain_model_path=None, njobs=0, is_bucket=True, is_memmap=True, device="cuda:0", eval_every_epoch=False)
    random.seed(531); np.random.seed(531); torch.manual_seed(531)
    s = get_trainer(MonoASRInterface, cfg, paras, {"af": "african", "au": "australia", "ca": "canada"})
    s.load_data(); s.set_model()
    s.asr_model.load_state_dict(blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11))
    s.evaluate = lambda: None
    rec = []
    orig = s.run_batch

    def spy(cur_b, x, ilens, ys, olens, train, accent_idx=None):
        info = orig(cur_b, x, ilens, ys, olens, train=train)
        rec.append((int(cur_b), ilens.clone(), [y.clone() for y in ys], dict(info)))
        return info
    s._train = partial(spy, train=True)
    s.exec()
    assert len(rec) == int(g["n_calls"]) == 26 and s.global_step == int(g["global_step"]) and s.ep == int(g["ep"])
    worst = 0.0
    for i, (idx, il, ys, info) in enumerate(rec):
        assert idx == int(g[f"call{i}/accent"])
        np.testing.assert_array_equal(il.numpy(), g[f"call{i}/ilens"])
        np.testing.assert_array_equal(np.concatenate([y.numpy() for y in ys]), g[f"call{i}/ys"])
        ref = float(g[f"call{i}/loss"])
        if ref == 0.0:
            assert info["loss"] == 0.0                                # every target of the batch is infeasible: zero_infinity
            continue
        rel = abs(info["loss"] - ref) / ref
        worst = max(worst, rel)
        assert rel < (1e-3 if i == 0 else 5e-3), (i, info["loss"], ref)             # measured worst 1.1e-3 after 25 SGD steps
    print(f"BLSTM mono run: worst per-batch CTC loss rel err over 26 steps {worst:.2e}; last {rec[-1][3]['loss']:.4f} vs {float(g['call25/loss']):.4f}")
    got = s.asr_model.engine.state_dict()
    sd0 = blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11)
    for n in ("head.bias", "encoder.blstm.bt1.bias", "encoder.blstm.rnn0.bias_hh_l0_reverse", "encoder.vgg.7.bias"):
        du = (got[n].cpu() - sd0[n]).double(); dr = (torch.from_numpy(g[f"param/{n}"]) - sd0[n]).double()
        cos = float((du * dr).sum() / (du.norm() * dr.norm()))
        print(f"   {n}: update cosine vs reference {cos:.4f}, norm ratio {float(du.norm() / dr.norm()):.4f}")
        assert cos > 0.9
    assert {p.name for p in s.log_dir.iterdir()} >= set(g["files"].tolist()) - {"exp_key"}


def test_shipped_geometry_runs_deterministically():
    """config/blstm/mono-test.yaml widths (enc_dim = proj_dim = odim = 360 -> recurrent operands padded to 384, 3 layers,
    idim 83) on a ragged batch of 8 x 400 frames: finite loss / gradients, bit-identical repeat, loss falls under SGD."""
    import time
    cfg = {"encoder": {"idim": 83, "enc_dim": 360, "proj_dim": 360, "odim": 360, "sample_rate": "1_1_1", "dropout": "0_0_0"}}
    torch.manual_seed(531)
    from masr_amd.blstm_engine import reference_init_state_dict
    eng = BlstmEngine(cfg, ODIM)
    eng.load_state_dict(reference_init_state_dict(cfg, ODIM))
    B, T = 8, 400
    g = torch.Generator().manual_seed(4)
    ilens = torch.tensor([400, 400, 388, 371, 350, 333, 300, 257])
    xs = torch.randn(B, T, 83, generator=g)
    for b in range(B):
        xs[b, ilens[b]:] = 0
    olens = torch.tensor([25, 30, 12, 18, 9, 22, 15, 7])
    ys = [torch.randint(1, 366, (int(n),), generator=g) for n in olens]
    eng.run_batch(xs, ilens, ys, olens, train=True)
    s1, g1 = eng.read_stats(), eng.grads.clone()
    eng.run_batch(xs, ilens, ys, olens, train=True)
    s2, g2 = eng.read_stats(), eng.grads.clone()
    assert s1["loss"] == s2["loss"] and torch.equal(g1, g2)
    assert np.isfinite(s1["loss"]) and bool(torch.isfinite(g1).all()) and float(g1.norm()) > 0
    mom = torch.zeros_like(eng.params)
    losses = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        eng.run_batch(xs, ilens, ys, olens, train=True)
        eng.clip_sgd_step(mom, 5.0, 0.01, 0.9, True, first_step=(i == 0))
        losses.append(eng.read_stats()["loss"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"shipped BLSTM geometry, B=8 x 400 frames: {dt * 1e3:.1f} ms per training step; CTC loss {[round(l, 3) for l in losses]}")
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("B,T,ilens", [(8, 160, [160, 160, 148, 131, 110, 93, 60, 17]), (20, 96, None), (3, 40, [40, 9, 33])])
def test_resident_recurrence_equals_per_step_launches(B, T, ilens):
    """csrc/lstm_rec.hip: the recurrence of a layer as one launch per pass (workgroups resident over the sequence, W_hh in registers, h_t / dz_t
    exchanged as self-flagging granules) against lstm.hip's launch per timestep (masr_blstm_set_resident_recurrence(0)), at the shipped widths
    (enc_dim 360 -> 12 workgroups per direction, the last one with 8 units): ragged lengths incl. a sequence shorter than one pooled frame
    pair, a batch over 16 rows (two MFMA row tiles), a tiny one.  Same fp32 formulas; the recurrent product is one MFMA chain per tile instead
    of four partial chains, and h_t travels as bf16 either way (a last-bit difference in fp32 can move that rounding), so: logits to 1e-2 of
    their range (0.2 at these initial weights), loss to 2e-4, every gradient tensor cos > 0.9995 and 1 % in norm.  Repeating the
    resident step gives the same bits (nothing in it depends on which workgroup arrives first)."""
    from masr_amd.blstm_engine import reference_init_state_dict
    cfg = {"encoder": {"idim": 83, "enc_dim": 360, "proj_dim": 360, "odim": 360, "sample_rate": "1_1_1", "dropout": "0_0_0"}}
    torch.manual_seed(531)
    sd = reference_init_state_dict(cfg, ODIM)
    g = torch.Generator().manual_seed(B + T)
    il = torch.tensor(ilens) if ilens is not None else torch.tensor(sorted([int(x) for x in torch.randint(T // 3, T + 1, (B,), generator=g)], reverse=True))
    il[0] = T
    xs = torch.randn(B, T, 83, generator=g)
    for b in range(B):
        xs[b, il[b]:] = 0
    olens = torch.tensor([max(1, min(int(n) // 8 - 1, 12)) for n in il])
    ys = [torch.randint(1, 366, (int(n),), generator=g) for n in olens]
    outs = []
    for resident in (True, True, False):
        eng = BlstmEngine(cfg, ODIM)
        eng.load_state_dict(sd)
        eng.set_resident_recurrence(resident)
        eng.run_batch(xs, il, ys, olens, train=True)
        st = eng.read_stats()
        lg, el = eng.last_logits()
        outs.append((st["loss"], lg.clone(), el.clone(), eng.grads.clone(), eng))
    (l0, lg0, el0, g0, eng), (l1, lg1, _, g1, _), (l2, lg2, el2, g2, _) = outs
    assert l0 == l1 and torch.equal(lg0, lg1) and torch.equal(g0, g1)
    assert torch.equal(el0, el2) and np.isfinite(l0)
    assert abs(l0 - l2) <= 2e-4 * abs(l2), (l0, l2)
    assert float((lg0 - lg2).abs().max()) <= 1e-2 * float(lg2.abs().max())
    worst = 1.0
    for n, (off, shape) in eng.table.items():
        k = int(np.prod(shape))
        a, b = g0[off:off + k].double(), g2[off:off + k].double()
        if float(b.norm()) > 0:
            worst = min(worst, float((a * b).sum() / (a.norm() * b.norm())))
            assert abs(float(a.norm() / b.norm()) - 1) < 1e-2, n
    print(f"B={B} T={T}: resident vs per-step: loss {l0:.6f} / {l2:.6f}, max logit difference {float((lg0 - lg2).abs().max()):.2e}, worst gradient cos {worst:.6f}")
    assert worst > 0.9995


def test_resident_recurrence_times_out_loudly():
    """A workgroup of the resident recurrence that never publishes its part of h_t (fault injection: it leaves at start) must not hang the
    step: its peers give up after a bounded number of polls, the launch ends, and masr_blstm_read_stats reports the step as failed.  The
    next step (all workgroups present) runs and gives the numbers of an undisturbed engine."""
    import time
    from masr_amd import _cabi
    sd = blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11)
    xs, il, ys, ol = synth_batch(5, [64, 52, 40], [6, 4, 3])
    ref = BlstmEngine(BLSTM_TINY, ODIM)
    ref.load_state_dict(sd)
    ref.run_batch(xs, il, ys, ol, train=True)
    want = ref.read_stats()
    eng = BlstmEngine(BLSTM_TINY, ODIM)
    eng.load_state_dict(sd)
    L = _cabi.lib()
    L.masr_test_blstm_stall(eng.h, 1)
    try:
        t0 = time.perf_counter()
        eng.run_batch(xs, il, ys, ol, train=True)
        with pytest.raises(_cabi.MasrError, match="timed out"):
            eng.read_stats()
        dt = time.perf_counter() - t0
        # the forward-only path (Tester / BlstmEngine.forward) reads no stats: masr_blstm_check reports the time-out there
        with pytest.raises(_cabi.MasrError, match="timed out"):
            eng.forward(xs, il)
    finally:
        L.masr_test_blstm_stall(eng.h, 0)
    print(f"stalled step reported after {dt:.1f} s")
    assert dt < 60                                               # one bounded wait per step: the launches behind the first time-out return at once
    lg, _ = eng.forward(xs, il)                                  # healthy again: the mark was cleared by the check
    assert bool(torch.isfinite(lg).all())
    eng.run_batch(xs, il, ys, ol, train=True)
    got = eng.read_stats()
    assert got["loss"] == want["loss"] and got["grad_norm"] == want["grad_norm"]


def test_tester_best_hyp_matches_reference(golden_dir, tmp_path, monkeypatch):
    """train.py --test --model_name blstm: the best-hyp file of the reference's Tester (arg-max over all frames, trim,
    collapse repeats, drop blanks) for the deterministic tiny model, line by line."""
    from types import SimpleNamespace
    from masr_amd.tester import Tester
    from oracle.make_goldens import write_toy_shard
    monkeypatch.chdir(tmp_path)
    data = tmp_path / "data"
    data.mkdir()
    write_toy_shard(data, "african", "test", 6, seed=300)
    (data / "units.txt").write_text("".join(f"u{i} {i}\n" for i in range(1, 366)))
    cfg = {"asr_model": dict(BLSTM_TINY), "solver": {"setting": "gold", "data_root": str(data), "spm_mapping": str(data / "units.txt"),
                                                     "spm_model": "unused", "beam_decode": {"beam_size": 1}}}
    log_dir = tmp_path / "testing-logs" / "evaluation" / "gold" / "no" / "ev" / "ev" / "african" / "0"
    log_dir.mkdir(parents=True)
    torch.save(blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11), log_dir / "model.wer.best")
    paras = SimpleNamespace(accent="af", algo="no", pretrain_suffix=None, eval_suffix="ev", runs=0, model_name="blstm", test_model="model.wer.best",
                            decode_suffix="greedy_decode", decode_mode="greedy", decode_batch_size=4, njobs=1, resume=False, overwrite=True,
                            is_memmap=True, device="cuda:0")
    t = Tester(cfg, paras, {"af": "african"})
    t.load_data(); t.set_model(); t.exec()
    lines = (log_dir / "greedy_decode" / "best-hyp").read_text().splitlines()
    gold = np.load(golden_dir / "blstm_tester_toy.npz")["lines"].tolist()
    print("ours :", lines[:2]); print("ref  :", gold[:2])
    same = sum(a.rstrip() == b.rstrip() for a, b in zip(lines, gold))
    assert len(lines) == len(gold) and same >= len(gold) - 1, (lines, gold)
