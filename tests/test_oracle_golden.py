"""The CPU oracle (oracle/ref_cpu.py) against vectors captured from the real reference
(oracle/make_goldens.py).  CPU only; pins the oracle (prompt section 3 / SURVEY 8c)."""
import math
import random
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import ref_cpu
from oracle.make_goldens import TINY, ODIM, synth_batch, flat_checks

torch.set_num_threads(4)


def _fp_close(a, b, rtol=2e-4, atol=1e-6):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.abs(b[2]), 1e-12)            # l2 norm of the tensor
    assert np.all(np.abs(a - b) <= rtol * np.abs(b) + atol + 1e-5 * scale), (a, b)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "model_tiny.npz")


@pytest.fixture(scope="module")
def sd():
    return ref_cpu.deterministic_state_dict(TINY, ODIM, seed=7)


CASES = {"ragged": ([64, 52, 40, 33], [9, 7, 5, 3]), "same": ([48, 48, 48], [6, 6, 4]), "single": ([37], [5])}


@pytest.mark.parametrize("cname", list(CASES))
@pytest.mark.parametrize("eps", [0.2, 0.0])
def test_forward_loss_grads(G, sd, cname, eps):
    ilens, olens = CASES[cname]
    batch = synth_batch(11, ilens, olens)
    p = ref_cpu.leafify(sd, TINY)
    info, grads, logit, gold = ref_cpu.run_batch_train(p, TINY, batch, eps)
    key = f"{cname}_eps{eps}"
    np.testing.assert_allclose(logit.numpy(), G[f"{key}/logit"], rtol=1e-4, atol=2e-5)
    np.testing.assert_array_equal(gold.numpy(), G[f"{key}/gold"])
    assert abs(info["loss"] - float(G[f"{key}/loss"])) <= 1e-5 * abs(float(G[f"{key}/loss"]))
    assert info["acc"] == float(G[f"{key}/acc"])
    for n, g in grads.items():
        _fp_close(flat_checks(g), G[f"{key}/gradfp/{n}"])
    for k in G.files:
        if k.startswith(f"{key}/grad/"):
            n = k[len(f"{key}/grad/"):]
            ref = G[k]
            np.testing.assert_allclose(grads[n].numpy(), ref, rtol=1e-3, atol=1e-5 * np.abs(ref).max() + 1e-9)


def test_inner_steps(G, sd):
    """two inner steps: run_batch -> clip 5 -> SGD(momentum .9, nesterov) with lr x1000."""
    ilens, olens = CASES["ragged"]
    p = ref_cpu.leafify(sd, TINY)
    bufs = {}
    lr = ref_cpu.inner_lr(TINY) * 1000
    i0 = ref_cpu.inner_step(p, TINY, synth_batch(11, ilens, olens), 0.2, bufs, lr)
    i1 = ref_cpu.inner_step(p, TINY, synth_batch(12, ilens, olens), 0.2, bufs, lr)
    assert abs(i0["grad_norm"] - float(G["inner/gradnorm0"])) <= 1e-4 * float(G["inner/gradnorm0"])
    assert abs(i1["grad_norm"] - float(G["inner/gradnorm1"])) <= 1e-4 * float(G["inner/gradnorm1"])
    assert abs(i1["loss"] - float(G["inner/loss1"])) <= 1e-5 * float(G["inner/loss1"])
    for n in ref_cpu.grad_param_names(p, TINY):
        _fp_close(flat_checks(p[n]), G[f"inner/paramfp/{n}"], rtol=1e-5)
    np.testing.assert_allclose(p["vgg2enc.bias"].detach().numpy(), G["inner/param/vgg2enc.bias"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(p["char_trans.weight"].detach().numpy(), G["inner/param/char_trans.weight"], rtol=1e-5, atol=5e-7)


def test_recog_and_eval(G, sd):
    ilens, olens = CASES["ragged"]
    xs, il, ys, ol = synth_batch(11, ilens, olens)
    with torch.no_grad():
        hyp = ref_cpu.recog_greedy(sd, TINY, xs, il)
        logit, _ = ref_cpu.model_forward(sd, TINY, xs, il, ys, ol.clone())
    np.testing.assert_array_equal(hyp.numpy(), G["recog/hyp"])
    np.testing.assert_allclose(logit.numpy(), G["eval/logit"], rtol=1e-4, atol=2e-5)


def test_olens_mutated_in_place(sd):
    """quirk Q6: preprocess does olens += 1 on the caller's tensor."""
    xs, il, ys, ol = synth_batch(11, [37], [5])
    with torch.no_grad():
        ref_cpu.model_forward(sd, TINY, xs, il, ys, ol)
    assert ol.tolist() == [6]


def test_masks_noam_adam(golden_dir):
    g = np.load(golden_dir / "masks_noam.npz")
    np.testing.assert_array_equal(ref_cpu.make_bool_pad_mask(torch.tensor([7, 3, 5])).numpy(), g["pad_mask"])
    np.testing.assert_array_equal(ref_cpu.generate_square_subsequent_mask(5).numpy(), g["causal5"])
    lrs = [ref_cpu.noam_lr(t, 1.0, 512, 25000) for t in range(1, 13)]
    np.testing.assert_allclose(lrs, g["noam_lr_512_25000"], rtol=1e-12)
    lrs = [ref_cpu.noam_lr(t, 0.7, 64, 4) for t in range(1, 13)]
    np.testing.assert_allclose(lrs, g["noam_lr_64_4_k0.7"], rtol=1e-12)
    w = {"w": torch.from_numpy(g["adam_w0"].copy())}
    st = {}
    for t, gr in enumerate(g["adam_grads"], 1):
        ref_cpu.adam_step(w, {"w": torch.from_numpy(gr.copy())}, st, ref_cpu.noam_lr(t, 1.0, 64, 4))
    np.testing.assert_allclose(w["w"].numpy(), g["adam_w5"], rtol=1e-6, atol=1e-7)


def test_bucket_sampler(golden_dir):
    g = np.load(golden_dir / "bucket_sampler.npz")
    random.seed(531)
    np.random.seed(531)
    plan = ref_cpu.bucket_sampler_plan(g["ilens"], 10, 50, 30, 4)
    for ep in range(2):
        batches = list(ref_cpu.bucket_sampler_epoch(plan))
        np.testing.assert_array_equal(np.array([i for b in batches for i in b]), g[f"epoch{ep}_flat"])
        np.testing.assert_array_equal(np.array([len(b) for b in batches]), g[f"epoch{ep}_sizes"])
    assert len(batches) == int(g["len"])
    # every batch holds utterances of one length (bucket_size 1) and ilen in (min_ilen, max_ilen-2]
    il = g["ilens"]
    for b in batches:
        assert len({int(il[i]) for i in b}) == 1
        assert 10 < il[b[0]] <= 48


def test_ctc(golden_dir):
    g = np.load(golden_dir / "ctc.npz")
    for pre in ("", "inf_"):
        logits = torch.from_numpy(g[pre + "logits"])
        lp = torch.log_softmax(logits, -1)
        loss, grad_lp = ref_cpu.ctc_loss_np(lp.numpy(), g[pre + "targets"], g[pre + "il"], g[pre + "tl"])
        assert abs(loss - float(g[pre + "loss"])) <= 1e-6 * max(1.0, abs(float(g[pre + "loss"])))
        glp = torch.from_numpy(grad_lp).float()
        grad_logits = glp - lp.exp() * glp.sum(-1, keepdim=True)       # log_softmax backward
        np.testing.assert_allclose(grad_logits.numpy(), g[pre + "grad_logits"], rtol=1e-4, atol=1e-6)


def test_fomaml_toy(golden_dir, tmp_path):
    """Replay of the reference's full FOMAML run (2 accents, meta_k=2, 2 meta-steps): same toy
    shards, same RNG streams -> same batches -> same losses and same final meta weights."""
    from oracle.make_goldens import write_toy_shard
    g = np.load(golden_dir / "fomaml_toy.npz")
    cfg = dict(TINY)
    cfg["meta"] = {"optimizer_opt": {"k": 1.0, "warmup_steps": 4}}
    accents = ["african", "australia"]
    shards = []
    for ai, a in enumerate(accents):
        write_toy_shard(tmp_path, a, "train", 16, seed=100 + ai)
        d = tmp_path / a / "train"
        feat = np.load(d / "feat.dat", mmap_mode="r")
        ilens, olens, label = np.load(d / "ilens.npy"), np.load(d / "olens.npy"), np.load(d / "label.npy")
        iptr = np.concatenate([[0], np.cumsum(ilens)])
        optr = np.concatenate([[0], np.cumsum(olens)])
        shards.append((feat, iptr, label, optr, ilens, olens))
    random.seed(531)
    np.random.seed(531)

    def new_iter(ai):
        feat, iptr, label, optr, ilens, olens = shards[ai]
        plan = ref_cpu.bucket_sampler_plan(ilens, 10, 50, 30, 4)
        for idxs in ref_cpu.bucket_sampler_epoch(plan):
            yield ref_cpu.collate(feat, iptr, label, optr, ilens, olens, idxs)

    # DataContainer.__init__ builds train iterators (and samplers) in accent order (dataset.py:223-235)
    its = []
    for ai in range(2):
        feat, iptr, label, optr, ilens, olens = shards[ai]
        plan = ref_cpu.bucket_sampler_plan(ilens, 10, 50, 30, 4)     # consumes `random` at construction
        its.append([plan, None])

    def get_item(ai):
        plan, it = its[ai]
        feat, iptr, label, optr, ilens, olens = shards[ai]
        if it is None:
            it = iter(ref_cpu.bucket_sampler_epoch(plan))            # np.random consumed lazily at first next()
            its[ai][1] = it
        try:
            idxs = next(it)
        except StopIteration:
            plan = ref_cpu.bucket_sampler_plan(ilens, 10, 50, 30, 4)
            it = iter(ref_cpu.bucket_sampler_epoch(plan))
            its[ai] = [plan, it]
            idxs = next(it)
        return ref_cpu.collate(feat, iptr, label, optr, ilens, olens, idxs)

    meta = OrderedDict((n, t.clone()) for n, t in ref_cpu.deterministic_state_dict(cfg, ODIM, 7).items())
    if cfg["tgt_share_weight"]:
        meta["pre_embed.weight"] = meta["char_trans.weight"]
    adam_state = {}
    task_ids = [0, 1]
    call = 0
    for step in (1, 2):
        random.shuffle(task_ids)
        tasks = []
        for ai in task_ids[:2]:
            tr = [get_item(ai) for _ in range(2)]
            val = get_item(ai)
            tasks.append((tr, val))
            for b in tr + [val]:
                assert int(g[f"call{call}/accent"]) == ai
                np.testing.assert_array_equal(b[1].numpy(), g[f"call{call}/ilens"])
                np.testing.assert_array_equal(np.concatenate([y.numpy() for y in b[2]]), g[f"call{call}/ys"])
                call += 1
        infos, lr = ref_cpu.fomaml_meta_step(meta, cfg, tasks, 0.2, adam_state, step)
        # val losses are the 3rd, 6th call of each meta step
        base = (step - 1) * 6
        for ti, info in enumerate(infos):
            ref = float(g[f"call{base + ti * 3 + 2}/loss"])
            assert abs(info["loss"] - ref) <= 2e-5 * ref
    assert call == int(g["n_calls"])
    assert abs(lr - float(g["meta/lr"])) <= 1e-12
    # Adam's update is +-lr * m/sqrt(v): a 1e-5 relative wobble in a near-cancelling conv gradient moves the
    # update by ~1e-4*lr, so weights are compared at 2% of the step size (lr), i.e. far below 1e-3 relative.
    tol = 0.02 * lr
    for n in ref_cpu.grad_param_names(meta, cfg):
        a, b = flat_checks(meta[n]), g[f"meta/fp/{n}"]
        # the key third of every in_proj_bias has an exactly-zero true gradient (softmax is invariant to a
        # per-query shift), so Adam turns pure rounding noise into +-lr steps there: not comparable.
        if not n.endswith("in_proj_bias"):
            assert abs(a[2] - b[2]) <= 1e-3 * b[2] + 1e-9, n             # l2 norm
        assert np.all(np.abs(a[3:] - b[3:]) <= tol), n                   # first/last 4 values
    for n in ("vgg2enc.bias", "char_trans.bias", "decoder.norm.weight"):
        assert np.abs(meta[n].numpy() - g[f"meta/param/{n}"]).max() <= tol, n
    assert abs(ref_cpu.inner_lr(cfg) - float(g["inner_lr"])) < 1e-15


def test_fbank_oracle_properties():
    """oracle/fbank_np.py has no reference fixture to pin it (the reference ships no extraction code): check the
    published algorithm's invariants instead -- frame count, scale law, DC invariance, tone localisation."""
    from oracle import fbank_np as F
    assert [F.num_frames(n) for n in (0, 399, 400, 559, 560, 16000)] == [0, 0, 1, 1, 2, 98]
    banks = F.mel_banks(80)
    assert banks.shape == (80, 256) and banks.min() >= 0 and banks.max() <= 1 and (banks.sum(1) > 0).all()
    t = np.arange(8000) / 16000.0
    tone = 5000 * np.sin(2 * np.pi * 1000.0 * t)
    f = F.fbank(tone, 80)
    centre = F.mel(F.LOW) + (np.arange(80) + 1) * (F.mel(8000.0) - F.mel(F.LOW)) / 81
    assert abs(int(f.mean(0).argmax()) - int(np.abs(centre - F.mel(1000.0)).argmin())) <= 1      # energy sits at the 1 kHz filter
    np.testing.assert_allclose(F.fbank(2 * tone, 80), f + 2 * np.log(2.0), atol=1e-9)            # power scales with amplitude^2
    np.testing.assert_allclose(F.fbank(tone + 1234.0, 80), f, atol=1e-6)                         # remove_dc_offset


def _voiced(f0_fn, dur, sr=16000, amp=3000.0, harmonics=7):
    t = np.arange(int(dur * sr)) / sr
    ph = 2 * np.pi * np.cumsum(f0_fn(t)) / sr
    return amp * sum(np.sin(k * ph) / k for k in range(1, harmonics + 1))


def test_pitch_oracle_properties():
    """oracle/pitch_np.py restates Kaldi's compute-kaldi-pitch-feats | process-kaldi-pitch-feats (the 3 pitch dims of the shipped
    83-dim rows); nothing in the container can pin it (no Kaldi, torchaudio, ... -- parity unpinned), so the published algorithm's
    invariants are checked: lag grid, frame count against the fbank stream (the recipe's paste-feats tolerance), tracking of a known
    f0 glide, voicing separation, gain invariance, the shape of the three processed dims."""
    from oracle import fbank_np as F
    from oracle import pitch_np as P
    lg = P.lags()
    assert len(lg) == 417 and abs(lg[0] - 1 / 400) < 1e-12 and lg[-1] <= 1 / 50 < lg[-1] * 1.005
    assert (P.OUTER_MIN_LAG, P.OUTER_MAX_LAG, P.NLAG_IN, P.FULL) == (8, 82, 75, 182)
    for n in (0, 724, 725, 885, 16000, 160000):
        tp, tf = P.num_frames(n), F.num_frames(n)
        assert tp <= tf and (tf - tp <= 2 or tp == 0), (n, tp, tf)
    assert [P.num_frames(n) for n in (724, 725, 884, 885, 160000)] == [0, 1, 1, 2, 996]
    W = P.upsample_matrix(lg)
    assert W.shape == (417, 75) and (np.count_nonzero(W, axis=1) <= 11).all()
    np.testing.assert_allclose(W.sum(1), 1.0, atol=0.02)                 # the windowed sinc interpolates a constant to (almost) itself
    # resampling: a 300 Hz tone passes the 1 kHz low-pass with its amplitude, a 3 kHz tone does not
    t = np.arange(8000) / 16000.0
    lo, hi = P.downsample(1000 * np.sin(2 * np.pi * 300 * t)), P.downsample(1000 * np.sin(2 * np.pi * 3000 * t))
    assert len(lo) == 2000 and 700 < np.abs(lo[100:-100]).max() < 1050 and np.abs(hi[100:-100]).max() < 60      # (one zero crossing: a soft filter, 0.78 at 300 Hz)
    # a glide 120 -> 180 Hz with harmonics, then noise
    wav = _voiced(lambda t: 120 + 40 * t, 1.5)
    wav[16000:] = np.random.RandomState(0).randn(len(wav) - 16000) * 800
    raw = P.compute_kaldi_pitch(wav)
    assert raw.shape == (P.num_frames(len(wav)), 2)
    centre = np.arange(len(raw)) * 0.01 + 0.0125 + 0.02                 # window centre + half the average lag span: within the tolerance below
    v = slice(5, 90)
    assert np.abs(raw[v, 1] - (120 + 40 * centre[v])).max() < 2.5       # Hz (the lag grid is 0.5 % wide)
    assert raw[v, 0].min() > 0.95 and np.abs(raw[108:, 0]).mean() < 0.3  # NCCF ~ 1 on voiced frames, small on noise
    np.testing.assert_allclose(P.compute_kaldi_pitch(0.1 * wav), raw, atol=1e-6)     # the ballast scales with the signal: gain invariant
    f3 = P.process_kaldi_pitch(raw)
    assert f3.shape == (len(raw), 3)
    assert f3[v, 0].max() < -1.0 and f3[110:, 0].min() > -0.4            # pov feature 2 ((1.0001 - nccf)^0.15 - 1): -> -1.5 voiced, -> 0 unvoiced
    slope = np.log(180 / 120) / 150 * 10.0                              # d log f0 per frame, times delta_pitch_scale
    assert np.abs(f3[10:80, 2] - slope).max() < 0.03
    const = P.process_kaldi_pitch(np.stack([np.full(50, 0.9), np.full(50, 200.0)], 1))
    assert np.abs(const[:, 1]).max() < 1e-12 and np.abs(const[:, 2]).max() < 1e-12      # constant pitch: zero normalised log pitch and delta
    rows = P.fbank_pitch(wav[:8000])
    assert rows.shape == (min(F.num_frames(8000), P.num_frames(8000)), 83)


@pytest.mark.parametrize("tag,ilens,olens", [("ragged", [61, 50, 38, 30], [7, 5, 4, 3]), ("single", [45], [6])])
def test_blstm_oracle_matches_reference(golden_dir, tag, ilens, olens):
    """oracle/blstm_cpu.py (explicit LSTM recurrence with own packed-sequence handling) vs the real MonoBLSTM +
    BLSTMTrainer.run_batch loss: logits, CTC loss, every parameter gradient."""
    from oracle import blstm_cpu
    from oracle.make_goldens import BLSTM_TINY, ODIM, flat_checks, synth_batch
    g = np.load(golden_dir / "blstm_tiny.npz")
    sd = blstm_cpu.deterministic_state_dict(BLSTM_TINY, ODIM, seed=11)
    assert list(sd.keys()) == g["state_dict_keys"].tolist()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xs, il, ys, ol = synth_batch(21, ilens, olens)
    loss, logits, lens = blstm_cpu.run_batch(p, BLSTM_TINY, (xs, il, ys, ol), ODIM)
    loss.backward()
    assert lens.tolist() == g[f"{tag}/enc_lens"].tolist()
    np.testing.assert_allclose(logits.detach().numpy(), g[f"{tag}/logits"], rtol=1e-4, atol=2e-5)
    assert abs(float(loss) - float(g[f"{tag}/loss"])) < 1e-5 * float(g[f"{tag}/loss"])
    for n in sd:
        np.testing.assert_allclose(flat_checks(p[n].grad), g[f"{tag}/grad/{n}"], rtol=2e-3, atol=1e-6, err_msg=n)
    np.testing.assert_allclose(p["head.bias"].grad.numpy(), g[f"{tag}/gradfull/head.bias"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(p["encoder.blstm.rnn0.weight_hh_l0_reverse"].grad.numpy(),
                               g[f"{tag}/gradfull/encoder.blstm.rnn0.weight_hh_l0_reverse"], rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("tag,ilens,olens", [("ragged", [118, 101, 77, 60], [4, 3, 2, 2]), ("single", [90], [3])])
def test_blstm_oracle_with_time_subsampling_matches_reference(golden_dir, tag, ilens, olens):
    """sample_rate 1_2_2 (RNNP.forward, src/modules/encoder.py:118-121: ys_pad[:, ::sub], enc_lens -> (len + 1) // sub): the oracle against
    the real MonoBLSTM -- output lengths, logits, CTC loss, every parameter gradient."""
    from oracle import blstm_cpu
    from oracle.make_goldens import BLSTM_SUB, ODIM, flat_checks, synth_batch
    g = np.load(golden_dir / "blstm_sub.npz")
    sd = blstm_cpu.deterministic_state_dict(BLSTM_SUB, ODIM, seed=12)
    assert list(sd.keys()) == g["state_dict_keys"].tolist()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xs, il, ys, ol = synth_batch(23, ilens, olens)
    loss, logits, lens = blstm_cpu.run_batch(p, BLSTM_SUB, (xs, il, ys, ol), ODIM)
    loss.backward()
    assert lens.tolist() == g[f"{tag}/enc_lens"].tolist()
    ref = g[f"{tag}/logits"]                                            # (the reference's time axis ends at the longest utterance's length)
    np.testing.assert_allclose(logits.detach().numpy()[:, :ref.shape[1]], ref, rtol=1e-4, atol=2e-5)
    assert abs(float(loss) - float(g[f"{tag}/loss"])) < 1e-5 * float(g[f"{tag}/loss"])
    for n in sd:
        np.testing.assert_allclose(flat_checks(p[n].grad), g[f"{tag}/grad/{n}"], rtol=2e-3, atol=1e-6, err_msg=n)
    for n in ("head.bias", "encoder.blstm.bt1.bias", "encoder.blstm.rnn2.weight_hh_l0_reverse"):
        np.testing.assert_allclose(p[n].grad.numpy(), g[f"{tag}/gradfull/{n}"], rtol=1e-3, atol=1e-7)


def test_fomaml_8acc_oracle_matches_reference(golden_dir, tmp_path, monkeypatch):
    """BASELINE configs[3] in one process (pretrain.py --algo fomaml, EIGHT accents, meta_batch_size 8, meta_k 1), run by the
    reference from its own seed-531 initialisation (tests/golden/fomaml_8acc.npz): the oracle's meta loop over the product's
    DataContainer, started from the product's init replay, reproduces it -- initial weights, same 16 tasks in the same order,
    every train / eval loss within 2e-5, every meta-gradient tensor, the meta weights after both Adam steps."""
    from replay import eight_accent_setup, oracle_fomaml_run
    g = np.load(golden_dir / "fomaml_8acc.npz")
    monkeypatch.chdir(tmp_path)
    cfg, _, dc, init = eight_accent_setup(tmp_path, golden_dir)
    for n, t in init.items():
        if f"init/fp/{n}" in g.files:
            _fp_close(flat_checks(t), g[f"init/fp/{n}"], rtol=1e-6)

    def on_batch(i, accent, train, batch):
        assert accent == int(g[f"call{i}/accent"]) and int(train) == int(g[f"call{i}/train"]), i
        np.testing.assert_array_equal(batch[1].numpy(), g[f"call{i}/ilens"])
        np.testing.assert_array_equal(np.concatenate([y.numpy() for y in batch[2]]), g[f"call{i}/ys"])

    run = oracle_fomaml_run(cfg, dc, 1, 8, 3, cfg["solver"]["label_smoothing"], on_batch=on_batch, init_sd=init)
    assert len(run["calls"]) == int(g["n_calls"]) == 40 and len(run["steps"]) == int(g["n_meta_steps"]) == 2
    for i, (accent, train, info) in enumerate(run["calls"]):
        for k in ("loss", "acc"):
            ref = float(g[f"call{i}/{k}"])
            assert abs(info[k] - ref) <= 2e-5 * abs(ref) + 1e-12, (i, k, info[k], ref)
    assert abs(run["lr"] - float(g["meta/lr"])) <= 1e-15
    names = ref_cpu.grad_param_names(run["steps"][0][1], cfg["asr_model"])
    for si, (mg, meta) in enumerate(run["steps"]):
        for n in names:
            a, b = flat_checks(mg[n]), g[f"step{si}/metagrad/fp/{n}"]
            if not n.endswith("in_proj_bias"):
                assert abs(a[2] - b[2]) <= 1e-4 * b[2] + 1e-9, (si, n, a[2], b[2])
            wa, wb = flat_checks(meta[n]), g[f"step{si}/meta/fp/{n}"]
            # (the reference's in_proj_bias starts at exactly 0 and its key third has an exactly-zero true gradient: Adam turns that
            # third's rounding noise into +-lr steps, in the reference too -- one lr of slack on the norm for it)
            assert abs(wa[2] - wb[2]) <= 1e-6 * wb[2] + (run["lr"] if n.endswith("in_proj_bias") else 0.0), (si, n)
        for n in ("vgg2enc.bias", "decoder.norm.weight", "char_trans.bias", "feat_extractor.0.weight", "encoder.layers.1.linear2.bias"):
            ref = torch.from_numpy(g[f"step{si}/metagrad/full/{n}"])
            # 5e-4 (cfg3: 2e-4): the learnable shards' frames are one codebook vector + 10 % noise, so conv1's weight gradient sums
            # thousands of near-equal terms -- the fp32 summation order of torch's conv backward vs the oracle's shows (measured 2.1e-4)
            assert float((mg[n] - ref).norm()) <= 5e-4 * float(ref.norm()) + 1e-9, (si, n)


@pytest.mark.parametrize("meta_k", [1, 2])
def test_fomaml_cfg3_oracle_matches_reference(golden_dir, tmp_path, monkeypatch, meta_k):
    """BASELINE configs[2] literally (pretrain.py --algo fomaml, 4 accents, inner_steps = meta_k, shipped warmup 25000; plus
    a meta_k = 2 variant): the oracle's meta loop over the product's DataContainer reproduces the reference run captured in
    tests/golden/fomaml_cfg3.npz -- same batches in the same order, EVERY train and eval loss within 2e-5, every meta-gradient
    tensor of every meta-step (norm within 1e-4, first/last elements), the meta weights after every Adam step."""
    import masr_amd  # noqa: F401
    from masr_amd.io.dataset import DataContainer
    from oracle.make_goldens import cfg3_workspace
    from replay import oracle_fomaml_run
    g = np.load(golden_dir / "fomaml_cfg3.npz")
    pre = f"k{meta_k}/"
    cfg = cfg3_workspace(tmp_path, golden_dir)
    monkeypatch.chdir(tmp_path)
    sv = cfg["solver"]
    random.seed(531); np.random.seed(531); torch.manual_seed(531)
    dc = DataContainer([tmp_path / "data" / a for _, a in (("af", "african"), ("au", "australia"), ("en", "england"), ("us", "us"))],
                       batch_size=sv["batch_size"], dev_batch_size=sv["dev_batch_size"], is_memmap=True, is_bucket=True,
                       min_ilen=sv["min_ilen"], max_ilen=sv["max_ilen"], half_batch_ilen=sv["half_batch_ilen"])

    def on_batch(i, accent, train, batch):
        assert accent == int(g[f"{pre}call{i}/accent"]) and int(train) == int(g[f"{pre}call{i}/train"]), i
        np.testing.assert_array_equal(batch[1].numpy(), g[f"{pre}call{i}/ilens"])
        np.testing.assert_array_equal(np.concatenate([y.numpy() for y in batch[2]]), g[f"{pre}call{i}/ys"])
        _fp_close(flat_checks(batch[0]), g[f"{pre}call{i}/x_fp"], rtol=1e-6)

    run = oracle_fomaml_run(cfg, dc, meta_k, 4, 5, sv["label_smoothing"], on_batch=on_batch)
    assert len(run["calls"]) == int(g[pre + "n_calls"]) and len(run["steps"]) == int(g[pre + "n_meta_steps"]) == 4
    worst = 0.0
    for i, (accent, train, info) in enumerate(run["calls"]):
        for k in ("loss", "acc"):
            ref = float(g[f"{pre}call{i}/{k}"])
            worst = max(worst, abs(info[k] - ref) / max(abs(ref), 1e-9)) if k == "loss" else worst
            assert abs(info[k] - ref) <= 2e-5 * abs(ref) + 1e-12, (i, k, info[k], ref)
    print(f"meta_k {meta_k}: {len(run['calls'])} calls, worst relative loss error {worst:.2e}")
    assert abs(run["lr"] - float(g[pre + "meta/lr"])) <= 1e-15
    names = ref_cpu.grad_param_names(run["steps"][0][1], cfg["asr_model"])
    for si, (mg, meta) in enumerate(run["steps"]):
        for n in names:
            a, b = flat_checks(mg[n]), g[f"{pre}step{si}/metagrad/fp/{n}"]
            if not n.endswith("in_proj_bias"):               # (key third: exactly-zero true gradient, rounding noise only)
                assert abs(a[2] - b[2]) <= 1e-4 * b[2] + 1e-9, (si, n, a[2], b[2])
                assert np.all(np.abs(a[3:] - b[3:]) <= 2e-4 * np.abs(b[3:]) + 1e-6 * b[2]), (si, n)
            # meta weights after Adam: every element moved by at most lr; compare at a fraction of that step
            wa, wb = flat_checks(meta[n]), g[f"{pre}step{si}/meta/fp/{n}"]
            assert abs(wa[2] - wb[2]) <= 1e-6 * wb[2], (si, n)
        for n in ("vgg2enc.bias", "decoder.norm.weight", "char_trans.bias", "feat_extractor.0.weight", "encoder.layers.1.linear2.bias"):
            ref = g[f"{pre}step{si}/metagrad/full/{n}"]
            err = np.linalg.norm(mg[n].numpy() - ref) / np.linalg.norm(ref)
            assert err < 2e-4, (si, n, err)


def test_radam_restatement_coincides_with_torch_radam_where_the_conventions_agree():
    """oracle.ref_cpu.radam_torch_optimizer_step restates the un-vendored torch_optimizer.RAdam (parity unpinned against the package).
    Where its conventions and torch.optim.RAdam's describe the same algorithm -- no weight decay, eps -> 0 -- the two must produce
    the same trajectory through both the unrectified and the rectified phase; the known differences (eps placement, decay on the
    weight) must show up when switched on."""
    import torch
    from oracle import ref_cpu
    g = torch.Generator().manual_seed(5)
    n = 257
    p0 = torch.randn(n, generator=g, dtype=torch.float64)
    w = torch.nn.Parameter(p0.clone())
    ref = torch.optim.RAdam([w], lr=1e-2, betas=(0.9, 0.999), eps=1e-30)
    p = p0.clone()
    st = {"step": 0, "exp_avg": torch.zeros(n, dtype=torch.float64), "exp_avg_sq": torch.zeros(n, dtype=torch.float64)}
    for t in range(12):
        gr = torch.randn(n, generator=g, dtype=torch.float64)
        w.grad = gr.clone(); ref.step()
        ref_cpu.radam_torch_optimizer_step(p, gr, st, lr=1e-2, eps=1e-30)
        torch.testing.assert_close(p, w.detach(), rtol=1e-9, atol=1e-12, msg=f"step {t + 1}")
    q = p0.clone()
    st2 = {"step": 0, "exp_avg": torch.zeros(n, dtype=torch.float64), "exp_avg_sq": torch.zeros(n, dtype=torch.float64)}
    ref_cpu.radam_torch_optimizer_step(q, torch.ones(n, dtype=torch.float64), st2, lr=1e-2, weight_decay=0.5)
    # decay on the weight: p * (1 - lr * wd) - lr * g / (1 - b1) * (1 - b1)  (unrectified first step: exp_avg / bias correction = g)
    torch.testing.assert_close(q, p0 * (1 - 1e-2 * 0.5) - 1e-2, rtol=1e-12, atol=1e-14)


def test_oracle_matches_reference_at_the_headline_shape(golden_dir):
    """fometa-hkust geometry (24.88 M parameters), the reference's seed-531 initialisation, B = 16 x T = 1000 x idim 80: the oracle's
    loss, accuracy, global gradient norm and EVERY per-tensor gradient norm against the reference's (tests/golden/hkust_fullsize.npz,
    oracle/make_goldens.py::gen_hkust_fullsize_goldens); small gradients element-wise.  ~5 s."""
    import torch
    from masr_amd.model import reference_init_state_dict
    from oracle import ref_cpu
    from oracle.make_goldens import ODIM, fullsize_batch
    g = np.load(golden_dir / "hkust_fullsize.npz")
    cfg = {"idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.0, "pos_dropout": 0.0, "tgt_share_weight": 1,
           "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}}
    torch.manual_seed(531)
    sd = reference_init_state_dict(cfg, ODIM)
    sd["pos_encoder.pe"] = ref_cpu.sinusoid_pe(3000, 512)
    xs, il, ys, ol = fullsize_batch(80)
    p = ref_cpu.leafify(sd, cfg)
    info, grads, _, _ = ref_cpu.run_batch_train(p, cfg, (xs, il, ys, ol.clone()), 0.2)
    assert abs(info["loss"] - float(g["d80/loss"])) <= 2e-6 * float(g["d80/loss"])
    assert abs(info["acc"] - float(g["d80/acc"])) <= 1.0 / int(g["d80/n_total"])
    tot = 0.0
    for n, gr in grads.items():
        fp = g[f"d80/gradfp/{n}"]
        assert abs(float(gr.double().norm()) - fp[2]) <= 1e-4 * fp[2] + 1e-9, n
        tot += float(gr.double().norm()) ** 2
    assert abs(tot ** 0.5 - float(g["d80/grad_norm"])) <= 1e-4 * float(g["d80/grad_norm"])
    for k in g.files:
        if k.startswith("d80/grad/"):
            n = k[len("d80/grad/"):]
            ref = torch.from_numpy(g[k])
            assert float((grads[n] - ref).norm() / (ref.norm() + 1e-20)) < 2e-4, n
