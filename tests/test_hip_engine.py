"""End-to-end parity of the HIP engine against the CPU oracle (GPU).  Same deterministic weights,
same seeded batches; bf16 MFMA operands vs the fp32 oracle -> tolerances stated per check
(north-star: meta-loss / CE within 1e-3 relative)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402
from masr_amd.engine import MasrEngine  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.make_goldens import TINY, ODIM, synth_batch  # noqa: E402

CASES = {"ragged": ([64, 52, 40, 33], [9, 7, 5, 3]), "same": ([48, 48, 48], [6, 6, 4]), "single": ([37], [5])}


def rel_l2(a, b):
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.fixture(scope="module")
def sd():
    return ref_cpu.deterministic_state_dict(TINY, ODIM, seed=7)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(golden_dir / "model_tiny.npz")


@pytest.mark.parametrize("cname", list(CASES))
@pytest.mark.parametrize("eps", [0.2, 0.0])
def test_run_batch_vs_oracle_and_golden(sd, G, cname, eps):
    ilens, olens = CASES[cname]
    xs, il, ys, ol = synth_batch(11, ilens, olens)
    eng = MasrEngine(TINY, ODIM, label_smoothing=eps)
    eng.load_state_dict(sd)
    eng.run_batch(xs, il, ys, ol, train=True)
    st = eng.read_stats()
    # oracle
    p = ref_cpu.leafify(sd, TINY)
    info, grads, logit, gold = ref_cpu.run_batch_train(p, TINY, (xs, il, ys, ol.clone()), eps)
    key = f"{cname}_eps{eps}"
    ref_loss = float(G[f"{key}/loss"])                      # the REAL reference's loss
    assert abs(info["loss"] - ref_loss) <= 1e-5 * ref_loss
    assert abs(st["loss"] - ref_loss) <= 1e-3 * ref_loss, (st["loss"], ref_loss)      # north-star tolerance
    assert st["n_total"] == sum(olens) + len(olens)
    lg, gd = eng.last_logits()
    np.testing.assert_array_equal(gd.cpu().numpy(), G[f"{key}/gold"])
    assert rel_l2(lg.cpu(), torch.from_numpy(G[f"{key}/logit"])) < 2e-2
    # gradients vs the fp32 oracle: bf16 operands flip ~0.3 % of the ReLU / max-pool decisions, which alone moves
    # gradients by 3-10 % (the oracle shows the same drift when IT is run with bf16 emulation), so this check is loose
    g_all = eng.state_dict(flat=eng.grads)
    names = ref_cpu.grad_param_names(p, TINY)

    def cmp(a, b, n):
        a, b = a.cpu(), b
        if n.endswith("in_proj_bias"):
            E = TINY["d_model"]                              # key-bias third has zero true gradient
            a = torch.cat([a[:E], a[2 * E:]]); b = torch.cat([b[:E], b[2 * E:]])
        return rel_l2(a, b)
    worst32 = max(cmp(g_all[n], grads[n], n) for n in names)
    assert worst32 < 0.15, worst32
    # ... and tight against the oracle with the same bf16 rounding points (same ReLU / pool decisions)
    with ref_cpu.bf16_emulation():
        pq = ref_cpu.leafify(sd, TINY)
        infoq, gradsq, logitq, _ = ref_cpu.run_batch_train(pq, TINY, (xs, il, ys, ol.clone()), eps)
    assert abs(st["loss"] - infoq["loss"]) <= 2e-4 * infoq["loss"], (st["loss"], infoq["loss"])
    assert rel_l2(lg.cpu(), logitq) < 4e-3
    worst = 0.0
    for n in names:
        r = cmp(g_all[n], gradsq[n], n)
        worst = max(worst, r)
        # per-tensor rel-l2 is dominated by the handful of ReLU units whose pre-activation sits within the ~1e-3
        # forward disagreement of two different bf16 implementations (each flipped unit moves a whole row/column
        # of a weight gradient): a few percent on this 30-token batch.  Structure errors show up as >> 10 %.
        assert r < 8e-2, (n, r)
    flat_a = torch.cat([g_all[n].cpu().reshape(-1) for n in names if not n.endswith("in_proj_bias")]).double()
    flat_b = torch.cat([gradsq[n].reshape(-1) for n in names if not n.endswith("in_proj_bias")]).double()
    cos = float((flat_a * flat_b).sum() / (flat_a.norm() * flat_b.norm()))
    assert cos > 0.999, cos
    assert abs(float(flat_a.norm()) - float(flat_b.norm())) < 1e-2 * float(flat_b.norm())
    print(f"{key}: loss {st['loss']:.6f} ref {ref_loss:.6f} bf16-oracle {infoq['loss']:.6f}; worst grad rel-l2 {worst:.4f} (fp32 oracle {worst32:.4f})")


def test_inner_steps_vs_oracle(sd, G):
    """two inner steps (run_batch -> clip 5 -> SGD momentum .9 nesterov), lr x1000 as in the golden"""
    ilens, olens = CASES["ragged"]
    eng = MasrEngine(TINY, ODIM, label_smoothing=0.2)
    eng.load_state_dict(sd)
    lr = ref_cpu.inner_lr(TINY) * 1000
    mom = torch.zeros_like(eng.params)
    losses, norms = [], []
    for i, seed in enumerate((11, 12)):
        xs, il, ys, ol = synth_batch(seed, ilens, olens)
        eng.run_batch(xs, il, ys, ol, train=True)
        eng.clip_sgd_step(mom, 5.0, lr, 0.9, True, first_step=(i == 0))
        st = eng.read_stats()
        losses.append(st["loss"]); norms.append(st["grad_norm"])
    assert abs(norms[0] - float(G["inner/gradnorm0"])) <= 2e-2 * float(G["inner/gradnorm0"])
    assert abs(losses[1] - float(G["inner/loss1"])) <= 1e-3 * float(G["inner/loss1"])
    ref = torch.from_numpy(G["inner/param/char_trans.weight"])
    got = eng.view("char_trans.weight").cpu()
    start = sd["char_trans.weight"]
    # compare the UPDATE (post - pre), which is what the step computed
    assert rel_l2(got - start, ref - start) < 5e-2


def test_eval_mode_no_backward(sd):
    eng = MasrEngine(TINY, ODIM, label_smoothing=0.2)
    eng.load_state_dict(sd)
    eng.grads.fill_(3.0)
    xs, il, ys, ol = synth_batch(11, *CASES["same"])
    eng.run_batch(xs, il, ys, ol, train=False)
    st = eng.read_stats()
    assert math.isfinite(st["loss"]) and torch.all(eng.grads == 3.0)


def test_state_dict_layout(sd):
    eng = MasrEngine(TINY, ODIM)
    eng.load_state_dict(sd)
    out = eng.state_dict()
    assert list(out.keys()) == list(sd.keys())
    for k in sd:
        assert out[k].shape == sd[k].shape
        torch.testing.assert_close(out[k].cpu(), sd[k], rtol=0, atol=1e-6 if k == "pos_encoder.pe" else 0)


def _flat_dot(eng, a, b):
    return float((a.double() * b.double()).sum())


def test_dropout_forward_backward_consistency(sd):
    """Dropout masks are regenerated (not stored) in the backward from a stateless hash of (seed, site, index).
    For a fixed seed the loss is a deterministic function of the weights, so the analytic gradient must match a
    central finite difference along the gradient direction -- this fails if any forward/backward mask pair disagrees."""
    cfg = dict(TINY); cfg["dropout"] = 0.2; cfg["pos_dropout"] = 0.1
    xs, il, ys, ol = synth_batch(11, *CASES["ragged"])
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
    eng.load_state_dict(sd)

    def loss_at(params, seed=77):
        eng.params.copy_(params); eng.mark_dirty(); eng.set_seed(seed)
        eng.run_batch(xs, il, ys, ol, train=True)
        return eng.read_stats()["loss"], eng.grads.clone()

    p0 = eng.params.clone()
    l0, g0 = loss_at(p0)
    l0b, g0b = loss_at(p0)
    assert l0 == l0b and torch.equal(g0, g0b), "same seed must reproduce the same masks"
    l_other, _ = loss_at(p0, seed=78)
    assert l_other != l0, "a different seed must draw different masks"
    eng.cfg_eval = None
    # eval mode ignores dropout
    eng.params.copy_(p0); eng.mark_dirty()
    eng.run_batch(xs, il, ys, ol, train=False)
    l_eval = eng.read_stats()["loss"]
    eng0 = MasrEngine(TINY, ODIM, label_smoothing=0.2); eng0.load_state_dict(sd)
    eng0.run_batch(xs, il, ys, ol, train=False)
    assert abs(l_eval - eng0.read_stats()["loss"]) < 1e-6
    assert abs(l0 - l_eval) < 0.15 * l_eval                      # dropout perturbs, it does not destroy, the loss
    # directional derivative along the gradient
    v = g0 / g0.norm()
    eps = 0.02
    lp, _ = loss_at(p0 + eps * v)
    lm, _ = loss_at(p0 - eps * v)
    fd = (lp - lm) / (2 * eps)
    an = float(g0.norm())
    assert abs(fd - an) < 0.08 * an, (fd, an)


def test_recog_matches_reference_tokens(sd, G):
    """greedy decode (masr_recog) vs the reference's MyTransformer.recog tokens (golden).  The tiny random model has
    near-flat logits, so bf16 rounding can flip an arg-max; once a token differs the rest of that utterance diverges
    (autoregressive), so agreement is checked on the leading tokens and against the bf16-emulated oracle."""
    ilens, olens = CASES["ragged"]
    xs, il, ys, ol = synth_batch(11, ilens, olens)
    eng = MasrEngine(TINY, ODIM)
    eng.load_state_dict(sd)
    hyp = eng.recog(xs, il).cpu()
    ref = torch.from_numpy(G["recog/hyp"])
    assert hyp.shape == ref.shape == (max(ilens) // 4, len(ilens))
    with ref_cpu.bf16_emulation(), torch.no_grad():
        hq = ref_cpu.recog_greedy(sd, TINY, xs, il)
    agree_ref = float((hyp == ref).float().mean())
    agree_q = float((hyp == hq).float().mean())
    first = int((hyp[0] == ref[0]).sum())
    print(f"recog agreement: reference {agree_ref:.3f}, bf16-emulated oracle {agree_q:.3f}, first tokens {first}/{len(ilens)}")
    assert first == len(ilens)
    assert agree_ref >= 0.9 and agree_q >= 0.9


def test_recog_tokens_exact_where_the_reference_arg_max_is_well_defined(sd, G):
    """An arg-max is an index: where the reference's own decision is well defined the tokens must be IDENTICAL, not 90 % alike.
    The decode's last projection runs in fp32 on the master weights (mk_logits_f32), so what can still move a decision is the
    bf16 rounding inherited from the layers below -- a relative 2^-9 per operand, far less than the spread of a position's
    logits.  Per utterance: every token up to the first position whose reference margin (best - second logit, in units of that
    position's logit std, from the fp32 oracle == the golden's tokens) is below MARGIN must equal the reference's; a divergence
    may only start AT such a near-tie (after it the prefixes differ and nothing is comparable)."""
    MARGIN = 0.05
    n_checked = n_total = 0
    for case, seed in (("ragged", 11), ("same", 12), ("single", 13)):
        ilens, olens = CASES[case]
        xs, il, ys, ol = synth_batch(seed, ilens, olens)
        eng = MasrEngine(TINY, ODIM)
        eng.load_state_dict(sd)
        hyp = eng.recog(xs, il).cpu()
        with torch.no_grad():
            ref, margin = ref_cpu.recog_greedy(ref_cpu.leafify(sd, TINY), TINY, xs, il, margins=True)
        if case == "ragged":
            assert torch.equal(ref, torch.from_numpy(G["recog/hyp"]))       # the oracle's tokens ARE the reference's (golden)
        for b in range(len(ilens)):
            tie = (margin[:, b] < MARGIN).nonzero()
            stop = int(tie[0]) if len(tie) else ref.shape[0]              # first near-tie of the reference's own decisions
            n_checked += stop; n_total += ref.shape[0]
            assert torch.equal(hyp[:stop, b], ref[:stop, b]), (case, b, stop, hyp[:, b].tolist(), ref[:, b].tolist(), margin[:, b].tolist())
            diff = (hyp[:, b] != ref[:, b]).nonzero()
            if len(diff):
                first = int(diff[0])
                assert float(margin[first, b]) < MARGIN, f"{case} utt {b}: tokens diverge at step {first} where the reference's margin is {float(margin[first, b]):.3f}"
    print(f"greedy decode: {n_checked} of {n_total} tokens lie before the reference's first near-tie (margin < {MARGIN} logit std) -- all identical")
    assert n_checked >= 0.5 * n_total


def test_recog_cached_equals_full_redecode(sd):
    """the KV-cached incremental decode (direct launches on the default stream, hipGraph replay on a
    side stream) emits exactly the tokens of the reference's literal schedule (whole prefix decoded again per step)."""
    ilens, olens = CASES["ragged"]
    xs, il, ys, ol = synth_batch(11, ilens, olens)
    eng = MasrEngine(TINY, ODIM)
    eng.load_state_dict(sd)
    full = eng.recog(xs, il, full=True).cpu()
    direct = eng.recog(xs, il).cpu()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g1 = eng.recog(xs, il)
        g2 = eng.recog(xs, il)                       # second call replays the cached graph
    side.synchronize()
    assert torch.equal(direct, full), (direct.T, full.T)
    assert torch.equal(g1.cpu(), full) and torch.equal(g2.cpu(), full)
    # a different batch geometry re-captures
    xs2, il2, _, _ = synth_batch(5, [40, 24], [3, 3])
    f2 = eng.recog(xs2, il2, full=True).cpu()
    with torch.cuda.stream(side):
        c2 = eng.recog(xs2, il2)
    side.synchronize()
    assert torch.equal(c2.cpu(), f2)


def test_recog_cached_hkust_geometry():
    """hkust widths (E 512, H 8, F 2048, 4 decoder layers, B 16): cached decode vs full re-decode, random-init weights.
    Random-init logits are nearly flat, so a bf16-level difference between the flash kernel (full) and the one-query
    kernel (cached) may flip an arg-max, after which that utterance diverges; require the common prefix to cover
    almost everything and the first 8 steps to be identical for every utterance."""
    torch.manual_seed(3)
    eng = MasrEngine(HKUST, ODIM)
    eng.load_state_dict(ref_cpu.deterministic_state_dict(HKUST, ODIM, seed=3))
    B, T = 16, 200
    xs = torch.randn(B, T, 83)
    il = torch.tensor([T - 4 * (i % 5) for i in range(B)])
    il, _ = il.sort(descending=True)
    full = eng.recog(xs, il, full=True).cpu()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        cached = eng.recog(xs, il).cpu()
    side.synchronize()
    assert cached.shape == full.shape == (T // 4, B)
    same = (cached == full)
    prefix = torch.cummin(same.int(), dim=0).values.sum(0)          # per-utterance length of the identical prefix
    print(f"cached vs full: identical prefix per utterance {prefix.tolist()} of {T // 4}")
    assert bool(same[:8].all())
    assert float(prefix.float().mean()) >= 0.9 * (T // 4)


HKUST = {"idim": 83, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.0, "pos_dropout": 0.0, "tgt_share_weight": 1,
         "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4},
         "meta": {"optimizer_opt": {"k": 1.0, "warmup_steps": 25000}}}


def test_hkust_geometry_inner_steps_vs_oracle():
    """The shipped fometa-hkust.yaml geometry (24.88 M parameters, head dim 64, idim 83 -> odd widths in the VGG) on a
    ragged batch: loss of two consecutive inner steps (run_batch + clip 5 + Nesterov SGD at the shipped inner lr) against
    the fp32 oracle within the north-star 1e-3, and the flat gradient's direction / size against the bf16-emulated oracle."""
    sd = ref_cpu.deterministic_state_dict(HKUST, ODIM, seed=3)
    assert sum(v.numel() for k, v in sd.items() if k not in ("pos_encoder.pe", "pre_embed.weight")) == 24881455
    ilens, olens = [203, 160, 121], [12, 9, 7]
    b1, b2 = synth_batch(5, ilens, olens), synth_batch(6, ilens, olens)
    eng = MasrEngine(HKUST, ODIM, label_smoothing=0.2)
    eng.load_state_dict(sd)
    mom = torch.zeros_like(eng.params)
    lr = ref_cpu.inner_lr(HKUST)
    got, g_first = [], None
    for i, b in enumerate((b1, b2)):
        eng.run_batch(b[0], b[1], b[2], b[3], train=True)
        if i == 0:
            g_first = eng.state_dict(flat=eng.grads)
        eng.clip_sgd_step(mom, 5.0, lr, 0.9, True, first_step=(i == 0))
        got.append(eng.read_stats())
    p = ref_cpu.leafify(sd, HKUST)
    bufs = {}
    ref = [ref_cpu.inner_step(p, HKUST, (b[0], b[1], b[2], b[3].clone()), 0.2, bufs, lr) for b in (b1, b2)]
    for g, r in zip(got, ref):
        assert abs(g["loss"] - r["loss"]) <= 1e-3 * r["loss"], (g["loss"], r["loss"])
        assert abs(g["grad_norm"] - r["grad_norm"]) <= 3e-2 * r["grad_norm"], (g["grad_norm"], r["grad_norm"])
    with ref_cpu.bf16_emulation():
        pq = ref_cpu.leafify(sd, HKUST)
        _, gq, _, _ = ref_cpu.run_batch_train(pq, HKUST, (b1[0], b1[1], b1[2], b1[3].clone()), 0.2)
    names = [n for n in ref_cpu.grad_param_names(pq, HKUST) if not n.endswith("in_proj_bias")]
    a = torch.cat([g_first[n].cpu().reshape(-1) for n in names]).double(); bq = torch.cat([gq[n].reshape(-1) for n in names]).double()
    cos = float((a * bq).sum() / (a.norm() * bq.norm()))
    print(f"hkust: losses {[round(g['loss'], 5) for g in got]} vs {[round(r['loss'], 5) for r in ref]}; grad cos {cos:.5f}")
    assert cos > 0.995 and abs(float(a.norm() / bq.norm()) - 1) < 2e-2


def test_graph_replayed_steps_equal_direct_launches(sd):
    """(opt-in, masr_set_step_graphs) A batch shape that repeats is captured into a hipGraph on its second occurrence and replayed afterwards; everything that
    changes between steps (tokens, lengths, dropout seed, 1/n_total) reaches the kernels through device memory.  Replayed
    steps must be BIT-identical to directly launched ones: same losses, same gradients, same dropout masks, step after step
    (the direct engine is kept off the graph path by its profiling switch), including a change of labels at a fixed shape."""
    cfg = dict(TINY); cfg["dropout"] = 0.1; cfg["pos_dropout"] = 0.1
    xs, il, ys, ol = synth_batch(11, *CASES["ragged"])
    ys2 = [(y + 1) % 365 + 1 for y in ys]                     # other labels, same lengths -> same shape, other n_correct / loss
    xs_dev = xs.cuda()
    engs = []
    for direct in (False, True):
        e = MasrEngine(cfg, ODIM, label_smoothing=0.2)
        e.load_state_dict(sd); e.set_seed(99)
        if direct:
            e.profile(True)
        e.set_step_graphs(True)
        engs.append(e)
    moms = [torch.zeros_like(e.params) for e in engs]
    with torch.cuda.stream(torch.cuda.Stream()):               # (steps on the legacy NULL stream are never captured)
        for step in range(6):
            labels = ys if step % 3 else ys2
            out = []
            for e, mom in zip(engs, moms):
                e.run_batch(xs_dev, il, labels, ol, train=True)
                g = e.grads.clone()
                e.clip_sgd_step(mom, 5.0, 0.01, 0.9, True, step == 0)
                out.append((e.read_stats(), g, e.params.clone()))
            (s0, g0, p0), (s1, g1, p1) = out
            assert s0 == s1, (step, s0, s1)
            assert torch.equal(g0, g1) and torch.equal(p0, p1), f"step {step}: replayed and direct launches differ"
        # evaluation at the same shape (separate graph: no dropout, no backward)
        for _ in range(3):
            r = []
            for e in engs:
                e.run_batch(xs_dev, il, ys, ol, train=False)
                r.append(e.read_stats()["loss"])
            assert r[0] == r[1]
        torch.cuda.current_stream().synchronize()
    c0 = engs[0].step_counters()
    assert (c0["direct"], c0["captured"], c0["replayed"]) == (2, 2, 7), c0
    assert engs[1].step_counters()["replayed"] == 0


def test_eight_encoder_layers_build_and_step():
    """config/transformer/mono-test.yaml / mono-test-new-trick-8e4d.yaml geometry (8 encoder / 4 decoder layers = 69 shadow
    jobs, more than one by-value job list holds): the engine is created, the shadow refresh takes two launches, and a training
    step matches the CPU oracle like the 2e2d model does."""
    cfg = dict(TINY, encoder={"nlayers": 8}, decoder={"nlayers": 4})
    sd8 = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=5)
    xs, il, ys, ol = synth_batch(21, [64, 52, 40, 33], [9, 7, 5, 3])
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
    eng.load_state_dict(sd8)
    eng.run_batch(xs, il, ys, ol, train=True)
    g_all = {k: v.clone() for k, v in eng.state_dict(flat=eng.grads).items()}
    mom = torch.zeros_like(eng.params)
    eng.clip_sgd_step(mom, 5.0, 0.05, 0.9, True, True)
    st = eng.read_stats()
    with ref_cpu.bf16_emulation():
        pq = ref_cpu.leafify(sd8, cfg)
        infoq, gradsq, _, _ = ref_cpu.run_batch_train(pq, cfg, (xs, il, ys, ol.clone()), 0.2)
    p = ref_cpu.leafify(sd8, cfg)
    info, _, _, _ = ref_cpu.run_batch_train(p, cfg, (xs, il, ys, ol.clone()), 0.2)
    assert abs(st["loss"] - info["loss"]) <= 1e-3 * info["loss"], (st["loss"], info["loss"])
    for n in ("encoder.layers.7.linear1.weight", "encoder.layers.0.self_attn.in_proj_weight", "decoder.layers.3.multihead_attn.out_proj.weight",
              "decoder.layers.3.linear2.weight", "vgg2enc.weight"):
        assert rel_l2(g_all[n].cpu(), gradsq[n]) < 8e-2, n
    # the second step reads the shadows of EVERY layer refreshed by the (two-launch) refresh after the update
    eng.run_batch(xs, il, ys, ol, train=True)
    st2 = eng.read_stats()
    assert math.isfinite(st2["loss"]) and st2["loss"] < st["loss"]


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_ksplit_gemms_with_the_combine_inside_the_layernorm(dropout):
    """engine.hip ksplit_of: at the hkust geometry the decoder's FFN second layer (K = 2048), its first layer's dgrad and the packed q/k/v dgrad
    (K = 1536) run k-split, and the LayerNorm (backward) behind each sums the fp32 partial products and applies the GEMM's epilogue (bias, dropout
    with the GEMM's element index, residual).  The pair itself is checked number by number in test_hip_kernels.py
    (test_ksplit_gemm_summed_by_the_layernorm) and on a one-decoder-layer model below (test_ksplit_one_decoder_layer_is_tight); here the whole
    step against whole reductions (masr_set_ksplit(0), the engine's default).  The two differ in fp32
    summation order only, but not bit for bit downstream: a last-bit difference moves some bf16 operand roundings of the following layers, so
    the step agrees to the engine's bf16 noise floor -- the same distance either schedule keeps from the bf16-emulated oracle (logits ~5e-3 of
    their range, loss ~1e-4): loss to 3e-4, logits to 1 % of their range, every gradient tensor's direction to cos > 0.998, the flat gradient to
    cos > 0.9995 and 0.5 % in norm (the engine against the bf16-emulated oracle: cos > 0.995).  The first decoder row of the batch (causal attention:
    the fewest such roundings upstream) agrees to 2e-3 of the logits' range, which a wrong element index in the dropout or a missing bias would
    not leave.  Since round 5 the attention out-projections (K = 512, two halves) take the same route."""
    cfg = dict(HKUST)
    cfg["dropout"] = cfg["pos_dropout"] = dropout
    sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=4)
    xs, il, ys, ol = synth_batch(23, [203, 160, 121, 96, 90], [12, 9, 7, 30, 2])
    outs = []
    for on in (True, False):
        eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
        eng.load_state_dict(sd)
        eng.set_seed(17)
        eng.set_ksplit(on)
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        assert (eng.step_counters()["ksplit_gemms"] > 0) == on       # (4 decoder layers x {3 forward + 3 backward} launches, less layer 0's input dgrad)
        outs.append((dict(eng.read_stats()), eng.last_logits()[0].clone(), eng.grads.clone(), eng))
    (sa, la, ga, eng), (sb, lb, gb, _) = outs
    assert abs(sa["loss"] - sb["loss"]) <= 3e-4 * abs(sb["loss"]), (sa["loss"], sb["loss"])
    assert float((la - lb).abs().max()) <= 1e-2 * float(lb.abs().max())
    assert not torch.equal(ga, gb)                            # (the two schedules do differ: this is not the same code path twice)
    assert float((la[0, 0] - lb[0, 0]).abs().max()) <= 2e-3 * float(lb.abs().max())      # first decoder row of the batch: the fewest re-roundings upstream
    worst = 1.0
    for n, (off, shape) in eng.table.items():
        k = int(np.prod(shape))
        a, b = ga[off:off + k].double(), gb[off:off + k].double()
        if float(b.norm()) > 0:
            worst = min(worst, float((a * b).sum() / (a.norm() * b.norm())))
    a, b = ga.double(), gb.double()
    cos = float((a * b).sum() / (a.norm() * b.norm()))
    print(f"dropout {dropout}: k-split vs whole reductions: loss {sa['loss']:.6f} / {sb['loss']:.6f}, flat gradient cos {cos:.6f}, worst tensor cos {worst:.5f}")
    assert worst > 0.998 and cos > 0.9995 and abs(float(a.norm() / b.norm()) - 1) < 5e-3

def test_ksplit_one_decoder_layer_is_tight():
    """The tight engine-level check of the k-split schedule (the whole-step test above can only hold the two schedules to the bf16 noise of four
    decoder layers): hkust width, ONE decoder layer, dropout off.  Behind the layer's three k-split GEMMs (self / cross out-projection halves,
    FFN second layer in four) there are only three bf16 re-rounding points (the LayerNorm outputs), so the two schedules may differ by a handful of
    last-bit flips of bf16 operands (an fp32 sum-order difference of ~1e-7 crosses a bf16 rounding boundary with probability ~3e-5 per element:
    ~10 of a 592 x 512 tensor) and by nothing else: at least 90 % of the logits rows are EQUAL to 2e-6 of the range (measured 93.5 %: the rows
    no flip reached), the whole tensor agrees to a rel-L2 of 1e-3 (measured 4e-4: the rows a flip did reach), the loss to 2e-5 (measured 5e-6)
    and the gradients of that layer's attention / FFN weights to a rel-L2 of 1.5e-2 (measured 5e-3: the backward has its own three
    k-split dgrads, each followed by bf16 gradient operands; a wrong partial, bias, residual or dropout index is a 1e-1 effect on every row)."""
    cfg = dict(HKUST)
    cfg["decoder"] = dict(cfg["decoder"], nlayers=1)
    cfg["dropout"] = cfg["pos_dropout"] = 0.0
    sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=14)
    xs, il, ys, ol = synth_batch(29, [203, 160, 121, 96, 90], [12, 9, 7, 30, 2])
    outs = []
    for on in (True, False):
        eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
        eng.load_state_dict(sd)
        eng.set_seed(5)
        eng.set_ksplit(on)
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        assert (eng.step_counters()["ksplit_gemms"] > 0) == on
        outs.append((dict(eng.read_stats()), eng.last_logits()[0].clone(), eng.grads.clone(), eng))
    (sa, la, ga, eng), (sb, lb, gb, _) = outs
    rng = float(lb.abs().max())
    rel = float((la - lb).double().norm() / lb.double().norm())
    row_ok = float(((la - lb).abs().amax(-1) <= 2e-6 * rng).float().mean())
    print(f"one decoder layer, k-split vs whole: loss {sa['loss']:.7f} / {sb['loss']:.7f}, logits rel-L2 {rel:.2e}, rows equal to 2e-6 of the range: {row_ok:.3f}")
    assert abs(sa["loss"] - sb["loss"]) <= 2e-5 * abs(sb["loss"]) and rel <= 1e-3 and row_ok >= 0.9
    worst = 0.0
    for n, (off, shape) in eng.table.items():
        if not n.startswith("decoder.layers.0.") or ".norm" in n:
            continue
        k = int(np.prod(shape))
        a, b = ga[off:off + k].double(), gb[off:off + k].double()
        worst = max(worst, float((a - b).norm() / b.norm()))
    print(f"  worst decoder-layer weight-gradient rel-L2 between the schedules: {worst:.2e}")
    assert 0 < worst <= 1.5e-2


@pytest.mark.parametrize("ksplit", [False, True])
def test_task_slot_hint_never_changes_bits(ksplit):
    """include/masr.h masr_set_concurrency: a hint (LDS footprint of a few launches), no result follows it -- a C-ABI user who runs
    set_concurrency(1) and then set_concurrency(4) gets the same bits, with the k-split on or off (the split follows masr_set_ksplit only)."""
    cfg = dict(HKUST)
    sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=9)
    xs, il, ys, ol = synth_batch(31, [203, 160, 121, 96, 90], [12, 9, 7, 30, 2])
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
    eng.set_ksplit(ksplit)
    outs = []
    for slots in (1, 4, 1):
        eng.load_state_dict(sd)
        eng.set_seed(21)
        eng.set_concurrency(slots)
        mom = torch.zeros_like(eng.params)
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        g = eng.grads.clone()
        eng.clip_sgd_step(mom, 5.0, 0.05, 0.9, True, True)
        outs.append((dict(eng.read_stats()), eng.last_logits()[0].clone(), g, eng.params.clone()))
        assert (eng.step_counters()["ksplit_gemms"] > 0) == ksplit
    for o in outs[1:]:
        assert o[0] == outs[0][0] and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2]) and torch.equal(o[3], outs[0][3])


@pytest.mark.parametrize("cfg_name", ["tiny", "hkust"])
def test_merged_weight_gradient_launch_equals_two_launches(cfg_name):
    """engine.hip flush_wgrads: the decoder-row weight gradients ride in the encoder rows' launch (two-segment tile list, long tiles
    dispatched first).  Against the two separate launches (masr_set_split_wgrad_launches): every gradient of a training step bit for bit."""
    cfg = dict(TINY if cfg_name == "tiny" else HKUST)
    sd = ref_cpu.deterministic_state_dict(cfg, ODIM, seed=6)
    ilens, olens = ([64, 52, 40, 33], [9, 7, 5, 3]) if cfg_name == "tiny" else ([203, 160, 121, 96, 90], [12, 9, 7, 30, 2])
    xs, il, ys, ol = synth_batch(22, ilens, olens)
    grads = []
    for split in (False, True):
        eng = MasrEngine(cfg, ODIM, label_smoothing=0.2)
        eng.load_state_dict(sd)
        eng.set_seed(3)
        eng.set_split_wgrad_launches(split)
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        grads.append(eng.grads.clone())
        assert np.isfinite(eng.read_stats()["loss"])
    assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0


def test_sgd_step_with_nan_norm_is_skipped():
    """math.isnan(grad_norm) -> the reference skips the step (fo_meta_interface.py:245): the update pass decides on the device (the norm
    never travels to the host) and leaves the parameters untouched"""
    cfg = dict(TINY)
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.1)
    eng.load_state_dict(ref_cpu.deterministic_state_dict(cfg, ODIM, seed=8))
    xs, il, ys, ol = synth_batch(22, [64, 52], [9, 7])
    eng.run_batch(xs, il, ys, ol.clone(), train=True)
    eng.grads[5] = float("nan")
    before = eng.params.clone()
    eng.clip_sgd_step(torch.zeros_like(eng.params), 5.0, 0.05, 0.9, True, 3)
    assert torch.equal(eng.params, before)


def test_nan_val_gradient_accumulates_by_default_and_is_dropped_on_request():
    """Quirk Q5 (fo_meta_interface.py:151-154): the reference warns about a NaN val-batch gradient and adds it to `_updates` all the same.
    Default = that; masr_set_drop_nan_grads (pretrain.py --fix_nan_meta_grad): the clip zeroes such a gradient / the accumulate leaves
    `updates` alone, decided on the device; the reported norm stays NaN (the warning is still logged); a finite gradient is unaffected."""
    cfg = dict(TINY)
    eng = MasrEngine(cfg, ODIM, label_smoothing=0.1)
    eng.load_state_dict(ref_cpu.deterministic_state_dict(cfg, ODIM, seed=8))
    xs, il, ys, ol = synth_batch(22, [64, 52], [9, 7])

    def fresh(poison):
        eng.run_batch(xs, il, ys, ol.clone(), train=True)
        if poison:
            eng.grads[5] = float("nan")
    # reference behaviour
    fresh(True)
    upd = torch.ones_like(eng.params)
    eng.clip_accumulate(upd, 5.0)
    assert bool(torch.isnan(upd).all())
    fresh(True)
    eng.clip_grads(5.0)
    assert bool(torch.isnan(eng.grads).all()) and math.isnan(eng.read_stats()['grad_norm'])
    # a finite gradient: the switch changes nothing
    fresh(False); eng.clip_grads(5.0); want = eng.grads.clone()
    eng.set_drop_nan_grads(True)
    fresh(False); eng.clip_grads(5.0)
    assert torch.equal(eng.grads, want) and float(want.abs().max()) > 0
    # NaN norm with the switch on
    fresh(True)
    upd = torch.ones_like(eng.params)
    eng.clip_accumulate(upd, 5.0)
    assert bool((upd == 1).all())
    fresh(True)
    eng.clip_grads(5.0)
    assert bool((eng.grads == 0).all()) and math.isnan(eng.read_stats()['grad_norm'])
