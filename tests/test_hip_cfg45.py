"""BASELINE configs[3] and configs[4] as far as ONE MI355X takes them, through the drop-in CLI (`pretrain.main(argv)`):

  configs[3]  pretrain.py --algo fomaml, EIGHT accents, meta_batch_size 8 (on the 8-GPU node: one task per GPU).  Here the eight
              tasks of a meta-step run as two waves of four concurrent task slots (--tasks_per_gpu 4), checked against
                * the REFERENCE's own run of that command line (tests/golden/fomaml_8acc.npz, its seed-531 initialisation):
                  initial weights, task order, batch identities, every train / eval loss within 1e-3, logs line by line;
                * the sequential (--tasks_per_gpu 1) run, bit for bit: snapshots, meta weights, Adam moments, logs;
                * the oracle's meta loop on the CPU, tensor by tensor: meta-gradient and post-Adam meta weights.
  configs[4]  pretrain.py --algo reptile (--fix_reptile: the reference raises ValueError; parity unpinned),
              8 accents, inner_steps = 5 -> train.py fine-tune -> train.py --test.  The pseudo-gradient theta_meta - theta_5 of the
              first meta-step tensor by tensor against oracle.ref_cpu.reptile_meta_step with five inner steps; the chain runs end to
              end and learns.
What only the 8-GPU node can add is the RCCL transport itself (tests/test_parallel_train_gloo.py runs the same loops on 8 gloo ranks).
Reference: pretrain.py:43,55-57, src/fo_meta_interface.py:128-250."""
import random
from functools import partial

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402,F401
from oracle import ref_cpu  # noqa: E402
from oracle.make_goldens import EIGHT_ACCENTS, ODIM, eight_workspace, flat_checks  # noqa: E402
from replay import eight_accent_setup, oracle_fomaml_run  # noqa: E402

CODES = [c for c, _ in EIGHT_ACCENTS]


def spied_pretrain(monkeypatch, argv, spy=True):
    """pretrain.main(argv) with the solver it builds instrumented the way tests/test_hip_cfg3.py instruments its own: every
    run_batch call (want_info forced: the product skips the host sync of calls whose info the reference discards), the initial
    meta weights, and per meta-step the meta-gradient (or Reptile pseudo-gradient) before Adam + the meta weights after it."""
    import pretrain
    import masr_amd.transformer_torch_trainer as ttt
    rec = {"calls": [], "steps": [], "init": None, "solver": None}
    orig_get = ttt.get_trainer

    def get_trainer(cls, config, paras, id2accent):
        solver = orig_get(cls, config, paras, id2accent)
        rec["solver"] = solver
        if not spy:
            return solver
        orig = solver.run_batch

        def run_batch(idx, x, ilens, ys, olens, train, accent_idx=None, **kw):
            r = (int(idx), bool(train), ilens.clone(), [y.clone() for y in ys])
            kw["want_info"] = True
            info = orig(idx, x, ilens, ys, olens, train=train, accent_idx=accent_idx, **kw)
            rec["calls"].append(r + (dict(info),))
            return info
        solver._train, solver._eval = partial(run_batch, train=True), partial(run_batch, train=False)
        orig_final = solver._final_meta_update

        def final(n_tasks=None):
            if rec["init"] is None:
                rec["init"] = solver._original.cpu().clone()
            mg = (solver._updates / solver._counter).cpu()
            orig_final(n_tasks)
            torch.cuda.synchronize()
            rec["steps"].append((mg, solver._original.cpu().clone()))
        solver._final_meta_update = final
        return solver
    monkeypatch.setattr(ttt, "get_trainer", get_trainer)
    pretrain.main(argv)
    torch.cuda.synchronize()
    monkeypatch.setattr(ttt, "get_trainer", orig_get)
    return rec


def pretrain_argv(algo, suffix, meta_k, max_step, k_slots, extra=()):
    return ["--config", "pre.yaml", "--pretrain_suffix", suffix, "--pretrain_accents", *CODES, "--num_pretrain", "8", "--tgt_accent", "ca",
            "--algo", algo, "--meta_k", str(meta_k), "--meta_batch_size", "8", "--max_step", str(max_step), "--njobs", "2",
            "--tasks_per_gpu", str(k_slots), "--overwrite", *extra]


def per_tensor_report(eng, flat, want, skip_key_third_E=None):
    """rel-L2 of every tensor of a flat buffer against a name -> tensor dict; -> [(err, name)] (largest first), cosine, norm ratio"""
    rows, dots = [], np.zeros(3)
    for n, ref in want.items():
        off, shape = eng.table[n]
        a = flat[off:off + int(np.prod(shape))].view(shape).double()
        b = ref.double()
        if skip_key_third_E and n.endswith("in_proj_bias"):          # key third: exactly-zero true gradient (softmax shift invariance)
            E = skip_key_third_E
            a, b = torch.cat([a[:E], a[2 * E:]]), torch.cat([b[:E], b[2 * E:]])
        rows.append((float((a - b).norm() / (b.norm() + 1e-30)), n))
        dots += [float((a * b).sum()), float((a * a).sum()), float((b * b).sum())]
    rows.sort(reverse=True)
    return rows, dots[0] / np.sqrt(dots[1] * dots[2]), np.sqrt(dots[1] / dots[2])


def test_cfg4_fomaml_eight_accents_two_waves_of_four_slots(golden_dir, tmp_path, monkeypatch):
    g = np.load(golden_dir / "fomaml_8acc.npz")
    monkeypatch.chdir(tmp_path)
    cfg, _ = eight_workspace(tmp_path, golden_dir)
    yaml.safe_dump(cfg, open(tmp_path / "pre.yaml", "w"))

    # ---- 1. the sequential run, instrumented: against the reference's run of the same command line
    seq = spied_pretrain(monkeypatch, pretrain_argv("fomaml", "seq", 1, 3, 1))
    solver, eng = seq["solver"], seq["solver"].asr_model.engine
    for n, (off, shape) in eng.table.items():
        if f"init/fp/{n}" in g.files:
            got, want = flat_checks(seq["init"][off:off + int(np.prod(shape))]), g[f"init/fp/{n}"]
            assert np.allclose(got, want, rtol=1e-6, atol=1e-9), f"initial {n} differs from the reference's seed-531 initialisation"
    calls = seq["calls"]
    assert len(calls) == int(g["n_calls"]) == 40
    worst = {True: 0.0, False: 0.0}
    for i, (accent, train, il, ys, info) in enumerate(calls):
        assert accent == int(g[f"call{i}/accent"]) and int(train) == int(g[f"call{i}/train"]), i
        np.testing.assert_array_equal(il.numpy(), g[f"call{i}/ilens"])
        np.testing.assert_array_equal(np.concatenate([y.numpy() for y in ys]), g[f"call{i}/ys"])
        ref = float(g[f"call{i}/loss"])
        rel = abs(info["loss"] - ref) / ref
        worst[train] = max(worst[train], rel)
        assert rel <= 1e-3, f"call {i}: loss {info['loss']} vs reference {ref} (rel {rel:.2e})"
        assert abs(info["acc"] - float(g[f"call{i}/acc"])) <= 1.0 / sum(len(y) + 1 for y in ys) + 1e-6
    print(f"8 accents, meta-batch 8: {len(calls)} run_batch calls, worst relative loss error train {worst[True]:.2e}, eval {worst[False]:.2e}")
    files = sorted(p.name for p in solver.log_dir.iterdir())
    assert [f for f in files if f not in ("meta_state.latest", "dashboard.jsonl")] == [str(f) for f in g["files"]]
    for key in g.files:
        if not key.startswith("log/"):
            continue
        name = key[4:]
        ours, ref = (solver.log_dir / name).read_text().split(), str(g[key]).split()
        assert len(ours) == len(ref), name
        for j in range(0, len(ref) - 1, 2):
            assert ours[j] == ref[j], (name, ours, ref)
            vo, vr = float(ours[j + 1]), float(ref[j + 1])
            tol = 1e-3 * abs(vr) if name.endswith("_loss") else (0.06 if name.endswith("_acc") else 0.02 * abs(vr) + 1e-9)
            assert abs(vo - vr) <= tol, (name, vo, vr)
    assert int(g["global_step"]) == solver.global_step and int(g["meta/step_num"]) == solver.meta_opt.step_num
    assert abs(float(g["meta/lr"]) - solver.meta_opt.lr) <= 1e-15

    # ---- 2. two waves of four concurrent task slots == the sequential run, bit for bit
    par = spied_pretrain(monkeypatch, pretrain_argv("fomaml", "par", 1, 3, 4), spy=False)
    assert par["solver"].tasks_per_gpu == 4 and par["solver"]._slots is not None and len(par["solver"]._slots) == 4
    d_seq, d_par = seq["solver"].log_dir, par["solver"].log_dir
    for name in ("snapshot.latest", "snapshot.step.2", "model.wer.best"):
        a, b = torch.load(d_seq / name), torch.load(d_par / name)
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), name
    ma, mb = torch.load(d_seq / "meta_state.latest", weights_only=False), torch.load(d_par / "meta_state.latest", weights_only=False)
    assert torch.equal(ma["original"], mb["original"]) and ma["step_num"] == mb["step_num"] == 1    # (written when global_step became 2)
    for k, v in ma["adam"].items():
        assert torch.equal(v, mb["adam"][k]) if torch.is_tensor(v) else v == mb["adam"][k], k
    assert torch.equal(seq["steps"][-1][1], par["solver"]._original.cpu())
    for p in d_seq.iterdir():
        if p.name.startswith(("train_", "dev_", "best_")):
            assert p.read_text() == (d_par / p.name).read_text(), p.name

    # ---- 3. the oracle's meta loop beside it (fp32, and with the engine's bf16 rounding points emulated), tensor by tensor
    def oracle(emulate):
        _, _, dc, init = eight_accent_setup(tmp_path, golden_dir)
        if emulate:
            with ref_cpu.bf16_emulation():
                return oracle_fomaml_run(cfg, dc, 1, 8, 3, cfg["solver"]["label_smoothing"], init_sd=init), init
        return oracle_fomaml_run(cfg, dc, 1, 8, 3, cfg["solver"]["label_smoothing"], init_sd=init), init
    (o32, init), (o16, _) = oracle(False), oracle(True)
    assert len(seq["steps"]) == len(o32["steps"]) == 2
    lr_sum = 0.0
    for si, (mg_flat, meta_flat) in enumerate(seq["steps"]):
        lr_sum += ref_cpu.noam_lr(si + 1, 1.0, cfg["asr_model"]["d_model"], 25000)
        r32, cos, ratio = per_tensor_report(eng, mg_flat, o32["steps"][si][0])
        r16, _, _ = per_tensor_report(eng, mg_flat, o16["steps"][si][0])
        print(f"  meta-step {si}: meta-gradient cosine vs the fp32 oracle {cos:.5f}, norm ratio {ratio:.4f}; worst tensors "
              f"(fp32 oracle) {[(n, round(e, 4)) for e, n in r32[:3]]}; (bf16-emulating oracle) {[(n, round(e, 4)) for e, n in r16[:3]]}")
        assert cos > 0.995 and abs(ratio - 1) < 0.02
        # per-tensor bounds from THIS run's own measurement (worst tensor 0.041 vs the fp32 oracle, 0.025 vs the bf16-emulating one: decoder
        # linear1 and the first conv), with half again as margin; in_proj_bias: its key third has a zero gradient
        for e, n in r32:
            assert n.endswith("in_proj_bias") or e < 0.06, f"meta-gradient step {si} {n}: rel-L2 {e:.3f} vs the fp32 oracle"
        for e, n in r16:
            assert n.endswith("in_proj_bias") or e < 0.04, f"meta-gradient step {si} {n}: rel-L2 {e:.3f} vs the bf16-emulating oracle"
        for n, w_ref in o32["steps"][si][1].items():
            if n not in eng.table or n == "pos_encoder.pe":
                continue
            off, shape = eng.table[n]
            w = meta_flat[off:off + int(np.prod(shape))].view(shape)
            ulp = float(w_ref.abs().max()) * 2.0 ** -23
            assert float((w - w_ref).abs().max()) <= 2.5 * lr_sum + ulp, (si, n)


def test_cfg5_reptile_five_inner_steps_eight_accents_chain(golden_dir, tmp_path, monkeypatch):
    import train
    monkeypatch.chdir(tmp_path)
    pre, ft = eight_workspace(tmp_path, golden_dir)
    yaml.safe_dump(pre, open(tmp_path / "pre.yaml", "w"))
    yaml.safe_dump(ft, open(tmp_path / "ft.yaml", "w"))
    # ---- 1. pretrain.py --algo reptile --meta_k 5 on 8 accents (two waves of four slots), 4 meta-steps
    run = spied_pretrain(monkeypatch, pretrain_argv("reptile", "rep", 5, 5, 4, extra=("--fix_reptile",)))
    solver, eng = run["solver"], run["solver"].asr_model.engine
    assert len(run["steps"]) == 4 and solver.meta_opt.step_num == 4
    assert sum(1 for c in run["calls"] if c[1]) == 4 * 8 * 6                 # per task: 5 inner batches + the val batch
    pre_dir = solver.log_dir
    assert (pre_dir / "snapshot.step.4").exists()
    # ---- 2. the first two meta-steps against the oracle (bf16 rounding points emulated), EVERY tensor of the pseudo-gradient
    _, _, dc, init = eight_accent_setup(tmp_path, golden_dir)
    with ref_cpu.bf16_emulation():
        o = oracle_fomaml_run(pre, dc, 5, 8, 3, pre["solver"]["label_smoothing"], init_sd=init, algo="reptile")
    E = pre["asr_model"]["d_model"]
    got_train = {}
    for accent, train_, il, ys, info in run["calls"]:
        if train_:
            got_train.setdefault(accent, []).append(info["loss"])
    want_train = {}
    for accent, train_, info in o["calls"]:
        if train_:
            want_train.setdefault(accent, []).append(info["loss"])
    worst = 0.0
    for a, want in want_train.items():                                   # per accent, in call order (slots interleave ACROSS accents only)
        for x, y in zip(got_train[a], want):
            worst = max(worst, abs(x - y) / y)
    print(f"reptile meta_k 5: worst relative loss error over the first two meta-steps' {sum(map(len, want_train.values()))} calls {worst:.2e}")
    assert worst <= 1e-3
    lr_sum = 0.0
    for si in range(2):
        lr_sum += ref_cpu.noam_lr(si + 1, 1.0, E, 25000)
        rows, cos, ratio = per_tensor_report(eng, run["steps"][si][0], o["steps"][si][0], skip_key_third_E=E)
        print(f"  meta-step {si}: pseudo-gradient cosine {cos:.5f}, norm ratio {ratio:.4f}, worst tensors {[(n, round(e, 4)) for e, n in rows[:4]]}")
        assert cos > 0.999 and abs(ratio - 1) < 0.02
        for e, n in rows:                                                  # bound of test_reptile_behind_fix_flag_matches_oracle
            assert e < 0.03, f"pseudo-gradient step {si} {n}: rel-L2 {e:.3f}"        # (measured: worst tensor 0.015)
        for n, w_ref in o["steps"][si][1].items():
            if n not in eng.table or n == "pos_encoder.pe":
                continue
            off, shape = eng.table[n]
            w = run["steps"][si][1][off:off + int(np.prod(shape))].view(shape)
            assert float((w - w_ref).abs().max()) <= 2.5 * lr_sum + 1.2e-7 * float(w_ref.abs().max() + 1.0), (si, n)
    # ---- 3. fine-tune on the target accent from snapshot.step.4, decode the test shard
    common = ["--config", "ft.yaml", "--accent", "ca", "--algo", "reptile", "--eval_suffix", "ft", "--njobs", "1"]
    train.main(common + ["--pretrain", "--pretrain_suffix", "rep", "--pretrain_setting", "eight", "--pretrain_step", "4",
                         "--pretrain_tgt_accent", "ca", "--overwrite"])
    ft_dir = tmp_path / "testing-logs" / "evaluation" / "eight-ft" / "reptile" / "rep" / "ft" / "canada" / "0"
    train.main(common + ["--pretrain_suffix", "rep", "--test", "--decode_batch_size", "4", "--overwrite"])
    lines = (ft_dir / "greedy_decode" / "best-hyp").read_text().splitlines()
    log = lambda p: [(int(l.split()[0]), float(l.split()[1])) for l in p.read_text().splitlines() if l.strip()]
    da, dl = log(ft_dir / "dev_acc"), log(ft_dir / "dev_loss")
    print("reptile k5 chain: dev_loss", dl[0], "->", dl[-1], "dev_acc", da[-3:])
    assert dl[-1][1] < 0.3 * dl[0][1] and np.median([a for _, a in da[-5:]]) >= 0.75
    assert len(lines) == 12 and (ft_dir / "model.wer.best").exists()
