"""BASELINE configs[2] on the GPU: `pretrain.py --algo fomaml`, 4 accents, inner_steps (meta_k) = 1 -- and a meta_k = 2 variant --
with the SHIPPED Noam schedule (warmup_steps 25000), through get_trainer(FOMetaASRInterface...).exec() with evaluate() ON.

Checked against the reference run captured in tests/golden/fomaml_cfg3.npz (oracle/make_goldens.py::gen_fomaml_cfg3_goldens):
  * same batches in the same order (train and eval calls);
  * EVERY run_batch loss -- inner steps, the val-batch CE after k inner steps (the north-star quantity), dev batches --
    within 1e-3 relative; accuracies within one token;
  * evaluate(): per-accent and average dev loss/acc/cer/wer log lines, best_wer / best_cer, model.wer.best, the file set;
and against the oracle (pinned to the same golden at 2e-5 by tests/test_oracle_golden.py) run beside it on the CPU:
  * the meta-gradient of every meta-step, tensor by tensor;
  * the meta weights after every Noam-Adam step, tensor by tensor.
Reference: src/fo_meta_interface.py:128-298, src/transformer_torch_trainer.py:59-99."""
import random
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402,F401
from masr_amd.fo_meta_interface import FOMetaASRInterface  # noqa: E402
from masr_amd.io.dataset import DataContainer  # noqa: E402
from masr_amd.transformer_torch_trainer import get_trainer  # noqa: E402
from oracle import ref_cpu  # noqa: E402
from oracle.make_goldens import ODIM, CFG3_ACCENTS, cfg3_paras, cfg3_workspace  # noqa: E402
from replay import oracle_fomaml_run  # noqa: E402


def product_run(tmp_path, golden_dir, meta_k, tasks_per_gpu=1):
    cfg = cfg3_workspace(tmp_path, golden_dir)
    # njobs 2: batches are assembled by the collate pool and the next meta-step is drawn ahead -- the golden's call order and
    # batch contents below prove that this changes neither the RNG consumption nor the data
    paras = cfg3_paras(meta_k, device="cuda:0", tasks_per_gpu=tasks_per_gpu, cuda=True, no_cuda=False, njobs=2)
    random.seed(531); np.random.seed(531); torch.manual_seed(531)
    solver = get_trainer(FOMetaASRInterface, cfg, paras, dict(CFG3_ACCENTS + [("ca", "canada")]))
    solver.load_data()
    solver.set_model()
    solver.asr_model.load_state_dict(ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7))
    solver.load_model()
    calls, steps = [], []
    orig = solver.run_batch

    def spy(idx, x, ilens, ys, olens, train, accent_idx=None, **kw):
        rec = (int(idx), bool(train), ilens.clone(), [y.clone() for y in ys])
        kw["want_info"] = True                                   # (run_task skips the host sync of its inner steps; the check wants every loss)
        info = orig(idx, x, ilens, ys, olens, train=train, accent_idx=accent_idx, **kw)
        calls.append(rec + (dict(info),))
        return info
    solver._train = partial(spy, train=True)
    solver._eval = partial(spy, train=False)
    orig_final = solver._final_meta_update

    def final_spy(n_tasks=None):
        if solver._updates is not None:
            mg = (solver._updates / solver._counter).cpu()
        else:                                                      # fused meta update: the tasks' gradient buffers, summed in task order
            acc = torch.zeros_like(solver._task_grads[0])
            for g_ in solver._task_grads:
                acc += g_
            mg = (acc / solver._counter).cpu()
        orig_final(n_tasks)
        torch.cuda.synchronize()
        steps.append((mg, solver._original.cpu().clone()))
    solver._final_meta_update = final_spy
    solver.exec()
    torch.cuda.synchronize()
    return cfg, solver, calls, steps


@pytest.mark.parametrize("meta_k", [1, 2])
def test_cfg3_run_matches_reference_and_oracle(golden_dir, tmp_path, monkeypatch, meta_k):
    g = np.load(golden_dir / "fomaml_cfg3.npz")
    pre = f"k{meta_k}/"
    monkeypatch.chdir(tmp_path)
    cfg, solver, calls, steps = product_run(tmp_path, golden_dir, meta_k)
    eng = solver.asr_model.engine

    # ---- every call: same batch, loss within 1e-3 relative (north-star), accuracy within one token
    assert len(calls) == int(g[pre + "n_calls"]), (len(calls), int(g[pre + "n_calls"]))
    worst = {True: 0.0, False: 0.0}
    for i, (accent, train, il, ys, info) in enumerate(calls):
        assert accent == int(g[f"{pre}call{i}/accent"]) and int(train) == int(g[f"{pre}call{i}/train"]), i
        np.testing.assert_array_equal(il.numpy(), g[f"{pre}call{i}/ilens"])
        np.testing.assert_array_equal(np.concatenate([y.numpy() for y in ys]), g[f"{pre}call{i}/ys"])
        ref = float(g[f"{pre}call{i}/loss"])
        rel = abs(info["loss"] - ref) / ref
        worst[train] = max(worst[train], rel)
        assert rel <= 1e-3, f"call {i} ({'train' if train else 'eval'}): loss {info['loss']} vs reference {ref} (rel {rel:.2e})"
        n_tok = sum(len(y) + 1 for y in ys)
        assert abs(info["acc"] - float(g[f"{pre}call{i}/acc"])) <= 1.0 / n_tok + 1e-6
    print(f"meta_k {meta_k}: {len(calls)} run_batch calls; worst relative loss error train {worst[True]:.2e}, eval {worst[False]:.2e}")

    # ---- evaluate(): log files line by line (loss 1e-3; acc one token of the 4-utterance dev batch; cer/wer from arg-max tokens)
    files = sorted(p.name for p in solver.log_dir.iterdir())
    ref_files = [str(f) for f in g[pre + "files"]]
    # ours adds meta_state.latest (resume extension) and dashboard.jsonl (stand-in for the comet.ml dashboard)
    assert [f for f in files if f not in ("meta_state.latest", "dashboard.jsonl")] == ref_files, (files, ref_files)
    for key in g.files:
        if not key.startswith(pre + "log/"):
            continue
        name = key[len(pre) + 4:]
        ours, ref = (solver.log_dir / name).read_text().split("\n"), str(g[key]).split("\n")
        assert len(ours) == len(ref), name
        for lo, lr_ in zip(ours, ref):
            if not lo and not lr_:
                continue
            so, sr = lo.split(), lr_.split()
            assert so[0] == sr[0], (name, lo, lr_)                      # same global step
            if len(sr) == 1:
                continue
            vo, vr = float(so[1]), float(sr[1])
            if name.endswith("_loss"):
                assert abs(vo - vr) <= 1e-3 * abs(vr), (name, lo, lr_)
            elif name.endswith("_acc"):
                assert abs(vo - vr) <= 0.06, (name, lo, lr_)
            else:                                                       # cer / wer / best_*: percent of edit distance over text
                assert abs(vo - vr) <= 0.02 * abs(vr) + 1e-9, (name, lo, lr_)
    assert int(g[pre + "global_step"]) == solver.global_step
    assert int(g[pre + "meta/step_num"]) == solver.meta_opt.step_num
    assert abs(float(g[pre + "meta/lr"]) - solver.meta_opt.lr) <= 1e-15
    best = torch.load(solver.log_dir / "model.wer.best")
    assert list(best.keys()) == list(eng.state_dict().keys())

    # ---- the oracle beside it (fp32, and with the engine's bf16 rounding points emulated): meta-gradients and meta weights
    sv = cfg["solver"]

    def oracle(emulate):
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        dc = DataContainer([tmp_path / "data" / a for _, a in CFG3_ACCENTS], batch_size=sv["batch_size"], dev_batch_size=sv["dev_batch_size"],
                           is_memmap=True, is_bucket=True, min_ilen=sv["min_ilen"], max_ilen=sv["max_ilen"], half_batch_ilen=sv["half_batch_ilen"])
        if emulate:
            with ref_cpu.bf16_emulation():
                return oracle_fomaml_run(cfg, dc, meta_k, 4, 5, sv["label_smoothing"])
        return oracle_fomaml_run(cfg, dc, meta_k, 4, 5, sv["label_smoothing"])
    o32, o16 = oracle(False), oracle(True)
    assert len(steps) == len(o32["steps"]) == 4
    names = [n for n in eng.table]
    sd0 = ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7)
    lr_sum, report = 0.0, []
    for si, (mg_flat, meta_flat) in enumerate(steps):
        lr_sum += ref_cpu.noam_lr(si + 1, 1.0, cfg["asr_model"]["d_model"], 25000)
        dots = np.zeros(3)
        for n in names:
            off, shape = eng.table[n]
            k = int(np.prod(shape))
            mine = mg_flat[off:off + k].view(shape).double()
            r32, r16 = o32["steps"][si][0][n].double(), o16["steps"][si][0][n].double()
            e32 = float((mine - r32).norm() / r32.norm())
            e16 = float((mine - r16).norm() / r16.norm())
            dots += [float((mine * r32).sum()), float((mine * mine).sum()), float((r32 * r32).sum())]
            report.append((si, n, e32, e16))
            # in_proj_bias: its key third has an exactly-zero true gradient (softmax shift invariance) -- rounding noise only
            if not n.endswith("in_proj_bias"):
                assert e32 < 0.15, f"meta-gradient step {si} {n}: rel-L2 {e32:.3f} vs the fp32 oracle"
                assert e16 < 0.08, f"meta-gradient step {si} {n}: rel-L2 {e16:.3f} vs the bf16-emulating oracle"
            # meta weights after Adam: each element moves by at most ~lr per step whatever the gradient's size, so the two
            # trajectories can differ by at most 2 * sum(lr) per element
            w_mine = meta_flat[off:off + k].view(shape)
            w_ref = o32["steps"][si][1][n]
            ulp = float(w_ref.abs().max()) * 2.0 ** -23              # (lr 3e-8 is below one fp32 ulp of a LayerNorm gain of 1.0)
            assert float((w_mine - w_ref).abs().max()) <= 2.5 * lr_sum + ulp, (si, n)
            assert float((w_mine - w_ref).norm()) <= 1e-5 * float(w_ref.norm()) + 1e-9, (si, n)
        cos = dots[0] / np.sqrt(dots[1] * dots[2])
        ratio = np.sqrt(dots[1] / dots[2])
        print(f"  meta-step {si}: whole meta-gradient cosine vs fp32 oracle {cos:.5f}, norm ratio {ratio:.4f}")
        assert cos > 0.995 and abs(ratio - 1) < 0.02
    report.sort(key=lambda r: -r[3])
    print("  largest per-tensor meta-gradient rel-L2 (vs fp32 oracle | vs bf16-emulating oracle):")
    for si, n, e32, e16 in report[:6]:
        print(f"    step {si} {n}: {e32:.4f} | {e16:.4f}")
    # the update direction of the meta weights over the whole run, per tensor (Adam normalises each element's step to ~lr, so
    # this is the fraction of elements whose gradient sign agrees, weighted by nothing: the harshest view of bf16 noise)
    final = steps[-1][1]
    agree = []
    for n in names:
        off, shape = eng.table[n]
        k = int(np.prod(shape))
        du = final[off:off + k].view(shape) - sd0[n]
        dr = o32["steps"][-1][1][n] - sd0[n]
        moved = (dr != 0) | (du != 0)
        if int(moved.sum()) == 0:
            continue
        agree.append((float((torch.sign(du) == torch.sign(dr))[moved].double().mean()), n))
    agree.sort()
    print("  lowest per-tensor sign agreement of the accumulated meta update:", [(n, round(a, 3)) for a, n in agree[:5]])
    assert np.mean([a for a, n in agree if not n.endswith("in_proj_bias")]) > 0.9


def test_cfg3_concurrent_slots_bit_identical(golden_dir, tmp_path, monkeypatch):
    """the same run with --tasks_per_gpu 4 (all four accent-tasks of a meta-step concurrently, one replica + stream + host
    thread each): meta weights and dev logs bit-identical to the sequential run"""
    monkeypatch.chdir(tmp_path)
    outs = []
    for k in (1, 4):
        cfg, solver, calls, steps = product_run(tmp_path, golden_dir, 1, tasks_per_gpu=k)
        outs.append((steps[-1][1], (solver.log_dir / "train_loss").read_text(), sorted(round(c[4]["loss"], 6) for c in calls if c[1])))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][2] == outs[1][2]


@pytest.mark.parametrize("k", [1, 4])
def test_host_running_ahead_books_the_same_stats(golden_dir, tmp_path, monkeypatch, k):
    """default: the task stats come back asynchronously and are booked a meta-step later (the host queues the next meta-step
    meanwhile); --sync_stats: read back per task.  Same meta weights bit for bit, same running averages, same train_* / dev_* log
    files (evaluate() runs for real after meta-step 4, the first reader of train_info in this run)."""
    monkeypatch.chdir(tmp_path)
    runs = []
    for sync in (True, False):
        cfg = cfg3_workspace(tmp_path, golden_dir)
        cfg["solver"].update(log_ival=1000, eval_ival=4, save_ival=1000)
        paras = cfg3_paras(1, device="cuda:0", tasks_per_gpu=k, cuda=True, no_cuda=False, njobs=2, sync_stats=sync)
        random.seed(531); np.random.seed(531); torch.manual_seed(531)
        solver = get_trainer(FOMetaASRInterface, cfg, paras, dict(CFG3_ACCENTS + [("ca", "canada")]))
        solver.load_data(); solver.set_model()
        solver.asr_model.load_state_dict(ref_cpu.deterministic_state_dict(cfg["asr_model"], ODIM, seed=7))
        solver.load_model()
        made, booked_at = [], []
        orig = solver._clip_and_stats

        def clip_and_stats(info, engine=None):
            h = orig(info, engine=engine)
            made.append(type(h).__name__)
            return h
        solver._clip_and_stats = clip_and_stats
        orig_add = solver.train_info.add
        solver.train_info.add = lambda info, n=1: (booked_at.append(solver.global_step), orig_add(info, n))[1]
        solver.exec()
        torch.cuda.synchronize()
        logs = {p.name: p.read_text() for p in sorted(solver.log_dir.iterdir()) if p.name.startswith(("train_", "dev_"))}
        runs.append((solver._original.clone(), dict(solver.train_info), logs, set(made), booked_at))
        assert not solver._pending
    assert runs[0][3] == {"_Resolved"} and runs[1][3] == {"_H"}             # the two paths were really taken
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2] and any(n.startswith("dev_") for n in runs[0][2])
    assert runs[1][4] == runs[0][4] and runs[1][4][:4] == [2, 2, 2, 2]      # meta-step 1's four tasks are booked during meta-step 2
