"""--resume continues a run EXACTLY (VERDICT r2 missing #3): N steps, stop, resume, N more == 2N uninterrupted.

The reference cannot resume pretraining at all (it asserts an optimizer.latest that pretraining never writes, SURVEY section 5 /
quirk Q3) and restarts its data stream when a mono run resumes.  Here `snapshot.latest` + `meta_state.latest` (pretrain) /
`optimizer.latest` (mono) also hold the meta weights, the optimiser state, the three RNG streams, the samplers' bucket
arrangement and cursor, the best-so-far error rates and the dropout streams' positions, so that the continued run draws the same
batches in the same task order and -- the kernels being deterministic -- ends on bit-identical weights, optimiser state and log
files.  Reference: src/pretrain_interface.py:71-98, src/fo_meta_interface.py:56-111, src/mono_interface.py:34-73,83-94."""
import pickle

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

import masr_amd  # noqa: E402,F401
from oracle.make_goldens import cfg3_workspace, chain_workspace  # noqa: E402

SKIP = ("dashboard.jsonl", "exp_key")


def _text_files(d):
    out = {}
    for p in sorted(d.iterdir()):
        if p.is_file() and p.name not in SKIP:
            try:
                out[p.name] = p.read_text()
            except UnicodeDecodeError:
                pass
    return out


def _dedup(text):
    seen, out = set(), []
    for l in text.splitlines():
        if l not in seen:
            seen.add(l); out.append(l)
    return out


def _assert_state_equal(a, b, path=""):
    if isinstance(a, torch.Tensor):
        assert torch.equal(a, b), f"{path}: tensors differ (max |d| {float((a.double() - b.double()).abs().max())})"
    elif isinstance(a, dict):
        assert a.keys() == b.keys(), path
        for k in a:
            _assert_state_equal(a[k], b[k], f"{path}/{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _assert_state_equal(x, y, f"{path}[{i}]")
    elif hasattr(a, "dtype") and hasattr(a, "shape"):                      # numpy
        assert (a == b).all(), path
    else:
        assert a == b, (path, a, b)


@pytest.mark.parametrize("algo,eval_ival,save_ival,stop,total,tasks_per_gpu", [
    ("fomaml", 2, 3, 3, 7, 1),        # the stop coincides with a checkpoint at the end of a chunk: every file must be identical
    ("fomaml", 3, 5, 7, 10, 2),       # checkpoint in the MIDDLE of an eval chunk, two more steps run after it; two task slots
    ("multi", 2, 3, 3, 7, 1),
])
def test_pretrain_resume_is_an_exact_continuation(golden_dir, tmp_path, monkeypatch, algo, eval_ival, save_ival, stop, total, tasks_per_gpu):
    import pretrain
    monkeypatch.chdir(tmp_path)
    cfg = cfg3_workspace(tmp_path, golden_dir)
    cfg["solver"].update(eval_ival=eval_ival, save_ival=save_ival)
    if algo == "multi":
        m = cfg["asr_model"]
        for k in ("inner_optimizer_cls", "inner_optimizer_opt", "meta_opt_cls", "meta"):
            m.pop(k)
        m.update({"optimizer_cls": "noam", "optimizer_opt": {"k": 1.0, "warmup_steps": 20}})
    yaml.safe_dump(cfg, open(tmp_path / "cfg.yaml", "w"))

    def cli(suffix, max_step, *extra):
        pretrain.main(["--config", "cfg.yaml", "--pretrain_suffix", suffix, "--pretrain_accents", "af", "au", "en", "us", "--num_pretrain", "4",
                       "--tgt_accent", "ca", "--algo", algo, "--meta_k", "1", "--meta_batch_size", "4", "--max_step", str(max_step), "--njobs", "2",
                       "--tasks_per_gpu", str(tasks_per_gpu), *extra])
        torch.cuda.synchronize()
        return tmp_path / "testing-logs" / "pretrain" / "cfg3" / algo / suffix / "canada" / "0"
    full = cli("full", total, "--overwrite")
    part = cli("part", stop, "--overwrite")
    saved_at = int((part / "global_step").read_text())
    assert saved_at == (stop // save_ival) * save_ival and saved_at < total
    part = cli("part", total, "--resume")
    assert int((part / "global_step").read_text()) == int((full / "global_step").read_text())
    for name in ("snapshot.latest", "meta_state.latest", f"snapshot.step.{(total // save_ival) * save_ival}", "model.wer.best"):
        a, b = torch.load(full / name, weights_only=False), torch.load(part / name, weights_only=False)
        _assert_state_equal(a, b, name)
    assert pickle.load(open(full / "info_dict.latest", "rb")) == pickle.load(open(part / "info_dict.latest", "rb"))
    fa, fb = _text_files(full), _text_files(part)
    assert fa.keys() == fb.keys()
    for name in fa:
        if saved_at == stop:
            assert fa[name] == fb[name], name
        else:                                   # steps saved_at .. stop-1 ran twice (before the stop and again after the resume): same lines twice
            assert fa[name].splitlines() == _dedup(fb[name]), name
    assert len(fa["train_loss"].splitlines()) >= (total - 1) // eval_ival            # (train_* logs are written at every evaluate())


@pytest.mark.parametrize("opt", ["noam", "SGD"])
def test_mono_resume_is_an_exact_continuation(golden_dir, tmp_path, monkeypatch, opt):
    import train
    monkeypatch.chdir(tmp_path)
    _, ft = chain_workspace(tmp_path, golden_dir)
    for k in ("pretrain_module", "freeze_module"):
        ft["solver"].pop(k)
    ft["solver"].update(eval_ival=20, log_ival=5)
    if opt == "SGD":
        ft["asr_model"].update({"optimizer_cls": "SGD", "optimizer_opt": {"lr": 0.02, "momentum": 0.9, "nesterov": True}})

    def cli(suffix, epochs, *extra):
        ft["solver"]["total_epochs"] = epochs
        yaml.safe_dump(ft, open(tmp_path / "ft.yaml", "w"))
        train.main(["--config", "ft.yaml", "--accent", "ca", "--algo", "no", "--eval_suffix", suffix, "--njobs", "1", *extra])
        torch.cuda.synchronize()
        return tmp_path / "testing-logs" / "evaluation" / "chain-ft" / "no" / suffix / suffix / "canada" / "0"
    full = cli("full", 4, "--overwrite")
    part = cli("part", 2, "--overwrite")
    saved_at = int((part / "global_step").read_text())
    assert int((part / "epoch").read_text()) == 2
    part = cli("part", 4, "--resume")
    assert int((part / "epoch").read_text()) == int((full / "epoch").read_text()) == 4
    assert int((part / "global_step").read_text()) == int((full / "global_step").read_text())
    _assert_state_equal(torch.load(full / "snapshot.latest"), torch.load(part / "snapshot.latest"), "snapshot.latest")
    oa, ob = pickle.load(open(full / "optimizer.latest", "rb")), pickle.load(open(part / "optimizer.latest", "rb"))
    _assert_state_equal(oa["opt"], ob["opt"], "optimizer.latest/opt")
    assert oa["step_num"] == ob["step_num"]
    assert pickle.load(open(full / "info_dict.latest", "rb")) == pickle.load(open(part / "info_dict.latest", "rb"))
    # logs: the resumed run holds every line of the uninterrupted one, in order, plus the lines of the evaluation the reference
    # runs at every start (src/mono_interface.py:131) -- all stamped with the step the run resumed at
    for name in ("train_loss", "train_acc", "dev_loss", "dev_wer"):
        la, lb = (full / name).read_text().splitlines(), (part / name).read_text().splitlines()
        extra = list(lb)
        for l in la:
            assert l in extra, (name, l)
            extra.remove(l)
        assert all(int(l.split()[0]) == saved_at for l in extra) and len(extra) <= 1, (name, extra)
