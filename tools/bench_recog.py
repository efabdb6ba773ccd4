"""Time masr_recog (KV-cached, hipGraph) against masr_recog_full (reference schedule) on the hkust model.
usage: python tools/bench_recog.py [B] [T]"""
import sys, time
import torch
sys.path.insert(0, ".")
import masr_amd  # noqa
from masr_amd.engine import MasrEngine
from masr_amd.model import reference_init_state_dict

HKUST = {"idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.1, "pos_dropout": 0.1, "tgt_share_weight": 1,
         "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
torch.manual_seed(531)
eng = MasrEngine(HKUST, 367)
eng.load_state_dict(reference_init_state_dict(HKUST, 367))
xs = torch.randn(B, T, 80, device="cuda")
il = torch.full((B,), T, dtype=torch.int64)
side = torch.cuda.Stream()


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


def cached_graph():
    with torch.cuda.stream(side):
        return eng.recog(xs, il)


ms_g, rg = timed(cached_graph, 5)
ms_d, rd = timed(lambda: eng.recog(xs, il), 3)
ms_f, rf = timed(lambda: eng.recog(xs, il, full=True), 1)
same = (rg == rf)
prefix = torch.cummin(same.int(), dim=0).values.sum(0)
print(f"B={B} T={T} Ldec={T // 4}: cached+graph {ms_g:.1f} ms, cached direct {ms_d:.1f} ms, full re-decode {ms_f:.1f} ms "
      f"(x{ms_f / ms_g:.1f}); graph==direct {bool((rg == rd).all())}; identical prefix vs full: min {int(prefix.min())} mean {float(prefix.float().mean()):.1f}")
