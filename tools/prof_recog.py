"""rocprofv3 target: a few KV-cached decodes (hkust, B=16, T=1000) on a side stream."""
import sys
import torch
sys.path.insert(0, ".")
import masr_amd  # noqa
from masr_amd.engine import MasrEngine
from masr_amd.model import reference_init_state_dict
HKUST = {"idim": 80, "nheads": 8, "d_model": 512, "d_inner": 2048, "dropout": 0.1, "pos_dropout": 0.1, "tgt_share_weight": 1,
         "encoder": {"nlayers": 2}, "decoder": {"nlayers": 4}}
torch.manual_seed(531)
eng = MasrEngine(HKUST, 367)
eng.load_state_dict(reference_init_state_dict(HKUST, 367))
xs = torch.randn(16, 1000, 80, device="cuda")
il = torch.full((16,), 1000, dtype=torch.int64)
side = torch.cuda.Stream()
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
for _ in range(3):
    if mode == "graph":
        with torch.cuda.stream(side):
            eng.recog(xs, il)
    else:
        eng.recog(xs, il, full=(mode == "full"))
torch.cuda.synchronize()
