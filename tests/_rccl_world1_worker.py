"""Worker of tests/test_hip_rccl_world1.py: TaskSharder's exchange through the nccl backend (= RCCL) with ONE rank on the one GPU
of the box -- process-group init with device_id, all-reduce on the side stream behind the main stream's producer, work.wait() +
wait_stream before the consumer, several reductions pending at once, and the issued-from-the-main-stream flavour."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import torch
import torch.distributed as dist

import masr_amd  # noqa: F401
from masr_amd.parallel import TaskSharder


def main():
    os.environ["MASR_FORCE_COLLECTIVE"] = "1"
    TaskSharder.init_process_group()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    sh = TaskSharder.from_env()
    assert sh.collective and sh.world == 1 and sh.backend == "nccl"
    dev = torch.device("cuda:0")
    n = 24_881_455                                                  # the hkust meta-gradient: 99.5 MB
    base = torch.arange(n, device=dev, dtype=torch.float32) * 1e-3
    for side in (True, False):
        bufs = []
        for k in range(3):                                          # three rounds in flight, each produced on the main stream right before
            b = torch.empty_like(base)
            torch.mul(base, float(k + 1), out=b)                    # producer on the main stream: the exchange must wait for it
            sh.reduce_async(b, side_stream=side)
            bufs.append(b)
            junk = base * 2.0                                       # main-stream work queued behind the (asynchronous) exchange
        assert len(sh._pending) == 3
        sh.wait_all()
        total = bufs[0] + bufs[1] + bufs[2]                         # consumer on the main stream
        torch.cuda.synchronize()
        assert torch.equal(total, base * 1.0 + base * 2.0 + base * 3.0), f"side_stream={side}: reduced buffers differ"
        del junk
    t = sh.all_reduce_scalar_sum(3.5)
    assert t == 3.5
    sh.barrier()
    print("rccl-world1-ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
