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
    native = os.environ.get("MASR_NATIVE_ALLREDUCE") == "1"
    sh = TaskSharder.from_env()
    assert sh.collective and sh.world == 1 and sh.backend == "nccl" and sh.native == native
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
        assert (sh._native_issued and not sh._pending) if native else len(sh._pending) == 3
        sh.wait_all()
        total = bufs[0] + bufs[1] + bufs[2]                         # consumer on the main stream
        torch.cuda.synchronize()
        assert torch.equal(total, base * 1.0 + base * 2.0 + base * 3.0), f"side_stream={side}: reduced buffers differ"
        del junk
    if native:
        # ---- the C ABI's own features (include/masr.h masr_allreduce): clip_grad_norm_'s scale pipelined with the collective, chunk
        # by chunk, == scale first, then reduce; any chunk count, a length that is no multiple of anything
        import ctypes as C
        from masr_amd import _cabi
        L = _cabi.lib()
        g = torch.Generator(device=dev).manual_seed(3)
        for n_, chunk_counts in ((n, (1, 4, 16)), (1_000_003, (1, 5)), (777, (1, 3))):
            x = torch.randn(n_, device=dev, generator=g) * 0.01
            norm = torch.linalg.vector_norm(x.double()).float().reshape(1)
            for max_norm in (1e4, 5.0, 0.5 * float(norm), 0.0013):  # no clipping / clipping
                want = x * torch.clamp(max_norm / (norm + 1e-6), max=1.0)
                outs = []
                for chunks in chunk_counts:
                    got = x.clone()
                    sh.nchunks = chunks
                    sh.reduce_async(got, clip=(norm.data_ptr(), max_norm))
                    sh.wait_all()
                    torch.cuda.synchronize()
                    outs.append(got)
                    # the coefficient is one fp32 division: within an ulp of torch's, whatever the chunking
                    torch.testing.assert_close(got, want, rtol=3e-7, atol=0.0)
                for o in outs[1:]:                                  # chunked == one pass, bit for bit
                    assert torch.equal(o, outs[0]), (n_, chunk_counts, max_norm)
                if max_norm == 1e4:
                    assert torch.equal(outs[0], x)                  # coefficient clamped to exactly 1
        nan = torch.full((1,), float("nan"), device=dev)            # a NaN norm poisons the buffer, as torch's clip does (quirk Q5)
        y = torch.ones(4096, device=dev)
        sh.reduce_async(y, clip=(nan.data_ptr(), 5.0)); sh.wait_all(); torch.cuda.synchronize()
        assert bool(torch.isnan(y).all())
        # bad arguments are refused with a message, nothing is launched
        assert L.masr_allreduce(sh._comm, None, 10, None, 0.0, 1, None) != 0 and b"masr_allreduce" in L.masr_last_error()
        assert not L.masr_allreduce_init(3, 2, b"x" * 128) and b"masr_allreduce_init" in L.masr_last_error()
        # the host-side health check: nothing pending -> 0 at once; an exchange in flight -> 1 (one poll) and 0 within the time limit
        assert sh.transport == "native" and L.masr_allreduce_check(sh._comm, 0) == 0
        big = torch.ones(n, device=dev)
        sh.reduce_async(big); sh.wait_all()
        assert L.masr_allreduce_check(sh._comm, 0) in (0, 1) and L.masr_allreduce_check(sh._comm, 60000) == 0
        sh.watchdog()
        assert L.masr_allreduce_check(None, 0) != 0
        sh.close()
        assert sh._comm is None
    t = sh.all_reduce_scalar_sum(3.5)
    assert t == 3.5
    sh.barrier()
    print("rccl-world1-ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
