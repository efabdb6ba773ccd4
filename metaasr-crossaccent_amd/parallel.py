"""Multi-GPU meta-training: accent-tasks sharded one-per-GPU, meta-gradient reduced with RCCL.

The reference is single-process.  FOMAML's tasks inside one meta-step are independent given the
meta weights (src/fo_meta_interface.py:139-156), so rank r runs the tasks  task_ids[r::world]  of the meta-batch
and the only exchange is ONE all-reduce(sum) of the flat meta-gradient (24.88 M fp32 = 99.5 MB for hkust), after
which every rank applies the identical Noam-Adam step (replicated state, no broadcast).

Overlap (SURVEY 8(e) option (i)): when a rank owns >= 2 tasks of a meta-step, each task's clipped gradient is
all-reduced on a side HIP stream while the next task's inner loop runs on the main stream -- all tasks of a
meta-step read the same meta weights, so nothing is stale.  xGMI is point-to-point, so the payload goes out as a
few large chunks (ring all-reduce is per-link bound) rather than per-tensor buckets.

Backends: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU tensors (tests, world_size 2).
"""
import os

import torch


class TaskSharder:
    def __init__(self, rank=0, world=1, backend=None, collective=None):
        self.rank, self.world, self.backend = rank, world, backend
        # does the meta-gradient go through the collective?  Always with several ranks; with ONE rank only when a process group
        # exists and MASR_FORCE_COLLECTIVE=1 asks for it (an all-reduce over one rank is the identity): that runs RCCL init, the
        # side-stream ordering and work.wait() on a single MI355X (tests/test_hip_rccl_world1.py)
        self.collective = (world > 1) if collective is None else bool(collective)
        self._side = None
        self._pending = []
        # Transport of the exchange.  Default: ProcessGroupNCCL's all_reduce (= RCCL) on a side stream.  OPT-IN (MASR_NATIVE_ALLREDUCE=1):
        # the C ABI's own exchange (include/masr.h masr_allreduce: librccl bound directly, our own side stream and event ordering,
        # clip-scale pipelined with the collective); torch.distributed then only carries the control plane (the communicator id,
        # barriers, gathered eval numbers).  It stays opt-in until a run on two or more physical GPUs has shown its meta weights
        # bit-equal to the default transport's: so far it has only ever run with one rank (tests/test_hip_rccl_world1.py).
        self.native = self.collective and backend == "nccl" and os.environ.get("MASR_NATIVE_ALLREDUCE") == "1"
        self.transport_note = ""                                 # why the transport is what it is (bench line: meta_step.transport)
        self.nchunks = int(os.environ.get("MASR_ALLREDUCE_CHUNKS", "4"))
        self.timeout_s = float(os.environ.get("MASR_ALLREDUCE_TIMEOUT_S", "300"))
        self._comm = None
        self._native_issued = False

    @classmethod
    def from_env(cls):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size()
            return cls(dist.get_rank(), world, dist.get_backend(), collective=world > 1 or os.environ.get("MASR_FORCE_COLLECTIVE") == "1")
        return cls()

    @staticmethod
    def init_process_group(backend=None):
        """one process per GPU; reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment."""
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if (world <= 1 and os.environ.get("MASR_FORCE_COLLECTIVE") != "1") or dist.is_initialized():
            return
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("MASR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)

    # ------------------------------------------------------------------ task assignment
    def owns(self, pos):
        """is the task at position `pos` of the (rank-identical) shuffled meta-batch this rank's?  Round-robin by
        position keeps the ranks balanced for any meta_batch_size; which accent lands where changes every meta-step,
        so shards are opened / uploaded lazily by the rank that first runs them (io/dataset.py)."""
        return pos % self.world == self.rank

    def my_tasks(self, meta_batch):
        return [t for pos, t in enumerate(meta_batch) if self.owns(pos)]

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()

    # ------------------------------------------------------------------ meta-gradient exchange
    def _side_stream(self, device):
        if self._side is None and device.type == "cuda":
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def _native_comm(self, device):
        """the C-ABI communicator of this rank (made on first use: rank 0 draws the RCCL id, the process group ships its 128 bytes)"""
        if self._comm is None:
            import ctypes as C
            import torch.distributed as dist
            from . import _cabi
            L = _cabi.lib()
            torch.cuda.set_device(device)
            box = [None]
            if self.rank == 0:
                raw = C.create_string_buffer(128)
                _cabi.check(L.masr_allreduce_unique_id(raw), "masr_allreduce_unique_id")
                box[0] = raw.raw
            if self.world > 1:
                dist.broadcast_object_list(box, src=0)
            comm = L.masr_allreduce_init(self.rank, self.world, box[0]) if box[0] else None
            if comm and os.environ.get("MASR_TEST_FAIL_ALLREDUCE_INIT") == "1":      # (tests: the agreed fallback below)
                L.masr_allreduce_destroy(comm)
                comm = None
            # every rank must take the same transport: agree on the outcome (a rank whose init failed would otherwise wait for ever in
            # the process group's collective while the others sit in ours)
            ok = torch.tensor([1 if comm else 0], device=device, dtype=torch.int32)
            if self.world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                err = L.masr_last_error().decode()
                if comm:
                    L.masr_allreduce_destroy(comm)
                import warnings
                warnings.warn(f"masr_allreduce_init failed on some rank ({err!r}): the meta-gradient goes through ProcessGroupNCCL instead")
                self.native = False
                self.transport_note = f"native init failed ({err})"
                return None
            self._comm = comm
            self._L = L
        return self._comm

    @property
    def transport(self):
        """what carries the meta-gradient: 'none' (one rank, no collective), 'native' (masr_allreduce), 'pg_nccl', 'pg_gloo'"""
        if not self.collective:
            return "none"
        return "native" if self.native else f"pg_{self.backend}"

    def watchdog(self):
        """host-side health check of the native exchange, called where a new exchange is about to be issued: RCCL's asynchronous error
        state, and a bounded wait for the PREVIOUS exchange (long complete in a healthy run, so this costs nothing; a collective that
        never completes would otherwise hang the job silently -- the waits of wait_all are device-side).  Raises: the caller lets the
        exception end the process, so the launcher tears the other ranks down."""
        if self._comm is not None:
            from . import _cabi
            _cabi.check(self._L.masr_allreduce_check(self._comm, int(self.timeout_s * 1000)), "masr_allreduce_check")

    def close(self):
        """end of train() (normal or SIGINT): drain and destroy the native communicator"""
        if self._comm is not None:
            self._L.masr_allreduce_destroy(self._comm)
            self._comm = None
            self._native_issued = False

    def reduce_async(self, buf, side_stream=True, clip=None):
        """all-reduce(sum) `buf` in place, overlapped with whatever the caller enqueues next on the main stream.
        The caller must not touch `buf` until wait_all().  side_stream=False: issued from the current stream (RCCL still runs it on
        its own stream) -- for callers that already keep four streams busy, where one more stream with work queued displaces a
        task's and nothing is left to overlap with anyway.
        clip=(address of a device float holding buf's L2 norm, max_norm): buf is scaled by clip_grad_norm_'s coefficient on the
        way out, chunk by chunk beside the collective (native path; elsewhere the scale pass runs first)."""
        if not self.collective:
            return
        comm = self._native_comm(buf.device) if self.native and buf.device.type == "cuda" else None
        if comm is not None:
            from . import _cabi
            if not self._native_issued:                          # first exchange of a meta-step: the previous step's must be done by now
                self.watchdog()
            norm, max_norm = clip if clip is not None else (None, 0.0)
            _cabi.check(self._L.masr_allreduce(comm, buf.data_ptr(), buf.numel(), norm, max_norm, self.nchunks,
                                               torch.cuda.current_stream(buf.device).cuda_stream), "masr_allreduce")
            self._native_issued = True
            return
        if clip is not None:
            # (the native exchange was asked for and is not available: the scale pass runs first, on the producer stream)
            from . import _cabi
            norm, max_norm = clip
            L = _cabi.lib()
            import ctypes as C
            _cabi.check(L.masr_clip_scale_flat(buf.data_ptr(), buf.numel(), C.c_void_p(norm), max_norm,
                                               torch.cuda.current_stream(buf.device).cuda_stream), "masr_clip_scale_flat")
        import torch.distributed as dist
        if buf.device.type == "cuda" and self.backend == "gloo":
            # rehearsal of the multi-rank path on a box with fewer GPUs than ranks (gloo has no device collectives worth the
            # name): through host memory, synchronously.  The RCCL path below is the product.
            torch.cuda.current_stream(buf.device).synchronize()
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)
            return
        if buf.device.type == "cuda" and not side_stream:
            self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), None))
        elif buf.device.type == "cuda":
            side = self._side_stream(buf.device)
            side.wait_stream(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(side):
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
            self._pending.append((work, side))
        else:
            self._pending.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), None))

    def wait_all(self):
        if self._native_issued:                                  # a device-side wait: the host does not block
            from . import _cabi
            _cabi.check(self._L.masr_allreduce_wait(self._comm, torch.cuda.current_stream().cuda_stream), "masr_allreduce_wait")
            self._native_issued = False
        for work, side in self._pending:
            work.wait()
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
        self._pending = []

    def all_reduce(self, buf):
        self.reduce_async(buf)
        self.wait_all()

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def all_reduce_scalar_sum(self, x: float) -> float:
        if self.world == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        dist.all_reduce(t)
        return float(t.item())
