"""Host helpers of the CLIs (reference: src/utils.py:8-38 -- usable CPU count for `--njobs`)."""
import os


def usable_cpus():
    """cores this process may actually use: the affinity mask capped by the cgroup CPU quota.  A GPU box hands one GPU's job a
    16-core share of a much larger host; sizing thread pools (collate pool, torch intra-op threads) by the visible core count
    oversubscribes that share many times over (measured: 36 ms instead of 1.7 ms per collated batch)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def setup_host_threads(njobs):
    """torch CPU ops are plumbing here (index tensors, checkpoint I/O): keep their thread pool within the job's share"""
    import torch
    torch.set_num_threads(max(1, min(int(njobs), usable_cpus(), 16)))
