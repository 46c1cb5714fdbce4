"""Host-side environment facts the harnesses need: how many CPUs this process may actually use.

`os.cpu_count()` reports the machine (256 hardware threads on the MI355X hosts of the pool) while the container's cgroup grants 16 CPUs of quota.
PyTorch sizes its intra-op pool from the former (128 threads), and one CPU-side tensor op per pass -- e.g. materialising an expanded coordinate grid --
then spins 128 OpenMP workers against a 16-CPU quota: the kernel throttles the whole process for the rest of the 100 ms period and the GPU queue
runs dry (measured in round 6, tools/exp/pass_jitter.py: LINF-LP passes of 21 ms each took 21 / 57 / 96 ms at random, +7 s of throttling in 40
passes; with the pool capped: 21 ms flat).  The engines no longer run CPU tensor ops per pass (linf/prep.py caches its grids on the device);
`cap_torch_threads()` is what bench.py, the tools and the tests call so that the CPU baseline and any host-side glue stay inside the quota."""
import os


def effective_cpus():
    """CPUs this process can use: min(affinity mask, cgroup CPU quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:                                                            # cgroup v2: "<quota> <period>" or "max <period>"
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:                                                        # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def cap_torch_threads(limit=None):
    """torch.set_num_threads(min(current, effective_cpus(), limit)); returns the thread count in force."""
    import torch
    n = min(torch.get_num_threads(), effective_cpus())
    if limit is not None:
        n = min(n, int(limit))
    n = max(1, n)
    if n != torch.get_num_threads():
        torch.set_num_threads(n)
    return n
