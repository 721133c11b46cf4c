"""CPU time this process is really allowed to use, and torch's CPU thread pool sized to it.

Found in round 4 (``profiles/r04_experiments.txt``, "host CPU quota"): the MI355X boxes show 256 CPUs, but the process lives in a
cgroup with a CFS quota of 16 CPUs.  torch sizes its intra-op pool from the CPU count (128 threads); every small CPU tensor
operation of the host code between launches wakes that pool, its workers spin for a while after the parallel region, the
cgroup's quota for the 100 ms period is burnt within a few milliseconds and EVERY thread of the process -- the one that
launches kernels included -- is frozen until the next period: launches and host waits that sporadically take 5 ... 85 ms
(quantised by the scheduler's slices), training iterations that take 150-290 ms instead of a reproducible 90-125.  Nothing
on the device is slow; the host is stopped.  ``cap_host_threads`` is called when the package is imported.
"""
from __future__ import annotations

import math
import os


def cpu_quota() -> float:
    """CPUs' worth of time per wall second this process may consume: the smaller of its affinity mask and its cgroup's
    CFS quota (cgroup v2 ``cpu.max``, v1 ``cpu.cfs_quota_us`` / ``cpu.cfs_period_us``; no quota = the affinity mask)."""
    try:
        n = float(len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        n = float(os.cpu_count() or 1)
    try:
        rel = "/"
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
        # the limit of the process's own group and of every ancestor applies (walk up inside the mounted hierarchy)
        path = os.path.normpath("/sys/fs/cgroup/" + rel)
        while path.startswith("/sys/fs/cgroup"):
            f = os.path.join(path, "cpu.max")
            if os.path.exists(f):
                quota, period = open(f).read().split()[:2]
                if quota != "max":
                    n = min(n, float(quota) / float(period))
            if path == "/sys/fs/cgroup":
                break
            path = os.path.dirname(path)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, q / p)
    except (OSError, ValueError):
        pass
    return max(n, 1.0)


def cpu_throttle_stats() -> dict:
    """The cgroup's CFS counters of this process (cgroup v2 ``cpu.stat``, v1 ``cpu/cpu.stat``): {'nr_periods', 'nr_throttled',
    'throttled_usec', 'usage_usec'} -- whatever the file has; {} when it cannot be read.  ``nr_throttled`` moving during a
    timed region means the kernel froze every thread of the process (the launch thread included) for the rest of a period:
    the failure an over-subscribed multi-rank run meets first (bench.py prints the difference around its timed region)."""
    out = {}
    paths = []
    try:
        rel = "/"
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
        paths.append(os.path.normpath("/sys/fs/cgroup/" + rel + "/cpu.stat"))
    except OSError:
        pass
    paths += ["/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"]
    for f in paths:
        try:
            for line in open(f):
                k, _, v = line.strip().partition(" ")
                if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
                    out[k] = int(v)
            if out:
                return out
        except (OSError, ValueError):
            continue
    return out


def cap_host_threads() -> int:
    """torch's intra-op CPU threads = at most HALF this process's share of the quota (the launch thread, the HIP runtime's
    helpers and the other streams' launch threads need the rest; the ranks of a node -- LOCAL_WORLD_SIZE, set by
    torch.distributed.run -- share one quota), never more than torch chose itself (OMP_NUM_THREADS is respected).
    ``A3D_HOST_THREADS=<n>`` forces n, ``A3D_HOST_THREADS=0`` leaves torch alone.  Returns the thread count in force."""
    import torch
    forced = os.environ.get("A3D_HOST_THREADS")
    cur = torch.get_num_threads()
    if forced is not None:
        try:
            want = int(forced)
        except ValueError:       # a malformed value must not make `import agile3d_amd` raise: say so, fall back to the policy
            import warnings
            warnings.warn(f"agile3d_amd: A3D_HOST_THREADS={forced!r} is not an integer; using the quota-based default")
            forced = None
    if forced is not None:
        if want <= 0:
            return cur
    if forced is None:
        try:
            ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
        except ValueError:
            ranks = 1
        want = min(cur, max(1, int(math.floor(cpu_quota() / (2 * ranks)))))
    if want != cur:
        torch.set_num_threads(want)
    return want
