"""Host CPU budget of this process.  On a container whose cgroup CPU quota is far below the machine's core count (the MI355X boxes
here: 256 hardware threads visible, quota 16) torch sizes its intra-op OpenMP pool from the visible count (128 threads).  Every
host-side tensor op above the parallel grain then wakes 128 spinning threads, the cgroup burns its quota in a few milliseconds and
the kernel throttles the WHOLE process for the rest of the 100 ms period -- the Python thread that feeds the GPU included.  Measured
on the masked-reconstruction step: host issue time 37-118 ms/step with the default pool, 7.8 ms/step with one thread (the GPU needs
33 ms).  The host side of this package only touches tiny tensors (a 128 x 513 filter bank, a 32 x 128 gain table), so the TRAINING
ENTRY POINTS (`MatSedTrainer`, `PmamTrainer`, `bench.py`) cap the pool when they are constructed -- importing the package, building a
model or running inference / evaluation never touches the host process's threading.  SED_HOST_THREADS overrides (0 = leave torch's
setting alone), an explicit OMP_NUM_THREADS is respected, and the value chosen is logged once."""
import logging
import os

log = logging.getLogger("transformer4sed_amd")


def usable_cpus():
    """CPUs this process may actually use: min(affinity mask, cgroup v2 / v1 CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:            # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


_capped = False


def cap_torch_threads():
    global _capped
    if _capped:
        return
    _capped = True
    want = os.environ.get("SED_HOST_THREADS") or None
    if want is None and "OMP_NUM_THREADS" in os.environ:
        return
    import torch
    if want is not None:
        try:
            n = int(want)
        except ValueError:
            raise ValueError(f"SED_HOST_THREADS must be an integer (0 = leave torch's thread pool alone), got {want!r}") from None
        if n > 0:
            torch.set_num_threads(n)
            log.info("host threads: torch intra-op pool set to %d (SED_HOST_THREADS)", n)
        return
    cap = max(1, usable_cpus() // 4)
    if torch.get_num_threads() > cap:
        log.info("host threads: torch intra-op pool %d -> %d (usable CPUs %d; SED_HOST_THREADS=0 keeps torch's setting)",
                 torch.get_num_threads(), cap, usable_cpus())
        torch.set_num_threads(cap)


def recommended_env():
    """Process environment an entry point should set BEFORE HIP initialises (transformer4sed_amd/__init__.py): os.environ.setdefault(k, v)
    for each item.  bench.py and tests/conftest.py do."""
    return {"GPU_MAX_HW_QUEUES": "8"}

