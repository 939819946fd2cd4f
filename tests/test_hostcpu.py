"""Host-thread budget (transformer4sed_amd/hostcpu.py): the cgroup / affinity arithmetic and the cap the training entry points apply."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_usable_cpus_within_affinity():
    from transformer4sed_amd.hostcpu import usable_cpus
    n = usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def _child(env_extra):
    code = ("import torch; a = torch.get_num_threads(); import transformer4sed_amd.ops, transformer4sed_amd.trainer; "
            "b = torch.get_num_threads(); from transformer4sed_amd.hostcpu import cap_torch_threads; cap_torch_threads(); "
            "print(a, b, torch.get_num_threads())")
    env = {k: v for k, v in os.environ.items() if k not in ("SED_HOST_THREADS", "OMP_NUM_THREADS")}
    env.update(env_extra, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [int(v) for v in r.stdout.split()[-3:]]


def test_import_leaves_threads_alone_entry_points_cap_and_honour_overrides():
    from transformer4sed_amd.hostcpu import usable_cpus
    before, imported, after = _child({})
    assert imported == before, "importing the package must not touch the host process's thread pool"
    assert after == min(before, max(1, usable_cpus() // 4))
    assert _child({"SED_HOST_THREADS": "3"})[2] == 3
    before, _, after = _child({"SED_HOST_THREADS": "0"})
    assert after == before
    assert _child({"OMP_NUM_THREADS": "5"}) == [5, 5, 5]
