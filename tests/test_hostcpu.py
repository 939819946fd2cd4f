"""Host-thread budget (transformer4sed_amd/hostcpu.py): the cgroup / affinity arithmetic and the import-time cap."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_usable_cpus_within_affinity():
    from transformer4sed_amd.hostcpu import usable_cpus
    n = usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def _child(env_extra):
    code = "import torch; a = torch.get_num_threads(); import transformer4sed_amd.ops; print(a, torch.get_num_threads())"
    env = {k: v for k, v in os.environ.items() if k not in ("SED_HOST_THREADS", "OMP_NUM_THREADS")}
    env.update(env_extra, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [int(v) for v in r.stdout.split()[-2:]]


def test_import_caps_the_intra_op_pool_and_honours_overrides():
    from transformer4sed_amd.hostcpu import usable_cpus
    before, after = _child({})
    assert after == min(before, max(1, usable_cpus() // 4))
    assert _child({"SED_HOST_THREADS": "3"})[1] == 3
    before, after = _child({"SED_HOST_THREADS": "0"})
    assert after == before
    before, after = _child({"OMP_NUM_THREADS": "5"})
    assert (before, after) == (5, 5)
