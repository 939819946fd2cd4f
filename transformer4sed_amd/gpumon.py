"""Shader clock / socket power of the GPU while a benchmark's timed region runs (bench.py `gpu_state`).

The MI355X clocks to its power budget (MI355X_MICROARCH.md "DVFS give-back"): the MFMA-bound GEMMs of the train step run at 1.6-2.0 GHz
of the 2.4 GHz maximum, which is what `roofline.frac` against the 2.4 GHz peak pays first.  This module puts the evidence into the bench
line: a sampler thread reads the amdgpu hwmon files (`freq1_input` = sclk in Hz, `power1_average` / `power1_input` in microwatts) of
the device every `period` seconds -- two small file reads, no subprocess -- and reports min / mean / max over the timed steps.  The card
is found by the HIP device's PCI address (a node lists all its GPUs in sysfs whatever the process may use).  Nothing here is needed by
the product path; a box without the files yields `{"source": None}`."""
import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _pci_address(index):
    """'dddd:bb:dd.f' of HIP device `index` (torch.cuda device properties), or None."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None


def _sysfs_device(index):
    """hwmon directory of HIP device `index`: the AMD card (PCI vendor 0x1002) whose sysfs device node is the HIP device's PCI address.
    A host exposes every GPU of the node in /sys/class/drm (plus one platform card per compute partition) even when the process sees one
    device -- card order says nothing; without a PCI match nothing is reported rather than another GPU's idle clock."""
    want = _pci_address(index)
    if want is None:
        return None
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(dev, "vendor")) != "0x1002":
            continue
        if os.path.basename(os.path.realpath(dev)).lower() == want:
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            return hw[0] if hw else None
    return None


class GpuSampler:
    def __init__(self, index=0, period=0.05):
        self.index, self.period = index, period
        self.hw = _sysfs_device(index)
        self.sclk, self.power, self.t = [], [], []
        self._stop = threading.Event()
        self._thr = None
        self.source = None
        if self.hw is not None and (_read(os.path.join(self.hw, "freq1_input")) or _read(os.path.join(self.hw, "power1_average"))
                                    or _read(os.path.join(self.hw, "power1_input"))):
            self.source = "sysfs hwmon of " + str(_pci_address(index)) + " (freq1_input, power1_input)"
        # (no rocm-smi fallback: its device index is the node's, not the process's -- it would report another GPU)

    # one sample: (sclk MHz or None, power W or None)
    def _sysfs(self):
        f = _read(os.path.join(self.hw, "freq1_input"))
        p = _read(os.path.join(self.hw, "power1_average")) or _read(os.path.join(self.hw, "power1_input"))
        return (float(f) / 1e6 if f else None, float(p) / 1e6 if p else None)

    def _loop(self):
        while not self._stop.is_set():
            s = self._sysfs()
            if s is not None:
                self.t.append(time.perf_counter())
                self.sclk.append(s[0])
                self.power.append(s[1])
            self._stop.wait(self.period)

    def start(self):
        if self.source is not None:
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thr is not None:
            self._thr.join(timeout=10)
        return self.summary()

    def summary(self):
        def stat(xs, nd):
            xs = [x for x in xs if x is not None]
            return None if not xs else {"min": round(min(xs), nd), "mean": round(sum(xs) / len(xs), nd), "max": round(max(xs), nd)}
        return {"source": self.source, "samples": len(self.t), "period_s": self.period, "sclk_MHz": stat(self.sclk, 0),
                "socket_power_W": stat(self.power, 1), "sclk_max_MHz": 2400,
                "note": "sampled over the timed steps only; the GEMM roofline's 2.5 PFLOP/s peak is quoted at 2400 MHz"}
