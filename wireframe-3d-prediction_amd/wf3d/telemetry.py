"""Board power and shader clock of the card a process drives, from the amdgpu hwmon files (readable by an ordinary user).

The split GEMMs run against the board's power cap, not against an issue limit: the clock the firmware
holds under the cap is part of every roofline statement, so bench.py samples it while the dominant kernel runs."""
import glob
import os
import threading
import time

import torch


def _read(path):
    try:
        with open(path) as f:
            return int(f.read())
    except (OSError, ValueError):
        return None


def hwmon_dir(device=0):
    """hwmon directory of CUDA/HIP device `device`, matched by PCI address; None if sysfs does not show it."""
    try:
        p = torch.cuda.get_device_properties(device)
        want = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
    except (AttributeError, RuntimeError):
        return None
    for d in glob.glob("/sys/class/drm/card*/device"):
        if os.path.basename(os.path.realpath(d)).startswith(want):
            h = sorted(glob.glob(d + "/hwmon/hwmon*"))
            if h:
                return h[0]
    return None


class Sampler:
    """with Sampler(dir) as s: ...; s.watts / s.ghz = means over the last three quarters of the samples taken inside."""

    def __init__(self, hw, period=0.02):
        self.hw, self.period = hw, period
        self._pw, self._fq, self._stop = [], [], False
        self.watts = self.ghz = None

    def _poll(self):
        while not self._stop:
            self._pw.append(_read(self.hw + "/power1_input"))
            self._fq.append(_read(self.hw + "/freq1_input"))
            time.sleep(self.period)

    def __enter__(self):
        if self.hw:
            self._th = threading.Thread(target=self._poll, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self.hw:
            self._th.join()
            pw = [v for v in self._pw[len(self._pw) // 4:] if v]
            fq = [v for v in self._fq[len(self._fq) // 4:] if v]
            self.watts = sum(pw) / len(pw) / 1e6 if pw else None
            self.ghz = sum(fq) / len(fq) / 1e9 if fq else None
        return False


def power_cap_watts(hw):
    v = _read(hw + "/power1_cap") if hw else None
    return v / 1e6 if v else None


def run_sampled(fn, hw, seconds=1.5, batch=20):
    """Runs fn() back to back for `seconds`; returns (seconds per call, mean board W, mean shader GHz)."""
    fn()
    torch.cuda.synchronize()
    with Sampler(hw) as s:
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(batch):
                fn()
            torch.cuda.synchronize()
            n += batch
        el = (time.perf_counter() - t0) / n
    return el, s.watts, s.ghz
