"""Clock / power samples of GPU 0 for the measurement tools (rocm-smi JSON; every field optional).

sample() -> {"sclk_mhz", "mclk_mhz", "fclk_mhz", "socclk_mhz", "power_w", "temp_c", ...} with whatever
the installed rocm-smi reports; {} if the tool is missing or fails (never raises)."""
import json
import re
import shutil
import subprocess
import threading
import time


def _num(v):
    m = re.search(r"-?\d+(\.\d+)?", str(v))
    return float(m.group(0)) if m else None


def sample(device=0, timeout=10):
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([exe, "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=timeout).stdout
        card = next(iter(json.loads(out).values()))
    except Exception:
        return {}
    res = {}
    for k, v in card.items():
        kl = k.lower()
        n = _num(v)
        if n is None:
            continue
        if "clock level" in kl:
            continue                              # "sclk clock level: 1" — the speed is in "... clock speed: (2100Mhz)"
        if "sclk" in kl:
            res["sclk_mhz"] = n
        elif "mclk" in kl:
            res["mclk_mhz"] = n
        elif "fclk" in kl:
            res["fclk_mhz"] = n
        elif "socclk" in kl:
            res["socclk_mhz"] = n
        elif "power" in kl and "w" in kl:
            res.setdefault("power_w", n)
        elif "temperature" in kl and "junction" in kl:
            res["temp_junction_c"] = n
        elif "temperature" in kl and ("memory" in kl or "hbm" in kl):
            res.setdefault("temp_hbm_c", n)
        elif "temperature" in kl:
            res.setdefault("temp_c", n)
    return res


class Sampler(object):
    """Background sampling at `period` seconds: with Sampler(0.25) as s: ...; s.rows -> [(t, sample), ...]"""

    def __init__(self, period=0.25, device=0):
        self.period, self.device, self.rows = period, device, []
        self._stop = threading.Event()
        self._t0 = time.perf_counter()

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter() - self._t0
            s = sample(self.device)
            self.rows.append((round(t, 3), s))
            self._stop.wait(self.period)

    def __enter__(self):
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join(timeout=15)

    def now(self):
        return time.perf_counter() - self._t0
