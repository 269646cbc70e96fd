"""Board telemetry from the amdgpu driver's hwmon files (sysfs: no HIP call, no child process) — used by bench.py inside its timed region and by the entry points'
measurement knobs.  Not on any compute path."""
import os


class PowerSampler:
    """Board power and shader clock of the device DURING the timed steps, read from the amdgpu driver's hwmon files (sysfs: no HIP call, no child process; a few
    microseconds per read, every 20 ms from a thread).  Why it is in the line: the headline step runs into the board's power cap (round 6: 1.35-1.37 kW of
    1.4 kW, shader clock 2.0 GHz against the 2.4 GHz the MFMA peak is quoted at) — the roofline fraction is quoted against the datasheet peak, the clock the
    part could hold says how much of the distance is the power limit.  Every figure is None where the files are not readable."""

    def __init__(self, torch, device):
        import glob
        import threading
        self.rows, self._stop, self._th, self.cap = [], threading.Event(), None, None
        self.power_f = self.freq_f = None
        try:
            pr = torch.cuda.get_device_properties(device)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hw:
                for name in ("power1_average", "power1_input"):
                    if os.path.exists(os.path.join(hw[0], name)):
                        self.power_f = os.path.join(hw[0], name)
                        break
                if os.path.exists(os.path.join(hw[0], "freq1_input")):
                    self.freq_f = os.path.join(hw[0], "freq1_input")
                if os.path.exists(os.path.join(hw[0], "power1_cap")):
                    self.cap = int(open(os.path.join(hw[0], "power1_cap")).read()) / 1e6
        except Exception:
            pass
        self._threading = threading

    def _read(self, path):
        try:
            with open(path) as f:
                return int(f.read())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            self.rows.append((self._read(self.power_f) if self.power_f else None, self._read(self.freq_f) if self.freq_f else None))
            self._stop.wait(0.02)

    def start(self):
        if self.power_f or self.freq_f:
            self._th = self._threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=1.0)
        pw = [p / 1e6 for p, _ in self.rows if p]
        fq = [f / 1e6 for _, f in self.rows if f]
        if not pw and not fq:
            return None
        return {"samples": len(self.rows), "power_W": {"mean": round(sum(pw) / len(pw), 1), "max": round(max(pw), 1)} if pw else None, "power_cap_W": self.cap,
                "sclk_MHz": {"mean": round(sum(fq) / len(fq)), "min": round(min(fq)), "max": round(max(fq))} if fq else None,
                "source": "amdgpu hwmon (power1_average / freq1_input) sampled every 20 ms inside the timed region"}
