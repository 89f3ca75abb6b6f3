"""Shader clock / socket power sampling while a workload runs (bench.py's `sclk_mhz_under_load` / `power_w`, tools/power_trace.py).

MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): an MFMA-dense kernel on real data runs at
1.8 - 2.0 GHz instead of the 2.4 GHz the dense peak is quoted at, so a roofline fraction needs the clock it was measured at.
Sources, first one that answers: the amdsmi python binding (gpu_metrics table: per-XCD gfx clocks + socket power, ~1 ms per
read), then the `rocm-smi` CLI (~0.2 s per read).  Host-side only; nothing here touches the kernels."""
import re
import subprocess
import threading
import time


class _AmdSmiSource:
    name = "amdsmi"

    def __init__(self, index=0):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.h = amdsmi.amdsmi_get_processor_handles()[index]
        self.read()                                     # raises when the binding cannot read this device

    @staticmethod
    def _num(v):
        try:
            v = float(v)
        except (TypeError, ValueError):
            return None
        return v if 0 < v < 60000 else None             # 0xFFFF / "N/A" sentinels

    def read(self):
        sclk = power = None
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
            clks = [self._num(c) for c in (m.get("current_gfxclks") or [])]
            clks = [c for c in clks if c]
            sclk = sum(clks) / len(clks) if clks else self._num(m.get("current_gfxclk"))
            power = self._num(m.get("current_socket_power")) or self._num(m.get("average_socket_power"))
        except Exception:
            pass
        if sclk is None:
            sclk = self._num(self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX).get("clk"))
        if power is None:
            p = self.smi.amdsmi_get_power_info(self.h)
            power = self._num(p.get("current_socket_power")) or self._num(p.get("average_socket_power")) or self._num(p.get("socket_power"))
        if sclk is None and power is None:
            raise RuntimeError("amdsmi returned neither a clock nor a power reading")
        return sclk, power


class _RocmSmiSource:
    name = "rocm-smi"

    def __init__(self, index=0):
        self.index = index
        self.read()

    def read(self):
        txt = subprocess.run(["rocm-smi", "-d", str(self.index), "-c", "-P"], capture_output=True, text=True, timeout=5).stdout
        sclk = re.search(r"sclk clock level:?\s*\S*:?\s*\(?(\d+)Mhz", txt)
        pw = re.search(r"Power \(W\):\s*([\d.]+)", txt)
        if not sclk and not pw:
            raise RuntimeError("rocm-smi output not understood")
        return (float(sclk.group(1)) if sclk else None), (float(pw.group(1)) if pw else None)


def open_source(index=0):
    for cls in (_AmdSmiSource, _RocmSmiSource):
        try:
            return cls(index)
        except Exception:
            continue
    return None


class Sampler:
    """`with Sampler() as s: work()` -> s.summary(): median / min / max of the shader clock (MHz) and socket power (W) over the
    samples taken while the block ran (the first `skip_s` seconds are dropped: clock ramp of an idle chip)."""

    def __init__(self, period=0.05, index=0, skip_s=0.0):
        self.period, self.skip_s = period, skip_s
        self.src = open_source(index)
        self.samples = []                                # (t, sclk MHz, power W)
        self._stop = threading.Event()
        self._th = None

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            try:
                sclk, power = self.src.read()
                self.samples.append((time.perf_counter() - t0, sclk, power))
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        if self.src is not None:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join()
        return False

    def summary(self):
        if self.src is None:
            return {"source": None}
        rows = [s for s in self.samples if s[0] >= self.skip_s] or self.samples

        def stat(i):
            v = sorted(x[i] for x in rows if x[i] is not None)
            return None if not v else {"median": v[len(v) // 2], "min": v[0], "max": v[-1]}

        return {"source": self.src.name, "samples": len(rows), "period_s": self.period, "sclk_mhz": stat(1), "power_w": stat(2)}
