#!/usr/bin/env python
"""
What clock and board power do the lossless kernels run at?  Samples the amdgpu hwmon files (sclk, power, power cap) every
few ms from a thread while a phase loops one kernel for ~2.5 s, and prints per-phase means next to the kernel's time.
    python tools/clock_power_probe.py            (GPU box)
"""
import glob
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hwmon_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for k in ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap", "power1_cap_max", "temp1_input"):
            p = os.path.join(d, k)
            if os.path.isfile(p):
                out.setdefault(d, {})[k] = p
    return out


def rd(p):
    try:
        with open(p) as f:
            return float(f.read().strip())
    except Exception:
        return float("nan")


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            self.rows.append((t,) + tuple(rd(p) for p in self.files))
            time.sleep(0.004)


def main():
    import torch

    torch.cuda.init()
    import bench
    from magphase_amd import engine as em

    for cmd in (["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--showperflevel", "--showmemuse"],):
        try:
            print(subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout[-3000:], flush=True)
        except Exception as e:
            print("rocm-smi failed:", e)
    hw = hwmon_files()
    if not hw:
        return
    # the card this process computes on: PCI address of HIP device 0 -> its hwmon directory
    pr = torch.cuda.get_device_properties(0)
    bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    mine = [d for d in hw if bus in os.path.realpath(d)]
    print("HIP device 0 = PCI %s -> %s" % (bus, mine), flush=True)
    d = mine[0] if mine else sorted(hw)[0]
    keys = [k for k in ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap") if k in hw[d]]
    files = [hw[d][k] for k in keys]
    for k in hw[d]:
        print(k, rd(hw[d][k]))
    for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") + glob.glob("/sys/class/drm/card*/device/pp_dpm_mclk") + \
            glob.glob("/sys/class/drm/card*/device/power_dpm_force_performance_level"):
        try:
            print(f, open(f).read().replace("\n", " | "))
        except Exception as e:
            print(f, e)

    torch.cuda.set_device(0)
    utts = bench.make_batch(0)
    eng = em.Engine()
    aplan = em.LosslessAnalysisPlan(eng, utts)
    splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
    H, F = aplan.fft_len // 2 + 1, aplan.total_frames
    feats = tuple(eng.empty_feats(F, H) for _ in range(3))
    strips = eng.empty((max(splan.strip_floats, 1) + 65536,))
    pcm = eng.empty((splan.total_out,))
    a = lambda: aplan.run(out=feats)
    s = lambda: splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
    big2 = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    x = torch.randn(8192, 8192, device="cuda")

    def copy():
        big2.copy_(big)

    def mm():
        torch.mm(x, x)

    phases = [("idle", None), ("analysis", a), ("synthesis", s), ("analysis+synthesis", lambda: (a(), s())), ("torch copy 1GiB", copy),
              ("torch fp32 mm 8192", mm), ("synthesis again", s), ("idle", None)]
    a(); s(); torch.cuda.synchronize()
    for name, fn in phases:
        smp = Sampler(files)
        smp.start()
        t0 = time.perf_counter()
        n = 0
        if fn is None:
            time.sleep(1.0)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            while time.perf_counter() - t0 < 2.5:
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                n += 20
                last = e0.elapsed_time(e1) / 20
        t1 = time.perf_counter()
        smp.stop = True
        smp.join()
        rows = [r for r in smp.rows if r[0] - t0 > 0.5 * (t1 - t0)]   # second half: past the ramp
        means = [sum(r[i + 1] for r in rows) / max(len(rows), 1) for i in range(len(keys))]
        first = [smp.rows[min(5, len(smp.rows) - 1)][i + 1] for i in range(len(keys))]
        print("%-22s %s  calls %d  ms/call(last 20) %s  | early: %s" % (
            name, "  ".join("%s=%.4g" % (k, m) for k, m in zip(keys, means)), n,
            ("%.4f" % last) if fn else "-", " ".join("%.4g" % v for v in first)), flush=True)
        time.sleep(0.5)


if __name__ == "__main__":
    main()
