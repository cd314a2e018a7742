#!/usr/bin/env python
"""
Energy per launch of k_synth_ola_pair and of its ablations: each variant loops for ~2.5 s while a thread samples the board
power of THIS device (hwmon power1_input); energy = mean power x time per launch.

    for v in full "noload:-DMPX_ABL_NOLOAD" ...; do python tools/ab_bench.py --prepare $v; done     # here
    python tools/energy_probe.py full noload noola nofft nomerge ...                                # GPU box
"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def my_hwmon(torch):
    pr = torch.cuda.get_device_properties(0)
    bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if bus in os.path.realpath(d):
            return d
    return None


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            vals = []
            for p in self.files:
                try:
                    vals.append(float(open(p).read()))
                except Exception:
                    vals.append(float("nan"))
            self.rows.append((time.perf_counter(),) + tuple(vals))
            time.sleep(0.004)


def measure(torch, files, fn, seconds=2.5, batch=20):
    smp = Sampler(files)
    smp.start()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    while time.perf_counter() - t0 < seconds:
        e0.record()
        for _ in range(batch):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / batch)
    t1 = time.perf_counter()
    smp.stop = True
    smp.join()
    rows = [r for r in smp.rows if r[0] - t0 > 0.5 * (t1 - t0)]
    mean = [sum(r[i + 1] for r in rows) / max(len(rows), 1) for i in range(len(files))]
    half = ms[len(ms) // 2:]
    return sum(half) / len(half), mean


def main():
    import torch

    torch.cuda.init()
    import ab_bench
    import bench

    d = my_hwmon(torch)
    files = [os.path.join(d, "power1_input"), os.path.join(d, "freq1_input")]
    names = sys.argv[1:] or ["cur"]
    utts = bench.make_batch(0)
    shared = None
    time.sleep(1.0)
    idle = [float(open(f).read()) for f in files]
    print("idle: %.0f W, sclk %.0f MHz" % (idle[0] / 1e6, idle[1] / 1e6), flush=True)
    print("%-12s %9s %9s %9s %11s %11s" % ("variant", "ms", "W", "sclk MHz", "J/launch", "J - idle"))
    for name in names:
        em = ab_bench.load(name)
        eng = em.Engine()
        aplan = em.LosslessAnalysisPlan(eng, utts)
        splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
        H, F = aplan.fft_len // 2 + 1, aplan.total_frames
        if shared is None:
            shared = (tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((splan.total_out,)))
        feats, pcm = shared
        strips = eng.empty((max(splan.strip_floats, 1) + 65536,))
        aplan.run(out=feats)
        torch.cuda.synchronize()
        for what, fn in (("synthesis", lambda: splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm)),) + (
                (("analysis", lambda: aplan.run(out=feats)),) if name == names[0] else ()):
            ms, (pw, fq) = measure(torch, files, fn)
            print("%-12s %9.4f %9.0f %9.0f %11.4f %11.4f   %s" % (name, ms, pw / 1e6, fq / 1e6, pw / 1e6 * ms / 1e3,
                                                                 (pw - idle[0]) / 1e6 * ms / 1e3, what), flush=True)
            time.sleep(0.3)


if __name__ == "__main__":
    main()
