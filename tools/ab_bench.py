#!/usr/bin/env python
"""
Within-process A/B timing of kernel build variants (guide rule: perf deltas < 10 % need interleaved rounds in ONE
process).  Builds libmagphase_hip variants with extra -D flags into gpurun_out/ab/, loads each with ctypes and
times the three hot-path launches on the bench workload, interleaved, reporting median / min per variant.

    python tools/ab_bench.py base: nostore:-DMPX_PROBE_NOSTORE w8:-DMPX_WAVES_PER_BLOCK=8
"""
import ctypes
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402
from magphase_amd import _lib, build  # noqa: E402
from magphase_amd.engine import Engine, LosslessAnalysisPlan, LosslessSynthesisPlan  # noqa: E402


def build_variant(name, flags):
    out_dir = os.path.join(ROOT, "gpurun_out", "ab")
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libmagphase_hip_%s.so" % name)
    srcs = build.SRCS
    if "PREV" in flags:   # build the snapshot of the previous kernel sources kept (untracked) under tools/_ab_prev
        flags = [f for f in flags if f != "PREV"]
        srcs = [s.replace(ROOT, os.path.join(ROOT, "tools", "_ab_prev")) for s in build.SRCS]
    cmd = [build.hipcc_path()] + build.FLAGS + flags + srcs + ["-o", lib]
    subprocess.check_call(cmd)
    return lib


def main():
    specs = [a.split(":", 1) for a in sys.argv[1:]] or [["base", ""]]
    rounds = int(os.environ.get("AB_ROUNDS", "15"))
    torch.cuda.set_device(0)
    engines = {}
    for name, fl in specs:
        path = build_variant(name, [f for f in fl.split(",") if f])
        _lib._lib = None
        _lib.LIB_PATH = path
        engines[name] = Engine()
    utts = bench.make_batch(0)
    first = engines[specs[0][0]]
    aplan = LosslessAnalysisPlan(first, utts)
    terr = int(os.environ.get("MAGPHASE_OLA_TERRITORY", aplan.fft_len))
    splan = LosslessSynthesisPlan(first, aplan.v_f0, aplan.fs, aplan.fft_len, territory=terr)
    splans = {n: LosslessSynthesisPlan(engines[n], aplan.v_f0, aplan.fs, aplan.fft_len, territory=terr) for n, _ in specs}
    N, H, F = aplan.fft_len, aplan.fft_len // 2 + 1, aplan.total_frames
    feats = tuple(first.empty_feats(F, H) for _ in range(3))
    strips = first.empty((splan.strip_floats,))
    pcm = first.empty((splan.total_out,))
    times = {n: ([], [], []) for n, _ in specs}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for r in range(rounds + 2):
        for name, _ in specs:
            e = engines[name]
            ev[0].record()
            e.analysis_frames(N, aplan.sig, aplan.pos, aplan.left, aplan.right, out=feats)
            ev[1].record()
            e.synthesis_lossless_ola(N, feats[0], feats[1], feats[2], splans[name], strips)
            ev[2].record()
            e.ola_fixup(N, splan.territory, strips, splan.utt_chunk_off, splan.strip_id, splan.out_start,
                        splan.out_off, splan.max_territories, splan.total_out, out=pcm)
            ev[3].record()
            torch.cuda.synchronize()
            if r >= 2:
                for k in range(3):
                    times[name][k].append(ev[k].elapsed_time(ev[k + 1]))
    print("%-14s %22s %22s %22s   (ms: median / min over %d interleaved rounds)" % ("variant", "k_analysis", "k_synth_ola", "k_ola_fixup", rounds))
    for name, _ in specs:
        t = times[name]
        print("%-14s " % name + " ".join("%10.4f /%9.4f" % (statistics.median(x), min(x)) for x in t))


if __name__ == "__main__":
    main()
