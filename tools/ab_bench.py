#!/usr/bin/env python
"""
Within-process A/B timing of kernel / planner versions (guide rule: perf deltas < 10 % need interleaved rounds in ONE
process).  A variant is a complete copy of the package (python + csrc + include) plus extra hipcc flags:

    NAME             the working tree
    NAME:-DFOO,-DBAR the working tree built with extra flags
    NAME@REF         the tree of git ref REF (e.g. prev@HEAD~2), optionally NAME@REF:-DFOO

    python tools/ab_bench.py --prepare prev@HEAD cur        # HERE (needs git + hipcc): snapshots + builds under tools/_ab/
    python tools/ab_bench.py prev@HEAD cur                  # on the GPU box: loads tools/_ab/*, times interleaved

Timed per variant and round, each bracketed by HIP events on the launch stream: the lossless analysis launch
(aplan.run), the fused synthesis (splan.run = k_synth_ola_pair + k_ola_fixup) -- the bench.py workload.  Reports
median / min per variant.  AB_WORKLOAD=lowdim times the configs[2] analysis (k_analysis + warp) and synthesis instead.
"""
import importlib
import importlib.util
import os
import shutil
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AB = os.path.join(ROOT, "tools", "_ab")


def parse(spec):
    flags = []
    if ":" in spec:
        spec, fl = spec.split(":", 1)
        flags = [f for f in fl.split(",") if f]
    name, ref = (spec.split("@", 1) + [None])[:2]
    return name, ref, flags


def prepare(name, ref, flags):
    dst = os.path.join(AB, name)
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    if ref:
        tar = subprocess.Popen(["git", "-C", ROOT, "archive", ref, "magphase_amd", "include"], stdout=subprocess.PIPE)
        subprocess.check_call(["tar", "-x", "-C", dst], stdin=tar.stdout)
        tar.wait()
    else:
        for d in ("magphase_amd", "include"):
            shutil.copytree(os.path.join(ROOT, d), os.path.join(dst, d),
                            ignore=shutil.ignore_patterns("__pycache__", "*.so", "*.pyc", "_obj"))
    csrc = os.path.join(dst, "magphase_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".cpp"))
            and f != "magphase_pyhost.cpp"]   # (the marshalling extension is not part of the C-ABI library)
    from magphase_amd import build

    if not ref:   # the working tree: per-unit objects, compiled in parallel and cached per flag set (magphase_amd/_obj)
        build.build(extra_flags=flags, out=os.path.join(dst, "magphase_amd", "libmagphase_hip.so"), verbose=False)
        print("built %s %s" % (name, " ".join(flags)), flush=True)
        return
    cmd = [build.hipcc_path()] + build.FLAGS + flags + srcs + ["-o", os.path.join(dst, "magphase_amd", "libmagphase_hip.so")]
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def load(name):
    pkg_dir = os.path.join(AB, name, "magphase_amd")
    alias = "mpa_" + name
    spec = importlib.util.spec_from_file_location(alias, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    return importlib.import_module(alias + ".engine")


def main():
    args = sys.argv[1:]
    if args and args[0] == "--prepare":
        for spec in args[1:]:
            prepare(*parse(spec))
        return
    import torch

    import bench

    names = [parse(a)[0] for a in args] or ["cur"]
    rounds = int(os.environ.get("AB_ROUNDS", "15"))
    lowdim = os.environ.get("AB_WORKLOAD", "lossless") == "lowdim"
    extract = os.environ.get("AB_WORKLOAD", "lossless") == "extract"   # configs[3] kernel: fused compressed analysis, 60 / 10 and 60 / 45
    roundtrip = os.environ.get("AB_WORKLOAD", "lossless") == "roundtrip"   # the one-launch copy synthesis (first column)
    torch.cuda.set_device(0)
    utts = bench.make_batch(0)
    if os.environ.get("AB_FS"):   # other sample rates / batch sizes: AB_FS=16000 AB_UTTS=128 (the N = 2048 kernels)
        from magphase_amd import synthetic as syn
        fs_ = int(os.environ["AB_FS"])
        utts = []
        for i in range(int(os.environ.get("AB_UTTS", "64"))):
            pcm_, pm_, voi_ = syn.make_utterance(i, dur_s=5.0, fs=fs_)
            utts.append((pcm_, fs_, pm_, voi_))
    steps = {}
    shared = None   # ONE set of feature matrices / strips / output for all variants: where they land in memory is worth
    # +-3-5 % by itself (tools/archive/alloc_lottery_probe.py) -- per-variant buffers turned that into a fake A/B difference
    for name in names:
        em = load(name)
        eng = em.Engine()
        if extract:
            # AB_ENV_<name>="K=V,...": environment of this variant's plan construction (e.g. MAGPHASE_FUSED_ANALYSIS=f32)
            extra = dict(kv.split("=", 1) for kv in os.environ.get("AB_ENV_" + name, "").split(",") if "=" in kv)
            saved = {k: os.environ.get(k) for k in extra}
            os.environ.update(extra)
            plans = [em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=pd, alpha_phase=False) for pd in (10, 45)]
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            outs = [pl.run() for pl in plans]
            assert all(pl.fused for pl in plans)
            steps[name] = (lambda plans=plans, outs=outs: plans[0].run(out=outs[0]),
                           lambda plans=plans, outs=outs: plans[1].run(out=outs[1]))
        elif not lowdim:
            aplan = em.LosslessAnalysisPlan(eng, utts)
            splan = em.LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
            H, F = aplan.fft_len // 2 + 1, aplan.total_frames
            print("%s: fft_len %d, frames %d, feature bytes %.3f GB, slots %d" % (name, aplan.fft_len, F, 12e-9 * H * F, splan.n_slots), flush=True)
            if shared is None:
                shared = (tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((splan.total_out,)))
            feats, pcm = shared
            strips = eng.empty((max(splan.strip_floats, 1) + 65536,))   # per variant: its size follows the variant's slot count
            steps[name] = (lambda aplan=aplan, feats=feats: aplan.run(out=feats),
                           lambda splan=splan, feats=feats, strips=strips, pcm=pcm: splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm))
            if roundtrip:
                rt = em.LosslessRoundTripPlan(eng, utts)
                rstrips = eng.empty((max(rt.synthesis.strip_floats, 1) + 65536,))
                steps[name] = (lambda rt=rt, feats=feats, rstrips=rstrips, pcm=pcm: rt.run(feats=feats, strips=rstrips, out=pcm),
                               steps[name][1])
        else:
            if shared is None:
                shared = {}
            # AB_ENV_<name>="K=V,K2=V2": environment of this variant's plan construction (e.g. MAGPHASE_UNWARP_BF16=0)
            extra = dict(kv.split("=", 1) for kv in os.environ.get("AB_ENV_" + name, "").split(",") if "=" in kv)
            saved = {k: os.environ.get(k) for k in extra}
            os.environ.update(extra)
            sa, ss = bench.lowdim_plans(em, eng, utts, shared)
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            steps[name] = (sa, ss)
    times = {n: ([], [], []) for n in names}
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for r in range(rounds + 2):
        for name in names:
            ev[0].record()
            steps[name][0]()
            ev[1].record()
            steps[name][1]()
            ev[2].record()
            steps[name][1]()      # again: reads not preceded by the analysis' 1.4 GB of writes (dirty lines in the Infinity Cache)
            ev[3].record()
            torch.cuda.synchronize()
            if r >= 2:
                for k in range(3):
                    times[name][k].append(ev[k].elapsed_time(ev[k + 1]))
    if os.environ.get("AB_ZERO") and not lowdim:   # DVFS check: the same synthesis launches on all-zero / constant features
        for fill in (0.0, 1.0):
            for t_ in shared[0]:
                t_.fill_(fill)
            for name in names:
                ts = []
                for r in range(rounds):
                    ev[0].record()
                    steps[name][1]()
                    ev[1].record()
                    torch.cuda.synchronize()
                    ts.append(ev[0].elapsed_time(ev[1]))
                print("%-14s synthesis on features == %.0f: %.4f / %.4f ms" % (name, fill, statistics.median(ts), min(ts)))
    print("%-14s %22s %22s %22s  (ms: median / min over %d interleaved rounds)"
          % ("variant", "analysis", "synthesis (+fixup)", "synthesis repeated", rounds))
    for name in names:
        t = times[name]
        print("%-14s " % name + " ".join("%10.4f /%9.4f" % (statistics.median(x), min(x)) for x in t))


if __name__ == "__main__":
    main()
