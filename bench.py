#!/usr/bin/env python
"""
bench.py -- MagPhase hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1], per GPU): 64 synthetic 48 kHz 5 s utterances, lossless analysis +
synthesis, FFT=4096, variable (pitch-synchronous) frame rate.  A step = one pass of the hot path over the batch:
k_analysis -> k_synth_ola_pair -> k_ola_fixup, with PCM and frame descriptors already resident in HBM.  Utterances
shard across ranks with no data-path collective (weak scaling: every rank owns 64 utterances).
Metric: frames/s (whole job) = frames processed by all ranks / max-over-ranks wall time of the K steps.

ONE JSON line.  Next to the headline (metric / value / roofline / cpu_baseline) rank 0 of a 1-GPU run adds
  "configs2": BASELINE configs[2] on the same 64 utterances -- analysis_compressed (mag 60 / phase 45, constant 5 ms
              rate) -> post-filter -> synthesis_from_compressed(b_const_rate): ms per step, per-kernel durations (HIP
              events) with their bound (bytes or fp32-MFMA flops), and its own cpu_baseline;
  "e2e":      what a caller gets -- the numpy-in / numpy-out array API and the file interface (wav + .est files ->
              feature files -> wavs through iobatch), as multiples of real time.
--quick skips configs2 / e2e / the CPU baselines.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

UTTS_PER_GPU = 64
DUR_S = 5.0
FS = 48000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: f32-in / f32-accumulate MFMA = the fp32 vector peak


def make_batch(rank):
    from magphase_amd import synthetic as syn

    utts = []
    for i in range(UTTS_PER_GPU):
        pcm, pm, voi = syn.make_utterance(rank * UTTS_PER_GPU + i, dur_s=DUR_S, fs=FS)
        utts.append((pcm, FS, pm, voi))
    return utts


# ------------------------------------------------------------------------------------------------------------------
# CPU baselines: the oracle (a parity-pinned numpy fp64 port of the reference), timed on this box's host cores
# ------------------------------------------------------------------------------------------------------------------
def _cpu_lossless(u):
    from oracle import magphase_oracle as orc  # checker / CPU baseline only

    pcm, fs, pm, voi = u
    x = pcm.astype(np.float64) / 32768.0
    o = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
    orc.synthesis_from_lossless(o[0], o[1], o[2], o[3], fs)
    return len(o[5])


def _cpu_lowdim(u):
    import warnings

    from oracle import magphase_oracle as orc  # checker / CPU baseline only

    pcm, fs, pm, voi = u
    x = pcm.astype(np.float64) / 32768.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        c = orc.analysis_compressed_from_epochs(x, fs, pm, voi, mag_dim=60, phase_dim=45, b_const_rate=True)
        m = orc.post_filter(c[0], fs)
        np.random.seed(0)
        orc.synthesis_from_compressed(m, c[1], c[2], c[3], fs, b_const_rate=True)
    return int(c[0].shape[0])


def _cpu_init():
    """Pool worker start-up, before the clock: one BLAS / FFT thread per worker (the pool is the parallelism, as in the
    reference's one-utterance-per-process model; N workers x all-core BLAS thrashes) and the oracle imported."""
    try:
        import threadpoolctl

        _cpu_init.limit = threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    from oracle import magphase_oracle  # noqa: F401


def _cpu_warm(_):
    return os.getpid()


def cpu_baseline(utts, fn, what, unit, budget_s=12.0):
    """
    Same parallel model as the reference (libutils.py:32-63: one utterance per multiprocessing.Pool worker).  The pool
    is created and warmed (imports, one BLAS thread per worker) BEFORE the clock starts; it has min(cores, 64) workers and
    every worker gets at least 4 tasks; tasks cycle through the batch's utterances.  Reports the pool rate, the
    per-worker rate inside the pool and the rate of one process alone (which may use BLAS threads).
    """
    import multiprocessing as mpc

    from oracle import magphase_oracle  # noqa: F401  (imported before the clock starts; checker / CPU baseline only)

    ncores = os.cpu_count() or 1
    fn(utts[0])   # untimed: first-call costs (scipy imports, FFT plan caches)
    t0 = time.perf_counter()
    n1, f1 = 0, 0
    while (time.perf_counter() - t0 < budget_s / 3 and n1 < len(utts)) or n1 == 0:   # one process alone
        f1 += fn(utts[n1])
        n1 += 1
    dt1 = time.perf_counter() - t0
    rate1, t_task = f1 / dt1, dt1 / n1
    what_unit = unit.split("/")[0]
    workers = max(1, min(ncores, 64))
    tasks = int(max(4 * workers, min(8 * workers, workers * (budget_s / 2) / t_task)))
    sample = [utts[i % len(utts)] for i in range(tasks)]
    out = {"value": round(rate1, 1), "unit": unit, "cores": 1, "kind": "port",
           "sample": "%s, numpy fp64 oracle, one process: %d utterances (%d %s) in %.1f s" % (what, n1, f1, what_unit, dt1),
           "value_1core": round(rate1, 1)}
    try:
        with mpc.get_context("fork").Pool(workers, initializer=_cpu_init) as pool:
            pool.map(_cpu_warm, range(4 * workers), chunksize=1)
            t0 = time.perf_counter()
            fp = sum(pool.map(fn, sample, chunksize=1))
            dtp = time.perf_counter() - t0
        out.update({"value": round(fp / dtp, 1), "cores": workers, "value_per_worker": round(fp / dtp / workers, 1),
                    "host_cores": ncores,
                    "sample": "%s, numpy fp64 oracle, Pool(%d of %d cores, created and warmed before timing), %d tasks "
                              "(%d per worker, cycling through the %d utterances of the batch) = %d %s in %.1f s; one "
                              "process alone: %.1f %s" % (what, workers, ncores, tasks, tasks // workers, len(utts), fp,
                                                          what_unit, dtp, rate1, unit)})
    except Exception as e:   # no fork / no semaphores: the single-process rate stands
        out["sample"] += " (pool unavailable: %s)" % type(e).__name__
    return out


# ------------------------------------------------------------------------------------------------------------------
# configs[2]: low-dimensional path
# ------------------------------------------------------------------------------------------------------------------
def lowdim_plans(em, eng, utts, shared=None):
    """(analysis step, synthesis step) closures of configs[2] for engine module `em` (tools/ab_bench.py uses this too;
    ``shared``: a dict it passes to every variant so that all of them work in the same large buffers)."""
    state = _lowdim_state(em, eng, utts, shared)
    return (lambda: state["aplan"].run(feats=state["feats"], out=state["out"]),
            lambda: state["splan"].run(out=state["pcm"]))


def _lowdim_state(em, eng, utts, shared=None):
    import torch
    from scipy import signal

    aplan = em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
    H = aplan.fft_len // 2 + 1
    if shared is not None and "feats" in shared:
        feats = shared["feats"]
    else:
        feats = tuple(eng.empty_feats(aplan.lossless.total_frames, H) for _ in range(3))
        if shared is not None:
            shared["feats"] = feats
    out = aplan.run(feats=feats)
    torch.cuda.synchronize()
    res = [t.cpu().numpy().astype(np.float64) for t in out]
    sutts = []
    for u in range(len(utts)):
        a, b = int(aplan.out_off[u]), int(aplan.out_off[u + 1])
        v_f0 = aplan.f0_out[u]
        with np.errstate(divide="ignore"):
            v_lf0 = np.log((v_f0 > 0).astype(float) * signal.medfilt(v_f0))
        v_lf0[np.isinf(v_lf0) | np.isnan(v_lf0)] = -1.0e10                 # la.f0_to_lf0 (libaudio.py:458-465)
        sutts.append((res[0][a:b], res[1][a:b], res[2][a:b], v_lf0))
    np.random.seed(0)
    try:
        splan = em.CompressedSynthesisPlan(eng, sutts, FS, b_const_rate=True, post_filter=True)
    except TypeError:
        splan = em.CompressedSynthesisPlan(eng, sutts, FS, b_const_rate=True)
    if shared is not None:   # the unwarped spectra (3 x 0.47 GB) and the output as well
        if hasattr(splan, "_buffers"):
            if "splan_buf" in shared:
                splan._buf = shared["splan_buf"]
            else:
                shared["splan_buf"] = splan._buffers()
        shared.setdefault("pcm", eng.empty((splan.total_out,)))
        pcm = shared["pcm"]
    else:
        pcm = eng.empty((splan.total_out,))
    return dict(aplan=aplan, splan=splan, feats=feats, out=out, pcm=pcm)


class _Marks:
    def __init__(self, torch):
        self.torch, self.ev = torch, []

    def __call__(self, name):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append((name, e))

    def durations(self):
        return [(n1, e0.elapsed_time(e1)) for (_n0, e0), (n1, e1) in zip(self.ev[:-1], self.ev[1:])]


def measure_lowdim(eng, utts, steps, warmup):
    import torch

    from magphase_amd import engine as em

    st = _lowdim_state(em, eng, utts)
    aplan, splan = st["aplan"], st["splan"]

    def step(mark=None):
        aplan.run(feats=st["feats"], out=st["out"], mark=mark)
        splan.run(out=st["pcm"], mark=mark)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    acc, reps = {}, max(5, min(steps, 20))
    for _ in range(reps):
        m = _Marks(torch)
        step(m)
        torch.cuda.synchronize()
        for name, ms in m.durations():
            if name != "start":
                acc[name] = acc.get(name, 0.0) + ms / reps
    N = aplan.fft_len
    H = N // 2 + 1
    Fv, Fc, Fs = aplan.lossless.total_frames, aplan.total_out_frames, splan.total_frames
    n_in, n_out, n_noise = aplan.lossless.total_smpls, splan.total_out, int(sum(splan.ns_len))
    dims = aplan.mag_dim + 2 * aplan.phase_dim
    n_per, n_voiced = int(splan.n_per), int(splan.voiced_host.sum())
    n_phase_rows = int(aplan.rows_in_use.sum().item()) if getattr(aplan, "phase_on_rows", False) else Fv
    # per-kernel bound.  bytes: what THIS kernel reads + writes as the path is staged today; flops: the GEMM's 2 m n k
    kinfo = {
        "k_analysis": ("hbm", 12.0 * H * Fv + 4.0 * n_in),
        # magnitude row of every frame, phase rows of the frames a voiced constant-rate frame interpolates from
        "k_analysis_f64": ("hbm", 4.0 * H * Fv + 8.0 * H * n_phase_rows + 4.0 * n_in),
        "k_mel_warp_mfma": ("mfma", 2.0 * H * dims * Fc),
        "k_post_filter": ("hbm", 8.0 * aplan.mag_dim * Fc),
        # variable-rate rows.  Magnitudes: two products over all H bins (k_mel_unwarp_tiled, timed under this mark too);
        # phases: real + imaginary, voiced frames only, bins below the periodic / aperiodic crossfade only
        "k_mel_unwarp_mfma": ("mfma", 2.0 * H * 2 * aplan.mag_dim * Fs
                              + 2.0 * n_per * 2 * aplan.phase_dim * n_voiced),
        "k_noise_stats": ("hbm", 4.0 * n_noise + 4.0 * Fs),
        "k_noise_gains": ("hbm", 12.0 * Fs),
        # one unwarped magnitude row per frame, the two phase rows of voiced frames below the crossfade
        "k_synth_comp_pair": ("hbm", 4.0 * H * Fs + 8.0 * n_per * n_voiced + 4.0 * n_noise + 4.0 * n_out),
        "k_ola_fixup": ("hbm", 12.0 * splan.n_runs * N),
    }
    kern = []
    for name, ms in acc.items():
        bound, work = kinfo.get(name, ("hbm", 0.0))
        if bound == "hbm":
            ach = work / (ms * 1e-3) / 1e9
            kern.append({"name": name, "ms": round(ms, 4), "bound": "hbm", "staged_bytes": work,
                         "achieved": round(ach, 1), "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)})
        else:
            ach = work / (ms * 1e-3) / 1e12
            kern.append({"name": name, "ms": round(ms, 4), "bound": "mfma", "flops": work,
                         "achieved": round(ach, 1), "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TF, 4)})
    dom = max(kern, key=lambda k: k["ms"])
    ms_step = dt / steps * 1e3
    # algorithmic bytes of C3 = C4 + C5 per 5 ms frame, + the staged lossless features the constant-rate
    # interpolation works on (2 x 12 H per variable-rate frame), which 8(d) allows for this configuration
    alg_fused = (4.0 * n_in + 4.0 * (dims + 2) * Fc) + (4.0 * (dims + 1) * Fc + 4.0 * n_noise + 4.0 * n_out)
    alg_staged = alg_fused + 24.0 * H * Fv
    traffic, src = _committed_traffic("lowdim_step")
    return {
        "workload": "configs[2]: the same 64 x 5 s @48 kHz; analysis_compressed(mag 60, phase 45, constant 5 ms rate) -> "
                    "post-filter -> synthesis_from_compressed(b_const_rate=True, per_phase_type='magphase')",
        "ms_per_step": round(ms_step, 4), "steps": steps,
        "value": round(Fc / (ms_step * 1e-3), 1), "unit": "5ms-frames/s",
        "x_realtime": round(UTTS_PER_GPU * DUR_S / (ms_step * 1e-3), 1),
        "const_rate_frames": Fc, "variable_rate_frames_analysed": Fv, "variable_rate_frames_resynthesised": Fs,
        "voiced_frames_resynthesised": n_voiced, "periodic_bins": n_per, "analysis_rows_with_phase": n_phase_rows,
        "kernels": kern,
        "roofline": {"kernel": dom["name"], "bound": dom["bound"], "achieved": dom["achieved"], "unit": dom["unit"],
                     "peak": HBM_PEAK_GBS if dom["bound"] == "hbm" else MFMA_F32_PEAK_TF, "frac": dom["frac"]},
        "path_bytes": {"algorithmic_fused (SURVEY 8d: C4 + C5)": alg_fused,
                       "algorithmic_with_staged_lossless_features (8d allowance for constant rate)": alg_staged,
                       "hbm_traffic_measured": traffic, "hbm_traffic_source": src,
                       "traffic_over_algorithmic_fused": (round(traffic / alg_fused, 2) if traffic else None),
                       "traffic_over_algorithmic_staged": (round(traffic / alg_staged, 2) if traffic else None)},
    }


# ------------------------------------------------------------------------------------------------------------------
# what a caller gets
# ------------------------------------------------------------------------------------------------------------------
def measure_e2e(utts):
    """Array API (numpy in -> numpy out, PCIe both ways) and file interface (tools/corpus_throughput.py), x real time."""
    import warnings

    from magphase_amd import magphase as mp

    out = {}
    sub = utts[:16]
    audio = len(sub) * DUR_S
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        feats = mp.analysis_lossless_batch(sub)                        # warm-up (pinned buffers, tables)
        fin = [(f[0], f[1], f[2], f[3], f[4]) for f in feats]
        mp.synthesis_from_lossless_batch(fin[:2])
        runs = []
        for _ in range(3):                                             # median of three (fresh 0.7 GB of pages per run)
            t0 = time.perf_counter()
            feats = mp.analysis_lossless_batch(sub)
            t_a = time.perf_counter() - t0
            fin = [(f[0], f[1], f[2], f[3], f[4]) for f in feats]
            t0 = time.perf_counter()
            mp.synthesis_from_lossless_batch(fin)
            runs.append((t_a + time.perf_counter() - t0, t_a))
        runs.sort()
        t_a = runs[1][1]
        t_s = runs[1][0] - t_a
    nfr = int(sum(f[0].shape[0] for f in feats))
    out["array_api_lossless"] = {
        "what": "mp.analysis_lossless_batch + mp.synthesis_from_lossless_batch on %d utterances: int16 PCM + epochs in, "
                "float64 numpy features out (%.2f GB of float32 across PCIe, widened on the host by native threads), the same "
                "features back in (narrowed into pinned staging), float64 PCM out" % (len(sub), 3 * 4.0 * nfr * 2049 / 1e9),
        "analysis_s": round(t_a, 3), "synthesis_s": round(t_s, 3), "timing": "median of 3 runs",
        "samples_s": [round(r[0], 3) for r in runs],
        "frames_per_s": round(nfr / (t_a + t_s), 1), "x_realtime": round(audio / (t_a + t_s), 1)}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import corpus_throughput

        out["file_interface"] = corpus_throughput.run(n_utt=int(os.environ.get("BENCH_E2E_UTTS", 128)))
    except Exception as e:
        out["file_interface"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


# ------------------------------------------------------------------------------------------------------------------
def _kernel_source_hash():
    """sha1 of the DEVICE sources (csrc/*.hip, *.hpp): what the committed PMC traffic was measured on.  The host-only
    .cpp files (file helpers, planners) do not change a kernel's traffic."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "magphase_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if not f.endswith((".hip", ".hpp")):
            continue
        with open(os.path.join(d, f), "r") as fh:
            for line in fh:      # the code, not the commentary: // comments and blank lines do not count
                code = line.split("//", 1)[0].strip()
                if code:
                    h.update(code.encode() + b"\n")
    return h.hexdigest()[:12]


def _committed_traffic(kernel):
    """
    HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/pmc_summary.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be
    read from inside this process, so the number is a committed measurement -- it is returned only if the kernel
    sources it was measured on are the ones in this tree (sha1 recorded with it); otherwise null ("stale").
    """
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        if t.get("csrc_sha1") != _kernel_source_hash():
            return None, "profiles/traffic.json is stale (measured on csrc %s, this tree is %s)" % (
                t.get("csrc_sha1"), _kernel_source_hash())
        v = t.get(kernel, {}).get("hbm_bytes_per_launch")
        return v, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (%s), csrc %s" % (
            t.get("source", "profiles/"), t.get("csrc_sha1"))
    except Exception:
        return None, "no committed PMC measurement"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="headline only: no configs2 / e2e / CPU baselines")
    ap.add_argument("--no-e2e", action="store_true", help="skip the array-API / file-interface block (profiling runs)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_DEVICE=1: test hook to exercise the N > 1 code path on a 1-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("BENCH_SHARE_DEVICE") else local_rank
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":   # "nccl" is RCCL on ROCm; only the barrier and two scalar reductions use it
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan, get_engine

    eng = get_engine()
    utts = make_batch(rank)
    t_plan0 = time.perf_counter()
    aplan = LosslessAnalysisPlan(eng, utts)
    splan = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
    torch.cuda.synchronize()
    t_plan_cold = time.perf_counter() - t_plan0
    t_plan0 = time.perf_counter()       # again: steady state (pinned staging and tables exist)
    aplan = LosslessAnalysisPlan(eng, utts)
    splan = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan0
    N = aplan.fft_len
    H = N // 2 + 1
    F = aplan.total_frames
    feats = tuple(eng.empty_feats(F, H) for _ in range(3))
    strips = eng.empty((max(splan.strip_floats, 1),))
    pcm_out = eng.empty((splan.total_out,))

    def step():
        aplan.run(out=feats)
        splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm_out)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        fr = torch.tensor([float(F)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = float(fr.item())
    else:
        total_frames = float(F)

    # ---- per-kernel durations with HIP events on the launch stream (separate, untimed-for-value loop)
    names = ("k_analysis", "k_synth_ola_pair", "k_ola_fixup")
    acc = [0.0, 0.0, 0.0]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    reps = max(5, min(args.steps, 20))
    for _ in range(reps):
        ev[0].record()
        aplan.run(out=feats)
        ev[1].record()
        eng.synthesis_lossless_ola(N, feats[0], feats[1], feats[2], splan, strips, pcm_out)
        ev[2].record()
        eng.ola_fixup(N, splan, strips, pcm_out)
        ev[3].record()
        torch.cuda.synchronize()
        for k in range(3):
            acc[k] += ev[k].elapsed_time(ev[k + 1])
    ms = [a / reps for a in acc]
    # algorithmic bytes per launch: features are materialised once (the API returns them),
    # every PCM sample is read once and written once; the run-boundary head strips are NOT algorithmic traffic.
    alg = [12.0 * H * F + 4.0 * aplan.total_smpls, 12.0 * H * F + 4.0 * splan.total_out, 0.0]
    kern = [{"name": names[k], "ms": round(ms[k], 4), "alg_bytes": alg[k],
             "alg_GBps": round(alg[k] / (ms[k] * 1e-3) / 1e9, 1)} for k in range(3)]
    fix_elems = int(np.sum(np.maximum(splan.runs_host["fix_hi"] - splan.runs_host["fix_lo"], 0)))
    kern[2]["note"] = "run-boundary fix-up: %d floats read twice and written once; not algorithmic traffic" % fix_elems
    dom = int(np.argmax(ms[:2]))
    traffic, traffic_src = _committed_traffic(names[dom])
    roof = {"bound": "hbm", "kernel": names[dom], "achieved": kern[dom]["alg_GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": traffic_src, "kernels": kern,
            "path_alg_GBps": round(sum(alg) / (sum(ms) * 1e-3) / 1e9, 1)}

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "frames/sec analysis+synthesis @48kHz FFT=4096",
            "value": round(total_frames * args.steps / dt, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic 48 kHz %.0f s utterances per GPU, lossless analysis+synthesis, "
                                   "FFT=4096, variable frame rate" % (UTTS_PER_GPU, DUR_S),
                       "frames_per_gpu": F, "audio_s_per_gpu": UTTS_PER_GPU * DUR_S,
                       "x_realtime": round(UTTS_PER_GPU * DUR_S * world / (dt / args.steps), 1),
                       "parallelism": "utterance-sharded x%d, no collective" % world,
                       "ola_runs": splan.n_runs,
                       "host_plan_build_s": round(t_plan, 4), "host_plan_build_cold_s": round(t_plan_cold, 3)},
            "roofline": roof,
        }
        full = world == 1 and not args.quick
        if full and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(utts, _cpu_lossless, "lossless analysis+synthesis of 5 s utterances",
                                               "frames/s")
        if full:
            try:
                c2 = measure_lowdim(eng, utts, max(10, args.steps // 4), max(2, args.warmup // 2))
                if not args.no_cpu_baseline:
                    c2["cpu_baseline"] = cpu_baseline(
                        utts, _cpu_lowdim, "configs[2] (analysis_compressed at constant rate -> post_filter -> "
                        "synthesis_from_compressed) of 5 s utterances", "5ms-frames/s", budget_s=15.0)
                out["configs2"] = c2
            except Exception as e:
                out["configs2"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_e2e:
                try:
                    out["e2e"] = measure_e2e(utts)
                except Exception as e:
                    out["e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
