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
  "corpus_shard": one 1 250-utterance shard of the 8-GPU corpus job (configs[3] extraction + configs[4] mixed-rate
              generation through the batch API, tools/corpus_workload.py; x_realtime = the first pass of the process,
              x_realtime_second_pass = the same pass again).
roofline carries, besides the spec-peak fraction: the HBM traffic of the dominant kernel measured IN THIS RUN (two
rocprofv3 --pmc child passes; --traffic committed|none skips them) and this device's measured streaming read / write /
copy ceilings (mpx_bw_probe).
--quick skips configs2 / e2e / corpus_shard / the CPU baselines / the PMC passes.

    python bench.py --workload corpus --utts 10000      BASELINE configs[3] + configs[4] as the whole job: the corpus is
                                                        LPT-sharded over the ranks (strong scaling), --gpus 1 --utts 1250
                                                        is one shard
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


def cpu_baseline(utts, fn, what, unit, budget_s=12.0, full_pool=True):
    """
    Same parallel model as the reference (libutils.py:32-63: one utterance per multiprocessing.Pool worker).  A pool
    is created and warmed (imports, one BLAS thread per worker) BEFORE its clock starts and every worker gets at least 4
    tasks; tasks cycle through the batch's utterances.  TWO pool sizes are timed: min(cores, 64) workers and
    Pool(os.cpu_count()) -- what lu.run_multithreaded does -- and `value` is the faster of the two
    (`pools` lists both).  Also reported: the per-worker rate inside the pool and one process alone (which may use BLAS
    threads).
    """
    import multiprocessing as mpc

    from oracle import magphase_oracle  # noqa: F401  (imported before the clock starts; checker / CPU baseline only)

    ncores = os.cpu_count() or 1
    fn(utts[0])   # untimed: first-call costs (scipy imports, FFT plan caches)
    t0 = time.perf_counter()
    n1, f1 = 0, 0
    while (time.perf_counter() - t0 < budget_s / 4 and n1 < len(utts)) or n1 == 0:   # one process alone
        f1 += fn(utts[n1])
        n1 += 1
    dt1 = time.perf_counter() - t0
    rate1, t_task = f1 / dt1, dt1 / n1
    what_unit = unit.split("/")[0]
    out = {"value": round(rate1, 1), "unit": unit, "cores": 1, "kind": "port", "host_cores": ncores,
           "sample": "%s, numpy fp64 oracle, one process: %d utterances (%d %s) in %.1f s" % (what, n1, f1, what_unit, dt1),
           "value_1core": round(rate1, 1), "pools": []}
    # (full_pool=False: only the min(cores, 64) pool -- a Pool(256) of oversubscribed numpy workers needs a minute for one
    # configs[2] task each)
    sizes = sorted({max(1, min(ncores, 64)), ncores}) if full_pool else [max(1, min(ncores, 64))]
    for workers in sizes:
        if workers == sizes[0]:
            tasks = int(max(4 * workers, min(8 * workers, workers * (budget_s / 2) / t_task)))
        else:   # Pool(os.cpu_count()): one task per worker (oversubscribed numpy workers run ~4 x slower each)
            tasks = workers
        sample = [utts[i % len(utts)] for i in range(tasks)]
        try:
            with mpc.get_context("fork").Pool(workers, initializer=_cpu_init) as pool:
                pool.map(_cpu_warm, range(4 * workers), chunksize=1)
                t0 = time.perf_counter()
                fp = sum(pool.map(fn, sample, chunksize=1))
                dtp = time.perf_counter() - t0
            out["pools"].append({"workers": workers, "value": round(fp / dtp, 1), "tasks": tasks,
                                 "value_per_worker": round(fp / dtp / workers, 1), "seconds": round(dtp, 2),
                                 what_unit: fp})
        except Exception as e:   # no fork / no semaphores: the single-process rate stands
            out["pools"].append({"workers": workers, "error": type(e).__name__})
    ok = [p_ for p_ in out["pools"] if "value" in p_]
    if ok:
        best = max(ok, key=lambda p_: p_["value"])
        out.update({"value": best["value"], "cores": best["workers"], "value_per_worker": best["value_per_worker"],
                    "sample": "%s, numpy fp64 oracle; pools created and warmed before timing, tasks cycle through the %d "
                              "utterances of the batch: %s; one process alone: %.1f %s" % (
                                  what, len(utts), "; ".join("Pool(%d of %d cores) %d tasks = %d %s in %.1f s -> %.1f %s" % (
                                      p_["workers"], ncores, p_["tasks"], p_[what_unit], what_unit, p_["seconds"],
                                      p_["value"], unit) for p_ in ok), rate1, unit)})
    else:
        out["sample"] += " (pool unavailable)"
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
    if getattr(aplan, "fused_cr", False):   # one kernel, no staged lossless rows (mpx_analysis_compressed_fused_cr)
        feats = None
    elif shared is not None and "feats" in shared:
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


def measure_extraction_kernel(eng, utts, live=None, live_src=None, reps=12):
    """
    The device part of BASELINE configs[3] on this batch: analysis_compressed at the VARIABLE frame rate, mag_dim 60 /
    phase_dim 10 (analysis_for_acoustic_modelling's Q7 setting: alpha_phase 0), as ONE fused kernel
    (mpx_analysis_compressed_fused) and, for comparison, as the staged pair k_analysis_f64 -> k_mel_warp_mfma.
    Algorithmic bytes: 4 S + 4 (mag_dim + 2 phase_dim + 2) per frame -- samples in, coefficients out.
    """
    import torch

    from magphase_amd import engine as em

    out = {"what": "configs[3] extraction kernel(s): 64 x 5 s, variable frame rate, mag 60 / phase 10 (Q7), HIP events, "
                   "median of %d launches" % (reps - 2)}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    old = os.environ.get("MAGPHASE_COMP_FUSED")
    try:
        for name, flag in (("fused", "1"), ("staged", "0")):
            os.environ["MAGPHASE_COMP_FUSED"] = flag
            plan = em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=10, alpha_phase=False)
            res = plan.run()
            feats = None if plan.fused else tuple(eng.empty_feats(plan.lossless.total_frames, plan.fft_len // 2 + 1) for _ in range(3))
            ts = []
            for r in range(reps):
                e0.record()
                plan.run(feats=feats, out=res)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts = sorted(ts[2:])
            out[name + "_ms"] = round(ts[len(ts) // 2], 4)
            if name == "fused":
                F = plan.lossless.total_frames
                alg = 4.0 * plan.lossless.total_smpls + 4.0 * (60 + 2 * 10 + 2) * F
                out.update({"frames": F, "alg_bytes": alg, "alg_bytes_note": "4 S + 4 (60 + 2 x 10 + 2) per frame (SURVEY.md 8d, C4)",
                            "staged_path_bytes": alg + 2 * 12.0 * (plan.fft_len // 2 + 1) * F})
            del plan, res, feats
    finally:
        if old is None:
            os.environ.pop("MAGPHASE_COMP_FUSED", None)
        else:
            os.environ["MAGPHASE_COMP_FUSED"] = old
    out["frames_per_s_fused"] = round(out["frames"] / (out["fused_ms"] * 1e-3), 1)
    if live and "k_analysis_warp_fused" in live:
        out["hbm_traffic_measured_fused"] = round(live["k_analysis_warp_fused"], 1)
        out["traffic_over_algorithmic"] = round(live["k_analysis_warp_fused"] / out["alg_bytes"], 3)
        out["traffic_source"] = live_src
    return out


class _Marks:
    def __init__(self, torch):
        self.torch, self.ev = torch, []

    def __call__(self, name):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append((name, e))

    def durations(self):
        return [(n1, e0.elapsed_time(e1)) for (_n0, e0), (n1, e1) in zip(self.ev[:-1], self.ev[1:])]


def pick_streams(n, enqueue, steps=24, tries=1):
    """
    n HIP streams that really run side by side.  HIP multiplexes its streams onto a few hardware queues (4 by default) and
    two streams that land on the same queue serialise (tools/archive/stream_pair_probe.py: of the pairs among 8 streams about one
    in four does) -- which streams share a queue is not something the API tells.  So: time `steps` steps on one stream,
    then draw streams until alternating between the candidate and EVERY chosen stream is faster than one stream alone
    (two streams on one queue: 3-4 % slower than one stream; on two queues: 0.5-6 % faster, depending on the box); if none
    is after `tries` draws, the best candidate is taken.  Since round 4 `tries` is 1: the FIRST pair of fresh streams is
    used whatever it measures (12 of 12 fresh pairs overlapped on the round-4 boxes, tools/archive/stream_priority_probe.py; a benchmark
    should not re-draw its own configuration) -- the probe only REPORTS whether the pair overlaps (`stream_pick`).
    enqueue(stream, k): enqueue one step on `stream` with buffer set k.  Returns (streams, report).
    """
    import torch

    def block(ss):
        dt = None
        for _rep in range(2):      # the first pass creates the hardware queue of a stream that has not been used yet
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(steps):
                enqueue(ss[i % len(ss)], i % len(ss))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / steps * 1e3
        return dt

    chosen = [torch.cuda.Stream()]
    block(chosen)
    single = block(chosen)
    report = {"single_stream_ms": round(single, 4), "pairs_tried_ms": []}
    while len(chosen) < n:
        best = None
        for _ in range(tries):
            c = torch.cuda.Stream()
            t = max(block([s_, c]) for s_ in chosen)
            report["pairs_tried_ms"].append(round(t, 4))
            if best is None or t < best[0]:
                best = (t, c)
            if t < single:
                break
        chosen.append(best[1])
    return chosen, report


def measure_lowdim(eng, utts, steps, warmup, live=None, live_src=None, n_streams=1, streams=None, side_forms=True):
    import torch

    from magphase_amd import engine as em

    st = _lowdim_state(em, eng, utts)
    aplan, splan = st["aplan"], st["splan"]
    # as in the headline: consecutive (independent) steps alternate between n_streams HIP streams, each with its own plans
    # and buffers; the per-kernel events below are taken one step at a time on the current stream
    states = [st] + [_lowdim_state(em, eng, utts) for _ in range(max(1, int(n_streams)) - 1)]
    if len(states) == 1:
        streams = [torch.cuda.current_stream()]
    elif streams is None or len(streams) < len(states):
        streams = [torch.cuda.Stream() for _ in states]
    counter = [0]

    def step(mark=None):
        if mark is not None or len(states) == 1:
            aplan.run(feats=st["feats"], out=st["out"], mark=mark)
            splan.run(out=st["pcm"], mark=mark)
            return
        k = counter[0] % len(states)
        counter[0] += 1
        sk = states[k]
        with torch.cuda.stream(streams[k]):
            sk["aplan"].run(feats=sk["feats"], out=sk["out"])
            sk["splan"].run(out=sk["pcm"])

    # Timing, independent of the headline's --steps / --warmup (round 2's driver run timed 10 steps after 2 warm-ups,
    # right behind 15 s of GPU idle under the CPU baseline, and got 3.6 x the kernels' own sum): warm up until two
    # consecutive single steps agree to 3 % (clocks back up, every first-use allocation done), then time a fixed number
    # of steps with BOTH the host clock around the synchronised loop and a HIP event pair around the same loop.
    steps, warm_used, last = max(int(steps), 50), 0, None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while warm_used < 60:
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        warm_used += 1
        cur = e0.elapsed_time(e1)
        if warm_used >= max(int(warmup), 3) and last is not None and abs(cur - last) <= 0.03 * min(cur, last):
            break
        last = cur
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt_events = e0.elapsed_time(e1) * 1e-3
    dt_one = None
    if len(states) > 1:     # for the record: the same number of steps one at a time
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            aplan.run(feats=st["feats"], out=st["out"])
            splan.run(out=st["pcm"])
        torch.cuda.synchronize()
        dt_one = time.perf_counter() - t1
    acc, reps = {}, 20
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
        # the one-kernel form (mpx_analysis_compressed_fused_cr): samples in, constant-rate magnitudes and the variable-rate
        # phase coefficients of the rows in use out; then those rows interpolated to the constant rate
        "k_analysis_warp_fused_cr": ("hbm", 4.0 * n_in + 4.0 * aplan.mag_dim * Fc + 8.0 * aplan.phase_dim * n_phase_rows),
        "k_warp_phase_rows": ("hbm", 8.0 * aplan.phase_dim * (n_phase_rows + Fc)),
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
    if live is not None and "lowdim_step" in live:
        traffic, src = live["lowdim_step"], live_src
        for k in kern:
            if k["name"] in live:
                k["hbm_traffic"] = round(live[k["name"]], 1)
    else:
        traffic, src = _committed_traffic("lowdim_step")
    # "noise spectra once" (opt-in, MAGPHASE_NOISE_SPECTRA=store): the synthesis side of the same step in both forms,
    # interleaved in this process (HIP events, one launch chain at a time); skipped when the run itself is in that form
    nso = None
    # (not in profiling runs, --traffic none: rocprofv3's per-kernel averages are to be those of the default form)
    if side_forms and not getattr(splan, "noise_spectra", False) and hasattr(eng.lib, "mpx_noise_stats_spectra"):
        try:
            saved = os.environ.get("MAGPHASE_NOISE_SPECTRA")
            os.environ["MAGPHASE_NOISE_SPECTRA"] = "store"
            try:
                st2 = _lowdim_state(em, eng, utts)
            finally:
                if saved is None:
                    os.environ.pop("MAGPHASE_NOISE_SPECTRA", None)
                else:
                    os.environ["MAGPHASE_NOISE_SPECTRA"] = saved
            forms = (("recompute", st), ("store", st2))
            ts = {n: [] for n, _ in forms}
            for r in range(17):
                for n, s_ in forms:
                    e0.record()
                    s_["splan"].run(out=s_["pcm"])
                    e1.record()
                    torch.cuda.synchronize()
                    if r >= 2:
                        ts[n].append(e0.elapsed_time(e1))
            spec_bytes = 4.0 * int(eng.lib.mpx_noise_spectra_floats(N, Fs))
            nso = {"what": "synthesis side (unwarp -> noise statistics -> gains -> synthesis -> fix-up) with every noise frame "
                           "transformed twice (default) or once, its spectrum stored by the statistics launch and loaded by "
                           "the synthesis launch; median of 15 interleaved rounds, HIP events",
                   "synthesis_ms_recompute": round(float(np.median(ts["recompute"])), 4),
                   "synthesis_ms_store": round(float(np.median(ts["store"])), 4),
                   "extra_hbm_bytes_store": 2.0 * spec_bytes, "default": "recompute"}
            del st2
        except Exception as e:   # the comparison is a side measurement: never the reason a bench line is lost
            nso = {"error": "%s: %s" % (type(e).__name__, e)}
    # the analysis side as ONE kernel (opt-in, MAGPHASE_COMP_FUSED_CR=1): interleaved with the default form in this process
    aok = None
    if side_forms and not getattr(aplan, "fused_cr", False) and hasattr(eng.lib, "mpx_analysis_compressed_fused_cr"):
        try:
            saved = os.environ.get("MAGPHASE_COMP_FUSED_CR")
            os.environ["MAGPHASE_COMP_FUSED_CR"] = "1"
            try:
                ap1 = em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
            finally:
                if saved is None:
                    os.environ.pop("MAGPHASE_COMP_FUSED_CR", None)
                else:
                    os.environ["MAGPHASE_COMP_FUSED_CR"] = saved
            if ap1.fused_cr:
                o1 = ap1.run()
                torch.cuda.synchronize()
                diff = [float((x - y).abs().max().item()) for x, y in zip(o1, st["out"])]
                forms = (("staged", lambda: aplan.run(feats=st["feats"], out=st["out"])), ("one_kernel", lambda: ap1.run(out=o1)))
                ts = {n: [] for n, _ in forms}
                for r in range(17):
                    for n, fn in forms:
                        e0.record()
                        fn()
                        e1.record()
                        torch.cuda.synchronize()
                        if r >= 2:
                            ts[n].append(e0.elapsed_time(e1))
                one_b = (sum(live[k] for k in ("k_cr_index", "k_analysis_warp_fused_cr", "k_warp_phase_rows") if k in live)
                         if live is not None and "k_analysis_warp_fused_cr" in live else None)
                stg_b = (sum(live[k] for k in ("k_analysis_f64", "k_mel_warp_mfma", "k_warp_phase_rows") if k in live)
                         if live is not None and "k_analysis_f64" in live else None)
                aok = {"what": "analysis side of the same step, default (k_analysis_f64 -> k_mel_warp_mfma + k_warp_phase_rows: the "
                               "lossless rows staged in HBM) vs ONE transform-and-warp kernel with the row interpolation inside "
                               "(mpx_analysis_compressed_fused_cr + mpx_warp_phase_rows, MAGPHASE_COMP_FUSED_CR=1); median of 15 "
                               "interleaved rounds, HIP events; traffic from the same PMC child passes as the rest",
                       "analysis_ms_staged": round(float(np.median(ts["staged"])), 4),
                       "analysis_ms_one_kernel": round(float(np.median(ts["one_kernel"])), 4),
                       "hbm_traffic_staged": (round(stg_b, 1) if stg_b else None),
                       "hbm_traffic_one_kernel": (round(one_b, 1) if one_b else None),
                       "step_hbm_traffic_with_one_kernel": (round(traffic - stg_b + one_b, 1) if (traffic and stg_b and one_b) else None),
                       "staged_rows_bytes_not_allocated": 12.0 * H * Fv,
                       "max_abs_difference_of_outputs": {"mag": diff[0], "real": diff[1], "imag": diff[2]},
                       "default": "staged"}
                del o1
            del ap1
        except Exception as e:
            aok = {"error": "%s: %s" % (type(e).__name__, e)}
    return {
        "analysis_one_kernel": aok,
        "noise_spectra_once": nso,
        "workload": "configs[2]: the same 64 x 5 s @48 kHz; analysis_compressed(mag 60, phase 45, constant 5 ms rate) -> "
                    "post-filter -> synthesis_from_compressed(b_const_rate=True, per_phase_type='magphase')",
        "ms_per_step": round(ms_step, 4), "steps": steps, "warmup_steps_used": warm_used,
        "ms_per_step_hip_events": (round(dt_events / steps * 1e3, 4) if len(states) == 1 else None),
        "streams": len(states),
        "ms_per_step_single_stream": (round(dt_one / steps * 1e3, 4) if dt_one else None),
        "kernel_sum_ms": round(sum(k["ms"] for k in kern), 4),
        "timing": "fixed %d steps after warming up until two consecutive steps agree to 3 %% (%d used); ms_per_step = host "
                  "clock around the synchronised loop (consecutive steps alternate between `streams` HIP streams with their "
                  "own plans and buffers), ms_per_step_single_stream = the same steps one at a time, ms_per_step_hip_events = "
                  "one event pair around the loop (single stream only), kernel_sum_ms = sum of the per-kernel event durations "
                  "of a separate 20-step loop, one step at a time" % (steps, warm_used),
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
        feats = mp.analysis_lossless_batch(sub, copy=False)            # warm-up (pinned buffers, tables)
        fin = [(f[0], f[1], f[2], f[3], f[4]) for f in feats]
        mp.synthesis_from_lossless_batch(fin[:2])
        runs = []
        for _ in range(7):                                             # median of seven (fresh 0.7 GB of pages per run: single runs scatter 0.02-0.15 s)
            t0 = time.perf_counter()
            feats = mp.analysis_lossless_batch(sub, copy=False)        # row views of the batch's arrays
            t_a = time.perf_counter() - t0
            fin = [(f[0], f[1], f[2], f[3], f[4]) for f in feats]
            t0 = time.perf_counter()
            mp.synthesis_from_lossless_batch(fin)
            runs.append((t_a + time.perf_counter() - t0, t_a))
        runs.sort()
        t_a = runs[len(runs) // 2][1]
        t_s = runs[len(runs) // 2][0] - t_a
    nfr = int(sum(f[0].shape[0] for f in feats))
    out["array_api_lossless"] = {
        "what": "mp.analysis_lossless_batch + mp.synthesis_from_lossless_batch on %d utterances: int16 PCM + epochs in, "
                "float64 numpy features out (%.2f GB of float32 across PCIe, widened on the host by native threads), the same "
                "features back in (narrowed into pinned staging), float64 PCM out" % (len(sub), 3 * 4.0 * nfr * 2049 / 1e9),
        "analysis_s": round(t_a, 3), "synthesis_s": round(t_s, 3), "timing": "median of %d runs" % len(runs),
        "samples_s": [round(r[0], 3) for r in runs],
        "frames_per_s": round(nfr / (t_a + t_s), 1), "x_realtime": round(audio / (t_a + t_s), 1)}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import corpus_throughput

        out["file_interface"] = corpus_throughput.run(n_utt=int(os.environ.get("BENCH_E2E_UTTS", 128)))
    except Exception as e:
        out["file_interface"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:    # the extraction batch path under EIGHT processes (all on this device: a 1-GPU box), files inside the clock
        import file_interface_nproc

        n8 = int(os.environ.get("BENCH_NPROC_UTTS", 1024))
        r1 = file_interface_nproc.run(procs=1, n_utt=n8, share_device=True, reps=3, layouts=("one_directory",))
        r8 = file_interface_nproc.run(procs=8, n_utt=n8, share_device=True, reps=3,
                                      layouts=("one_directory", "rank_subdirs", "stage_then_rename"))
        out["file_interface_8proc"] = dict(r8, one_process_same_corpus=r1["one_directory"],
                                           note="8 ranks creating 5 files per utterance in ONE directory serialise on its lock "
                                                "(one_directory = scripts/batch_feature_extraction_for_tts.py --direct); the "
                                                "script's multi-rank default writes into OUT/.rank<r>/ and moves the files up "
                                                "at the end (stage_then_rename: the reference's final layout); --rank-subdirs "
                                                "keeps one subdirectory per rank; best of 3 runs (all listed) with os.sync() before each")
    except Exception as e:
        out["file_interface_8proc"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


# ------------------------------------------------------------------------------------------------------------------
# measured ceilings and live HBM traffic
# ------------------------------------------------------------------------------------------------------------------
def measure_ceilings(eng):
    """
    What THIS device sustains for a plain streaming read, fill and copy of 1 GiB (mpx_bw_probe: float4 kernels in
    mpx_bw_probe_shapes() launch shapes -- grid x block, accesses in flight per lane, non-temporal bit; tools/archive/bw_sweep.hip),
    timed with HIP events in this process: the "measured device copy-kernel ceiling" of SURVEY.md 8(d), quoted in the
    roofline object beside the 8 TB/s spec peak.  Per kind: the BEST shape's median of 5 launches after 2 warm-ups (round 3
    quoted one shape, 2048 x 256 with one access in flight, which under-drives fills and copies).
    """
    import statistics

    import torch

    from magphase_amd import _lib

    n = 1 << 28
    a, b = eng.empty((n,)), eng.empty((n,))
    a.zero_()
    b.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_shapes = int(eng.lib.mpx_bw_probe_shapes())
    out = {"bytes": 4 * n, "how": "mpx_bw_probe over 1 GiB in %d launch shapes (tools/archive/bw_sweep.hip), HIP events, best shape's "
                                  "median of 5 launches after 2 warm-ups, this process, this device" % n_shapes}
    with torch.cuda.device(eng.device):
        for kind, name, nbytes in ((0, "read", 4.0 * n), (1, "write", 4.0 * n), (2, "copy", 8.0 * n)):
            best, per_shape = 0.0, []
            for shape in range(n_shapes):
                ts = []
                for r in range(7):
                    e0.record()
                    _lib.check(eng.lib.mpx_bw_probe(eng.stream_ptr(), kind + 16 * shape, a.data_ptr(), b.data_ptr(), n), "mpx_bw_probe")
                    e1.record()
                    torch.cuda.synchronize()
                    if r >= 2:
                        ts.append(e0.elapsed_time(e1))
                gbps = round(nbytes / (statistics.median(ts) * 1e-3) / 1e9, 1)
                per_shape.append(gbps)
                best = max(best, gbps)
            out[name + "_GBps"] = best
            out[name + "_GBps_by_shape"] = per_shape
    del a, b
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------------------------
# board power: the lossless kernels run AT the device's power cap, so the bench samples it
# ------------------------------------------------------------------------------------------------------------------
def _hwmon_dir(torch, dev_index=0):
    """hwmon directory of the card HIP device `dev_index` is (matched by PCI address), or None."""
    import glob

    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bus = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        if bus in os.path.realpath(d) and os.path.isfile(os.path.join(d, "power1_input")):
            return d
    return None


def _read_num(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except Exception:
        return None


class PowerSampler:
    """Samples power1_input (uW) of one hwmon directory every few ms from a thread; mean_w() over the samples taken
    after `skip_s` seconds (the sensor is a moving average: the first few hundred ms still show the previous phase)."""

    def __init__(self, hwmon, period_s=0.004):
        import threading

        self.file = os.path.join(hwmon, "power1_input")
        self.period, self.rows, self._stop = period_s, [], False
        self.t0 = time.perf_counter()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        while not self._stop:
            v = _read_num(self.file)
            if v is not None:
                self.rows.append((time.perf_counter() - self.t0, v * 1e-6))
            time.sleep(self.period)

    def stop(self):
        self._stop = True
        self.thread.join()

    def mean_w(self, skip_s=0.0):
        v = [w for t, w in self.rows if t >= skip_s]
        return (sum(v) / len(v)) if v else None


def measure_power(torch, dev_index, phases, seconds=1.6):
    """
    Board power while each phase (name -> callable enqueueing one launch / step) loops for `seconds`: mean of the sensor
    over the second half of the loop, with the time per call from HIP events.  Returns None without an hwmon directory.
    energy_J = power x time per call; energy_above_idle_J subtracts the device's idle power measured first.
    """
    d = _hwmon_dir(torch, dev_index)
    if d is None:
        return None
    torch.cuda.synchronize()
    time.sleep(1.2)
    idle = _read_num(os.path.join(d, "power1_input"))
    cap = _read_num(os.path.join(d, "power1_cap"))
    out = {"cap_W": (round(cap * 1e-6, 1) if cap else None), "idle_W": (round(idle * 1e-6, 1) if idle else None),
           "source": "amdgpu hwmon power1_input of this device (%s), sampled every 4 ms while the phase loops for %.1f s; mean "
                     "over the second half" % (d, seconds), "phases": {}}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn in phases:
        smp = PowerSampler(d)
        t0 = time.perf_counter()
        ms = []
        while time.perf_counter() - t0 < seconds:
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / 20)
        smp.stop()
        w = smp.mean_w(skip_s=0.5 * seconds)
        m = sum(ms[len(ms) // 2:]) / max(1, len(ms[len(ms) // 2:]))
        ph = {"ms": round(m, 4), "board_W": (round(w, 1) if w else None)}
        if w:
            ph["energy_J"] = round(w * m * 1e-3, 4)
            if idle:
                ph["energy_above_idle_J"] = round((w - idle * 1e-6) * m * 1e-3, 4)
            if cap:
                ph["frac_of_cap"] = round(w / (cap * 1e-6), 3)
        out["phases"][name] = ph
        time.sleep(0.3)
    return out


# kernels of one configs[2] step (each launched once per step; the unwarp is two launches of different kernels)
LOWDIM_ONE_KERNEL = ("k_cr_index", "k_analysis_warp_fused_cr")   # instead of the first two with MAGPHASE_COMP_FUSED_CR=1
LOWDIM_KERNELS = ("k_analysis_f64", "k_mel_warp_mfma", "k_warp_phase_rows", "k_post_filter", "k_mel_unwarp_tiled",
                  "k_mel_unwarp_mfma", "k_noise_stats", "k_noise_gains", "k_synth_comp_pair")


def _short_kernel_name(k):
    return k.split("(")[0].split("::")[-1].split("<")[0].replace("void ", "").strip()


def live_traffic(timeout_s=170):
    """
    HBM bytes per launch measured NOW, on this box: two child runs of this script (``--pmc-child``: 4 lossless steps +
    3 configs[2] steps, nothing else) under ``rocprofv3 --pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE`` -- separate passes
    (the two do not fit the TCC's counter slots together), each with --kernel-trace only, as MI355X_MICROARCH.md's
    rocprofv3 section prescribes.  bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KiB and gfx950's
    FETCH_SIZE tallies a coalesced stream's 128-byte requests at 64 bytes (same guide; it self-calibrates here:
    k_synth_ola_pair must fetch the 1.40 GB k_analysis wrote).  Returns ({kernel: bytes per launch, "lowdim_step": ...},
    description) or (None, why not).
    """
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("BENCH_PMC_CHILD"):
        return None, "inside a profiled child"
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(rocprof):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    env = dict(os.environ, BENCH_PMC_CHILD="1", TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    vals, launches = {}, {}
    t_start = time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"):
            left = timeout_s - (time.perf_counter() - t_start)
            if left < 20:
                if ctr.startswith("SQ_"):
                    break                  # the instruction counts are an extra: the traffic passes are what must be there
                return None, "time budget for the PMC passes used up"
            d = os.path.join(tmp, ctr.split()[0])
            cmd = [rocprof, "--pmc"] + ctr.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", "3", "--warmup", "1", "--streams", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                if ctr.startswith("SQ_"):
                    break
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (ctr, r.returncode)
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") not in ctr.split():
                            continue
                        ctr_ = row["Counter_Name"]
                        k = _short_kernel_name(row["Kernel_Name"])
                        if k.startswith("k_"):
                            vals.setdefault(k, {}).setdefault(ctr_, []).append(float(row["Counter_Value"]))
    except Exception as e:   # timeout, no permission for the counters, ...
        return None, "live PMC passes unavailable: %s" % type(e).__name__
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k, c in vals.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f_, w_ = c["FETCH_SIZE"], c["WRITE_SIZE"]
            out[k] = (2.0 * sum(f_) / len(f_) + sum(w_) / len(w_)) * 1024.0
            launches[k] = len(f_)
    # wave-level instruction counts per launch (SQ_INSTS_*: the third pass), for roofline.valu
    out["_insts"] = {k: {c_: sum(v_) / len(v_) for c_, v_ in c.items() if c_.startswith("SQ_")} for k, c in vals.items()
                     if any(c_.startswith("SQ_") for c_ in c)}
    if "k_noise_stats" in out:
        names = LOWDIM_KERNELS
        if os.environ.get("MAGPHASE_COMP_FUSED_CR") == "1":
            names = LOWDIM_ONE_KERNEL + tuple(k for k in LOWDIM_KERNELS if k not in ("k_analysis_f64", "k_mel_warp_mfma"))
        ks = [k for k in names if k in out]
        if "k_analysis_warp_fused_cr" in launches and "k_analysis_f64" in launches and "k_warp_phase_rows" in launches:
            # (the child ran the analysis side in both forms: each launched k_warp_phase_rows -- same bytes -- once per step)
            launches["k_warp_phase_rows"] = max(1, launches["k_warp_phase_rows"] - min(launches["k_analysis_warp_fused_cr"],
                                                                                      launches["k_analysis_f64"]))
        out["lowdim_step"] = sum(max(1, round(launches[k] / launches["k_noise_stats"])) * out[k] for k in ks)
    return out, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of `bench.py --pmc-child` "
                 "(4 lossless + 3 configs[2] steps) on this box, (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, %.0f s"
                 % (time.perf_counter() - t_start))


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


def _self_launch(n):
    """
    `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment, the same command line), as the reference's "call it, it fans out"
    (libutils.py:32-63).  Rank 0 prints the JSON line on our stdout.  Returns the exit code.
    """
    import socket
    import subprocess

    if not os.environ.get("BENCH_SHARE_DEVICE"):
        import torch

        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but %d device(s) visible (BENCH_SHARE_DEVICE=1 puts every rank on cuda:0)\n" % (n, have))
            return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=(None if r == 0 else subprocess.DEVNULL)))
    rc = 0
    try:
        while procs:
            for p_ in list(procs):
                c = p_.poll()
                if c is None:
                    continue
                procs.remove(p_)
                if c != 0:      # one rank failed: the others would wait in a barrier forever
                    rc = rc or c
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams of the SECONDARY figure (value_overlapped / ms_per_step_overlapped): consecutive "
                         "(independent) steps alternating between that many streams.  `value` / `ms_per_step` are always "
                         "one step at a time on one stream; 1 = no secondary figure")
    ap.add_argument("--form", choices=("one", "two"), default=os.environ.get("BENCH_FORM", "one"),
                    help="one: a step = ONE launch that analyses every frame, writes its feature rows and overlap-adds the "
                         "frame rebuilt from them (mpx_roundtrip_lossless_ola + mpx_ola_fixup); two: mpx_analysis_frames, then "
                         "mpx_synthesis_lossless_ola reading the rows back (+ mpx_ola_fixup).  Same outputs; the other form is "
                         "measured beside the chosen one (roofline.two_launch / roofline.one_launch)")
    ap.add_argument("--no-idle-probe", action="store_true",
                    help="skip the from-idle repeat of the timed loop (profiling runs: keeps every launch in the steady state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="headline only: no configs2 / e2e / CPU baselines")
    ap.add_argument("--no-e2e", action="store_true", help="skip the array-API / file-interface block (profiling runs)")
    ap.add_argument("--no-power", action="store_true",
                    help="skip the board-power loops (rocprofv3 --stats runs: their thousands of back-to-back launches of one "
                         "kernel would dominate its average duration; the steps are what the profile is about)")
    ap.add_argument("--workload", choices=("configs1", "corpus"), default="configs1",
                    help="'corpus' = BASELINE configs[3] + configs[4]: a --utts corpus, LPT-sharded over the ranks "
                         "(tools/corpus_workload.py); the default is the headline configs[1] (+ configs2 / e2e / a corpus shard)")
    ap.add_argument("--utts", type=int, default=10000, help="corpus size of --workload corpus (whole job, all ranks)")
    ap.add_argument("--traffic", choices=("live", "committed", "none"), default="live",
                    help="roofline.traffic: 'live' = rocprofv3 --pmc child passes in this run (falls back to the committed "
                         "measurement), 'committed' = profiles/traffic.json, 'none'")
    ap.add_argument("--pmc-child", action="store_true",
                    help="internal: the short workload the live PMC passes profile (lossless steps + 3 configs[2] steps)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        sys.exit(_self_launch(args.gpus))   # `python bench.py --gpus N` on its own: fan out like lu.run_multithreaded does

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from magphase_amd import sharding as _sharding
    # every rank on its own share of the cores next to its GPU (its native staging threads inherit it): the reference's
    # model is one worker per core with nothing shared (libutils.py:61-62)
    core_binding = _sharding.bind_rank_to_cores(local_rank, world) if world > 1 else None
    # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_DEVICE=1: test hook to exercise the N > 1 code path on a 1-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("BENCH_SHARE_DEVICE") else local_rank
    if world > 1:
        import datetime

        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rank 0 measures its CPU baseline, PMC traffic and the configs[2] / e2e blocks AFTER the timed region while the
        # other ranks wait in the closing barrier: minutes, not the default collective timeout
        PG_TIMEOUT = datetime.timedelta(seconds=2400)
        if args.gpus != world and rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs\n" % (args.gpus, world))
        if not os.environ.get("BENCH_SHARE_DEVICE") and torch.cuda.device_count() <= dev_index:
            raise SystemExit("bench.py: rank %d wants cuda:%d but only %d device(s) are visible" % (rank, dev_index, torch.cuda.device_count()))
        torch.cuda.set_device(dev_index)
        if backend == "nccl":   # "nccl" is RCCL on ROCm; only the barrier and two scalar reductions use it -- nothing on the
            try:                # data path -- so a node whose RCCL cannot initialise still runs, over gloo
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index), timeout=PG_TIMEOUT)
                probe = torch.zeros(1, device="cuda")
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as e:
                sys.stderr.write("bench.py rank %d: RCCL unavailable (%s: %s); falling back to gloo\n" % (rank, type(e).__name__, str(e)[:200]))
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                backend = "gloo"
                dist.init_process_group(backend="gloo", timeout=PG_TIMEOUT)
        else:
            dist.init_process_group(backend=backend, timeout=PG_TIMEOUT)
    else:
        dist = None
        torch.cuda.set_device(0)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    from magphase_amd.engine import LosslessAnalysisPlan, LosslessRoundTripPlan, LosslessSynthesisPlan, get_engine

    eng = get_engine()
    if args.workload == "corpus":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import corpus_workload

        def barrier_c():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        rep = corpus_workload.run(args.utts, rank=rank, world=world, dist=dist, barrier=barrier_c)
        if rank == 0:
            c3, c4 = rep["configs3_extraction"], rep["configs4_generation"]
            t = c3["seconds_max_over_ranks"] + c4["seconds_max_over_ranks"]
            print(json.dumps({
                "metric": "frames/sec corpus feature extraction + waveform generation (BASELINE configs[3] + configs[4])",
                "value": round((c3["frames"] + c4["frames"]) / t, 1), "unit": "frames/s", "n_gpus": world,
                "steps": 1, "warmup": 1, "ms_per_step": round(t * 1e3, 2), "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": "configs[3] + configs[4]: %d-utterance corpus, utterance-sharded x%d (LPT), no "
                                       "collective" % (args.utts, world), "x_realtime": round(
                                           (c3["audio_s"] + c4["audio_s"]) / t, 1)},
                "corpus": rep}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    utts = make_batch(rank)
    one = args.form == "one"

    def build_plans():   # the chosen form's plans (what a caller of that form builds); the other form's come afterwards
        if one:
            r = LosslessRoundTripPlan(eng, utts)
            return r, r.analysis, None
        a = LosslessAnalysisPlan(eng, utts)
        return None, a, LosslessSynthesisPlan(eng, a.v_f0, a.fs, a.fft_len)

    t_plan0 = time.perf_counter()
    rt, aplan, splan = build_plans()
    torch.cuda.synchronize()
    t_plan_cold = time.perf_counter() - t_plan0
    t_plan0 = time.perf_counter()       # again: steady state (pinned staging and tables exist)
    rt, aplan, splan = build_plans()
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan0
    if splan is None:
        splan = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
    if rt is None:
        rt = LosslessRoundTripPlan(eng, utts)
    N = aplan.fft_len
    H = N // 2 + 1
    F = aplan.total_frames
    # Consecutive steps are independent (one batch in, one batch out), so they alternate between --streams HIP streams,
    # each with its own feature matrices / strips / output: the next step's analysis fills the tail of this step's
    # synthesis launch (a SIMD serves its waves by age: the last waves of a launch run alone).  Every step
    # does all of its work; nothing is shared between steps but the read-only inputs.  --streams 1 = one step at a time.
    n_streams = max(1, int(args.streams))
    assert rt.total_out == splan.total_out and rt.total_frames == F
    bufs = [(tuple(eng.empty_feats(F, H) for _ in range(3)),
             eng.empty((max(splan.strip_floats, rt.synthesis.strip_floats, 1),)),
             eng.empty((splan.total_out,))) for _ in range(n_streams)]
    feats, strips, pcm_out = bufs[0]

    def step_two(f_, s_, p_):
        aplan.run(out=f_)
        splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)

    def step_one(f_, s_, p_):
        rt.run(feats=f_, strips=s_, out=p_)

    step_form = step_one if one else step_two

    def enqueue(stream, k):
        with torch.cuda.stream(stream):
            step_form(*bufs[k])

    stream_pick = None
    if n_streams > 1:      # streams that map to different hardware queues (pick_streams)
        streams, stream_pick = pick_streams(n_streams, enqueue)
    else:
        streams = [torch.cuda.current_stream()]

    def step(i=0, one_stream=False):
        k = 0 if one_stream else i % n_streams
        enqueue(streams[k], k)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(one_stream=False):
        for i in range(args.warmup):
            step(i, one_stream)
        barrier()
        t = time.perf_counter()
        for i in range(args.steps):
            step(i, one_stream)
        barrier()
        return time.perf_counter() - t

    # The plans above were built on the host with the GPU idle, and a GPU coming out of idle goes through a power-management
    # transient of about 30 ms of busy time (tools/archive/step_curve_probe.py: the analysis launch runs 0.31 -> 0.41 -> 0.30 ms over
    # the first ~40 steps, after ANY idle period, whatever ran before it).  A corpus job is never in that state, so the
    # device is taken out of it before the W warm-up steps; `ms_per_step_from_idle` below is the same W + K steps started
    # 0.5 s after the last launch, for the record.
    PRECOND_STEPS = 0 if args.pmc_child else 64
    for i in range(PRECOND_STEPS):
        step(i, one_stream=True)
    # THE timed region: W warm-up steps, then exactly K steps between barriers, one step at a time on ONE stream -- the
    # figure a reader can reconcile with rocprofv3's kernel durations (ms_per_step = the dominant kernel + the fix-up)
    dt = timed(one_stream=True)
    dt_ov = dt_idle = dt_two = None
    if not args.pmc_child and not args.no_idle_probe:
        time.sleep(0.5)
        dt_idle = timed(one_stream=True)
        for i in range(PRECOND_STEPS):
            step(i, one_stream=True)
        torch.cuda.synchronize()
    if n_streams > 1 and not args.pmc_child:   # secondary figure: consecutive (independent) steps alternating between streams
        dt_ov = timed()                        # (the next step's analysis fills the tail of this step's launch)
    if not args.pmc_child and not args.quick:  # the same K steps in the OTHER form (what a caller of the reference's two
        step_other = step_two if one else step_one   # functions, analysis_lossless then synthesis_from_lossless, gets)
        with torch.cuda.stream(streams[0]):
            for _ in range(args.warmup):
                step_other(*bufs[0])
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_other(*bufs[0])
            barrier()
            dt_two = time.perf_counter() - t1
    if args.pmc_child:      # profiled by live_traffic(): the other form and a few configs[2] steps as well, then done (no JSON line)
        from magphase_amd import engine as em

        for _ in range(3):
            (step_two if one else step_one)(*bufs[0])
        torch.cuda.synchronize()

        st = _lowdim_state(em, eng, utts)
        for _ in range(3):
            st["aplan"].run(feats=st["feats"], out=st["out"])
            st["splan"].run(out=st["pcm"])
        torch.cuda.synchronize()
        del st
        xp = em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=10, alpha_phase=False)   # configs[3]: the fused kernel
        for _ in range(3):
            xp.run()
        torch.cuda.synchronize()
        if not os.environ.get("MAGPHASE_COMP_FUSED_CR"):   # configs[2]'s analysis side in its one-kernel form
            os.environ["MAGPHASE_COMP_FUSED_CR"] = "1"
            try:
                xc = em.CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
            finally:
                os.environ.pop("MAGPHASE_COMP_FUSED_CR", None)
            if getattr(xc, "fused_cr", False):
                for _ in range(3):
                    xc.run()
                torch.cuda.synchronize()
        return
    dt_own = dt
    if dist is not None:
        t = torch.tensor([dt, dt_ov or 0.0, dt_two or 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        dt_ov = float(t[1].item()) if dt_ov else None
        dt_two = float(t[2].item()) if dt_two else None
        fr = torch.tensor([float(F)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = float(fr.item())
    else:
        total_frames = float(F)

    # ---- per-kernel durations with HIP events on the launch stream (separate, untimed-for-value loop)
    reps = max(5, min(args.steps, 20))

    def event_ms(calls):   # mean duration of each call of the sequence, HIP events on the launch stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(calls) + 1)]
        acc = [0.0] * len(calls)
        for _ in range(reps):
            ev[0].record()
            for k, c in enumerate(calls):
                c()
                ev[k + 1].record()
            torch.cuda.synchronize()
            for k in range(len(calls)):
                acc[k] += ev[k].elapsed_time(ev[k + 1])
        return [a / reps for a in acc]

    # algorithmic bytes per launch: features are materialised once (the API returns them) and
    # read once by the synthesis, every PCM sample is read once and written once: 24 H + 8 S per frame for analysis +
    # synthesis; the run-boundary head strips are NOT algorithmic traffic.  The one-launch form does a frame's analysis AND
    # synthesis, so its launch carries the whole per-frame figure -- of which it MOVES only the writes (12 H + 8 S per
    # frame: the rows are not read back), reported beside it as moved_bytes / moved_GBps.
    def kernel_rows(form_one):
        if form_one:
            nm = ("k_roundtrip_pair", "k_ola_fixup")
            ms_ = event_ms((lambda: eng.roundtrip_lossless_ola(N, aplan, rt.synthesis, feats, strips, pcm_out),
                            lambda: eng.ola_fixup(N, rt.synthesis, strips, pcm_out)))
            alg_ = [24.0 * H * F + 4.0 * aplan.total_smpls + 4.0 * splan.total_out, 0.0]
            sp = rt.synthesis
        else:
            nm = ("k_analysis", "k_synth_ola_pair", "k_ola_fixup")
            ms_ = event_ms((lambda: aplan.run(out=feats),
                            lambda: eng.synthesis_lossless_ola(N, feats[0], feats[1], feats[2], splan, strips, pcm_out),
                            lambda: eng.ola_fixup(N, splan, strips, pcm_out)))
            alg_ = [12.0 * H * F + 4.0 * aplan.total_smpls, 12.0 * H * F + 4.0 * splan.total_out, 0.0]
            sp = splan
        rows = [{"name": nm[k], "ms": round(ms_[k], 4), "alg_bytes": alg_[k],
                 "alg_GBps": round(alg_[k] / (ms_[k] * 1e-3) / 1e9, 1)} for k in range(len(nm))]
        if form_one:
            moved = 12.0 * H * F + 4.0 * aplan.total_smpls + 4.0 * splan.total_out
            rows[0]["moved_bytes"] = moved
            rows[0]["moved_GBps"] = round(moved / (ms_[0] * 1e-3) / 1e9, 1)
            rows[0]["note"] = ("analysis + synthesis of every frame in one launch: alg_bytes is SURVEY 8d's per-frame figure "
                               "(24 H + 8 S: rows written once, read once); the launch writes the rows and does not read them "
                               "back, moved_bytes (12 H + 8 S per frame) is what it has to move")
        fix_elems = int(np.sum(np.maximum(sp.runs_host["fix_hi"] - sp.runs_host["fix_lo"], 0)))
        rows[-1]["note"] = "run-boundary fix-up: %d floats read twice and written once; not algorithmic traffic" % fix_elems
        return rows, ms_, alg_

    kern, ms, alg = kernel_rows(one)
    names = [k["name"] for k in kern]
    kern_other, ms_other, alg_other = kernel_rows(not one)
    dom = int(np.argmax(ms[:-1]))
    full = rank == 0 and not args.quick
    # ---- per rank (every rank of an N > 1 run measures these on ITS device; rank 0 gathers them into `per_rank`): the
    # rank's own time for the K steps, its dominant kernel's duration (HIP events above), the board power of its device
    # while the headline step loops.  The reference's model is one worker per utterance with nothing shared
    # (libutils.py:32-63); the N > 1 line says how even the ranks are.
    moved_dom = kern[dom].get("moved_bytes", alg[dom])     # bytes the dominant launch has to move (== alg for two launches)
    rank_power = None
    if not args.quick and not args.no_power:
        try:
            rank_power = measure_power(torch, dev_index, (("headline_step", lambda: step_form(*bufs[0])),), seconds=1.2)
        except Exception:
            rank_power = None
    rp = (rank_power or {}).get("phases", {}).get("headline_step", {})
    # Host side of ONE launch (Engine.prepare_analysis of this rank's 64-utterance batch: native planner + the samples
    # into page-locked memory, no stream touched), every rank at the same time between barriers -- and, for N > 1, rank 0
    # once more ALONE while the others wait: what the ranks cost each other on the host (memory system, cores).
    def host_prepare_ms(reps=12):
        ts = []
        for _ in range(reps):
            t0_ = time.perf_counter()
            p_ = eng.prepare_analysis(utts)
            ts.append(time.perf_counter() - t0_)
            if p_ is not None:
                p_.release()
        ts.sort()
        return 1e3 * ts[len(ts) // 2]

    host_ms = host_ms_alone = float("nan")
    if not args.quick:
        try:
            host_prepare_ms(3)
            barrier()
            host_ms = host_prepare_ms()
            barrier()
            if world > 1:
                if rank == 0:
                    host_ms_alone = host_prepare_ms()
                barrier()
        except Exception:
            pass
    n_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    mine = [float(rank), float(dev_index), dt_own / args.steps * 1e3, ms[dom], moved_dom / (ms[dom] * 1e-3) / 1e9,
            alg[dom] / (ms[dom] * 1e-3) / 1e9, float(F), rp.get("board_W") or float("nan"),
            rp.get("energy_above_idle_J") or float("nan"), rp.get("frac_of_cap") or float("nan"),
            host_ms, float(n_cores), float(eng.host_threads(32 << 20)),
            float(core_binding["numa_node"]) if (core_binding and core_binding.get("numa_node") is not None) else float("nan")]
    if dist is not None:
        mt = torch.tensor(mine, dtype=torch.float64, device=red_dev)
        allr = [torch.zeros_like(mt) for _ in range(world)]
        dist.all_gather(allr, mt)
        allr = [[float(v) for v in t_.cpu().tolist()] for t_ in allr]
    else:
        allr = [mine]

    def _num(v, nd):
        return None if v != v else round(v, nd)

    per_rank = [{"rank": int(r_[0]), "device": int(r_[1]), "ms_per_step": _num(r_[2], 4), "kernel": names[dom],
                 "kernel_ms": _num(r_[3], 4), "moved_GBps": _num(r_[4], 1), "frac": _num(r_[4] / HBM_PEAK_GBS, 4),
                 "alg_8d_GBps": _num(r_[5], 1), "frac_8d": _num(r_[5] / HBM_PEAK_GBS, 4), "frames": int(r_[6]),
                 "board_W": _num(r_[7], 1), "energy_above_idle_J": _num(r_[8], 4), "frac_of_cap": _num(r_[9], 3),
                 "host_prepare_ms": _num(r_[10], 4), "host_cores": int(r_[11]), "native_threads_cap": int(r_[12]),
                 "numa_node": (None if r_[13] != r_[13] else int(r_[13]))}
                for r_ in allr]
    hp_ = [r_["host_prepare_ms"] for r_ in per_rank if r_["host_prepare_ms"] is not None]
    host_contention = None
    if hp_:
        host_contention = {
            "what": "host side of one 64-utterance launch (Engine.prepare_analysis: native planner + 30 MB of samples into "
                    "page-locked memory), median of 12, every rank at the same time",
            "ms_max_over_ranks": max(hp_), "ms_mean_over_ranks": round(sum(hp_) / len(hp_), 4),
            "ms_rank0_alone": (_num(host_ms_alone, 4) if world > 1 else None),
            "ratio_together_over_alone": (round(max(hp_) / host_ms_alone, 3) if (world > 1 and host_ms_alone == host_ms_alone
                                                                                 and host_ms_alone > 0) else None),
            "cores_bound": bool(core_binding), "cores_per_rank": per_rank[0]["host_cores"],
            "native_threads_cap": per_rank[0]["native_threads_cap"]}
    live, live_src = (live_traffic() if (full and args.traffic == "live") else (None, "not requested"))
    if live is not None and names[dom] in live:
        traffic, traffic_src = live[names[dom]], live_src
        for k in kern:
            if k["name"] in live:
                k["hbm_traffic"] = round(live[k["name"]], 1)
    elif args.traffic == "none":
        traffic, traffic_src = None, "not requested"
    else:
        traffic, traffic_src = _committed_traffic(names[dom])
        if live is None and args.traffic == "live" and full:
            traffic_src += " (live passes: %s)" % live_src
    # The headline fraction is what the launch MOVES over its duration against 8 TB/s.  A fused launch (the one-launch
    # step) does not read its feature rows back, so it moves 12 H + 8 S per frame; SURVEY 8d's per-frame figure
    # (24 H + 8 S: rows written once AND read once) is kept beside it as achieved_8d / frac_8d -- for the two-launch
    # form the two are the same number.
    moved_GBps = round(moved_dom / (ms[dom] * 1e-3) / 1e9, 1)
    fr_ = [r_["frac"] for r_ in per_rank if r_["frac"] is not None]
    roof = {"bound": "hbm", "kernel": names[dom], "achieved": moved_GBps, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(moved_GBps / HBM_PEAK_GBS, 4),
            "frac_definition": "moved_bytes / kernel time / peak (bytes the launch has to move: rows written once, samples in "
                               "and out); frac_8d prices the launch at SURVEY 8d's analysis + synthesis bytes (rows written "
                               "once and read once), which a fused launch does not move",
            "moved_bytes": moved_dom, "alg_bytes_8d": alg[dom],
            "achieved_8d": kern[dom]["alg_GBps"], "frac_8d": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 4),
            "frac_ranks": {"min": min(fr_), "mean": round(sum(fr_) / len(fr_), 4), "max": max(fr_)},
            "traffic": traffic, "traffic_source": traffic_src,
            "traffic_over_moved": (round(traffic / moved_dom, 4) if traffic else None),
            "traffic_over_algorithmic": (round(traffic / alg[dom], 4) if traffic else None),
            "kernels": kern,
            "kernel_time_source": "HIP events on the launch stream, mean of %d launches in this process (the rocprofv3 "
                                  "--kernel-trace --stats summary of this command: profiles/, latest r05_*kernel_stats.csv)" % reps,
            "path_alg_GBps": round(sum(alg) / (sum(ms) * 1e-3) / 1e9, 1)}
    # The VALU side (SQ_INSTS_VALU of the live third PMC pass): wave-level vector instructions per frame, and their issue rate
    # against what this chip sustains -- a pure v_fma_f32 stream saturates at 0.80 G wave-instructions / s per SIMD whatever
    # the occupancy (tools/archive/pk_probe.hip, clock_probe.hip: the board's power limit sets the clock), 1024 SIMDs.
    insts = (live or {}).get("_insts", {}).get(names[dom]) if live is not None else None
    VALU_SAT_G = 0.80
    n_simd = 4 * int(torch.cuda.get_device_properties(dev_index).multi_processor_count)
    if insts and insts.get("SQ_INSTS_VALU"):
        v_ = insts["SQ_INSTS_VALU"]
        rate = v_ / (ms[dom] * 1e-3) / n_simd / 1e9
        roof["valu"] = {"insts_per_launch": round(v_, 0), "insts_per_frame": round(v_ / F, 1),
                        "salu_per_frame": (round(insts.get("SQ_INSTS_SALU", 0.0) / F, 1) if insts.get("SQ_INSTS_SALU") else None),
                        "lds_per_frame": (round(insts.get("SQ_INSTS_LDS", 0.0) / F, 1) if insts.get("SQ_INSTS_LDS") else None),
                        "butterfly_floor_per_frame": 2112,
                        "rate_G_per_s_per_simd": round(rate, 4), "sat_rate_G_per_s_per_simd": VALU_SAT_G, "simds": n_simd,
                        "frac": round(rate / VALU_SAT_G, 4),
                        "source": "SQ_INSTS_VALU of this run's third rocprofv3 --pmc child pass / HIP-event kernel time; "
                                  "saturation rate: tools/archive/pk_probe.hip (v_fma_f32 stream at the board's power limit)"}
        cand = {"hbm": roof["frac"], "valu": roof["valu"]["frac"]}
        pw_ = (rank_power or {}).get("phases", {}).get("headline_step", {}).get("frac_of_cap")
        if pw_:
            cand["power"] = pw_
        roof["bound"] = max(cand, key=cand.get)
        roof["bound_candidates"] = {k_: round(v__, 4) for k_, v__ in cand.items()}
        roof["bound_note"] = ("fractions of the three ceilings this launch runs against: hbm = bytes moved / time / 8 TB/s, valu = "
                              "wave-level VALU instructions / time / saturation rate, power = board watts / cap while the step "
                              "loops; `bound` names the largest (frac / achieved / peak above stay the HBM ones, per the bench contract)")
    other_key = "two_launch" if one else "one_launch"
    roof[other_key] = {"kernels": kern_other, "kernel_sum_ms": round(sum(ms_other), 4),
                       "path_alg_GBps": round(sum(alg_other) / (sum(ms_other) * 1e-3) / 1e9, 1),
                       "note": "the same step in the other form (--form %s), same process, one launch at a time" % ("two" if one else "one")}
    if live is not None:
        for k in kern_other:
            if k["name"] in live:
                k["hbm_traffic"] = round(live[k["name"]], 1)
    if full:
        try:    # what this device sustains for plain streams: the read ceiling bounds k_synth_ola_pair, the write one k_analysis
            ceil = measure_ceilings(eng)   # and k_roundtrip_pair (it moves writes only)
            roof["measured_ceilings"] = ceil
            for k in kern + kern_other:
                if k["name"] == "k_analysis":
                    k["frac_of_measured_write"] = round(k["alg_GBps"] / ceil["write_GBps"], 4)
                elif k["name"] == "k_synth_ola_pair":
                    k["frac_of_measured_read"] = round(k["alg_GBps"] / ceil["read_GBps"], 4)
                elif k["name"] == "k_roundtrip_pair":
                    k["moved_frac_of_measured_write"] = round(k["moved_GBps"] / ceil["write_GBps"], 4)
            if not one:
                roof["frac_of_measured_read"] = kern[1]["frac_of_measured_read"]
        except Exception as e:
            roof["measured_ceilings"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:    # board power while each kernel loops: both lossless kernels sit at the device's power cap
            from magphase_amd import _lib as _l

            n_pr = 1 << 28   # the streaming probes too: what a GB read / written costs on this board (pJ per byte)
            pa, pb = eng.empty((n_pr,)), eng.empty((n_pr,))
            pa.zero_(), pb.zero_()
            ceil_ = roof.get("measured_ceilings", {})

            def probe(kind):
                shapes = ceil_.get(("read", "write", "copy")[kind] + "_GBps_by_shape")
                shape = int(np.argmax(shapes)) if shapes else 0
                return lambda: _l.check(eng.lib.mpx_bw_probe(eng.stream_ptr(), kind + 16 * shape, pa.data_ptr(), pb.data_ptr(), n_pr),
                                        "mpx_bw_probe")

            pw = None if args.no_power else measure_power(torch, dev_index, (
                ("k_analysis", lambda: aplan.run(out=feats)),
                ("k_synth_ola_pair+k_ola_fixup", lambda: splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm_out)),
                ("step", lambda: (aplan.run(out=feats), splan.run(feats[0], feats[1], feats[2], strips=strips, out=pcm_out))),
                ("k_roundtrip_pair+k_ola_fixup", lambda: rt.run(feats=feats, strips=strips, out=pcm_out)),
                ("probe_read_1GiB", probe(0)), ("probe_fill_1GiB", probe(1)), ("probe_copy_1GiB", probe(2))))
            del pa, pb
            if pw is not None:
                for nm, nb in (("probe_read_1GiB", 4.0 * n_pr), ("probe_fill_1GiB", 4.0 * n_pr), ("probe_copy_1GiB", 8.0 * n_pr)):
                    ph = pw["phases"].get(nm, {})
                    if "energy_above_idle_J" in ph:
                        ph["pJ_per_byte_above_idle"] = round(ph["energy_above_idle_J"] / nb * 1e12, 1)
            if pw is not None:
                roof["power"] = pw
                roof["power_note"] = ("a launch at the cap lasts energy / (cap - idle) whatever its memory rate: the 8 TB/s "
                                      "fraction above is bounded by the board's power limit, not by HBM (tools/energy_probe.py)")
        except Exception as e:
            roof["power"] = {"error": "%s: %s" % (type(e).__name__, e)}

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
            "value_two_launch": (round(total_frames * args.steps / dt_two, 1) if (dt_two and one) else None),
            "value_overlapped": (round(total_frames * args.steps / dt_ov, 1) if dt_ov else None),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: %d synthetic 48 kHz %.0f s utterances per GPU, lossless analysis+synthesis, "
                                   "FFT=4096, variable frame rate" % (UTTS_PER_GPU, DUR_S),
                       "form": ("one launch per step: every frame analysed, its feature rows written, the frame rebuilt from "
                                "them and overlap-added (mpx_roundtrip_lossless_ola + mpx_ola_fixup); --form two = "
                                "mpx_analysis_frames, then mpx_synthesis_lossless_ola reading the rows back: roofline.two_launch"
                                if one else "two launches per step (mpx_analysis_frames, mpx_synthesis_lossless_ola + mpx_ola_fixup)"),
                       "frames_per_gpu": F, "audio_s_per_gpu": UTTS_PER_GPU * DUR_S,
                       "x_realtime": round(UTTS_PER_GPU * DUR_S * world / (dt / args.steps), 1),
                       "parallelism": "utterance-sharded x%d, no collective" % world,
                       "streams": 1,
                       "ms_per_step_single_stream": round(ms_step, 4),   # (== ms_per_step since round 6; kept for readers of old lines)
                       "overlapped": ({"streams": n_streams, "stream_pick": stream_pick,
                                       "ms_per_step_overlapped": round(dt_ov / args.steps * 1e3, 4),
                                       "value_overlapped": round(total_frames * args.steps / dt_ov, 1),
                                       "note": "the same K steps alternating between %d HIP streams with their own feature / "
                                               "output buffers (the next step's analysis fills the tail of this step's launch): "
                                               "rounds 3-5 printed THIS as `value`" % n_streams} if dt_ov else None),
                       "other_form": ({"form": "two launches per step (mpx_analysis_frames, then mpx_synthesis_lossless_ola "
                                               "reading the rows back, + mpx_ola_fixup): what analysis_lossless followed by "
                                               "synthesis_from_lossless costs" if one else "one launch per step",
                                       "ms_per_step": round(dt_two / args.steps * 1e3, 4),
                                       "value": round(total_frames * args.steps / dt_two, 1)} if dt_two else None),
                       "ms_per_step_from_idle": (round(dt_idle / args.steps * 1e3, 4) if dt_idle else None),
                       "power_state_note": "%d untimed steps take the device out of its post-idle power transient before the "
                                           "W warm-up steps (the plans are built on the host with the GPU idle; "
                                           "tools/archive/step_curve_probe.py); ms_per_step_from_idle = the same W + K steps started "
                                           "0.5 s after the last launch" % PRECOND_STEPS,
                       "streams_note": "value / ms_per_step: one step at a time on one stream (the figure rocprofv3's kernel "
                                       "durations add up to); config.overlapped: the same steps alternating between HIP streams",
                       "ola_runs": (rt.synthesis.n_runs if one else splan.n_runs),
                       "host_plan_build_s": round(t_plan, 4), "host_plan_build_cold_s": round(t_plan_cold, 3)},
            "roofline": roof,
            "per_rank": per_rank,
            "host_contention": host_contention,
        }
        if full and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(utts, _cpu_lossless, "lossless analysis+synthesis of 5 s utterances",
                                               "frames/s")
        if full:
            try:
                c2 = measure_lowdim(eng, utts, 50, 3, live=live, live_src=live_src, n_streams=n_streams, streams=streams,
                                    side_forms=(args.traffic != "none"))
                if not args.no_cpu_baseline:
                    c2["cpu_baseline"] = cpu_baseline(
                        utts, _cpu_lowdim, "configs[2] (analysis_compressed at constant rate -> post_filter -> "
                        "synthesis_from_compressed) of 5 s utterances", "5ms-frames/s", budget_s=15.0, full_pool=False)
                out["configs2"] = c2
            except Exception as e:
                out["configs2"] = {"error": "%s: %s" % (type(e).__name__, e)}
            try:
                out["configs3_extraction_kernel"] = measure_extraction_kernel(eng, utts, live=live, live_src=live_src)
            except Exception as e:
                out["configs3_extraction_kernel"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_e2e:
                try:
                    out["e2e"] = measure_e2e(utts)
                except Exception as e:
                    out["e2e"] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:    # one shard of the 8-GPU corpus job (10 000 utterances / 8): `bench.py --workload corpus --utts 1250`
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import corpus_workload

                    out["corpus_shard"] = corpus_workload.run(int(os.environ.get("BENCH_CORPUS_UTTS", 1250)))
                except Exception as e:
                    out["corpus_shard"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
