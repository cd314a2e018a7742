#!/usr/bin/env python
"""
bench.py -- MagPhase hot-path benchmark on MI355X (contract: see the task statement / DESIGN.md section 6).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], per GPU): 64 synthetic 48 kHz 5 s utterances, lossless analysis +
synthesis, FFT=4096, variable (pitch-synchronous) frame rate.  A step = one pass of the hot path over the
batch: k_analysis -> k_synth_ola_pair -> k_ola_fixup, with PCM and frame descriptors already resident in HBM.
Utterances shard across ranks with no data-path collective (weak scaling: every rank owns 64 utterances).
Metric: frames/s (whole job) = frames processed by all ranks / max-over-ranks wall time of the K steps.
"""
import argparse
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def make_batch(rank):
    from magphase_amd import synthetic as syn

    utts = []
    for i in range(UTTS_PER_GPU):
        pcm, pm, voi = syn.make_utterance(rank * UTTS_PER_GPU + i, dur_s=DUR_S, fs=FS)
        utts.append((pcm, FS, pm, voi))
    return utts


def _cpu_one(u):
    from oracle import magphase_oracle as orc  # checker / CPU baseline only

    pcm, fs, pm, voi = u
    x = pcm.astype(np.float64) / 32768.0
    o = orc.analysis_lossless_from_epochs(x, fs, pm, voi)
    orc.synthesis_from_lossless(o[0], o[1], o[2], o[3], fs)
    return len(o[5])


def cpu_baseline(utts, budget_s=12.0):
    """The oracle (a parity-pinned numpy fp64 port of the reference) timed on this box's host cores.

    Same parallel model as the reference (libutils.py:32-63: one utterance per Pool worker, all cores)."""
    import multiprocessing as mpc

    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    n1, f1 = 0, 0
    while time.perf_counter() - t0 < budget_s / 3 and n1 < len(utts):  # single core first: sizes the pool sample
        f1 += _cpu_one(utts[n1])
        n1 += 1
    dt1 = time.perf_counter() - t0
    rate1 = f1 / dt1
    sample = utts[: min(len(utts), max(ncores, int(ncores * (budget_s * 2 / 3) / (dt1 / n1))))]
    t0 = time.perf_counter()
    try:
        with mpc.get_context("fork").Pool(ncores) as pool:
            fp = sum(pool.map(_cpu_one, sample))
        rate_pool = fp / (time.perf_counter() - t0)
    except Exception:
        rate_pool, fp, ncores, sample = rate1, f1, 1, utts[:n1]
    return {
        "value": round(rate_pool, 1),
        "unit": "frames/s",
        "cores": ncores,
        "kind": "port",
        "sample": "%d of the %d utterances (%.0f s audio, %d frames), lossless analysis+synthesis, numpy fp64 oracle, "
                  "Pool(%d) one utterance per task; single-core rate %.1f frames/s on %d utterances"
                  % (len(sample), len(utts), len(sample) * DUR_S, fp, ncores, rate1, n1),
        "value_1core": round(rate1, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="lossless", choices=["lossless", "lowdim"],
                    help="lossless = BASELINE configs[1] (the metric; default); lowdim = configs[2]: compressed "
                         "analysis (60/45, constant 5 ms rate) + post-filter + compressed synthesis, 1 GPU only")
    args = ap.parse_args()
    if args.workload == "lowdim":
        return main_lowdim(args)

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
    # every PCM sample is read once and written once; the OLA strips are NOT algorithmic traffic.
    alg = [12.0 * H * F + 4.0 * aplan.total_smpls, 12.0 * H * F, 4.0 * splan.total_out]
    kern = [{"name": names[k], "ms": round(ms[k], 4), "alg_bytes": alg[k],
             "alg_GBps": round(alg[k] / (ms[k] * 1e-3) / 1e9, 1)} for k in range(3)]
    dom = int(np.argmax(ms))
    # HBM traffic per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    # (profiles/traffic.json, written by tools/pmc_summary.py --traffic; FETCH_SIZE doubled as MI355X_MICROARCH.md
    # prescribes for gfx950); null when no such measurement is committed for the kernel.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            traffic = json.load(fh).get(names[dom], {}).get("hbm_bytes_per_launch")
    except Exception:
        traffic = None
    roof = {"bound": "hbm", "kernel": names[dom], "achieved": kern[dom]["alg_GBps"], "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(kern[dom]["alg_GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernels": kern,
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
                       "host_plan_build_s": round(t_plan, 3)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(utts)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_lowdim(args):
    """Secondary workload (BASELINE configs[2]); not the headline metric.  Prints one JSON line of its own."""
    import torch

    torch.cuda.set_device(0)
    from magphase_amd import magphase as mp
    from magphase_amd.engine import CompressedAnalysisPlan, CompressedSynthesisPlan, get_engine

    eng = get_engine()
    utts = make_batch(0)
    aplan = CompressedAnalysisPlan(eng, utts, mag_dim=60, phase_dim=45, b_const_rate=True)
    N, H = aplan.fft_len, aplan.fft_len // 2 + 1
    feats = tuple(eng.empty_feats(aplan.lossless.total_frames, H) for _ in range(3))
    out = aplan.run(feats=feats)
    torch.cuda.synchronize()
    res = [t.cpu().numpy().astype(np.float64) for t in out]
    sutts = []
    from scipy import signal
    from magphase_amd import libaudio as la
    for u in range(len(utts)):
        a, b = int(aplan.out_off[u]), int(aplan.out_off[u + 1])
        v_f0 = aplan.f0_out[u]
        v_lf0 = la.f0_to_lf0((v_f0 > 0).astype(float) * signal.medfilt(v_f0))
        sutts.append((mp.post_filter(res[0][a:b], FS), res[1][a:b], res[2][a:b], v_lf0))
    np.random.seed(0)
    splan = CompressedSynthesisPlan(eng, sutts, FS, b_const_rate=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ta, ts = [], []
    for it in range(args.warmup + args.steps):
        ev[0].record()
        aplan.run(feats=feats, out=out)
        ev[1].record()
        splan.run()
        ev[2].record()
        torch.cuda.synchronize()
        if it >= args.warmup:
            ta.append(ev[0].elapsed_time(ev[1]))
            ts.append(ev[1].elapsed_time(ev[2]))
    fa, fs_ = aplan.total_out_frames, splan.total_frames
    ms_a, ms_s = float(np.median(ta)), float(np.median(ts))
    print(json.dumps({
        "metric": "frames/sec low-dim analysis+synthesis @48kHz FFT=4096 (secondary workload, configs[2])",
        "value": round(fa / ((ms_a + ms_s) * 1e-3), 1), "unit": "5ms-frames/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_a + ms_s, 4), "higher_is_better": True, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "configs[2]: 64 x 5 s @48k, mag_dim 60, phase_dim 45, constant 5 ms rate, post-filter on",
                   "const_rate_frames": fa, "variable_rate_frames_resynthesised": fs_,
                   "ms_analysis (k_analysis + k_mel_warp_mfma)": round(ms_a, 4),
                   "ms_synthesis (k_post_filter + k_mel_unwarp_mfma + k_noise_stats + k_noise_gains + "
                   "k_synth_comp_pair + k_ola_fixup, incl. buffer allocations)": round(ms_s, 4),
                   "x_realtime": round(UTTS_PER_GPU * DUR_S / ((ms_a + ms_s) * 1e-3), 1)}}))


if __name__ == "__main__":
    main()
