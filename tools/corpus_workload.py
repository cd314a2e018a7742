#!/usr/bin/env python
"""
BASELINE.json configs[3] / configs[4] as a workload: a 10 000-utterance corpus, LPT-sharded over the ranks with no
data-path collective (magphase_amd.sharding.shard_by_cost -- the MI355X form of the reference's one-utterance-per-Pool-
worker model, libutils.py:32-63; scripts/batch_feature_extraction_for_tts.py:57, scripts/batch_waveform_generation.py:
28-64), processed in 64-utterance launches through the batch API of the drop-in module:

  configs[3]  feature extraction, analysis_for_acoustic_modelling semantics: 48 kHz, mag 60 / phase 10, variable frame
              rate, Q7 (alpha_phase = False, as magphase.py:3010 forwards it)             mp.analysis_compressed_batch
  configs[4]  waveform generation from "predicted" features (the build's own analysis + N(0, 0.05) perturbations):
              mag 60 / phase 45, MagPhase post-filter, variable rate, output high-pass on, 16-bit PCM out, sample rates
              MIXED 48 kHz (even utterances) / 16 kHz (odd) -- one launch per rate        mp.synthesis_from_compressed_batch

    python bench.py --workload corpus --utts 10000          (under torchrun: one rank per GPU)
    python bench.py --workload corpus --gpus 1 --utts 1250  (one shard of the 8-GPU job)

The corpus: utterance u lasts dur[u] ~ U[2, 8] s (seeded, every rank derives the same list and the same LPT shards from
it); its signal / epochs are one of 64 synthetic base utterances per rank and rate (synthetic.make_utterance, 8 s) cut to
that duration -- a rank only ever materialises its own shard.  What is timed per rank: the wall clock of its whole shard
through the array-level batch API -- host plan build, H2D of the 16-bit PCM / the coefficient matrices, kernels, D2H of
the features / the 16-bit PCM -- i.e. everything but the files (the file interface is bench.py's e2e block).  Reported:
frames/s and x real time of the whole job (max over ranks), the per-rank cost and time imbalance.
"""
import os
import sys
import gc
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

POOL = 64
BASE_DUR_S = 8.0
# utterances per launch.  Extraction: 64 (30 MB of PCM per launch; 128 / 256 measured 14 / 10 % slower: the staging copy grows
# with the launch and the plan build is the bound).  Generation: 256 = 128 per sample rate (per-launch fixed costs -- plan
# build, noise stream, PCIe copies -- amortise: 64 / 128 / 256 / 512 per launch = 70.9 / 84.9 / 96.1 / 44.2 k x real time on one
# box; at 512 the page-locked buffers re-grow).  CORPUS_BATCH overrides both.
BATCH = int(os.environ.get("CORPUS_BATCH", "64"))
# Warm-up of a timed pass: the largest launch repeated for this long (the passes are 30-50 ms of device work, and a device that
# comes out of idle -- the corpus is generated on the host for seconds before -- runs its first tens of ms at idle clocks:
# tools/archive/step_curve_probe.py; a production job of minutes does not see that)
WARM_S = float(os.environ.get("CORPUS_WARM_S", "0.25"))
BATCH_GEN = int(os.environ.get("CORPUS_BATCH", "256"))


def corpus_spec(n_utts, mixed_rates):
    """(durations [s], sample rates) of the whole corpus: the same on every rank."""
    rng = np.random.RandomState(424242)
    dur = np.round(rng.uniform(2.0, BASE_DUR_S, int(n_utts)), 3)
    fs = np.where(np.arange(int(n_utts)) % 2 == 0, 48000, 16000) if mixed_rates else np.full(int(n_utts), 48000)
    return dur, fs.astype(np.int64)


def utterance_cost(dur, fs):
    """LPT weight: frames (~175 per second of audio at any rate) x transform length."""
    return np.asarray(dur, dtype=np.float64) * np.where(np.asarray(fs) > 24000, 4096.0, 2048.0)


def _pool(rank, fs):
    from magphase_amd import synthetic as syn

    return [syn.make_utterance(50000 + 1000 * rank + i, dur_s=BASE_DUR_S, fs=int(fs)) for i in range(POOL)]


def _cut(base, dur, fs):
    pcm, pm, voi = base
    n = int(round(dur * fs))
    keep = pm * fs < n - 2
    return pcm[:n], int(fs), pm[keep], voi[keep]


def _batches(items, n=None):
    n = n or BATCH
    return [items[i:i + n] for i in range(0, len(items), n)]


def run_extraction(rank, mine, dur, fs):
    """configs[3] on this rank's shard `mine` (global utterance indices).  Returns dict(seconds, frames, audio_s, utts)."""
    from magphase_amd import magphase as mp

    pool = _pool(rank, 48000)
    items = [_cut(pool[int(u) % POOL], float(dur[u]), 48000) for u in mine]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw = dict(mag_dim=60, phase_dim=10, alpha_phase=False, as_float32=True)
        # (the cyclic collector is paused from here to the end of the timed pass, as timeit does: one generation-2 sweep over the
        # job's tens of thousands of arrays is 50 ms, as long as the whole pass.  Collected BEFORE the warm-up launches so that
        # the device does not sit idle -- and drop its clocks -- between them and the clock's start)
        gc.collect()
        gc.disable()
        from magphase_amd.engine import get_engine

        eng = get_engine()

        def one_pass(todo):
            # pipelined form (see run_generation): a launch's features are taken one launch later; the host side of launch
            # i + 1 (native planners, samples into page-locked memory) is prepared on the engine's planner thread while this
            # thread enqueues launch i (engine.prepare_async; the plan constructors take the result)
            frames, prev = 0, None
            fut = eng.prepare_async("analysis", todo[0]) if todo else None
            for i, b in enumerate(todo):
                prep = fut.result()
                fut = eng.prepare_async("analysis", todo[i + 1]) if i + 1 < len(todo) else None
                cur = mp.analysis_compressed_batch(b, async_out=True, prepared=prep, **kw)
                if prev is not None:
                    prev[1].wait()
                    frames += sum(int(r[0].shape[0]) for r in prev[0])
                    prev[1].release()
                prev = cur
            if prev is not None:
                prev[1].wait()
                frames += sum(int(r[0].shape[0]) for r in prev[0])
                prev[1].release()
            return frames

        todo = _batches(items)
        if items:   # warm-up: the LARGEST launch of the job first (page-locked slots and ring slots, device pools, tables: a
            # launch that outgrows them re-pins / re-allocates inside the clock -- a production job pays that once in minutes),
            # then a few launches through the SAME pipelined loop as the timed pass (planner thread, copy streams, every slot in
            # rotation), until the device is out of its idle clocks
            big = max(todo, key=lambda b: sum(int(x[0].shape[0]) for x in b))
            mp.analysis_compressed_batch(big, **kw)
            tw, nw = time.perf_counter(), 0
            while nw < 2 or time.perf_counter() - tw < WARM_S:
                one_pass([big] + todo[:3])
                nw += 1
        t0 = time.perf_counter()
        frames = one_pass(todo)
        dt = time.perf_counter() - t0
        gc.enable()
    return {"seconds": dt, "frames": frames, "audio_s": float(np.sum(dur[mine])) if len(mine) else 0.0, "utts": len(mine)}


def run_generation(rank, mine, dur, fs):
    """configs[4] on this rank's shard.  Features come from the build's own analysis of the base utterances (untimed)."""
    from magphase_amd import magphase as mp

    rng = np.random.RandomState(777 + rank)
    feats = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for rate in sorted(set(int(f) for f in fs[mine])):
            pool = _pool(rank, rate)
            res = []
            for b in _batches([(p[0], rate, p[1], p[2]) for p in pool]):
                res += mp.analysis_compressed_batch(b, mag_dim=60, phase_dim=45, as_float32=True)
            feats[rate] = [(r[0] + rng.normal(0, 0.05, r[0].shape).astype(np.float32),
                            np.clip(r[1] + rng.normal(0, 0.05, r[1].shape).astype(np.float32), -1, 1),
                            np.clip(r[2] + rng.normal(0, 0.05, r[2].shape).astype(np.float32), -1, 1), r[3]) for r in res]
        items = []
        for u in mine:
            rate = int(fs[u])
            m, re_, im, lf0 = feats[rate][int(u) % POOL]
            n = max(8, int(round(m.shape[0] * float(dur[u]) / BASE_DUR_S)))
            items.append((rate, (m[:n], re_[:n], im[:n], lf0[:n])))

        # The pipelined form of the batch API, as iobatch uses it: a launch's 16-bit PCM lands in a page-locked ring slot
        # (async_out: engine.HostTicket) and is taken one launch later, so the host plans launch i + 1 while the device
        # works on launch i; numpy's noise stream stays on the device between launches (defer_rng) and goes back to numpy
        # once, at the end.  Every result is waited for and every state put back INSIDE the clock.
        from collections import deque

        from magphase_amd.engine import get_engine

        eng = get_engine()
        pending = deque()

        def take(keep):
            smpls = 0
            while len(pending) > keep:
                sigs, ticket = pending.popleft()
                ticket.wait()
                smpls += sum(int(x.shape[0]) for x in sigs)
                ticket.release()
            return smpls

        def groups_of(batch):   # one launch per sample rate
            return [(rate, [x for r, x in batch if r == rate]) for rate in sorted(set(r for r, _ in batch))]

        def synth(groups, prepared=None):
            frames = 0
            for k, (rate, group) in enumerate(groups):
                pending.append(mp.synthesis_from_compressed_batch(group, rate, b_out_hpf=True, b_post_filter=True,
                                                                  pcm16_norm=0.98, async_out=True, defer_rng=True,
                                                                  prepared=prepared[k].result() if prepared else None))
                frames += sum(int(g[0].shape[0]) for g in group)
            return frames

        def prepare(groups):   # the launches' host side on the planner thread, one batch ahead (see run_extraction)
            return [eng.prepare_async("synthesis", group, rate) for rate, group in groups]

        np.random.seed(1000 + rank)
        gc.collect()
        gc.disable()   # (see run_extraction)
        def one_pass(todo):
            frames, smpls = 0, 0
            fut = prepare(todo[0]) if todo else None
            for i, g in enumerate(todo):
                cur, fut = fut, (prepare(todo[i + 1]) if i + 1 < len(todo) else None)
                frames += synth(g, cur)
                smpls += take(2)            # the two rate groups of the launch just issued stay in flight
            smpls += take(0)
            eng.mt_sync()
            return frames, smpls

        todo = [groups_of(b) for b in _batches(items, BATCH_GEN)]
        if items:   # warm-up (see run_extraction): the largest launch, then launches through the same pipelined loop as the
            # timed pass (planner thread, generator stream, every slot in rotation)
            big = max(_batches(items, BATCH_GEN), key=lambda b: sum(int(x[1][0].shape[0]) for x in b))
            tw, nw = time.perf_counter(), 0
            while nw < 2 or time.perf_counter() - tw < WARM_S:
                one_pass([groups_of(big)] + todo[:2])
                nw += 1
        t0 = time.perf_counter()
        frames, smpls = one_pass(todo)
        dt = time.perf_counter() - t0
        gc.enable()
    return {"seconds": dt, "frames": frames, "audio_s": float(np.sum(dur[mine])) if len(mine) else 0.0, "utts": len(mine)}


def run(n_utts, rank=0, world=1, dist=None, barrier=None):
    """Both configs on a corpus of n_utts utterances sharded over `world` ranks; returns the report dict on every rank
    (per-rank scalars gathered with all_gather_object: the only communication)."""
    from magphase_amd import sharding

    out = {}
    for name, mixed, fn in (("configs3_extraction", False, run_extraction), ("configs4_generation", True, run_generation)):
        dur, fs = corpus_spec(n_utts, mixed)
        cost = utterance_cost(dur, fs)
        shards = sharding.shard_by_cost(cost, world)
        mine = shards[rank]
        if barrier:
            barrier()
        r = fn(rank, mine, dur, fs)
        if barrier:
            barrier()
        # the same pass once more in the same process: what a job longer than one pass runs at (the first pass of a process has
        # two or three 5-10 ms stalls inside asynchronous enqueues on top: docs/LAB_NOTES.md, round 5 item 16)
        r["seconds_second"] = fn(rank, mine, dur, fs)["seconds"]
        if barrier:
            barrier()
        r["cost"] = float(np.sum(cost[mine]))
        if dist is not None and world > 1:
            allr = [None] * world
            dist.all_gather_object(allr, r)
        else:
            allr = [r]
        t_max = max(x["seconds"] for x in allr)
        frames, audio = sum(x["frames"] for x in allr), sum(x["audio_s"] for x in allr)
        costs, secs = [x["cost"] for x in allr], [x["seconds"] for x in allr]
        out[name] = {
            "utterances": int(n_utts), "ranks": world, "launch_utts": BATCH_GEN if fn is run_generation else BATCH,
            "frames": frames, "audio_s": round(audio, 1), "seconds_max_over_ranks": round(t_max, 4),
            "frames_per_s": round(frames / t_max, 1), "x_realtime": round(audio / t_max, 1),
            "x_realtime_second_pass": round(audio / max(x["seconds_second"] for x in allr), 1),
            "per_rank_utts": [x["utts"] for x in allr], "per_rank_seconds": [round(s, 4) for s in secs],
            "lpt_cost_imbalance_max_over_mean": round(max(costs) / (sum(costs) / len(costs)), 5),
            "time_imbalance_max_over_mean": round(max(secs) / (sum(secs) / len(secs)), 4),
        }
    out["what"] = ("synthetic corpus of %d utterances (2-8 s, mean 5 s), LPT-sharded over %d rank(s) by frames x transform "
                   "length, %d utterances per launch (generation: %d), array-level batch API in its pipelined form: a launch's results are taken one launch later from a page-locked ring, "
                   "numpy's noise stream stays on the device between launches (plan build + H2D + kernels + D2H + every wait "
                   "and the final state hand-back inside the "
                   "clock, files outside): configs[3] = analysis_compressed (48 kHz, 60 / 10, Q7), configs[4] = post-filter "
                   "+ synthesis_from_compressed + output high-pass + 16-bit PCM (60 / 45, 48 kHz and 16 kHz mixed, numpy's "
                   "global noise stream)" % (n_utts, world, BATCH, BATCH_GEN))
    return out
