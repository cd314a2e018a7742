#!/usr/bin/env python
"""
The file interface under N processes: N ranks (one per GPU on a multi-GPU node; all on device 0 on a
1-GPU box: --share-device) run the feature-extraction batch path -- wav + .est files -> iobatch.extract_features_corpus ->
<tok>.mag/.real/.imag/.lf0/.shift -- on their shard of ONE corpus, files INSIDE the clock.  Reports per rank: x real time,
busy seconds of the reader / compute / writer stages; for the job: wall time = slowest rank, x real time.  Two layouts of
the output: every rank into the same directory (what the reference's scripts do, magphase.py:3014-3020: five files per
utterance, all created in one directory = one inode lock) and one subdirectory per rank (--rank-subdirs of
scripts/batch_feature_extraction_for_tts.py).

    python tools/file_interface_nproc.py --procs 8 --utts 1024 --share-device
bench.py imports run() for its e2e.file_interface_8proc block.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "demos"))


STAGES = ("load_s", "compute_s", "store_s", "store_wait_device_s", "store_write_files_s")


def make_corpus(wav_dir, n_utt, dur=5.0, distinct=64):
    """n_utt wav + .est pairs: `distinct` different utterances, the rest hard links to them (reads come from the page cache
    either way; what is measured is the pipeline and the file creation, not the disk)."""
    import make_demo_data
    from magphase_amd import libaudio as la, synthetic as syn

    os.makedirs(wav_dir)
    toks = []
    for u in range(n_utt):
        tok = "u%05d" % u
        w, e = os.path.join(wav_dir, tok + ".wav"), os.path.join(wav_dir, tok + ".est")
        if u < distinct:
            pcm, pm, voi = syn.make_utterance(3000 + u, dur_s=dur)
            la.write_audio_file(w, pcm / 32768.0, 48000, norm=None)
            make_demo_data.write_est(e, pm, voi)
        else:
            src = "u%05d" % (u % distinct)
            os.link(os.path.join(wav_dir, src + ".wav"), w)
            os.link(os.path.join(wav_dir, src + ".est"), e)
        toks.append(tok)
    return toks


def child(args):
    """One rank: warm-up batch, wait for the common start, extract its shard, write its JSON record."""
    import torch

    from magphase_amd import iobatch, sharding

    rank, _lr, world = sharding.dist_env()
    torch.cuda.set_device(sharding.local_device_index())
    toks = [t.strip() for t in open(args.scp) if t.strip()]
    wavs = [os.path.join(args.wav_dir, t + ".wav") for t in toks]
    sizes = [os.path.getsize(w) for w in wavs]
    mine = sharding.shard_by_cost(sizes, world)[rank]
    out_dir = os.path.join(args.out_dir, "rank%d" % rank) if (args.rank_subdirs or args.stage_rename) else args.out_dir
    os.makedirs(out_dir, exist_ok=True)
    warm = os.path.join(args.sync_dir, "warm%d" % rank)
    iobatch.extract_features_corpus([wavs[i] for i in mine[:args.batch]], warm, batch_utts=args.batch, verbose=False)
    open(os.path.join(args.sync_dir, "ready%d" % rank), "w").close()
    go = os.path.join(args.sync_dir, "go")
    while not os.path.exists(go):
        time.sleep(0.002)
    rep = iobatch.CorpusReport()
    t0 = time.perf_counter()
    iobatch.extract_features_corpus([wavs[i] for i in mine], out_dir, batch_utts=args.batch, verbose=False, report=rep)
    t_ren = 0.0
    if args.stage_rename:   # the files of this rank move up into the common directory (same final layout as the reference's)
        t1 = time.perf_counter()
        for n in os.listdir(out_dir):
            os.rename(os.path.join(out_dir, n), os.path.join(args.out_dir, n))
        os.rmdir(out_dir)
        t_ren = time.perf_counter() - t1
    dt = time.perf_counter() - t0
    rec = {"rank": rank, "rename_s": t_ren, "utts": len(mine), "seconds": dt, "load_s": rep.get("load_s", 0.0), "compute_s": rep.get("compute_s", 0.0),
           "store_s": rep.get("store_s", 0.0), "store_wait_device_s": rep.get("store_wait_device_s", 0.0),
           "store_write_files_s": rep.get("store_write_files_s", 0.0), "failed": len(rep.get("failed", []))}
    with open(os.path.join(args.sync_dir, "result%d.json" % rank), "w") as fh:
        json.dump(rec, fh)


def run(procs=8, n_utt=1024, dur=5.0, batch=32, share_device=True, layouts=("one_directory", "rank_subdirs"), keep=None,
        base_dir=None, reps=3):
    """Builds the corpus once and runs the job `reps` times in every layout (os.sync() before each run: a run leaves
    ~0.3 GB of dirty pages, and on a disk-backed directory the NEXT run's writers are throttled by their write-back --
    single runs differed by 7 x); the best run is reported, all wall times listed.  base_dir: where the files live
    (default: the system temp directory; /dev/shm takes the disk out of the picture)."""
    tmp = keep or tempfile.mkdtemp(prefix="mpx_nproc_", dir=base_dir)
    try:
        wav_dir = os.path.join(tmp, "wavs")
        toks = make_corpus(wav_dir, n_utt, dur)
        scp = os.path.join(tmp, "list.scp")
        with open(scp, "w") as fh:
            fh.write("\n".join(toks) + "\n")
        out = {"what": "%d processes%s, %d utterances of %.0f s @48 kHz (wav + .est -> .mag/.real/.imag/.lf0/.shift, 5 files per "
                       "utterance), %d utterances per launch, files inside the clock, all ranks start together after one "
                       "warm-up batch each" % (procs, " sharing device 0" if share_device else " (one GPU each)", n_utt, dur, batch),
               "audio_s": n_utt * dur}
        out["files_on"] = tmp
        for layout, rep_i in [(l, i) for l in layouts for i in range(reps)]:
            sync = os.path.join(tmp, "sync_%s_%d" % (layout, rep_i))
            os.makedirs(sync)
            out_dir = os.path.join(tmp, "feats_" + layout)
            os.sync()
            ps = []
            for r in range(procs):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(procs), OMP_NUM_THREADS="4")
                if share_device:
                    env["MAGPHASE_SHARE_DEVICE"] = "1"
                cmd = [sys.executable, os.path.abspath(__file__), "--child", "--scp", scp, "--wav-dir", wav_dir, "--out-dir", out_dir,
                       "--sync-dir", sync, "--batch", str(batch)] + (["--rank-subdirs"] if layout == "rank_subdirs" else []) + (
                           ["--stage-rename"] if layout == "stage_then_rename" else [])
                ps.append(subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
            t_wait = time.time()
            while sum(os.path.exists(os.path.join(sync, "ready%d" % r)) for r in range(procs)) < procs:
                if any(p.poll() not in (None, 0) for p in ps) or time.time() - t_wait > 600:
                    errs = [p.stderr.read().decode()[-500:] for p in ps if p.poll() not in (None, 0)]
                    for p in ps:
                        p.kill()
                    raise RuntimeError("a rank failed before the start: %s" % errs)
                time.sleep(0.01)
            t0 = time.perf_counter()
            open(os.path.join(sync, "go"), "w").close()
            rcs = [p.wait() for p in ps]
            wall = time.perf_counter() - t0
            if any(rcs):
                raise RuntimeError("rank exit codes %s: %s" % (rcs, [p.stderr.read().decode()[-300:] for p in ps]))
            recs = [json.load(open(os.path.join(sync, "result%d.json" % r))) for r in range(procs)]
            n_files = sum(len(fs) for _d, _s, fs in os.walk(out_dir))
            slow = max(r["seconds"] for r in recs)
            out.setdefault("_runs_" + layout, []).append({
                "seconds_slowest_rank": round(slow, 4), "x_realtime_job": round(n_utt * dur / slow, 1),
                "seconds_parent_wall": round(wall, 4), "files_written": n_files,
                "per_rank_seconds": [round(r["seconds"], 4) for r in recs],
                "per_rank_x_realtime": [round(r["utts"] * dur / r["seconds"], 1) for r in recs],
                "stage_busy_s_mean": {k: round(sum(r[k] for r in recs) / procs, 4) for k in STAGES},
                "stage_busy_s_max": {k: round(max(r[k] for r in recs), 4) for k in STAGES},
                "rename_s_max": round(max(r.get("rename_s", 0.0) for r in recs), 4),
                "failed": sum(r["failed"] for r in recs)})
            shutil.rmtree(out_dir, ignore_errors=True)
        for layout in layouts:
            runs = sorted(out.pop("_runs_" + layout), key=lambda r: r["seconds_slowest_rank"])
            # the BEST run is reported, all are listed: what spoils a run is the box's page-cache write-back of the files
            # earlier runs (and the bench before them) left behind -- 0.12 s / 0.94 s / 1.02 s for the same one-process job
            out[layout] = dict(runs[0], seconds_slowest_rank_all_runs=[r["seconds_slowest_rank"] for r in runs])
        return out
    finally:
        if keep is None:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--utts", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--share-device", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--rank-subdirs", action="store_true")
    ap.add_argument("--stage-rename", action="store_true")
    ap.add_argument("--scp"), ap.add_argument("--wav-dir"), ap.add_argument("--out-dir"), ap.add_argument("--sync-dir")
    args = ap.parse_args()
    if args.child:
        return child(args)
    rep = {}
    for where, base in (("tmp", None), ("shm", "/dev/shm")):
        rep["1proc_" + where] = run(1, args.utts, batch=args.batch, share_device=True, layouts=("one_directory",), base_dir=base)
        rep["%dproc_%s" % (args.procs, where)] = run(args.procs, args.utts, batch=args.batch, share_device=args.share_device, base_dir=base,
                                                     layouts=("one_directory", "rank_subdirs", "stage_then_rename"))
    print(json.dumps(rep, indent=1))
    for k, v in rep.items():
        for lay in ("one_directory", "rank_subdirs", "stage_then_rename"):
            if lay in v:
                print("%-12s %-18s %8.4f s  %9.1f x real time   runs %s   write %.3f s" % (
                    k, lay, v[lay]["seconds_slowest_rank"], v[lay]["x_realtime_job"], v[lay]["seconds_slowest_rank_all_runs"],
                    v[lay]["stage_busy_s_mean"]["store_write_files_s"]))


if __name__ == "__main__":
    main()
