#!/usr/bin/env python
"""
Corpus feature extraction for TTS (Merlin) on the MI355X path.

Counterpart (python 3) of the reference script of the same name: every token of the .scp list goes through
mp.analysis_for_acoustic_modelling(wav, out_dir), which writes <token>.mag/.real/.imag/.lf0 (+ .shift) as raw float32.
The reference fans tokens out over a multiprocessing.Pool; here each process drives one GPU and, under
`python -m torch.distributed.run --nproc-per-node N`, takes its share of the list (magphase_amd.sharding: by file
size, longest first; no data is exchanged between ranks).

    python scripts/batch_feature_extraction_for_tts.py [--scp LIST] [--wav-dir DIR] [--out-dir DIR]
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.realpath(__file__))
sys.path.append(os.path.realpath(os.path.join(HERE, "..", "src")))

import libutils as lu  # noqa: E402
import magphase as mp  # noqa: E402
from magphase_amd import iobatch, sharding  # noqa: E402


def main():
    demo = os.path.realpath(os.path.join(HERE, "..", "demos", "data_48k"))
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scp", default=os.path.join(demo, "file_id.scp"))
    ap.add_argument("--wav-dir", default=os.path.join(demo, "wavs_nat"))
    ap.add_argument("--out-dir", default=os.path.join(demo, "params_nat"))
    ap.add_argument("--batch", type=int, default=16, help="utterances per kernel launch (0: one call per file, like the reference)")
    ap.add_argument("--rank-subdirs", action="store_true",
                    help="every rank writes into OUT_DIR/rank<r>/ instead of all ranks into OUT_DIR: file creation in ONE "
                         "directory serialises on its inode lock (tools/file_interface_nproc.py measures it)")
    ap.add_argument("--direct", action="store_true",
                    help="multi-rank runs: create the files directly in OUT_DIR (the reference's way; by default every rank "
                         "writes into OUT_DIR/.rank<r>/ and moves its files up when it is done: the same final layout, "
                         "2.7 x faster under 8 ranks)")
    args = ap.parse_args()
    lu.mkdir(args.out_dir)
    tokens = [str(t) for t in lu.read_text_file2(args.scp, dtype="string", comments="#").tolist()]
    rank, local_rank, world = sharding.dist_env()
    if world > 1:
        import torch
        torch.cuda.set_device(sharding.local_device_index())
        sharding.bind_rank_to_cores(local_rank, world)   # this rank's share of the cores next to its GPU (libutils.py:61-62)
    sizes = [os.path.getsize(os.path.join(args.wav_dir, t + ".wav")) if os.path.isfile(os.path.join(args.wav_dir, t + ".wav"))
             else 0 for t in tokens]
    mine = sharding.shard_by_cost(sizes, world)[rank]
    final_dir = None
    if args.rank_subdirs and world > 1:
        args.out_dir = os.path.join(args.out_dir, "rank%d" % rank)
        lu.mkdir(args.out_dir)
    elif world > 1 and not args.direct:
        final_dir, args.out_dir = args.out_dir, os.path.join(args.out_dir, ".rank%d" % rank)
        if os.path.isdir(args.out_dir):   # left by a run that was killed: its files are NOT this run's results
            import shutil
            stale = os.listdir(args.out_dir)
            for n in stale:
                q = os.path.join(args.out_dir, n)
                shutil.rmtree(q, ignore_errors=True) if os.path.isdir(q) and not os.path.islink(q) else os.remove(q)
            if stale:
                print("[rank %d] removed %d stale files of an earlier, interrupted run from %s" % (rank, len(stale), args.out_dir))
        lu.mkdir(args.out_dir)

    def move_up():   # this rank's files (and its crash list) move up into the common directory
        if final_dir is None or not os.path.isdir(args.out_dir):
            return
        left = 0
        for n in os.listdir(args.out_dir):
            try:
                os.rename(os.path.join(args.out_dir, n), os.path.join(final_dir, n))
            except OSError as e:
                left += 1
                print("[rank %d] could not move %s up: %s" % (rank, n, e))
        if not left:
            try:
                os.rmdir(args.out_dir)
            except OSError:
                pass

    try:
        if args.batch > 0:   # reader thread / kernels / writer thread overlapped, args.batch utterances per launch
            rep = iobatch.CorpusReport()
            n = iobatch.extract_features_corpus([os.path.join(args.wav_dir, tokens[i] + ".wav") for i in mine], args.out_dir,
                                                batch_utts=args.batch, report=rep)
            print("[rank %d] %d of %d files analysed in %d batches" % (rank, rep.get("done", 0), len(mine), n))
            if rep.get("failed"):
                print("[rank %d] %d files failed, listed in %s" % (rank, len(rep["failed"]), os.path.join(
                    final_dir or args.out_dir, os.path.basename(rep["crash_list"]))))
        else:
            for i in mine:
                print("[rank %d] analysing %s.wav" % (rank, tokens[i]))
                mp.analysis_for_acoustic_modelling(os.path.join(args.wav_dir, tokens[i] + ".wav"), args.out_dir)
    except BaseException:
        # Interrupted mid-corpus (an exception, Ctrl-C): the writer thread may have been stopped inside a file, so NOTHING is
        # moved into the common directory -- what this rank finished stays in its own directory, named here, and the
        # exception goes out as it is (partial files must not appear in OUT_DIR, cleanup must not mask the error)
        if final_dir is not None:
            print("[rank %d] interrupted: its files stay in %s (not moved into %s)" % (rank, args.out_dir, final_dir))
        raise
    else:
        move_up()
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
