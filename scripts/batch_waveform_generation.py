#!/usr/bin/env python
"""
Waveform generation from predicted low-dimensional features on the MI355X path.

Counterpart (python 3) of the reference script of the same name: for every token,
mp.synthesis_from_acoustic_modelling(feats_dir, token, out_dir, mag_dim, phase_dim, fs, pf_type, b_const_rate=False)
reads <token>.mag/.real/.imag/.lf0, applies the MagPhase post-filter and writes <token>.wav.  Defaults point at the
predicted features the reference bundles (copied as data under demos/data_48k/params_predicted).  One process per
GPU under torch.distributed.run; tokens are dealt to ranks by feature-file size, nothing is exchanged.

    python scripts/batch_waveform_generation.py [--scp LIST] [--feats-dir DIR] [--out-dir DIR] [--pf-type magphase|no]
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
    ap.add_argument("--scp", default=os.path.join(demo, "file_id_predict.scp"))
    ap.add_argument("--feats-dir", default=os.path.join(demo, "params_predicted"))
    ap.add_argument("--out-dir", default=os.path.join(demo, "wavs_syn_from_predicted"))
    ap.add_argument("--fs", type=int, default=48000, help="sample rate of every token (the reference's script)")
    ap.add_argument("--fs-map", default=None, help="text file of `token fs` lines for corpora that mix sample rates "
                                                   "(tokens not listed use --fs); batches are split by rate")
    ap.add_argument("--mag-dim", type=int, default=60)
    ap.add_argument("--phase-dim", type=int, default=45)
    ap.add_argument("--pf-type", default="magphase", choices=["magphase", "merlin", "no"])
    ap.add_argument("--batch", type=int, default=16, help="utterances per kernel launch (0: one call per file, like the reference)")
    ap.add_argument("--noise", default="reference", choices=["reference", "device"],
                    help="reference: aperiodic source from numpy's global RNG as magphase.py:883; device: generated on the GPU, "
                         "seeded by the token (fast, and the same wav whatever the batching / number of GPUs)")
    args = ap.parse_args()
    lu.mkdir(args.out_dir)
    tokens = [str(t) for t in lu.read_text_file2(args.scp, dtype="string", comments="#").tolist()]
    rank, local_rank, world = sharding.dist_env()
    if world > 1:
        import torch
        torch.cuda.set_device(sharding.local_device_index())
        sharding.bind_rank_to_cores(local_rank, world)   # this rank's share of the cores next to its GPU (libutils.py:61-62)
    fs_map = {}
    if args.fs_map:
        for line in open(args.fs_map):
            parts = line.split("#")[0].split()
            if len(parts) >= 2:
                fs_map[parts[0]] = int(parts[1])
    fs_of = {t: fs_map.get(t, args.fs) for t in tokens}
    sizes = [os.path.getsize(os.path.join(args.feats_dir, t + ".mag")) if os.path.isfile(os.path.join(args.feats_dir, t + ".mag"))
             else 0 for t in tokens]
    mine = sharding.shard_by_cost(sizes, world)[rank]
    if args.batch > 0:   # reader thread / kernels / writer thread overlapped, args.batch utterances per launch
        rep = iobatch.CorpusReport()
        iobatch.generate_waveforms_corpus(args.feats_dir, [tokens[i] for i in mine], args.out_dir, args.mag_dim,
                                          args.phase_dim, fs_of, pf_type=args.pf_type, b_const_rate=False,
                                          batch_utts=args.batch, report=rep, noise_mode=args.noise)
        if rep.get("failed"):
            print("[rank %d] %d tokens failed, listed in %s" % (rank, len(rep["failed"]), rep["crash_list"]))
    else:
        for i in mine:
            mp.synthesis_from_acoustic_modelling(args.feats_dir, tokens[i], args.out_dir, args.mag_dim, args.phase_dim,
                                                 fs_of[tokens[i]], pf_type=args.pf_type, b_const_rate=False)
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
