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
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--mag-dim", type=int, default=60)
    ap.add_argument("--phase-dim", type=int, default=45)
    ap.add_argument("--pf-type", default="magphase", choices=["magphase", "merlin", "no"])
    ap.add_argument("--batch", type=int, default=16, help="utterances per kernel launch (0: one call per file, like the reference)")
    args = ap.parse_args()
    lu.mkdir(args.out_dir)
    tokens = [str(t) for t in lu.read_text_file2(args.scp, dtype="string", comments="#").tolist()]
    rank, local_rank, world = sharding.dist_env()
    if world > 1:
        import torch
        torch.cuda.set_device(local_rank)
    sizes = [os.path.getsize(os.path.join(args.feats_dir, t + ".mag")) for t in tokens]
    mine = sharding.shard_by_cost(sizes, world)[rank]
    if args.batch > 0:   # reader thread / kernels / writer thread overlapped, args.batch utterances per launch
        iobatch.generate_waveforms_corpus(args.feats_dir, [tokens[i] for i in mine], args.out_dir, args.mag_dim,
                                          args.phase_dim, args.fs, pf_type=args.pf_type, b_const_rate=False,
                                          batch_utts=args.batch)
    else:
        for i in mine:
            mp.synthesis_from_acoustic_modelling(args.feats_dir, tokens[i], args.out_dir, args.mag_dim, args.phase_dim,
                                                 args.fs, pf_type=args.pf_type, b_const_rate=False)
    print("rank %d done" % rank)


if __name__ == "__main__":
    main()
