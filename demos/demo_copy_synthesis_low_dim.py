#!/usr/bin/env python
"""
Low-dimensional copy synthesis on the MI355X path: analysis_compressed -> post_filter -> synthesis_from_compressed.

Counterpart (python 3) of the reference demo of the same name (same calls, same defaults: 60 magnitude and 45 phase
coefficients, variable frame rate, post-filter on, no output high-pass) and the same output file name pattern.

    python demos/demo_copy_synthesis_low_dim.py [--wav FILE] [--out-dir DIR] [--const-rate] [--no-postfilter]
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.realpath(__file__))
sys.path.append(os.path.realpath(os.path.join(HERE, "..", "src")))

import libaudio as la  # noqa: E402
import libutils as lu  # noqa: E402
import magphase as mp  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--wav", default=os.path.join(HERE, "data_48k", "wavs_nat", "syn_000.wav"))
    ap.add_argument("--out-dir", default=os.path.join(HERE, "data_48k", "wavs_syn"))
    ap.add_argument("--mag-dim", type=int, default=60)
    ap.add_argument("--phase-dim", type=int, default=45)
    ap.add_argument("--const-rate", action="store_true", help="5 ms constant frame rate instead of pitch-synchronous")
    ap.add_argument("--no-postfilter", action="store_true")
    args = ap.parse_args()
    if not os.path.isfile(args.wav):
        sys.path.insert(0, HERE)
        import make_demo_data
        make_demo_data.main(out_dir=os.path.dirname(args.wav))
    lu.mkdir(args.out_dir)

    mag_mel_log, real_mel, imag_mel, lf0, _shift, fs, _n = mp.analysis_compressed(
        args.wav, mag_dim=args.mag_dim, phase_dim=args.phase_dim, b_const_rate=args.const_rate)
    if not args.no_postfilter:
        mag_mel_log = mp.post_filter(mag_mel_log, fs)
    v_syn = mp.synthesis_from_compressed(mag_mel_log, real_mel, imag_mel, lf0, fs, b_const_rate=args.const_rate,
                                         b_out_hpf=False)
    name = "%s_copy_syn_low_dim_mag_dim_%d_ph_dim_%d_const_rate_%d.wav" % (
        lu.get_filename(args.wav), args.mag_dim, args.phase_dim, args.const_rate)
    la.write_audio_file(os.path.join(args.out_dir, name), v_syn, fs)
    print("wrote", os.path.join(args.out_dir, name))


if __name__ == "__main__":
    main()
