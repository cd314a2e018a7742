#!/usr/bin/env python
"""
HTS state-aligned labels -> "variable frame rate" labels for a constant-frame-rate trainer (Merlin).

Counterpart (python 3) of the reference script of the same name: for every token of the .scp list, the frames per
label state are counted from the token's .shift file (written by batch_feature_extraction_for_tts.py) with
mp.get_num_of_frms_per_state and the label times are rewritten with
la.convert_label_state_align_to_var_frame_rate (state i lasts n_i frames of 5 ms).  Host-only text processing: no
kernel is involved.  Tokens that cannot be converted (e.g. a phone without any frame) are appended to a crash list,
like the reference does.

    python scripts/batch_convert_label_state_aligned_to_variable_frame_rate.py [--scp LIST] [--lab-dir DIR]
                                                                               [--shift-dir DIR] [--out-dir DIR]
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
    demo = os.path.realpath(os.path.join(HERE, "..", "demos", "data_48k"))
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--scp", default=os.path.join(demo, "file_id.scp"))
    ap.add_argument("--lab-dir", default=os.path.join(demo, "labs"))
    ap.add_argument("--shift-dir", default=os.path.join(demo, "params_nat"))
    ap.add_argument("--out-dir", default=os.path.join(demo, "labs_var_rate"))
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--prevent-zeros", action="store_true",
                    help="give every state at least one frame (only useful when too many utterances crash)")
    args = ap.parse_args()
    lu.mkdir(args.out_dir)
    tokens = [str(t) for t in lu.read_text_file2(args.scp, dtype="string", comments="#").tolist()]
    crashlist_file = lu.ins_pid("crash_file_list.scp")
    for tok in tokens:
        print("\nAnalysing file: " + tok + "................................")
        in_lab = os.path.join(args.lab_dir, tok + ".lab")
        try:
            v_shift = lu.read_binfile(os.path.join(args.shift_dir, tok + ".shift"), dim=1)
            v_n = mp.get_num_of_frms_per_state(v_shift, in_lab, args.fs, b_prevent_zeros=args.prevent_zeros,
                                               n_states_x_phone=5, nfrms_tolerance=6)
            la.convert_label_state_align_to_var_frame_rate(in_lab, v_n, os.path.join(args.out_dir, tok + ".lab"))
        except (KeyboardInterrupt, SystemExit):
            raise
        except Exception:
            with open(crashlist_file, "a") as f:
                f.write(tok + "\n")
    print("Done!")


if __name__ == "__main__":
    main()
