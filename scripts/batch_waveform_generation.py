#!/usr/bin/env python
"""
Python-3 counterpart of the reference's scripts/batch_waveform_generation.py (:28-64): for every token,
mp.synthesis_from_acoustic_modelling(in_feats_dir, tok, out_dir, mag_dim, phase_dim, fs, pf_type, b_const_rate=False).
Default input: the predicted features the reference bundles (demos/data_48k/params_predicted, copied as data).
"""
import os
import sys

curr_dir = os.path.dirname(os.path.realpath(__file__))
sys.path.append(os.path.realpath(curr_dir + '/../src'))
import libutils as lu  # noqa: E402
import magphase as mp  # noqa: E402


def synthesis(in_feats_dir, filename_token, out_syn_dir, mag_dim, phase_dim, fs, pf_type):
    mp.synthesis_from_acoustic_modelling(in_feats_dir, filename_token, out_syn_dir, mag_dim, phase_dim, fs,
                                         pf_type=pf_type, b_const_rate=False)
    return


if __name__ == '__main__':
    fs = 48000
    files_scp = sys.argv[1] if len(sys.argv) > 1 else curr_dir + '/../demos/data_48k/file_id_predict.scp'
    in_feats_dir = sys.argv[2] if len(sys.argv) > 2 else curr_dir + '/../demos/data_48k/params_predicted'
    out_syn_dir = sys.argv[3] if len(sys.argv) > 3 else curr_dir + '/../demos/data_48k/wavs_syn_from_predicted'
    mag_dim = 60
    phase_dim = 45
    pf_type = 'magphase'
    lu.mkdir(out_syn_dir)
    l_file_tokns = lu.read_text_file2(files_scp, dtype='string', comments='#').tolist()
    for file_tokn in l_file_tokns:
        synthesis(in_feats_dir, file_tokn, out_syn_dir, mag_dim, phase_dim, fs, pf_type)
    print('Done!')
