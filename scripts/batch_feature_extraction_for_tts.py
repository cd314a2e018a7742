#!/usr/bin/env python
"""
Python-3 counterpart of the reference's scripts/batch_feature_extraction_for_tts.py (:33-62): for every token of the
scp list, mp.analysis_for_acoustic_modelling(wav, out_dir) -> <tok>.{mag,real,imag,lf0,shift} float32 files.
On the GPU path the utterances of a process are handled in sequence on its device (b_multiproc is ignored); shard the
scp list over processes / GPUs with magphase_amd.sharding for a node (one process per GPU, no collective).
"""
import os
import sys

curr_dir = os.path.dirname(os.path.realpath(__file__))
sys.path.append(os.path.realpath(curr_dir + '/../src'))
import libutils as lu  # noqa: E402
import magphase as mp  # noqa: E402


def feat_extraction(in_wav_dir, file_name_token, out_feats_dir):
    print("\nAnalysing file: " + file_name_token + '.wav............................')
    wav_file = os.path.join(in_wav_dir, file_name_token + '.wav')
    mp.analysis_for_acoustic_modelling(wav_file, out_feats_dir)
    return


if __name__ == '__main__':
    files_scp = sys.argv[1] if len(sys.argv) > 1 else curr_dir + '/../demos/data_48k/file_id.scp'
    in_wav_dir = sys.argv[2] if len(sys.argv) > 2 else curr_dir + '/../demos/data_48k/wavs_nat'
    out_feats_dir = sys.argv[3] if len(sys.argv) > 3 else curr_dir + '/../demos/data_48k/params_nat'
    lu.mkdir(out_feats_dir)
    l_file_tokns = lu.read_text_file2(files_scp, dtype='string', comments='#').tolist()
    for file_name_token in l_file_tokns:
        feat_extraction(in_wav_dir, file_name_token, out_feats_dir)
    print('Done!')
