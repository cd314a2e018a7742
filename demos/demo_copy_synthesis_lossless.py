#!/usr/bin/env python
"""
Python-3 counterpart of the reference's demos/demo_copy_synthesis_lossless.py (same call sequence, :57-91):
analysis_lossless -> synthesis_from_lossless -> write_audio_file.  Runs on the MI355X path.
Input wav needs epochs: a <stem>.est next to it (demos/make_demo_data.py writes synthetic wav + est pairs).
"""
import os
import sys

this_dir = os.path.dirname(os.path.realpath(__file__))
sys.path.append(os.path.realpath(this_dir + '/../src'))

import libutils as lu  # noqa: E402
import libaudio as la  # noqa: E402
import magphase as mp  # noqa: E402

if __name__ == '__main__':
    wav_file_orig = sys.argv[1] if len(sys.argv) > 1 else os.path.join(this_dir, 'data_48k/wavs_nat/syn_000.wav')
    out_dir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(this_dir, 'data_48k/wavs_syn')
    if not os.path.isfile(wav_file_orig):
        import make_demo_data
        make_demo_data.main()
    lu.mkdir(out_dir)

    print("Analysing.....................................................")
    m_mag, m_real, m_imag, v_f0, fs, v_shift = mp.analysis_lossless(wav_file_orig)

    print("Synthesising.................................................")
    v_syn_sig = mp.synthesis_from_lossless(m_mag, m_real, m_imag, v_f0, fs)

    print("Saving wav file..............................................")
    wav_file_syn = out_dir + '/' + lu.get_filename(wav_file_orig) + '_copy_syn_lossless.wav'
    la.write_audio_file(wav_file_syn, v_syn_sig, fs)
    print('Done!')
