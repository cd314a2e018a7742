#!/usr/bin/env python
"""
Python-3 counterpart of the reference's demos/demo_copy_synthesis_low_dim.py (:60-90): analysis_compressed ->
(optional post_filter) -> synthesis_from_compressed(b_out_hpf=False) -> write_audio_file.  MI355X path.
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
    mag_dim = 60         # Number of Mel-scaled frequency bins.
    phase_dim = 45       # Number of Mel-scaled frequency bins kept for phase features (real and imag).
    b_const_rate = False
    b_postfilter = True
    if not os.path.isfile(wav_file_orig):
        import make_demo_data
        make_demo_data.main()
    lu.mkdir(out_dir)

    print("Analysing.....................................................")
    m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0, v_shift, fs, fft_len = mp.analysis_compressed(
        wav_file_orig, mag_dim=mag_dim, phase_dim=phase_dim, b_const_rate=b_const_rate)

    if b_postfilter:
        print("Postfiltering.................................................")
        m_mag_mel_log = mp.post_filter(m_mag_mel_log, fs)

    print("Synthesising.................................................")
    v_syn_sig = mp.synthesis_from_compressed(m_mag_mel_log, m_real_mel, m_imag_mel, v_lf0, fs,
                                             b_const_rate=b_const_rate, b_out_hpf=False)

    print("Saving wav file..............................................")
    wav_file_syn = out_dir + '/' + lu.get_filename(wav_file_orig) + \
        '_copy_syn_low_dim_mag_dim_%d_ph_dim_%d_const_rate_%d.wav' % (mag_dim, phase_dim, b_const_rate)
    la.write_audio_file(wav_file_syn, v_syn_sig, fs)
    print('Done!')
