#!/usr/bin/env python
"""
End-to-end corpus throughput through the file interface: wav + .est files on disk ->
iobatch.extract_features_corpus (reader thread, batched kernels, writer thread) -> feature files ->
iobatch.generate_waveforms_corpus -> wav files.  Everything the reference's two batch scripts do, timed wall-clock.

    python tools/corpus_throughput.py            # N_UTT=128 utterances of 5 s
bench.py imports run() for its "e2e" block.
"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "demos"))
import numpy as np  # noqa: E402


def run(n_utt=128, dur=5.0, batch_utts=32, noise_mode=None, reps=3):
    """Returns a dict with the two rates (x real time) and the wall times; files live in a temporary directory.
    Every timed section runs `reps` times; the MEDIAN is reported (the box's page-cache write-back makes single runs
    differ by 2-3 x), all samples are listed."""
    import make_demo_data
    from magphase_amd import iobatch, libaudio as la, synthetic as syn

    tmp = tempfile.mkdtemp(prefix="mpx_corpus_")
    try:
        wav_dir = os.path.join(tmp, "wavs")
        os.makedirs(wav_dir)
        toks = []
        for u in range(n_utt):
            pcm, pm, voi = syn.make_utterance(3000 + u, dur_s=dur)
            tok = "u%04d" % u
            la.write_audio_file(os.path.join(wav_dir, tok + ".wav"), pcm / 32768.0, 48000, norm=None)
            make_demo_data.write_est(os.path.join(wav_dir, tok + ".est"), pm, voi)
            toks.append(tok)
        wavs = [os.path.join(wav_dir, t + ".wav") for t in toks]
        feats = os.path.join(tmp, "feats")
        # warm-up at the timed batch size: the page-locked staging buffers and the device pools are sized by the batch, a
        # corpus pays for them once
        nw = min(n_utt, batch_utts)
        iobatch.extract_features_corpus(wavs[:nw], os.path.join(tmp, "warm"), batch_utts=batch_utts, phase_dim=45, verbose=False)
        ext = []
        for _ in range(reps):
            rep_e = iobatch.CorpusReport()
            t = time.time()
            iobatch.extract_features_corpus(wavs, feats, batch_utts=batch_utts, phase_dim=45, verbose=False, report=rep_e)
            ext.append((time.time() - t, rep_e))
        ext.sort(key=lambda x: x[0])
        t_ext, rep_e = ext[len(ext) // 2]
        audio = n_utt * dur
        gen = {}
        for mode in ([noise_mode] if noise_mode else ["reference", "device"]):
            np.random.seed(1)
            iobatch.generate_waveforms_corpus(feats, toks[:nw], os.path.join(tmp, "warm_syn"), 60, 45, 48000,
                                              pf_type="magphase", batch_utts=batch_utts, verbose=False, noise_mode=mode)
            runs = []
            for _ in range(reps):
                rep_g = iobatch.CorpusReport()
                np.random.seed(1)
                t = time.time()
                iobatch.generate_waveforms_corpus(feats, toks, os.path.join(tmp, "syn_" + mode), 60, 45, 48000,
                                                  pf_type="magphase", batch_utts=batch_utts, verbose=False, report=rep_g,
                                                  noise_mode=mode)
                runs.append((time.time() - t, rep_g))
            runs.sort(key=lambda x: x[0])
            t_gen, rep_g = runs[len(runs) // 2]
            gen[mode] = {"s": round(t_gen, 3), "x_realtime": round(audio / t_gen, 1),
                         "samples_s": [round(x[0], 3) for x in runs],
                         "stage_busy_s": {k: round(v, 3) for k, v in rep_g.items() if k.endswith("_s")}}
        first = gen.get("reference") or next(iter(gen.values()))
        return {"what": "%d wav + .est files of %.0f s @48 kHz on local disk, one process, one GPU, iobatch reader / "
                        "compute / writer pipeline, %d utterances per launch: extraction = analysis_for_acoustic_modelling "
                        "(60 + 45 + 45 + lf0 + shift files), generation = post-filter + synthesis_from_compressed + "
                        "16-bit wav; noise 'reference' = numpy's global RNG drawn on the host exactly as magphase.py:883 "
                        "(the default), 'device' = counter-based generator on the GPU (opt-in)" % (n_utt, dur, batch_utts),
                "audio_s": audio, "extraction_s": round(t_ext, 3), "extraction_x_realtime": round(audio / t_ext, 1),
                "extraction_samples_s": [round(x[0], 3) for x in ext], "timing": "median of %d runs" % reps,
                "generation_s": first["s"], "generation_x_realtime": first["x_realtime"],
                "generation": gen,
                "stage_busy_s": {"extraction": {k: round(v, 3) for k, v in rep_e.items() if k.endswith("_s")}}}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    r = run(n_utt=int(os.environ.get("N_UTT", 128)), noise_mode=os.environ.get("NOISE_MODE"))
    print("feature extraction (wav+est -> .mag/.real/.imag/.lf0/.shift): %.0f s of audio in %.3f s = %.0f x real time"
          % (r["audio_s"], r["extraction_s"], r["extraction_x_realtime"]))
    for mode, g in r["generation"].items():
        print("waveform generation (features -> post-filter -> wav), %9s noise: %.3f s = %.0f x real time  %s"
              % (mode, g["s"], g["x_realtime"], g["stage_busy_s"]))
    print("extraction stage busy seconds (reader / compute / writer threads overlap):", r["stage_busy_s"])
