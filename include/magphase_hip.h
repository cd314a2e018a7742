/*
 * magphase_hip.h -- C ABI of libmagphase_hip.so: the MI355X (gfx950) implementation of the MagPhase
 * per-frame analysis / synthesis hot path.
 *
 * The reference (CSTR-Edinburgh/magphase) has no FFI: its boundary is the Python module API of
 * src/magphase.py.  These entry points are what magphase_amd/magphase.py binds through ctypes in
 * place of the numpy loops cited on each function; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers
 *     (PyTorch-ROCm tensors are only the allocator); nothing is allocated or freed here;
 *   - `stream` is a hipStream_t (passed as void*); all work is enqueued on it, nothing synchronises;
 *   - fft_len N is 4096, 2048 or 1024; H = N/2+1; feature matrices are row-major float32 [F x H];
 *   - return 0 on success, <0 on error (text via mpx_last_error()); re-entrant per stream.
 */
#ifndef MAGPHASE_HIP_H
#define MAGPHASE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPX_ABI_VERSION 2

#define MPX_OK 0
#define MPX_ERR_ARG (-1)     /* bad argument (unsupported fft_len, null pointer, negative count) */
#define MPX_ERR_HIP (-2)     /* a HIP runtime call failed */
#define MPX_ERR_HOST (-3)    /* a host-side helper ran out of memory (or threw) in one of its worker threads */

int mpx_version(void);
const char* mpx_last_error(void);

/* Number of bytes of the per-fft_len twiddle table, and its initialisation (a small kernel enqueued on `stream`
 * evaluates it in float64 and rounds to float32: no host copy, no synchronisation).  The table is read-only
 * afterwards and may be shared by every stream of the device. */
size_t mpx_tables_bytes(int fft_len);
int mpx_tables_init(void* stream, int fft_len, void* tables);

/*
 * Analysis of pitch-synchronous frames.  Replaces, for all frames of a batch at once:
 *   magphase.py:74-119   windowing()               (Hann half windows, frame = sig[pm-left .. pm+right])
 *   magphase.py:309-323  zero-pad/truncate to N, circular rotation (epoch -> index 0)
 *   magphase.py:325      np.fft.fft, first H bins
 *   magphase.py:457-476  compute_lossless_feats(): mag=|X|, real=Re X/|X|, imag=Im X/|X| (0 where |X|==0)
 * sig        : float32 PCM of all utterances of the batch, concatenated
 * frame_pos  : int64[n_frames]  absolute index in `sig` of each frame's epoch (utterance offset + pm)
 * frame_left : int32[n_frames]  left length  (pm - previous epoch; == v_shift of the reference)
 * frame_right: int32[n_frames]  right length (next epoch - pm)
 * out_*      : float32 [n_frames x H], row pitch `ld` floats (ld >= H; dense ld == H is the fastest, see mpx_feat_ld)
 * The host keeps the fp64 epoch/index math (np.round, int casts) -- it is never recomputed on the device.
 */
int mpx_analysis_frames(void* stream, int fft_len, const void* tables, const float* sig,
                        const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                        int64_t n_frames, float* out_mag, float* out_real, float* out_imag, int64_t ld);

/*
 * The same analysis with the window, the transform and the epilogue in float64 (features still float32: correctly
 * rounded values of the reference's float64 features instead of values carrying the fp32 FFT's ~1e-6-of-peak noise).
 * For callers whose next step amplifies that noise on weak bins -- the compressed analysis takes ln(mag^2 + 1e-8)
 * and uses Re X/|X| of bins 60-80 dB below the frame peak (magphase.py:2508-2521, libaudio.py:575-601).  About twice
 * the time of mpx_analysis_frames (float64 vector rate, 8 waves per CU).  tables_f64: mpx_tables_f64_bytes() bytes
 * initialised by mpx_tables_f64_init() (float64 twiddles; same life cycle as mpx_tables_init's table).
 * rows_in_use [n_frames] (DEVICE) or NULL: frames with a 0 get their magnitude row only -- out_real / out_imag keep what
 * they held (the compressed analysis never reads the phase features of rows no voiced frame interpolates from:
 * mpx_mel_warp_rows' rows_in_use).
 */
size_t mpx_tables_f64_bytes(int fft_len);
int mpx_tables_f64_init(void* stream, int fft_len, void* tables);
int mpx_analysis_frames_f64(void* stream, int fft_len, const void* tables_f64, const float* sig,
                            const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                            int64_t n_frames, float* out_mag, float* out_real, float* out_imag, int64_t ld,
                            const float* rows_in_use);

/*
 * mpx_analysis_frames_f64 with the WINDOW WEIGHTS READ FROM A HOST-BUILT TABLE instead of evaluated on the device:
 * win_tab (DEVICE, float64) holds, for every half length h <= win_cap, the rising half np.hanning(2 h + 1)[0 .. h] at
 * offset h (h + 1) / 2 -- (win_cap + 1)(win_cap + 2) / 2 doubles, built by the host's numpy (hostmath.hann_half_table), i.e.
 * by the very function the reference windows with (libaudio.py:70-84).  The products sample x weight are then the
 * reference's own float64 values, and a bin that cancels exactly carries the reference's own residue (0.0 or a few 2^-53,
 * with its sign) instead of being flushed to zero.  Frames with a half longer than win_cap use the analytic window.
 * win_tab == NULL: exactly mpx_analysis_frames_f64.
 */
int mpx_analysis_frames_f64w(void* stream, int fft_len, const void* tables_f64, const float* sig,
                             const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                             int64_t n_frames, float* out_mag, float* out_real, float* out_imag, int64_t ld,
                             const float* rows_in_use, const double* win_tab, int32_t win_cap);

/*
 * FUSED compressed analysis at the variable frame rate (magphase.py:2947-2988 with b_const_rate=False: analysis_lossless ->
 * format_for_modelling; SURVEY.md section 8d, configuration C4 "fused; lossless features never hit HBM"): float64 analysis
 * of every frame (as mpx_analysis_frames_f64w) -> log power / unit phasor of every bin -> both mel warps on the matrix
 * cores -> voicing mask and clip, in ONE kernel; the [n_frames x H] lossless features are never written.
 *   wpack, whalf : the two warp matrices in MFMA fragment order (hostmath.pack_warp_fused of the [mag_dim x H] and
 *                  [phase_dim x H] matrices mpx_mel_warp takes, in the layout mpx_analysis_compressed_fused_layout()
 *                  names; DEVICE, float32)
 *   voiced       : float32[n_frames] (DEVICE): 0 = unvoiced frame, phase outputs +0 (magphase.py:2527-2529)
 *   mag_fbank    : 0 = cepstral mel warp of ln(mag^2 + 1e-8) (la.sp_mel_warp), 1 = mel filter bank (la.sp_mel_warp_fbank)
 *   out_mag [n_frames x mag_dim], out_real / out_imag [n_frames x phase_dim], dense float32
 * mag_dim <= 64, phase_dim <= 48, fft_len 2048 or 4096.  Equals mpx_analysis_frames_f64w followed by mpx_mel_warp up to the
 * float32 summation order of the warp.
 */
int mpx_analysis_compressed_fused(void* stream, int fft_len, const void* tables_f64, const float* sig,
                                  const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                                  int64_t n_frames, const double* win_tab, int32_t win_cap, const float* wpack,
                                  const float* whalf, int32_t mag_dim, int32_t phase_dim, const float* voiced,
                                  int32_t mag_fbank, float* out_mag, float* out_real, float* out_imag);
/*
 * The same at the CONSTANT frame rate (analysis_compressed_type1(..., const_rate_ms=5.0), magphase.py:2967-2983: the
 * lossless features are interpolated to the 5 ms grid by interp_from_variable_to_const_frm_rate, :2219-2239, BEFORE
 * format_for_modelling, :2490-2544).  Constant-rate frame c lies between the frames row0[c] and row1[c] (== row0[c] + 1, or
 * == row0[c] for the duplicated first row; row1 ascending over the batch) with weight row_t[c]:
 *   out_mag [n_const x mag_dim]            ln((|X_row0| + t (|X_row1| - |X_row0|))^2 + 1e-8) . W_mag -- the operand rows of
 *                                          the matrix product are built per constant-rate frame inside the kernel;
 *   tmp_real / tmp_imag [n_frames x phase_dim]   the phase streams' warp of the VARIABLE-rate frames with rows_in_use != 0
 *                                          (no mask, no clip), to be finished by mpx_warp_phase_rows -- the split
 *                                          mpx_mel_warp_rows makes.
 * wpack / whalf as for mpx_analysis_compressed_fused (layout 1, eight waves; mel-warp magnitudes only).  work: DEVICE
 * scratch of mpx_analysis_compressed_fused_cr_work_bytes(fft_len, n_frames) bytes (an index of row1 and 8 KB per
 * workgroup for the frame a round hands to the next).  The lossless features never reach HBM.
 */
int mpx_analysis_compressed_fused_cr(void* stream, int fft_len, const void* tables_f64, const float* sig,
                                     const int64_t* frame_pos, const int32_t* frame_left, const int32_t* frame_right,
                                     int64_t n_frames, const double* win_tab, int32_t win_cap, const float* wpack,
                                     const float* whalf, int32_t mag_dim, int32_t phase_dim, const float* rows_in_use,
                                     const int32_t* row0, const int32_t* row1, const float* row_t, int64_t n_const,
                                     float* out_mag, float* tmp_real, float* tmp_imag, void* work);
int64_t mpx_analysis_compressed_fused_cr_work_bytes(int fft_len, int64_t n_frames);
/* column tiles of 16 the fused kernel runs for the magnitude / phase job (what pack_warp_fused must produce) */
int mpx_analysis_compressed_fused_tiles(int32_t mag_dim, int32_t phase_dim, int32_t* ntm, int32_t* ntp);
/* waves per workgroup = frames per round = K slices the packed weights are cut into (pack_warp_fused's n_waves) */
int mpx_analysis_compressed_fused_waves(void);
/* fragment layout of `wpack` this build's mpx_analysis_compressed_fused expects (pack_warp_fused's `layout`): 0 = one
 * v_mfma_f32_16x16x4_f32 fragment per column tile; 1 = the magnitude product on v_mfma_f32_4x4x1_16b_f32 (eight
 * fragments [bin group][half of the 64 coefficients] per wave and chunk), then the phase tiles' fragments */
int mpx_analysis_compressed_fused_layout(void);
/* resident workgroups per CU of the fused kernel (the runtime's occupancy query), < 0 on error: diagnostics */
int mpx_analysis_compressed_fused_blocks_per_cu(int fft_len, int32_t phase_dim);

/*
 * Row pitch (in floats) the lossless feature matrices should be allocated with.  Any ld >= H is CORRECT for every
 * entry point that takes `ld` (so the matrices may live inside wider buffers); mpx_feat_ld() returns the pitch
 * measured fastest on MI355X, which is the reference's dense [F x H] layout, ld == H: padding the rows to a
 * multiple of 32 or 64 floats leaves one partially written 128-byte line per row (the lone Nyquist bin) and costs
 * 7 % of the analysis kernel.  Returns 0 for an unsupported fft_len.
 */
int64_t mpx_feat_ld(int fft_len);

/*
 * Lossless synthesis, per-frame part.  Replaces magphase.py:1761-1770 (synthesis_from_lossless):
 *   X = mag * (real + j imag)/|real + j imag|  (|.|==0 -> 1), Hermitian extension with Im X[0]=Im X[N/2]=0
 *   (libaudio.py:369-388), np.fft.ifft(.).real, np.fft.fftshift  (epoch at index N/2).
 * mag/real/imag : float32 [n_frames x H], row pitch `ld` floats
 * frames_out : float32 [n_frames x N]
 */
int mpx_synthesis_lossless_frames(void* stream, int fft_len, const void* tables, const float* mag,
                                  const float* real, const float* imag, int64_t n_frames, float* frames_out,
                                  int64_t ld);

/*
 * PSOLA overlap-add, gather form, deterministic (ascending frame order).  Replaces magphase.py:34-62 ola():
 * for utterance u, out[t] = sum_i frames[i][t + out_start[u] - pm_rel[i]] over its frames i with
 * 0 <= t + out_start[u] - pm_rel[i] < N, for t in [0, out_off[u+1]-out_off[u]).
 * utt_frame_off : int32[n_utts+1]  frame range of each utterance
 * pm_rel        : int32[n_frames]  pm_i - pm_0 within the utterance (non-decreasing)
 * out_start     : int32[n_utts]    first kept sample of the OLA buffer (python slice start N/2 - pm_0, resolved on host)
 * out_off       : int64[n_utts+1]  sample offsets of each utterance in pcm_out
 * max_out_len   : max over utterances of the output length (grid sizing)
 */
int mpx_ola_gather(void* stream, int fft_len, const float* frames, int32_t n_utts, const int32_t* utt_frame_off,
                   const int32_t* pm_rel, const int32_t* out_start, const int64_t* out_off, int64_t max_out_len,
                   float* pcm_out);

/*
 * Fused lossless synthesis + PSOLA (the production path; the two calls above stay as the reference form).
 * Replaces magphase.py:1759-1776 (synthesis_from_lossless) + :34-62 (ola) for a batch.
 * The frames of every utterance are cut into RUNS of consecutive frames (host: hostmath.ola_runs).  A run is
 * overlap-added, in ascending frame order, in an on-chip ring buffer by one wavefront pair; finished samples go
 * straight to pcm_out, except the first head_end elements of a run that has a predecessor in its utterance -- those
 * positions also receive the predecessor's last frames -- which go to the run's head strip and are added to pcm_out by
 * mpx_ola_fixup afterwards (predecessor's sum + head strip: a fixed order, so the result is deterministic; it differs
 * from the reference's single ascending sum only by fp32 re-association).  Runs of one utterance must be long
 * enough that only ADJACENT runs overlap: pm_rel[frame_end] - pm_rel[frame_begin - 1] >= N for every run that has
 * both neighbours (hostmath.ola_runs guarantees it).
 * Element coordinates e of a run are OLA-buffer positions (magphase.py:38) minus x0.
 */
typedef struct mpx_ola_run {
    int32_t frame_begin, frame_end; /* global frame indices: rows of mag/real/imag, entries of pm_rel */
    int32_t x0;                     /* OLA-buffer position of element 0; chosen so that out_base is a multiple of 64 */
    int32_t head_end;               /* elements e < head_end go to the head strip (0: first run of its utterance) */
    int32_t out_lo, out_hi;         /* elements out_lo <= e < out_hi are written to pcm_out[out_base + e] */
    int32_t flush_end;              /* the ring is streamed out and cleared up to this element when the run ends */
    int32_t fix_lo, fix_hi;         /* mpx_ola_fixup: pcm_out[out_base + e] += strip[e] for fix_lo <= e < fix_hi */
    int32_t pad;
    int64_t out_base;               /* pcm_out index of element 0 (may be negative) */
    int64_t strip_off;              /* float offset of the head strip in `strips`; head_end <= mpx_ola_strip_floats() */
} mpx_ola_run;

/* slot_off : int32[n_slots+1], slot_runs : int32[n_runs] -- work list: pair slot s processes the runs
 *            slot_runs[slot_off[s] .. slot_off[s+1]) in that order (the host balances the lists for
 *            mpx_synth_ola_slots() slots; with runs of equal length every slot gets one)
 * pm_rel   : int32[n_frames]  pm_i - pm_0 within the utterance (as mpx_ola_gather)
 * strips   : float32 [n_runs x mpx_ola_strip_floats(fft_len)]
 * pcm_out  : float32, the utterances' outputs concatenated (every kept sample is written exactly once by
 *            mpx_synthesis_lossless_ola; mpx_ola_fixup then adds the head strips) */
int mpx_synth_ola_slots(void); /* pair slots the current device runs concurrently (CUs x pairs per workgroup) */
/* Relative speed of the slots (HOST float32[n_slots], n_slots = mpx_synth_ola_slots()): a SIMD serves its resident waves
 * by age, so the wave groups of a workgroup run at different, fixed rates; the planner (hostmath.ola_runs) deals the frames
 * of a batch in proportion to these weights so that all slots finish together. */
int mpx_synth_ola_slot_weights(float* weights_host, int32_t n_slots);
int64_t mpx_ola_strip_floats(int fft_len); /* floats per head strip: fft_len + 64 */
int mpx_synthesis_lossless_ola(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                               const float* imag, const mpx_ola_run* runs, int32_t n_runs, const int32_t* slot_off,
                               const int32_t* slot_runs, int32_t n_slots, const int32_t* pm_rel, float* strips,
                               float* pcm_out, int64_t ld /* row pitch of mag/real/imag */);
int mpx_ola_fixup(void* stream, int fft_len, const mpx_ola_run* runs, int32_t n_runs, const float* strips,
                  float* pcm_out);

/*
 * Copy synthesis in one launch: analysis_lossless (magphase.py:2869-2906: windowing :74-119, analysis_with_del_comp_from_pm
 * :266-334, compute_lossless_feats :457-476) followed by synthesis_from_lossless (:1759-1776, ola :34-62) on the same
 * frames, as demos/demo_copy_synthesis_lossless.py:44-50 calls them back to back.  Frame f is cut out of `sig` exactly as
 * by mpx_analysis_frames (frame_pos / frame_left / frame_right), its three feature rows are written to out_mag / out_real /
 * out_imag (row pitch ld) -- analysis_lossless' return values -- and the frame is rebuilt from those float32 values and
 * overlap-added exactly as by mpx_synthesis_lossless_ola (runs / slot_off / slot_runs / pm_rel / strips / pcm_out: the
 * tables of a synthesis plan built from the analysis' v_f0; mpx_ola_fixup afterwards).  pcm_out equals what
 * mpx_synthesis_lossless_ola gives on the rows written here, and the rows equal mpx_analysis_frames', to the last bits
 * of float32 (the same arithmetic, contracted differently by the compiler / another order of the same transform).  The
 * feature rows are not read back from memory.
 * Every frame of [0, n_frames) must belong to exactly one run (slots: mpx_synth_comp_slots(), weights:
 * mpx_roundtrip_slot_weights).
 */
int mpx_roundtrip_slot_weights(float* weights_host, int32_t n_slots); /* as mpx_synth_comp_slot_weights, for this kernel */
int mpx_roundtrip_lossless_ola(void* stream, int fft_len, const void* tables, const float* sig, const int64_t* frame_pos,
                               const int32_t* frame_left, const int32_t* frame_right, int64_t n_frames,
                               const mpx_ola_run* runs, int32_t n_runs, const int32_t* slot_off, const int32_t* slot_runs,
                               int32_t n_slots, const int32_t* pm_rel, float* out_mag, float* out_real, float* out_imag,
                               float* strips, float* pcm_out, int64_t ld);

/* ------------------------------------------------------------------------------------------------------------------
 * Compressed-feature synthesis (magphase.py:825-997 synthesis_from_compressed, b_fbank_mel=False, per_phase_type='magphase')
 * ------------------------------------------------------------------------------------------------------------------ */

/*
 * Mel unwarp of the three feature streams as the linear maps they are:
 *   out_mag  = exp(a_mag  [F x k_mag]   @ u_mag   [k_mag   x n_bins])   la.sp_mel_unwarp + np.exp   (magphase.py:854)
 *   out_real =     a_real [F x k_phase] @ u_phase [k_phase x n_bins]    phase_uncompress_type1_mcep (magphase.py:1219-1235,
 *   out_imag =     a_imag [F x k_phase] @ u_phase                        nearest-neighbour extension folded into u_phase)
 * The matrices are computed on the host in float64 (magphase_amd/hostmath.py: unwarp_matrix, phase_unwarp_matrix).
 */
int mpx_mel_unwarp(void* stream, int64_t n_frames, int32_t n_bins, const float* a_mag, int32_t k_mag,
                   const float* u_mag, float* out_mag, const float* a_real, const float* a_imag, int32_t k_phase,
                   const float* u_phase, float* out_real, float* out_imag,
                   int64_t ld /* row pitch of the three outputs in floats, >= n_bins; see mpx_spec_ld */);

/*
 * mpx_mel_unwarp with the constant -> variable frame-rate interpolation of synthesis_from_compressed folded in
 * (magphase.py:861-868, interp_from_const_to_variable_rate :2242-2252): output row f (n_frames = VARIABLE-rate frames) is
 * the linear interpolation, weight row_t[f], between the unwarped rows row0[f] and row1[f] of the constant-rate
 * coefficient matrices -- out_mag = lerp(exp(U a[row0]), exp(U a[row1])), out_real / out_imag = U lerp(a[row0], a[row1])
 * (the phase unwarp is linear).  The [rows x n_bins] spectra are written once, at the variable rate, and the synthesis
 * kernel reads one row per frame.  row0, row1: DEVICE int32[n_frames]; row_t: DEVICE float32[n_frames].
 * tile_first (optional, with n_rows = number of constant-rate rows): DEVICE int32[ceil(n_rows / 31) + 1], tile_first[T] =
 * the first frame whose row0 >= 31 T (row0 ascending, row1 - row0 in {0, 1}: what the constant -> variable scan yields);
 * the magnitudes are then unwarped once per constant-rate ROW and interpolated out of an LDS tile (same values, 44 % fewer
 * products).  NULL: two products per frame.
 * voiced (optional): DEVICE int32[n_frames]; the PHASE rows of a 32-frame tile without a voiced frame are not computed (out_real /
 * out_imag keep their previous content there): mpx_synthesis_compressed_ola never reads the phase rows of unvoiced frames.
 * n_phase_bins: only the first n_phase_bins bins (rounded up to a multiple of 64) of out_real / out_imag are produced (0: all):
 * above the periodic / aperiodic crossfade (magphase.py:873-876; 6 kHz at 48 kHz = bin 512 of 2049) the periodic curve is
 * exactly zero and the synthesis (n_per_bins below) does not read them.
 */
int mpx_mel_unwarp_rows(void* stream, int64_t n_frames, int32_t n_bins, const float* a_mag, int32_t k_mag,
                        const float* u_mag, float* out_mag, const float* a_real, const float* a_imag, int32_t k_phase,
                        const float* u_phase, float* out_real, float* out_imag, int64_t ld, const int32_t* row0,
                        const int32_t* row1, const float* row_t, int64_t n_rows, const int32_t* tile_first,
                        const int32_t* voiced, int32_t n_phase_bins);

/*
 * Row pitch the unwarped spectra (outputs of mpx_mel_unwarp / mpx_min_phase, inputs of mpx_synthesis_compressed_ola)
 * should be allocated with: n_bins rounded up to a multiple of 32 floats.  mpx_mel_unwarp runs on the matrix cores
 * (v_mfma_f32_32x32x2_f32) and a wave stores 32-float row segments; with 128-byte aligned rows every segment is one
 * full line (measured: 0.80 -> 0.55 ms for the three matrices of 57 k frames), with the dense pitch every segment
 * leaves two partially written lines behind.  Any ld >= n_bins is correct.
 */
int64_t mpx_spec_ld(int32_t n_bins);

/*
 * Device noise source -- OPT-IN replacement of np.random.uniform(-1, 1, n) (magphase.py:883; the reference draws the
 * aperiodic excitation from numpy's global Mersenne twister on the host).  out[offsets[u] + i] = sample i of utterance
 * u = word (i & 3) of Philox4x32-10(counter = i >> 2, key = seeds[u]) mapped to (w >> 8) * 2^-23 - 1 in [-1, 1).
 * A sample depends on (seed, i) only: the result is independent of batching and sharding (bit-identical files from 1 or
 * 8 GPUs).  Same distribution as the reference's source, NOT its sample values; the default mode of the Python layer
 * keeps drawing on the host so that a seeded run reproduces the reference sample for sample.
 * seeds: uint64[n_utts]; offsets: int64[n_utts+1]; max_len: longest utterance (grid sizing).
 */
int mpx_noise_uniform(void* stream, int32_t n_utts, const uint64_t* seeds, const int64_t* offsets, int64_t max_len,
                      float* out);

/*
 * The reference's noise source, np.random.uniform(-1, 1, n) drawn from numpy's GLOBAL generator (magphase.py:883), produced
 * on the device from numpy's own state: key [624] (DEVICE uint32) and pos are np.random.get_state()[1:3]; out [n_samples]
 * gets float32(-1 + 2 d), d = the 53-bit doubles random_sample() would return (two MT19937 words each) -- bit-identical to
 * the host draw --, key_out [624] / pos_out [1] (DEVICE) the state numpy is left in after those draws (the caller puts it
 * back with np.random.set_state).  raw: DEVICE scratch of 2 * n_samples uint32.  work: DEVICE scratch of
 * mpx_noise_numpy_mt19937_work_words() uint32, or NULL.  With work and more than one segment of 319 488 words to draw,
 * the stream is produced by one workgroup per segment: the 624-word window each segment starts from is a jump-ahead of
 * the first one (X[n + J] = xor of X[n + i] over the set bits of x^J mod the generator's characteristic polynomial;
 * log2(segments) rounds of k_mt_seq + k_mt_xor, polynomials from mpx_host_mt19937_jump_polys).  Without work, or for short draws,
 * one workgroup runs the recurrence from the key (454 words per barrier).  A second kernel converts.  Bit-identical
 * either way.  The many-workgroup form waits for the stream once (the polynomials are uploaded from pageable memory);
 * its caller reads key_out / pos_out back right afterwards anyway.
 */
int mpx_noise_numpy_mt19937(void* stream, const uint32_t* key, int32_t pos, int64_t n_samples, uint32_t* raw,
                            float* out, uint32_t* key_out, int32_t* pos_out, uint32_t* work);
int64_t mpx_noise_numpy_mt19937_work_words(void);

/*
 * Host helper (no device): out [n_levels x 624] uint32 (HOST), level l = the bits of x^(jump_words * 2^l) mod phi, phi the
 * characteristic polynomial (degree 19937) of MT19937's word recurrence; bit i of a level = bit (i & 31) of word i >> 5.
 * For the words X[n] the recurrence produces (n >= 624 counted from a key), X[n + jump] = xor_{i : bit i} X[n + i].
 * phi is found once per process (Berlekamp-Massey), ladders are cached per jump_words.
 */
int32_t mpx_host_mt19937_jump_poly(int64_t jump_words, int32_t n_levels, uint32_t* out);
/* The same for n arbitrary jumps, independent of each other (no cache), on up to n_threads host threads: out [n x 624].
 * mpx_noise_numpy_mt19937's ladder takes the multiples p * J * R^l (p < R; R = 2: the doubling ladder) of its segment
 * length from here. */
int32_t mpx_host_mt19937_jump_polys(const int64_t* jumps, int32_t n, uint32_t* out, int32_t n_threads);

/*
 * Noise-gain statistics (magphase.py:886-903, Q10/Q11): for every frame, the windowed noise frame
 * (frame_wtype 0: Hann halves, 1: np.bartlett**2.5 halves; epoch at index 0) is transformed and
 * out_sum[f] = sum_{k=1}^{N/2-1} (ln|Ns[k]|)^2.  The host turns the per-class means into the two gains per utterance.
 * noise / frame_pos / frame_left / frame_right as sig / frame_* of mpx_analysis_frames.
 */
int mpx_noise_stats(void* stream, int fft_len, const void* tables, const float* noise, const int64_t* frame_pos,
                    const int32_t* frame_left, const int32_t* frame_right, const int32_t* frame_wtype,
                    int64_t n_frames, float* out_sum);
/*
 * "Noise spectra once" form of the pair mpx_noise_stats -> mpx_synthesis_compressed_ola (fft_len 4096 only; the same
 * reference lines, magphase.py:886-903 and :908-976): mpx_noise_stats_spectra also stores every frame's noise spectrum
 * (mpx_noise_spectra_floats(fft_len, n_frames) floats, in the kernels' own register layout), and
 * mpx_synthesis_compressed_ola_spectra (one row per frame: row0 / row1 / row_t NULL) loads it instead of transforming
 * the noise frame a second time.  Trades the second FFT for 17.4 KB of HBM traffic per frame each way; opt-in
 * (MAGPHASE_NOISE_SPECTRA=store), measured in docs/LAB_NOTES.md (round 5).
 */
int64_t mpx_noise_spectra_floats(int fft_len, int64_t n_frames);
int mpx_noise_stats_spectra(void* stream, int fft_len, const void* tables, const float* noise, const int64_t* frame_pos,
                            const int32_t* frame_left, const int32_t* frame_right, const int32_t* frame_wtype,
                            int64_t n_frames, float* out_sum, float* spectra);

/*
 * Spectrum assembly + inverse FFT + anti-ringing window + PSOLA for compressed-feature synthesis
 * (magphase.py:908-976; run / slot / strip tables and pcm_out exactly as mpx_synthesis_lossless_ola, followed by mpx_ola_fixup).
 * Per frame f (all arrays int32/float32[n_frames] unless noted):
 *   noise_pos(int64)/noise_left/noise_right/noise_wtype : this frame's noise frame (recomputed here)
 *   voiced, inv_gain                                    : class flag and 1/gain of its class
 *   row0,row1,row_t : the frame's mag/real/imag row = (1-row_t)*row[row0] + row_t*row[row1] of the [rows x H] matrices
 *                     (constant -> variable frame rate, magphase.py:2242-2252; row0 == row1 for variable-rate input);
 *                     all three NULL: one row per frame, row index = frame index (variable-rate input, or rows already
 *                     interpolated by mpx_mel_unwarp_rows -- half the feature loads)
 *   win_left, win_right : anti-ringing window half lengths (Q14);  pm_rel : as mpx_ola_gather
 * per_v, ap_v, ap_u : float32[H] per-bin constants (hostmath.synthesis_bin_curves: tilt x sqrt(mask) etc., Q12/Q13)
 * n_per_bins : per_v[k] == 0 for k >= n_per_bins (the caller's promise; 0 or >= H: no promise): real / imag and per_v of those
 *              bins are not read (one row per frame form only).
 */
int mpx_synth_comp_slots(void); /* wave slots of mpx_synthesis_compressed_ola on the current device */
/* Relative speed of the compressed synthesis kernel's slots (see mpx_synth_ola_slot_weights). */
int mpx_synth_comp_slot_weights(float* weights_host, int32_t n_slots);
int mpx_synthesis_compressed_ola(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                                 const float* imag, const float* noise, const int64_t* noise_pos,
                                 const int32_t* noise_left, const int32_t* noise_right, const int32_t* noise_wtype,
                                 const int32_t* voiced, const float* inv_gain, const int32_t* row0,
                                 const int32_t* row1, const float* row_t, const int32_t* win_left,
                                 const int32_t* win_right, const int32_t* pm_rel, const float* per_v,
                                 const float* ap_v, const float* ap_u, const mpx_ola_run* runs, int32_t n_runs,
                                 const int32_t* slot_off, const int32_t* slot_runs, int32_t n_slots,
                                 float* strips, float* pcm_out,
                                 int64_t ld /* row pitch of mag/real/imag in floats, >= fft_len/2 + 1 */,
                                 int32_t n_per_bins);
int mpx_synthesis_compressed_ola_spectra(void* stream, int fft_len, const void* tables, const float* mag, const float* real,
                                 const float* imag, const float* noise, const int64_t* noise_pos,
                                 const int32_t* noise_left, const int32_t* noise_right, const int32_t* noise_wtype,
                                 const int32_t* voiced, const float* inv_gain, const int32_t* row0,
                                 const int32_t* row1, const float* row_t, const int32_t* win_left,
                                 const int32_t* win_right, const int32_t* pm_rel, const float* per_v,
                                 const float* ap_v, const float* ap_u, const mpx_ola_run* runs, int32_t n_runs,
                                 const int32_t* slot_off, const int32_t* slot_runs, int32_t n_slots,
                                 float* strips, float* pcm_out, int64_t ld, int32_t n_per_bins,
                                 const float* spectra /* as stored by mpx_noise_stats_spectra */);

/*
 * HOST function (no device work, no stream): the serial constant -> variable frame-rate scan of
 * magphase.py:1426-1449 (get_shifts_and_frm_locs_from_const_shifts, Q16) in the reference's float64 operation
 * sequence (scipy interp1d's linear formula, no FMA): bit-identical shifts / frame locations, ~100x faster than one
 * scipy call per step.  centres, shift_c: HOST float64[n] (centres ascending); shifts_out, locs_out: HOST float64[2n].
 * Returns the index `start` of the first valid element (results are out[start .. 2n)), or -1 on bad arguments.
 */
int64_t mpx_host_const_to_var_scan(const double* centres, const double* shift_c, int64_t n, double* shifts_out,
                                   double* locs_out);

/* ------------------------------------------------------------------------------------------------------------------
 * Compressed-feature analysis (magphase.py:2490-2544 format_for_modelling, :2947-2988 analysis_compressed)
 * ------------------------------------------------------------------------------------------------------------------ */

/*
 * Mel warp of the lossless features (la.sp_mel_warp, libaudio.py:643-661 = SPTK-3.9 ``mcep -j 0`` + alpha=0 cosine matrix;
 * the SPTK part is a restatement, PARITY UNPINNED) as linear maps of the log-periodogram, plus the epilogues of
 * format_for_modelling:
 *   out_mag  [F x mag_dim]   = W_mag   . ln(mag^2 + 1e-8)                                  (== la.log(sp_mel_warp(mag)))
 *   out_real [F x phase_dim] = clip(voiced * (W_phase . ln(exp(real)^2 + 1e-8)), -1, 1)    (same for imag)
 * W_mag [mag_dim x n_bins], W_phase [phase_dim x n_bins] float32 from hostmath.warp_matrix (float64 on the host).
 * row0/row1/row_t (all null, or int32/int32/float32 [F]): output frame f reads the interpolated input row
 * (1-row_t)*x[row0] + row_t*x[row1] (variable -> constant frame rate, magphase.py:2219-2239); null = row f.
 */
int mpx_mel_warp(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real, const float* imag,
                 const int32_t* row0, const int32_t* row1, const float* row_t, const float* w_mag, int32_t mag_dim,
                 const float* w_phase, int32_t phase_dim, const float* voiced, float* out_mag, float* out_real,
                 float* out_imag, int64_t ld /* row pitch of mag/real/imag in floats, >= n_bins */);

/*
 * mpx_mel_warp with the magnitudes compressed by the mel FILTER BANK instead of the cepstral warp
 * (format_for_modelling(b_mag_fbank_mel=True), magphase.py:2504-2510; la.sp_mel_warp_fbank / apply_fbank 'average',
 * libaudio.py:721-769): out_mag = la.log(exp(W_fbank . la.log(mag))), la.log's floor (-1e10 for mag == 0, and where the
 * exp underflows) included.  w_fbank: [mag_dim x n_bins] (hostmath.warp_fbank_matrix).  The phase streams are unchanged.
 */
int mpx_mel_warp_fbank(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real,
                       const float* imag, const int32_t* row0, const int32_t* row1, const float* row_t,
                       const float* w_fbank, int32_t mag_dim, const float* w_phase, int32_t phase_dim, const float* voiced,
                       float* out_mag, float* out_real, float* out_imag, int64_t ld);

/*
 * mpx_mel_warp / mpx_mel_warp_fbank (mag_fbank != 0) with the row tables given, and the phase streams warped on the
 * VARIABLE-rate rows: the phase prologue ln(e^{2x} + 1e-8) = 2x + 1e-8 e^{-2x} is linear up to its floor term, so the
 * product is formed once per variable-rate row (n_var_rows rows of real / imag, no row interpolation in the operand
 * load: half the loads of the phase jobs, 11 % fewer rows at a 5 ms rate) into tmp_real / tmp_imag [n_var_rows x phase_dim]
 * (DEVICE scratch) and the phase_dim outputs are interpolated to the constant rate, masked and clipped afterwards
 * (magphase.py:2219-2239 then :2520-2532 in the other order; difference < 1e-8 per bin).  rows_in_use [n_var_rows]: != 0
 * for the rows a voiced constant-rate frame interpolates from -- 64-row tiles without one are not computed.  With
 * tmp_real == NULL: exactly mpx_mel_warp / mpx_mel_warp_fbank.
 */
int mpx_mel_warp_rows(void* stream, int64_t n_frames, int32_t n_bins, const float* mag, const float* real,
                      const float* imag, const int32_t* row0, const int32_t* row1, const float* row_t, const float* w_mag,
                      int32_t mag_dim, const float* w_phase, int32_t phase_dim, const float* voiced, float* out_mag,
                      float* out_real, float* out_imag, int64_t ld, int32_t mag_fbank, int64_t n_var_rows,
                      const float* rows_in_use, float* tmp_real, float* tmp_imag);

/*
 * The second half of mpx_mel_warp_rows' phase streams on its own: out[f] = clip(voiced[f] * ((1 - t) tmp[row0[f]] +
 * t tmp[row1[f]])) for the phase_dim columns of both streams, 0 where voiced[f] == 0 (magphase.py:2219-2239, :2527-2532).
 */
int mpx_warp_phase_rows(void* stream, int64_t n_frames, int32_t phase_dim, const float* tmp_real, const float* tmp_imag,
                        const int32_t* row0, const int32_t* row1, const float* row_t, const float* voiced, float* out_real,
                        float* out_imag);

/*
 * Minimum-phase spectrum of a magnitude spectrum by the complex cepstrum (la.build_min_phase_from_mag_spec,
 * libaudio.py:920-934; synthesis_from_compressed(per_phase_type='min_phase'), magphase.py:937-938).
 * For output frame f: m = (1-row_t)*mag[row0] + row_t*mag[row1] ([rows x H]); out_mag[f] = m and
 * (out_real, out_imag)[f] = (cos phi, sin phi), phi = Im FFT(causal fold(IFFT(ln m))) -- the unit phasor the
 * synthesis kernel expects in place of the unwarped phase features ([n_frames x H] each).
 */
int mpx_min_phase(void* stream, int fft_len, const void* tables, const float* mag, const int32_t* row0,
                  const int32_t* row1, const float* row_t, int64_t n_frames, float* out_mag, float* out_real,
                  float* out_imag, int64_t ld /* row pitch of mag and of the three outputs, floats */);

/*
 * Noise gains on the device (magphase.py:902-906, Q10): per utterance u and class c (0 voiced, 1 unvoiced)
 * g = sqrt(exp( sum of out_sum over the class's frames / (n_frames_of_class * bins_per_frame) )), float64;
 * inv_gain[f] = 1/g(class of f) (float32, input of mpx_synthesis_compressed_ola); gains (optional, may be null):
 * float64 [n_utts x 2].  bins_per_frame = N/2 - 1.
 */
int mpx_noise_gains(void* stream, const float* sums, const int32_t* voiced, const int32_t* utt_frame_off,
                    int32_t n_utts, int32_t bins_per_frame, float* inv_gain, double* gains);

/*
 * MagPhase post-filter (magphase.py:2300-2378, Q20) on [n_frames x dim] log-mel magnitudes.  half_len: int32
 * [nx_last - nx_first + 1] half lengths of the centred moving average of bins nx_first..nx_last (host table,
 * magphase.py:2342-2344); tilt: float32[dim] enhancement factors (np.linspace(boost_at_zero, boost_at_nyq, dim)).
 */
int mpx_post_filter(void* stream, const float* mag_mel_log, int64_t n_frames, int32_t dim, const int32_t* half_len,
                    int32_t nx_first, int32_t nx_last, const float* tilt, float* out);

/*
 * Output high-pass filter (magphase.py:981-995: scipy.signal.butter(4, 40 Hz) + lfilter) in float64 on the device,
 * as a cascade of the two second-order sections of the same Butterworth design (scipy output='sos'), each run as
 * a blocked scan: zero-state recurrence per block of mpx_hpf_block() samples, block end states chained per
 * utterance through A^B, free responses added from G[n] = C A^n.  (The 4th-order direct form cannot be chained in
 * float64: clustered poles at |z| ~ 0.997.  The cascade agrees with lfilter to ~1e-7 of peak = lfilter's own noise.)
 * sos_host: HOST pointer to 2 x 6 float64 (b0 b1 b2 a0 a1 a2 per section); pmat: float64 [2 x 4] (A^B row-major per
 * section); gtab: float64 [2 x B x 2]; blk_off: int32[n_utts+1] cumulative block counts; zend/zstart: float64
 * scratch [total_blocks x 2]; y_tmp, y: float64 [total samples] (y = output).  Tables: hostmath.hpf_tables.
 */
int mpx_hpf_block(void);
int mpx_output_hpf(void* stream, const float* pcm, const int64_t* out_off, const int32_t* blk_off, int32_t n_utts,
                   int64_t max_len, const double* sos_host, const double* pmat, const double* gtab, double* zend,
                   double* zstart, double* y_tmp, double* y);

/* ------------------------------------------------------------------------------------------------------------------
 * Built-in epoch / voicing front end.  NOT REAPER (libaudio.py:450-455 shells out to it): parity
 * unpinned, opt-in (magphase_amd/epochs.py documents the algorithm and its quality checks).  Batched over utterances.
 * ------------------------------------------------------------------------------------------------------------------ */

/*
 * F0 / voicing candidates: box-decimation by `dec` (to ~4 kHz), then per 5 ms frame the normalised cross-correlation
 * for n_lags lags starting at l_min; the shortest lag within 0.06 of the best is refined by a parabola.
 * sig: float32 PCM, utterances at off[0..n_utts]; dec_off / frame_off: int64[n_utts+1] offsets of the decimated
 * signals (scratch xd, float64) and of the frames in the outputs; means: float64[2 n_utts] scratch.
 * Outputs per frame: f0 (fs_d / lag), peak (NCCF at the chosen lag), energy (sum of squares of the frame's window).
 */
int mpx_epoch_f0_track(void* stream, const float* sig, const int64_t* off, int32_t n_utts, int32_t dec,
                       const int64_t* dec_off, int64_t max_dec_len, double* xd, double* means, const int64_t* frame_off,
                       int64_t max_frames, int32_t hop, int32_t win, int32_t l_min, int32_t n_lags, double fs_d,
                       float* f0, float* peak, float* energy);

/*
 * Epoch candidates by zero-frequency filtering: x differenced, through two zero-frequency resonators (cumulative
 * sums, float64) with the local mean over 2 half_win[u] + 1 samples removed after the first and three times after the
 * second; then the zero crossings of both directions of the result.  List p of utterance u (p = 0: negative-going,
 * 1: positive-going) holds counts[2u + p] entries at [(2u + p) * cap ...): sample index, |slope|, and the excitation
 * energy of the w_score samples after the crossing minus the w_score before it (which direction carries the epochs
 * depends on the recording's polarity: the host keeps the one with the larger mean score).  Entries are unordered.
 * cross_frac (may be null): the zero of the line through the two samples of the sign change, as the fraction of a sample
 * BEFORE sample index cross_idx (0 .. 1): sub-sample crossing position = cross_idx - cross_frac.
 * buf_a/b/c: float64 scratch of the total sample count each.
 */
int mpx_epoch_zff(void* stream, const float* sig, const int64_t* off, int32_t n_utts, int64_t max_len,
                  const int32_t* half_win, int32_t w_score, double* buf_a, double* buf_b, double* buf_c, int32_t cap,
                  int32_t* counts, int32_t* cross_idx, float* cross_slope, float* cross_score, float* cross_frac);

/*
 * 16-bit PCM of the synthesised utterances for the wav writer (libaudio.py:352-365 write_audio_file, Q17): per
 * utterance v = norm * y / max|y| in float64 (skipped when norm <= 0), then lrint(v * 0x7FFF) as libsndfile converts
 * floats to PCM_16 -- the host form's IEEE operations in the same order, so the samples are bit-identical to
 * la.write_audio_file's.  y: float64 (y_is_f64 != 0; the output of mpx_output_hpf) or float32 samples, utterances
 * concatenated at out_off (int64[n_utts+1]); peaks: float64[n_utts] scratch (receives max|y|); out: int16, same layout.
 */
int mpx_pcm16(void* stream, const void* y, int32_t y_is_f64, const int64_t* out_off, int32_t n_utts, int64_t max_len,
              double norm, double* peaks, int16_t* out);

/*
 * int16 PCM -> float32 samples in [-1, 1) (x / 32768, exact): the conversion la.read_audio_file's caller needs
 * (libaudio.py:369-377 returns float64 in [-1, 1); the analysis reads float32, magphase.py:2869-2879), done on the
 * device so that 16-bit wavs cross PCIe as int16.  pcm: DEVICE int16[n], 8-byte aligned; out: DEVICE float32[n],
 * 16-byte aligned.
 */
int mpx_pcm16_to_f32(void* stream, const int16_t* pcm, int64_t n, float* out);

/*
 * Merlin / HTS style post-filter of the log mel magnitudes (magphase.py:3375-3465: the reference pipes each utterance
 * through nine SPTK-3.9 binaries -- x2x | freqt | c2acr, vopr, mc2b | bcp | merge | b2mc -- SPTK restated from its published
 * algorithms, PARITY UNPINNED) for all frames of a batch (csrc/magphase_merlin.hip):
 *   mcep = x . c1 (la.rceps 'log' / 'compact' as a matrix);  mcep_w = mcep * lifter (1, 1, pf, pf, ...);
 *   r0, p_r0 = sum_k wk[k] exp(2 (mcep | mcep_w) . g)[k]   (frame energy: freqt to the linear axis + c2acr -M 0 -l 4096);
 *   b = mc2b(mcep_w, alpha); b[0] += ln(r0 / p_r0) / 2; mcep_pf = b2mc(b, alpha);  out = mcep_pf . cf, NaN -> magic.
 * mag_mel_log, out: float32 [n_frames x dim] (3 <= dim <= 64); c1, cf: [dim x dim]; lifter: [dim]; g: [dim x n_bins];
 * wk: [n_bins] (hostmath.merlin_tables builds them in float64); mcep, mcep_w: scratch [n_frames x dim]; r0, p_r0: scratch
 * [n_frames].  Deterministic (no atomics).
 */
int mpx_post_filter_merlin(void* stream, const float* mag_mel_log, int64_t n_frames, int32_t dim, const float* c1,
                           const float* lifter, const float* g, const float* wk, int32_t n_bins, double alpha,
                           const float* cf, double magic, float* mcep, float* mcep_w, float* r0, float* p_r0, float* out);

/*
 * Memory-rate probe (csrc/magphase_probe.hip; measurement only, not on the MagPhase path): one streaming float4 kernel
 * over n_floats (a multiple of 4) elements.  mode = kind + 16 * shape: kind 0 reads `a` (b: one float of scratch), kind 1
 * fills `a`, kind 2 copies a -> b; shape < mpx_bw_probe_shapes() selects the launch shape (grid x block, independent
 * 16-byte accesses in flight per lane, non-temporal bit; shape 0 = 2048 x 256 with one access in flight).  bench.py times
 * every shape with HIP events and quotes the best per kind as the device's own streaming read / write / copy ceilings
 * beside the 8 TB/s spec peak in its roofline object.  The reference has no counterpart.
 */
int mpx_bw_probe_shapes(void);
int mpx_bw_probe(void* stream, int32_t mode, float* a, float* b, int64_t n_floats);

/* ------------------------------------------------------------------------------------------------------------------
 * Host-side file helpers of the batch scripts (csrc/magphase_host.cpp; no device work, no stream).  Called from the
 * reader / writer threads of iobatch.py: the FFI call drops the interpreter lock, so reading, computing and writing overlap.
 * ------------------------------------------------------------------------------------------------------------------ */

/*
 * Batch planners (csrc/magphase_plan.cpp): the reference's float64 / integer index arithmetic for all utterances of a
 * batch in one call, the same IEEE-754 operation sequence as the numpy forms in magphase_amd/hostmath.py / engine.py
 * (np.round = half to even, astype(int) = truncation, sequential cumsum).  All pointers are HOST memory.  A negative
 * return value -(u + 2) names utterance u as the one the numpy form would raise on; -1 = bad arguments.
 *
 * mpx_host_plan_analysis: libaudio.py:435-447 (epoch clean-up), magphase.py:77-98 (frame bounds), :2198-2207 (f0).
 *   pm_sec, voi [ep_off[n_utts]]: epochs (seconds) and voicing flags, utterance u at [ep_off[u], ep_off[u+1]);
 *   n_smpls, fs [n_utts]; sig_off [n_utts]: offset of the utterance's samples in the batch's sample buffer.
 *   Outputs (capacity = number of epochs): pos = rounded epoch + sig_off, pm = rounded epoch, left / right = distances to
 *   the neighbouring epochs (0 and n_smpls - 1 at the ends), f0 = voi * fs / left; frame_off [n_utts + 1].
 *   Returns the total number of frames.
 */
int64_t mpx_host_plan_analysis(int32_t n_utts, const double* pm_sec, const double* voi, const int64_t* ep_off,
                               const int64_t* n_smpls, const double* fs, const int64_t* sig_off, int64_t* pos,
                               int64_t* pm_out, int64_t* left, int64_t* right, double* f0, int64_t* frame_off);

/*
 * mpx_host_plan_analysis_batch: the WHOLE host side of a lossless / compressed analysis launch in one call, the utterances
 * given by pointer (nothing is concatenated by the caller), on n_threads threads -- what the reference does once per
 * utterance in its Pool worker (libutils.py:32-63; scripts/batch_feature_extraction_for_tts.py:40-57):
 *   - the samples of all utterances copied into `stage`, the page-locked buffer the H2D DMA reads (stage_kind 0: every
 *     utterance is int16, staged as int16 and widened on the device by mpx_pcm16_to_f32; 1: float32, int16 * 2^-15 /
 *     float64 rounded to nearest even; pcm_kind[u] 0 int16, 1 float32, 2 float64; stage may be null: no copy);
 *   - mpx_host_plan_analysis' arithmetic per utterance (libaudio.py:435-447, magphase.py:77-98, :2198-2207) with the device
 *     tables written in their final types: pos int64, left32 / right32 int32, voi32 float32 (f0 > 0; may be null);
 *   - host results: pm, left64 (the reference's v_shift), f0, f0_med = scipy.signal.medfilt(f0) per utterance (kernel 3,
 *     zero-padded; magphase.py:2499-2500; may be null), frame_off [n_utts + 1];
 *   - frames longer than fft_len (the reference warns once per such frame, magphase.py:311-315): their global frame index
 *     and length in long_frame / long_len (capacity long_cap), their number in *n_long_out (fft_len <= 0: not looked for).
 * Every per-frame output has capacity sum(n_epochs).  Returns the number of frames, -(u + 2) for the first utterance the
 * numpy form raises on (no epochs), -1 for bad arguments.
 */
int64_t mpx_host_plan_analysis_batch(int32_t n_utts, const void* const* pcm, const int32_t* pcm_kind,
                                     const int64_t* n_smpls, const double* fs, const double* const* pm_sec,
                                     const double* const* voi, const int64_t* n_epochs, void* stage, int32_t stage_kind,
                                     int64_t* pos, int32_t* left32, int32_t* right32, float* voi32, int64_t* pm_out,
                                     int64_t* left64, double* f0, double* f0_med, int64_t* frame_off, int32_t fft_len,
                                     int64_t* long_frame, int64_t* long_len, int64_t long_cap, int64_t* n_long_out,
                                     int32_t n_threads);

/*
 * mpx_host_plan_synthesis: the per-utterance part of synthesis_from_compressed before any spectrum (magphase.py:846-848
 * f0 -> voicing / shifts, :861-868 constant -> variable rate via mpx_host_const_to_var_scan, :879-882 epochs and noise
 * length, :77-98 noise frame bounds, :969-973 anti-ringing window lengths, :34-62 OLA offsets).  f0 = exp(lf0) of the
 * rows [row_off[u], row_off[u+1]) (the caller evaluates the exp).  Per-frame outputs have capacity `cap` (2 x rows is
 * always enough); row0 / row1 index the batch's coefficient rows; npos is relative to the batch's noise buffer.
 * Returns the total number of (variable-rate) frames.
 */
int64_t mpx_host_plan_synthesis(int32_t n_utts, const double* f0, const int64_t* row_off, double fs, int32_t fft_len,
                                int32_t b_const_rate, int32_t b_voi_ap_win, int64_t cap, int64_t* v_shift, int64_t* v_pm,
                                int64_t* npos, int32_t* nleft, int32_t* nright, int32_t* wtype, int32_t* voiced,
                                int32_t* row0, int32_t* row1, double* rowt, int32_t* win_l, int32_t* win_r,
                                int64_t* pm_rel, int64_t* frame_off, int64_t* ns_len_out, int64_t* out_start,
                                int64_t* out_len);

/*
 * mpx_host_plan_lossless_synthesis: synthesis_from_lossless's epochs and OLA offsets (magphase.py:1771-1772: v_pm =
 * cumsum(f0_to_shift(f0, fs)).astype(int) -- float cumsum, then truncation (Q3); :34-62) for the utterances
 * [frame_off[u], frame_off[u+1]) of f0.  Outputs v_pm, pm_rel [frames], out_start, out_len [n_utts].
 */
int64_t mpx_host_plan_lossless_synthesis(int32_t n_utts, const double* f0, const int64_t* frame_off, const double* fs,
                                         int32_t fft_len, int64_t* v_pm, int64_t* pm_rel, int64_t* out_start,
                                         int64_t* out_len);

/*
 * mpx_host_ola_runs: the run planner of mpx_synthesis_lossless_ola / mpx_synthesis_compressed_ola (see mpx_ola_run) in
 * its default mode: the batch's frames are cut at `gcuts` (equal shares of the frame sequence, one per wave-pair slot)
 * and at utterance boundaries; cuts that would let non-adjacent runs overlap are dropped.  pm_rel / frame_off: frame
 * positions relative to each utterance's first frame; starts / out_lens: ola's kept part per utterance; out_offs:
 * utterance offsets in pcm_out.  Returns the number of runs written (capacity n_utts + n_gcuts is always enough).
 */
int64_t mpx_host_ola_runs(int32_t n_utts, const int64_t* pm_rel, const int64_t* frame_off, const int64_t* starts,
                          const int64_t* out_lens, const int64_t* out_offs, int32_t fft_len, const int64_t* gcuts,
                          int64_t n_gcuts, mpx_ola_run* runs, int64_t cap_runs);

/*
 * mpx_host_plan_synthesis_batch: the whole host side of a compressed-feature synthesis launch (what the reference does per
 * utterance in scripts/batch_waveform_generation.py:28-58 -> magphase.py:3229-3275 -> :836-897, :969-976), utterances by
 * pointer, on n_threads threads:
 *   - the coefficient matrices (kind[u] 1 float32 / 2 float64, C-contiguous [n_rows[u] x mag_dim | phase_dim]) copied /
 *     narrowed into `stage` (page-locked, float32) as [R x mag_dim | R x phase_dim | R x phase_dim], R = sum n_rows; null:
 *     no copy;
 *   - mpx_host_plan_synthesis' arithmetic per utterance on f0 = exp(lf0) (concatenated, evaluated by the caller with
 *     numpy's exp), mpx_host_ola_runs over n_slots slots (wcum / wsum: np.concatenate(([0.], np.cumsum(w))) and w.sum() of
 *     the slots' float64 weights, null = equal shares), the slots' work lists and, with want_tiles, the frames of every
 *     31-row tile of the coefficient matrix (mpx_mel_unwarp_rows);
 *   - every device table written in its final type into `desc` (page-locked; one H2D copy), table k at byte offset
 *     desc_off[k], 256-byte aligned, in this order: utt_frame_off i32[U+1], npos i64[F], nleft, nright, wtype, voiced i32[F],
 *     tile_first i32[tiles+1], row0, row1 i32[F], rowt f32[F], win_l, win_r, pm_rel i32[F], out_start i32[U], out_off
 *     i64[U+1], runs (mpx_ola_run[n_runs]), slot_off i32[slots+1], slot_runs i32[n_runs]      (desc_off: int64[18]);
 *   - host results: v_shift, v_pm (int64), voiced_host (int32) with capacity 2 R + 2 U; frame_off [U+1]; ns_len, out_start,
 *     out_len [U]; runs_host (capacity runs_cap >= U + n_slots + 1);
 *     counts[8] = {frames, runs, slots in use, bytes of desc in use, noise samples, output samples, tiles + 1, R}.
 * Returns the number of frames; -(u + 2): the numpy form raises on utterance u; -4000000: weighted shares and fewer frames
 * than slots (counts[0] = frames: call again with n_slots = frames and the first `frames` weights' cumsum / sum); other
 * values <= -1000000: a capacity / table-order case left to the numpy form; -1: bad arguments.
 */
int64_t mpx_host_plan_synthesis_batch(int32_t n_utts, const void* const* mag, const void* const* real,
                                      const void* const* imag, const int32_t* kind, const int64_t* n_rows,
                                      int32_t mag_dim, int32_t phase_dim, float* stage, const double* f0, double fs,
                                      int32_t fft_len, int32_t b_const_rate, int32_t b_voi_ap_win, int32_t n_slots,
                                      const double* wcum, double wsum, int32_t want_tiles, uint8_t* desc,
                                      int64_t desc_cap, int64_t* desc_off, int64_t* v_shift, int64_t* v_pm,
                                      int32_t* voiced_host, int64_t* frame_off, int64_t* ns_len, int64_t* out_start,
                                      int64_t* out_len, mpx_ola_run* runs_host, int64_t runs_cap, int64_t* counts,
                                      int32_t n_threads);

/* dst[i] = (double)src[i], i < n, on a few threads: the float32 -> float64 widening of the array API's outputs (the
 * reference's arrays are float64, magphase.py:457-476; numpy's astype is one thread at ~1.5 GB/s). */
int32_t mpx_host_widen_f32(const float* src, double* dst, int64_t n, int32_t n_threads);

/* n byte ranges src[i][0 .. nbytes[i]) copied to dst + dst_off[i] on n_threads threads (HOST pointers): the samples of a
 * batch's utterances into the page-locked staging buffer of the analysis plan.  Returns 0, or MPX_ERR_ARG. */
int32_t mpx_host_copy_many(int32_t n, const void* const* src, const int64_t* nbytes, const int64_t* dst_off, void* dst,
                           int32_t n_threads);

/* dst[i] = (float)src[i] (round to nearest even, numpy's astype), i < n, on a few threads: the array API's float64 inputs
 * on their way to the device. */
int32_t mpx_host_narrow_f64(const double* src, float* dst, int64_t n, int32_t n_threads);

/* sizes[i] = size of paths[i] in bytes, or -errno. */
int32_t mpx_host_file_sizes(int32_t n, const char* const* paths, int64_t* sizes);

/*
 * np.loadtxt(est_file, skiprows=skiprows, usecols=[0, 1]) for n files at once (libaudio.py:421-447 read_reaper_est_file's
 * read): the first two whitespace-separated columns of every non-blank line after `skiprows` lines, as float64 (correctly
 * rounded decimal -> binary, the values strtod gives).  File i writes its rows to col0 / col1 [row_off[i], row_off[i+1])
 * (HOST float64); counts[i] = rows, or -errno (-EINVAL: a row with one column, -ENOSPC: more rows than its slice holds;
 * a file has at most size / 4 + 1 rows).  n_threads <= 1: in the calling thread.
 */
int32_t mpx_host_read_est_batch(int32_t n, const char* const* paths, int32_t skiprows, const int64_t* row_off,
                                double* col0, double* col1, int64_t* counts, int32_t n_threads);

/*
 * n files written at once: headers[i] (header_bytes[i] bytes; headers may be NULL) followed by bodies[i]
 * (body_bytes[i] bytes) -- lu.write_binfile (libutils.py:193-199) for the feature files, the RIFF header + samples of
 * la.write_audio_file (libaudio.py:352-365) for wavs.  status[i] = 0 or errno.
 */
int32_t mpx_host_write_files(int32_t n, const char* const* paths, const void* const* headers, const int64_t* header_bytes,
                             const void* const* bodies, const int64_t* body_bytes, int32_t* status, int32_t n_threads);

/* n files read at once into bufs[i] (at most cap[i] bytes): got[i] = bytes read or -errno (lu.read_binfile's np.fromfile). */
int32_t mpx_host_read_files(int32_t n, const char* const* paths, void* const* bufs, const int64_t* cap, int64_t* got,
                            int32_t n_threads);


#ifdef __cplusplus
}
#endif
#endif /* MAGPHASE_HIP_H */
