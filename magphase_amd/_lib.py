"""
ctypes binding of libmagphase_hip.so (C ABI: include/magphase_hip.h).

There is NO CPU fallback: if the library is missing or no ROCm device is visible the product raises.
"""
import ctypes
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmagphase_hip.so")

# every symbol include/magphase_hip.h declares (tests/test_cabi_symbols.py checks the .so exports them all)
SYMBOLS = (
    "mpx_version",
    "mpx_last_error",
    "mpx_tables_bytes",
    "mpx_tables_init",
    "mpx_feat_ld",
    "mpx_analysis_frames",
    "mpx_tables_f64_bytes",
    "mpx_tables_f64_init",
    "mpx_analysis_frames_f64",
    "mpx_analysis_frames_f64w",
    "mpx_analysis_compressed_fused",
    "mpx_analysis_compressed_fused_cr",
    "mpx_analysis_compressed_fused_cr_work_bytes",
    "mpx_analysis_compressed_fused_tiles",
    "mpx_analysis_compressed_fused_waves",
    "mpx_analysis_compressed_fused_layout",
    "mpx_analysis_compressed_fused_blocks_per_cu",
    "mpx_synthesis_lossless_frames",
    "mpx_ola_gather",
    "mpx_synth_ola_slots",
    "mpx_synth_ola_slot_weights",
    "mpx_ola_strip_floats",
    "mpx_synthesis_lossless_ola",
    "mpx_ola_fixup",
    "mpx_roundtrip_lossless_ola",
    "mpx_roundtrip_slot_weights",
    "mpx_mel_unwarp",
    "mpx_mel_unwarp_rows",
    "mpx_spec_ld",
    "mpx_noise_uniform",
    "mpx_noise_numpy_mt19937",
    "mpx_noise_numpy_mt19937_work_words",
    "mpx_host_mt19937_jump_poly",
    "mpx_host_mt19937_jump_polys",
    "mpx_noise_stats",
    "mpx_noise_spectra_floats",
    "mpx_noise_stats_spectra",
    "mpx_synth_comp_slots",
    "mpx_synth_comp_slot_weights",
    "mpx_synthesis_compressed_ola",
    "mpx_synthesis_compressed_ola_spectra",
    "mpx_host_const_to_var_scan",
    "mpx_host_plan_analysis",
    "mpx_host_plan_analysis_batch",
    "mpx_host_plan_synthesis",
    "mpx_host_plan_synthesis_batch",
    "mpx_host_plan_lossless_synthesis",
    "mpx_host_ola_runs",
    "mpx_host_widen_f32",
    "mpx_host_narrow_f64",
    "mpx_host_copy_many",
    "mpx_host_file_sizes",
    "mpx_host_read_est_batch",
    "mpx_host_write_files",
    "mpx_host_read_files",
    "mpx_mel_warp",
    "mpx_mel_warp_fbank",
    "mpx_mel_warp_rows",
    "mpx_warp_phase_rows",
    "mpx_min_phase",
    "mpx_noise_gains",
    "mpx_post_filter",
    "mpx_epoch_f0_track",
    "mpx_epoch_zff",
    "mpx_pcm16",
    "mpx_pcm16_to_f32",
    "mpx_hpf_block",
    "mpx_output_hpf",
    "mpx_bw_probe",
    "mpx_bw_probe_shapes",
    "mpx_post_filter_merlin",
)

_lib = None


class MagphaseHipError(RuntimeError):
    pass


_load_lock = threading.Lock()


def load():
    """Loads the shared library (once) and declares the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:   # the reader / writer threads of iobatch call the host helpers: one loader at a time
        return _load_locked()


def _load_locked():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch first: it brings the process's HIP runtime (its own libamdhip64).  Loading this library before it binds
    # the kernels to a second copy of the runtime from /opt/rocm, which torch never initialises ("no ROCm-capable
    # device is detected" at the first launch).
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise MagphaseHipError(
            "libmagphase_hip.so not found at %s -- build it with `python -m magphase_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t
    lib.mpx_version.restype = ctypes.c_int
    lib.mpx_version.argtypes = []
    lib.mpx_last_error.restype = ctypes.c_char_p
    lib.mpx_last_error.argtypes = []
    lib.mpx_tables_bytes.restype = sz
    lib.mpx_tables_bytes.argtypes = [ctypes.c_int]
    lib.mpx_tables_init.restype = ctypes.c_int
    lib.mpx_tables_init.argtypes = [vp, ctypes.c_int, vp]
    lib.mpx_analysis_frames.restype = ctypes.c_int
    lib.mpx_analysis_frames.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64]
    lib.mpx_tables_f64_bytes.restype = sz
    lib.mpx_tables_f64_bytes.argtypes = [ctypes.c_int]
    lib.mpx_tables_f64_init.restype = ctypes.c_int
    lib.mpx_tables_f64_init.argtypes = [vp, ctypes.c_int, vp]
    lib.mpx_analysis_frames_f64.restype = ctypes.c_int
    lib.mpx_analysis_frames_f64.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp]
    lib.mpx_analysis_compressed_fused.restype = ctypes.c_int
    lib.mpx_analysis_compressed_fused.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, i32, vp, vp, i32, i32, vp, i32,
                                                  vp, vp, vp]
    lib.mpx_analysis_compressed_fused_cr.restype = ctypes.c_int
    lib.mpx_analysis_compressed_fused_cr.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, i32, vp, vp, i32, i32, vp,
                                                     vp, vp, vp, i64, vp, vp, vp, vp]
    lib.mpx_analysis_compressed_fused_cr_work_bytes.restype = i64
    lib.mpx_analysis_compressed_fused_cr_work_bytes.argtypes = [ctypes.c_int, i64]
    lib.mpx_analysis_compressed_fused_blocks_per_cu.restype = ctypes.c_int
    lib.mpx_analysis_compressed_fused_blocks_per_cu.argtypes = [ctypes.c_int, i32]
    lib.mpx_analysis_compressed_fused_waves.restype = ctypes.c_int
    lib.mpx_analysis_compressed_fused_waves.argtypes = []
    lib.mpx_analysis_compressed_fused_layout.restype = ctypes.c_int
    lib.mpx_analysis_compressed_fused_layout.argtypes = []
    lib.mpx_analysis_compressed_fused_tiles.restype = ctypes.c_int
    lib.mpx_analysis_compressed_fused_tiles.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    lib.mpx_analysis_frames_f64w.restype = ctypes.c_int
    lib.mpx_analysis_frames_f64w.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, i32]
    lib.mpx_feat_ld.restype = i64
    lib.mpx_feat_ld.argtypes = [ctypes.c_int]
    lib.mpx_synthesis_lossless_frames.restype = ctypes.c_int
    lib.mpx_synthesis_lossless_frames.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, i64, vp, i64]
    lib.mpx_ola_gather.restype = ctypes.c_int
    lib.mpx_ola_gather.argtypes = [vp, ctypes.c_int, vp, i32, vp, vp, vp, vp, i64, vp]
    lib.mpx_synthesis_lossless_ola.restype = ctypes.c_int
    lib.mpx_synth_ola_slots.restype = ctypes.c_int
    lib.mpx_synth_ola_slots.argtypes = []
    lib.mpx_synth_comp_slot_weights.restype = ctypes.c_int
    lib.mpx_synth_comp_slot_weights.argtypes = [vp, i32]
    lib.mpx_synth_ola_slot_weights.restype = ctypes.c_int
    lib.mpx_synth_ola_slot_weights.argtypes = [vp, i32]
    lib.mpx_ola_strip_floats.restype = i64
    lib.mpx_ola_strip_floats.argtypes = [ctypes.c_int]
    lib.mpx_synthesis_lossless_ola.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i64]
    lib.mpx_roundtrip_slot_weights.restype = ctypes.c_int
    lib.mpx_roundtrip_slot_weights.argtypes = [vp, i32]
    lib.mpx_roundtrip_lossless_ola.restype = ctypes.c_int
    lib.mpx_roundtrip_lossless_ola.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, i32, vp, vp, i32, vp, vp, vp,
                                               vp, vp, vp, i64]
    lib.mpx_ola_fixup.restype = ctypes.c_int
    lib.mpx_ola_fixup.argtypes = [vp, ctypes.c_int, vp, i32, vp, vp]
    lib.mpx_mel_unwarp.restype = ctypes.c_int
    lib.mpx_mel_unwarp.argtypes = [vp, i64, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i64]
    lib.mpx_mel_unwarp_rows.restype = ctypes.c_int
    lib.mpx_mel_unwarp_rows.argtypes = [vp, i64, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, i64, vp, vp, vp, i64, vp, vp, i32]
    lib.mpx_spec_ld.restype = i64
    lib.mpx_spec_ld.argtypes = [i32]
    lib.mpx_noise_uniform.restype = ctypes.c_int
    lib.mpx_noise_uniform.argtypes = [vp, i32, vp, vp, i64, vp]
    lib.mpx_noise_numpy_mt19937.restype = ctypes.c_int
    lib.mpx_noise_numpy_mt19937.argtypes = [vp, vp, i32, i64, vp, vp, vp, vp, vp]
    lib.mpx_noise_numpy_mt19937_work_words.restype = i64
    lib.mpx_noise_numpy_mt19937_work_words.argtypes = []
    lib.mpx_host_mt19937_jump_poly.restype = ctypes.c_int
    lib.mpx_host_mt19937_jump_poly.argtypes = [i64, i32, vp]
    lib.mpx_noise_stats.restype = ctypes.c_int
    lib.mpx_noise_stats.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.mpx_noise_spectra_floats.restype = i64
    lib.mpx_noise_spectra_floats.argtypes = [ctypes.c_int, i64]
    lib.mpx_noise_stats_spectra.restype = ctypes.c_int
    lib.mpx_noise_stats_spectra.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    lib.mpx_synth_comp_slots.restype = ctypes.c_int
    lib.mpx_synth_comp_slots.argtypes = []
    lib.mpx_synthesis_compressed_ola.restype = ctypes.c_int
    lib.mpx_synthesis_compressed_ola.argtypes = [vp, ctypes.c_int, vp] + [vp] * 19 + [vp, i32, vp, vp, i32, vp, vp, i64, i32]
    lib.mpx_synthesis_compressed_ola_spectra.restype = ctypes.c_int
    lib.mpx_synthesis_compressed_ola_spectra.argtypes = ([vp, ctypes.c_int, vp] + [vp] * 19 +
                                                         [vp, i32, vp, vp, i32, vp, vp, i64, i32, vp])
    lib.mpx_host_const_to_var_scan.restype = i64
    lib.mpx_host_const_to_var_scan.argtypes = [vp, vp, i64, vp, vp]
    lib.mpx_host_plan_analysis.restype = i64
    lib.mpx_host_plan_analysis.argtypes = [i32] + [vp] * 12
    lib.mpx_host_plan_synthesis.restype = i64
    lib.mpx_host_plan_synthesis.argtypes = [i32, vp, vp, ctypes.c_double, i32, i32, i32, i64] + [vp] * 17
    lib.mpx_host_plan_lossless_synthesis.restype = i64
    lib.mpx_host_plan_lossless_synthesis.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, vp]
    lib.mpx_host_plan_analysis_batch.restype = i64
    lib.mpx_host_plan_analysis_batch.argtypes = [i32] + [vp] * 7 + [vp, i32] + [vp] * 9 + [i32, vp, vp, i64, vp, i32]
    lib.mpx_host_plan_synthesis_batch.restype = i64
    lib.mpx_host_plan_synthesis_batch.argtypes = ([i32] + [vp] * 5 + [i32, i32, vp, vp, ctypes.c_double, i32, i32, i32, i32,
                                                   vp, ctypes.c_double, i32, vp, i64] + [vp] * 9 + [i64, vp, i32])
    lib.mpx_host_ola_runs.restype = i64
    lib.mpx_host_ola_runs.argtypes = [i32, vp, vp, vp, vp, vp, i32, vp, i64, vp, i64]
    lib.mpx_host_narrow_f64.restype = i32
    lib.mpx_host_narrow_f64.argtypes = [vp, vp, i64, i32]
    lib.mpx_host_copy_many.restype = i32
    lib.mpx_host_copy_many.argtypes = [i32, vp, vp, vp, vp, i32]
    lib.mpx_host_widen_f32.restype = i32
    lib.mpx_host_widen_f32.argtypes = [vp, vp, i64, i32]
    lib.mpx_host_file_sizes.restype = i32
    lib.mpx_host_file_sizes.argtypes = [i32, vp, vp]
    lib.mpx_host_read_est_batch.restype = i32
    lib.mpx_host_read_est_batch.argtypes = [i32, vp, i32, vp, vp, vp, vp, i32]
    lib.mpx_host_write_files.restype = i32
    lib.mpx_host_write_files.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32]
    lib.mpx_host_read_files.restype = i32
    lib.mpx_host_read_files.argtypes = [i32, vp, vp, vp, vp, i32]
    lib.mpx_mel_warp.restype = ctypes.c_int
    lib.mpx_mel_warp.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, i64]
    lib.mpx_mel_warp_fbank.restype = ctypes.c_int
    lib.mpx_mel_warp_fbank.argtypes = lib.mpx_mel_warp.argtypes
    lib.mpx_mel_warp_rows.restype = ctypes.c_int
    lib.mpx_mel_warp_rows.argtypes = lib.mpx_mel_warp.argtypes + [i32, i64, vp, vp, vp]
    lib.mpx_warp_phase_rows.restype = ctypes.c_int
    lib.mpx_warp_phase_rows.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mpx_min_phase.restype = ctypes.c_int
    lib.mpx_min_phase.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64]
    lib.mpx_noise_gains.restype = ctypes.c_int
    lib.mpx_noise_gains.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    lib.mpx_post_filter.restype = ctypes.c_int
    lib.mpx_post_filter.argtypes = [vp, vp, i64, i32, vp, i32, i32, vp, vp]
    lib.mpx_epoch_f0_track.restype = ctypes.c_int
    lib.mpx_epoch_f0_track.argtypes = [vp, vp, vp, i32, i32, vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, ctypes.c_double,
                                       vp, vp, vp]
    lib.mpx_epoch_zff.restype = ctypes.c_int
    lib.mpx_epoch_zff.argtypes = [vp, vp, vp, i32, i64, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    lib.mpx_pcm16.restype = ctypes.c_int
    lib.mpx_pcm16.argtypes = [vp, vp, i32, vp, i32, i64, ctypes.c_double, vp, vp]
    lib.mpx_pcm16_to_f32.restype = ctypes.c_int
    lib.mpx_pcm16_to_f32.argtypes = [vp, vp, i64, vp]
    lib.mpx_hpf_block.restype = ctypes.c_int
    lib.mpx_hpf_block.argtypes = []
    lib.mpx_output_hpf.restype = ctypes.c_int
    lib.mpx_output_hpf.argtypes = [vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.mpx_post_filter_merlin.restype = ctypes.c_int
    lib.mpx_post_filter_merlin.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp, i32, ctypes.c_double, vp, ctypes.c_double,
                                           vp, vp, vp, vp, vp]
    lib.mpx_bw_probe_shapes.restype = ctypes.c_int
    lib.mpx_bw_probe_shapes.argtypes = []
    lib.mpx_bw_probe.restype = ctypes.c_int
    lib.mpx_bw_probe.argtypes = [vp, i32, vp, vp, i64]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mpx_last_error()
        raise MagphaseHipError("%s failed (%d): %s" % (what, rc, msg.decode("utf-8", "replace") if msg else ""))
