"""CPU: the C-ABI library builds, loads, and exports every symbol include/magphase_hip.h declares."""
import ctypes
import os
import re

import pytest

from magphase_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "magphase_hip.h")).read()
    return sorted(set(re.findall(r"\b(mpx_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_loads_and_exports_all_symbols():
    if not os.path.isfile(_lib.LIB_PATH):
        build.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    l2 = _lib.load()
    assert l2.mpx_version() == 2
    assert l2.mpx_tables_bytes(4096) == 64 * (2 * 32 + 4) * 4   # one padded row per lane (wave_fft.hpp)
    assert l2.mpx_tables_bytes(2048) == 64 * (2 * 16 + 4) * 4
    assert l2.mpx_tables_bytes(1024) == 64 * (2 * 8 + 4) * 4
    assert l2.mpx_tables_bytes(1000) == 0


def test_argument_errors_without_gpu():
    lib = _lib.load()
    rc = lib.mpx_analysis_frames(None, 1234, None, None, None, None, None, 1, None, None, None, 618)
    assert rc == -1 and b"fft_len" in lib.mpx_last_error()
    rc = lib.mpx_analysis_frames(None, 4096, None, None, None, None, None, 5, None, None, None, 2049)
    assert rc == -1 and b"null" in lib.mpx_last_error()
    assert lib.mpx_analysis_frames(None, 4096, None, None, None, None, None, 0, None, None, None, 2112) == 0


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from magphase_amd import magphase as mp
    with pytest.raises(_lib.MagphaseHipError):
        mp.synthesis_from_lossless(None, None, None, None, 48000)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "magphase_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_every_exported_symbol_is_mapped_to_the_reference_in_integration_md():
    """INTEGRATION.md section 3 is the map reference interface -> C entry point: a symbol without a row is an
    undocumented piece of the boundary."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in _declared() if s not in doc]
    assert not missing, missing
