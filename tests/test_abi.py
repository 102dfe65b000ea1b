"""The C-ABI shared library builds for gfx950, loads, and exports every symbol include/lvae.h declares with the
signature table the ctypes binding uses.  No compute calls (no GPU needed)."""
import ctypes
import os
import re
import sys

from vae_lagging_encoder_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    with open(os.path.join(ROOT, "include", "lvae.h")) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(lv_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = header_symbols()
    assert syms, "no declarations parsed from include/lvae.h"
    assert sorted(_lib.SIGNATURES.keys()) == syms


def test_hip_library_builds_loads_and_exports_everything():
    path = build.build_hip()
    assert os.path.exists(path)
    cdll = ctypes.CDLL(path)
    for name in header_symbols():
        assert hasattr(cdll, name), name
    lib = _lib.bind(cdll, path)
    # value-returning helpers are host-only and safe to call without a GPU
    assert lib.lv_lstm_bwd_ksplit(1024) == 4
    assert lib.lv_sumsq_workspace_floats() >= 256


def test_argument_checks_return_negative_status_without_touching_the_gpu():
    lib = _lib.bind(ctypes.CDLL(build.build_hip()))
    raw = lib.cdll.lv_gemm_f32
    assert raw(0, 1, -1, 4, 4, 1.0, None, 4, None, 4, None, 4, 0, None, 0, 1, None, 0, 1, None, 0, None) < 0
    assert raw(0, 1, 4, 4, 4, 1.0, None, 4, None, 4, None, 4, 0, None, 0, 1, None, 0, 1, None, 0, None) < 0   # NULL operands
    assert lib.cdll.lv_lstm_fwd_f32(None, None, None, None, None, None, 1.0, None, None, 1, 1, 1, None) < 0
    assert lib.lv_lstm_ws_floats(32, 1024) > 4 * 1024 * 1024


def test_package_has_no_cpu_fallback():
    import torch
    from vae_lagging_encoder_amd import engine
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import install as emu_install
    saved = emu_install.install(None)       # the session fixture may have installed the emulator
    try:
        try:
            engine.backend_for(torch.device("cpu"))
            raise AssertionError("CPU tensors must be refused")
        except _lib.LvaeError:
            pass
        # and the package itself carries no switch for it
        assert not any("TEST_BACKEND" in n.upper() or "install_test" in n for n in dir(engine))
    finally:
        emu_install.install(saved)
