"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/rvb.h
declares; without a GPU the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import HAVE_GPU, ROOT
from reverb_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rvb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rvb_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()                                   # the PRODUCT library
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"librvb.so does not export {n}"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in reverb_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_product_library_carries_no_test_hooks(lib):
    """the rvb_test_* hooks live in librvb_test.so (csrc/test_api.h); the product .so exports the two
    public headers and nothing else of the C ABI."""
    import subprocess
    assert not any(n.startswith("rvb_test_") for n in declared_symbols())
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if nm.returncode == 0:
        exported = {l.split()[-1] for l in nm.stdout.splitlines() if l.strip()}
        assert not [n for n in exported if n.startswith(("rvb_test_", "rvd_test_"))]
        c_abi = {n for n in exported if re.fullmatch(r"rv[bd]_[a-z0-9_]+", n)}
        assert c_abi == set(_lib.SIGNATURES) | set(_lib.DIAR_SIGNATURES), c_abi ^ (set(_lib.SIGNATURES) | set(_lib.DIAR_SIGNATURES))
    product = _lib.load()
    text = open(os.path.join(ROOT, "reverb_amd", "csrc", "test_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    hooks = sorted(set(re.findall(r"\b(rvb_test_[a-z0-9_]+)\s*\(", text)))
    assert hooks == sorted(_lib.TEST_SIGNATURES) and len(hooks) >= 20
    for n in hooks:
        assert hasattr(lib, n), f"librvb_test.so does not export {n}"
        assert not hasattr(product, n), f"librvb.so exports the test hook {n}"


def test_version_and_frame_count(lib):
    assert b"gfx950" in lib.rvb_version()
    assert lib.rvb_num_frames(399) == 0 and lib.rvb_num_frames(400) == 1 and lib.rvb_num_frames(57600000) == 359998


def test_struct_layout_matches_header():
    text = open(os.path.join(ROOT, "include", "rvb.h")).read()
    body = text[text.index("typedef struct rvb_model_cfg {"):text.index("} rvb_model_cfg;")]
    fields = re.findall(r"int32_t\s+([a-z_0-9]+);", body)
    assert fields == [f[0] for f in _lib.ModelCfg._fields_]
    assert fields[0] == "struct_size" and _lib.ModelCfg().struct_size == 4 * len(fields)


def _integration_stub_fields(struct_comment):
    """Field names of the ctypes stub INTEGRATION.md shows for `struct_comment` (the list a maintainer would paste)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    at = text.index("# struct " + struct_comment)
    body = text[at:text.index(")]", at)]
    return re.findall(r'"([a-z_0-9]+)"', body)


def test_integration_stub_binds_the_header_field_for_field():
    """the document's stub had 19 fields against the header's 20.  The stub is parsed out of the
    document and compared with the headers, for both config structs and the audio info struct."""
    for header, struct, comment in (("rvb.h", "rvb_model_cfg", "rvb_model_cfg"), ("rvd.h", "rvd_model_cfg", "rvd_model_cfg")):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        body = text[text.index("typedef struct %s {" % struct):text.index("} %s;" % struct)]
        assert _integration_stub_fields(comment) == re.findall(r"int32_t\s+([a-z_0-9]+);", body), header
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rvb.h")).read(), flags=re.S)
    body = text[text.index("typedef struct rvb_audio_info {"):text.index("} rvb_audio_info;")]
    assert _integration_stub_fields("rvb_audio_info") == re.findall(r"int(?:32|64)_t\s+([a-z_0-9]+);", body)


def test_a_binding_of_another_size_is_refused_by_name(lib):
    """A struct one field short (what following the stale document produced) or long must not be read: RVB_E_ARG."""
    assert lib.rvb_model_cfg_size() == ctypes.sizeof(_lib.ModelCfg) and lib.rvd_model_cfg_size() == ctypes.sizeof(_lib.DiarCfg)
    h = ctypes.c_void_p()
    for size in (ctypes.sizeof(_lib.ModelCfg) - 4, ctypes.sizeof(_lib.ModelCfg) + 4, 0):
        cfg = _lib.ModelCfg(struct_size=size)
        assert lib.rvb_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1 and b"ABI mismatch" in lib.rvb_last_error()
    dcfg = _lib.DiarCfg(struct_size=ctypes.sizeof(_lib.DiarCfg) - 4)
    assert lib.rvd_create(ctypes.byref(dcfg), 0, ctypes.byref(h)) == -1 and b"ABI mismatch" in lib.rvd_last_error()


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu(lib):
    from reverb_amd import synth
    from reverb_amd.engine import Engine
    cfg = synth.make_config("tiny")
    with pytest.raises(_lib.RvbError, match="no HIP device"):
        Engine(cfg, {}, dtype="f32")
    out = np.zeros(4, np.float32)
    rc = lib.rvb_test_gemm(0, _lib.fptr(out), _lib.fptr(out), None, None, _lib.fptr(out), 1, 1, 4, 1.0, 0, 1, 0, 0, 0, 0, 0)
    assert rc != 0 and b"no CPU fallback" in lib.rvb_last_error()


def test_bad_arguments_are_reported_not_crashed(lib):
    cfg = _lib.ModelCfg()
    h = ctypes.c_void_p()
    assert lib.rvb_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1      # RVB_E_ARG: all-zero dims
    assert b"unsupported model dimensions" in lib.rvb_last_error() or b"dtype" in lib.rvb_last_error()
    assert lib.rvb_create(None, 0, ctypes.byref(h)) == -1


def test_diarization_header_symbols_and_struct():
    """include/rvd.h (diarization networks): every declared function is exported and bound."""
    from reverb_amd import _lib as L
    lib = L.load()
    text = open(os.path.join(ROOT, "include", "rvd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(rvd_[a-z0-9_]+)\s*\(", code)))
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"librvb.so does not export {n}"
    assert sorted(L.DIAR_SIGNATURES) == names
    body = code[code.index("typedef struct rvd_model_cfg {"):code.index("} rvd_model_cfg;")]
    assert re.findall(r"int32_t\s+([a-z_0-9]+);", body) == [f[0] for f in L.DiarCfg._fields_]


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_diarization_fails_loudly_without_gpu():
    from reverb_amd import synth_diar
    from reverb_amd.diar_engine import DiarEngine
    with pytest.raises(_lib.RvbError, match="no HIP device"):
        DiarEngine(synth_diar.make_diar_config(), {}, dtype="f32")


def test_compute_feats_refuses_what_the_kernel_does_not_cover():
    """host-side argument checks of rvb_compute_feats (no GPU needed for the frame count or the refusals)"""
    lib = _lib.load()
    n = ctypes.c_int64(-1)
    x = np.zeros(16000, np.float32)
    assert lib.rvb_compute_feats(0, _lib.fptr(x), x.size, 23, 25.0, 10.0, None, ctypes.byref(n)) == 0 and n.value == 98
    assert lib.rvb_compute_feats(0, _lib.fptr(x), x.size, 80, 32.0, 8.0, None, ctypes.byref(n)) == 0 and n.value == 1 + (16000 - 512) // 128
    for bins, flen, fshift in ((80, 10.0, 10.0), (80, 40.0, 10.0), (200, 25.0, 10.0), (80, 25.0, 0.0)):
        assert lib.rvb_compute_feats(0, _lib.fptr(x), x.size, bins, flen, fshift, None, ctypes.byref(n)) == -5      # RVB_E_UNSUPPORTED
        assert b"rvb_compute_feats" in lib.rvb_last_error()
