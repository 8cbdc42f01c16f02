"""The C-ABI library loads without a GPU, exports every symbol include/frbch.h declares, and
fails loudly (no CPU fallback) when asked to compute without a device."""
import ctypes as C
import os
import re
import subprocess

import pytest

from frb_baseband_amd import _lib
from frb_baseband_amd import channeliser as ch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "frbch.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frbch_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (frbch_[a-z0-9_]+)", out))
    assert set(declared_functions()) <= exported


def test_struct_sizes_match(hip_lib):
    cfg = ch.new_config(hip_lib)
    assert cfg.size == C.sizeof(_lib.FrbchConfig)       # frbch_config_init writes sizeof(frbch_config)
    assert cfg.abi_version == _lib.ABI_VERSION


def test_version_is_hip_backend(hip_lib):
    assert b"hip-gfx950" in hip_lib.frbch_version()


def test_shim_exists_and_rejects_bad_args():
    shim = os.path.join(ROOT, "frb_baseband_amd", "csrc", "digifil")
    assert os.access(shim, os.X_OK)
    r = subprocess.run([shim, "-q", "-o", "x.fil", "y.hdr"], capture_output=True)
    assert r.returncode != 0 and b"unknown option" in r.stderr


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ch.RunError) as ei:
        ch.Channeliser(ch.new_config())
    assert "no CPU fallback" in ei.value.message


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(_lib.LibraryMissing):
        _lib.load(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "frb_baseband_amd")
    for dirpath, _d, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".h", ".inc", ".hip")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "frb_oracle" not in text and "import oracle" not in text and "from oracle" not in text, fn
                assert "libfrbch_emu" not in text, fn
                assert not re.search(r'#\s*include\s*[<"][^>"]*emu', text), fn
