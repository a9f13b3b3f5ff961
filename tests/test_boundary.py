"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/gmpi_render.h declares, the ctypes struct matches, the product never touches the oracle,
and the product refuses to run without a GPU / without the HIP library (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ml-gmpi_amd")


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "gmpi_render.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gmpi_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from ml_gmpi_amd import _lib
    names = declared_functions()
    assert len(names) >= 6 and set(names) == set(_lib.EXPORTS), names
    lib = _lib.load_library()
    for n in names:
        assert hasattr(lib, n), n
    assert lib.gmpi_query(0) == _lib.ABI_VERSION
    assert lib.gmpi_query(1) == ctypes.sizeof(_lib.GmpiRenderParams)
    assert lib.gmpi_query(2) == 950
    assert b"gfx950" in lib.gmpi_version_string()


def test_header_compiles_as_plain_c(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "gmpi_render.h"\nint main(void){GmpiRenderParams p; p.struct_size=sizeof p; return (int)p.struct_size==0;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)


def test_product_never_references_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+(oracle|ref_import)\b", txt, flags=re.M) or "libgmpi_oracle" in txt \
                        or "/root/reference" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_no_cpu_fallback():
    from ml_gmpi_amd import MPI, GmpiError
    mpi = MPI()
    rgba = torch.rand(1, 2, 4, 8, 8)
    with pytest.raises(GmpiError, match="no CPU path"):
        with torch.no_grad():
            mpi.render_views(rgba, torch.rand(1, 2, 3), torch.rand(1, 3, 8, 8), torch.rand(1, 3), torch.rand(1, 3))


def test_missing_library_fails_loudly(monkeypatch):
    from ml_gmpi_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "_SO", os.path.join(PKG, "does_not_exist.so"))
    with pytest.raises(_lib.GmpiError, match="HIP extension not built"):
        _lib.load_library()
