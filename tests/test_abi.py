"""CPU-side checks of the boundary: the C-ABI library builds, loads and exports every symbol that
include/emagls.h declares; without a GPU every compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from emagls_amd import build, _lib
    build.build(jobs=4, verbose=False)
    return _lib.load()


def test_header_and_binding_agree(lib):
    from emagls_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "emagls.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(emagls_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors have the size and field offsets the C compiler gives the header's structs."""
    import subprocess
    from emagls_amd import _lib
    src = tmp_path / "layout.c"
    fields_d = [f[0] for f in _lib.DesignDesc._fields_]
    fields_i = [f[0] for f in _lib.PlanInfo._fields_]
    body = "".join('printf("%%zu\\n", offsetof(emagls_design_desc, %s));' % f for f in fields_d)
    body += "".join('printf("%%zu\\n", offsetof(emagls_plan_info, %s));' % f for f in fields_i)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "emagls.h"\nint main(void){printf("%zu %zu\\n", '
                   'sizeof(emagls_design_desc), sizeof(emagls_plan_info));' + body + 'return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert [int(out[0]), int(out[1])] == [C.sizeof(_lib.DesignDesc), C.sizeof(_lib.PlanInfo)]
    offs = [int(x) for x in out[2:]]
    want = [getattr(_lib.DesignDesc, f).offset for f in fields_d] + [getattr(_lib.PlanInfo, f).offset for f in fields_i]
    assert offs == want


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import emagls_amd as E
    from emagls_amd._lib import EmaglsError
    with pytest.raises(EmaglsError):
        E.getSH(2, np.zeros((4, 2)), "real")
    with pytest.raises(EmaglsError):
        E.getLsFilters(np.zeros((8, 30)), np.zeros((8, 30)), np.zeros(30), np.zeros(30), 1)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under emagls_amd/ may reference it."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "emagls_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "emagls_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_design_out_shape_needs_no_gpu(lib):
    """emagls_design_out_shape (include/emagls.h): the filter shape of every design kind from the descriptor alone -- what a C or MEX
    caller allocates for a job of emagls_jobs_run (lib/getLsFilters.m:33 keeps the HRIR length; lib/getEMagLs2Filters.m:132-135 one
    filter per microphone; getMagLsFilters2D / EMAinCH 2N+1 circular harmonics; FromAtf real whatever the basis)."""
    import ctypes as C
    from emagls_amd import _lib as L
    cases = [  # kind, basis, order, len, nsamp, nmics -> rows, cols, complex
        (L.KIND_LS, "complex", 3, 512, 128, 0, (128, 16, True)), (L.KIND_MAGLS, "real", 4, 512, 128, 0, (512, 25, False)),
        (L.KIND_MAGLS_2D, "complex", 7, 256, 64, 0, (256, 15, True)), (L.KIND_EMAGLS, "complex", 4, 512, 128, 32, (512, 25, True)),
        (L.KIND_EMAGLS2, "complex", 4, 1024, 128, 32, (1024, 32, True)), (L.KIND_EMAGLS2, "real", 1, 1024, 128, 64, (1024, 64, False)),
        (L.KIND_EMA_CH, "real", 5, 512, 128, 16, (512, 11, False)), (L.KIND_EMA_SH, "real", 3, 512, 128, 16, (512, 16, False)),
        (L.KIND_FROM_ATF, "complex", 0, 2048, 256, 8, (2048, 8, False))]
    for kind, basis, order, ln, nsamp, nmics, want in cases:
        d = L.DesignDesc(kind, L.BASIS[basis], order, 48000.0, ln, nsamp, 2702, 0.042, nmics, 0.0, 0, 0, 0, 0, 0)
        r, c, z = C.c_int64(0), C.c_int64(0), C.c_int(0)
        assert lib.emagls_design_out_shape(C.byref(d), C.byref(r), C.byref(c), C.byref(z)) == 0
        assert (r.value, c.value, bool(z.value)) == want, (kind, basis)
    bad = L.DesignDesc(99, 0, 1, 48000.0, 16, 16, 10, 0.0, 0, 0.0, 0, 0, 0, 0, 0)
    r, c, z = C.c_int64(0), C.c_int64(0), C.c_int(0)
    assert lib.emagls_design_out_shape(C.byref(bad), C.byref(r), C.byref(c), C.byref(z)) != 0
