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
