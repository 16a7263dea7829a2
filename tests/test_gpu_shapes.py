"""Ragged, minimal and maximal shapes through every entry point, against the oracle (tolerance 1e-6 relative, north_star).
The full sweep (60 cases, incl. 3000+ directions and 512-tap filters): `python tools/fuzz_shapes.py` on the GPU box."""
import pytest

from shape_cases import CASES, FAST, run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx", FAST, ids=[f"{i}-{CASES[i][0]}-D{CASES[i][1]}" for i in FAST])
def test_shape_case(idx):
    err = run(CASES[idx])
    print(f"case {CASES[idx]}: rel = {err:.2e}")
    assert err < 1e-6


def test_too_few_directions_for_the_orthonormal_route_are_refused():
    """165 directions against the 256+ SH channels the low bins of a 3.6 cm array need on the orthonormal route: a message, not a
    failed Cholesky factorisation."""
    from emagls_amd._lib import EmaglsError
    with pytest.raises(EmaglsError, match="orthonormal route"):
        run(("emagls2", 165, 64, 224, 48000.0, 0.03608547132175317, 30, 4, "complex"))
