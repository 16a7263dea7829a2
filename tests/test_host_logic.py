"""Host-side checks that need neither the GPU nor the library."""



def test_wave_fft_lds_layouts_are_conflict_free():
    """wave_fft.hpp's three LDS layouts against the lane groups of ds_write_b128 (8 contiguous lanes, 8 slots) and ds_read_b128 (four
    non-contiguous 16-lane groups, 16 slots) -- MI355X_MICROARCH.md, LDS table: the transpositions conflict-free, the mirrored reads of
    the natural-order layout at most 2-way (the simulation the header cites)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "experiments", "lds_layout_check.py")
    spec = importlib.util.spec_from_file_location("lds_layout_check", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    r = m.report()
    assert sorted(m.slot(k) for k in range(1024)) == list(range(1024))
    assert all(r[k] == 1 for k in ("transposition 1 write", "transposition 1 read", "transposition 2 write", "transposition 2 read", "natural-order write")), r
    assert r["natural-order read k"] <= 2 and r["natural-order read N-k"] <= 2, r
