"""Fixture / data I/O: the harness's save() variable lists round-trip through MAT -v7, the reference's own
golden files are read back by the same loader, and the HRIR loader takes the MIRO field names as plain arrays."""
import os

import numpy as np
import pytest
import scipy.io as sio

from emagls_amd import io as IO


def test_fixture_name_matches_the_harness_pattern():
    # verifyEMagLs.m:25-32 with filterLen 512, 32 microphones, order 4
    assert IO.fixture_name(512, 32, 4, "real", "eMagLS", dc=False) == "HRIR_L2702_512samples_32channels_sh4_real_eMagLS_woDC.mat"
    assert IO.fixture_name(512, 32, 4, "complex", "LS") == "HRIR_L2702_512samples_32channels_sh4_complex_LS.mat"
    assert IO.fixture_name(512, 32, 4, "real", "MagLS", dc=True).endswith("_real_MagLS_wDC.mat")
    with pytest.raises(ValueError):
        IO.fixture_name(512, 32, 4, "real", "eMagLS3")


@pytest.mark.parametrize("cplx", [False, True])
def test_save_fixture_round_trip(tmp_path, golden, grids, cplx):
    """Write the eMagLS set the way verifyEMagLs.m:218-221 does and read it back: same variables, values, dtypes, shapes
    (grids as column vectors, scalars 1x1) -- and byte-compatible with how the reference's own fixture loads."""
    tag = ("complex" if cplx else "real") + "_eMagLS_woDC"
    wL, wR = golden[tag + "/wEMlsL"], golden[tag + "/wEMlsR"]
    assert np.iscomplexobj(wL) == cplx
    f = str(tmp_path / IO.fixture_name(512, 32, 4, "complex" if cplx else "real", "eMagLS", dc=False))
    IO.save_fixture(f, "eMagLS", wEMlsL=wL, wEMlsR=wR, hrirGridAziRad=grids["azi"], hrirGridZenRad=grids["zen"],
                    micRadius=grids["mic_radius"], micGridAziRad=grids["mic_azi"], micGridZenRad=grids["mic_zen"], shOrder=4,
                    fs=48000.0, filterLen=512)
    raw = sio.loadmat(f)
    assert sorted(k for k in raw if not k.startswith("__")) == sorted(IO.FIXTURE_FIELDS["eMagLS"])
    assert raw["hrirGridAziRad"].shape == (2702, 1) and raw["shOrder"].shape == (1, 1) and raw["wEMlsL"].shape == (512, 25)
    assert raw["wEMlsL"].dtype == (np.complex128 if cplx else np.float64)
    with open(f, "rb") as fh:
        assert fh.read(19) == b"MATLAB 5.0 MAT-file"          # '-v7' is the level-5 container (with compression)
    d = IO.load_fixture(f)
    assert np.array_equal(d["wEMlsL"], wL) and np.array_equal(d["wEMlsR"], wR)
    assert np.array_equal(d["hrirGridAziRad"], grids["azi"]) and d["micRadius"] == grids["mic_radius"] and d["filterLen"] == 512.0


def test_save_fixture_checks_the_variable_list(tmp_path):
    with pytest.raises(ValueError, match="missing"):
        IO.save_fixture(str(tmp_path / "x.mat"), "LS", wLsL=np.zeros((4, 4)), wLsR=np.zeros((4, 4)))
    with pytest.raises(ValueError, match="unexpected"):
        IO.save_fixture(str(tmp_path / "x.mat"), "LS", wLsL=np.zeros((4, 4)), wLsR=np.zeros((4, 4)), hrirGridAziRad=[0.0],
                        hrirGridZenRad=[0.0], shOrder=1, fs=48000.0)


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources"), reason="reference checkout not present (GPU box)")
def test_load_fixture_reads_the_reference_files(golden):
    d = IO.load_fixture("/root/reference/resources/HRIR_L2702_512samples_32channels_sh4_complex_MagLS_woDC.mat")
    assert set(d) == set(IO.FIXTURE_FIELDS["MagLS"]) | {"applyDiffusenessConst"} and d["applyDiffusenessConst"] == 0.0
    assert np.array_equal(d["wMlsL"], golden["complex_MagLS_woDC/wMlsL"]) and d["fs"] == 48000.0 and d["shOrder"] == 4.0
    assert d["hrirGridAziRad"].shape == (2702,)


def _hrir_arrays():
    rng = np.random.default_rng(0)
    return rng.standard_normal((16, 6)), rng.standard_normal((16, 6)), np.linspace(0, 5, 6), np.linspace(0.1, 3, 6)


def test_load_hrir_set_containers(tmp_path):
    hL, hR, azi, zen = _hrir_arrays()
    # the MIRO field names at top level (save -struct) and inside a struct variable, as single precision like the original
    sio.savemat(str(tmp_path / "plain.mat"), dict(irChOne=hL.astype(np.float32), irChTwo=hR.astype(np.float32), azimuth=azi[None, :],
                                                  elevation=zen[None, :], fs=np.array([[48000]])))
    sio.savemat(str(tmp_path / "struct.mat"), dict(HRIR_L2702=dict(irChOne=hL, irChTwo=hR, azimuth=azi, elevation=zen, fs=48000.0)))
    np.savez(str(tmp_path / "a.npz"), hL=hL, hR=hR)
    np.savez(str(tmp_path / "b.npz"), irChOne=hL, irChTwo=hR, azimuth=azi, elevation=zen, fs=48000)
    for name, exact, grid in (("plain.mat", False, True), ("struct.mat", True, True), ("a.npz", True, False), ("b.npz", True, True)):
        d = IO.load_hrir_set(str(tmp_path / name))
        assert d["hL"].dtype == np.float64 and d["hL"].shape == (16, 6)
        assert np.allclose(d["hL"], hL, atol=0 if exact else 1e-6) and np.allclose(d["hR"], hR, atol=0 if exact else 1e-6)
        if grid:
            assert np.array_equal(d["azi"], azi) and np.array_equal(d["zen"], zen) and d["fs"] == 48000.0
        else:
            assert "azi" not in d


def test_load_hrir_set_errors(tmp_path):
    hL, hR, azi, zen = _hrir_arrays()
    sio.savemat(str(tmp_path / "bad.mat"), dict(something=hL))
    with pytest.raises(ValueError, match="neither"):
        IO.load_hrir_set(str(tmp_path / "bad.mat"))
    sio.savemat(str(tmp_path / "obj.mat"), dict(HRIR_L2702=np.zeros((6, 1), dtype=np.uint32)))   # what a missing class leaves (:57)
    with pytest.raises(ValueError, match="MATLAB object"):
        IO.load_hrir_set(str(tmp_path / "obj.mat"))
    sio.savemat(str(tmp_path / "grid.mat"), dict(irChOne=hL, irChTwo=hR, azimuth=azi[:3], elevation=zen, fs=1.0))
    with pytest.raises(ValueError, match="grid angles"):
        IO.load_hrir_set(str(tmp_path / "grid.mat"))
    with pytest.raises(ValueError, match="unsupported"):
        IO.load_hrir_set(str(tmp_path / "x.sofa"))
