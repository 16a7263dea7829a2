"""Fixture and data I/O of the reference's harness -- host-side file handling, no device work.

    load_hrir_set     verifyEMagLs.m:54-71   load(hrirFile); hL = HRIR_L2702.irChOne; ... (the fields the harness reads)
    load_fixture      verifyEMagLs.m:152-200 the golden filter sets under resources/ (MAT v5/v7)
    save_fixture      verifyEMagLs.m:203-227 save(refFile, <variables>, '-v7') with the harness's variable lists
    fixture_name      verifyEMagLs.m:25-32   resources/HRIR_L2702_<len>samples_<mics>channels_sh<order>_<shDef>_<method>.mat

The published HRIR set is a MIRO *object* (a MATLAB class instance; MATLAB stores those in an undocumented subsystem blob that
only MATLAB with miro.m on its path can decode -- the harness itself downloads miro.m for that, :59-66).  What crosses to this
side is the set as plain arrays: an .npz, or a .mat holding either the five fields or a struct with them, e.g. from

    load HRIR_L2702.mat; s = struct('irChOne', HRIR_L2702.irChOne, 'irChTwo', HRIR_L2702.irChTwo, 'azimuth', HRIR_L2702.azimuth, ...
        'elevation', HRIR_L2702.elevation, 'fs', HRIR_L2702.fs); save('hrir_l2702_plain.mat', '-struct', 's', '-v7')
"""
from __future__ import annotations

import os

import numpy as np

# the variable lists of the harness's save() calls (verifyEMagLs.m:206-224), keyed by the method part of the file name
FIXTURE_FIELDS = {
    "LS": ("wLsL", "wLsR", "hrirGridAziRad", "hrirGridZenRad", "shOrder"),
    "MagLS": ("wMlsL", "wMlsR", "hrirGridAziRad", "hrirGridZenRad", "shOrder", "fs", "filterLen"),
    "eMagLS": ("wEMlsL", "wEMlsR", "hrirGridAziRad", "hrirGridZenRad", "micRadius", "micGridAziRad", "micGridZenRad", "shOrder", "fs",
               "filterLen"),
    "eMagLS2": ("wEMls2L", "wEMls2R", "hrirGridAziRad", "hrirGridZenRad", "micRadius", "micGridAziRad", "micGridZenRad", "fs", "filterLen"),
}
# the shipped fixtures predate the removal of the diffuseness constraint and also carry this flag (CHANGELOG.md:10-12)
OPTIONAL_FIELDS = ("applyDiffusenessConst",)
_ROW_VECTORS = ("hrirGridAziRad", "hrirGridZenRad", "micGridAziRad", "micGridZenRad")   # column vectors in the harness's workspace
_HRIR_FIELDS = ("irChOne", "irChTwo", "azimuth", "elevation", "fs")


def fixture_name(filter_len, num_mics, sh_order, sh_definition, method, dc=None, hrir="HRIR_L2702"):
    """verifyEMagLs.m:25-32: '<hrir>_<len>samples_<mics>channels_sh<order>_<real|complex>_<LS|MagLS_woDC|...>.mat'.
    dc: None for LS (no suffix), False -> '_woDC', True -> '_wDC'."""
    if method not in FIXTURE_FIELDS:
        raise ValueError("method must be one of %s" % (sorted(FIXTURE_FIELDS),))
    suffix = "" if dc is None else ("_wDC" if dc else "_woDC")
    return "%s_%dsamples_%dchannels_sh%d_%s_%s%s.mat" % (hrir, filter_len, num_mics, sh_order, sh_definition, method, suffix)


def _loadmat(path):
    import scipy.io as sio
    try:
        return sio.loadmat(path, squeeze_me=False, struct_as_record=False)
    except NotImplementedError as e:       # MAT v7.3 is HDF5
        raise ValueError("%s is a MAT v7.3 (HDF5) file; re-save it with '-v7'" % path) from e


def load_fixture(path):
    """One golden filter set as {variable: array}; scalars come back as Python floats, grids as 1-D arrays."""
    out = {}
    for k, v in _loadmat(path).items():
        if k.startswith("__"):
            continue
        v = np.asarray(v)
        if v.size == 1:
            out[k] = float(np.real(v.ravel()[0]))
        elif k in _ROW_VECTORS:
            out[k] = np.asarray(v, dtype=np.float64).ravel()
        else:
            out[k] = v
    return out


def save_fixture(path, method, **variables):
    """save(refFile, ..., '-v7') with exactly the variables the harness writes for `method` (LS / MagLS / eMagLS / eMagLS2).
    Filters keep their dtype (complex filters stay complex); grids are written as column vectors, scalars as 1x1 doubles."""
    import scipy.io as sio
    want = FIXTURE_FIELDS[method]
    missing = [k for k in want if k not in variables]
    extra = [k for k in variables if k not in want and k not in OPTIONAL_FIELDS]
    want = tuple(want) + tuple(k for k in OPTIONAL_FIELDS if k in variables)
    if missing or extra:
        raise ValueError("fixture '%s' takes exactly %s (missing %s, unexpected %s)" % (method, want, missing, extra))
    md = {}
    for k in want:
        v = np.asarray(variables[k])
        if k in _ROW_VECTORS:
            v = np.asarray(v, dtype=np.float64).reshape(-1, 1)
        elif v.ndim == 0:
            v = np.asarray(v, dtype=np.float64).reshape(1, 1)
        elif not np.iscomplexobj(v):
            v = np.asarray(v, dtype=np.float64)
        md[k] = v
    sio.savemat(path, md, format="5", do_compression=True, oned_as="column")     # '-v7' == level-5 format with compression


def _field(obj, name):
    if isinstance(obj, dict):
        return obj[name]
    return getattr(obj, name)


def load_hrir_set(path):
    """-> dict(hL, hR [numSamples x numDirections], azi, zen [numDirections], fs) as verifyEMagLs.m:67-71 derives them
    (`elevation` holds zenith angles, :70).  Accepts .npz (hL, hR[, azi, zen, fs] or the MIRO field names) and .mat with
    the five MIRO fields at top level or inside one struct variable (any name, e.g. HRIR_L2702)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        d = dict(np.load(path))
        if "hL" in d:
            out = dict(hL=d["hL"], hR=d["hR"])
            for k in ("azi", "zen", "fs"):
                if k in d:
                    out[k] = d[k]
            return _finish_hrirs(out, path)
        src = d
    elif ext == ".mat":
        d = {k: v for k, v in _loadmat(path).items() if not k.startswith("__")}
        if all(f in d for f in _HRIR_FIELDS[:2]):
            src = d
        else:
            src = None
            for k, v in d.items():
                v = np.asarray(v)
                if v.dtype == object and v.size == 1 and hasattr(v.ravel()[0], "_fieldnames"):
                    if all(f in v.ravel()[0]._fieldnames for f in _HRIR_FIELDS[:2]):
                        src = v.ravel()[0]
                        break
                if v.dtype.kind in "uV" and k.upper().startswith("HRIR"):    # what loadmat leaves of a class instance
                    raise ValueError("%s holds '%s' as a MATLAB object (MIRO class); export its fields as plain arrays first "
                                     "(see the module docstring of emagls_amd.io)" % (path, k))
            if src is None:
                raise ValueError("%s has neither irChOne/irChTwo arrays nor a struct with them" % path)
    else:
        raise ValueError("unsupported HRIR container '%s' (use .npz or .mat -v7)" % ext)
    out = dict(hL=_field(src, "irChOne"), hR=_field(src, "irChTwo"))
    for name, key in (("azimuth", "azi"), ("elevation", "zen"), ("fs", "fs")):
        try:
            out[key] = _field(src, name)
        except (KeyError, AttributeError):
            pass
    return _finish_hrirs(out, path)


def _finish_hrirs(out, path):
    hL = np.asarray(out["hL"], dtype=np.float64)
    hR = np.asarray(out["hR"], dtype=np.float64)
    if hL.ndim != 2 or hL.shape != hR.shape:
        raise ValueError("%s: irChOne / irChTwo must be equal-shaped [numSamples x numDirections] arrays" % path)
    res = dict(hL=hL, hR=hR)
    for k in ("azi", "zen"):
        if k in out:
            v = np.asarray(out[k], dtype=np.float64).ravel()
            if v.size != hL.shape[1]:
                raise ValueError("%s: %d grid angles for %d directions" % (path, v.size, hL.shape[1]))
            res[k] = v
    if "fs" in out:
        res["fs"] = float(np.asarray(out["fs"]).ravel()[0])
    return res
