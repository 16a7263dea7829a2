"""Fixture and data I/O of the reference's harness -- host-side file handling, no device work.

    load_hrir_set     verifyEMagLs.m:54-71   load(hrirFile); hL = HRIR_L2702.irChOne; ... (the fields the harness reads)
    load_fixture      verifyEMagLs.m:152-200 the golden filter sets under resources/ (MAT v5/v7)
    save_fixture      verifyEMagLs.m:203-227 save(refFile, <variables>, '-v7') with the harness's variable lists
    fixture_name      verifyEMagLs.m:25-32   resources/HRIR_L2702_<len>samples_<mics>channels_sh<order>_<shDef>_<method>.mat

The published HRIR set comes as a MIRO *object* in a .mat (a MATLAB class instance; the harness itself downloads miro.m to
decode it, :59-66) and as its SOFA twin `HRIR_L2702.sofa` (netCDF-4 / HDF5, SimpleFreeFieldHRIR).  This side reads the SOFA file
directly (emagls_amd/hdf5_min.py, a plain-Python HDF5 reader: the image has no HDF5 module), MAT files of either generation
(-v7 through scipy, -v7.3 through the same HDF5 reader) that hold the five fields as plain arrays or one struct, and .npz.
A -v7 file that holds the MIRO instance itself goes through emagls_amd/mcos.py, a decoder of MATLAB's undocumented object
storage that could only be tried on constructed files here (its docstring says so): prefer the SOFA file.
Damaged files: the HDF5 reader refuses them with Hdf5Error (a ValueError; tests/test_io.py damages the committed files 240 ways);
-v7 MAT files are parsed by scipy, whose compiled reader can crash the process on a damaged object file (600 damaged copies of
a constructed MIRO file: a segmentation fault inside scipy.io.matlab._mio5 among them) -- do not feed it files from strangers.
The plain-array export from MATLAB, where that is the handier route:

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


class _Struct:
    """A MAT v7.3 struct (an HDF5 group of datasets) with the attribute access scipy's mat_struct offers."""

    def __init__(self, fields):
        self.__dict__.update(fields)
        self._fieldnames = list(fields)


def _mat73_value(node):
    from . import hdf5_min as H5
    if isinstance(node, H5.Group):
        if node.attrs.get("MATLAB_class", "struct") != "struct":
            raise ValueError("'%s' is a MATLAB object of class '%s', not plain data; export its fields as plain arrays "
                             "(see the module docstring of emagls_amd.io)" % (node.name, node.attrs.get("MATLAB_class")))
        return _Struct({k: _mat73_value(node[k]) for k in node.keys()})
    cls = node.attrs.get("MATLAB_class", "double")
    if cls not in ("double", "single", "int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "logical"):
        raise ValueError("'%s' is a MATLAB %s, not a numeric array" % (node.name, cls))
    v = node.read()
    if node.attrs.get("MATLAB_empty", 0):
        return np.zeros(tuple(int(x) for x in np.asarray(v).ravel()))
    return np.asarray(v).T          # MATLAB writes column-major data with the dimensions reversed


def _loadmat73(path):
    """MAT v7.3 = HDF5 (emagls_amd/hdf5_min.py): numeric variables and structs of them; '#refs#' / '#subsystem#' (cells,
    objects) are skipped here and reported by the caller when the variable it wants is one of them."""
    from . import hdf5_min as H5
    f = H5.File(path)
    out = {}
    for k in f.keys():
        if k.startswith("#"):
            continue
        try:
            out[k] = _mat73_value(f[k])
        except ValueError as e:
            out[k] = e
    return out


def _loadmat(path):
    import scipy.io as sio
    try:
        return sio.loadmat(path, squeeze_me=False, struct_as_record=False)
    except NotImplementedError:            # MAT v7.3 is HDF5
        return _loadmat73(path)
    except FileNotFoundError:
        raise
    except ValueError:
        raise
    except Exception as e:                 # what scipy's reader raises on a damaged file: OSError, TypeError, UnboundLocalError ...
        raise ValueError("%s could not be read as a MAT file (%s: %s)" % (path, type(e).__name__, e)) from e


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


def _mat_objects(raw, path):
    """The properties of the first classdef object in a -v7 file that has irChOne / irChTwo (the MIRO instance itself,
    verifyEMagLs.m:56: `load(hrirFile)` with miro.m on the path), decoded by emagls_amd/mcos.py -- see its STATUS note."""
    from . import mcos
    if not any(mcos.is_opaque(v) for v in raw.values() if isinstance(v, np.ndarray)):
        return None
    try:
        objs = mcos.object_properties(raw)
    except Exception as e:   # McosError, or whatever scipy / the decoder tripped over in a damaged subsystem element
        raise ValueError("%s holds a MATLAB object that could not be decoded (%s: %s); use the set's .sofa file or export the "
                         "fields as plain arrays (see the module docstring of emagls_amd.io)" % (path, type(e).__name__, e)) from e
    for var, (cls, props) in objs.items():
        if all(f in props for f in _HRIR_FIELDS[:2]):
            return props
    raise ValueError("%s: the objects %s have no irChOne / irChTwo properties" % (path, sorted(objs)))


def load_hrir_set(path):
    """-> dict(hL, hR [numSamples x numDirections], azi, zen [numDirections], fs) as verifyEMagLs.m:67-71 derives them
    (`elevation` holds zenith angles, :70).  Accepts the set's SOFA twin (.sofa, SimpleFreeFieldHRIR), .mat (-v7 or -v7.3)
    with the five MIRO fields at top level or inside one struct variable (any name, e.g. HRIR_L2702), and .npz (hL, hR[, azi,
    zen, fs] or the MIRO field names)."""
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
        raw = _loadmat(path)
        d = {k: v for k, v in raw.items() if not k.startswith("__")}
        objects = _mat_objects(raw, path)
        if all(f in d for f in _HRIR_FIELDS[:2]):
            src = d
        elif objects:
            src = objects
        else:
            src, refused = None, None
            for k, v in d.items():
                if isinstance(v, Exception):       # a v7.3 variable that is not plain data
                    refused = refused or v
                    continue
                v = np.asarray(v)
                if v.dtype == object and v.size == 1 and hasattr(v.ravel()[0], "_fieldnames"):
                    if all(f in v.ravel()[0]._fieldnames for f in _HRIR_FIELDS[:2]):
                        src = v.ravel()[0]
                        break
                if v.dtype.kind in "uV" and k.upper().startswith("HRIR"):    # what loadmat leaves of a class instance
                    raise ValueError("%s holds '%s' as a MATLAB object (MIRO class); export its fields as plain arrays first "
                                     "(see the module docstring of emagls_amd.io)" % (path, k))
            if src is None and refused is not None:
                raise ValueError("%s: %s" % (path, refused))
            if src is None:
                raise ValueError("%s has neither irChOne/irChTwo arrays nor a struct with them" % path)
    elif ext == ".sofa":
        return _finish_hrirs(_load_sofa(path), path)
    else:
        raise ValueError("unsupported HRIR container '%s' (use .sofa, .mat or .npz)" % ext)
    out = dict(hL=_field(src, "irChOne"), hR=_field(src, "irChTwo"))
    for name, key in (("azimuth", "azi"), ("elevation", "zen"), ("fs", "fs")):
        try:
            out[key] = _field(src, name)
        except (KeyError, AttributeError):
            pass
    return _finish_hrirs(out, path)


def _load_sofa(path):
    """The SOFA twin of the set (`HRIR_L2702.sofa`; SimpleFreeFieldHRIR, AES69): Data.IR [M x R x N] with receivers (left,
    right), SourcePosition [M x 3] as (azimuth, elevation, radius) in degrees / metres or Cartesian metres, Data.SamplingRate.
    -> the five quantities verifyEMagLs.m:67-71 takes from the MIRO object (its `elevation` is the zenith angle)."""
    from . import hdf5_min as H5
    f = H5.File(path)
    conv = f.attrs.get("SOFAConventions")
    if "Data.IR" not in f or "SourcePosition" not in f:
        raise ValueError("%s: no Data.IR / SourcePosition (SOFAConventions = %r; a FIR HRIR set is needed)" % (path, conv))
    ir = np.asarray(f["Data.IR"].read(), dtype=np.float64)
    if ir.ndim != 3 or ir.shape[1] != 2:
        raise ValueError("%s: Data.IR is %s, expected [M x 2 x N]" % (path, ir.shape))
    pos_ds = f["SourcePosition"]
    pos = np.asarray(pos_ds.read(), dtype=np.float64)
    if pos.shape[0] == 1 and ir.shape[0] > 1:
        pos = np.repeat(pos, ir.shape[0], axis=0)
    kind = str(pos_ds.attrs.get("Type", "spherical")).strip().lower()
    units = str(pos_ds.attrs.get("Units", "degree, degree, metre")).lower()
    if kind == "spherical":
        scale = 1.0 if units.startswith("rad") else np.pi / 180.0
        azi = np.mod(pos[:, 0] * scale, 2.0 * np.pi)
        zen = np.pi / 2.0 - pos[:, 1] * scale
    elif kind == "cartesian":
        azi = np.mod(np.arctan2(pos[:, 1], pos[:, 0]), 2.0 * np.pi)
        zen = np.arctan2(np.hypot(pos[:, 0], pos[:, 1]), pos[:, 2])
    else:
        raise ValueError("%s: SourcePosition:Type '%s'" % (path, kind))
    out = dict(hL=ir[:, 0, :].T, hR=ir[:, 1, :].T, azi=azi, zen=zen)
    if "Data.SamplingRate" in f:
        out["fs"] = float(np.asarray(f["Data.SamplingRate"].read()).ravel()[0])
    return out


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
