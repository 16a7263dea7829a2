"""Test helper: writes HDF5 files through the image's libhdf5 (ctypes; the library is test tooling only, the product reads
HDF5 with emagls_amd/hdf5_min.py).  `find_libhdf5()` returns None when the library is absent -- tests that need fresh
files skip then, the committed fixtures under tests/golden/ still run.

    python tests/h5gen.py        regenerates tests/golden/hrir_small.sofa and tests/golden/hrir_small_v73.mat
"""
from __future__ import annotations

import ctypes as C
import glob
import os

import numpy as np

hid = C.c_int64
_LIB = None


def find_libhdf5():
    global _LIB
    if _LIB is not None:
        return _LIB or None
    cands = [os.environ.get("EMAGLS_LIBHDF5", "")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + ["libhdf5.so"]
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            lib.H5open()
            _LIB = lib
            _declare(lib)
            return lib
        except OSError:
            continue
    _LIB = False
    return None


def _declare(h):
    def f(name, res, *args):
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = list(args)
    P = C.c_void_p
    f("H5Fcreate", hid, C.c_char_p, C.c_uint, hid, hid)
    f("H5Fclose", C.c_int, hid)
    f("H5Pcreate", hid, hid)
    f("H5Pclose", C.c_int, hid)
    f("H5Pset_libver_bounds", C.c_int, hid, C.c_int, C.c_int)
    f("H5Pset_link_creation_order", C.c_int, hid, C.c_uint)
    f("H5Pset_attr_creation_order", C.c_int, hid, C.c_uint)
    f("H5Pset_chunk", C.c_int, hid, C.c_int, P)
    f("H5Pset_deflate", C.c_int, hid, C.c_uint)
    f("H5Pset_shuffle", C.c_int, hid)
    f("H5Pset_fletcher32", C.c_int, hid)
    f("H5Pset_userblock", C.c_int, hid, C.c_uint64)
    f("H5Screate_simple", hid, C.c_int, P, P)
    f("H5Screate", hid, C.c_int)
    f("H5Sclose", C.c_int, hid)
    f("H5Dcreate2", hid, hid, C.c_char_p, hid, hid, hid, hid, hid)
    f("H5Dwrite", C.c_int, hid, hid, hid, hid, hid, P)
    f("H5Dclose", C.c_int, hid)
    f("H5Gcreate2", hid, hid, C.c_char_p, hid, hid, hid)
    f("H5Gclose", C.c_int, hid)
    f("H5Acreate2", hid, hid, C.c_char_p, hid, hid, hid, hid)
    f("H5Awrite", C.c_int, hid, hid, P)
    f("H5Aclose", C.c_int, hid)
    f("H5Tcopy", hid, hid)
    f("H5Tset_size", C.c_int, hid, C.c_size_t)
    f("H5Tset_strpad", C.c_int, hid, C.c_int)
    f("H5Tclose", C.c_int, hid)


def _g(h, name):
    return hid.in_dll(h, name).value


_NATIVE = {"f8": "H5T_NATIVE_DOUBLE_g", "f4": "H5T_NATIVE_FLOAT_g", "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g",
           "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g", "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g",
           "i2": "H5T_NATIVE_INT16_g", "i1": "H5T_NATIVE_INT8_g"}
_FILE_BE = {"f8": "H5T_IEEE_F64BE_g", "f4": "H5T_IEEE_F32BE_g", "i4": "H5T_STD_I32BE_g"}


class Writer:
    """w = Writer(path, latest=False, track_order=False, userblock=0); w.dataset(...); w.attr(...); w.group(...); w.close()"""

    def __init__(self, path, latest=False, track_order=False, userblock=0):
        h = find_libhdf5()
        if h is None:
            raise RuntimeError("libhdf5 not found")
        self.h = h
        self.track = track_order
        fapl = h.H5Pcreate(_g(h, "H5P_CLS_FILE_ACCESS_ID_g"))
        if latest:
            h.H5Pset_libver_bounds(fapl, 2, 2)      # H5F_LIBVER_LATEST in 1.10.x (EARLIEST 0, V18 1, V110 2)
        fcpl = h.H5Pcreate(_g(h, "H5P_CLS_FILE_CREATE_ID_g"))
        if track_order:
            h.H5Pset_link_creation_order(fcpl, 3)   # tracked | indexed, as netCDF-4 sets on every group
        if userblock:
            h.H5Pset_userblock(fcpl, userblock)
        self.fid = h.H5Fcreate(path.encode(), 2, fcpl, fapl)   # H5F_ACC_TRUNC
        if self.fid < 0:
            raise RuntimeError("H5Fcreate failed for %s" % path)
        h.H5Pclose(fapl)
        h.H5Pclose(fcpl)
        self.groups = {"/": self.fid}

    def group(self, name):
        h = self.h
        gcpl = h.H5Pcreate(_g(h, "H5P_CLS_GROUP_CREATE_ID_g"))
        if self.track:
            h.H5Pset_link_creation_order(gcpl, 3)
        gid = h.H5Gcreate2(self.fid, name.encode(), 0, gcpl, 0)
        h.H5Pclose(gcpl)
        self.groups[name] = gid
        return gid

    def _space(self, shape):
        h = self.h
        if len(shape) == 0:
            return h.H5Screate(0)       # H5S_SCALAR
        dims = (C.c_uint64 * len(shape))(*shape)
        return h.H5Screate_simple(len(shape), dims, None)

    def _types(self, a, big_endian=False):
        """-> (memory type, file type, owned) for a numeric ndarray or a fixed-length bytes array"""
        h = self.h
        if a.dtype.kind == "S":
            t = h.H5Tcopy(_g(h, "H5T_C_S1_g"))
            h.H5Tset_size(t, a.dtype.itemsize)
            h.H5Tset_strpad(t, 1)       # null padded, like netCDF char arrays
            return t, t, [t]
        key = a.dtype.str[1:]
        mem = _g(h, _NATIVE[key])
        return mem, (_g(h, _FILE_BE[key]) if big_endian else mem), []

    def dataset(self, name, array, chunks=None, deflate=None, shuffle=False, fletcher=False, big_endian=False, attrs=None):
        h = self.h
        a = np.ascontiguousarray(array)
        mem, ftype, owned = self._types(a, big_endian)
        sp = self._space(a.shape)
        dcpl = h.H5Pcreate(_g(h, "H5P_CLS_DATASET_CREATE_ID_g"))
        if self.track:
            h.H5Pset_attr_creation_order(dcpl, 3)
        if chunks is not None:
            cd = (C.c_uint64 * len(chunks))(*chunks)
            h.H5Pset_chunk(dcpl, len(chunks), cd)
            if shuffle:
                h.H5Pset_shuffle(dcpl)
            if deflate is not None:
                h.H5Pset_deflate(dcpl, deflate)
            if fletcher:
                h.H5Pset_fletcher32(dcpl)
        did = h.H5Dcreate2(self.fid, name.encode(), ftype, sp, 0, dcpl, 0)
        if did < 0:
            raise RuntimeError("H5Dcreate2 failed for %s" % name)
        if h.H5Dwrite(did, mem, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) < 0:
            raise RuntimeError("H5Dwrite failed for %s" % name)
        for k, v in (attrs or {}).items():
            self._attr(did, k, v)
        h.H5Dclose(did)
        h.H5Pclose(dcpl)
        h.H5Sclose(sp)
        for t in owned:
            h.H5Tclose(t)

    def attr(self, where, name, value, vlen=False):
        self._attr(self.groups[where], name, value, vlen)

    def _attr(self, loc, name, value, vlen=False):
        h = self.h
        if isinstance(value, str):
            raw = value.encode()
            if vlen:
                t = h.H5Tcopy(_g(h, "H5T_C_S1_g"))
                h.H5Tset_size(t, C.c_size_t(-1).value)      # H5T_VARIABLE
                sp = h.H5Screate(0)
                aid = h.H5Acreate2(loc, name.encode(), t, sp, 0, 0)
                buf = C.c_char_p(raw)
                h.H5Awrite(aid, t, C.byref(buf))
            else:
                t = h.H5Tcopy(_g(h, "H5T_C_S1_g"))
                h.H5Tset_size(t, max(len(raw), 1))
                h.H5Tset_strpad(t, 1)
                sp = h.H5Screate(0)
                aid = h.H5Acreate2(loc, name.encode(), t, sp, 0, 0)
                buf = C.create_string_buffer(raw, max(len(raw), 1))
                h.H5Awrite(aid, t, buf)
            h.H5Aclose(aid)
            h.H5Sclose(sp)
            h.H5Tclose(t)
            return
        a = np.ascontiguousarray(value)
        mem, ftype, owned = self._types(a)
        sp = self._space(a.shape)
        aid = h.H5Acreate2(loc, name.encode(), ftype, sp, 0, 0)
        h.H5Awrite(aid, mem, a.ctypes.data_as(C.c_void_p))
        h.H5Aclose(aid)
        h.H5Sclose(sp)
        for t in owned:
            h.H5Tclose(t)

    def close(self):
        for k, g in self.groups.items():
            if k != "/":
                self.h.H5Gclose(g)
        self.h.H5Fclose(self.fid)


def small_hrir_set(seed=5, ndirs=38, nsamp=24, fs=48000.0):
    rng = np.random.default_rng(seed)
    azi = rng.uniform(0.0, 2 * np.pi, ndirs)
    zen = np.arccos(rng.uniform(-1.0, 1.0, ndirs))
    hL = rng.standard_normal((nsamp, ndirs)) * np.exp(-np.arange(nsamp) / 6.0)[:, None]
    hR = rng.standard_normal((nsamp, ndirs)) * np.exp(-np.arange(nsamp) / 6.0)[:, None]
    return dict(hL=hL, hR=hR, azi=azi, zen=zen, fs=fs)


def write_sofa(path, hrirs, position_type="spherical", chunked=True, vlen_attrs=False, latest=False, ir_dtype="f8"):
    """A SimpleFreeFieldHRIR file the way the SOFA API (netCDF-4) lays it out: creation-order-tracked root group with more
    than eight links (dense link storage), dimension variables, Data.IR [M x R x N] chunked + shuffled + deflated."""
    M, N = hrirs["hL"].shape[1], hrirs["hL"].shape[0]
    w = Writer(path, latest=latest, track_order=True)
    for k, v in (("Conventions", "SOFA"), ("SOFAConventions", "SimpleFreeFieldHRIR"), ("SOFAConventionsVersion", "1.0"),
                 ("Version", "1.0"), ("DataType", "FIR"), ("RoomType", "free field"), ("Title", "small test set"),
                 ("APIName", "h5gen"), ("APIVersion", "0"), ("AuthorContact", "-"), ("Organization", "-"), ("License", "-"),
                 ("DateCreated", "2020-01-01 00:00:00"), ("DateModified", "2020-01-01 00:00:00"), ("DatabaseName", "-"),
                 ("ListenerShortName", "KU100")):
        w.attr("/", k, v, vlen=vlen_attrs)
    for name, n in (("I", 1), ("C", 3), ("R", 2), ("E", 1), ("N", N), ("M", M)):
        w.dataset(name, np.zeros(n, dtype=np.float32), attrs={"CLASS": "DIMENSION_SCALE", "NAME": "This is a netCDF dimension but not a netCDF variable."})
    ir = np.stack([hrirs["hL"].T, hrirs["hR"].T], axis=1).astype(ir_dtype)        # [M x R x N]
    w.dataset("Data.IR", ir, chunks=(min(M, 16), 2, N) if chunked else None, deflate=4 if chunked else None, shuffle=chunked)
    w.dataset("Data.SamplingRate", np.array([hrirs["fs"]]), attrs={"Units": "hertz"})
    w.dataset("Data.Delay", np.zeros((1, 2)))
    ele = np.pi / 2 - hrirs["zen"]
    if position_type == "spherical":
        pos = np.stack([np.degrees(hrirs["azi"]), np.degrees(ele), np.full(M, 3.25)], axis=1)
        units = "degree, degree, metre"
    else:
        rad = 3.25
        pos = np.stack([rad * np.cos(ele) * np.cos(hrirs["azi"]), rad * np.cos(ele) * np.sin(hrirs["azi"]), rad * np.sin(ele)], axis=1)
        units = "metre"
    w.dataset("SourcePosition", pos, chunks=(M, 3) if chunked else None, deflate=1 if chunked else None,
              attrs={"Type": position_type, "Units": units})
    w.dataset("ListenerPosition", np.zeros((1, 3)), attrs={"Type": "cartesian", "Units": "metre"})
    w.dataset("ListenerUp", np.array([[0.0, 0.0, 1.0]]))
    w.dataset("ListenerView", np.array([[1.0, 0.0, 0.0]]), attrs={"Type": "cartesian", "Units": "metre"})
    w.dataset("ReceiverPosition", np.array([[[0.0], [0.09], [0.0]], [[0.0], [-0.09], [0.0]]]).reshape(2, 3, 1),
              attrs={"Type": "cartesian", "Units": "metre"})
    w.dataset("EmitterPosition", np.zeros((1, 3, 1)), attrs={"Type": "cartesian", "Units": "metre"})
    w.close()


def write_mat73(path, hrirs, as_struct=True):
    """What `save(..., '-v7.3')` writes for the five MIRO fields exported as a struct (or at top level): a 512-byte user
    block with the MAT header, datasets in MATLAB's column-major order (dimensions reversed), MATLAB_class attributes."""
    w = Writer(path, userblock=512)
    prefix = ""
    if as_struct:
        w.group("HRIR_L2702")
        w.attr("HRIR_L2702", "MATLAB_class", "struct")
        prefix = "HRIR_L2702/"
    for k, v in (("irChOne", hrirs["hL"]), ("irChTwo", hrirs["hR"]), ("azimuth", hrirs["azi"][None, :]),
                 ("elevation", hrirs["zen"][None, :]), ("fs", np.array([[hrirs["fs"]]]))):
        w.dataset(prefix + k, np.ascontiguousarray(np.asarray(v, dtype=np.float64).T), attrs={"MATLAB_class": "double"})
    w.close()
    stamp_mat73_header(path)


def stamp_mat73_header(path):
    head = ("MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Thu Jan  1 00:00:00 2020 HDF5 schema 1.00 .").encode()
    with open(path, "r+b") as f:
        f.write(head.ljust(116, b" ") + b"\0" * 8 + b"\x00\x02IM")


if __name__ == "__main__":
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    hs = small_hrir_set()
    write_sofa(os.path.join(here, "hrir_small.sofa"), hs)
    write_mat73(os.path.join(here, "hrir_small_v73.mat"), hs)
    np.savez(os.path.join(here, "hrir_small_expected.npz"), **hs)
    print("wrote", sorted(f for f in os.listdir(here) if f.startswith("hrir_small")))
