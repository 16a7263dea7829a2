"""Test helper: builds a MAT -v7 file that holds one classdef object (e.g. a MIRO instance) the way emagls_amd/mcos.py
describes MATLAB's layout -- opaque variable + subsystem element -- byte by byte (scipy cannot write either).  The file is
a construction from the description, not MATLAB output."""
from __future__ import annotations

import struct

import numpy as np

miINT8, miUINT8, miINT32, miUINT32, miDOUBLE, miMATRIX = 1, 2, 5, 6, 9, 14
mxCELL, mxSTRUCT, mxCHAR, mxDOUBLE, mxSINGLE, mxUINT8, mxUINT32, mxOPAQUE = 1, 2, 4, 6, 7, 9, 13, 17
miSINGLE, miUINT16 = 7, 4


def _elem(mdtype, payload):
    pad = (-len(payload)) % 8
    return struct.pack("<II", mdtype, len(payload)) + payload + b"\0" * pad


def _flags(mclass):
    return _elem(miUINT32, struct.pack("<II", mclass, 0))


def _dims(shape):
    return _elem(miINT32, struct.pack("<%di" % len(shape), *shape))


def _name(name):
    return _elem(miINT8, name.encode())


def matrix(a, name=""):
    """numeric ndarray / str / list (cell column) / dict (1x1 struct) -> miMATRIX element"""
    if isinstance(a, dict):
        flen = max(len(k) for k in a) + 1 if a else 1
        names = b"".join(k.encode().ljust(flen, b"\0") for k in a)
        body = _flags(mxSTRUCT) + _dims((1, 1)) + _name(name) + _elem(miINT32, struct.pack("<i", flen)) + _elem(miINT8, names)
        body += b"".join(matrix(v) for v in a.values())
        return _elem(miMATRIX, body)
    if isinstance(a, list):
        body = _flags(mxCELL) + _dims((len(a), 1)) + _name(name) + b"".join(matrix(v) for v in a)
        return _elem(miMATRIX, body)
    if isinstance(a, bytes):        # an element that is already encoded (an opaque)
        return a
    if isinstance(a, str):
        body = _flags(mxCHAR) + _dims((1, len(a))) + _name(name) + _elem(miUINT16, np.array([ord(c) for c in a], dtype="<u2").tobytes())
        return _elem(miMATRIX, body)
    a = np.asarray(a)
    if a.ndim < 2:
        a = a.reshape(-1, 1) if a.ndim == 1 else a.reshape(1, 1)
    table = {"float64": (mxDOUBLE, miDOUBLE), "float32": (mxSINGLE, miSINGLE), "uint8": (mxUINT8, miUINT8), "uint32": (mxUINT32, miUINT32)}
    mclass, mdtype = table[a.dtype.name]
    body = _flags(mclass) + _dims(a.shape) + _name(name) + _elem(mdtype, np.asfortranarray(a).tobytes(order="F"))
    return _elem(miMATRIX, body)


def opaque(var, class_name, payload):
    body = _flags(mxOPAQUE) + _name(var) + _name("MCOS") + _name(class_name) + payload
    return _elem(miMATRIX, body)


def linking_metadata(class_name, prop_names, version=3, extra_names=()):
    """One class, one object whose properties (all of kind 1) sit in cells 2, 3, ... in the order given."""
    names = [class_name] + list(extra_names) + list(prop_names)
    strings = b"".join(n.encode() + b"\0" for n in names)
    strings += b"\0" * ((-(40 + len(strings))) % 8)
    r1 = struct.pack("<8I", 0, 0, 0, 0, 0, 1, 0, 0)
    r2 = struct.pack("<2I", 0, 0)
    r3 = struct.pack("<12I", 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1)
    trip = [len(prop_names)]
    for i, p in enumerate(prop_names):
        trip += [names.index(p) + 1, 1, i]
    if len(trip) % 2:
        trip.append(0)
    r4 = struct.pack("<2I", 0, 0) + struct.pack("<%dI" % len(trip), *trip)
    r5 = struct.pack("<2I", 0, 0)
    offs, pos = [], 40 + len(strings)
    for r in (r1, r2, r3, r4, r5):
        offs.append(pos)
        pos += len(r)
    offs += [pos, pos, pos]
    blob = struct.pack("<2I", version, len(names)) + struct.pack("<8I", *offs) + strings + r1 + r2 + r3 + r4 + r5
    return np.frombuffer(blob, dtype=np.uint8)


def write_object_mat(path, var, class_name, props, defaults=None, version=3):
    """props: {name: ndarray or str}; defaults: {name: value} the class carries for properties the object leaves unset."""
    meta = linking_metadata(class_name, list(props), version=version, extra_names=list(defaults or {}))
    cells = [meta, np.zeros((0, 0))] + list(props.values())
    cells += [[np.zeros((0, 0)), np.zeros((0, 0))], [np.zeros((0, 0)), np.zeros((0, 0))]]
    cells += [[np.zeros((0, 0)), dict(defaults) if defaults else dict(unused=np.zeros((0, 0)))]]
    wrapper = opaque("", "FileWrapper__", matrix(cells))
    subsystem = b"\x00\x01IM\0\0\0\0" + matrix({"MCOS": wrapper})
    ref = np.array([0xDD000000, 2, 1, 1, 1, 1], dtype=np.uint32)
    head = ("MATLAB 5.0 MAT-file, Platform: GLNXA64, Created by tests/mcosgen.py").encode().ljust(116, b" ")
    var_elem = opaque(var, class_name, matrix(ref))
    sub_elem = matrix(np.frombuffer(subsystem, dtype=np.uint8).reshape(1, -1))
    with open(path, "wb") as f:
        f.write(head + struct.pack("<Q", 128 + len(var_elem)) + b"\x00\x01IM")
        f.write(var_elem + sub_elem)
