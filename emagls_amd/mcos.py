"""Property values of MATLAB class instances (classdef objects, e.g. the MIRO object `HRIR_L2702` of verifyEMagLs.m:56-71)
stored in MAT -v7 files -- host-side file handling for `emagls_amd.io`, no device work.

MATLAB writes such a variable as an "opaque" element (type system 'MCOS', class name, and a small uint32 reference:
[0xDD000000, ndims, dims..., object ids..., class id]) and keeps the property data in the file's subsystem element, which
scipy hands over undecoded as `__function_workspace__`.  The subsystem is a nested MAT stream holding a struct with a
field `MCOS`: an opaque `FileWrapper__` whose payload is a cell array

    cell 0        linking metadata (uint8): version, number of names, eight region offsets, the names (class and property
                  names, NUL-terminated), then the regions:
                    1  class table        4 x uint32 per class   (namespace name index, class name index, 0, 0)
                    2  property blocks of objects saved through saveobj ("type 1")
                    3  object table       6 x uint32 per object  (class id, 0, 0, type-1 block id, type-2 block id, dependency id)
                    4  property blocks of ordinary objects ("type 2"):  nprops, then nprops x (name index, kind, value),
                       padded to 8 bytes; kind 1: value = index of the cell that holds the data (counted from cell 2),
                       kind 0: value = index of a name (the property is that string), kind 2: the value itself (logical)
                    5+ dynamic properties and two regions of unknown use
                  (every table starts with an all-zero entry; ids and name indices count from 1)
    cell 1        unused
    cell 2 ...    property values
    last cell     per class, a struct of the property defaults (what an unset property reads as)

MATLAB does not document this layout; the description above is the one the open-source readers agree on.  STATUS: no
MATLAB-written object file exists in this image, so the decoder is exercised only on files that tests/mcosgen.py builds from
this same description -- it is unverified against MATLAB's own output, and io.load_hrir_set says which route it took.
"""
from __future__ import annotations

import io as _io

import numpy as np

_OPAQUE_FIELDS = ("s0", "s1", "s2", "arr")


class McosError(ValueError):
    pass


def is_opaque(v):
    return isinstance(v, np.ndarray) and v.dtype.names == _OPAQUE_FIELDS


def _text(v):
    if isinstance(v, bytes):
        return v.decode("latin1")
    if isinstance(v, np.ndarray):
        return v.tobytes().decode("latin1") if v.dtype.kind in "Su" and v.dtype.itemsize == 1 else "".join(str(x) for x in v.ravel())
    return str(v)


def _subsystem_cells(workspace):
    """`__function_workspace__` (uint8 [1 x n]) -> the FileWrapper__ cell array as a list."""
    import scipy.io as sio
    raw = np.asarray(workspace, dtype=np.uint8).tobytes()
    if len(raw) < 16 or raw[2:4] not in (b"IM", b"MI"):
        raise McosError("the subsystem element does not start with a MAT stream header")
    head = b"MATLAB 5.0 MAT-file, subsystem".ljust(116, b" ") + b"\0" * 8 + raw[0:4]
    inner = sio.loadmat(_io.BytesIO(head + raw[8:]), squeeze_me=False, struct_as_record=False)
    top = inner.get("__function_workspace__")
    if top is None:
        raise McosError("the subsystem holds no unnamed element")
    node = np.asarray(top).ravel()[0]
    if not hasattr(node, "MCOS"):
        raise McosError("the subsystem has no MCOS field (fields: %s)" % getattr(node, "_fieldnames", None))
    wrapper = node.MCOS
    if not is_opaque(wrapper) or "FileWrapper__" not in _text(wrapper["s2"][0]):
        raise McosError("the MCOS field is not a FileWrapper__ object")
    cells = np.asarray(wrapper["arr"][0], dtype=object).ravel()
    if cells.size < 3:
        raise McosError("FileWrapper__ holds %d cells" % cells.size)
    return list(cells)


class _Metadata:
    def __init__(self, blob):
        b = np.asarray(blob, dtype=np.uint8).tobytes()
        if len(b) < 40:
            raise McosError("linking metadata of %d bytes" % len(b))
        u = lambda p: int.from_bytes(b[p:p + 4], "little")
        self.version = u(0)
        if not 2 <= self.version <= 4:
            raise McosError("linking metadata version %d (2-4 are known)" % self.version)
        nnames = u(4)
        offs = [u(8 + 4 * i) for i in range(8)]
        if any(o > len(b) for o in offs) or offs[0] < 40:
            raise McosError("region offsets %s outside the %d metadata bytes" % (offs, len(b)))
        names = b[40:offs[0]].split(b"\0")
        self.names = [n.decode("latin1") for n in names[:nnames]]
        if len(self.names) < nnames:
            raise McosError("%d names announced, %d found" % (nnames, len(self.names)))
        w = lambda lo, hi: [u(p) for p in range(lo, hi - 3, 4)]
        r1, r2, r3, r4 = (w(offs[i], offs[i + 1]) for i in range(4))
        self.classes = [tuple(r1[i:i + 4]) for i in range(0, len(r1) - 3, 4)]
        self.objects = [tuple(r3[i:i + 6]) for i in range(0, len(r3) - 5, 6)]
        self.blocks = {1: self._blocks(r2), 2: self._blocks(r4)}

    @staticmethod
    def _blocks(words):
        """property blocks: [0, 0] then per block nprops + triplets, padded to an even number of words"""
        out = [[]]
        p = 2
        while p < len(words):
            n = words[p]
            if p + 1 + 3 * n > len(words):
                raise McosError("property block of %d entries runs past its region" % n)
            out.append([tuple(words[p + 1 + 3 * i:p + 4 + 3 * i]) for i in range(n)])
            p += 1 + 3 * n
            p += p % 2
        return out

    def name(self, idx):
        if not 1 <= idx <= len(self.names):
            raise McosError("name index %d of %d" % (idx, len(self.names)))
        return self.names[idx - 1]


def object_properties(matdict):
    """{variable name: (class name, {property: value})} for every classdef object among the variables of a loadmat() result
    (struct_as_record=False).  Properties the file does not set read as the class defaults the file carries."""
    found = {k: v for k, v in matdict.items() if is_opaque(v)}
    if not found:
        return {}
    if "__function_workspace__" not in matdict:
        raise McosError("the file holds class instances but no subsystem element")
    cells = _subsystem_cells(matdict["__function_workspace__"])
    md = _Metadata(cells[0])
    out = {}
    for key, v in found.items():
        rec = v.ravel()[0]
        var = _text(rec["s0"]) or key
        if _text(rec["s1"]) != "MCOS":
            continue
        ref = np.asarray(rec["arr"]).ravel().astype(np.uint64)
        if ref.size < 6 or int(ref[0]) != 0xDD000000:
            raise McosError("'%s': not an object reference (a class that was unknown when the file was written?)" % var)
        ndims = int(ref[1])
        count = int(np.prod(ref[2:2 + ndims]))
        if count != 1:
            raise McosError("'%s' is a %s object array; one object is expected" % (var, "x".join(str(int(d)) for d in ref[2:2 + ndims])))
        obj_id, class_id = int(ref[2 + ndims]), int(ref[2 + ndims + count])
        if not 1 <= obj_id < len(md.objects) or not 1 <= class_id < len(md.classes):
            raise McosError("'%s': object id %d / class id %d outside the tables" % (var, obj_id, class_id))
        cls, _, _, t1, t2, _ = md.objects[obj_id]
        if t2 >= len(md.blocks[2]) or t1 >= len(md.blocks[1]):
            raise McosError("'%s': property block %d / %d outside the tables" % (var, t1, t2))
        block = md.blocks[2][t2] if t2 else md.blocks[1][t1]
        props = {}
        defaults = np.asarray(cells[-1], dtype=object).ravel()
        if class_id < defaults.size and hasattr(defaults[class_id], "ravel"):
            d = defaults[class_id].ravel()
            if d.size and hasattr(d[0], "_fieldnames"):
                props.update({f: getattr(d[0], f) for f in d[0]._fieldnames})
        for name_idx, kind, value in block:
            if kind == 1:
                if not 0 <= value + 2 < len(cells) - 1:
                    raise McosError("'%s.%s' points at cell %d of %d" % (var, md.name(name_idx), value + 2, len(cells)))
                props[md.name(name_idx)] = cells[value + 2]
            elif kind == 0:
                props[md.name(name_idx)] = md.name(value)
            else:
                props[md.name(name_idx)] = bool(value)
        out[var] = (md.name(md.classes[class_id][1]), props)
    return out
