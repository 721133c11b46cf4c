"""Binary PLY ingest / export with the reference's interface (``utils/ply.py``): the scan format of the datasets
(``datasets/InterMultiObj3DSegDataset.py:49``: fields x, y, z, R, G, B, label) and of the mask exports.

    read_ply(filename, triangular_mesh=False)      utils/ply.py:116-189
    write_ply(filename, field_list, field_names, triangular_faces=None)   utils/ply.py:219-312

Same behaviour on the cases the reference's callers rely on: binary little/big endian files only (an ASCII file is a
``ValueError``), ``read_ply`` returns one numpy structured array with the vertex properties under their own names
(with ``triangular_mesh=True``: ``[vertices, faces int32 [F, 3]]``), ``write_ply`` takes 1-D / 2-D arrays whose columns
become the fields, appends ``.ply`` when missing and returns ``True`` (``False`` + a message for inconsistent input).
Host code: the bytes go straight from the file into one ``np.fromfile`` record read; the next step of the path
(voxelisation, ``agile3d_amd.sparse.sparse_quantize``) is where the GPU starts.
"""
from __future__ import annotations

import sys

import numpy as np

# PLY scalar type names (both spellings) -> numpy type codes
_SCALAR = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
           "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
           "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
_BYTE_ORDER = {"binary_little_endian": "<", "binary_big_endian": ">"}


class _Header:
    """Parsed PLY header: byte order and, per element in file order, (name, count, [(property, dtype) | list spec])."""

    def __init__(self, order, elements):
        self.order = order
        self.elements = elements

    def element(self, name):
        for e in self.elements:
            if e[0] == name:
                return e
        return None


def _read_header(f) -> _Header:
    magic = f.readline()
    if b"ply" not in magic:
        raise ValueError("The file does not start whith the word ply")
    fmt_line = f.readline().split()
    if len(fmt_line) < 2 or fmt_line[0] != b"format":
        raise ValueError("PLY header: second line must be the format line")
    fmt = fmt_line[1].decode()
    if fmt == "ascii":
        raise ValueError("The file is not binary")
    if fmt not in _BYTE_ORDER:
        raise ValueError(f"PLY header: unknown format {fmt}")
    order = _BYTE_ORDER[fmt]
    elements = []
    while True:
        line = f.readline()
        if line == b"":
            raise ValueError("PLY header: end_header not found")
        tok = line.split()
        if not tok or tok[0] == b"comment" or tok[0] == b"obj_info":
            continue
        if tok[0] == b"end_header":
            break
        if tok[0] == b"element":
            elements.append((tok[1].decode(), int(tok[2]), []))
        elif tok[0] == b"property":
            if not elements:
                raise ValueError("PLY header: property before any element")
            props = elements[-1][2]
            if tok[1] == b"list":
                props.append((tok[4].decode(), ("list", order + _SCALAR[tok[2].decode()], order + _SCALAR[tok[3].decode()])))
            else:
                name = tok[1].decode()
                if name not in _SCALAR:
                    raise ValueError(f"PLY header: unknown property type {name}")
                props.append((tok[2].decode(), order + _SCALAR[name]))
    return _Header(order, elements)


def read_ply(filename, triangular_mesh=False):
    """Read a binary ``.ply`` file (utils/ply.py:116-189).  Point clouds: one structured array of the vertex
    element.  ``triangular_mesh=True``: ``[vertex array, int32 faces [F, 3]]`` (faces must be ``uchar``-counted
    triangles of 32-bit indices, the only layout the reference accepts)."""
    with open(filename, "rb") as f:
        hdr = _read_header(f)
        if not hdr.elements:
            raise ValueError("PLY file without elements")
        if not triangular_mesh:
            # the reference reads the properties it finds as ONE record type counted by the last element line; files
            # of the datasets hold a single vertex element
            name, count, props = hdr.elements[0]
            if any(isinstance(t, tuple) for _, t in props):
                raise ValueError("list properties need triangular_mesh=True")
            if len(hdr.elements) > 1:
                raise ValueError("more than one element: read with triangular_mesh=True")
            return np.fromfile(f, dtype=[(n, t) for n, t in props], count=count)
        vert = hdr.element("vertex")
        face = hdr.element("face")
        if vert is None or face is None or hdr.elements[0][0] != "vertex":
            raise ValueError("triangular_mesh=True needs a vertex element followed by a face element")
        vertices = np.fromfile(f, dtype=[(n, t) for n, t in vert[2]], count=vert[1])
        if len(face[2]) != 1 or not isinstance(face[2][0][1], tuple):
            raise ValueError("Unsupported faces property")
        _, cnt_t, idx_t = face[2][0][1]
        if np.dtype(cnt_t).itemsize != 1 or np.dtype(idx_t).itemsize != 4:
            raise ValueError("Unsupported faces property : only 'list uchar int' triangles")
        rec = np.fromfile(f, dtype=[("k", cnt_t), ("v1", idx_t), ("v2", idx_t), ("v3", idx_t)], count=face[1])
        if rec.size and (rec["k"] != 3).any():
            raise ValueError("faces are not all triangles")
        faces = np.vstack((rec["v1"], rec["v2"], rec["v3"])).T
        return [vertices, faces]


def write_ply(filename, field_list, field_names, triangular_faces=None):
    """Write a binary ``.ply`` file in the machine's byte order (utils/ply.py:219-312).  ``field_list``: one array
    or a list/tuple of arrays, every 1-D array and every column of a 2-D array is one field, named by
    ``field_names`` in order.  Returns True, or False (after printing why) when the fields do not fit together."""
    fields = list(field_list) if isinstance(field_list, (list, tuple)) else [field_list]
    columns = []
    for a in fields:
        a = np.asarray(a)
        if a.ndim > 2:
            print("fields have more than 2 dimensions")
            return False
        if a.ndim < 2:
            a = a.reshape(-1, 1)
        columns += [a[:, c] for c in range(a.shape[1])]
    if any(len(c) != len(columns[0]) for c in columns):
        print("wrong field dimensions")
        return False
    if len(columns) != len(field_names):
        print("wrong number of field names")
        return False
    if not filename.endswith(".ply"):
        filename += ".ply"
    n = len(columns[0]) if columns else 0
    header = ["ply", f"format binary_{sys.byteorder}_endian 1.0", f"element vertex {n}"]
    header += [f"property {c.dtype.name} {name}" for c, name in zip(columns, field_names)]
    if triangular_faces is not None:
        header += [f"element face {triangular_faces.shape[0]:d}", "property list uchar int vertex_indices"]
    header.append("end_header")
    record = np.empty(n, dtype=[(name, c.dtype.str) for c, name in zip(columns, field_names)])
    for c, name in zip(columns, field_names):
        record[name] = c
    with open(filename, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        record.tofile(f)
        if triangular_faces is not None:
            tri = np.asarray(triangular_faces).astype(np.int32)
            faces = np.empty(tri.shape[0], dtype=[("k", "uint8"), ("0", "int32"), ("1", "int32"), ("2", "int32")])
            faces["k"] = 3
            faces["0"], faces["1"], faces["2"] = tri[:, 0], tri[:, 1], tri[:, 2]
            faces.tofile(f)
    return True
