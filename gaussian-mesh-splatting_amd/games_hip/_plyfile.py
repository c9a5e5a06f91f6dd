"""Minimal stand-in for the third-party `plyfile` package the reference imports (scene/gaussian_model.py:19,
scene/dataset_readers.py:22; absent from this image) -- SURVEY.md section 8(f) item 4.

Implements the part of its API the reference calls:
    PlyData.read(path)                       scene/gaussian_model.py:227, scene/dataset_readers.py:108
    plydata['vertex'] / plydata.elements[0]  element lookup by name / position
    element['x'], element.properties[i].name, element.count, element.data
    PlyElement.describe(structured_array, 'vertex')      scene/gaussian_model.py:215, scene/dataset_readers.py:129
    PlyData([element], text=False).write(path)
for the PLY 1.0 format (ascii, binary_little_endian, binary_big_endian), scalar properties and list properties
(faces of mesh files).  numpy only; files written here are byte-compatible with what `plyfile` writes for the same
structured array (header line order: format, element, properties in dtype order, end_header)."""
from __future__ import annotations

import numpy as np

_PLY_TO_NP = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
              "f8": "double"}


class PlyProperty:
    def __init__(self, name, val_dtype, len_dtype=None):
        self.name, self.val_dtype, self.len_dtype = name, val_dtype, len_dtype      # numpy codes without byte order

    @property
    def is_list(self):
        return self.len_dtype is not None

    def __repr__(self):
        return f"PlyProperty({self.name!r}, {self.val_dtype!r})"


class PlyElement:
    def __init__(self, name, properties, count, data=None):
        self.name, self.properties, self.count, self.data = name, list(properties), int(count), data

    @staticmethod
    def describe(data, name, **_unused):
        data = np.asarray(data)
        if data.dtype.names is None:
            raise ValueError("PlyElement.describe needs a structured array")
        props = []
        for n in data.dtype.names:
            dt = data.dtype.fields[n][0]
            if dt.shape or dt.kind not in "iuf" or dt.str[1:] not in _NP_TO_PLY:
                raise ValueError(f"unsupported dtype for property {n}: {dt}")
            props.append(PlyProperty(n, dt.str[1:]))
        return PlyElement(name, props, data.shape[0], data)

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return self.count

    def ply_property(self, name):
        for p in self.properties:
            if p.name == name:
                return p
        raise KeyError(name)


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="<", comments=()):
        self.elements, self.text, self.byte_order, self.comments = list(elements), bool(text), byte_order, list(comments)

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def __contains__(self, name):
        return any(e.name == name for e in self.elements)

    # ------------------------------------------------------------------ reading
    @staticmethod
    def read(stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            if f.readline().strip() != b"ply":
                raise ValueError("not a PLY file")
            fmt, elements, comments = None, [], []
            while True:
                line = f.readline()
                if not line:
                    raise ValueError("unexpected end of PLY header")
                tok = line.decode("ascii", "replace").strip().split()
                if not tok:
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] in ("comment", "obj_info"):
                    comments.append(" ".join(tok[1:]))
                elif tok[0] == "element":
                    elements.append(PlyElement(tok[1], [], int(tok[2])))
                elif tok[0] == "property":
                    if tok[1] == "list":
                        elements[-1].properties.append(PlyProperty(tok[4], _PLY_TO_NP[tok[3]], _PLY_TO_NP[tok[2]]))
                    else:
                        elements[-1].properties.append(PlyProperty(tok[2], _PLY_TO_NP[tok[1]]))
                elif tok[0] == "end_header":
                    break
            if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
                raise ValueError(f"unsupported PLY format {fmt!r}")
            bo = ">" if fmt == "binary_big_endian" else "<"
            for e in elements:
                e.data = PlyData._read_element(f, e, fmt, bo)
            return PlyData(elements, text=fmt == "ascii", byte_order=bo, comments=comments)
        finally:
            if own:
                f.close()

    @staticmethod
    def _read_element(f, e, fmt, bo):
        has_list = any(p.is_list for p in e.properties)
        native = np.dtype([(p.name, "O" if p.is_list else p.val_dtype) for p in e.properties])
        if fmt != "ascii" and not has_list:
            disk = np.dtype([(p.name, bo + p.val_dtype) for p in e.properties])
            raw = f.read(disk.itemsize * e.count)
            if len(raw) != disk.itemsize * e.count:
                raise ValueError(f"PLY element {e.name}: file truncated")
            return np.frombuffer(raw, dtype=disk, count=e.count).astype(native)
        out = np.empty(e.count, dtype=native)
        for i in range(e.count):
            if fmt == "ascii":
                tok = f.readline().split()
                pos = 0
                for p in e.properties:
                    if p.is_list:
                        n = int(tok[pos]); pos += 1
                        out[p.name][i] = np.array(tok[pos:pos + n], dtype=np.float64).astype(p.val_dtype)
                        pos += n
                    else:
                        out[p.name][i] = np.array(tok[pos], dtype=np.float64).astype(p.val_dtype)
                        pos += 1
            else:
                for p in e.properties:
                    if p.is_list:
                        ld = np.dtype(bo + p.len_dtype)
                        n = int(np.frombuffer(f.read(ld.itemsize), dtype=ld)[0])
                        vd = np.dtype(bo + p.val_dtype)
                        out[p.name][i] = np.frombuffer(f.read(vd.itemsize * n), dtype=vd).astype(p.val_dtype)
                    else:
                        vd = np.dtype(bo + p.val_dtype)
                        out[p.name][i] = np.frombuffer(f.read(vd.itemsize), dtype=vd)[0]
        return out

    # ------------------------------------------------------------------ writing
    def header(self):
        fmt = "ascii" if self.text else ("binary_big_endian" if self.byte_order == ">" else "binary_little_endian")
        lines = ["ply", f"format {fmt} 1.0"] + [f"comment {c}" for c in self.comments]
        for e in self.elements:
            lines.append(f"element {e.name} {e.count}")
            for p in e.properties:
                if p.is_list:
                    lines.append(f"property list {_NP_TO_PLY[p.len_dtype]} {_NP_TO_PLY[p.val_dtype]} {p.name}")
                else:
                    lines.append(f"property {_NP_TO_PLY[p.val_dtype]} {p.name}")
        lines.append("end_header")
        return "\n".join(lines) + "\n"

    def write(self, stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            f.write(self.header().encode("ascii"))
            for e in self.elements:
                if any(p.is_list for p in e.properties):
                    raise NotImplementedError("writing list properties is not needed by the reference")
                if self.text:
                    for row in e.data:
                        f.write((" ".join(repr(v.item()) if np.issubdtype(type(v), np.floating) else str(v) for v in row) + "\n").encode("ascii"))
                else:
                    disk = np.dtype([(p.name, self.byte_order + p.val_dtype) for p in e.properties])
                    f.write(np.ascontiguousarray(e.data.astype(disk)).tobytes())
        finally:
            if own:
                f.close()
