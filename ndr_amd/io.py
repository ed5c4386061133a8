"""On-disk formats around the hot path: Gmsh ``.msh`` field files as written / read by the reference's
``mesh.MSHFieldWriter`` / ``mesh.MSHFieldParser3`` (utils.py:302-325, 411-417) and the ParaView ``.vtr`` rectilinear grids its
drivers write through ``pyevtk.gridToVTK`` (utils.py:350-376).  Host-side, numpy only.

The ``.msh`` dialect is MeshFEM's (paths relative to VoxelFEM/3rdparty/MeshFEM/src/lib/MeshFEM): the mesh block of
``MeshIO_MSH::save`` (MeshIO.cc:533-614) followed by one ``$ElementData`` / ``$NodeData`` block per field
(MSHFieldWriter.hh:128-205).  Format 2.2, BINARY by default (python_bindings/MSHFieldWriter_bindings.cc:19-23):
  $MeshFormat / "2.2 1 8" / int32 1 + newline / $EndMeshFormat
  $Nodes / count / per node: int32 index (1-based) + 3 float64, then one newline / $EndNodes
  $Elements / count / int32 element type, int32 count, int32 0 (no tags), per element: int32 index + int32 node ids
      (1-based), then one newline / $EndElements
  $ElementData / 1 / "name" / 0 (no real tags) / 3 / 0 / components / count / per entry: int32 index + float64 values,
      immediately followed by $EndElementData
The ASCII form (binary=False) writes the same sections as text, coordinates and values with 17 significant digits
(MeshIO.cc:570: the precision set for the node list stays on the stream).  The element type is chosen by the node count of the
LAST element from MeshIO.cc:527-531 in table order, so a 4-node element is written as type 4 (the table lists the
tetrahedron before the quadrilateral) -- reproduced as is: the reference's own parser maps it back the same way."""
import base64
import struct

import numpy as np

_ELEMENT_TYPE_FOR_NODE_COUNT = {3: 2, 4: 4, 8: 5, 6: 9, 10: 11, 2: 1}      # first table entry with that node count, MeshIO.cc:527-531
_NODE_COUNT_FOR_ELEMENT_TYPE = {2: 3, 4: 4, 3: 4, 5: 8, 9: 6, 11: 10, 1: 2, 8: 3}


class MSHFieldWriter:
    """``mesh.MSHFieldWriter(path, V, F, binary=True)`` then ``addField(name, values)``; scalar / vector fields per element
    (rows == #elements) or per node (rows == #nodes); 2-vectors are padded to 3 components (MSHFieldWriter.hh:147-151)."""

    def __init__(self, path, V, F, binary=True):
        self._path, self._binary = str(path), bool(binary)
        V = np.asarray(V, dtype=np.float64)
        F = np.asarray(F, dtype=np.int64)
        if V.ndim != 2 or F.ndim != 2:
            raise RuntimeError("V must be #V x dim and F #F x nodesPerElement")
        if V.shape[0] == 0:
            raise RuntimeError("Empty mesh.")
        self._nv, self._nf = V.shape[0], F.shape[0]
        etype = _ELEMENT_TYPE_FOR_NODE_COUNT.get(F.shape[1])
        if etype is None:
            raise RuntimeError("Unsupported node count for MSH I/O")
        V3 = np.zeros((V.shape[0], 3))
        V3[:, :V.shape[1]] = V
        with open(self._path, "wb") as fh:
            fh.write(b"$MeshFormat\n2.2 %d 8\n" % (1 if self._binary else 0))
            if self._binary:
                fh.write(struct.pack("<i", 1) + b"\n")
            fh.write(b"$EndMeshFormat\n$Nodes\n%d\n" % self._nv)
            if self._binary:
                rec = np.zeros(self._nv, dtype=[("i", "<i4"), ("p", "<f8", 3)])
                rec["i"] = np.arange(1, self._nv + 1)
                rec["p"] = V3
                fh.write(rec.tobytes() + b"\n")
            else:
                for i, p in enumerate(V3):
                    fh.write(("%d %.17g %.17g %.17g\n" % (i + 1, p[0], p[1], p[2])).encode())
            fh.write(b"$EndNodes\n$Elements\n%d\n" % self._nf)
            if self._binary:
                if self._nf > 0:
                    fh.write(struct.pack("<iii", etype, self._nf, 0))
                rec = np.empty((self._nf, 1 + F.shape[1]), dtype="<i4")
                rec[:, 0] = np.arange(1, self._nf + 1)
                rec[:, 1:] = F + 1
                fh.write(rec.tobytes() + b"\n")
            else:
                for i, e in enumerate(F):
                    fh.write(("%d %d 0 %s\n" % (i + 1, etype, " ".join(str(int(n) + 1) for n in e))).encode())
            fh.write(b"$EndElements\n")

    def addField(self, name, values, domain=None):
        a = np.asarray(values, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        if domain is None:                                       # DomainType::GUESS, MSHFieldWriter.hh:332-338
            domain = "element" if a.shape[0] == self._nf else ("node" if a.shape[0] == self._nv else None)
        if domain is None or a.shape[0] != (self._nf if domain == "element" else self._nv):
            raise RuntimeError("Invalid field domain size.")
        ncomp = {1: 1, 2: 3, 3: 3}.get(a.shape[1])
        if ncomp is None:
            raise RuntimeError("Invalid field dimension.")
        out = np.zeros((a.shape[0], ncomp))
        out[:, :a.shape[1]] = a
        sec = "ElementData" if domain == "element" else "NodeData"
        with open(self._path, "ab") as fh:
            fh.write(("$%s\n1\n\"%s\"\n0\n3\n0\n%d\n%d\n" % (sec, name, ncomp, a.shape[0])).encode())
            if self._binary:
                rec = np.zeros(a.shape[0], dtype=[("i", "<i4"), ("v", "<f8", ncomp)])
                rec["i"] = np.arange(1, a.shape[0] + 1)
                rec["v"] = out.reshape(a.shape[0], ncomp) if ncomp > 1 else out[:, 0:1].reshape(a.shape[0], 1)
                fh.write(rec.tobytes())
            else:
                for i, row in enumerate(out):
                    fh.write(("%d %s\n" % (i + 1, " ".join("%.17g" % v for v in row))).encode())
            fh.write(("$End%s\n" % sec).encode())


class MSHFieldParser3:
    """``mesh.MSHFieldParser3(mshPath=...)``: ``scalarField(name)``, ``vectorField(name)``, ``vertices()``, ``elements()``;
    reads the binary and the ASCII form (MeshIO.cc:625-760, MSHFieldParser.cc:170-260)."""

    def __init__(self, mshPath):
        self._fields = {}
        data = open(mshPath, "rb").read()
        pos = 0

        def line():
            nonlocal pos
            while pos < len(data) and data[pos:pos + 1] in (b"\n", b"\r", b" ", b"\t"):
                pos += 1
            end = data.index(b"\n", pos)
            out = data[pos:end].decode().strip()
            pos = end + 1
            return out

        if line() != "$MeshFormat":
            raise RuntimeError("Bad MSH file format")
        version, ftype, dsize = line().split()
        if int(ftype) > 1 or int(dsize) != 8:
            raise RuntimeError("Unsupported MSH file format")
        binary = int(ftype) == 1
        if binary:
            if struct.unpack_from("<i", data, pos)[0] != 1:
                raise RuntimeError("Unsupported MSH file format")
            pos += 5
        if line() != "$EndMeshFormat" or line() != "$Nodes":
            raise RuntimeError("Bad MSH file format")
        n = int(line())
        if binary:
            rec = np.frombuffer(data, dtype=[("i", "<i4"), ("p", "<f8", 3)], count=n, offset=pos)
            self._V = np.array(rec["p"])
            pos += rec.nbytes
        else:
            self._V = np.array([[float(v) for v in line().split()[1:4]] for _ in range(n)]).reshape(n, 3)
        if line() != "$EndNodes" or line() != "$Elements":
            raise RuntimeError("Bad MSH file format")
        ne = int(line())
        if binary:
            if ne > 0:
                etype, cnt, ntags = struct.unpack_from("<iii", data, pos)
                pos += 12
                npe = _NODE_COUNT_FOR_ELEMENT_TYPE[etype]
                rec = np.frombuffer(data, dtype="<i4", count=ne * (1 + ntags + npe), offset=pos).reshape(ne, 1 + ntags + npe)
                self._F = np.array(rec[:, 1 + ntags:], dtype=np.int64) - 1
                pos += rec.nbytes
            else:
                self._F = np.zeros((0, 0), dtype=np.int64)
        else:
            F = []
            for _ in range(ne):
                t = line().split()
                F.append([int(v) - 1 for v in t[3 + int(t[2]):]])
            self._F = np.array(F, dtype=np.int64)
        if line() != "$EndElements":
            raise RuntimeError("Bad MSH file format")
        while True:
            while pos < len(data) and data[pos:pos + 1] in (b"\n", b"\r", b" ", b"\t"):
                pos += 1
            if pos >= len(data):
                break
            tag = line()
            if tag not in ("$ElementData", "$NodeData"):
                raise RuntimeError("Unrecognized MSH section: " + tag)
            if int(line()) != 1:
                raise RuntimeError("Bad MSH field format")
            name = line().strip('"')
            for _ in range(int(line())):
                line()                                           # real tags are discarded
            if int(line()) != 3:
                raise RuntimeError("Bad MSH field format")
            line()
            ncomp, cnt = int(line()), int(line())
            if cnt != (ne if tag == "$ElementData" else n):
                raise RuntimeError("Illegal number of field values")
            if binary:
                rec = np.frombuffer(data, dtype=[("i", "<i4"), ("v", "<f8", ncomp)], count=cnt, offset=pos)
                vals = np.array(rec["v"]).reshape(cnt, ncomp)
                pos += rec.nbytes
            else:
                vals = np.array([[float(v) for v in line().split()[1:1 + ncomp]] for _ in range(cnt)]).reshape(cnt, ncomp)
            if line() != "$End" + tag[1:]:
                raise RuntimeError("Bad MSH field format")
            self._fields[name] = (tag[1:], vals)

    def vertices(self):
        return self._V

    def elements(self):
        return self._F

    def fieldNames(self):
        return list(self._fields)

    def _get(self, name):
        if name not in self._fields:
            raise RuntimeError("Field '%s' not found" % name)
        return self._fields[name][1]

    def scalarField(self, name):
        d = self._get(name)
        if d.shape[1] != 1:
            raise RuntimeError("Field '%s' is not scalar valued" % name)
        return d[:, 0].copy()

    def vectorField(self, name):
        return self._get(name).copy()


def grid_to_vtr(path, x, y, z, cellData=None, pointData=None):
    """``pyevtk.hl.gridToVTK(path, x, y, z, cellData=..., pointData=...)`` for rectilinear grids: writes ``path + '.vtr'`` and
    returns the file name, as pyevtk does.  VTK XML RectilinearGrid, version 1.0, the layout pyevtk emits: every DataArray
    is ``format="appended"`` with a byte offset into one ``<AppendedData encoding="raw">`` section, where each array is a
    UInt64 byte count followed by its little-endian values in Fortran order (x fastest)."""
    x, y, z = (np.asarray(a, dtype=np.float64) for a in (x, y, z))
    nx, ny, nz = x.size - 1, y.size - 1, z.size - 1
    ext = "0 %d 0 %d 0 %d" % (nx, ny, nz)
    blobs, offset = [], 0

    def arr(name, a, ncomp=1):
        nonlocal offset
        a = np.asarray(a)
        vt = {"float64": "Float64", "float32": "Float32", "int32": "Int32", "int64": "Int64"}.get(str(a.dtype))
        if vt is None:
            a, vt = a.astype(np.float64), "Float64"
        raw = np.ascontiguousarray(a).astype(a.dtype.newbyteorder("<")).tobytes()
        line = '<DataArray Name="%s" NumberOfComponents="%d" type="%s" format="appended" offset="%d"/>\n' % (name, ncomp, vt, offset)
        blobs.append(struct.pack("<Q", len(raw)) + raw)
        offset += 8 + len(raw)
        return line

    fname = str(path) + ".vtr"
    head = ['<?xml version="1.0"?>\n<VTKFile type="RectilinearGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">\n',
            '<RectilinearGrid WholeExtent="%s">\n<Piece Extent="%s">\n' % (ext, ext)]
    for tag, data, shape in (("PointData", pointData, (nx + 1, ny + 1, nz + 1)), ("CellData", cellData, (nx, ny, nz))):
        head.append("<%s>\n" % tag)
        for name, a in (data or {}).items():
            a = np.asarray(a)
            if a.shape != shape:
                raise RuntimeError("%s array '%s' has shape %s, expected %s" % (tag, name, a.shape, shape))
            head.append(arr(name, a.ravel(order="F")))
        head.append("</%s>\n" % tag)
    head.append("<Coordinates>\n" + arr("x_coordinates", x) + arr("y_coordinates", y) + arr("z_coordinates", z) + "</Coordinates>\n")
    head.append('</Piece>\n</RectilinearGrid>\n<AppendedData encoding="raw">_')
    with open(fname, "wb") as fh:
        fh.write("".join(head).encode("ascii"))
        for bl in blobs:
            fh.write(bl)
        fh.write(b"</AppendedData>\n</VTKFile>\n")
    return fname


def read_vtr(fname):
    """inverse of grid_to_vtr: returns (x, y, z, cellData, pointData)"""
    import re
    data = open(fname, "rb").read()
    cut = data.index(b'<AppendedData encoding="raw">_') + len(b'<AppendedData encoding="raw">_')
    txt, blob = data[:cut].decode("ascii"), data[cut:]
    ext = [int(v) for v in re.search(r'Piece Extent="([^"]+)"', txt).group(1).split()]
    nx, ny, nz = ext[1], ext[3], ext[5]
    np_t = {"Float64": "<f8", "Float32": "<f4", "Int32": "<i4", "Int64": "<i8"}

    def section(tag):
        m = re.search(r"<%s>(.*?)</%s>" % (tag, tag), txt, re.S)
        out = {}
        for name, t, off in re.findall(r'<DataArray Name="([^"]+)" NumberOfComponents="\d+" type="(\w+)" format="appended" offset="(\d+)"/>',
                                       m.group(1) if m else ""):
            off = int(off)
            nbytes = struct.unpack_from("<Q", blob, off)[0]
            out[name] = np.frombuffer(blob, dtype=np_t[t], count=nbytes // np.dtype(np_t[t]).itemsize, offset=off + 8).copy()
        return out

    c = section("Coordinates")
    cd = {k: v.reshape((nx, ny, nz), order="F") for k, v in section("CellData").items()}
    pd = {k: v.reshape((nx + 1, ny + 1, nz + 1), order="F") for k, v in section("PointData").items()}
    return c["x_coordinates"], c["y_coordinates"], c["z_coordinates"], cd, pd
