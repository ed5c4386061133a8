"""On-disk formats around the hot path: Gmsh ``.msh`` field files written/read by the reference's
``mesh.MSHFieldWriter`` / ``mesh.MSHFieldParser3`` (utils.py:302-325, 411-417) and the ParaView ``.vtr`` rectilinear
grids its drivers write through ``pyevtk.gridToVTK`` (utils.py:350-376).  Host-side, numpy only.

The MeshFEM sources that define the reference's exact ``.msh`` dialect are not part of the reference checkout
(un-vendored submodule); the files written here are standard Gmsh 2.2 ASCII ($MeshFormat / $Nodes / $Elements /
$ElementData / $NodeData), which is the format family MeshFEM's MSHFieldWriter emits."""
import base64
import struct

import numpy as np

_GMSH_TYPE = {(4, 2): 3, (8, 3): 5, (3, 2): 2, (4, 3): 4}      # (nodes per element, dim) -> quad, hexahedron, triangle, tet


class MSHFieldWriter:
    """``mesh.MSHFieldWriter(path, V, F)`` then ``addField(name, values)``; scalar or vector fields per element
    (len == #elements) or per node (len == #nodes)."""

    def __init__(self, path, V, F, linearSubsample=True):
        self._path = str(path)
        V = np.asarray(V, dtype=np.float64)
        F = np.asarray(F, dtype=np.int64)
        if V.ndim != 2 or F.ndim != 2:
            raise RuntimeError("V must be #V x dim and F #F x nodesPerElement")
        self._nv, self._nf, self._dim = V.shape[0], F.shape[0], V.shape[1]
        key = (F.shape[1], 3 if (F.shape[1] == 8 or (F.shape[1] == 4 and V.shape[1] == 3 and False)) else 2)
        if F.shape[1] == 8:
            key = (8, 3)
        etype = _GMSH_TYPE.get(key)
        if etype is None:
            raise RuntimeError("unsupported element with %d nodes" % F.shape[1])
        V3 = np.zeros((V.shape[0], 3))
        V3[:, :V.shape[1]] = V
        with open(self._path, "w") as fh:
            fh.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % self._nv)
            for i, p in enumerate(V3):
                fh.write("%d %.17g %.17g %.17g\n" % (i + 1, p[0], p[1], p[2]))
            fh.write("$EndNodes\n$Elements\n%d\n" % self._nf)
            for i, e in enumerate(F):
                fh.write("%d %d 2 0 0 %s\n" % (i + 1, etype, " ".join(str(int(n) + 1) for n in e)))
            fh.write("$EndElements\n")

    def addField(self, name, values, domain=None):
        a = np.asarray(values, dtype=np.float64)
        if a.ndim == 1:
            a = a[:, None]
        if domain is None:
            domain = "element" if a.shape[0] == self._nf else ("node" if a.shape[0] == self._nv else None)
        if domain is None or a.shape[0] != (self._nf if domain == "element" else self._nv):
            raise RuntimeError("Invalid field size")
        ncomp = {1: 1, 2: 3, 3: 3, 9: 9}.get(a.shape[1])
        if ncomp is None:
            raise RuntimeError("fields must be scalar, vector or tensor valued")
        out = np.zeros((a.shape[0], ncomp))
        out[:, :a.shape[1]] = a
        sec = "ElementData" if domain == "element" else "NodeData"
        with open(self._path, "a") as fh:
            fh.write("$%s\n1\n\"%s\"\n1\n0.0\n3\n0\n%d\n%d\n" % (sec, name, ncomp, a.shape[0]))
            for i, row in enumerate(out):
                fh.write("%d %s\n" % (i + 1, " ".join("%.17g" % v for v in row)))
            fh.write("$End%s\n" % sec)


class MSHFieldParser3:
    """``mesh.MSHFieldParser3(mshPath=...)``: ``scalarField(name)``, ``vectorField(name)``, ``vertices()``, ``elements()``."""

    def __init__(self, mshPath):
        self._fields = {}
        self._V, self._F = None, None
        with open(mshPath) as fh:
            lines = [l.rstrip("\n") for l in fh]
        i = 0
        while i < len(lines):
            tag = lines[i].strip()
            if tag == "$Nodes":
                n = int(lines[i + 1])
                self._V = np.array([[float(v) for v in lines[i + 2 + k].split()[1:4]] for k in range(n)])
                i += n + 3
            elif tag == "$Elements":
                n = int(lines[i + 1])
                F = []
                for k in range(n):
                    t = lines[i + 2 + k].split()
                    ntags = int(t[2])
                    F.append([int(v) - 1 for v in t[3 + ntags:]])
                self._F = np.array(F, dtype=np.int64)
                i += n + 3
            elif tag in ("$ElementData", "$NodeData"):
                j = i + 1
                ns = int(lines[j]); names = [lines[j + 1 + k].strip().strip('"') for k in range(ns)]; j += 1 + ns
                nr = int(lines[j]); j += 1 + nr
                ni = int(lines[j]); ints = [int(lines[j + 1 + k]) for k in range(ni)]; j += 1 + ni
                ncomp, cnt = ints[1], ints[2]
                data = np.array([[float(v) for v in lines[j + k].split()[1:1 + ncomp]] for k in range(cnt)])
                self._fields[names[0]] = (tag[1:], data)
                i = j + cnt + 1
            else:
                i += 1

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
    """``pyevtk.hl.gridToVTK(path, x, y, z, cellData=..., pointData=...)`` for rectilinear grids: writes ``path + '.vtr'``
    (VTK XML RectilinearGrid, base64-encoded inline binary, little endian, Fortran (x fastest) ordering) and returns the
    file name, as pyevtk does."""
    x, y, z = (np.asarray(a, dtype=np.float64) for a in (x, y, z))
    nx, ny, nz = x.size - 1, y.size - 1, z.size - 1
    ext = "0 %d 0 %d 0 %d" % (nx, ny, nz)

    def enc(a):
        raw = np.ascontiguousarray(a).tobytes()
        return base64.b64encode(struct.pack("<I", len(raw)) + raw).decode("ascii")

    def arr(name, a, ncomp=1):
        a = np.asarray(a)
        vt = {"float64": "Float64", "float32": "Float32", "int32": "Int32", "int64": "Int64"}.get(str(a.dtype))
        if vt is None:
            a, vt = a.astype(np.float64), "Float64"
        return '<DataArray type="%s" Name="%s" NumberOfComponents="%d" format="binary">%s</DataArray>\n' % (vt, name, ncomp, enc(a))

    fname = str(path) + ".vtr"
    with open(fname, "w") as fh:
        fh.write('<?xml version="1.0"?>\n<VTKFile type="RectilinearGrid" version="0.1" byte_order="LittleEndian">\n')
        fh.write('<RectilinearGrid WholeExtent="%s">\n<Piece Extent="%s">\n' % (ext, ext))
        for tag, data, shape in (("PointData", pointData, (nx + 1, ny + 1, nz + 1)), ("CellData", cellData, (nx, ny, nz))):
            fh.write("<%s>\n" % tag)
            for name, a in (data or {}).items():
                a = np.asarray(a)
                if a.shape != shape:
                    raise RuntimeError("%s array '%s' has shape %s, expected %s" % (tag, name, a.shape, shape))
                fh.write(arr(name, a.ravel(order="F")))
            fh.write("</%s>\n" % tag)
        fh.write("<Coordinates>\n" + arr("x_coordinates", x) + arr("y_coordinates", y) + arr("z_coordinates", z) + "</Coordinates>\n")
        fh.write("</Piece>\n</RectilinearGrid>\n</VTKFile>\n")
    return fname


def read_vtr(fname):
    """inverse of grid_to_vtr (tests and round trips): returns (x, y, z, cellData, pointData)"""
    import re
    txt = open(fname).read()
    ext = [int(v) for v in re.search(r'Piece Extent="([^"]+)"', txt).group(1).split()]
    nx, ny, nz = ext[1], ext[3], ext[5]
    np_t = {"Float64": np.float64, "Float32": np.float32, "Int32": np.int32, "Int64": np.int64}

    def section(tag):
        m = re.search(r"<%s>(.*?)</%s>" % (tag, tag), txt, re.S)
        out = {}
        for t, name, b64 in re.findall(r'<DataArray type="(\w+)" Name="([^"]+)"[^>]*>([^<]*)</DataArray>', m.group(1) if m else ""):
            raw = base64.b64decode(b64)
            out[name] = np.frombuffer(raw[4:4 + struct.unpack("<I", raw[:4])[0]], dtype=np_t[t]).copy()
        return out

    c = section("Coordinates")
    cd = {k: v.reshape((nx, ny, nz), order="F") for k, v in section("CellData").items()}
    pd = {k: v.reshape((nx + 1, ny + 1, nz + 1), order="F") for k, v in section("PointData").items()}
    return c["x_coordinates"], c["y_coordinates"], c["z_coordinates"], cd, pd
