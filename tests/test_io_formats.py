"""On-disk formats around the path (host only): .msh field files and .vtr grids -- byte-level goldens spelled out from the
format rules (MeshFEM's MeshIO.cc:533-614 + MSHFieldWriter.hh:128-205 for .msh, the VTK XML appended-raw layout for .vtr),
then round trips."""
import struct
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _hex_grid(ne):
    nn = np.array(ne) + 1
    idx = np.stack(np.meshgrid(*[np.arange(n) for n in nn], indexing="ij"), -1).reshape(-1, 3)
    V = idx * np.array([0.5, 0.25, 1.0])
    eidx = np.stack(np.meshgrid(*[np.arange(n) for n in ne], indexing="ij"), -1).reshape(-1, 3)
    nstr = np.array([nn[1] * nn[2], nn[2], 1])
    order = [0, 1, 3, 2, 4, 5, 7, 6]
    loc = [np.array([(m >> 2) & 1, (m >> 1) & 1, m & 1]) @ nstr for m in range(8)]
    F = np.stack([eidx @ nstr + loc[m] for m in order], axis=1)
    return V, F


def test_msh_bytes_follow_the_meshfem_dialect(tmp_path):
    """one hexahedron, one element field, one node vector field: every byte written out by hand from the writer's rules"""
    from ndr_amd import io
    V = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 2], [1, 0, 2], [1, 1, 2], [0, 1, 2]], dtype=float)
    F = np.array([[0, 1, 2, 3, 4, 5, 6, 7]])
    p = str(tmp_path / "one.msh")
    w = io.MSHFieldWriter(p, V, F)                                   # binary is the default (MSHFieldWriter_bindings.cc:22)
    w.addField("density", np.array([0.25]))
    w.addField("u", np.arange(16, dtype=float).reshape(8, 2))        # 2-vectors are padded to 3 components
    want = b"$MeshFormat\n2.2 1 8\n" + struct.pack("<i", 1) + b"\n$EndMeshFormat\n$Nodes\n8\n"
    for i, pt in enumerate(V):
        want += struct.pack("<i3d", i + 1, *pt)
    want += b"\n$EndNodes\n$Elements\n1\n" + struct.pack("<iii", 5, 1, 0) + struct.pack("<9i", 1, 1, 2, 3, 4, 5, 6, 7, 8) + b"\n$EndElements\n"
    want += b'$ElementData\n1\n"density"\n0\n3\n0\n1\n1\n' + struct.pack("<id", 1, 0.25) + b"$EndElementData\n"
    want += b'$NodeData\n1\n"u"\n0\n3\n0\n3\n8\n'
    for i in range(8):
        want += struct.pack("<i3d", i + 1, 2.0 * i, 2.0 * i + 1, 0.0)
    want += b"$EndNodeData\n"
    assert open(p, "rb").read() == want
    r = io.MSHFieldParser3(p)
    assert np.array_equal(r.vertices(), V) and np.array_equal(r.elements(), F) and r.scalarField("density")[0] == 0.25
    # ASCII form: same sections as text, 17 significant digits (the precision set for the node list stays on the stream)
    q = str(tmp_path / "one_ascii.msh")
    w = io.MSHFieldWriter(q, V[:4, :2], np.array([[0, 1, 2, 3]]), binary=False)
    w.addField("rho", np.array([1.0 / 3.0]))
    want = ("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n4\n1 0 0 0\n2 1 0 0\n3 1 1 0\n4 0 1 0\n$EndNodes\n$Elements\n1\n"
            "1 4 0 1 2 3 4\n$EndElements\n"          # a 4-node element gets type 4: first 4-node row of MeshIO.cc:527-531
            '$ElementData\n1\n"rho"\n0\n3\n0\n1\n1\n1 0.33333333333333331\n$EndElementData\n')
    assert open(q).read() == want
    assert io.MSHFieldParser3(q).scalarField("rho")[0] == 1.0 / 3.0


def test_vtr_bytes_follow_the_vtk_appended_raw_layout(tmp_path):
    from ndr_amd import io
    d = np.array([[[1.5]], [[-2.0]]])                                # 2 x 1 x 1 cells
    f = io.grid_to_vtr(str(tmp_path / "g"), np.array([0.0, 1.0, 2.0]), np.array([0.0, 1.0]), np.array([0.0, 4.0]), cellData={"data": d})
    head = ('<?xml version="1.0"?>\n<VTKFile type="RectilinearGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">\n'
            '<RectilinearGrid WholeExtent="0 2 0 1 0 1">\n<Piece Extent="0 2 0 1 0 1">\n<PointData>\n</PointData>\n<CellData>\n'
            '<DataArray Name="data" NumberOfComponents="1" type="Float64" format="appended" offset="0"/>\n</CellData>\n<Coordinates>\n'
            '<DataArray Name="x_coordinates" NumberOfComponents="1" type="Float64" format="appended" offset="24"/>\n'
            '<DataArray Name="y_coordinates" NumberOfComponents="1" type="Float64" format="appended" offset="56"/>\n'
            '<DataArray Name="z_coordinates" NumberOfComponents="1" type="Float64" format="appended" offset="80"/>\n'
            '</Coordinates>\n</Piece>\n</RectilinearGrid>\n<AppendedData encoding="raw">_')
    blob = (struct.pack("<Q2d", 16, 1.5, -2.0) + struct.pack("<Q3d", 24, 0.0, 1.0, 2.0) + struct.pack("<Q2d", 16, 0.0, 1.0) +
            struct.pack("<Q2d", 16, 0.0, 4.0))
    assert open(f, "rb").read() == head.encode() + blob + b"</AppendedData>\n</VTKFile>\n"


def test_msh_field_round_trip(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "VoxelFEM", "python"))
    import mesh
    ne = (3, 2, 4)
    V, F = _hex_grid(ne)
    rho = np.random.default_rng(0).uniform(size=F.shape[0])
    u = np.random.default_rng(1).standard_normal((V.shape[0], 3))
    p = str(tmp_path / "d.msh")
    w = mesh.MSHFieldWriter(p, V, F)
    w.addField("density", rho)
    w.addField("u", u)
    r = mesh.MSHFieldParser3(mshPath=p)
    assert np.array_equal(r.scalarField("density"), rho)          # binary by default: exact
    assert np.array_equal(r.vectorField("u"), u)
    assert np.array_equal(r.vertices(), V) and np.array_equal(r.elements(), F)
    with pytest.raises(RuntimeError):
        r.scalarField("nope")
    with pytest.raises(RuntimeError):
        w.addField("bad", np.zeros(5))


def test_vtr_round_trip(tmp_path):
    from ndr_amd import io
    d = np.random.default_rng(2).uniform(size=(4, 3, 5))
    f = io.grid_to_vtr(str(tmp_path / "g"), np.arange(5), np.arange(4), np.arange(6), cellData={"data": d})
    assert f.endswith(".vtr")
    x, y, z, cd, pd = io.read_vtr(f)
    assert np.array_equal(cd["data"], d) and x.size == 5 and z.size == 6 and pd == {}
    with pytest.raises(RuntimeError):
        io.grid_to_vtr(str(tmp_path / "h"), np.arange(5), np.arange(4), np.arange(6), cellData={"data": d.T})


def test_side_module_shims_import():
    sys.path.insert(0, os.path.join(ROOT, "VoxelFEM", "python"))
    import parallelism
    parallelism.set_max_num_tbb_threads(8)
    parallelism.set_gradient_assembly_num_threads(4)
    parallelism.unset_max_num_tbb_threads()
    with pytest.raises(RuntimeError):
        parallelism.set_max_num_tbb_threads(0)
