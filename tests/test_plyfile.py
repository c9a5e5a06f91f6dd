"""CPU: the `plyfile` stand-in (gaussian-mesh-splatting_amd/games_hip/_plyfile.py) on the reference's own call patterns
(scene/gaussian_model.py:194-268 save/load of a Gaussian cloud, scene/dataset_readers.py:107-130 fetchPly/storePly)
and against hand-built PLY bytes (the format is the published PLY 1.0 layout)."""
import io
import struct

import numpy as np

from games_hip._plyfile import PlyData, PlyElement


def test_gaussian_cloud_roundtrip_with_the_reference_attribute_layout(tmp_path):
    P, rest = 37, 45
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(rest)]
             + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    attrs = np.random.default_rng(0).normal(size=(P, len(names))).astype(np.float32)
    elements = np.empty(P, dtype=[(n, "f4") for n in names])
    elements[:] = list(map(tuple, attrs))                       # scene/gaussian_model.py:213-214
    path = tmp_path / "point_cloud.ply"
    PlyData([PlyElement.describe(elements, "vertex")]).write(str(path))
    ply = PlyData.read(str(path))
    el = ply.elements[0]
    assert el.name == "vertex" and el.count == P and [p.name for p in el.properties] == names
    xyz = np.stack((np.asarray(el["x"]), np.asarray(el["y"]), np.asarray(el["z"])), axis=1)    # :229-231
    assert np.array_equal(xyz, attrs[:, :3])
    extra = sorted([p.name for p in el.properties if p.name.startswith("f_rest_")], key=lambda x: int(x.split("_")[-1]))
    assert len(extra) == rest and np.array_equal(np.asarray(el["f_rest_44"]), attrs[:, names.index("f_rest_44")])
    assert np.array_equal(np.asarray(ply["vertex"]["rot_3"]), attrs[:, -1])


def test_binary_layout_is_the_ply_specification(tmp_path):
    data = np.array([(1.5, -2.0, 3.25, 255, 0, 7)], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    buf = io.BytesIO()
    PlyData([PlyElement.describe(data, "vertex")]).write(buf)
    raw = buf.getvalue()
    head, body = raw.split(b"end_header\n")
    assert head.decode().splitlines() == ["ply", "format binary_little_endian 1.0", "element vertex 1", "property float x",
                                          "property float y", "property float z", "property uchar red",
                                          "property uchar green", "property uchar blue"]
    assert body == struct.pack("<fffBBB", 1.5, -2.0, 3.25, 255, 0, 7)


def test_reads_ascii_big_endian_and_face_lists():
    ascii_ply = b"ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n" \
                b"element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0.5\n3 0 1 2\n"
    ply = PlyData.read(io.BytesIO(ascii_ply))
    assert ply.text and ply["vertex"].count == 3 and float(ply["vertex"]["z"][2]) == 0.5
    assert list(ply["face"]["vertex_indices"][0]) == [0, 1, 2]
    be = b"ply\nformat binary_big_endian 1.0\nelement vertex 2\nproperty double x\nproperty short k\nend_header\n" + \
         struct.pack(">dh", 0.125, -3) + struct.pack(">dh", 8.0, 300)
    ply = PlyData.read(io.BytesIO(be))
    assert list(ply["vertex"]["x"]) == [0.125, 8.0] and list(ply["vertex"]["k"]) == [-3, 300]
    le_faces = b"ply\nformat binary_little_endian 1.0\nelement face 2\nproperty list uchar uint vertex_indices\nend_header\n" + \
               struct.pack("<BIII", 3, 5, 6, 7) + struct.pack("<BIIII", 4, 1, 2, 3, 4)
    f = PlyData.read(io.BytesIO(le_faces))["face"]["vertex_indices"]
    assert list(f[0]) == [5, 6, 7] and list(f[1]) == [1, 2, 3, 4]


def test_store_and_fetch_ply_pattern(tmp_path):
    """scene/dataset_readers.py:117-130 storePly then :107-115 fetchPly."""
    rng = np.random.default_rng(1)
    xyz, rgb = rng.normal(size=(20, 3)), rng.integers(0, 256, size=(20, 3))
    dtype = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")]
    elements = np.empty(20, dtype=dtype)
    elements[:] = list(map(tuple, np.concatenate((xyz, np.zeros_like(xyz), rgb), axis=1)))
    path = str(tmp_path / "points3d.ply")
    PlyData([PlyElement.describe(elements, "vertex")]).write(path)
    v = PlyData.read(path)["vertex"]
    pos = np.vstack([v["x"], v["y"], v["z"]]).T
    col = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    assert np.allclose(pos, xyz.astype(np.float32)) and np.array_equal((col * 255).round().astype(int), rgb)
