"""CPU: OBJ reader / writer and the mesh point-cloud construction (games_hip/io_mesh.py) on the reference's call patterns
(games/mesh_splatting/scene/dataset_readers.py:40-105)."""
import numpy as np
import pytest
import torch

from games_hip import io_mesh, synthetic as syn


def test_obj_reader_handles_polygons_triplets_and_relative_indices(tmp_path):
    p = tmp_path / "mesh.obj"
    p.write_text("# comment\nmtllib x.mtl\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
                 "f 1/1/1 2/1/1 3/1/1 4/1/1\n"          # quad with v/vt/vn triplets -> two triangles (fan)
                 "v 0.5 0.5 1e0\n"
                 "f -1 1 2\n"                           # relative index: the vertex just read
                 "f 3//1 4//1 5//1\n")
    m = io_mesh.load_obj(str(p))
    assert m.vertices.shape == (5, 3) and m.vertices.dtype == np.float64
    assert m.faces.tolist() == [[0, 1, 2], [0, 2, 3], [4, 0, 1], [2, 3, 4]]
    assert m.triangles.shape == (4, 3, 3)


def test_obj_roundtrip_and_point_cloud_fields(tmp_path):
    v, f = syn.uv_sphere(6, 7)
    path = str(tmp_path / "mesh.obj")
    io_mesh.save_obj(path, v, f)
    m = io_mesh.load_obj(path)
    assert np.array_equal(m.faces, f.numpy()) and np.allclose(m.vertices, v.numpy(), rtol=0, atol=1e-7)
    pcd = io_mesh.mesh_point_cloud(m, num_splats=3, seed=1)
    F = f.shape[0]
    assert pcd.alpha.shape == (F, 3, 3) and pcd.points.shape == (3 * F, 3) and pcd.triangles.shape == (F, 3, 3)
    assert pcd.colors.shape == (3 * F, 3) and pcd.normals.shape == (3 * F, 3)
    # dataset_readers.py:31-37: (x, y, z) -> (x, -z, y)
    vt = io_mesh.transform_vertices_function(torch.tensor(m.vertices))
    assert torch.allclose(vt[:, 0], torch.tensor(m.vertices[:, 0])) and torch.allclose(vt[:, 1], -torch.tensor(m.vertices[:, 2]))
    assert torch.allclose(pcd.points, torch.matmul(pcd.alpha, pcd.triangles).reshape(-1, 3))


def test_point_cloud_drives_the_reference_create_from_pcd(tmp_path, monkeypatch):
    """The stand-in MeshPointCloud has what the reference's GaussianMeshModel.create_from_pcd reads
    (gaussian_mesh_model.py:53-84), and the reference's own transform_vertices_function agrees with ours."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.import_reference()
    import games
    from games_hip import model as hip_model
    import test_abi
    monkeypatch.setattr(hip_model, "mesh_to_gaussians", test_abi._cpu_op)
    try:
        import games.mesh_splatting.scene.dataset_readers as rdr
        v = torch.rand(11, 3, dtype=torch.float64)
        assert torch.equal(rdr.transform_vertices_function(v.clone()), io_mesh.transform_vertices_function(v.clone()))
    except Exception as e:          # the reader module needs more of the absent third-party stack than the stubs provide
        print("reference dataset_readers not importable here:", repr(e))
    v, f = syn.uv_sphere(5, 6)
    path = str(tmp_path / "mesh.obj")
    io_mesh.save_obj(path, v, f)
    pcd = io_mesh.mesh_point_cloud(io_mesh.load_obj(path), num_splats=2, seed=0)
    out = hip_model.install(games)
    try:
        with ref_import.cuda_literals_on_cpu():
            m = games.gaussianModel["gs_mesh"](3)
            m.create_from_pcd(pcd, 1.0)
        assert m.get_xyz.shape == (2 * f.shape[0], 3) and m._features_rest.shape == (2 * f.shape[0], 15, 3)
        assert torch.allclose(m.get_xyz.detach(), torch.matmul(m.alpha.detach(), pcd.triangles).reshape(-1, 3), atol=1e-6)
    finally:
        hip_model.uninstall(games, out)
        ref_import.drop_reference_stubs()


def _flame_model_on_cpu(monkeypatch):
    from games_hip import model as hip_model
    import test_abi
    monkeypatch.setattr(hip_model, "mesh_to_gaussians", test_abi._cpu_op)       # CPU stand-in for the HIP op (test only)
    return hip_model.HipGaussianFlameModel.from_scene(syn.mesh_scene("tiny"), "cpu", enlargement=1.1)


def test_flame_params_roundtrip_in_the_reference_file_layout(tmp_path, monkeypatch):
    """`HipGaussianFlameModel.save_ply` / `load_ply`: point_cloud.ply + flame_params.pt with the reference's nine keys
    (games/flame_splatting/scene/gaussian_flame_model.py:232-265) plus the raw `_alpha` / `_scales`."""
    from games_hip import model as hip_model
    m = _flame_model_on_cpu(monkeypatch)
    with torch.no_grad():
        m._flame_exp.copy_(torch.tensor([[0.3, -0.2, 0.1, 0.05]])); m._flame_pose[0, 0] = 0.25; m._flame_trans.copy_(torch.tensor([[0.01, 0.02, -0.03]]))
        m._scales.mul_(1.3)
    path = str(tmp_path / "point_cloud" / "iteration_3" / "point_cloud.ply")
    m.save_ply(path)
    params = torch.load(path.replace("point_cloud.ply", "flame_params.pt"), weights_only=False)
    assert list(params)[:9] == ["_flame_shape", "_flame_exp", "_flame_pose", "_flame_neck_pose", "_flame_trans", "_vertices_enlargement",
                                "faces", "alpha", "point_cloud"]
    assert set(params) - set(hip_model.HipGaussianFlameModel.FLAME_ATTRS) == {"_alpha", "_scales"}
    m2 = hip_model.HipGaussianFlameModel(3)
    m2.load_ply(path, device="cpu")
    for a in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest", "alpha", "_alpha", "_scales", "_flame_exp",
              "_flame_pose", "_flame_trans", "_vertices_enlargement", "faces", "vertices"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
    assert torch.equal(m.get_scaling.detach(), m2.get_scaling.detach()) and torch.equal(m.get_rotation.detach(), m2.get_rotation.detach())
    # the loaded model animates: a new expression moves the vertices and re-derives scale / rotation
    with torch.no_grad():
        m2._flame_exp.fill_(0.8)
    m2.update_alpha(); m2.prepare_scaling_rot()
    assert float((m2._xyz - m._xyz).abs().max()) > 1e-4 and torch.isfinite(m2._rotation).all()


def test_flame_file_written_by_the_reference_loads_as_the_reference_would_hold_it(tmp_path, monkeypatch):
    """A flame_params.pt with ONLY the reference's nine keys: xyz / scaling / rotation come from the PLY, `alpha` from the file,
    `vertices` is None (gaussian_flame_model.py:253-265), the getters serve the PLY columns."""
    from games_hip import model as hip_model
    m = _flame_model_on_cpu(monkeypatch)
    path = str(tmp_path / "point_cloud.ply")
    m.save_ply(path)
    pt = path.replace("point_cloud.ply", "flame_params.pt")
    params = torch.load(pt, weights_only=False)
    torch.save({k: params[k] for k in hip_model.HipGaussianFlameModel.FLAME_ATTRS}, pt)
    m2 = hip_model.HipGaussianFlameModel(3)
    m2.load_ply(path, device="cpu")
    assert m2.vertices is None and not hasattr(m2, "_scales")
    assert torch.equal(m2.alpha, m.alpha.detach()) and torch.equal(m2._xyz.detach(), m._xyz.detach())
    assert torch.equal(m2.get_scaling.detach(), torch.exp(m._scaling.detach()))
    assert torch.allclose(m2.get_rotation.detach(), torch.nn.functional.normalize(m._rotation.detach()))
    assert isinstance(m2._xyz, torch.nn.Parameter) and m2.active_sh_degree == m2.max_sh_degree


def test_reference_flame_model_loads_the_file_the_standalone_model_wrote(tmp_path, monkeypatch):
    """The reference's own `GaussianFlameModel.load_ply` (executed here, CPU) reads point_cloud.ply + flame_params.pt written
    by `HipGaussianFlameModel.save_ply`: same attributes, extra keys ignored.  Needs the reference tree: skipped on the GPU box."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference tree not present")
    ref_import.import_reference()
    try:
        import games
        m = _flame_model_on_cpu(monkeypatch)
        path = str(tmp_path / "point_cloud.ply")
        m.save_ply(path)
        real_load = torch.load
        monkeypatch.setattr(torch, "load", lambda *a, **k: real_load(*a, **{"weights_only": False, **k}))
        with ref_import.cuda_literals_on_cpu():
            r = games.gaussianModelRender["gs_flame"](3)
            r.load_ply(path)
        for a in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"):
            assert torch.equal(getattr(r, a).detach().cpu(), getattr(m, a).detach()), a
        assert torch.equal(r.alpha, m.alpha) and torch.equal(r.faces, m.faces) and torch.equal(r._flame_exp, m._flame_exp)
        assert type(r.point_cloud).__name__ == "_FlameCloud"
    finally:
        ref_import.drop_reference_stubs()
