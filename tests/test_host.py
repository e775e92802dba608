"""CPU: the C++ host mirror (bhray_host.cpp) against the Python restatement of the Rust host code
(oracle/host_oracle.py) and the golden fixtures.  Integer / byte work: bit-exact."""
import os

import numpy as np
import pytest

import bhusie_amd as B
from bhusie_amd import assets
from bhusie_amd.model import NODE_DTYPE
from oracle import host_oracle as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_uniform_bytes_match_host_oracle_and_golden():
    g = np.load(os.path.join(GOLD, "functions.npz"))
    assert B.BlackHole().uniform() == H.black_hole_uniform() == g["bh_uniform.default"].tobytes()
    # general rotation: sin/cos come from the platform libm on both sides (Rust f32::sin_cos in the
    # reference), so agreement is to a few ulp, not to the bit
    kw = dict(accretion_disk_rotation=(1.1, -0.4, 2.0), position=(1.0, 2.0, 3.0))
    a = np.frombuffer(B.BlackHole(**kw).uniform(), dtype="<f4")
    b = np.frombuffer(H.black_hole_uniform(**kw), dtype="<f4")
    assert H.black_hole_uniform(**kw) == g["bh_uniform.rot"].tobytes()
    assert np.allclose(a[:25], b[:25], rtol=0, atol=4e-7)
    assert a[:8].tobytes() == b[:8].tobytes()                      # radii, position, flags: copied verbatim
    assert B.Camera().uniform() == H.camera_uniform()
    assert B.Camera(position=(1, 2, 3), forward=(0.6, 0.0, 0.8), fov=0.7).uniform() == \
        H.camera_uniform(position=(1, 2, 3), forward=(0.6, 0.0, 0.8), fov=0.7)
    assert B.RayDetails().uniform() == H.ray_details()            # mod.rs:116-121 defaults
    d = B.RayDetails()
    assert (d.step_size, d.max_iterations, d.angle_division_threshold, d.integration_method) == (0.15, 2000, 0.02, 0)


def test_black_hole_default_orientation_values():
    """default disk rotation (0.15,0,0.25) -> normal ≈ (0.24740,-0.95803,-0.14479)."""
    import struct
    u = B.BlackHole().uniform()
    n = struct.unpack_from("<3f", u, 32)
    assert np.allclose(n, (0.24740, -0.95803, -0.14479), atol=2e-5)
    right = struct.unpack_from("<3f", u, 48)
    assert np.allclose(right, (0.95803, 0.24740, 0.0), atol=2e-5)
    assert abs(np.linalg.norm(n) - 1) < 1e-6


def _check_bvh_invariants(a):
    nodes, lookup, tris, pts = a["nodes"], a["bvh_lookup"], a["triangles"], a["points"]
    T = len(tris)
    assert sorted(lookup.tolist()) == list(range(T))              # a permutation
    seen = np.zeros(T, dtype=int)
    stack = [0]
    visited = 0
    while stack:
        i = stack.pop(); visited += 1
        n = nodes[i]
        if n["obj_count"] == 0:
            l = int(n["left_child"])
            assert l + 1 < len(nodes)                              # children adjacent (triangle.rs:239-243)
            for c in (l, l + 1):
                assert np.all(nodes[c]["min_corner"] >= n["min_corner"]) and np.all(nodes[c]["max_corner"] <= n["max_corner"])
            stack += [l, l + 1]
        else:
            ids = lookup[int(n["left_child"]):int(n["left_child"]) + int(n["obj_count"])]
            seen[ids] += 1
            P = pts[tris[ids, :3].reshape(-1), :3]
            assert np.array_equal(P.min(axis=0), n["min_corner"]) and np.array_equal(P.max(axis=0), n["max_corner"])
    assert visited == len(nodes)
    assert np.all(seen == 1)                                       # every triangle in exactly one leaf


@pytest.mark.parametrize("with_normals", [True, False])
def test_load_model_and_bvh_match_python_restatement(tmp_path, with_normals):
    text = assets.sphere_mesh_obj(14, 18, radius=7.0, bump=0.25, seed=9, with_normals=with_normals)
    p = tmp_path / "m.obj"; p.write_text(text)
    m = B.load_model(str(p))
    a = m.arrays()
    ref = H.load_model(text)
    d = ref.as_oracle_dict()
    assert np.array_equal(a["points"], d["points"])               # model.rs:36-38 scaling (0.5,-0.5,0.5)
    assert np.array_equal(a["normals"], d["normals"])
    assert np.array_equal(a["triangles"], d["triangles"])
    assert np.array_equal(a["bvh_lookup"], d["bvh_lookup"])
    assert a["nodes"].tobytes() == d["nodes"].tobytes()
    assert m.max_depth() == ref.max_depth()
    assert tuple(a["position"]) == (-10.0, 0.0, 30.0) and a["visible"] == 1     # triangle.rs:100,108
    _check_bvh_invariants(a)
    assert m.pack_uniform() == ref.pack_uniform()                  # exact 48 234 572-byte ModelUniform image


def test_bvh_matches_golden_fixture(tmp_path):
    g = np.load(os.path.join(GOLD, "mesh.npz"))
    for tag, key in (("m", "obj"), ("f", "obj_flat")):
        p = tmp_path / f"{tag}.obj"; p.write_bytes(g[key].tobytes())
        m = B.load_model(str(p))
        a = m.arrays()
        for k in ("points", "normals", "triangles", "bvh_lookup"):
            assert np.array_equal(a[k], g[f"{tag}.{k}"]), k
        assert a["nodes"].tobytes() == g[f"{tag}.nodes"].tobytes()
        assert m.max_depth() == int(g[f"{tag}.max_depth"][0])


def test_degenerate_models():
    m = B.Model()
    m.build_bvh()                                                  # empty
    a = m.arrays()
    assert len(a["nodes"]) == 1 and a["nodes"][0]["obj_count"] == 0
    m = B.Model()
    for p in ((0, 0, 0), (1, 0, 0), (0, 1, 0)):
        m.add_vertex(p)
    m.add_normal((0, 0, 1))
    for _ in range(5):                                             # identical centroids: split fails, one leaf of 5
        m.add_triangle((0, 1, 2, 0, 0, 0))
    m.build_bvh()
    a = m.arrays()
    assert len(a["nodes"]) == 1 and a["nodes"][0]["obj_count"] == 5
    with pytest.raises(B.BhrayError):
        bad = B.Model(); bad.add_vertex((0, 0, 0)); bad.add_triangle((0, 1, 2, 0, 0, 0)); bad.build_bvh()


def test_obj_reader_forms(tmp_path):
    text = "v 0 0 0\nv 2 0 0\nv 0 2 0\nv 0 0 2\nvn 0 0 1\nvt 0 0\nf 1/1/1 2/1/1 3/1/1\nf -4//-1 -2//-1 -1//-1  # comment\n"
    p = tmp_path / "a.obj"; p.write_text(text)
    a = B.load_model(str(p)).arrays()
    assert a["triangles"].tolist() == [[0, 1, 2, 0, 0, 0], [0, 2, 3, 0, 0, 0]]
    assert np.array_equal(a["points"][1], [1.0, -0.0, 0.0, 0.0])
    q = tmp_path / "quad.obj"; q.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(B.BhrayError) as e:                         # tobj default options do not triangulate
        B.load_model(str(q))
    assert e.value.code == -7
    with pytest.raises(B.BhrayError):
        B.load_model(str(tmp_path / "missing.obj"))


def test_disk_texture_generator_reproduces_reference_disk_png():
    """perlin/src/main.rs restated (C++ behind the ABI, and NumPy): both regenerate the reference's shipped disk.png up to
    libm rounding (cos/sin/atan2/powf of the platform) — the one artefact the reference itself provides for this path."""
    g = np.load(os.path.join(GOLD, "disk_png_samples.npz"))
    cpp = assets.reference_disk_texture(1000)
    assert cpp.shape == (1000, 1000, 4) and np.array_equal(cpp[..., 0], cpp[..., 3]) and np.array_equal(cpp[..., 0], cpp[..., 1])
    got = cpp[g["y"], g["x"]].astype(int)
    d = np.abs(got - g["rgba"].astype(int))
    assert d.max() <= 2 and (d == 0).mean() > 0.999
    assert abs(float(cpp.mean()) - float(g["mean"][0])) < 1e-3
    hist = np.bincount(cpp[..., 0].ravel(), minlength=256)
    assert np.abs(hist - g["hist"]).sum() <= 200                     # <= 100 of 10^6 pixels move to a neighbouring bin
    py = H.perlin_disk(1000)
    dd = np.abs(py.astype(int) - cpp.astype(int))
    assert dd.max() <= 2 and (dd == 0).mean() > 0.9999
    ref_png = "/root/reference/src/renderer/textures/disk.png"
    if os.path.exists(ref_png):                                       # the build container has the reference checkout
        from PIL import Image
        ref = np.array(Image.open(ref_png))
        e = np.abs(cpp.astype(int) - ref.astype(int))
        assert e.max() <= 2 and (e == 0).mean() > 0.9999


def test_sah_builder_behind_a_flag_is_a_valid_bvh(tmp_path):
    """bhray_model_build_bvh_sah: not the reference's tree, but the same format and invariants;
    it also repairs the huge leaves the midpoint builder leaves on lattice-like meshes."""
    text = assets.sphere_mesh_obj(40, 48, radius=7.0, bump=0.2, seed=9)
    p = tmp_path / "m.obj"; p.write_text(text)
    m = B.load_model(str(p))
    ref = m.arrays()
    ref_leaf_max = int(ref["nodes"]["obj_count"].max())
    m.build_bvh_sah()
    a = m.arrays()
    _check_bvh_invariants(a)
    assert int(a["nodes"]["obj_count"].max()) <= 4 < ref_leaf_max
    assert np.array_equal(a["points"], ref["points"]) and np.array_equal(a["triangles"], ref["triangles"])
    m.build_bvh()                                                   # and back: the reference-identical tree again
    b = m.arrays()
    assert b["nodes"].tobytes() == ref["nodes"].tobytes() and np.array_equal(b["bvh_lookup"], ref["bvh_lookup"])
