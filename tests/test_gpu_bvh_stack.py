"""-m gpu: the BVH traversal's short LDS stack + restart trail (bhray_kernels.hip: trace_ray_model) where it is under
stress - a pop that finds the ring overwritten re-descends from the root along the recorded path and must repeat no decision.

 1. A degenerate CHAIN tree (every split peels one triangle off: depth = triangle count) deeper than the ring's 8 entries: more far
    children are pending than the ring holds, so the re-descent runs for real in the shipped library.
 2. The same sources built with a ring of TWO entries (`make -C bhusie_amd/csrc stack2` -> libbhray_stack2.so, also built by
    __graft_entry__.build()): every ordinary mesh overflows it constantly.
Frames must equal the CPU oracle's (which keeps a plain 64-entry index stack) to the parity bar, direction pixels bit for bit, and the
node / triangle counters must be EQUAL - a repeated or skipped decision changes them.
 3. A chain deeper than BHRAY_BVH_STACK = 64 levels must come back as BHRAY_E_BVH_DEPTH, not as pixels."""
import ctypes as C
import os

import numpy as np
import pytest

import bhusie_amd as B
from bhusie_amd import assets
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu


def chain_model(n, x0=0.02, position=(0.0, 0.0, 0.0)):
    """n triangles facing the x axis at x_k = x0 * 2.5^k, size proportional to x_k: the midpoint split (triangle.rs:196-259) of every node
    separates the farthest triangle from all the others, so the tree is a chain about n levels deep."""
    m = B.Model()
    m.add_normal((1.0, 0.0, 0.0))
    for k in range(n):
        x = x0 * (2.5 ** k)
        s = 0.25 * x
        for p in ((x, -s, -s), (x, s, -s), (x, 0.0, s)):
            m.add_vertex(p)
        m.add_triangle((3 * k, 3 * k + 1, 3 * k + 2, 0, 0, 0))
    m.build_bvh()
    m.set_transform(position, 1)
    return m


def render(cfg, u, tex, model, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    rp.upload_model(model)
    rp.set_uniforms(*u)
    rp.render()
    return rp


def check_against_oracle(cfg, u, tex, model, what):
    rp = render(cfg, u, tex, model, counters=True)
    cnt = O.Counters()
    want = O.render_ladder(T.oracle_scene(*u, tex, [model.arrays()]), cfg.sizes(), cnt)
    got = rp.read_hdr()
    T.assert_parity(got, want[-1], what)
    d = want[-1][..., 3] == 0
    assert np.array_equal(got[d].view(np.uint32), want[-1][d].view(np.uint32)), what
    c = rp.counters()
    rp.close()
    assert c == cnt.as_dict(), (what, c, cnt.as_dict())
    return c


@pytest.mark.parametrize("method", [0, 1])
def test_chain_tree_deeper_than_the_ring(method):
    model = chain_model(22, position=(-30.0, 0.0, 0.0))                  # 17 of the far boxes lie within t_max = 1e5: twice the ring
    assert model.max_depth() >= 20
    tex = T.textures()
    # camera outside the relativity sphere, looking down the chain's axis: the first flat iteration of every ray traverses the chain
    cam = B.Camera(position=(-45.0, 0.5, 0.3), forward=(1.0, -0.01, -0.005), fov=0.9)
    u = T.uniforms(camera=cam, integration_method=method, model_count=1)
    cfg = B.ladder_from_base((40, 24), 3, 2)
    c = check_against_oracle(cfg, u, tex, model, f"chain, method {method}")
    assert c["node_pairs"] > 20 * 100 and c["triangles"] > 0


def test_chain_deeper_than_the_trail_is_an_error_not_pixels():
    model = chain_model(70, x0=1.0, position=(-30.0, 0.0, 0.0))          # 2.5^69 = 2.9e27: representable; 68 levels
    assert model.max_depth() > 64
    tex = T.textures()
    cam = B.Camera(position=(-45.0, 0.5, 0.3), forward=(1.0, -0.01, -0.005), fov=0.9)
    u = T.uniforms(camera=cam, integration_method=1, model_count=1)
    cfg = B.ladder_from_base((24, 14), 3, 2)
    rp = render(cfg, u, tex, model)
    with pytest.raises(B.BhrayError) as e:
        rp.sync()
    assert e.value.code == -6                                              # BHRAY_E_BVH_DEPTH
    rp.close()


@pytest.fixture()
def stack2_library():
    """libbhray_stack2.so: the same sources with -DBHRAY_BVH_LDS_STACK=2"""
    from bhusie_amd import _lib, layouts
    path = T.variant_library("stack2")
    saved = _lib.lib()
    L = C.CDLL(path)
    layouts.declare(L)
    _lib._lib = L
    yield
    _lib._lib = saved


@pytest.mark.parametrize("method", [0, 1])
def test_ring_of_two_entries_on_ordinary_meshes(stack2_library, tmp_path, method):
    tex = T.textures()
    p = tmp_path / "mesh.obj"
    p.write_text(assets.sphere_mesh_obj(20, 28, radius=8.0, bump=0.2, seed=5))
    model = B.load_model(str(p))
    d = np.array([-0.12, 0.0, 1.0]); d /= np.linalg.norm(d)
    cam = B.Camera(position=(0.0, 0.0, -40.0), forward=tuple(d), fov=1.2)
    u = T.uniforms(camera=cam, integration_method=method, model_count=1)
    c = check_against_oracle(B.ladder_from_base((40, 24), 3, 2), u, tex, model, f"ring of 2, sphere mesh, method {method}")
    assert c["triangles"] > 0 and c["node_pairs"] > 0
    p.write_text(assets.icosphere_mesh_obj(4, radius=8.0, bump=0.15, seed=3))       # 5 120 triangles, camera inside the sphere, mesh behind the hole
    model = B.load_model(str(p))
    model.set_transform((-6.0, 0.0, 32.0), 1)
    u = T.uniforms(integration_method=method, model_count=1)
    check_against_oracle(B.ladder_from_base((48, 27), 3, 2), u, tex, model, f"ring of 2, icosphere behind the hole, method {method}")
    c = check_against_oracle(B.ladder_from_base((40, 24), 3, 2), T.uniforms(camera=B.Camera(position=(-45.0, 0.5, 0.3), forward=(1.0, -0.01, -0.005), fov=0.9),
                                                                          integration_method=method, model_count=1),
                             tex, chain_model(22, position=(-30.0, 0.0, 0.0)), f"ring of 2, chain, method {method}")
    assert c["node_pairs"] > 0
