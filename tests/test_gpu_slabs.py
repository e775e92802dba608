"""-m gpu: contiguous row slabs (bhray_config.partition = BHRAY_PARTITION_SLABS), the per-row work counters that balance them
(bhray_get_row_work) and the whole chain calibration frame -> bhray_balance_slabs -> partitioned ctx, through the C ABI.

Every partitioned frame is compared byte for byte with the frame ONE partition-less ctx renders from the same uniforms; the row
work is compared with the CPU oracle's per-ray iteration counts (oracle_render_aux) over the pixels the oracle's own
classification traces - an exact integer identity, not a tolerance."""
import numpy as np
import pytest

import bhusie_amd as B
from oracle import oracle as O
from tests import common as T

pytestmark = pytest.mark.gpu


def _whole(cfg, u, tex, **kw):
    rp = B.RayPass(cfg, device=0, **kw)
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    rp.render()
    return rp


@pytest.mark.parametrize("spec", [0, 2])
@pytest.mark.parametrize("method", [0, 1])
def test_row_work_is_the_oracles_iteration_count_per_level_row(method, spec):
    tex = T.textures()
    u = T.uniforms(integration_method=method)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    rp = _whole(cfg, u, tex, counters=True, speculative_levels=spec)
    got = rp.row_work()
    total = rp.counters()
    rp.close()
    sc = T.oracle_scene(*u, tex)
    sizes = cfg.sizes()
    imgs = O.render_ladder(sc, sizes)
    # only the level rows a frame row depends on are rendered (the frame is a window of the last level): ray.wgsl:185-201, top down
    needed = [None] * len(sizes)
    rows = np.arange(cfg.crop_y, cfg.crop_y + cfg.frame_h)
    for l in range(len(sizes) - 1, -1, -1):
        needed[l] = np.zeros(sizes[l][1], dtype=bool)
        needed[l][rows] = True
        if l > 0:
            h, ph = sizes[l][1], sizes[l - 1][1]
            ry = np.float32(ph) / np.float32(h + ((h - 1) // (ph - 1) - 1))
            tl = np.floor(rows.astype(np.float32) * ry).astype(np.int64)
            rows = np.unique(np.concatenate([np.clip(tl, 0, ph - 1), np.clip(tl + 1, 0, ph - 1)]))
    for l, (w, h) in enumerate(sizes):
        its = O.render_aux(sc, (w, h))[..., 1].astype(np.int64)
        traced = np.ones((h, w), dtype=bool) if l < max(spec, 1) else O.classify_level(sc, (w, h), imgs[l - 1]) == 2
        traced &= needed[l][:, None]
        if l == len(sizes) - 1:                                           # the frame window of the last level
            win = np.zeros((h, w), dtype=bool)
            win[:, cfg.crop_x:cfg.crop_x + cfg.frame_w] = True
            traced &= win
        want = (its * traced).sum(axis=1)
        assert np.array_equal(got[l].astype(np.int64), want), (l, np.nonzero(got[l].astype(np.int64) != want)[0][:8])
    assert sum(int(a.sum()) for a in got) > 0 and total["traced"] > 0


@pytest.mark.parametrize("nparts,root", [(2, 0), (3, 2), (8, 0)])
def test_balanced_slabs_assemble_the_single_ctx_frame(nparts, root):
    """calibration frame -> row work -> balanced bounds -> ONE ctx, N slab partitions on device 0 (tiles travel as RCCL send/recv to
    self), frame batches and speculative levels as bench.py uses them: the frame, the levels and the sky pass equal the single ctx's."""
    tex = T.textures()
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    one = _whole(cfg, u, tex, counters=True, speculative_levels=2)
    want = one.read_hdr()
    bounds = B.balance_slabs(cfg, one.row_work(), nparts)
    assert bounds[0] == 0 and bounds[-1] == 110 and len(bounds) == nparts + 1
    rp = B.RayPass(cfg, devices=[0] * nparts, gather_root=root, slab_row0=bounds, frames_in_flight=2, frames_per_batch=3, speculative_levels=2)
    info = rp.gather_info()
    assert info["partitions"] == nparts
    assert info["bytes_received_per_frame"] == (110 - (bounds[root + 1] - bounds[root])) * (3 * 200 + 7) * 4         # packed rows: x, y, z + alpha bits
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    for _ in range(7):                                       # two full batches + a partial one; slots reused
        rp.render()
    got = rp.read_hdr()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for l in range(2):
        a, b = rp.read_level(l).view(np.uint32), one.read_level(l).view(np.uint32)
        have = ~(a == 0xFFFFFFFF).all(axis=-1)
        assert have.any() and np.array_equal(a[have], b[have])
    rp.resolve_sky(); one.resolve_sky()
    assert np.array_equal(rp.read_sky().view(np.uint16), one.read_sky().view(np.uint16))
    rp.close(); one.close()


def test_a_slab_rank_renders_its_rows_of_the_frame_and_uneven_bounds_are_fine():
    """one process per GPU without the gather (row_rank / row_world + slabs): every rank's packed rows are its rows of the frame -
    including an empty slab and a one-row slab."""
    tex = T.textures()
    u = T.uniforms(integration_method=0)
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    one = _whole(cfg, u, tex)
    want = one.read_hdr()
    one.close()
    bounds = [0, 37, 37, 38, 110]
    for rank in range(4):
        rp = B.RayPass(cfg, device=0, row_rank=rank, row_world=4, slab_row0=bounds)
        rp.set_textures(*tex)
        rp.set_uniforms(*u)
        rp.render()
        rows = rp.local_rows()
        assert rows.tolist() == list(range(bounds[rank], bounds[rank + 1]))
        got = rp.read_hdr()
        assert got.shape[0] == len(rows)
        if len(rows):
            assert np.array_equal(got.view(np.uint32), want[rows].view(np.uint32))
        rp.close()


def test_bad_slab_bounds_are_errors():
    cfg = B.ladder_for_frame((200, 110), 3, 3)
    for bounds in ([0, 50, 40, 110], [1, 50, 110], [0, 50, 109]):
        with pytest.raises(B.BhrayError):
            B.RayPass(cfg, device=0, row_rank=0, row_world=len(bounds) - 1, slab_row0=bounds)
    with pytest.raises(B.BhrayError):
        B.RayPass(cfg, devices=[0, 0], slab_row0=[0, 60, 111])
    rp = B.RayPass(cfg, device=0)                            # no counters: no row work
    with pytest.raises(B.BhrayError):
        rp.row_work()
    rp.close()


def test_bench_frame_on_eight_balanced_slabs_is_the_single_gpu_frame():
    """configs[1] at full 1920x1080 the way bench.py --gpus 8 now runs it: balanced slabs, 8 frames per batch, speculative levels."""
    tex = T.textures(small=False)
    u = T.uniforms(integration_method=1)
    cfg = B.ladder_for_frame((1920, 1080), 3, 4)
    one = _whole(cfg, u, tex, frames_in_flight=1, counters=True, speculative_levels=2)
    want = one.read_hdr()
    work = one.row_work()
    one.close()
    bounds = B.balance_slabs(cfg, work, 8)
    sizes = [b - a for a, b in zip(bounds, bounds[1:])]
    assert min(sizes) >= 8 and max(sizes) > 2 * min(sizes)              # the hole's slabs are thin, the sky's are thick
    rp = B.RayPass(cfg, devices=[0] * 8, slab_row0=bounds, frames_per_batch=8, frames_in_flight=2, speculative_levels=2)
    rp.set_textures(*tex)
    rp.set_uniforms(*u)
    for _ in range(19):
        rp.render()
    got = rp.read_hdr()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    rp.close()
