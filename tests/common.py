"""Shared scene builders for the tests (oracle side and product side get the SAME bytes)."""
from __future__ import annotations

import functools

import numpy as np

import bhusie_amd as B
from bhusie_amd import assets
from oracle import oracle as O

REL_TOL = 1e-4          # BASELINE.json north_star: "within 1e-4 relative per channel"
ABS_FLOOR = 1e-3        # channel magnitudes below this are compared absolutely (1e-7)


@functools.lru_cache(maxsize=None)
def textures(small=True):
    if small:
        return assets.temp_lut(64), assets.disk_texture(200, seed=1), assets.sky_texture(512, 256, seed=2)
    return assets.temp_lut(256), assets.disk_texture(1000, seed=1), assets.sky_texture(4096, 2048, seed=2)


def uniforms(camera=None, black_hole=None, **details):
    cam = camera or B.Camera()
    bh = black_hole or B.BlackHole()
    det = B.RayDetails(**details)
    return cam.uniform(), bh.uniform(), det.uniform()


def oracle_scene(cam_b, bh_b, det_b, tex, models=()):
    return O.OracleScene(cam_b, bh_b, det_b, tex[0], tex[1], tex[2], list(models))


def rel_err(got, want):
    return np.abs(got - want) / np.maximum(np.abs(want), ABS_FLOOR)


def assert_parity(got, want, what=""):
    """Classes (alpha) identical; every channel within REL_TOL; reports bit-exact fraction."""
    assert got.shape == want.shape, (got.shape, want.shape)
    fin = np.isfinite(want).all(axis=-1)
    assert np.isfinite(got[fin]).all(), f"{what}: non-finite output where the oracle is finite"
    assert np.array_equal(got[..., 3][fin], want[..., 3][fin]), f"{what}: pixel class (alpha) mismatch"
    e = rel_err(got[fin], want[fin])
    assert float(e.max(initial=0.0)) <= REL_TOL, f"{what}: max rel err {float(e.max()):.3g} > {REL_TOL}"
    return float(e.max(initial=0.0)), float((got[fin] == want[fin]).all(axis=-1).mean()) if fin.any() else 1.0


class DeviceBuffer:
    """A plain HIP allocation for tests that bind their own output memory (no torch: other tests hide the devices from it)."""

    def __init__(self, nbytes: int, fill: int = 0xFF):
        import ctypes as C
        self._C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0
        assert self.hip.hipMemset(self.ptr, fill, C.c_size_t(nbytes)) == 0
        assert self.hip.hipDeviceSynchronize() == 0

    def read(self, dtype=np.float32) -> np.ndarray:
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        assert self.hip.hipMemcpy(self._C.c_void_p(out.ctypes.data), self.ptr, self._C.c_size_t(self.nbytes), 2) == 0   # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = None


def variant_library(target: str) -> str:
    """Path of bhusie_amd/libbhray_<target>.so - another in-tree build of the SAME sources (`make -C bhusie_amd/csrc <target>`: stack2;
    __graft_entry__.build() makes it).  Built here if it is missing (a fresh checkout on a box with hipcc): never a fallback."""
    import os
    import subprocess
    from bhusie_amd import _lib
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    path = os.path.join(root, f"libbhray_{target}.so")
    if not os.path.exists(path):
        r = subprocess.run(["make", "-C", os.path.join(root, "csrc"), "-j4", target], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0 and os.path.exists(path), f"{path} not built: make -C bhusie_amd/csrc {target}\n{r.stdout[-1500:]}{r.stderr[-1500:]}"
    return path
