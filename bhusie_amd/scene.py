"""Mirrors of the reference scene types that become uniform bytes for the ray pass.

Camera / CameraUniform           /root/reference/src/scene/camera.rs:3-90
BlackHole / BlackHoleUniform     /root/reference/src/scene/blackhole.rs:3-98
RayDetails                       /root/reference/src/renderer/pipelines/ray_pipeline.rs:3-14 (+ defaults mod.rs:116-121)
The arithmetic (cgmath quaternion math for the disk orientation) runs in the C++ host code behind
the C ABI (bhray_host.cpp); these classes only hold fields and call it.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

from ._lib import lib
from .layouts import BhrayBlackHole, BhrayBlackHoleUniform, BhrayCameraUniform, BhrayDetails


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


@dataclass
class Camera:
    position: tuple = (0.0, 0.0, -19.0)      # camera.rs:12
    forward: tuple = (0.0, 0.0, 1.0)
    fov: float = 1.0

    def uniform(self) -> bytes:
        u = BhrayCameraUniform()
        lib().bhray_camera_uniform_update(C.byref(u), _f3(self.position), _f3(self.forward), float(self.fov))
        return bytes(u)


@dataclass
class BlackHole:
    position: tuple = (0.0, 0.0, 0.0)         # blackhole.rs:16-28
    accretion_disk_rotation: tuple = (0.15, 0.0, 0.25)
    accretion_disk_inner: float = 2.0
    accretion_disk_outer: float = 10.0
    rotation_speed: float = 1.0
    relativity_sphere_radius: float = 20.0
    show_disk_texture: int = 1
    show_red_shift: int = 1
    feather_amount: float = 0.3

    def uniform(self) -> bytes:
        b = BhrayBlackHole()
        b.position[:] = [float(x) for x in self.position]
        b.accretion_disk_rotation[:] = [float(x) for x in self.accretion_disk_rotation]
        b.accretion_disk_inner = self.accretion_disk_inner; b.accretion_disk_outer = self.accretion_disk_outer
        b.rotation_speed = self.rotation_speed; b.relativity_sphere_radius = self.relativity_sphere_radius
        b.show_disk_texture = self.show_disk_texture; b.show_red_shift = self.show_red_shift
        b.feather_amount = self.feather_amount
        u = BhrayBlackHoleUniform()
        lib().bhray_black_hole_uniform_update(C.byref(u), C.byref(b))
        return bytes(u)


@dataclass
class RayDetails:
    material_count: int = 0
    model_count: int = 0
    time: float = 0.0
    integration_method: int = 0               # 0 Euler (the reference default), 1 "Runge Kutta"
    step_size: float = 0.15
    max_iterations: int = 2000
    angle_division_threshold: float = 0.02
    highlight_interpolation: int = 0

    def uniform(self) -> bytes:
        d = BhrayDetails(self.material_count, self.model_count, self.time, self.integration_method, self.step_size,
                         self.max_iterations, self.angle_division_threshold, self.highlight_interpolation)
        return bytes(d)
