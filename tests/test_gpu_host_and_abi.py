"""-m gpu: the boundary and host-side tests (tests/test_abi.py, tests/test_host.py — CPU tests the driver does not run on the GPU
box) executed on the MI355X box as well, so that the C ABI's symbol table and layouts, the C++ host mirror (uniform packing,
OBJ loader, BVH builder), the disk-texture generator against the reference's shipped disk.png samples, and the Rust-binding /
header lock step carry evidence from the same run as the kernels.  Plus what only a GPU box can show: bhray_create succeeds,
device enumeration, the C++ host program runs."""
import os
import subprocess

import pytest

import bhusie_amd as B
from tests import test_abi, test_host

pytestmark = pytest.mark.gpu

NO_ARGS = [test_abi.test_every_declared_symbol_is_exported_and_bound, test_abi.test_layout_sizes_match_reference_structs,
           test_abi.test_reference_ladder_rule, test_abi.test_errors_are_codes_not_crashes, test_abi.test_product_does_not_touch_the_oracle,
           test_abi.test_integration_md_rust_binding_matches_the_header, test_abi.test_header_layout_asserts_are_compiled_into_the_library,
           test_host.test_uniform_bytes_match_host_oracle_and_golden, test_host.test_black_hole_default_orientation_values,
           test_host.test_degenerate_models, test_host.test_disk_texture_generator_reproduces_reference_disk_png]
TMP_PATH = [test_host.test_bvh_matches_golden_fixture, test_host.test_obj_reader_forms, test_host.test_sah_builder_behind_a_flag_is_a_valid_bvh]


@pytest.mark.parametrize("fn", NO_ARGS, ids=lambda f: f.__module__.split(".")[-1] + "." + f.__name__)
def test_cpu_suite_on_the_gpu_box(fn):
    fn()


@pytest.mark.parametrize("fn", TMP_PATH, ids=lambda f: f.__module__.split(".")[-1] + "." + f.__name__)
def test_cpu_suite_on_the_gpu_box_tmp(fn, tmp_path):
    fn(tmp_path)


@pytest.mark.parametrize("with_normals", [True, False])
def test_load_model_and_bvh_on_the_gpu_box(tmp_path, with_normals):
    test_host.test_load_model_and_bvh_match_python_restatement(tmp_path, with_normals)


def test_device_is_present_and_create_succeeds():
    assert B.lib().bhray_device_count() >= 1
    rp = B.RayPass(B.ladder_from_base((72, 41), 3, 4))
    info = rp.gather_info()
    assert info == dict(partitions=1, local_partitions=1, root=0, root_is_local=1, comm_ranks=0, rccl_version=info["rccl_version"],
                        bytes_sent_per_frame=0, bytes_received_per_frame=0)
    rp.set_materials()                                          # mod.rs:389: accepted, ignored
    with pytest.raises(B.BhrayError):
        rp.render()                                             # uniforms not set: a call-order error, not a crash
    rp.close()
