// bhray_layout.cpp — compile-time proof that include/bhray.h carries the reference's byte layouts.
//
// Every size and offset the Rust host's #[repr(C)] structs (and the WGSL structs they mirror) define is asserted here, so a
// drift of the header cannot compile.  Sources: RayDetails ray_pipeline.rs:3-14 (ray.wgsl:25-34); CameraUniform
// camera.rs:66-73 (ray.wgsl:41-45); BlackHoleUniform blackhole.rs:37-51 (ray.wgsl:112-123); NodeUniform triangle.rs:45-52
// (ray.wgsl:85-90); Triangle triangle.rs:54-63 (ray.wgsl:67-74); ModelUniform triangle.rs:268-285 (ray.wgsl:47-65);
// MaterialUniform material.rs:7-11.
#include <stddef.h>

#include "../../include/bhray.h"

#define OFF(T, m, v) static_assert(offsetof(T, m) == (v), #T "." #m " offset")
#define SZ(T, v) static_assert(sizeof(T) == (v), #T " size")

SZ(bhray_details, 32);
OFF(bhray_details, material_count, 0); OFF(bhray_details, model_count, 4); OFF(bhray_details, time, 8);
OFF(bhray_details, integration_method, 12); OFF(bhray_details, step_size, 16); OFF(bhray_details, max_iterations, 20);
OFF(bhray_details, angle_division_threshold, 24); OFF(bhray_details, highlight_interpolation, 28);

SZ(bhray_camera_uniform, 32);
OFF(bhray_camera_uniform, position, 0); OFF(bhray_camera_uniform, _padding, 12); OFF(bhray_camera_uniform, forward, 16);
OFF(bhray_camera_uniform, fov, 28);

SZ(bhray_black_hole_uniform, 132);
OFF(bhray_black_hole_uniform, accretion_disk_inner, 0); OFF(bhray_black_hole_uniform, accretion_disk_outer, 4);
OFF(bhray_black_hole_uniform, rotation_speed, 8); OFF(bhray_black_hole_uniform, relativity_sphere_radius, 12);
OFF(bhray_black_hole_uniform, position, 16); OFF(bhray_black_hole_uniform, show_disk_texture, 28);
OFF(bhray_black_hole_uniform, normal, 32); OFF(bhray_black_hole_uniform, show_red_shift, 44);
OFF(bhray_black_hole_uniform, rotation_matrix, 48); OFF(bhray_black_hole_uniform, feather_amount, 96);
OFF(bhray_black_hole_uniform, pad, 100);

SZ(bhray_node, 32);
OFF(bhray_node, min_corner, 0); OFF(bhray_node, left_child, 12); OFF(bhray_node, max_corner, 16); OFF(bhray_node, obj_count, 28);

SZ(bhray_triangle, 24);
OFF(bhray_triangle, p1, 0); OFF(bhray_triangle, p2, 4); OFF(bhray_triangle, p3, 8);
OFF(bhray_triangle, n1, 12); OFF(bhray_triangle, n2, 16); OFF(bhray_triangle, n3, 20);

// ModelUniform: 48-byte header, then points / normals (vec4 each), triangles (24 B), nodes (32 B), bvh_lookup (4 B),
// each BHRAY_MAX_MODEL_VERTICES long, then 28 bytes of tail padding (triangle.rs:268-285)
SZ(bhray_model_header, 48);
OFF(bhray_model_header, position, 0); OFF(bhray_model_header, visible, 12); OFF(bhray_model_header, rotation, 16);
OFF(bhray_model_header, pad3, 28); OFF(bhray_model_header, point_count, 32); OFF(bhray_model_header, normal_count, 36);
OFF(bhray_model_header, triangle_count, 40); OFF(bhray_model_header, pad0, 44);
static_assert(BHRAY_MAX_MODEL_VERTICES == 524288, "triangle.rs:7");
static_assert(BHRAY_MODEL_OFF_POINTS == 48u, "points");
static_assert(BHRAY_MODEL_OFF_NORMALS == 8388656u, "normals = 48 + 16 * 524288");
static_assert(BHRAY_MODEL_OFF_TRIANGLES == 16777264u, "triangles = normals + 16 * 524288");
static_assert(BHRAY_MODEL_OFF_NODES == 29360176u, "nodes = triangles + 24 * 524288");
static_assert(BHRAY_MODEL_OFF_LOOKUP == 46137392u, "bvh_lookup = nodes + 32 * 524288");
static_assert(BHRAY_MODEL_OFF_LOOKUP + 4u * BHRAY_MAX_MODEL_VERTICES + 28u == BHRAY_MODEL_UNIFORM_BYTES, "ModelUniform size");
static_assert(BHRAY_MODEL_UNIFORM_BYTES == 48234572u, "ModelUniform size (triangle.rs:268-285)");
static_assert(BHRAY_MAX_MODELS == 1 && BHRAY_MAX_MATERIALS == 8, "triangle.rs:6, material.rs:3");

// not reference layouts, but ABI the bindings restate (bhusie_amd/layouts.py)
static_assert(sizeof(bhray_counters) == 104, "bhray_counters");
static_assert(BHRAY_COMM_ID_BYTES == 128, "ncclUniqueId");
OFF(bhray_config, level_w, 12); OFF(bhray_config, level_h, 12 + 4 * BHRAY_MAX_LEVELS); OFF(bhray_config, crop_x, 12 + 8 * BHRAY_MAX_LEVELS);
OFF(bhray_config, superset_levels, 12 + 8 * BHRAY_MAX_LEVELS + 44);
OFF(bhray_config, device_count, 12 + 8 * BHRAY_MAX_LEVELS + 48);
OFF(bhray_config, devices, 12 + 8 * BHRAY_MAX_LEVELS + 52);
OFF(bhray_config, gather, 12 + 8 * BHRAY_MAX_LEVELS + 52 + 4 * BHRAY_MAX_DEVICES);
OFF(bhray_config, comm_id, 12 + 8 * BHRAY_MAX_LEVELS + 52 + 4 * BHRAY_MAX_DEVICES + 8);
OFF(bhray_config, partition, 12 + 8 * BHRAY_MAX_LEVELS + 52 + 4 * BHRAY_MAX_DEVICES + 8 + BHRAY_COMM_ID_BYTES);
OFF(bhray_config, slab_row0, 12 + 8 * BHRAY_MAX_LEVELS + 52 + 4 * BHRAY_MAX_DEVICES + 8 + BHRAY_COMM_ID_BYTES + 4);
SZ(bhray_config, 12 + 8 * BHRAY_MAX_LEVELS + 52 + 4 * BHRAY_MAX_DEVICES + 8 + BHRAY_COMM_ID_BYTES + 4 + 4 * (BHRAY_MAX_DEVICES + 1));
