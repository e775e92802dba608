// bhray_host.cpp — C++ mirror of the reference's Rust host code that feeds the ray pass.
// (The reference host is compiled Rust; no Rust toolchain exists in this image, so the host side
// above the C ABI is C++.  Names and argument meaning follow the Rust.)
//
//   Model / build_bvh / update_bounds / subdivide   /root/reference/src/renderer/triangle.rs:65-259
//   ModelUniform::update (byte image)               src/renderer/triangle.rs:268-325
//   load_model                                       src/renderer/model.rs:7-87
//   CameraUniform::update                            src/scene/camera.rs:84-88
//   BlackHole::new, BlackHoleUniform::update         src/scene/blackhole.rs:16-28, 68-98
//   RayDetails defaults                              src/renderer/mod.rs:116-121
//
// Compiled with -ffp-contract=off: the float arithmetic is one binary32 operation per Rust
// operator, in source order.  cgmath 0.18 (Euler→Quaternion, Quaternion*Vector3, normalize) is a
// crates.io dependency that is not vendored in /root/reference; its published formulas are
// restated here (parity unpinned by the reference: DESIGN.md §Oracle).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/bhray.h"

namespace {

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 normalize(V3 a) { float m = sqrtf(dot(a, a)); return a * (1.0f / m); }   // cgmath InnerSpace::normalize

}  // namespace

struct bhray_model {
    float position[3] = {-10.0f, 0.0f, 30.0f};          // triangle.rs:100
    float rotation[3] = {0.0f, 0.0f, 0.0f};
    int32_t visible = 1;
    std::vector<float> points;                            // 4 per point
    std::vector<float> normals;                           // 4 per normal
    std::vector<bhray_triangle> triangles;
    std::vector<bhray_node> nodes;
    std::vector<int32_t> bvh_lookup;
    size_t nodes_used = 0;

    V3 point(int32_t i) const { return v3(points[4 * (size_t)i], points[4 * (size_t)i + 1], points[4 * (size_t)i + 2]); }

    // update_bounds, triangle.rs:159-194
    void update_bounds(size_t ni) {
        bhray_node& n = nodes[ni];
        const float mx = 3.40282347e+38f;
        n.min_corner[0] = n.min_corner[1] = n.min_corner[2] = mx;
        n.max_corner[0] = n.max_corner[1] = n.max_corner[2] = -mx;
        for (int32_t i = 0; i < n.obj_count; i++) {
            const bhray_triangle& t = triangles[(size_t)bvh_lookup[(size_t)(n.left_child + i)]];
            const int32_t idx[3] = {t.p1, t.p2, t.p3};
            for (int k = 0; k < 3; k++) {
                const float* p = &points[4 * (size_t)idx[k]];
                for (int a = 0; a < 3; a++) {
                    n.min_corner[a] = fminf(n.min_corner[a], p[a]);
                    n.max_corner[a] = fmaxf(n.max_corner[a], p[a]);
                }
            }
        }
        // f32::min/max leave the sign of a zero result unspecified (minNum(-0,+0) may be either);
        // canonicalise zero bounds to +0 so that every builder produces the same bytes.
        for (int a = 0; a < 3; a++) { n.min_corner[a] += 0.0f; n.max_corner[a] += 0.0f; }
    }

    // subdivide, triangle.rs:196-259.  The reference recurses (and grows its stack to 1 GiB,
    // main.rs:1-5); this uses an explicit stack with the same visiting order (left subtree
    // completely before right), which is what fixes the node numbering.
    void subdivide_from(size_t root) {
        std::vector<size_t> todo;
        todo.push_back(root);
        while (!todo.empty()) {
            const size_t ni = todo.back();
            todo.pop_back();
            if (nodes[ni].obj_count <= 2) continue;
            const float ext[3] = {nodes[ni].max_corner[0] - nodes[ni].min_corner[0],
                                  nodes[ni].max_corner[1] - nodes[ni].min_corner[1],
                                  nodes[ni].max_corner[2] - nodes[ni].min_corner[2]};
            int axis = 0;
            if (ext[1] > ext[axis]) axis = 1;
            if (ext[2] > ext[axis]) axis = 2;
            const float split = nodes[ni].min_corner[axis] + ext[axis] / 2.0f;
            int32_t i = nodes[ni].left_child;
            int32_t j = i + nodes[ni].obj_count - 1;
            while (i <= j) {
                const bhray_triangle& t = triangles[(size_t)bvh_lookup[(size_t)i]];
                const float* a = &points[4 * (size_t)t.p1];
                const float* b = &points[4 * (size_t)t.p2];
                const float* c = &points[4 * (size_t)t.p3];
                const float centroid = ((a[axis] + b[axis]) + c[axis]) / 3.0f;
                if (centroid < split) {
                    i += 1;
                } else {
                    const int32_t tmp = bvh_lookup[(size_t)i];
                    bvh_lookup[(size_t)i] = bvh_lookup[(size_t)j];
                    bvh_lookup[(size_t)j] = tmp;
                    j -= 1;
                }
            }
            const int32_t left_count = i - nodes[ni].left_child;
            if (left_count == 0 || left_count == nodes[ni].obj_count) continue;
            const size_t li = nodes_used, ri = nodes_used + 1;
            nodes_used += 2;
            if (nodes.size() < nodes_used) nodes.resize(nodes_used, bhray_node{{0, 0, 0}, 0, {0, 0, 0}, 0});
            nodes[li].left_child = nodes[ni].left_child;
            nodes[li].obj_count = left_count;
            nodes[ri].left_child = i;
            nodes[ri].obj_count = nodes[ni].obj_count - left_count;
            nodes[ni].left_child = (int32_t)li;
            nodes[ni].obj_count = 0;
            update_bounds(li);
            update_bounds(ri);
            // The recursion is subdivide(left) then subdivide(right); node indices are handed out
            // in that depth-first order.  LIFO: push right first so that left is processed next,
            // and everything the left subtree allocates comes before anything the right does.
            todo.push_back(ri);
            todo.push_back(li);
        }
    }
};

extern "C" {

void bhray_camera_uniform_update(bhray_camera_uniform* u, const float position[3], const float forward[3], float fov) {
    if (!u || !position || !forward) return;
    memcpy(u->position, position, 12);
    u->_padding = 0;
    memcpy(u->forward, forward, 12);
    u->fov = fov;
}

void bhray_black_hole_default(bhray_black_hole* bh) {
    if (!bh) return;
    memset(bh, 0, sizeof *bh);
    bh->accretion_disk_rotation[0] = 0.15f; bh->accretion_disk_rotation[1] = 0.0f; bh->accretion_disk_rotation[2] = 0.25f;
    bh->accretion_disk_inner = 2.0f; bh->accretion_disk_outer = 10.0f;
    bh->rotation_speed = 1.0f; bh->relativity_sphere_radius = 20.0f;
    bh->show_disk_texture = 1; bh->show_red_shift = 1; bh->feather_amount = 0.3f;
}

void bhray_details_default(bhray_details* d) {
    if (!d) return;
    memset(d, 0, sizeof *d);
    d->angle_division_threshold = 0.02f; d->step_size = 0.15f; d->max_iterations = 2000;
}

void bhray_black_hole_uniform_update(bhray_black_hole_uniform* u, const bhray_black_hole* bh) {
    if (!u || !bh) return;
    memset(u, 0, sizeof *u);
    memcpy(u->position, bh->position, 12);
    u->accretion_disk_inner = bh->accretion_disk_inner;
    u->accretion_disk_outer = bh->accretion_disk_outer;
    u->rotation_speed = bh->rotation_speed;
    u->relativity_sphere_radius = bh->relativity_sphere_radius;
    u->show_disk_texture = bh->show_disk_texture;
    u->show_red_shift = bh->show_red_shift;
    u->feather_amount = bh->feather_amount;
    // Quaternion::from(Euler{x,y,z}) — cgmath 0.18 quaternion.rs (XYZ order)
    const float half = 0.5f;
    const float sx = sinf(bh->accretion_disk_rotation[0] * half), cx = cosf(bh->accretion_disk_rotation[0] * half);
    const float sy = sinf(bh->accretion_disk_rotation[1] * half), cy = cosf(bh->accretion_disk_rotation[1] * half);
    const float sz = sinf(bh->accretion_disk_rotation[2] * half), cz = cosf(bh->accretion_disk_rotation[2] * half);
    const float qs = ((-sx * sy) * sz) + ((cx * cy) * cz);
    const V3 qv = v3(((sx * cy) * cz) + ((sy * sz) * cx), ((-sx * sz) * cy) + ((sy * cx) * cz), ((sx * sy) * cz) + ((sz * cx) * cy));
    // Quaternion * Vector3: tmp = v x vec + vec*s ; (v x tmp)*2 + vec
    const V3 vec = v3(0.0f, -1.0f, 0.0f);
    const V3 tmp = cross(qv, vec) + vec * qs;
    const V3 up = normalize(cross(qv, tmp) * 2.0f + vec);
    const V3 right = cross(v3(0.0f, 0.0f, 1.0f), up);
    const V3 fwd = cross(right, up);
    const float m[12] = {right.x, right.y, right.z, 0.0f, up.x, up.y, up.z, 0.0f, fwd.x, fwd.y, fwd.z, 0.0f};
    memcpy(u->rotation_matrix, m, sizeof m);
    u->normal[0] = up.x; u->normal[1] = up.y; u->normal[2] = up.z;
}

int bhray_model_new(bhray_model** out) {
    if (!out) return BHRAY_E_INVALID;
    *out = new (std::nothrow) bhray_model();
    return *out ? BHRAY_OK : BHRAY_E_NOMEM;
}

void bhray_model_free(bhray_model* m) { delete m; }

int bhray_model_add_vertex(bhray_model* m, const float p[4]) {
    if (!m || !p) return BHRAY_E_INVALID;
    if (m->points.size() / 4 >= BHRAY_MAX_MODEL_VERTICES) return BHRAY_E_CAPACITY;
    m->points.insert(m->points.end(), p, p + 4);
    return BHRAY_OK;
}

int bhray_model_add_normal(bhray_model* m, const float n[4]) {
    if (!m || !n) return BHRAY_E_INVALID;
    if (m->normals.size() / 4 >= BHRAY_MAX_MODEL_VERTICES) return BHRAY_E_CAPACITY;
    m->normals.insert(m->normals.end(), n, n + 4);
    return BHRAY_OK;
}

int bhray_model_add_triangle(bhray_model* m, const bhray_triangle* t) {
    if (!m || !t) return BHRAY_E_INVALID;
    if (m->triangles.size() >= BHRAY_MAX_MODEL_VERTICES) return BHRAY_E_CAPACITY;
    m->triangles.push_back(*t);
    return BHRAY_OK;
}

int bhray_model_build_bvh(bhray_model* m) {
    if (!m) return BHRAY_E_INVALID;
    const size_t np = m->points.size() / 4;
    for (const bhray_triangle& t : m->triangles)
        if (t.p1 < 0 || t.p2 < 0 || t.p3 < 0 || (size_t)t.p1 >= np || (size_t)t.p2 >= np || (size_t)t.p3 >= np) return BHRAY_E_INVALID;
    m->nodes.clear();
    m->nodes.resize(1, bhray_node{{0, 0, 0}, 0, {0, 0, 0}, 0});
    m->bvh_lookup.resize(m->triangles.size());
    for (size_t i = 0; i < m->triangles.size(); i++) m->bvh_lookup[i] = (int32_t)i;
    m->nodes[0].left_child = 0;
    m->nodes[0].obj_count = (int32_t)m->triangles.size();
    m->nodes_used = 1;
    m->update_bounds(0);
    m->subdivide_from(0);
    m->nodes.resize(m->nodes_used);
    if (m->nodes_used > BHRAY_MAX_MODEL_VERTICES) return BHRAY_E_CAPACITY;
    return BHRAY_OK;
}

// Binned SAH builder (not the reference's algorithm; see include/bhray.h).  16 bins over the centroid bounds of the
// longest axis... of every axis, cost = area(L)*n(L) + area(R)*n(R); falls back to a median split when binning cannot
// separate the centroids; leaves hold <= 4 triangles.  Children are allocated adjacently and left subtree first, like
// the reference builder, so the device layout code is shared.
int bhray_model_build_bvh_sah(bhray_model* m) {
    if (!m) return BHRAY_E_INVALID;
    const size_t np = m->points.size() / 4;
    for (const bhray_triangle& t : m->triangles)
        if (t.p1 < 0 || t.p2 < 0 || t.p3 < 0 || (size_t)t.p1 >= np || (size_t)t.p2 >= np || (size_t)t.p3 >= np) return BHRAY_E_INVALID;
    const size_t T = m->triangles.size();
    m->nodes.clear();
    m->nodes.resize(1, bhray_node{{0, 0, 0}, 0, {0, 0, 0}, 0});
    m->bvh_lookup.resize(T);
    for (size_t i = 0; i < T; i++) m->bvh_lookup[i] = (int32_t)i;
    std::vector<float> cent(3 * T), tmin(3 * T), tmax(3 * T);
    for (size_t i = 0; i < T; i++) {
        const bhray_triangle& t = m->triangles[i];
        const float* p[3] = {&m->points[4 * (size_t)t.p1], &m->points[4 * (size_t)t.p2], &m->points[4 * (size_t)t.p3]};
        for (int a = 0; a < 3; a++) {
            cent[3 * i + a] = (p[0][a] + p[1][a] + p[2][a]) / 3.0f;
            tmin[3 * i + a] = fminf(p[0][a], fminf(p[1][a], p[2][a]));
            tmax[3 * i + a] = fmaxf(p[0][a], fmaxf(p[1][a], p[2][a]));
        }
    }
    m->nodes[0].left_child = 0;
    m->nodes[0].obj_count = (int32_t)T;
    m->nodes_used = 1;
    m->update_bounds(0);
    auto area = [](const float* lo, const float* hi) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
    };
    constexpr int BINS = 16;
    std::vector<size_t> todo;
    todo.push_back(0);
    while (!todo.empty()) {
        const size_t ni = todo.back();
        todo.pop_back();
        const int32_t first = m->nodes[ni].left_child, count = m->nodes[ni].obj_count;
        if (count <= 4) continue;
        float clo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, chi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
        for (int32_t i = 0; i < count; i++) {
            const size_t t = (size_t)m->bvh_lookup[(size_t)(first + i)];
            for (int a = 0; a < 3; a++) { clo[a] = fminf(clo[a], cent[3 * t + a]); chi[a] = fmaxf(chi[a], cent[3 * t + a]); }
        }
        int best_axis = -1, best_bin = -1;
        float best_cost = 3.4e38f;
        for (int a = 0; a < 3; a++) {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0.0f)) continue;
            int cnt[BINS] = {0};
            float blo[BINS][3], bhi[BINS][3];
            for (int b = 0; b < BINS; b++) for (int k = 0; k < 3; k++) { blo[b][k] = 3.4e38f; bhi[b][k] = -3.4e38f; }
            const float scale = (float)BINS / ext;
            for (int32_t i = 0; i < count; i++) {
                const size_t t = (size_t)m->bvh_lookup[(size_t)(first + i)];
                int b = (int)((cent[3 * t + a] - clo[a]) * scale);
                b = b < 0 ? 0 : (b > BINS - 1 ? BINS - 1 : b);
                cnt[b]++;
                for (int k = 0; k < 3; k++) { blo[b][k] = fminf(blo[b][k], tmin[3 * t + k]); bhi[b][k] = fmaxf(bhi[b][k], tmax[3 * t + k]); }
            }
            float llo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, lhi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
            float larea[BINS]; int lcnt[BINS]; int acc = 0;
            for (int b = 0; b < BINS - 1; b++) {
                for (int k = 0; k < 3; k++) { llo[k] = fminf(llo[k], blo[b][k]); lhi[k] = fmaxf(lhi[k], bhi[b][k]); }
                acc += cnt[b]; lcnt[b] = acc; larea[b] = area(llo, lhi);
            }
            float rlo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, rhi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
            acc = 0;
            for (int b = BINS - 1; b >= 1; b--) {
                for (int k = 0; k < 3; k++) { rlo[k] = fminf(rlo[k], blo[b][k]); rhi[k] = fmaxf(rhi[k], bhi[b][k]); }
                acc += cnt[b];
                if (lcnt[b - 1] == 0 || acc == 0) continue;
                const float cost = larea[b - 1] * (float)lcnt[b - 1] + area(rlo, rhi) * (float)acc;
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        int32_t mid;
        if (best_axis >= 0) {
            const float ext = chi[best_axis] - clo[best_axis], scale = (float)BINS / ext;
            int32_t i = first, j = first + count - 1;
            while (i <= j) {
                const size_t t = (size_t)m->bvh_lookup[(size_t)i];
                int b = (int)((cent[3 * t + best_axis] - clo[best_axis]) * scale);
                b = b < 0 ? 0 : (b > BINS - 1 ? BINS - 1 : b);
                if (b < best_bin) i++;
                else { std::swap(m->bvh_lookup[(size_t)i], m->bvh_lookup[(size_t)j]); j--; }
            }
            mid = i;
        } else {
            mid = first + count / 2;                       // all centroids coincide: split the index range
        }
        if (mid == first || mid == first + count) mid = first + count / 2;
        const size_t li = m->nodes_used, ri = m->nodes_used + 1;
        m->nodes_used += 2;
        if (m->nodes.size() < m->nodes_used) m->nodes.resize(m->nodes_used, bhray_node{{0, 0, 0}, 0, {0, 0, 0}, 0});
        m->nodes[li].left_child = first; m->nodes[li].obj_count = mid - first;
        m->nodes[ri].left_child = mid; m->nodes[ri].obj_count = first + count - mid;
        m->nodes[ni].left_child = (int32_t)li; m->nodes[ni].obj_count = 0;
        m->update_bounds(li); m->update_bounds(ri);
        todo.push_back(ri); todo.push_back(li);
    }
    m->nodes.resize(m->nodes_used);
    if (m->nodes_used > BHRAY_MAX_MODEL_VERTICES) return BHRAY_E_CAPACITY;
    return BHRAY_OK;
}

int bhray_model_max_depth(const bhray_model* m) {
    if (!m || m->nodes.empty()) return 0;
    int depth = 0;
    std::vector<std::pair<int32_t, int>> st;
    st.push_back({0, 1});
    while (!st.empty()) {
        auto [ni, d] = st.back();
        st.pop_back();
        if (d > depth) depth = d;
        const bhray_node& n = m->nodes[(size_t)ni];
        if (n.obj_count == 0 && m->nodes.size() > 1) { st.push_back({n.left_child, d + 1}); st.push_back({n.left_child + 1, d + 1}); }
    }
    return depth;
}

int bhray_model_desc_get(const bhray_model* m, bhray_model_desc* out) {
    if (!m || !out) return BHRAY_E_INVALID;
    memset(out, 0, sizeof *out);
    memcpy(out->position, m->position, 12);
    out->visible = m->visible;
    out->points = m->points.data(); out->normals = m->normals.data();
    out->triangles = m->triangles.data(); out->nodes = m->nodes.data(); out->bvh_lookup = m->bvh_lookup.data();
    out->point_count = (int32_t)(m->points.size() / 4); out->normal_count = (int32_t)(m->normals.size() / 4);
    out->triangle_count = (int32_t)m->triangles.size(); out->node_count = (int32_t)m->nodes.size();
    return BHRAY_OK;
}

int bhray_model_set_transform(bhray_model* m, const float position[3], int32_t visible) {
    if (!m || !position) return BHRAY_E_INVALID;
    memcpy(m->position, position, 12);
    m->visible = visible;
    return BHRAY_OK;
}

int bhray_model_pack_uniform(const bhray_model* m, void* dst, size_t size) {
    if (!m || !dst || size != BHRAY_MODEL_UNIFORM_BYTES) return BHRAY_E_INVALID;
    uint8_t* b = (uint8_t*)dst;
    memset(b, 0, size);
    bhray_model_header hd; memset(&hd, 0, sizeof hd);
    memcpy(hd.position, m->position, 12); hd.visible = m->visible; memcpy(hd.rotation, m->rotation, 12);
    hd.point_count = (int32_t)(m->points.size() / 4);
    hd.normal_count = 0;                                  // never copied by ModelUniform::update (triangle.rs:308-325)
    hd.triangle_count = (int32_t)m->triangles.size();
    memcpy(b, &hd, sizeof hd);
    if (!m->points.empty()) memcpy(b + BHRAY_MODEL_OFF_POINTS, m->points.data(), m->points.size() * 4);
    if (!m->normals.empty()) memcpy(b + BHRAY_MODEL_OFF_NORMALS, m->normals.data(), m->normals.size() * 4);
    if (!m->triangles.empty()) memcpy(b + BHRAY_MODEL_OFF_TRIANGLES, m->triangles.data(), m->triangles.size() * sizeof(bhray_triangle));
    if (!m->nodes.empty()) memcpy(b + BHRAY_MODEL_OFF_NODES, m->nodes.data(), m->nodes.size() * sizeof(bhray_node));
    if (!m->bvh_lookup.empty()) memcpy(b + BHRAY_MODEL_OFF_LOOKUP, m->bvh_lookup.data(), m->bvh_lookup.size() * 4);
    return BHRAY_OK;
}

// load_model, model.rs:7-87.  OBJ subset of tobj 4.0.2 with default LoadOptions (no
// triangulation, positions and normals indexed separately): `v`, `vn`, `f` with exactly three
// vertices written as a, a/t, a//n or a/t/n (1-based, negative = relative).
int bhray_load_model(const char* path, bhray_model** out) {
    if (!path || !out) return BHRAY_E_INVALID;
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return BHRAY_E_IO;
    std::vector<float> pos, nrm;
    std::vector<int32_t> idx, nidx;
    char line[1024];
    int rc = BHRAY_OK;
    while (fgets(line, sizeof line, f)) {
        char* hash = strchr(line, '#');
        if (hash) *hash = 0;
        char* s = line;
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            float a, b, c;
            if (sscanf(s + 2, "%f %f %f", &a, &b, &c) != 3) { rc = BHRAY_E_IO; break; }
            pos.push_back(a); pos.push_back(b); pos.push_back(c);
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            float a, b, c;
            if (sscanf(s + 3, "%f %f %f", &a, &b, &c) != 3) { rc = BHRAY_E_IO; break; }
            nrm.push_back(a); nrm.push_back(b); nrm.push_back(c);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            char* p = s + 2;
            int nv = 0;
            while (*p) {
                while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
                if (!*p) break;
                if (nv == 3) { nv = 4; break; }
                char* end;
                long a = strtol(p, &end, 10);
                if (end == p) { nv = -1; break; }
                long n = 0; bool has_n = false;
                p = end;
                if (*p == '/') {
                    p++;
                    if (*p != '/') { (void)strtol(p, &end, 10); p = end; }           // texcoord index ignored
                    if (*p == '/') { p++; n = strtol(p, &end, 10); if (end != p) has_n = true; p = end; }
                }
                a = a > 0 ? a - 1 : (long)(pos.size() / 3) + a;
                idx.push_back((int32_t)a);
                if (has_n) { n = n > 0 ? n - 1 : (long)(nrm.size() / 3) + n; nidx.push_back((int32_t)n); }
                nv++;
            }
            if (nv != 3) { rc = BHRAY_E_IO; break; }       // tobj default options do not triangulate
        }
    }
    fclose(f);
    if (rc) return rc;
    if (!nidx.empty() && nidx.size() != idx.size()) return BHRAY_E_IO;
    bhray_model* m = nullptr;
    rc = bhray_model_new(&m);
    if (rc) return rc;
    const int32_t mesh_offset = 0, normal_offset = 0;       // single object (model.rs:22-23)
    for (size_t i = 0; i < nrm.size() / 3 && !rc; i++) {
        const float n4[4] = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], 0.0f};
        rc = bhray_model_add_normal(m, n4);
    }
    for (size_t i = 0; i < pos.size() / 3 && !rc; i++) {
        const float p4[4] = {pos[3 * i] * 0.5f, pos[3 * i + 1] * -0.5f, pos[3 * i + 2] * 0.5f, 0.0f};   // model.rs:36-38
        rc = bhray_model_add_vertex(m, p4);
    }
    const size_t npts = pos.size() / 3;
    for (size_t i = 0; i < idx.size() / 3 && !rc; i++) {
        const int32_t p1 = idx[3 * i], p2 = idx[3 * i + 1], p3 = idx[3 * i + 2];
        if (p1 < 0 || p2 < 0 || p3 < 0 || (size_t)p1 >= npts || (size_t)p2 >= npts || (size_t)p3 >= npts) { rc = BHRAY_E_IO; break; }
        int32_t n1, n2, n3;
        if (!nidx.empty()) {
            n1 = nidx[3 * i]; n2 = nidx[3 * i + 1]; n3 = nidx[3 * i + 2];
            const int32_t nn = (int32_t)(nrm.size() / 3);
            if (n1 < 0 || n2 < 0 || n3 < 0 || n1 >= nn || n2 >= nn || n3 >= nn) { rc = BHRAY_E_IO; break; }
        } else {
            const V3 a = m->point(p1), b = m->point(p2), c = m->point(p3);
            const V3 dir = normalize(cross(b - a, c - a));                              // model.rs:59-63
            n1 = n2 = n3 = (int32_t)(m->normals.size() / 4);
            const float n4[4] = {dir.x, dir.y, dir.z, 0.0f};
            rc = bhray_model_add_normal(m, n4);
            if (rc) break;
        }
        const bhray_triangle t = {p1 + mesh_offset, p2 + mesh_offset, p3 + mesh_offset, n1 + normal_offset, n2 + normal_offset, n3 + normal_offset};
        rc = bhray_model_add_triangle(m, &t);
    }
    if (!rc) rc = bhray_model_build_bvh(m);
    if (rc) { bhray_model_free(m); return rc; }
    *out = m;
    return BHRAY_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// perlin/src/main.rs:1-148 — the disk-texture generator (offline asset tool of the reference)
// ---------------------------------------------------------------------------------------------------------------
namespace {

inline uint32_t rotl16(uint32_t v) { return (v << 16) | (v >> 16); }

// random_gradient, main.rs:6-24
inline void random_gradient(uint32_t ix, uint32_t iy, float& gx, float& gy) {
    uint32_t a = ix, b = iy;
    a *= 3284157443u;
    b ^= rotl16(a);
    b *= 1911520717u;
    a ^= rotl16(b);
    a *= 2048419325u;
    const float random = (float)a * (3.14159265358979323846f / (float)0xFFFFFFFFu);   // PI / (!(0u32 >> 1)) as f32
    gx = cosf(random); gy = sinf(random);
}
inline float dot_grid_gradient(uint32_t ix, uint32_t iy, float x, float y) {            // main.rs:26-33
    float gx, gy; random_gradient(ix, iy, gx, gy);
    const float dx = x - (float)ix, dy = y - (float)iy;
    return dx * gx + dy * gy;
}
inline float interpolate(float a0, float a1, float w) {                                  // main.rs:35-38
    return (a1 - a0) * ((w * (w * 6.0f - 15.0f) + 10.0f) * w * w * w) + a0;
}
inline float perlin(float x, float y) {                                                  // main.rs:40-58
    const uint32_t x0 = (uint32_t)floorf(x), x1 = x0 + 1, y0 = (uint32_t)floorf(y), y1 = y0 + 1;
    const float sx = x - (float)x0, sy = y - (float)y0;
    const float ix0 = interpolate(dot_grid_gradient(x0, y0, x, y), dot_grid_gradient(x1, y0, x, y), sx);
    const float ix1 = interpolate(dot_grid_gradient(x0, y1, x, y), dot_grid_gradient(x1, y1, x, y), sx);
    return interpolate(ix0, ix1, sy) * 0.5f + 0.5f;
}
inline uint8_t as_u8(float v) { return !(v == v) ? 0 : (v <= 0.0f ? 0 : (v >= 255.0f ? 255 : (uint8_t)v)); }            // Rust `as u8`
inline uint32_t as_u32(float v) { return !(v == v) ? 0u : (v <= 0.0f ? 0u : (v >= 4294967295.0f ? 0xFFFFFFFFu : (uint32_t)v)); }

// value[x * h + y] like ImageBuffer::put_pixel(x, y)
std::vector<uint8_t> generate(uint32_t w, uint32_t h, uint32_t density) {                // main.rs:61-77
    std::vector<uint8_t> buf((size_t)w * h);
    const float d = (float)density / (float)w;
    for (uint32_t x = 0; x < w; x++)
        for (uint32_t y = 0; y < h; y++) buf[(size_t)x * h + y] = as_u8(perlin((float)x * d, (float)y * d) * 256.0f);
    return buf;
}
std::vector<uint8_t> spiral(const std::vector<uint8_t>& buf, uint32_t w, uint32_t h, float amount, float power) {      // main.rs:79-110
    std::vector<uint8_t> out((size_t)w * h);
    const float PI = 3.14159265358979323846f;
    for (uint32_t x = 0; x < w; x++)
        for (uint32_t y = 0; y < h; y++) {
            float rx = ((float)x / (float)w) * 2.0f - 1.0f, ry = ((float)y / (float)h) * 2.0f - 1.0f;
            const float r = sqrtf(rx * rx + ry * ry);
            float theta = atan2f(ry, rx);
            theta = fmodf(theta + PI + powf(r, power) * PI * amount, 2.0f * PI) - PI;
            rx = r * cosf(theta); ry = r * sinf(theta);
            const uint32_t nx = as_u32((rx * 0.5f + 0.5f) * (float)w) % w, ny = as_u32((ry * 0.5f + 0.5f) * (float)h) % h;
            out[(size_t)x * h + y] = buf[(size_t)nx * h + ny];
        }
    return out;
}
std::vector<uint8_t> merge(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b, float amount) {              // main.rs:113-131
    std::vector<uint8_t> out(a.size());
    for (size_t i = 0; i < a.size(); i++) out[i] = as_u8((float)a[i] * amount + (float)b[i] * (1.0f - amount));
    return out;
}

}  // namespace

extern "C" int bhray_generate_disk_texture(uint32_t size, uint8_t* rgba8_out) {          // main(), main.rs:133-147
    if (!rgba8_out || size < 2 || size > 16384) return BHRAY_E_INVALID;
    const uint32_t dens[4] = {4, 20, 50, 100};
    std::vector<uint8_t> sp[4];
    for (int i = 0; i < 4; i++) sp[i] = spiral(generate(size, size, dens[i]), size, size, 2.0f, 0.5f);
    const std::vector<uint8_t> m3 = merge(merge(merge(sp[3], sp[2], 0.5f), sp[1], 0.5f), sp[0], 0.5f);
    for (uint32_t y = 0; y < size; y++)
        for (uint32_t x = 0; x < size; x++) {
            const uint8_t v = m3[(size_t)x * size + y];
            uint8_t* o = rgba8_out + 4 * ((size_t)y * size + x);
            o[0] = v; o[1] = v; o[2] = v; o[3] = v;
        }
    return BHRAY_OK;
}
