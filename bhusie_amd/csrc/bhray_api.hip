// bhray_api.hip — the C ABI of include/bhray.h on top of the gfx950 kernels.
//
// Plays the role of RayPipeline::{new,pass,output_view}
// (/root/reference/src/renderer/pipelines/ray_pipeline.rs:36-309) and of the ladder / per-frame
// upload code in Renderer::{new,render} (src/renderer/mod.rs:113-207, 378-420).
// All device memory is owned here; there is no CPU rendering path.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <cmath>
#include <string>
#include <vector>

#include "bhray_dev.h"
#include "bhray_internal.h"
#include "bhray_math.h"

using namespace bhray;

namespace {

// Frame slots run on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, so 4+ slots
// serialise pairwise unless the HOST raises the limit before the HIP runtime initialises (include/bhray.h, frames in flight).
// The library does not touch the process environment.

thread_local std::string g_create_error;

#define BHRAY_READ_RING 64
constexpr int SPAN_MAX = BHRAY_MAX_LEVELS + 2;      // trace launches per batch (per-level, + speculative / predicted)

#ifndef BHRAY_ROW_VARIANTS
#define BHRAY_ROW_VARIANTS 9         // orderings of a level's rows kept on the device (centre rows at 0, 1/8, ... 1 of the height); < 2 = ascending only
#endif
struct Level {                      // geometry of one ladder level (shared by all frame slots)
    int w = 0, h = 0;
    std::vector<int32_t> rows;      // rows to compute
    int32_t* d_rows = nullptr;
    int32_t* d_rows_near = nullptr; // BHRAY_ROW_VARIANTS orderings of the same rows, tile rows nearest a centre row first (variant v: centre v / (N-1) of the level's height)
    int32_t* d_rowmap = nullptr;    // final level only
    size_t queue_cap = 0;           // entries the level's queue can need: its rows x its columns
    size_t queue_alloc = 0;         // entries allocated per frame (>= queue_cap; grows when the partition changes: dev_set_partition)
};

// Resources of one frame: level images, work queues and output buffer.
struct FrameRes {
    std::vector<float4*> level_out;     // [levels-1] full-size images of the non-final levels
    std::vector<uint32_t*> queue;       // [levels]
    std::vector<float4*> spec_out;      // speculative mode: traced images of levels 1..S-1 (level 0 traces straight into level_out[0])
    uint32_t* spec_queue = nullptr;     // speculative mode: merged, level-tagged queue of levels 0..S-1
    uint32_t* super_queue = nullptr;    // superset speculation: merged, level-tagged queue of the last U levels
    // temporal speculation (BHRAY_F_TEMPORAL): the pixels the previous frame held here had to trace, all levels, level-tagged
    uint32_t* pred_queue = nullptr;     // the predicted launch's queue: built by predict_kernel from the previous frame's marks
    uint32_t* pred_ctl = nullptr;       // [0] entries [1] entries taken: the control words of level BHRAY_MAX_LEVELS-1 in d_qctl (temporal mode has <= 4 levels), reset with them
    std::vector<uint8_t*> need;         // per level: need[y * w + x] = the last exact classification had to trace that pixel
    std::vector<uint32_t*> stamp;       // per level: stamp[y * w + x] == stamp_value <=> the predicted launch traced that pixel this frame
    uint32_t stamp_value = 0;

    uint32_t* d_qctl = nullptr;         // [BHRAY_QCTL_WORDS]: qcount[l], qhead[l], then the work counters   (a slice of Slot::d_qctl)
    unsigned long long* d_work = nullptr;   // [BHRAY_WORK_WORDS] inside that slice: integrator steps the trace waves issued for the frame held here (FrameLaunch::work)
    Counters64* d_counters = nullptr;   // [BHRAY_MAX_LEVELS]                         (a slice of Slot::d_counters)
    unsigned long long* d_row_work = nullptr;   // BHRAY_F_COUNTERS: iterations per level row, level l at bhray_dev::row_work_off[l] (bhray_get_row_work)
    float4* own_out = nullptr;
    float4* out = nullptr;              // where this frame is written (own_out or a bound buffer)
    uint2* sky_out = nullptr;           // RGBA16F image of the sky resolve pass (allocated on first use)
    uint64_t sky_frame_id = ~0ull;      // frame_id of the frame sky_out was resolved from (a slot position is reused: an older frame's image is stale)
    uint64_t frame_id = 0;              // frame_counter value of the frame held here
    int row_variant = -1;               // which ordering of the level rows this frame classifies in (Level::d_rows_near), -1: ascending
};

// One batch in flight: a HIP stream, the frames of the batch (frames_per_batch of them) and the argument block of its
// launches.  dev_render stages a frame (host work only); the launches of the whole batch are enqueued when the batch is
// full or flushed, each launch covering all staged frames.
struct Slot {
    hipStream_t stream = nullptr;
    bool owns_stream = true;            // false: the stream belongs to an earlier slot (more slots than hardware queues)
    hipEvent_t done = nullptr;          // recorded after the slot's last launch
    hipEvent_t uploaded = nullptr;      // recorded after the argument block of the slot's batch has been copied to the device
    bool used = false;                  // `uploaded` has been recorded at least once
    std::vector<FrameRes> fr;
    uint32_t pending = 0;               // frames staged and not launched yet
    int method = 0; bool models = false;   // kernel variant of the staged frames (a batch is homogeneous)
    uint32_t* d_qctl = nullptr;         // [frames_per_batch][BHRAY_QCTL_WORDS]
    uint32_t launched_frames = 0;       // frames of the batch launched last from this slot (their work counters are valid once it has completed)
    Counters64* d_counters = nullptr;   // [frames_per_batch][BHRAY_MAX_LEVELS]
    uint8_t* h_args = nullptr;          // pinned staging of the argument block: FrameParams[B], then FrameLaunch[B] per launch
    uint8_t* d_args = nullptr;
    size_t args_cap = 0;
    uint64_t batch_id = 0;              // batch_counter value of the batch this slot holds
};

struct ModelStore {
    float pos[3] = {0, 0, 0};
    int visible = 0;
    float4* points = nullptr; float4* normals = nullptr; int32_t* triangles = nullptr;
    float4* nodes = nullptr; int32_t* lookup = nullptr; float4* leaf = nullptr;
    int point_count = 0, normal_count = 0, triangle_count = 0, node_count = 0;
    int root_cull = 0; float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};   // union of the root's child boxes (ModelDev)
    bool loaded = false;
};

}  // namespace

struct bhray_dev {
    bhray_config cfg{};
    int device = 0;
    std::vector<Level> levels;
    std::vector<Slot> slots;               // batches in flight
    uint32_t batch = 1;                    // frames per batch
    int last_slot = 0, last_sub = 0;       // slot / position in its batch of the most recently rendered frame
    uint64_t batch_counter = 0;            // batches launched so far; the staging slot is batch_counter % slots
    uint64_t retired = 0;                  // batches known to have completed (polled oldest-first at every launch): batch_counter - retired are in flight
    int dynamic_dense = -1;                // BHRAY_DYNAMIC_DENSE=n overrides the policy below (0: by the slot count; n > 0: by the batches in flight now, threshold n)
    bhray::DevOptions opt;
    float4* bound_out = nullptr;           // dev_bind_output: destination of the next frame (one-shot)
    std::vector<hipEvent_t> waits;         // dev_wait_event: pending dependencies of the next render (events owned by the caller)
    int launched_slot = -1;                // dev_take_launched
    uint32_t launched_frames = 0;
    size_t out_bytes = 0;
    size_t out_alloc = 0;                  // bytes of every frame's own output buffer (>= out_bytes)
    size_t spec_alloc = 0, pred_alloc = 0, super_alloc = 0;   // entries of the merged queues (speculative / temporal / superset modes)
    std::vector<uint32_t> local_rows;      // frame rows of this partition, increasing
    size_t row_work_off[BHRAY_MAX_LEVELS + 1] = {0};   // BHRAY_F_COUNTERS: offset of every level's rows in FrameRes::d_row_work; [levels] = total
    // scene
    uint8_t* tex[3] = {nullptr, nullptr, nullptr};
    int tex_w[3] = {0, 0, 0}, tex_h[3] = {0, 0, 0};
    ModelStore models[BHRAY_MAX_MODELS];
    bhray_camera_uniform cam{};
    bhray_black_hole_uniform bh{};
    bhray_details det{};
    bool have_uniforms = false;
    std::vector<hipEvent_t> events;        // ring: [BHRAY_TIMING_RING][3 * levels + 4]: per level (before classify, before trace, after trace), 2 around the sky pass, 2 around the temporal mode's prediction + predicted trace
    uint64_t frame_counter = 0, timing_begin = 0;   // timing_begin: first batch not yet reported by dev_get_timing
    uint8_t sky_recorded[BHRAY_TIMING_RING] = {0};
    uint8_t ring_frames[BHRAY_TIMING_RING] = {0};   // frames of the batch held by each timing-ring entry
    // execution spans of the trace launches of timed batches (FrameLaunch::span): [BHRAY_TIMING_RING][SPAN_MAX][2] device words
    unsigned long long* d_span = nullptr;
    uint8_t ring_spans[BHRAY_TIMING_RING] = {0};    // trace launches of each timing-ring entry
    double wall_clock_khz = 100000.0;               // hipDeviceAttributeWallClockRate (s_memrealtime: 100 MHz on gfx950)
    int* d_err = nullptr;
    int num_cus = 256;
    // BHRAY_F_TEMPORAL: what is predicted beyond the pixels the previous frame traced (measured at 1080p on a camera orbiting at 0.002
    // and 0.02 rad per frame, profiles/r02_experiments.json "temporal_prediction"): interpolated pixels whose widest neighbour angle
    // exceeded 0.8 x the threshold, pixels within 4 (levels below the last) / 1 (last level) pixels of a predicted one, and the clamped
    // border pixels (predict_kernel).  +8 % rays; the levels below the last then miss nothing, the last level a few hundred aliased
    // pixels: ONE fix-up launch instead of three.  BHRAY_TEMPORAL_MARGIN / BHRAY_TEMPORAL_RADIUS="last[,below]" override (tuning).
    float temporal_margin = 0.8f;
    uint32_t temporal_radius_coarse = 4;
    uint32_t temporal_radius = 1;
                                           // radius of 1-2 adds 2-4 % rays and does not shorten a moving camera's frames - the misses are scattered interpolate/trace flips)
    int bpc_override = 0;                  // BHRAY_TRACE_BLOCKS_PER_CU (tuning experiments only)
    int grid_override = 0;                 // BHRAY_TRACE_GRID: absolute number of persistent trace blocks (tuning experiments only)
    int wave_prio = -1;                    // BHRAY_PRIO=0/1 forces wave priority off / on for every latency-build launch; -1 = one frame per launch and ONE frame slot
    int quad_wps = -1;                     // BHRAY_QUAD: waves per SIMD the quad march may use for a short queue (bhray_quad.inc); 0 = the scalar thin shares only; -1 = the measured default (RK 2, Euler 1)
    int dense_override = -1;               // BHRAY_TRACE_DENSE=0/1 (tuning experiments only)
    int coarse_build = -1;                 // BHRAY_COARSE_BUILD=0/1 (experiment): the build of the trace launches below the ladder's last level (-1: the batch's build)
    bool rendered = false;
    // asynchronous hand-off (dev_read_hdr_async)
    hipEvent_t read_ev[BHRAY_READ_RING] = {nullptr};     // ticket t -> read_ev[t % BHRAY_READ_RING]
    uint64_t read_tickets = 0;
    std::string err;
};

namespace {

int fail(bhray_dev* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(ctx, BHRAY_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

}  // namespace
// rows of level k-1 that the rows `fine` of level k read (ray.wgsl:185-201), same binary32 math
std::vector<int32_t> bhray::coarse_rows_needed(const std::vector<int32_t>& fine, int h, int ph) {
    std::vector<uint8_t> need((size_t)ph, 0);
    if (ph == 1) return {};
    const int sf = (h - 1) / (ph - 1);
    const float ry = (float)ph / (float)(h + (sf - 1));
    for (int32_t y : fine) {
        const float ppy = (float)y * ry;
        int tl = (int)floorf(ppy);
        int a = tl < 0 ? 0 : (tl > ph - 1 ? ph - 1 : tl);
        int b = tl + 1; b = b < 0 ? 0 : (b > ph - 1 ? ph - 1 : b);
        need[(size_t)a] = 1; need[(size_t)b] = 1;
    }
    std::vector<int32_t> out;
    for (int y = 0; y < ph; y++) if (need[(size_t)y]) out.push_back(y);
    return out;
}
namespace {

// bh_acos(c) < thr  <=>  c > acos_threshold(thr)  for c in [-1, 1]: bh_acos is monotone non-increasing over all binary32 values
// of the interval (exhaustive check: dev_selftest), so the set where the predicate holds is an upper interval; its lower end is
// found by bisection over the ordered bit patterns with the same bh_acos the kernels use.
float acos_threshold(float thr) {
    if (!(thr == thr)) return INFINITY;                              // NaN threshold: never smaller
    if (!(bh_acos(1.0f) < thr)) return INFINITY;                     // not even angle 0 is below the threshold
    if (bh_acos(-1.0f) < thr) return u2f(0xbf800001u);               // every c of [-1, 1] is: c > (largest float below -1)
    // ordered keys: k < 0 <-> x = -|bits|, k >= 0 <-> x = +bits; x(k) increasing in k
    auto xk = [](int64_t k) { return k >= 0 ? u2f((uint32_t)k) : u2f(0x80000000u | (uint32_t)(-k)); };
    int64_t lo = -(int64_t)0x3f800000, hi = (int64_t)0x3f800000;     // predicate false at lo (x = -1), true at hi (x = 1)
    while (hi - lo > 1) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (bh_acos(xk(mid)) < thr) hi = mid; else lo = mid;
    }
    return xk(lo);                                                   // the largest c whose angle is not below the threshold
}

void derive_frame(const bhray_dev* c, FrameParams& P) {
    memset(&P, 0, sizeof P);
    const bhray_camera_uniform& cam = c->cam;
    const bhray_black_hole_uniform& bh = c->bh;
    const bhray_details& d = c->det;
    const F3 fwd = ld3(cam.forward);
    const F3 plane_up = f3(0.0f, -1.0f, 0.0f);
    const F3 right = normalize(cross(fwd, plane_up));               // ray.wgsl:276
    const F3 up = normalize(cross(fwd, right));                     // ray.wgsl:277
    const float fov_factor = 1.0f / bh_tan(cam.fov / 2.0f);         // ray.wgsl:279
    const F3 fwd_ff = fwd * fov_factor;
    const F3 cpos = ld3(cam.position), bpos = ld3(bh.position);
    P.cam[0] = cpos.x; P.cam[1] = cpos.y; P.cam[2] = cpos.z;
    P.right[0] = right.x; P.right[1] = right.y; P.right[2] = right.z;
    P.up[0] = up.x; P.up[1] = up.y; P.up[2] = up.z;
    P.fwd_ff[0] = fwd_ff.x; P.fwd_ff[1] = fwd_ff.y; P.fwd_ff[2] = fwd_ff.z;
    P.ray_distance = distance(cpos, bpos);
    P.ray_distance_f = fdistance(cpos, bpos);
    P.relativity0 = P.ray_distance < bh.relativity_sphere_radius ? 1 : 0;
    P.bh[0] = bpos.x; P.bh[1] = bpos.y; P.bh[2] = bpos.z;
    memcpy(P.bn, bh.normal, 12);
    P.bn_len = length(ld3(bh.normal));
    P.cull_outer_pad = bh.accretion_disk_outer + 0.0501f; P.cull_plane_c1 = 1.0101f * P.bn_len; P.cull_plane_c2 = 1.01e-4f * P.bn_len;
    P.inner = bh.accretion_disk_inner; P.outer = bh.accretion_disk_outer;
    P.rot_speed = bh.rotation_speed; P.R = bh.relativity_sphere_radius;
    P.show_tex = bh.show_disk_texture; P.show_shift = bh.show_red_shift;
    for (int col = 0; col < 3; col++) for (int r = 0; r < 3; r++) P.M[3 * col + r] = bh.rotation_matrix[4 * col + r];
    P.feather = bh.feather_amount;
    P.time = d.time; P.time_rot = d.time * bh.rotation_speed; P.method = d.integration_method != 0 ? 1 : 0; P.step_size = d.step_size;
    P.max_iter = d.max_iterations; P.thr = d.angle_division_threshold;
    P.acos_cstar = acos_threshold(P.thr);
    P.acos_cstar_near = c->temporal_margin < 1.0f ? acos_threshold(P.thr * c->temporal_margin) : P.acos_cstar;
    int mc = d.model_count; if (mc < 0) mc = 0; if (mc > BHRAY_MAX_MODELS) mc = BHRAY_MAX_MODELS;
    int usable = 0;
    for (int i = 0; i < mc; i++) {
        const ModelStore& m = c->models[i];
        ModelDev& md = P.models[i];
        memcpy(md.pos, m.pos, 12);
        md.visible = (m.loaded && m.triangle_count > 0) ? m.visible : 0;
        md.points = m.points; md.normals = m.normals; md.triangles = m.triangles; md.nodes = m.nodes; md.lookup = m.lookup; md.leaf = m.leaf;
        md.node_count = m.node_count;
        md.root_cull = m.root_cull; memcpy(md.root_lo, m.root_lo, 12); memcpy(md.root_hi, m.root_hi, 12);
        usable = i + 1;
    }
    P.model_count = usable;
    bool any_visible = false;
    for (int i = 0; i < usable; i++) any_visible |= P.models[i].visible != 0;
    if (!any_visible) P.model_count = 0;     // nothing to traverse: the no-mesh kernel variant is exact
    TexDev* t[3] = {&P.temp, &P.disk, &P.sky};
    for (int i = 0; i < 3; i++) { t[i]->rgba = c->tex[i]; t[i]->w = c->tex_w[i]; t[i]->h = c->tex_h[i]; }
}

int launch_batch(bhray_dev* c);

hipError_t sync_all(bhray_dev* c) {
    for (Slot& S : c->slots) {
        hipError_t e = hipStreamSynchronize(S.stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

void free_model(ModelStore& m) {
    if (m.points) (void)hipFree(m.points);
    if (m.normals) (void)hipFree(m.normals);
    if (m.triangles) (void)hipFree(m.triangles);
    if (m.nodes) (void)hipFree(m.nodes);
    if (m.lookup) (void)hipFree(m.lookup);
    if (m.leaf) (void)hipFree(m.leaf);
    m = ModelStore();
}

}  // namespace

extern "C" {

const char* bhray_strerror(int code) {
    switch (code) {
        case BHRAY_OK: return "ok";
        case BHRAY_E_INVALID: return "invalid argument";
        case BHRAY_E_NO_DEVICE: return "no usable HIP device";
        case BHRAY_E_HIP: return "HIP runtime error";
        case BHRAY_E_NOMEM: return "out of memory";
        case BHRAY_E_STATE: return "call order violated";
        case BHRAY_E_BVH_DEPTH: return "BVH deeper than the traversal stack";
        case BHRAY_E_IO: return "I/O or parse error";
        case BHRAY_E_CAPACITY: return "model exceeds reference capacity";
        case BHRAY_E_COMM: return "RCCL unavailable or collective failed";
        default: return "unknown error";
    }
}

uint32_t bhray_version(void) { return (BHRAY_VERSION_MAJOR << 16) | BHRAY_VERSION_MINOR; }


int bhray_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bhray_ladder_from_base(uint32_t base_w, uint32_t base_h, uint32_t m, uint32_t levels, bhray_config* cfg) {
    if (!cfg || levels < 1 || levels > BHRAY_MAX_LEVELS || base_w < 2 || base_h < 2 || m < 2) return BHRAY_E_INVALID;
    uint64_t w = base_w, h = base_h;
    for (uint32_t i = 0; i < levels; i++) {
        if (w > 32767 || h > 32767) return BHRAY_E_INVALID;          // queue entries pack tag<<30 | y<<15 | x
        cfg->level_w[i] = (uint32_t)w; cfg->level_h[i] = (uint32_t)h;
        w = w * m - (m - 1); h = h * m - (m - 1);                    // mod.rs:203-204
    }
    cfg->levels = levels;
    cfg->crop_x = 0; cfg->crop_y = 0;
    cfg->frame_w = cfg->level_w[levels - 1]; cfg->frame_h = cfg->level_h[levels - 1];
    if (cfg->row_world == 0) { cfg->row_world = 1; cfg->row_rank = 0; }
    if (cfg->stripe_rows == 0) cfg->stripe_rows = 27;
    cfg->struct_size = sizeof(bhray_config);
    return BHRAY_OK;
}

int bhray_ladder_for_frame(uint32_t frame_w, uint32_t frame_h, uint32_t m, uint32_t levels, bhray_config* cfg) {
    if (!cfg || levels < 1 || levels > BHRAY_MAX_LEVELS || frame_w < 2 || frame_h < 2 || m < 2) return BHRAY_E_INVALID;
    uint64_t s = 1;
    for (uint32_t i = 1; i < levels; i++) s *= m;                    // last = (base-1)*m^(levels-1) + 1
    const uint64_t bw = (frame_w - 1 + s - 1) / s + 1, bh_ = (frame_h - 1 + s - 1) / s + 1;
    int rc = bhray_ladder_from_base((uint32_t)(bw < 2 ? 2 : bw), (uint32_t)(bh_ < 2 ? 2 : bh_), m, levels, cfg);
    if (rc) return rc;
    const uint32_t lw = cfg->level_w[levels - 1], lh = cfg->level_h[levels - 1];
    cfg->crop_x = (lw - frame_w) / 2; cfg->crop_y = (lh - frame_h) / 2;
    cfg->frame_w = frame_w; cfg->frame_h = frame_h;
    return BHRAY_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// per-device engine (bhray_dev.h): one partition of the frame on one GPU.  The public bhray_ctx (bhray_group.hip) owns one
// of these per local partition.
// ------------------------------------------------------------------------------------------
const char* dev_last_error(const bhray_dev* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
void dev_set_create_error(const char* msg) { g_create_error = msg ? msg : ""; }

void dev_destroy(bhray_dev* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (Slot& S : c->slots) if (S.stream) (void)hipStreamSynchronize(S.stream);
    for (Slot& S : c->slots) {
        for (FrameRes& R : S.fr) {
            for (auto p : R.level_out) if (p) (void)hipFree(p);
            for (auto p : R.queue) if (p) (void)hipFree(p);
            for (auto p : R.spec_out) if (p) (void)hipFree(p);
            if (R.spec_queue) (void)hipFree(R.spec_queue);
            if (R.super_queue) (void)hipFree(R.super_queue);
            if (R.pred_queue) (void)hipFree(R.pred_queue);
            for (auto p : R.need) if (p) (void)hipFree(p);
            for (auto p : R.stamp) if (p) (void)hipFree(p);
            if (R.own_out) (void)hipFree(R.own_out);
            if (R.d_row_work) (void)hipFree(R.d_row_work);
            if (R.sky_out) (void)hipFree(R.sky_out);
        }
        if (S.d_qctl) (void)hipFree(S.d_qctl);
        if (S.d_counters) (void)hipFree(S.d_counters);
        if (S.h_args) (void)hipHostFree(S.h_args);
        if (S.d_args) (void)hipFree(S.d_args);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.uploaded) (void)hipEventDestroy(S.uploaded);
        if (S.stream && S.owns_stream) (void)hipStreamDestroy(S.stream);
    }
    for (Level& L : c->levels) {
        if (L.d_rows) (void)hipFree(L.d_rows);
        if (L.d_rows_near) (void)hipFree(L.d_rows_near);
        if (L.d_rowmap) (void)hipFree(L.d_rowmap);
    }
    for (auto& t : c->tex) if (t) (void)hipFree(t);
    for (auto& m : c->models) free_model(m);
    for (auto& e : c->events) if (e) (void)hipEventDestroy(e);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_span) (void)hipFree(c->d_span);
    for (auto& e : c->read_ev) if (e) (void)hipEventDestroy(e);
    delete c;
}

namespace {
// The partition's rows (bhray_config.partition of c->cfg), the level rows they depend on (top-down, ray.wgsl:185-201), and their
// device tables.  The tables are allocated for a whole level, so a partition set later (dev_set_partition) rewrites them in place.
int build_row_tables(bhray_dev* c) {
    const bhray_config* cfg = &c->cfg;
    const uint32_t nl = cfg->levels;
    c->local_rows = partition_row_list(*cfg, cfg->row_world, cfg->row_rank);
    for (uint32_t l = 0; l < nl; l++) c->levels[l].rows.clear();
    {
        Level& F = c->levels[nl - 1];
        for (uint32_t r : c->local_rows) F.rows.push_back((int32_t)(cfg->crop_y + r));
        for (int l = (int)nl - 1; l > 0; l--)
            c->levels[l - 1].rows = coarse_rows_needed(c->levels[l].rows, c->levels[l].h, c->levels[l - 1].h);
    }
    for (uint32_t l = 0; l < nl; l++) {
        Level& L = c->levels[l];
        const bool last = (l == nl - 1);
        const size_t nrows = L.rows.size();
        if (!L.d_rows) HIPCHK(c, hipMalloc(&L.d_rows, (size_t)L.h * sizeof(int32_t)));
        if (nrows) {
            HIPCHK(c, hipMemcpy(L.d_rows, L.rows.data(), nrows * sizeof(int32_t), hipMemcpyHostToDevice));
            if (BHRAY_ROW_VARIANTS >= 2) {
                // The same rows with the tile rows (8 list entries) nearest a centre row first: the rays that pass closest to the hole are the
                // longest, and a launch lasts as long as its last rays - classified first they enter the queue first and are traced first
                // (every pixel is classified independently of the others: the order changes nothing else).  One ordering per centre row;
                // a frame picks the one nearest the row the hole projects to (dev_render).
                const size_t nch = nrows / 8;                         // whole tile rows; a short last one stays last
                std::vector<int32_t> all((size_t)BHRAY_ROW_VARIANTS * nrows);
                for (int v = 0; v < BHRAY_ROW_VARIANTS; v++) {
                    const double centre = (double)v / (double)(BHRAY_ROW_VARIANTS - 1) * (double)(L.h - 1);
                    std::vector<size_t> ch(nch);
                    for (size_t k = 0; k < nch; k++) ch[k] = k;
                    std::stable_sort(ch.begin(), ch.end(), [&](size_t a, size_t b) {
                        return fabs((double)L.rows[a * 8 + 4] - centre) < fabs((double)L.rows[b * 8 + 4] - centre);
                    });
                    int32_t* o = all.data() + (size_t)v * nrows;
                    size_t n = 0;
                    for (size_t k : ch) for (size_t j = k * 8; j < k * 8 + 8; j++) o[n++] = L.rows[j];
                    for (size_t j = nch * 8; j < nrows; j++) o[n++] = L.rows[j];
                }
                if (!L.d_rows_near) HIPCHK(c, hipMalloc(&L.d_rows_near, (size_t)BHRAY_ROW_VARIANTS * (size_t)L.h * sizeof(int32_t)));
                HIPCHK(c, hipMemcpy(L.d_rows_near, all.data(), all.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            }
        }
        const size_t span = last ? cfg->frame_w : (size_t)L.w;
        L.queue_cap = nrows * span;
        if (last) {
            std::vector<int32_t> map((size_t)L.h, -1);
            for (size_t i = 0; i < c->local_rows.size(); i++) map[(size_t)(cfg->crop_y + c->local_rows[i])] = c->opt.frame_rowmap ? (int32_t)c->local_rows[i] : (int32_t)i;
            if (!L.d_rowmap) HIPCHK(c, hipMalloc(&L.d_rowmap, (size_t)L.h * sizeof(int32_t)));
            HIPCHK(c, hipMemcpy(L.d_rowmap, map.data(), (size_t)L.h * sizeof(int32_t), hipMemcpyHostToDevice));
        }
    }
    c->out_bytes = (c->opt.frame_rowmap ? (size_t)cfg->frame_h : c->local_rows.size()) * (size_t)cfg->frame_w * sizeof(float4);
    return BHRAY_OK;
}

// Every frame's ray queues and own output buffer, sized for the current partition.  `headroom` (a partition set at run time): a buffer
// that has to grow grows by a quarter more than needed, so that bounds that keep moving a few rows do not reallocate every time.
int ensure_frame_buffers(bhray_dev* c, bool headroom) {
    const bhray_config* cfg = &c->cfg;
    const uint32_t nl = cfg->levels;
    auto want = [&](size_t need, size_t full) { size_t w = headroom ? need + need / 4 + 64 : need; return w > full ? std::max(full, need) : w; };
    for (uint32_t l = 0; l < nl; l++) {
        Level& L = c->levels[l];
        if (L.queue_cap <= L.queue_alloc) continue;
        const size_t full = (size_t)L.h * (size_t)(l == nl - 1 ? cfg->frame_w : (uint32_t)L.w);
        const size_t n = want(L.queue_cap, full);
        for (Slot& S : c->slots) for (FrameRes& R : S.fr) {
            if (R.queue[l]) { HIPCHK(c, hipFree(R.queue[l])); R.queue[l] = nullptr; }
            HIPCHK(c, hipMalloc(&R.queue[l], n * sizeof(uint32_t)));
        }
        L.queue_alloc = n;
    }
    auto merged = [&](size_t need, size_t& alloc, uint32_t* FrameRes::*member) -> int {
        if (need <= alloc) return BHRAY_OK;
        const size_t n = headroom ? need + need / 4 + 64 : need;
        for (Slot& S : c->slots) for (FrameRes& R : S.fr) {
            if (R.*member) { HIPCHK(c, hipFree(R.*member)); R.*member = nullptr; }
            HIPCHK(c, hipMalloc(&(R.*member), n * sizeof(uint32_t)));
        }
        alloc = n;
        return BHRAY_OK;
    };
    if (cfg->speculative_levels) {
        size_t cap = 0;
        for (uint32_t l = 0; l < cfg->speculative_levels; l++) cap += c->levels[l].queue_cap;
        int rc = merged(cap, c->spec_alloc, &FrameRes::spec_queue); if (rc) return rc;
    }
    if (cfg->flags & BHRAY_F_TEMPORAL) {
        size_t cap = 0;
        for (uint32_t l = 0; l < nl; l++) cap += c->levels[l].queue_cap;
        int rc = merged(cap, c->pred_alloc, &FrameRes::pred_queue); if (rc) return rc;
    }
    if (cfg->superset_levels) {
        size_t cap = 0;
        for (uint32_t l = nl - cfg->superset_levels; l < nl; l++) cap += c->levels[l].queue_cap;
        int rc = merged(cap, c->super_alloc, &FrameRes::super_queue); if (rc) return rc;
    }
    if (!c->opt.external_out && c->out_bytes > c->out_alloc) {
        const size_t full = (size_t)cfg->frame_h * (size_t)cfg->frame_w * sizeof(float4);
        size_t n = headroom ? c->out_bytes + c->out_bytes / 4 : c->out_bytes;
        if (n > full) n = std::max(full, c->out_bytes);
        for (Slot& S : c->slots) for (FrameRes& R : S.fr) {
            if (R.own_out) { HIPCHK(c, hipFree(R.own_out)); R.own_out = nullptr; }
            HIPCHK(c, hipMalloc(&R.own_out, n));
            HIPCHK(c, hipMemset(R.own_out, 0xFF, n));
            R.out = R.own_out;
        }
        c->out_alloc = n;
    }
    return BHRAY_OK;
}
}  // namespace

int dev_create(const bhray_config* cfg, const bhray::DevOptions& opt, bhray_dev** out) {
    if (!cfg || !out) return fail(nullptr, BHRAY_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(bhray_config)) return fail(nullptr, BHRAY_E_INVALID, "bhray_config.struct_size mismatch");
    if (cfg->levels < 1 || cfg->levels > BHRAY_MAX_LEVELS) return fail(nullptr, BHRAY_E_INVALID, "levels out of range");
    for (uint32_t i = 0; i < cfg->levels; i++)
        if (cfg->level_w[i] < 2 || cfg->level_h[i] < 2 || cfg->level_w[i] > 32767 || cfg->level_h[i] > 32767)
            return fail(nullptr, BHRAY_E_INVALID, "level %u size %ux%u unsupported", i, cfg->level_w[i], cfg->level_h[i]);
    const uint32_t lw = cfg->level_w[cfg->levels - 1], lh = cfg->level_h[cfg->levels - 1];
    if (cfg->frame_w < 1 || cfg->frame_h < 1 || cfg->crop_x + cfg->frame_w > lw || cfg->crop_y + cfg->frame_h > lh)
        return fail(nullptr, BHRAY_E_INVALID, "frame window outside the last level");
    if (cfg->row_world < 1 || cfg->row_rank >= cfg->row_world)
        return fail(nullptr, BHRAY_E_INVALID, "bad row partition");
    if (const char* why = partition_error(*cfg, cfg->row_world)) return fail(nullptr, BHRAY_E_INVALID, "bad row partition: %s", why);
    if (cfg->frames_in_flight > BHRAY_MAX_FRAMES_IN_FLIGHT) return fail(nullptr, BHRAY_E_INVALID, "frames_in_flight > %d", BHRAY_MAX_FRAMES_IN_FLIGHT);
    if (cfg->frames_per_batch > BHRAY_MAX_FRAMES_PER_BATCH) return fail(nullptr, BHRAY_E_INVALID, "frames_per_batch > %d", BHRAY_MAX_FRAMES_PER_BATCH);
    if (cfg->speculative_levels == 1 || cfg->speculative_levels > BHRAY_MAX_SPEC_LEVELS || (cfg->speculative_levels && cfg->speculative_levels >= cfg->levels))
        return fail(nullptr, BHRAY_E_INVALID, "speculative_levels must be 0 or 2..min(%d, levels-1)", BHRAY_MAX_SPEC_LEVELS);
    if (cfg->superset_levels == 1 || cfg->superset_levels > BHRAY_MAX_SPEC_LEVELS ||
        (cfg->superset_levels && cfg->superset_levels + (cfg->speculative_levels ? cfg->speculative_levels : 1) > cfg->levels))
        return fail(nullptr, BHRAY_E_INVALID, "superset_levels must be 0 or 2..%d and leave at least one coarser level (beyond the speculative ones)", BHRAY_MAX_SPEC_LEVELS);
    if ((cfg->flags & BHRAY_F_TEMPORAL) && (cfg->levels > BHRAY_MAX_SPEC_LEVELS || cfg->speculative_levels || cfg->superset_levels))
        return fail(nullptr, BHRAY_E_INVALID, "BHRAY_F_TEMPORAL needs levels <= %d and no speculative / superset levels", BHRAY_MAX_SPEC_LEVELS);
    if (cfg->flags & ~(uint32_t)BHRAY_F_ALL)
        return fail(nullptr, BHRAY_E_INVALID, "unknown bits in bhray_config.flags (0x%x; bit 6 was BHRAY_F_FUSED, the fused ladder of rounds 3-5: measured slower, removed - profiles/variants_src/)", cfg->flags & ~(uint32_t)BHRAY_F_ALL);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, BHRAY_E_NO_DEVICE, "no HIP device visible (libbhray has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, BHRAY_E_NO_DEVICE, "device %d not present (%d visible)", cfg->device, ndev);

    bhray_dev* c = new (std::nothrow) bhray_dev();
    if (!c) return fail(nullptr, BHRAY_E_NOMEM, "host allocation failed");
    c->cfg = *cfg;
    c->device = cfg->device;
    c->opt = opt;
#define CHK(call)                                                                                             \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            int rc_ = fail(nullptr, BHRAY_E_HIP, "%s: %s", #call, hipGetErrorString(e_));                     \
            dev_destroy(c);                                                                                 \
            return rc_;                                                                                       \
        }                                                                                                     \
    } while (0)
    CHK(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, c->device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("BHRAY_TRACE_BLOCKS_PER_CU")) c->bpc_override = atoi(e);
    if (const char* e = getenv("BHRAY_TRACE_GRID")) c->grid_override = atoi(e);
    if (const char* e = getenv("BHRAY_PRIO")) c->wave_prio = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("BHRAY_QUAD")) { const int q = atoi(e); c->quad_wps = q < 0 ? 0 : (q > 4 ? 4 : q); }
    if (const char* e = getenv("BHRAY_TEMPORAL_MARGIN")) { const float m = (float)atof(e); c->temporal_margin = m < 0.05f ? 0.05f : (m > 1.0f ? 1.0f : m); }
    if (const char* e = getenv("BHRAY_TEMPORAL_RADIUS")) {          // "fine[,coarse]": the last level, the levels below it
        int rf = 0, rc = -1;
        const int n = sscanf(e, "%d,%d", &rf, &rc);
        if (n < 2) rc = rf;
        c->temporal_radius = rf < 0 ? 0 : (rf > 4 ? 4 : (uint32_t)rf);
        c->temporal_radius_coarse = rc < 0 ? 0 : (rc > 4 ? 4 : (uint32_t)rc);
    }
    if (const char* e = getenv("BHRAY_TRACE_DENSE")) c->dense_override = atoi(e) != 0;
    if (const char* e = getenv("BHRAY_COARSE_BUILD")) c->coarse_build = atoi(e);
    if (const char* e = getenv("BHRAY_DYNAMIC_DENSE")) c->dynamic_dense = atoi(e);
    const uint32_t nslots = cfg->frames_in_flight ? cfg->frames_in_flight : 4;
    c->cfg.frames_in_flight = nslots;
    c->slots.resize(nslots);
    c->batch = cfg->frames_per_batch ? cfg->frames_per_batch : 1;
    c->cfg.frames_per_batch = c->batch;

    const uint32_t nl = cfg->levels;
    c->levels.resize(nl);
    for (uint32_t l = 0; l < nl; l++) { c->levels[l].w = (int)cfg->level_w[l]; c->levels[l].h = (int)cfg->level_h[l]; }
    for (uint32_t l = 0; l < nl; l++) c->row_work_off[l + 1] = c->row_work_off[l] + (size_t)cfg->level_h[l];
    // rows of the frame owned by this partition, the level rows they depend on, their device tables
    { int rc_ = build_row_tables(c); if (rc_) { g_create_error = c->err; dev_destroy(c); return rc_; } }
    const size_t nlaunch = 5 * (size_t)nl + 3;                            // upper bound of launches per batch
    // Streams beyond the hardware queues ROCm maps them onto (GPU_MAX_HW_QUEUES, default 4; two are left to the null stream and a
    // communication stream) do not add concurrency, they alias - and a device with MORE streams than queues collapses (24 slots on 24
    // queues: 5 580 -> 4 380 Mrays/s, and a 20-frame block from 8.5 to 73 ms, measured).  The library only READS the variable: the slots
    // beyond the limit share the streams of the first ones (two frames on one stream are simply in order).
    size_t max_streams = 2;
    { const char* e = getenv("GPU_MAX_HW_QUEUES"); const int q = e ? atoi(e) : 4; max_streams = q > 3 ? (size_t)(q - 2) : 2; }
    if (opt.max_streams && max_streams > opt.max_streams) max_streams = opt.max_streams;
    for (size_t si = 0; si < c->slots.size(); si++) {
        Slot& S = c->slots[si];
        if (si < max_streams) CHK(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
        else { S.stream = c->slots[si % max_streams].stream; S.owns_stream = false; }
        CHK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
        CHK(hipEventCreateWithFlags(&S.uploaded, hipEventDisableTiming));
        const size_t B = c->batch;
        CHK(hipMalloc(&S.d_qctl, B * BHRAY_QCTL_WORDS * sizeof(uint32_t)));
        CHK(hipMemset(S.d_qctl, 0, B * BHRAY_QCTL_WORDS * sizeof(uint32_t)));
        CHK(hipMalloc(&S.d_counters, B * BHRAY_MAX_LEVELS * sizeof(Counters64)));
        CHK(hipMemset(S.d_counters, 0, B * BHRAY_MAX_LEVELS * sizeof(Counters64)));
        S.args_cap = (B * (sizeof(FrameParams) + nlaunch * sizeof(FrameLaunch) + 16) + 15) & ~(size_t)15;
        CHK(hipHostMalloc((void**)&S.h_args, S.args_cap, hipHostMallocDefault));
        CHK(hipMalloc(&S.d_args, S.args_cap));
        S.fr.resize(B);
        for (size_t k = 0; k < B; k++) {
            FrameRes& R = S.fr[k];
            R.d_qctl = S.d_qctl + k * BHRAY_QCTL_WORDS;
            R.d_work = reinterpret_cast<unsigned long long*>(R.d_qctl + 2 * BHRAY_MAX_LEVELS);
            R.d_counters = S.d_counters + k * BHRAY_MAX_LEVELS;
            R.level_out.assign(nl > 0 ? nl - 1 : 0, nullptr);
            R.queue.assign(nl, nullptr);
            for (uint32_t l = 0; l < nl; l++) {
                const Level& L = c->levels[l];
                if (l + 1 < nl) {
                    const size_t npix = (size_t)L.w * (size_t)L.h;
                    CHK(hipMalloc(&R.level_out[l], npix * sizeof(float4)));
                    CHK(hipMemset(R.level_out[l], 0xFF, npix * sizeof(float4)));      // NaN: "never rendered"
                }
            }
            if (cfg->speculative_levels) {
                const uint32_t ns = cfg->speculative_levels;
                R.spec_out.assign(ns, nullptr);
                for (uint32_t l = 0; l < ns; l++) {
                    const Level& L = c->levels[l];
                    if (l > 0) {
                        const size_t npix = (size_t)L.w * (size_t)L.h;
                        CHK(hipMalloc(&R.spec_out[l], npix * sizeof(float4)));
                        CHK(hipMemset(R.spec_out[l], 0xFF, npix * sizeof(float4)));
                    }
                }
            }
            if (cfg->flags & BHRAY_F_TEMPORAL) {
                R.stamp.assign(nl, nullptr); R.need.assign(nl, nullptr);
                for (uint32_t l = 0; l < nl; l++) {
                    const Level& L = c->levels[l];
                    const size_t npix = (size_t)L.w * (size_t)L.h;
                    CHK(hipMalloc(&R.stamp[l], npix * sizeof(uint32_t)));
                    CHK(hipMemset(R.stamp[l], 0, npix * sizeof(uint32_t)));
                    CHK(hipMalloc(&R.need[l], npix));
                    CHK(hipMemset(R.need[l], 0, npix));
                }
                static_assert(BHRAY_MAX_SPEC_LEVELS < BHRAY_MAX_LEVELS, "the predicted queue borrows the last level's control words");
                R.pred_ctl = R.d_qctl + 2 * (BHRAY_MAX_LEVELS - 1);
            }
            if (cfg->flags & BHRAY_F_COUNTERS) {
                CHK(hipMalloc(&R.d_row_work, c->row_work_off[nl] * sizeof(unsigned long long)));
                CHK(hipMemset(R.d_row_work, 0, c->row_work_off[nl] * sizeof(unsigned long long)));
            }
            R.out = R.own_out;
        }
    }
    // ray queues and own output buffers of every frame, sized for the partition (grown by dev_set_partition when it changes)
    { int rc_ = ensure_frame_buffers(c, false); if (rc_) { g_create_error = c->err; dev_destroy(c); return rc_; } }
    if (cfg->flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE)) {
        c->events.assign((size_t)BHRAY_TIMING_RING * (nl * 3 + 4), nullptr);
        for (auto& e : c->events) CHK(hipEventCreate(&e));
        CHK(hipMalloc(&c->d_span, (size_t)BHRAY_TIMING_RING * SPAN_MAX * 2 * sizeof(unsigned long long)));
        CHK(hipMemset(c->d_span, 0, (size_t)BHRAY_TIMING_RING * SPAN_MAX * 2 * sizeof(unsigned long long)));
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) c->wall_clock_khz = (double)khz;
    }
    CHK(hipMalloc(&c->d_err, sizeof(int)));
    CHK(hipMemset(c->d_err, 0, sizeof(int)));
    // 1x1 opaque-black defaults so a missing texture cannot fault
    for (int s = 0; s < 3; s++) {
        const uint8_t px[4] = {0, 0, 0, 255};
        CHK(hipMalloc(&c->tex[s], 4));
        CHK(hipMemcpy(c->tex[s], px, 4, hipMemcpyHostToDevice));
        c->tex_w[s] = 1; c->tex_h[s] = 1;
    }
#undef CHK
    *out = c;
    return BHRAY_OK;
}

int dev_set_texture(bhray_dev* c, int slot, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!c) return BHRAY_E_INVALID;
    if (slot < 0 || slot > 2 || !rgba8 || w < 1 || h < 1 || w > 32768 || h > 32768) return fail(c, BHRAY_E_INVALID, "bad texture arguments");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }          // staged frames hold the old texture's address
    HIPCHK(c, sync_all(c));
    uint8_t* d = nullptr;
    const size_t bytes = (size_t)w * h * 4;
    HIPCHK(c, hipMalloc(&d, bytes));
    hipError_t e = hipMemcpy(d, rgba8, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return fail(c, BHRAY_E_HIP, "texture upload: %s", hipGetErrorString(e)); }
    if (c->tex[slot]) (void)hipFree(c->tex[slot]);
    c->tex[slot] = d; c->tex_w[slot] = (int)w; c->tex_h[slot] = (int)h;
    return BHRAY_OK;
}

int dev_upload_model(bhray_dev* c, uint32_t mi, const bhray_model_desc* d) {
    if (!c) return BHRAY_E_INVALID;
    if (mi >= BHRAY_MAX_MODELS || !d) return fail(c, BHRAY_E_INVALID, "bad model arguments");
    if (d->point_count < 0 || d->normal_count < 0 || d->triangle_count < 0 || d->node_count < 0 ||
        d->point_count > BHRAY_MAX_MODEL_VERTICES || d->normal_count > BHRAY_MAX_MODEL_VERTICES ||
        d->triangle_count > BHRAY_MAX_MODEL_VERTICES || d->node_count > BHRAY_MAX_MODEL_VERTICES)
        return fail(c, BHRAY_E_CAPACITY, "model exceeds MAX_MODEL_VERTICES (triangle.rs:7)");
    if (d->triangle_count > 0 && (!d->points || !d->normals || !d->triangles || !d->nodes || !d->bvh_lookup || d->node_count < 1))
        return fail(c, BHRAY_E_INVALID, "model arrays missing");
    // validate indices so the kernel never reads out of bounds
    for (int i = 0; i < d->triangle_count; i++) {
        const bhray_triangle& t = d->triangles[i];
        if (t.p1 < 0 || t.p2 < 0 || t.p3 < 0 || t.p1 >= d->point_count || t.p2 >= d->point_count || t.p3 >= d->point_count ||
            t.n1 < 0 || t.n2 < 0 || t.n3 < 0 || t.n1 >= d->normal_count || t.n2 >= d->normal_count || t.n3 >= d->normal_count)
            return fail(c, BHRAY_E_INVALID, "triangle %d has an index out of range", i);
        if (d->bvh_lookup[i] < 0 || d->bvh_lookup[i] >= d->triangle_count) return fail(c, BHRAY_E_INVALID, "bvh_lookup[%d] out of range", i);
    }
    for (int i = 0; i < d->node_count && d->triangle_count > 0; i++) {     // an empty model's root is (0 children, 0 objects): never traversed
        const bhray_node& n = d->nodes[i];
        if (n.obj_count < 0) return fail(c, BHRAY_E_INVALID, "node %d: negative obj_count", i);
        if (n.obj_count == 0) {
            if (n.left_child < 1 || n.left_child + 1 >= d->node_count)
                return fail(c, BHRAY_E_INVALID, "node %d: children out of range", i);
        } else if (n.left_child < 0 || (int64_t)n.left_child + n.obj_count > d->triangle_count) {
            return fail(c, BHRAY_E_INVALID, "node %d: leaf range out of bounds", i);
        }
    }
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }          // staged frames hold the old model's addresses
    HIPCHK(c, sync_all(c));
    ModelStore& m = c->models[mi];
    free_model(m);
    memcpy(m.pos, d->position, 12);
    m.visible = d->visible;
    m.point_count = d->point_count; m.normal_count = d->normal_count; m.triangle_count = d->triangle_count; m.node_count = d->node_count;
    if (d->triangle_count > 0) {
        HIPCHK(c, hipMalloc(&m.points, (size_t)d->point_count * 16));
        HIPCHK(c, hipMalloc(&m.normals, (size_t)d->normal_count * 16));
        HIPCHK(c, hipMalloc(&m.triangles, (size_t)d->triangle_count * 24));
        HIPCHK(c, hipMalloc(&m.nodes, (size_t)d->node_count * 32));
        HIPCHK(c, hipMalloc(&m.lookup, (size_t)d->triangle_count * 4));
        HIPCHK(c, hipMemcpy(m.points, d->points, (size_t)d->point_count * 16, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(m.normals, d->normals, (size_t)d->normal_count * 16, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(m.triangles, d->triangles, (size_t)d->triangle_count * 24, hipMemcpyHostToDevice));
        // Nodes are re-numbered breadth-first for the device (the builder numbers them depth-first, triangle.rs:239-258):
        // the top of the tree, which every traversal touches, is then contiguous (cache locality).  Children stay adjacent
        // and leaf ranges are untouched, so traversal order and results do not change.
        {
            std::vector<bhray_node> bfs((size_t)d->node_count);
            std::vector<int32_t> order; order.reserve((size_t)d->node_count);
            order.push_back(0);
            size_t next = 1;
            for (size_t q = 0; q < order.size(); q++) {
                const bhray_node& n = d->nodes[(size_t)order[q]];
                bfs[q] = n;
                if (n.obj_count == 0) {
                    if (next + 2 > (size_t)d->node_count) return fail(c, BHRAY_E_INVALID, "malformed BVH (unreachable or shared nodes)");
                    bfs[q].left_child = (int32_t)next;
                    order.push_back(n.left_child); order.push_back(n.left_child + 1);
                    next += 2;
                }
            }
            m.node_count = (int)order.size();
            // what the traversal's first visit tests: the two children of the (untested) root.  A ray that misses the union
            // of their boxes misses both (the slab test is monotone in the box), so the kernel can skip that visit.
            if (bfs[0].obj_count == 0 && order.size() >= 3) {
                for (int a = 0; a < 3; a++) {
                    m.root_lo[a] = bfs[1].min_corner[a] < bfs[2].min_corner[a] ? bfs[1].min_corner[a] : bfs[2].min_corner[a];
                    m.root_hi[a] = bfs[1].max_corner[a] > bfs[2].max_corner[a] ? bfs[1].max_corner[a] : bfs[2].max_corner[a];
                }
                bool finite = true;
                for (int a = 0; a < 3; a++) finite = finite && std::isfinite(m.root_lo[a]) && std::isfinite(m.root_hi[a]);
                m.root_cull = finite ? 1 : 0;
            }
            HIPCHK(c, hipMemcpy(m.nodes, bfs.data(), order.size() * 32, hipMemcpyHostToDevice));
        }
        HIPCHK(c, hipMemcpy(m.lookup, d->bvh_lookup, (size_t)d->triangle_count * 4, hipMemcpyHostToDevice));
        // pre-gathered leaf geometry: slot k holds the points and normals of triangle bvh_lookup[k] (copies, same bits)
        {
            std::vector<float> leaf((size_t)d->triangle_count * 24);
            for (int k = 0; k < d->triangle_count; k++) {
                const bhray_triangle& t = d->triangles[d->bvh_lookup[k]];
                const int32_t pi[3] = {t.p1, t.p2, t.p3}, ni[3] = {t.n1, t.n2, t.n3};
                for (int v = 0; v < 3; v++) {
                    memcpy(&leaf[(size_t)k * 24 + 4 * v], d->points + 4 * (size_t)pi[v], 16);
                    memcpy(&leaf[(size_t)k * 24 + 12 + 4 * v], d->normals + 4 * (size_t)ni[v], 16);
                }
            }
            HIPCHK(c, hipMalloc(&m.leaf, leaf.size() * 4));
            HIPCHK(c, hipMemcpy(m.leaf, leaf.data(), leaf.size() * 4, hipMemcpyHostToDevice));
        }
    }
    m.loaded = true;
    return BHRAY_OK;
}

int dev_upload_model_uniform(bhray_dev* c, uint32_t mi, const void* bytes, size_t size) {
    if (!c) return BHRAY_E_INVALID;
    if (!bytes || size != BHRAY_MODEL_UNIFORM_BYTES) return fail(c, BHRAY_E_INVALID, "ModelUniform must be %u bytes", BHRAY_MODEL_UNIFORM_BYTES);
    const uint8_t* b = (const uint8_t*)bytes;
    bhray_model_header hd; memcpy(&hd, b, sizeof hd);
    bhray_model_desc d; memset(&d, 0, sizeof d);
    memcpy(d.position, hd.position, 12); d.visible = hd.visible;
    d.points = (const float*)(b + BHRAY_MODEL_OFF_POINTS);
    d.normals = (const float*)(b + BHRAY_MODEL_OFF_NORMALS);
    d.triangles = (const bhray_triangle*)(b + BHRAY_MODEL_OFF_TRIANGLES);
    d.nodes = (const bhray_node*)(b + BHRAY_MODEL_OFF_NODES);
    d.bvh_lookup = (const int32_t*)(b + BHRAY_MODEL_OFF_LOOKUP);
    d.point_count = hd.point_count; d.triangle_count = hd.triangle_count;
    if (d.triangle_count < 0 || d.triangle_count > BHRAY_MAX_MODEL_VERTICES || d.point_count < 0 || d.point_count > BHRAY_MAX_MODEL_VERTICES)
        return fail(c, BHRAY_E_CAPACITY, "counts in ModelUniform header out of range");
    // ModelUniform::update never copies normal_count (triangle.rs:308-325) and carries no node
    // count: recover both from the index data.
    int nmax = -1;
    for (int i = 0; i < d.triangle_count; i++) {
        const bhray_triangle& t = d.triangles[i];
        nmax = t.n1 > nmax ? t.n1 : nmax; nmax = t.n2 > nmax ? t.n2 : nmax; nmax = t.n3 > nmax ? t.n3 : nmax;
    }
    if (nmax >= BHRAY_MAX_MODEL_VERTICES) return fail(c, BHRAY_E_INVALID, "normal index out of range");
    d.normal_count = nmax + 1;
    int node_max = 0;
    if (d.triangle_count > 0) {
        std::vector<int> st; st.push_back(0);
        size_t visited = 0;
        while (!st.empty()) {
            int n = st.back(); st.pop_back();
            if (n < 0 || n >= BHRAY_MAX_MODEL_VERTICES || ++visited > (size_t)BHRAY_MAX_MODEL_VERTICES) return fail(c, BHRAY_E_INVALID, "malformed BVH");
            node_max = n > node_max ? n : node_max;
            if (d.nodes[n].obj_count == 0) { st.push_back(d.nodes[n].left_child); st.push_back(d.nodes[n].left_child + 1); }
        }
    }
    d.node_count = d.triangle_count > 0 ? node_max + 1 : 0;
    return dev_upload_model(c, mi, &d);
}

int dev_set_model_transform(bhray_dev* c, uint32_t mi, const float position[3], int32_t visible) {
    if (!c) return BHRAY_E_INVALID;
    if (mi >= BHRAY_MAX_MODELS || !position) return fail(c, BHRAY_E_INVALID, "bad model arguments");
    memcpy(c->models[mi].pos, position, 12);
    c->models[mi].visible = visible;
    return BHRAY_OK;
}

int dev_set_uniforms(bhray_dev* c, const void* cam32, const void* bh132, const void* det32) {
    if (!c) return BHRAY_E_INVALID;
    if (!cam32 || !bh132 || !det32) return fail(c, BHRAY_E_INVALID, "null uniform block");
    static_assert(sizeof(bhray_camera_uniform) == 32 && sizeof(bhray_black_hole_uniform) == 132 && sizeof(bhray_details) == 32, "layout");
    memcpy(&c->cam, cam32, 32); memcpy(&c->bh, bh132, 132); memcpy(&c->det, det32, 32);
    c->have_uniforms = true;
    return BHRAY_OK;
}

// Another row partition for this engine (bhray_set_partition): same frame, same ladder, other rows.  Everything enqueued so far is
// waited for; the row tables are rewritten in place, the per-frame queues and output buffers grow if the new rows need more.  The level
// images, textures, models and uniforms stay.  The temporal mode's per-pixel history stays too (rows that are new to this partition
// have none: their first frame is the plain ladder's).
int dev_set_partition(bhray_dev* c, uint32_t partition, uint32_t stripe_rows, const uint32_t* slab_row0, uint32_t row_rank, uint32_t row_world) {
    if (!c) return BHRAY_E_INVALID;
    bhray_config n = c->cfg;
    n.partition = partition; n.row_rank = row_rank; n.row_world = row_world;
    if (stripe_rows) n.stripe_rows = stripe_rows;
    if (partition == BHRAY_PARTITION_SLABS) {
        if (!slab_row0 || row_world > BHRAY_MAX_DEVICES) return fail(c, BHRAY_E_INVALID, "bad row partition");
        for (uint32_t p = 0; p <= row_world; p++) n.slab_row0[p] = slab_row0[p];
    }
    if (row_world < 1 || row_rank >= row_world) return fail(c, BHRAY_E_INVALID, "bad row partition");
    if (const char* why = partition_error(n, row_world)) return fail(c, BHRAY_E_INVALID, "bad row partition: %s", why);
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }
    HIPCHK(c, sync_all(c));
    c->cfg = n;
    for (Slot& S : c->slots) S.launched_frames = 0;           // the work the slots' frames counted belongs to the rows this engine had then (dev_get_work)
    { int rc = build_row_tables(c); if (rc) return rc; }
    return ensure_frame_buffers(c, true);
}

// ------------------------------------------------------------------------------------------
// One batch = the frames staged in a slot.  BatchPlan lays out the argument block (FrameParams[B], then one FrameLaunch[nb] array per
// launch, in the slot's pinned staging) and the launch sequence, one method per ladder mode; launch_batch enqueues it.
// ------------------------------------------------------------------------------------------
namespace {
struct Launch { int kind; const FrameLaunch* d; int blocks; bool count; std::vector<int> ev_before, ev_after; int build = -1; bool fixup = false; int levels = 1; FrameLaunch* h = nullptr; };   // kind 0 classify, 1 trace; timing events recorded around it; build: -1 the ctx's trace build, 0 latency, 1 dense

struct BatchPlan {
    bhray_dev* c;
    Slot& S;
    const uint32_t nb, nl;             // frames staged, ladder levels
    const bool count;                  // BHRAY_F_COUNTERS
    const int literal;                 // the integrator's evaluation (launch_trace's `eval`)
    const int grid;                    // persistent trace blocks of the ctx's trace build
    hipStream_t st;
    size_t args_used;
    std::vector<Launch> seq;
    uint32_t first_normal = 0;         // first level that still needs its own classify + trace pair

    // argument block of the next launch: nb FrameLaunch entries on the host, and their device address
    void next_launch(FrameLaunch*& h, const FrameLaunch*& d) {
        h = (FrameLaunch*)(S.h_args + args_used); d = (const FrameLaunch*)(S.d_args + args_used);
        args_used += (size_t)nb * sizeof(FrameLaunch);
        memset(h, 0, (size_t)nb * sizeof(FrameLaunch));
    }
    void level_params(const FrameRes& R, uint32_t l, LevelParams& L) const {
        const Level& Lv = c->levels[l];
        const bool last = (l == nl - 1);
        memset(&L, 0, sizeof L);
        L.w = Lv.w; L.h = Lv.h;
        if (l == 0) { L.pw = 1; L.ph = 1; L.rx = 1.0f; L.ry = 1.0f; L.prev = nullptr; }
        else {
            const Level& Pv = c->levels[l - 1];
            L.pw = Pv.w; L.ph = Pv.h; L.prev = R.level_out[l - 1];
            const int sfx = (L.w - 1) / (L.pw - 1), sfy = (L.h - 1) / (L.ph - 1);          // ray.wgsl:185
            L.rx = (float)L.pw / (float)(L.w + (sfx - 1)); L.ry = (float)L.ph / (float)(L.h + (sfy - 1));   // ray.wgsl:187
        }
        if (last) {
            L.out = R.out; L.out_pitch = (int)c->cfg.frame_w; L.out_x0 = (int)c->cfg.crop_x; L.rowmap = Lv.d_rowmap;
            L.x0 = (int)c->cfg.crop_x; L.x1 = (int)(c->cfg.crop_x + c->cfg.frame_w);
        } else {
            L.out = R.level_out[l]; L.out_pitch = Lv.w; L.out_x0 = 0; L.rowmap = nullptr; L.x0 = 0; L.x1 = Lv.w;
        }
        L.rows = (Lv.d_rows_near && R.row_variant >= 0) ? Lv.d_rows_near + (size_t)R.row_variant * Lv.rows.size() : Lv.d_rows;
        L.nrows = (int)Lv.rows.size();
    }
    int classify_blocks(uint32_t l) const {
        const Level& Lv = c->levels[l];
        const int span = (l == nl - 1) ? (int)c->cfg.frame_w : Lv.w;
        const int tiles_x = (span + 7) / 8, tiles_y = ((int)Lv.rows.size() + 7) / 8;
        return ((tiles_x + BHRAY_CLASSIFY_BX - 1) / BHRAY_CLASSIFY_BX) * ((tiles_y + BHRAY_CLASSIFY_BY - 1) / BHRAY_CLASSIFY_BY);
    }

    // speculative_levels = ns: every needed pixel of levels 0..ns-1 traced in ONE launch, then classified
    int speculative(uint32_t ns) {
        // (1) every needed pixel of levels 0..ns-1 into ONE level-tagged queue, (2) one trace launch over it,
        // (3) classify levels 1..ns-1 against the traced images.  Queue control words of level 0 serve the merged queue.
        for (uint32_t l = 0; l < ns; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].L.pw = 1; h[k].L.ph = 1; h[k].L.prev = nullptr; h[k].L.tag = (int)l;    // "base case": every pixel is traced
                h[k].queue = R.spec_queue; h[k].qctl = R.d_qctl; h[k].counters = nullptr;
            }
            seq.push_back({0, d, classify_blocks(l), false, {}, {}});
            if (l == 0) seq.back().ev_before = {0};
            if (l == ns - 1) seq.back().ev_after = {1};
        }
        {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, 0, h[k].L);
                h[k].SL.n = (int)ns;
                for (uint32_t l = 0; l < ns; l++) {
                    const Level& Lv = c->levels[l];
                    h[k].SL.l[l].w = Lv.w; h[k].SL.l[l].h = Lv.h; h[k].SL.l[l].out = (l == 0) ? R.level_out[0] : R.spec_out[l]; h[k].SL.l[l].out_pitch = Lv.w;
                    h[k].SL.l[l].row_work = (count && R.d_row_work) ? R.d_row_work + c->row_work_off[l] : nullptr;
                }
                h[k].queue = R.spec_queue; h[k].qctl = R.d_qctl; h[k].counters = count ? R.d_counters : nullptr;
            }
            seq.push_back({1, d, grid, count, {}, {2}});
            if (ns < nl) seq.back().build = c->coarse_build;
        }
        for (uint32_t l = 1; l < ns; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].L.spec = R.spec_out[l];
                h[k].queue = nullptr; h[k].qctl = R.d_qctl + 2 * l; h[k].counters = count ? R.d_counters + l : nullptr;
            }
            seq.push_back({0, d, classify_blocks(l), count, {(int)(3 * l)}, {(int)(3 * l + 1), (int)(3 * l + 2)}});   // no trace launch of its own
        }
        first_normal = ns;
        return BHRAY_OK;
    }

    // BHRAY_F_TEMPORAL: the previous frame's traced set in ONE launch, then the ladder fixes up what that prediction missed
    int temporal() {
        const bool shared_device = c->slots.size() > 1;     // other frames' launches share the device with this batch's
        // Temporal speculation: ONE launch traces, for every level, the pixels the previous frame held in this slot position had
        // to trace (its exact classification recorded them); then the ladder runs as usual, except that a pixel that needs tracing
        // and was delivered by the predicted launch (stamp) is not traced again.  With a perfect prediction the per-level trace
        // launches find empty queues and the chain of dependent long rays collapses into one launch; with no prediction (first
        // frame, scene cut) this is the plain ladder.  Pixels are identical in every case.
        for (uint32_t k = 0; k < nb; k++) {
            FrameRes& R = S.fr[k];
            R.stamp_value++;
            if (R.stamp_value == 0) {                          // 32-bit stamps wrapped: start over with clean images
                for (uint32_t l = 0; l < nl; l++) HIPCHK(c, hipMemsetAsync(R.stamp[l], 0, (size_t)c->levels[l].w * (size_t)c->levels[l].h * sizeof(uint32_t), st));
                R.stamp_value = 1;
            }
        }
        // (0) the prediction: per level, the previous frame's traced set dilated by temporal_radius pixels -> one level-tagged queue
        const FrameLaunch* pred_first = nullptr; int pred_blocks = 0;
        for (uint32_t l = 0; l < nl; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].L.tag = (int)l;
                h[k].queue = R.pred_queue; h[k].qctl = R.pred_ctl; h[k].need = R.need[l]; h[k].radius = (int)(l + 1 == nl ? c->temporal_radius : c->temporal_radius_coarse);
            }
            for (uint32_t k = 0; k < nb; k++) h[k].blocks = classify_blocks(l);
            if (l == 0) { pred_first = d; pred_blocks = 0; }
            if (classify_blocks(l) > pred_blocks) pred_blocks = classify_blocks(l);
        }
        seq.push_back({2, pred_first, pred_blocks, false, {(int)(3 * nl + 2)}, {}, -1, false, (int)nl});      // ONE launch, blockIdx.z = level (the entries are contiguous)
        {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, 0, h[k].L);
                h[k].SL.n = (int)nl;
                for (uint32_t l = 0; l < nl; l++) {
                    LevelParams Lp; level_params(R, l, Lp);
                    SpecLevel& sl = h[k].SL.l[l];
                    sl.w = Lp.w; sl.h = Lp.h; sl.out = Lp.out; sl.out_pitch = Lp.out_pitch; sl.out_x0 = Lp.out_x0; sl.rowmap = Lp.rowmap; sl.stamp = R.stamp[l];
                    sl.row_work = (count && R.d_row_work) ? R.d_row_work + c->row_work_off[l] : nullptr;
                }
                h[k].queue = R.pred_queue; h[k].qctl = R.pred_ctl; h[k].counters = count ? R.d_counters : nullptr;
                h[k].stamp_value = R.stamp_value; h[k].probe_empty = 1;
            }
            // the predicted launch holds a whole frame's rays: the dense build (0.61 against 0.66 ms at 1080p); the fix-up launches are
            // expected to be nearly empty: the latency build, which looks at the queue head before its first atomic.
            // Grids: with ONE frame slot the device is this frame's - full-occupancy grids; with several slots the ctx's rule for every
            // trace launch (2 persistent blocks per CU: `grid`) - a full-device persistent grid keeps the next batch's small kernels
            // (its prediction, its fix-up classification) waiting until it has drained, and two batches then run one after the other
            // (rank 3 of an 8-way 1080p partition, 20-frame blocks of a moving sequence: 0.0995 -> 0.0736 ms per frame, EXPERIMENTS R4.12)
            seq.push_back({1, d, shared_device ? grid : c->num_cus * trace_blocks_per_cu(S.method, S.models, count, 1, literal), count, {}, {(int)(3 * nl + 3)}, 1});
        }
        for (uint32_t l = 0; l < nl; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].L.pass = CLASSIFY_FIXUP; h[k].L.tag = (int)l;
                h[k].queue = R.queue[l]; h[k].qctl = R.d_qctl + 2 * l; h[k].counters = count ? R.d_counters + l : nullptr;
                h[k].need = R.need[l];
                h[k].stamp = R.stamp[l]; h[k].stamp_value = R.stamp_value; h[k].probe_empty = 1;
                h[k].row_work = (count && R.d_row_work) ? R.d_row_work + c->row_work_off[l] : nullptr;
            }
            seq.push_back({0, d, classify_blocks(l), count, {(int)(3 * l)}, {(int)(3 * l + 1)}, -1, true});
            seq.push_back({1, d, shared_device ? grid : c->num_cus * trace_blocks_per_cu(S.method, S.models, count, 0, literal), count, {}, {(int)(3 * l + 2)}, 0});
        }
        first_normal = nl;
        return BHRAY_OK;
    }

    // the plain ladder: levels [first_normal, u0), one classify + one trace launch each
    int levels(uint32_t u0) {
        for (uint32_t l = first_normal; l < u0; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].queue = R.queue[l]; h[k].qctl = R.d_qctl + 2 * l; h[k].counters = count ? R.d_counters + l : nullptr;
                h[k].row_work = (count && R.d_row_work) ? R.d_row_work + c->row_work_off[l] : nullptr;
            }
            seq.push_back({0, d, classify_blocks(l), count, {(int)(3 * l)}, {(int)(3 * l + 1)}});
            seq.push_back({1, d, grid, count, {}, {(int)(3 * l + 2)}});
            if (l + 1 < nl) seq.back().build = c->coarse_build;
        }
        return BHRAY_OK;
    }

    // superset_levels = nu: the last nu levels traced in ONE launch over a conservative superset
    int superset(uint32_t nu, uint32_t u0) {
        // Superset speculation over the last nu levels: ONE trace launch instead of nu dependent ones.
        //  (1) tentative classification of levels u0..nl-1 in order: a pixel whose coarser inputs are known is classified exactly,
        //      a pixel with an input that is itself queued (PENDING) is queued conservatively -> one level-tagged queue;
        //  (2) one trace launch over it, each ray stored at its level's own place;
        //  (3) the levels are classified again, exactly, now that the coarser level is final: copy / interpolate are stored,
        //      pixels that need tracing keep what (2) wrote (the queued set is a superset of the needed one).
        // Same pixels as the plain ladder; the extra rays are the conservatively queued pixels that turn out to interpolate.
        // Queue control words of level u0 serve the merged queue; the trace is timed as level u0's.
        for (uint32_t l = u0; l < nl; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].L.pass = CLASSIFY_TENTATIVE; h[k].L.tag = (int)(l - u0); h[k].L.no_store = (l == nl - 1) ? 1 : 0;
                h[k].queue = R.super_queue; h[k].qctl = R.d_qctl + 2 * u0; h[k].counters = nullptr;
            }
            seq.push_back({0, d, classify_blocks(l), false, {}, {}});
            if (l == u0) seq.back().ev_before = {(int)(3 * u0)};
            if (l == nl - 1) seq.back().ev_after = {(int)(3 * u0 + 1)};
        }
        {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, u0, h[k].L);
                h[k].SL.n = (int)nu;
                for (uint32_t l = u0; l < nl; l++) {
                    LevelParams Lp; level_params(R, l, Lp);
                    SpecLevel& sl = h[k].SL.l[l - u0];
                    sl.w = Lp.w; sl.h = Lp.h; sl.out = Lp.out; sl.out_pitch = Lp.out_pitch; sl.out_x0 = Lp.out_x0; sl.rowmap = Lp.rowmap;
                    sl.row_work = (count && R.d_row_work) ? R.d_row_work + c->row_work_off[l] : nullptr;
                }
                h[k].queue = R.super_queue; h[k].qctl = R.d_qctl + 2 * u0; h[k].counters = count ? R.d_counters + u0 : nullptr;
            }
            seq.push_back({1, d, grid, count, {}, {(int)(3 * u0 + 2)}});
        }
        for (uint32_t l = u0; l < nl; l++) {
            FrameLaunch* h; const FrameLaunch* d; next_launch(h, d);
            for (uint32_t k = 0; k < nb; k++) {
                const FrameRes& R = S.fr[k];
                level_params(R, l, h[k].L);
                h[k].L.pass = CLASSIFY_KEEP;
                h[k].queue = nullptr; h[k].qctl = R.d_qctl + 2 * l; h[k].counters = count ? R.d_counters + l : nullptr;
            }
            if (l == u0) seq.push_back({0, d, classify_blocks(l), count, {}, {}});      // inside level u0's trace interval
            else seq.push_back({0, d, classify_blocks(l), count, {(int)(3 * l)}, {(int)(3 * l + 1), (int)(3 * l + 2)}});
        }
        return BHRAY_OK;
    }
};
}  // namespace

// Enqueues every launch of the batch staged in the current slot: one argument block (FrameParams + per-launch FrameLaunch
// arrays) copied to the device, then the same launch sequence a single frame needs, each launch covering all staged frames.
namespace {
int launch_batch(bhray_dev* c) {
    Slot& S = c->slots[(size_t)(c->batch_counter % c->slots.size())];
    const uint32_t nb = S.pending;
    if (nb == 0) return BHRAY_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t nl = c->cfg.levels;
    // BHRAY_F_TIMING_SPARSE: events around the launches of every 4th batch only (every recorded event is a packet in the stream's
    // queue: 12 per frame cost a saturated device 1.6 %)
    const bool sparse = (c->cfg.flags & BHRAY_F_TIMING_SPARSE) != 0;
    const bool count = (c->cfg.flags & BHRAY_F_COUNTERS) != 0;
    const bool timing = (c->cfg.flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE)) != 0 && (!sparse || (c->batch_counter & 3u) == 0);
    if ((c->cfg.flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE)) != 0 && !timing) {
        const size_t ring0 = (size_t)(c->batch_counter % BHRAY_TIMING_RING);
        c->ring_frames[ring0] = 0; c->sky_recorded[ring0] = 0;          // this batch carries no events
    }
    hipStream_t st = S.stream;
    const uint32_t B = c->batch;
    const FrameParams* dP = (const FrameParams*)S.d_args;
    // Persistent trace grid: (resident blocks per CU) x CUs.  With several batches in flight each launch takes only
    // half of the block slots: the kernels of the other batches fill the rest, and a wave of a half-size grid pulls
    // more than one load of rays, so the refill keeps its lanes busy (+4 % at 16 slots).
    // Register budget of the no-mesh trace kernel (bhray_kernels.hip): the dense build when the device is saturated with
    // rays - at least ~4 whole frames' worth in flight (slots x frames per batch / row partitions) - otherwise the latency
    // build (measured on MI355X: 1920x1080, 16 slots: 4830 vs 4160 Mrays/s; 1/8 row tile, 16 slots x 8 frames: 0.070 vs
    // 0.080 ms per frame; one slot: the latency build is 7-15 % faster per launch).
    // ... with four or more row partitions a rank's launches are small and a timed block may hold only a few batches: there the count is
    // the batches IN FLIGHT when this one is launched (completed ones are retired oldest-first, one or two event queries per launch),
    // dense from 8 partitions' worth on (emulated ranks, 20-frame blocks: N = 8 0.1018 -> 0.0992 ms per frame, N = 4 0.1469 -> 0.1403;
    // 400-frame blocks unchanged; a whole frame per GPU loses 1-2 % with it: profiles/EXPERIMENTS.md R3.11).
    // ... and what counts is rays, not frames: a partition of a 3840x2160 frame holds four times the rays of the same partition of a
    // 1920x1080 one, so the frames in flight are weighted by the frame's pixels against 1920x1080 (the size the thresholds were measured
    // at; 3840x2160 over 8 partitions, 20-frame blocks: the dense build 0.2499 / 0.2407 ms per frame at 4 / 7 frames per batch against
    // 0.2614 / 0.2550 for the latency build - profiles/r04_emu_knobs.txt).
    const int dyn = c->dynamic_dense >= 0 ? c->dynamic_dense : (c->cfg.row_world >= 4 ? 8 : 0);
    size_t in_flight = c->slots.size();
    if (dyn > 0) {
        while (c->retired < c->batch_counter) {
            const Slot& O = c->slots[(size_t)(c->retired % c->slots.size())];
            if (O.batch_id == c->retired && hipEventQuery(O.done) != hipSuccess) break;
            c->retired++;
        }
        in_flight = (size_t)(c->batch_counter - c->retired) + 1;
    }
    // A launch that by itself holds 2.5 frames' worth of rays is dense whatever else is in flight (the first batches of a short block).
    const double weight = std::max(1.0, (double)c->cfg.frame_w * (double)c->cfg.frame_h / (1920.0 * 1080.0)) / (double)c->cfg.row_world;   // 1920x1080 frames' worth per frame of this partition
    const bool dense = c->dense_override >= 0 ? c->dense_override != 0
                                              : ((double)nb * weight >= 2.5 || (double)(in_flight * (size_t)c->batch) * weight >= (double)(dyn > 0 ? dyn : 4));
    const int literal = (c->cfg.flags & BHRAY_F_LITERAL) ? 1 : ((c->cfg.flags & BHRAY_F_EVAL_FMA) ? 2 : 0);   // the integrator's evaluation (launch_trace's `eval`)
    int bpc = trace_blocks_per_cu(S.method, S.models, count, dense, literal);
    if (c->slots.size() > 1 && bpc > 1) bpc = bpc > 4 ? 2 : (bpc / 2 > 1 ? bpc / 2 : 1);     // measured: 2 blocks per CU is best at 8-16 slots
    if (c->bpc_override > 0) bpc = c->bpc_override;
    int grid = c->num_cus * bpc;
    // ... and one and a half for the Euler kernels (their steps are short: a block's share of a queue is used up sooner, and a smaller grid leaves the later launches'
    // blocks room beside it): 512 -> 384 blocks at 22 slots +1.5 % (20- and 400-frame blocks), with the mesh +2.4-3 %, a rank of 8 0 / +2 %; the RK kernels: nothing
    // (profiles/r05_ab_euler_grid.txt, EXPERIMENTS.md R5.9)
    if (S.method == 0 && c->slots.size() >= 8 && bpc == 2 && c->bpc_override <= 0) grid = c->num_cus * 3 / 2;      // (measured at 22 slots only: from 8 slots on)
    if (c->grid_override > 0) grid = c->grid_override;
    BatchPlan plan{c, S, nb, nl, count, literal, grid, st, (size_t)B * sizeof(FrameParams)};
    const uint32_t ns = c->cfg.speculative_levels;
    const bool any_rows = !c->levels[nl - 1].rows.empty();    // a partition without rows has nothing to launch (then no level has rows)
    const bool temporal = (c->cfg.flags & BHRAY_F_TEMPORAL) != 0;
    const uint32_t nu = temporal ? 0 : c->cfg.superset_levels;
    const uint32_t u0 = nu ? nl - nu : nl;                     // first level of the superset group
    if (any_rows) {
        int rc = BHRAY_OK;
        {
            if (ns) rc = plan.speculative(ns);
            if (!rc && temporal) rc = plan.temporal();
            if (!rc) rc = plan.levels(u0);
            if (!rc && nu) rc = plan.superset(nu, u0);
        }
        if (rc) return rc;
    }
    std::vector<Launch>& seq = plan.seq;
    const size_t args_used = plan.args_used;
    if (args_used > S.args_cap) return fail(c, BHRAY_E_STATE, "internal: argument block overflow");
    const size_t ring = (size_t)(c->batch_counter % BHRAY_TIMING_RING);
    if (timing && c->d_span) {                 // execution spans of this batch's trace launches (entry 0 of each launch's FrameLaunch array)
        int nt = 0;
        for (const Launch& Ln : seq) {
            if (Ln.kind != 1 || nt >= SPAN_MAX) continue;
            FrameLaunch* h0 = (FrameLaunch*)(S.h_args + ((const uint8_t*)Ln.d - S.d_args));
            h0->span = c->d_span + (ring * SPAN_MAX + (size_t)nt) * 2;
            nt++;
        }
        c->ring_spans[ring] = (uint8_t)nt;
        HIPCHK(c, hipMemsetAsync(c->d_span + ring * SPAN_MAX * 2, 0, (size_t)SPAN_MAX * 2 * sizeof(unsigned long long), st));
    } else {
        c->ring_spans[ring] = 0;
    }
    for (const Launch& Ln : seq) {             // every trace launch adds the steps its waves issue for a frame to that frame's work counters
        if (Ln.kind != 1) continue;
        FrameLaunch* hl = (FrameLaunch*)(S.h_args + ((const uint8_t*)Ln.d - S.d_args));
        for (uint32_t k = 0; k < nb; k++) hl[k].work = S.fr[k].d_work;
        // a whole frame, one frame per launch: thin shares are dealt strided (bhray_kernels.hip, thin_stride; bit 1 of probe_empty)
        if (c->cfg.row_world <= 1 && nb == 1) hl[0].probe_empty |= 2;
        // the quad march (bhray_quad.inc) for launches whose queue turns out short: how many waves per SIMD it may use (bits 2-4; 0 = off)
        // Measured (profiles/EXPERIMENTS.md R6.1): one wave per SIMD -21 % per launch (RK, a level-0-sized queue), two -10 %, three +10 %: a second wave on a
        // SIMD slows both, and from the third on a scalar wave with four times the rays is faster.  The Euler step has too little 3-vector work to gain at two.
        // Only for a host that renders one frame at a time (one frame per launch, at most two frame slots): beside other frames' waves on the same SIMDs
        // the quad march's shorter iteration is gone and its four lanes per ray cost throughput - a rank of an 8-way partition, batches of 5 frames: +3.5 %
        // per frame in the driver's blocks against -11-14 % one frame at a time (R6.1).  BHRAY_QUAD=n forces it for every latency-build launch.
        // ... and wave priority by predicted ray length (bit 5; BHRAY_WAVE_PRIO in bhray_kernels.hip) for a host with ONE frame slot: it shortens a lone frame's launches (S = 2 1.16 -> 1.13 ms,
        // S = 3 0.97 -> 0.95), but as soon as two frames overlap it costs - the drop-in shim with two frames in flight 0.816 -> 0.873 ms per frame, a saturated device 3.5 % (EXPERIMENTS.md R6.5)
        if (c->wave_prio >= 0 ? c->wave_prio != 0 : (nb == 1 && c->slots.size() <= 1)) for (uint32_t k = 0; k < nb; k++) hl[k].probe_empty |= 32;
        const int quad_wps = c->quad_wps >= 0 ? c->quad_wps : ((nb == 1 && c->slots.size() <= 2) ? (S.method == 0 ? 1 : 2) : 0);
        for (uint32_t k = 0; k < nb; k++) hl[k].probe_empty |= (quad_wps & 7) << 2;
    }
    S.launched_frames = nb;
    // enqueue
    HIPCHK(c, launch_upload(S.h_args, S.d_args, (args_used + 15) / 16, S.d_qctl, (size_t)nb * BHRAY_QCTL_WORDS, st));   // + queue control reset
    HIPCHK(c, hipEventRecord(S.uploaded, st));
    if (count) HIPCHK(c, hipMemsetAsync(S.d_counters, 0, (size_t)nb * BHRAY_MAX_LEVELS * sizeof(Counters64), st));
    if (count) for (uint32_t k = 0; k < nb; k++) if (S.fr[k].d_row_work) HIPCHK(c, hipMemsetAsync(S.fr[k].d_row_work, 0, c->row_work_off[nl] * sizeof(unsigned long long), st));
    hipEvent_t* fev = timing ? &c->events[ring * (nl * 3 + 4)] : nullptr;
    if (timing) { c->sky_recorded[ring] = 0; c->ring_frames[ring] = (uint8_t)nb; }
    for (const Launch& Ln : seq) {
        if (timing) for (int e : Ln.ev_before) HIPCHK(c, hipEventRecord(fev[e], st));
        if (Ln.kind == 2) HIPCHK(c, launch_predict(dP, Ln.d, (int)nb, Ln.levels, Ln.blocks, st));
        else if (Ln.kind == 0) HIPCHK(c, launch_classify(dP, Ln.d, (int)nb, Ln.blocks, Ln.count, Ln.fixup, st));
        else HIPCHK(c, launch_trace(dP, Ln.d, (int)nb, S.method, S.models, Ln.count, Ln.build < 0 ? dense : Ln.build != 0, literal, c->d_err, Ln.blocks, st));
        if (timing) for (int e : Ln.ev_after) HIPCHK(c, hipEventRecord(fev[e], st));
    }
    HIPCHK(c, hipEventRecord(S.done, st));
    S.used = true;
    S.batch_id = c->batch_counter;
    S.pending = 0;
    c->launched_slot = (int)(c->batch_counter % c->slots.size());
    c->launched_frames = nb;
    c->batch_counter++;
    return BHRAY_OK;
}

// kernel variant the current uniforms and scene need (same rule as derive_frame: a mesh only when one is usable and visible)
void frame_variant(const bhray_dev* c, int& method, bool& models) {
    method = c->det.integration_method != 0 ? 1 : 0;
    int mc = c->det.model_count; if (mc < 0) mc = 0; if (mc > BHRAY_MAX_MODELS) mc = BHRAY_MAX_MODELS;
    models = false;
    for (int i = 0; i < mc; i++) {
        const ModelStore& m = c->models[i];
        if (m.loaded && m.triangle_count > 0 && m.visible != 0) models = true;
    }
}
}  // namespace

int dev_flush(bhray_dev* c) {
    if (!c) return BHRAY_E_INVALID;
    return launch_batch(c);
}

int dev_render(bhray_dev* c) {
    if (!c) return BHRAY_E_INVALID;
    if (!c->have_uniforms) return fail(c, BHRAY_E_STATE, "dev_set_uniforms has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    FrameParams P;
    derive_frame(c, P);
    {   // a batch runs one kernel variant: launch what is staged if this frame needs another
        Slot& S0 = c->slots[(size_t)(c->batch_counter % c->slots.size())];
        if (S0.pending > 0 && (S0.method != P.method || S0.models != (P.model_count > 0))) { int rc = launch_batch(c); if (rc) return rc; }
    }
    const int si = (int)(c->batch_counter % c->slots.size());
    Slot& S = c->slots[(size_t)si];
    if (S.pending == 0) {
        // the slot's previous argument block must have reached the device before the pinned copy is rewritten
        if (S.used && hipEventQuery(S.uploaded) != hipSuccess) HIPCHK(c, hipEventSynchronize(S.uploaded));
        S.method = P.method; S.models = P.model_count > 0;
    }
    for (hipEvent_t ev : c->waits) HIPCHK(c, hipStreamWaitEvent(S.stream, ev, 0));
    c->waits.clear();
    const uint32_t k = S.pending;
    FrameRes& R = S.fr[k];
    ((FrameParams*)S.h_args)[k] = P;
    R.out = c->bound_out ? c->bound_out : R.own_out;
    c->bound_out = nullptr;                                   // a binding applies to one frame
    if (!R.out && c->out_bytes) return fail(c, BHRAY_E_STATE, "internal: no output bound for this frame");
    R.frame_id = c->frame_counter;
    {   // the level row the hole projects to (create_ray, ray.wgsl:269-285, inverted): its rows are classified first
        const F3 d = ld3(P.bh) - ld3(P.cam), ff = ld3(P.fwd_ff);
        const float along = dot(d, ff) / dot(ff, ff);
        float frac = 0.5f;
        if (along > 0.0f) {
            const Level& Ll = c->levels[c->cfg.levels - 1];
            const float sm = (float)std::min(Ll.w - 1, Ll.h - 1);
            const float posy = dot(d, ld3(P.up)) / along;
            frac = (posy * sm * 0.5f + (float)(Ll.h - 1) * 0.5f) / (float)std::max(1, Ll.h - 1);
        }
        if (!(frac >= 0.0f)) frac = 0.0f;
        if (frac > 1.0f) frac = 1.0f;
        // ... when few frames are in flight (a host that presents every frame: a launch then lasts as long as its last rays - one frame at a time
        // 1.136 -> 1.087 ms, the drop-in shim with 2 frames in flight 2 515 -> 2 618 Mrays/s); with many frames in flight the launches overlap each
        // other's tails and the ascending order keeps its locality (4K / Euler -1 % with the ordering: EXPERIMENTS R4.16)
        R.row_variant = (BHRAY_ROW_VARIANTS >= 2 && c->slots.size() <= 4) ? (int)lrintf(frac * (float)(BHRAY_ROW_VARIANTS - 1)) : -1;
    }
    S.pending = k + 1;
    c->last_slot = si; c->last_sub = (int)k;
    c->rendered = true;
    c->frame_counter++;
    if (S.pending == c->batch) return launch_batch(c);
    return BHRAY_OK;
}

int dev_sync(bhray_dev* c) {
    if (!c) return BHRAY_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }
    HIPCHK(c, sync_all(c));
    if (c->rendered) {
        int e = 0;
        HIPCHK(c, hipMemcpy(&e, c->d_err, sizeof e, hipMemcpyDeviceToHost));
        if (e != 0) {
            HIPCHK(c, hipMemset(c->d_err, 0, sizeof(int)));
            return fail(c, e, "kernel reported: %s", bhray_strerror(e));
        }
    }
    return BHRAY_OK;
}

uint32_t dev_local_rows(const bhray_dev* c) { return c ? (uint32_t)c->local_rows.size() : 0; }

int dev_local_row_index(const bhray_dev* c, uint32_t i, uint32_t* frame_row) {
    if (!c || !frame_row || i >= c->local_rows.size()) return BHRAY_E_INVALID;
    *frame_row = c->local_rows[i];
    return BHRAY_OK;
}

int dev_read_hdr(bhray_dev* c, float* dst, size_t pitch) {
    if (!c) return BHRAY_E_INVALID;
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(float4);
    if (c->local_rows.empty()) return dev_sync(c);              // this partition owns no rows: nothing to copy
    if (!dst || pitch < rowb) return fail(c, BHRAY_E_INVALID, "bad destination / pitch");
    int rc = dev_sync(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy2D(dst, pitch, c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub].out, rowb, rowb, c->local_rows.size(), hipMemcpyDeviceToHost));
    return BHRAY_OK;
}

// Asynchronous hand-off of the most recently enqueued frame to host memory (a consumer on another device: the wgpu texture the
// reference's SkyPipeline samples, ray_pipeline.rs:297-299, mod.rs:215).  The copy (SDMA: no CU time, 55.7 GB/s into pinned memory
// on this box) is enqueued ON THE FRAME'S OWN SLOT STREAM, behind its kernels; the call returns at once and the other slots' frames
// render while it runs.  (First version: dedicated copy streams ordered by events both ways - ROCm 7.2 then ran copies and kernels
// strictly one after the other, 1.22 ms per 1080p frame; in stream order 0.62 ms, which is the link rate: profiles/EXPERIMENTS.md.)
// `dst` should be pinned (bhray_host_alloc / hipHostRegister): a pageable destination makes the runtime stage the copy.
int dev_read_hdr_async(bhray_dev* c, float* dst, size_t pitch, uint64_t* ticket) {
    if (!c || !ticket) return BHRAY_E_INVALID;
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(float4);
    if (!c->rendered) return fail(c, BHRAY_E_STATE, "nothing rendered yet");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }
    const uint64_t t = c->read_tickets;
    hipEvent_t& ev = c->read_ev[t % BHRAY_READ_RING];
    if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (t >= BHRAY_READ_RING) HIPCHK(c, hipEventSynchronize(ev));          // the ticket that held this event 64 copies ago
    Slot& S = c->slots[(size_t)c->last_slot];
    const size_t rows = c->local_rows.size();
    if (rows) {
        if (!dst || pitch < rowb) return fail(c, BHRAY_E_INVALID, "bad destination / pitch");
        const uint8_t* src = (const uint8_t*)S.fr[(size_t)c->last_sub].out;
        if (pitch == rowb) HIPCHK(c, hipMemcpyAsync(dst, src, rows * rowb, hipMemcpyDeviceToHost, S.stream));
        else HIPCHK(c, hipMemcpy2DAsync(dst, pitch, src, rowb, rowb, rows, hipMemcpyDeviceToHost, S.stream));
    }
    HIPCHK(c, hipEventRecord(ev, S.stream));
    HIPCHK(c, hipEventRecord(S.done, S.stream));           // whatever is ordered behind this slot's frame is ordered behind its copy too
    *ticket = t;
    c->read_tickets = t + 1;
    return BHRAY_OK;
}

// The same hand-off for the RGBA16F image of the sky pass (half the bytes): bhray_resolve_sky must have been called for the frame.
int dev_read_sky_async(bhray_dev* c, uint16_t* dst, size_t pitch, uint64_t* ticket) {
    if (!c || !ticket) return BHRAY_E_INVALID;
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(uint2);
    if (!c->rendered) return fail(c, BHRAY_E_STATE, "nothing rendered yet");
    HIPCHK(c, hipSetDevice(c->device));
    Slot& S = c->slots[(size_t)c->last_slot];
    FrameRes& R = S.fr[(size_t)c->last_sub];
    const size_t rows = c->local_rows.size();
    // refused BEFORE anything is launched: a misuse (render without resolve_sky) must not flush a partly filled batch on its way to the error
    if (rows && (!R.sky_out || R.sky_frame_id != R.frame_id)) return fail(c, BHRAY_E_STATE, "dev_resolve_sky has not been called for this frame");
    { int rc = launch_batch(c); if (rc) return rc; }          // (dev_resolve_sky launched the frame's batch: nothing is pending unless a partition without rows skipped it)
    const uint64_t t = c->read_tickets;
    hipEvent_t& ev = c->read_ev[t % BHRAY_READ_RING];
    if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (t >= BHRAY_READ_RING) HIPCHK(c, hipEventSynchronize(ev));
    if (rows) {
        if (!dst || pitch < rowb) return fail(c, BHRAY_E_INVALID, "bad destination / pitch");
        if (pitch == rowb) HIPCHK(c, hipMemcpyAsync(dst, R.sky_out, rows * rowb, hipMemcpyDeviceToHost, S.stream));
        else HIPCHK(c, hipMemcpy2DAsync(dst, pitch, R.sky_out, rowb, rowb, rows, hipMemcpyDeviceToHost, S.stream));
    }
    HIPCHK(c, hipEventRecord(ev, S.stream));
    HIPCHK(c, hipEventRecord(S.done, S.stream));
    *ticket = t;
    c->read_tickets = t + 1;
    return BHRAY_OK;
}

int dev_wait_read(bhray_dev* c, uint64_t ticket) {
    if (!c) return BHRAY_E_INVALID;
    if (ticket >= c->read_tickets) return fail(c, BHRAY_E_INVALID, "unknown read ticket");
    if (c->read_tickets - ticket > BHRAY_READ_RING) return BHRAY_OK;              // its event was waited for when it was recycled
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->read_ev[ticket % BHRAY_READ_RING]));
    return BHRAY_OK;
}

int dev_read_level(bhray_dev* c, uint32_t level, float* dst, size_t pitch) {
    if (!c) return BHRAY_E_INVALID;
    if (level >= c->cfg.levels) return fail(c, BHRAY_E_INVALID, "level out of range");
    const Level& L = c->levels[level];
    const size_t rowb = (size_t)L.w * sizeof(float4);
    if (!dst || pitch < rowb) return fail(c, BHRAY_E_INVALID, "bad destination / pitch");
    int rc = dev_sync(c);
    if (rc) return rc;
    if (level + 1 < c->cfg.levels) {
        HIPCHK(c, hipMemcpy2D(dst, pitch, c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub].level_out[level], rowb, rowb, (size_t)L.h, hipMemcpyDeviceToHost));
        return BHRAY_OK;
    }
    // last level: scatter the packed window back into a NaN canvas of the full level size
    for (int y = 0; y < L.h; y++) memset((uint8_t*)dst + (size_t)y * pitch, 0xFF, rowb);
    const size_t frb = (size_t)c->cfg.frame_w * sizeof(float4);
    std::vector<uint8_t> tmp(c->local_rows.size() * frb);
    if (!tmp.empty()) HIPCHK(c, hipMemcpy(tmp.data(), c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub].out, tmp.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < c->local_rows.size(); i++) {
        uint8_t* row = (uint8_t*)dst + (size_t)(c->cfg.crop_y + c->local_rows[i]) * pitch + (size_t)c->cfg.crop_x * sizeof(float4);
        memcpy(row, tmp.data() + i * frb, frb);
    }
    return BHRAY_OK;
}

int dev_hdr_device_ptr(bhray_dev* c, void** p, size_t* bytes) {
    if (!c || !p) return BHRAY_E_INVALID;
    *p = c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub].out; if (bytes) *bytes = c->out_bytes;
    return BHRAY_OK;
}

int dev_bind_output(bhray_dev* c, void* p, size_t bytes) {
    if (!c) return BHRAY_E_INVALID;
    if (!p) { c->bound_out = nullptr; return BHRAY_OK; }
    if (bytes < c->out_bytes) return fail(c, BHRAY_E_INVALID, "output binding needs %zu bytes", c->out_bytes);
    if (((uintptr_t)p & 15u) != 0) return fail(c, BHRAY_E_INVALID, "output binding must be 16-byte aligned");
    c->bound_out = (float4*)p;
    return BHRAY_OK;
}

int dev_resolve_sky(bhray_dev* c) {
    if (!c) return BHRAY_E_INVALID;
    if (!c->rendered) return fail(c, BHRAY_E_STATE, "nothing rendered yet");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }
    Slot& S = c->slots[(size_t)c->last_slot];
    FrameRes& R = S.fr[(size_t)c->last_sub];
    const size_t npix = c->local_rows.size() * (size_t)c->cfg.frame_w;
    if (!R.sky_out && npix) HIPCHK(c, hipMalloc(&R.sky_out, npix * sizeof(uint2)));
    TexDev sky; sky.rgba = c->tex[BHRAY_TEX_SKY]; sky.w = c->tex_w[BHRAY_TEX_SKY]; sky.h = c->tex_h[BHRAY_TEX_SKY];
    const bool timing = (c->cfg.flags & BHRAY_F_TIMING) != 0;
    const size_t ring = (size_t)(S.batch_id % BHRAY_TIMING_RING);
    hipEvent_t* ev = timing ? &c->events[ring * (c->cfg.levels * 3 + 4) + c->cfg.levels * 3] : nullptr;
    if (timing) HIPCHK(c, hipEventRecord(ev[0], S.stream));
    HIPCHK(c, launch_sky(sky, R.out, R.sky_out, npix, S.stream));
    R.sky_frame_id = R.frame_id;
    if (timing) { HIPCHK(c, hipEventRecord(ev[1], S.stream)); c->sky_recorded[ring] = 1; }
    HIPCHK(c, hipEventRecord(S.done, S.stream));
    return BHRAY_OK;
}

int dev_read_sky(bhray_dev* c, uint16_t* dst, size_t pitch) {
    if (!c) return BHRAY_E_INVALID;
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(uint2);
    if (c->local_rows.empty()) return dev_sync(c);
    if (!dst || pitch < rowb) return fail(c, BHRAY_E_INVALID, "bad destination / pitch");
    FrameRes& S = c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub];
    if (!S.sky_out || S.sky_frame_id != S.frame_id) return fail(c, BHRAY_E_STATE, "dev_resolve_sky has not been called for this frame");
    int rc = dev_sync(c);
    if (rc) return rc;
    if (c->local_rows.empty()) return BHRAY_OK;
    HIPCHK(c, hipMemcpy2D(dst, pitch, S.sky_out, rowb, rowb, c->local_rows.size(), hipMemcpyDeviceToHost));
    return BHRAY_OK;
}

int dev_sky_device_ptr(bhray_dev* c, void** p, size_t* bytes) {
    if (!c || !p) return BHRAY_E_INVALID;
    *p = nullptr;
    const FrameRes& R = c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub];
    // the same rule as the reads: the image belongs to the frame it was resolved from (a slot position is reused: an older frame's image is stale)
    if (!c->local_rows.empty() && (!R.sky_out || R.sky_frame_id != R.frame_id)) return fail(c, BHRAY_E_STATE, "dev_resolve_sky has not been called for this frame");
    *p = R.sky_out;
    if (bytes) *bytes = c->local_rows.size() * (size_t)c->cfg.frame_w * sizeof(uint2);
    return BHRAY_OK;
}

int dev_wait_event(bhray_dev* c, hipEvent_t ev) {
    if (!c || !ev) return BHRAY_E_INVALID;
    c->waits.push_back(ev);
    return BHRAY_OK;
}

int dev_device(const bhray_dev* c) { return c ? c->device : -1; }

int dev_next_position(bhray_dev* c, int* slot, uint32_t* sub) {
    if (!c || !slot || !sub) return BHRAY_E_INVALID;
    Slot& S0 = c->slots[(size_t)(c->batch_counter % c->slots.size())];
    if (S0.pending > 0) {
        int method; bool models;
        frame_variant(c, method, models);
        if (S0.method != method || S0.models != models) { int rc = launch_batch(c); if (rc) return rc; }
    }
    *slot = (int)(c->batch_counter % c->slots.size());
    *sub = c->slots[(size_t)*slot].pending;
    return BHRAY_OK;
}

void dev_peek_position(const bhray_dev* c, uint64_t* batch_counter, uint32_t* pending, int* method, bool* models) {
    const Slot& S = c->slots[(size_t)(c->batch_counter % c->slots.size())];
    *batch_counter = c->batch_counter; *pending = S.pending; *method = S.method; *models = S.models;
}

bool dev_take_launched(bhray_dev* c, int* slot, uint32_t* frames) {
    if (!c || c->launched_slot < 0) return false;
    if (slot) *slot = c->launched_slot;
    if (frames) *frames = c->launched_frames;
    c->launched_slot = -1; c->launched_frames = 0;
    return true;
}

int* dev_err_flag(bhray_dev* c) { return c->d_err; }
hipStream_t dev_slot_stream(bhray_dev* c, int slot) { return c->slots[(size_t)slot].stream; }
hipEvent_t dev_slot_done(bhray_dev* c, int slot) { return c->slots[(size_t)slot].done; }

int dev_launch_sky(bhray_dev* c, const void* src, void* dst, size_t npix, hipStream_t stream) {
    if (!c) return BHRAY_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    TexDev sky; sky.rgba = c->tex[BHRAY_TEX_SKY]; sky.w = c->tex_w[BHRAY_TEX_SKY]; sky.h = c->tex_h[BHRAY_TEX_SKY];
    HIPCHK(c, launch_sky(sky, (const float4*)src, (uint2*)dst, npix, stream));
    return BHRAY_OK;
}

int dev_next_stream(bhray_dev* c, void** s) {
    if (!c || !s) return BHRAY_E_INVALID;
    *s = (void*)c->slots[(size_t)(c->batch_counter % c->slots.size())].stream;
    return BHRAY_OK;
}

int dev_signal_stream(bhray_dev* c, void* s) {
    if (!c) return BHRAY_E_INVALID;
    if (!c->rendered) return BHRAY_OK;
    HIPCHK(c, hipSetDevice(c->device));
    { int rc = launch_batch(c); if (rc) return rc; }
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)s, c->slots[(size_t)c->last_slot].done, 0));
    return BHRAY_OK;
}

int dev_selftest(bhray_dev* c, uint64_t mismatches[3]) {
    if (!c || !mismatches) return BHRAY_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIPCHK(c, hipMalloc(&d, 24));
    hipError_t e = hipMemset(d, 0, 24);
    if (e == hipSuccess) e = launch_selftest(d, nullptr);
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, BHRAY_E_HIP, "selftest: %s", hipGetErrorString(e));
    mismatches[0] = h[0]; mismatches[1] = h[1]; mismatches[2] = h[2];
    return BHRAY_OK;
}

int dev_get_level_counters(bhray_dev* c, uint32_t level, bhray_counters* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    if (!(c->cfg.flags & BHRAY_F_COUNTERS)) return fail(c, BHRAY_E_STATE, "ctx created without BHRAY_F_COUNTERS");
    if (level >= c->cfg.levels) return fail(c, BHRAY_E_INVALID, "level out of range");
    int rc = dev_sync(c);
    if (rc) return rc;
    static_assert(sizeof(bhray_counters) == sizeof(Counters64), "counter layout");
    HIPCHK(c, hipMemcpy(out, c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub].d_counters + level, sizeof(Counters64), hipMemcpyDeviceToHost));
    return BHRAY_OK;
}

int dev_add_row_work(bhray_dev* c, uint32_t level, uint64_t* acc, uint32_t n) {
    if (!c || !acc) return BHRAY_E_INVALID;
    if (!(c->cfg.flags & BHRAY_F_COUNTERS)) return fail(c, BHRAY_E_STATE, "ctx created without BHRAY_F_COUNTERS");
    if (level >= c->cfg.levels || n != c->cfg.level_h[level]) return fail(c, BHRAY_E_INVALID, "level out of range, or n is not the level's height");
    int rc = dev_sync(c);
    if (rc) return rc;
    const FrameRes& R = c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub];
    std::vector<unsigned long long> tmp(n);
    HIPCHK(c, hipMemcpy(tmp.data(), R.d_row_work + c->row_work_off[level], (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < n; y++) acc[y] += tmp[y];
    return BHRAY_OK;
}

// What the frames of the batches still held by the slots cost this engine: integrator steps issued by its trace waves (FrameLaunch::work),
// mean per frame, and the pixels its classify launches visit per frame (an HBM-bound pass: its cost goes with the pixels).
int dev_get_work(bhray_dev* c, double* wave_steps_per_frame, double* classify_pixels_per_frame, uint32_t* frames, int* method) {
    if (!c || !wave_steps_per_frame || !classify_pixels_per_frame || !frames || !method) return BHRAY_E_INVALID;
    int rc = dev_sync(c);
    if (rc) return rc;
    std::vector<uint32_t> tmp((size_t)c->batch * BHRAY_QCTL_WORDS);
    double sum = 0.0; uint32_t n = 0;
    for (const Slot& S : c->slots) {
        if (!S.used || S.launched_frames == 0) continue;
        HIPCHK(c, hipMemcpy(tmp.data(), S.d_qctl, (size_t)S.launched_frames * BHRAY_QCTL_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint32_t k = 0; k < S.launched_frames; k++) {
            unsigned long long w[BHRAY_WORK_WORDS];
            memcpy(w, tmp.data() + (size_t)k * BHRAY_QCTL_WORDS + 2 * BHRAY_MAX_LEVELS, sizeof w);
            for (unsigned long long v : w) sum += (double)v;
            n++;
        }
    }
    double px = 0.0;
    for (const Level& L : c->levels) px += (double)L.queue_cap;
    *wave_steps_per_frame = n ? sum / (double)n : 0.0;
    *classify_pixels_per_frame = px;
    *frames = n;
    *method = c->det.integration_method != 0 ? 1 : 0;
    return BHRAY_OK;
}

// diagnostics (not part of include/bhray.h): the entries the last frame's own queue of `level` held (temporal mode: the fix-up set)
int dev_debug_read_queue(bhray_dev* c, uint32_t level, uint32_t* out, uint32_t cap, uint32_t* count) {
    if (!c || !count || level >= c->cfg.levels) return BHRAY_E_INVALID;
    int rc = dev_sync(c);
    if (rc) return rc;
    const FrameRes& R = c->slots[(size_t)c->last_slot].fr[(size_t)c->last_sub];
    uint32_t n = 0;
    HIPCHK(c, hipMemcpy(&n, R.d_qctl + 2 * level, sizeof n, hipMemcpyDeviceToHost));
    *count = n;
    if (out && R.queue[level]) HIPCHK(c, hipMemcpy(out, R.queue[level], (size_t)(n < cap ? n : cap) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BHRAY_OK;
}

int dev_get_counters(bhray_dev* c, bhray_counters* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    memset(out, 0, sizeof *out);
    for (uint32_t l = 0; l < c->cfg.levels; l++) {
        bhray_counters t;
        int rc = dev_get_level_counters(c, l, &t);
        if (rc) return rc;
        const uint64_t* a = (const uint64_t*)&t; uint64_t* b = (uint64_t*)out;
        for (size_t k = 0; k < sizeof(bhray_counters) / 8; k++) { if (k == 12) b[k] = a[k] > b[k] ? a[k] : b[k]; else b[k] += a[k]; }   // [12] max_ray_iterations
    }
    return BHRAY_OK;
}

int dev_get_timing(bhray_dev* c, bhray_timing* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    if (!(c->cfg.flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE))) return fail(c, BHRAY_E_STATE, "ctx created without BHRAY_F_TIMING");
    int rc = dev_sync(c);
    if (rc) return rc;
    memset(out, 0, sizeof *out);
    const uint32_t nl = c->cfg.levels;
    uint64_t begin = c->timing_begin;
    if (c->batch_counter - begin > BHRAY_TIMING_RING) begin = c->batch_counter - BHRAY_TIMING_RING;
    for (uint64_t f = begin; f < c->batch_counter; f++) {
        const size_t ring = (size_t)(f % BHRAY_TIMING_RING);
        hipEvent_t* ev = &c->events[ring * (nl * 3 + 4)];
        if (!c->ring_frames[ring]) continue;                           // a batch without events (BHRAY_F_TIMING_SPARSE)
        if (c->sky_recorded[ring]) {
            float t = 0; HIPCHK(c, hipEventElapsedTime(&t, ev[3 * nl], ev[3 * nl + 1])); out->sky_ms += t; out->sky_launches++;
        }
        hipEvent_t first = nullptr, last = nullptr;
        for (uint32_t l = 0; l < nl; l++) {
            if (c->levels[l].rows.empty()) continue;
            float a = 0, b = 0;
            HIPCHK(c, hipEventElapsedTime(&a, ev[3 * l], ev[3 * l + 1]));
            HIPCHK(c, hipEventElapsedTime(&b, ev[3 * l + 1], ev[3 * l + 2]));
            const bool spec_classified = (c->cfg.speculative_levels && l >= 1 && l < c->cfg.speculative_levels) ||
                                         (c->cfg.superset_levels && l > nl - c->cfg.superset_levels);               // no trace launch of its own
            out->classify_ms += a; out->level_classify_ms[l] += a; out->classify_launches++;
            if (!spec_classified) { out->trace_ms += b; out->level_trace_ms[l] += b; out->trace_launches++; }
            if (!first) first = ev[3 * l];
            last = ev[3 * l + 2];
        }
        if ((c->cfg.flags & BHRAY_F_TEMPORAL) && first) {      // prediction + predicted trace launch: the bulk of a temporal-mode frame
            float t = 0; HIPCHK(c, hipEventElapsedTime(&t, ev[3 * nl + 2], ev[3 * nl + 3]));
            out->predicted_trace_ms += t; out->predicted_launches++;
            first = ev[3 * nl + 2];
        }
        if (first && last) { float t = 0; HIPCHK(c, hipEventElapsedTime(&t, first, last)); out->total_ms += t; }
        out->frames += c->ring_frames[ring];
        out->batches++;
    }
    if (c->d_span) {
        std::vector<unsigned long long> sp((size_t)BHRAY_TIMING_RING * SPAN_MAX * 2);
        HIPCHK(c, hipMemcpy(sp.data(), c->d_span, sp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (uint64_t f = begin; f < c->batch_counter; f++) {
            const size_t ring = (size_t)(f % BHRAY_TIMING_RING);
            if (!c->ring_frames[ring]) continue;
            for (int t = 0; t < (int)c->ring_spans[ring]; t++) {
                const unsigned long long a = ~sp[(ring * SPAN_MAX + (size_t)t) * 2], b = sp[(ring * SPAN_MAX + (size_t)t) * 2 + 1];
                if (sp[(ring * SPAN_MAX + (size_t)t) * 2] == 0ull || b < a) continue;       // the launch never ran (no rows)
                out->trace_exec_ms += (float)((double)(b - a) / c->wall_clock_khz);
                out->trace_exec_launches++;
            }
        }
    }
    c->timing_begin = c->batch_counter;
    return BHRAY_OK;
}
