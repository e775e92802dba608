// bhray_internal.h — structures passed from the C-ABI host layer to the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bhray.h"

namespace bhray {

struct TexDev {
    const uint8_t* rgba;   // RGBA8 unorm, row-major, tightly packed (texture.rs:16-69)
    int w, h;
};

// Compact model resident in HBM.  Nodes are 32 B (two float4 halves); the two children of an
// inner node are adjacent (triangle.rs:239-243), so a child pair is one 64 B aligned read.
struct ModelDev {
    float pos[3];
    int visible;
    const float4* points;
    const float4* normals;
    const int32_t* triangles;   // 6 ints each
    const float4* nodes;        // 2 float4 per node
    const int32_t* lookup;
    const float4* leaf;         // pre-gathered leaf geometry: 6 float4 (p1,p2,p3,n1,n2,n3) per bvh_lookup slot
    int node_count;
    int root_cull;              // 1: root_lo/root_hi hold the union of the root's two child boxes (root is an inner node)
    float root_lo[3], root_hi[3];
};

// Everything that is uniform over a frame.  Lives in HBM (one entry per frame of a batch); the address is wave-uniform,
// so the members are fetched with scalar loads into SGPRs.
// Derived members are computed on the host with the SAME binary32 operation sequence the shader
// performs per pixel (ray.wgsl:270-281, 488, 511), so hoisting them cannot change a bit.
struct FrameParams {
    // camera (ray.wgsl:41-45, 269-285)
    float cam[3];
    float right[3];        // normalize(cross(forward, (0,-1,0)))
    float up[3];           // normalize(cross(forward, right))
    float fwd_ff[3];       // forward * (1 / tan(fov/2))
    float ray_distance;    // distance(camera.position, black_hole.position)  (ray.wgsl:511,555)
    float ray_distance_f;  // the same with the integrator's fused dot (N7): `dist` of a ray's first step
    int relativity0;       // ray_distance < relativity_radius                (ray.wgsl:488)
    // black hole (ray.wgsl:112-123)
    float bh[3];
    float bn[3];
    float bn_len;          // length(normal) (host), for the conservative disk cull
    float cull_outer_pad, cull_plane_c1, cull_plane_c2;   // black_hole_culls' folded constants: outer + 0.0501, 1.0101 |n|, 1.01e-4 |n|
    float inner, outer, rot_speed, R;
    int show_tex, show_shift;
    float M[9];            // rotation matrix columns c0,c1,c2
    float feather;
    // details (ray.wgsl:25-34)
    float time;
    float time_rot;        // time * rotation_speed (ray.wgsl:633): the same single binary32 product, formed once on the host
    int method;
    float step_size;
    int max_iter;
    float thr;
    float acos_cstar;      // largest c with bh_acos(c) >= thr (host, binary search): bh_acos(c) < thr <=> c > acos_cstar on [-1, 1]
    float acos_cstar_near; // the same for thr * temporal_margin (<= thr): temporal speculation also predicts the pixels this close to being traced
    int model_count;
    TexDev temp, disk, sky;
    ModelDev models[BHRAY_MAX_MODELS];
};

// One ladder level (one RayPipeline of the reference, ray_pipeline.rs:36-309).
struct LevelParams {
    int w, h;              // level size (textureDimensions(color_buffer))
    int pw, ph;            // previous level size; 1x1 => base case (ray.wgsl:178)
    float rx, ry;          // size_ratio (ray.wgsl:187), computed on the host in binary32
    const float4* prev;    // previous level, full pw*ph
    float4* out;           // this level's pixels
    int out_pitch;         // pixels per output row
    int out_x0;            // output column = x - out_x0
    const int32_t* rowmap; // level row y -> output row (nullptr: identity)
    const int32_t* rows;   // rows to compute (sorted), nrows entries
    int nrows;
    int x0, x1;            // columns to compute
    int tag;               // speculative mode: level id stored in bits 30-31 of the queue entries
    const float4* spec;    // speculative mode: already traced pixels of this level; a pixel that needs tracing copies from here
    int pass;              // CLASSIFY_NORMAL / CLASSIFY_TENTATIVE / CLASSIFY_KEEP (superset speculation, bhray_config.superset_levels)
    int no_store;          // tentative pass of the last level: nobody reads its image, only the queue entries are produced
};
enum : int { CLASSIFY_NORMAL = 0,
             // the coarser level may hold PENDING pixels (alpha = 2: queued for tracing, value not known yet).  A pixel whose inputs are all
             // known is classified exactly (copy / interpolate -> stored; trace -> queued, PENDING stored); a pixel with a PENDING
             // input is queued conservatively (a copy pixel: just marked PENDING).  The queued set is a superset of the exact one.
             CLASSIFY_TENTATIVE = 1,
             // after the trace of the superset: the coarser level is final.  Copy / interpolate are stored (overwriting speculative
             // traces the exact classification does not want); a pixel that needs tracing is left as the trace launch wrote it.
             CLASSIFY_KEEP = 2,
             // temporal speculation: exact classification (the coarser level is final).  Copy / interpolate are stored; a pixel that
             // needs tracing is recorded for the next frame's prediction and, unless this frame's predicted launch already traced it
             // (stamp), queued for this level's own trace launch.
             CLASSIFY_FIXUP = 3 };
#define BHRAY_PENDING_ALPHA 2.0f   // alpha of a final pixel is exactly 0 or 1

// Speculative tracing of several levels in one launch: per-level geometry and destination, selected by the entry's tag.
struct SpecLevel { int w, h; float4* out; int out_pitch; int out_x0; const int32_t* rowmap;    // rowmap/out_x0 as in LevelParams
                   uint32_t* stamp;      // temporal speculation: stamp[y*w + x] = FrameLaunch::stamp_value when the pixel is stored
                   unsigned long long* row_work; };   // counting builds: as FrameLaunch::row_work, for this level
struct SpecLevels { int n; SpecLevel l[BHRAY_MAX_SPEC_LEVELS]; };   // n == 0: one level, described by LevelParams

struct Counters64 { unsigned long long v[13]; };   // order = bhray_counters

// One frame's share of one launch.  A launch covers the `nb` frames of a batch: classify uses blockIdx.y as the frame
// index, the persistent trace blocks start on frame blockIdx.x % nb and move on to the other frames when theirs runs dry.
struct FrameLaunch {
    LevelParams L;
    SpecLevels SL;
    uint32_t* queue;       // ray queue of this frame and level(s)
    uint32_t* qctl;        // [0] entries appended (classify), [1] entries taken (trace)
    Counters64* counters;  // nullptr unless BHRAY_F_COUNTERS
    unsigned long long* row_work;   // counting builds: row_work[y] += iterations of every ray traced for level row y (bhray_get_row_work); nullptr: off
    // temporal speculation (BHRAY_F_TEMPORAL): the exact classification records every pixel that needs tracing for the NEXT frame's
    // predicted launch, and sends to this frame's queue only those the predicted launch has not traced already
    uint8_t* need;         // this level's per-pixel mark "the shader traces this pixel" (written by the exact classification, read by predict_kernel)
    int radius;            // predict_kernel: dilation of the previous frame's traced set, in pixels of this level
    const uint32_t* stamp; // this level's stamp image (classify); nullptr outside temporal mode
    uint32_t stamp_value;  // stamp of the current frame
    int probe_empty;       // trace, bit 0: this launch is expected to find its queue (nearly) used up - look before the first atomic; bit 1: thin shares are dealt strided (a whole frame, one frame per launch); bits 2-4: waves per SIMD the quad march may use (bhray_quad.inc); bit 5: waves set their issue priority by their rays' predicted length
    int blocks;            // predict (one launch, all levels): this level's own block count
    unsigned long long* span; // trace, entry 0 of a timed launch: [0] max(~first block start) [1] max(last block end), device wall clock; nullptr: untimed
    unsigned long long* work; // trace, entry 0 of a launch: work[blockIdx & (BHRAY_WORK_WORDS - 1)] += integrator steps this wave ISSUED for the frames of the batch
                              // (a step costs the wave the same whether 1 or 64 of its lanes march): what the batch's rays cost this GPU, whatever else shares
                              // it - the measure bhray_rebalance balances.  Counted in whole batches of BHRAY_REL_BATCH steps.
};
#define BHRAY_WORK_WORDS 16      // counters per frame (the waves of a launch spread over them: one word would serialise 2 048 atomics at every launch's end)
#define BHRAY_QCTL_WORDS (2 * BHRAY_MAX_LEVELS + 2 * BHRAY_WORK_WORDS)   // 32-bit words of a frame's control block: queue counts / heads, then the work counters (64-bit)

// classify / predict: a 256-thread block covers a rectangle of BX x BY 8x8-pixel tiles (4 waves, BX*BY/4 tiles each in turn)
#ifndef BHRAY_CLASSIFY_BX
#define BHRAY_CLASSIFY_BX 4
#endif
#ifndef BHRAY_CLASSIFY_BY
#define BHRAY_CLASSIFY_BY 2    // measured at 1080p RK, 20 slots: 4x1 5 315, 2x2 5 290, 4x2 5 406, 2x4 5 380, 4x4 5 360, 8x1 5 328, 8x2 5 345 Mrays/s
#endif
// launchers (bhray_kernels.hip); Pb / Fb are device arrays of nb entries
hipError_t launch_classify(const FrameParams* Pb, const FrameLaunch* Fb, int nb, int blocks, bool count, bool fixup, hipStream_t s);
hipError_t launch_trace(const FrameParams* Pb, const FrameLaunch* Fb, int nb, int method, bool models, bool count, bool dense, int eval, int* err_flag,
                        int grid_blocks, hipStream_t s);
int trace_blocks_per_cu(int method, int has_models, int count, int dense, int eval);   // eval: 0 contract, 1 BHRAY_F_LITERAL, 2 BHRAY_F_EVAL_FMA
// copies n16 16-byte words from pinned host memory to device memory with a kernel (stays on the compute queue: a DMA copy
// between the launches of a stream costs a cross-engine handshake each time)
// and zeroes `nzero` 32-bit words at `zero` (the queue control words of the batch) in the same launch
hipError_t launch_upload(const void* pinned_src, void* dst, size_t n16, uint32_t* zero, size_t nzero, hipStream_t s);
hipError_t launch_predict(const FrameParams* Pb, const FrameLaunch* Fb, int nb, int levels, int blocks, hipStream_t s);   // temporal speculation: F.need -> F.queue
hipError_t launch_selftest(unsigned long long* bad3, hipStream_t s);   // [0]: 1/x mismatches, [1]: sqrt mismatches, [2]: places where bh_acos increases
hipError_t launch_sky(const TexDev& sky, const float4* src, uint2* dst_rgba16f, size_t npix, hipStream_t s);

}  // namespace bhray
