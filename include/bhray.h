/*
 * bhray.h — C ABI of libbhray: the MI355X-native geodesic ray-trace pass.
 *
 * This is the drop-in boundary for ONE path of cleggacus/bhusie: the ray pass that the
 * reference runs as `RayPipeline` (src/renderer/pipelines/ray_pipeline.rs:28-310) over the
 * compute shader src/renderer/shaders/ray.wgsl:1-847, driven by `Renderer::new/render`
 * (src/renderer/mod.rs:170-207, 378-420).  The Rust host keeps its window, UI and scene graph
 * and calls these functions instead of wgpu for the ray pass (binding stub: INTEGRATION.md).
 *
 * Conventions
 *   - every function returns 0 (BHRAY_OK) or a negative BHRAY_E_* code; nothing unwinds or
 *     aborts across the boundary (the reference panics: mod.rs:66,75,89; model.rs:17).
 *   - all pointer arguments are borrowed for the duration of the call only.
 *   - a ctx is used from one thread at a time (the reference is single-threaded: app.rs:108-114).
 *   - byte layouts of the uniform blocks are the reference's #[repr(C)] structs, so the host
 *     passes `bytemuck::bytes_of(..)` unchanged.
 *   - there is NO CPU fallback: bhray_create fails with BHRAY_E_NO_DEVICE when no gfx950
 *     device is usable.
 */
#ifndef BHRAY_H
#define BHRAY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHRAY_VERSION_MAJOR 0
#define BHRAY_VERSION_MINOR 4

/* ------------------------------------------------------------------------------------------
 * Error codes
 * ---------------------------------------------------------------------------------------- */
enum {
    BHRAY_OK            = 0,
    BHRAY_E_INVALID     = -1,  /* bad argument / bad config                                  */
    BHRAY_E_NO_DEVICE   = -2,  /* no usable HIP device (no CPU fallback exists)              */
    BHRAY_E_HIP         = -3,  /* a HIP runtime call failed; see bhray_last_error            */
    BHRAY_E_NOMEM       = -4,
    BHRAY_E_STATE       = -5,  /* call order violated (e.g. render before set_uniforms)      */
    BHRAY_E_BVH_DEPTH   = -6,  /* BVH deeper than BHRAY_BVH_STACK levels                     */
    BHRAY_E_IO          = -7,  /* file could not be read / parsed (OBJ loader)               */
    BHRAY_E_CAPACITY    = -8,  /* model exceeds the reference's fixed capacities             */
    BHRAY_E_COMM        = -9   /* RCCL could not be loaded / a collective call failed / the gather did not complete within
                                      BHRAY_COMM_TIMEOUT_MS (watchdog, below): the ctx is failed, destroy it                    */
};

/* ------------------------------------------------------------------------------------------
 * Byte layouts shared with the reference (all little-endian, f32 = IEEE binary32)
 * ---------------------------------------------------------------------------------------- */

/* RayDetails — src/renderer/pipelines/ray_pipeline.rs:3-14  ⇄  ray.wgsl:25-34 (`Details`). */
typedef struct bhray_details {
    int32_t material_count;
    int32_t model_count;
    float   time;
    int32_t integration_method;        /* 0 Euler, 1 "Runge Kutta" (Cash–Karp), ray.wgsl:29 */
    float   step_size;
    int32_t max_iterations;
    float   angle_division_threshold;
    int32_t highlight_interpolation;   /* no-op in the reference, ray.wgsl:230-234          */
} bhray_details;                        /* 32 B */

/* CameraUniform — src/scene/camera.rs:66-73  ⇄  ray.wgsl:41-45. */
typedef struct bhray_camera_uniform {
    float    position[3];
    uint32_t _padding;
    float    forward[3];
    float    fov;
} bhray_camera_uniform;                 /* 32 B */

/* BlackHoleUniform — src/scene/blackhole.rs:37-51  ⇄  ray.wgsl:112-123. */
typedef struct bhray_black_hole_uniform {
    float   accretion_disk_inner;
    float   accretion_disk_outer;
    float   rotation_speed;
    float   relativity_sphere_radius;
    float   position[3];
    int32_t show_disk_texture;
    float   normal[3];
    int32_t show_red_shift;
    float   rotation_matrix[12];        /* 3 columns, each padded to vec4 (mat3x3 in WGSL)  */
    float   feather_amount;
    int32_t pad[8];
} bhray_black_hole_uniform;             /* 132 B */

/* NodeUniform — src/renderer/triangle.rs:45-52  ⇄  ray.wgsl:85-90. */
typedef struct bhray_node {
    float   min_corner[3];
    int32_t left_child;                 /* inner: index of first child (second = +1); leaf: first bvh_lookup slot */
    float   max_corner[3];
    int32_t obj_count;                  /* 0 ⇒ inner node                                   */
} bhray_node;                           /* 32 B */

/* Triangle (index record) — src/renderer/triangle.rs:54-63  ⇄  ray.wgsl:67-74. */
typedef struct bhray_triangle {
    int32_t p1, p2, p3;
    int32_t n1, n2, n3;
} bhray_triangle;                       /* 24 B */

/* ModelUniform — src/renderer/triangle.rs:268-285; the storage buffer bound at ray.wgsl:9.
 * Fixed capacity arrays; every size and offset below is static_assert-ed in bhusie_amd/csrc/bhray_layout.cpp. */
#define BHRAY_MAX_MODEL_VERTICES 524288  /* triangle.rs:7, ray.wgsl:1 */
#define BHRAY_MAX_MODELS         1       /* triangle.rs:6, ray.wgsl:2 */
#define BHRAY_MAX_MATERIALS      8       /* material.rs:3, ray.wgsl:3 */
#define BHRAY_MODEL_UNIFORM_BYTES 48234572u
#define BHRAY_MODEL_OFF_POINTS    48u
#define BHRAY_MODEL_OFF_NORMALS   (48u + 16u * BHRAY_MAX_MODEL_VERTICES)
#define BHRAY_MODEL_OFF_TRIANGLES (48u + 32u * BHRAY_MAX_MODEL_VERTICES)
#define BHRAY_MODEL_OFF_NODES     (48u + 56u * BHRAY_MAX_MODEL_VERTICES)
#define BHRAY_MODEL_OFF_LOOKUP    (48u + 88u * BHRAY_MAX_MODEL_VERTICES)

typedef struct bhray_model_header {     /* first 48 bytes of ModelUniform (Rust field order) */
    float    position[3];
    int32_t  visible;
    float    rotation[3];               /* uploaded, never applied by the shader (ray.wgsl:56) */
    uint32_t pad3;
    int32_t  point_count;
    int32_t  normal_count;
    int32_t  triangle_count;
    uint32_t pad0;
} bhray_model_header;                   /* 48 B */

/* Compact model: the same arrays, sized to the actual counts (what stays resident in HBM). */
typedef struct bhray_model_desc {
    float                 position[3];
    int32_t               visible;
    const float*          points;       /* point_count  × 4 f32 (xyz + pad)                 */
    const float*          normals;      /* normal_count × 4 f32                             */
    const bhray_triangle* triangles;    /* triangle_count                                    */
    const bhray_node*     nodes;        /* node_count                                        */
    const int32_t*        bvh_lookup;   /* triangle_count                                    */
    int32_t point_count, normal_count, triangle_count, node_count;
} bhray_model_desc;

/* ------------------------------------------------------------------------------------------
 * Context configuration
 * ---------------------------------------------------------------------------------------- */
#define BHRAY_MAX_LEVELS 8
#define BHRAY_MAX_FRAMES_IN_FLIGHT 32
#define BHRAY_MAX_SPEC_LEVELS 4
#define BHRAY_MAX_FRAMES_PER_BATCH 32
#define BHRAY_BVH_STACK  64            /* tree levels the traversal's restart trail covers (reference: a stack of 19 whole nodes, no overflow check, ray.wgsl:292) */
#define BHRAY_MAX_DEVICES 16           /* GPUs one ctx can drive (one node: 8 MI355X)                 */
#define BHRAY_COMM_ID_BYTES 128        /* an RCCL ncclUniqueId                                         */

enum {                                  /* bhray_config.flags */
    BHRAY_F_COUNTERS   = 1u << 0,       /* kernels also accumulate bhray_counters (slower)   */
    BHRAY_F_TIMING     = 1u << 1,       /* record HIP events around every launch             */
    BHRAY_F_TIMING_SPARSE = 1u << 4,    /* like BHRAY_F_TIMING, but only every 4th batch carries events (a recorded event is a packet in
                                           the stream: 12 per frame cost a saturated device 1.6 %); bhray_get_timing aggregates those */
    BHRAY_F_TEMPORAL   = 1u << 3,       /* temporal speculation: one launch first traces, at every level, the pixels the previous frame
                                           had to trace; the ladder then only traces what that prediction missed.  Same pixels; the
                                           chain of dependent trace launches collapses when consecutive frames are similar (an
                                           interactive host, one frame at a time).  The prediction is a superset of last frame's
                                           traced set (pixels close to the interpolation threshold, a few pixels around, the clamped
                                           border pixels: DESIGN.md §4), +8 % rays; 1080p, one frame at a time: 0.78 ms with a static
                                           camera, 0.84-0.87 ms with a moving one, 1.20 ms without the flag.  levels <= 4, no
                                           speculative / superset levels.                                                    */
    /* (1u << 6 was BHRAY_F_FUSED, the fused ladder of rounds 3-5 - one persistent launch per batch with tile dependencies instead of
       launch boundaries: measured slower on MI355X, 2.1 against 1.2 ms for one 1080p frame at a time, profiles/EXPERIMENTS.md R3.1 - removed
       in round 6, kept as a patch: profiles/variants_src/.  bhray_create refuses the bit.)                                              */
    BHRAY_F_EVAL_FMA   = 1u << 5,       /* a THIRD evaluation of the integrator: the shader text with fused multiply-add contraction only
                                           (every `x*y + z` of ray.wgsl:401-480 one fma), none of the contract's reassociations (N9/N10).
                                           Like BHRAY_F_LITERAL it exists for measurement: the pixels on which it differs from the literal
                                           text by more than 1e-4 are the pixels on which the default evaluation does
                                           (tests/test_gpu_literal.py).  Ignored when BHRAY_F_LITERAL is set.                        */
    BHRAY_F_GATHER_SKY = 1u << 7,       /* multi-GPU ctx only (device_count >= 2, or gather = BHRAY_GATHER_RCCL): gather the RGBA16F image of the
                                           sky pass instead of the RGBA32F frame - HALF the bytes over xGMI.  Every partition runs the sky
                                           pass (sky.wgsl, per pixel) over its own rows behind its render and sends 8-byte pixels; the root
                                           assembles the sky image.  bhray_resolve_sky is then implied by bhray_render (calling it is
                                           allowed and does nothing); bhray_read_sky / bhray_read_sky_async / bhray_sky_device_ptr deliver the
                                           assembled image; the RGBA32F frame is NOT assembled: bhray_read_hdr, bhray_read_hdr_async,
                                           bhray_hdr_device_ptr and bhray_bind_output return BHRAY_E_STATE.  For a host that lets the library
                                           run the sky pass.  Same pixels as the sky pass over the assembled frame.  */
    BHRAY_F_ALL        = 0xbfu,         /* every flag above and below: bhray_create refuses any other bit                    */
    BHRAY_F_LITERAL    = 1u << 2        /* the integrator (ray.wgsl:401-480, 533) operator by operator: one binary32 operation per
                                           WGSL operator in source order, no fused multiply-add, no reassociation.  Slower; exists
                                           to MEASURE how far the default evaluation (DESIGN.md §2, N3/N7/N9/N10 — permitted by
                                           WGSL, cheaper on CDNA4) is from the shader text: tests/test_gpu_literal.py.  Cost: bench.py's
                                           `literal` entry                                                          */
};

/* The ladder is the reference's chain of RayPipelines (mod.rs:170-207): level 0 traces every
 * pixel (its t_prev is the 1×1 base texture, ray.wgsl:178), level k>0 reads level k-1.
 * The delivered frame is the window [crop_x, crop_x+frame_w) × [crop_y, crop_y+frame_h) of
 * the last level; pixels outside it (and coarse pixels no window pixel depends on) are not
 * computed.  With crop = 0 and frame = last level size this is exactly the reference.
 *
 * Speculative levels.  The ladder is a chain of dependent launches whose coarse levels are latency-bound (a level
 * takes as long as its longest ray, however few rays it has).  With speculative_levels = S the pixels of levels
 * 0..S-1 are all traced in ONE launch before any of them is classified; classification then selects, per pixel, the
 * copy / the interpolation / the already traced value exactly as the shader would.  Same pixels, fewer dependent
 * launches, more rays traced (bhray_counters then count the speculative work).  Meant for small per-GPU frames
 * (row-tiled multi-GPU); off by default.
 *
 * Superset speculation.  speculative_levels shortens the chain at its coarse end by tracing everything; at the fine end
 * that would trace 6x too many rays.  With superset_levels = U the last U levels are first classified TENTATIVELY in order - a
 * pixel whose coarser inputs are known is classified exactly, a pixel with an input that is itself queued is queued
 * conservatively - then traced in ONE launch, then classified again exactly (copy / interpolate stored, traced pixels kept).
 * The queued set contains every pixel the shader would trace, so the frame is unchanged; the surplus (conservatively queued
 * pixels that turn out to interpolate) is a few per cent of the rays.  One dependent trace launch instead of U: meant for one
 * frame at a time (an interactive host) and small per-GPU frames.  speculative_levels + superset_levels < levels.
 *
 * Frame batches.  With frames_per_batch = B > 1, bhray_render only STAGES a frame (uniforms, output binding); the
 * launches are enqueued once B frames are staged (or bhray_flush / any call that waits for or reads a frame is made),
 * and every launch then covers the B frames: B times fewer dependent launches per frame and B times more rays per
 * launch.  Frames of a batch may have different uniforms; pixels are identical to B = 1.  For throughput rendering of
 * small per-GPU frames (row-tiled multi-GPU, offline sequences); an interactive host keeps B = 1.  A consumer that
 * orders its own work after a frame through bhray_next_stream must call bhray_flush before enqueueing that work.
 *
 * Watchdog (a ctx that gathers: device_count >= 2 or gather = BHRAY_GATHER_RCCL).  A peer that never posts its share of a batch's
 * RCCL group would make ncclGroupEnd or the communication stream - and with it bhray_sync and every read - wait for ever.  Every call
 * of the ABI on such a ctx, and every frame one of its issue threads enqueues, therefore carries a deadline of BHRAY_COMM_TIMEOUT_MS
 * milliseconds (environment, read by bhray_create; default 30000, 0 = no watchdog).  When a call is overdue a thread of the ctx aborts
 * its communicators (ncclCommAbort: RCCL's blocked host calls return, its kernels leave their spin loops), writes one line to stderr
 * and fails the ctx: the overdue call and every later one return BHRAY_E_COMM with the watchdog's message in bhray_last_error;
 * bhray_destroy still works.  Pick the deadline above the longest single call the host makes (a bhray_sync behind hundreds of staged
 * 8K frames is seconds).  Not covered: ncclCommInitRank / ncclCommInitAll inside bhray_create (no communicator to abort yet).
 *
 * Row partition (multi-GPU row tiling).  partition = BHRAY_PARTITION_STRIPES (default): frame row r belongs to partition
 * (r / stripe_rows) % row_world - interleaved stripes, balanced whatever the scene, at the price of coarse ladder rows that
 * several partitions compute (every stripe boundary costs two rows at each coarser level).  partition =
 * BHRAY_PARTITION_SLABS: partition p owns the consecutive rows [slab_row0[p], slab_row0[p + 1]) - one boundary per partition;
 * the bounds come from the host, which balances them by measured work (bhray_get_row_work of a calibration frame ->
 * bhray_balance_slabs; 1920x1080, 8 partitions, default scene: 2 % of the ray-steps computed twice and 2 % imbalance against
 * 7 % and 8.5 % for stripes of 27, profiles/partition_sim.py).  Either way this ctx renders only rows of partition row_rank
 * and packs them densely, in increasing r, into its output buffer.  row_world = 1 ⇒ the whole frame.
 *
 * Multi-GPU.  The reference host is one process on one thread (app.rs:108-114, mod.rs:415-420), so the
 * row tiling lives behind this ABI:
 *   device_count = N >= 2 — ONE ctx drives the N GPUs devices[0..N): partition i is rendered on devices[i] (scene and
 *     uniforms replicated, coarse ladder rows recomputed per partition, nothing exchanged during the levels), and every
 *     bhray_render also enqueues the gather of the row tiles to the GPU of partition gather_root (RCCL: grouped
 *     ncclSend/ncclRecv over xGMI, one message per partition per batch) and the de-interleave of the stripes into the
 *     frame (a HIP kernel on the root GPU).  On the wire a row of the RGBA32F frame is its x, y, z floats plus ONE BIT of alpha per
 *     pixel (alpha is exactly 0 or 1: ray.wgsl:589-594): 12.1 bytes per pixel instead of 16, packed behind the render on the sending
 *     GPU and unpacked by the de-interleave - the assembled frame is the same bits (bhray_gather_info counts the bytes that travel).  Output calls (bhray_read_hdr, bhray_hdr_device_ptr, bhray_bind_output,
 *     bhray_resolve_sky) then refer to the WHOLE frame on the root GPU; bhray_local_rows = frame_h.  row_rank/row_world
 *     are ignored (row_world is set to N).  A device may appear more than once (functional tests on a one-GPU box): its
 *     partitions share one RCCL rank and their tiles travel as send/recv-to-self.
 *   one process per GPU (a launcher such as torchrun/mpirun): every process creates its ctx with device_count <= 1,
 *     row_rank = its rank, row_world = N, gather = BHRAY_GATHER_RCCL and the SAME comm_id (bhray_comm_unique_id on one
 *     rank, distributed by the launcher's own means).  The gather is enqueued by bhray_render exactly as above; the frame
 *     exists on rank gather_root only (the other ranks' output calls succeed and deliver nothing).
 * RCCL is loaded (dlopen "librccl.so.1") when the first such ctx is created; a single-GPU host never loads it. */
enum { BHRAY_PARTITION_STRIPES = 0,     /* interleaved stripes of stripe_rows rows                                   */
       BHRAY_PARTITION_SLABS = 1 };     /* contiguous slabs bounded by slab_row0[0..row_world] (row_world <= BHRAY_MAX_DEVICES) */
enum { BHRAY_GATHER_NONE = 0,           /* row_world > 1: this ctx delivers its packed rows, the caller moves them   */
       BHRAY_GATHER_RCCL = 1 };         /* the library gathers (always on when device_count >= 2)                    */

typedef struct bhray_config {
    uint32_t struct_size;               /* = sizeof(bhray_config)                            */
    int32_t  device;                    /* HIP device ordinal (device_count == 0)            */
    uint32_t levels;                    /* 1..BHRAY_MAX_LEVELS                               */
    uint32_t level_w[BHRAY_MAX_LEVELS];
    uint32_t level_h[BHRAY_MAX_LEVELS];
    uint32_t crop_x, crop_y;
    uint32_t frame_w, frame_h;
    uint32_t row_rank, row_world, stripe_rows;
    uint32_t flags;
    uint32_t frames_in_flight;          /* 0 = default (4); 1 = strictly one frame at a time; more than 22 are served with 22  */
    uint32_t speculative_levels;        /* 0 = off; S>=2: trace EVERY needed pixel of levels 0..S-1 in one launch   */
    uint32_t frames_per_batch;          /* 0/1 = every bhray_render launches; B>1: launches cover B staged frames  */
    uint32_t superset_levels;           /* 0 = off; U>=2: the LAST U levels are traced in one launch over a conservative superset    */
    uint32_t device_count;              /* 0: one GPU, `device`; N: devices[0..N) (N >= 2: single-process multi-GPU) */
    int32_t  devices[BHRAY_MAX_DEVICES];
    uint32_t gather;                    /* BHRAY_GATHER_*: one process per GPU only (see above)                      */
    uint32_t gather_root;               /* partition whose GPU receives the frame (default 0)                        */
    uint8_t  comm_id[BHRAY_COMM_ID_BYTES]; /* one process per GPU: the communicator id shared by all ranks           */
    uint32_t partition;                 /* BHRAY_PARTITION_*                                                         */
    uint32_t slab_row0[BHRAY_MAX_DEVICES + 1]; /* BHRAY_PARTITION_SLABS: first frame row of every partition, then frame_h
                                           (non-decreasing; a partition may own no rows)                            */
} bhray_config;

/* Reference ladder rule `r ← r·m − (m−1)` (mod.rs:177-205): fills level_w/h[0..levels).    */
int bhray_ladder_from_base(uint32_t base_w, uint32_t base_h, uint32_t multiplier,
                           uint32_t levels, bhray_config* cfg);
/* Smallest reference-rule ladder whose last level covers frame_w × frame_h; the frame is the
 * centred window of it.  1918×1081/levels 4 gives base 72×41, crop 0 — the shipped config.  */
int bhray_ladder_for_frame(uint32_t frame_w, uint32_t frame_h, uint32_t multiplier,
                           uint32_t levels, bhray_config* cfg);

/* ------------------------------------------------------------------------------------------
 * Lifecycle — replaces RayPipeline::new ×levels (ray_pipeline.rs:36-295, mod.rs:181-207)
 * ---------------------------------------------------------------------------------------- */
typedef struct bhray_ctx bhray_ctx;

int  bhray_create(const bhray_config* cfg, bhray_ctx** out);
void bhray_destroy(bhray_ctx* ctx);
const char* bhray_last_error(const bhray_ctx* ctx);   /* ctx may be NULL: last create error  */
const char* bhray_strerror(int code);
uint32_t    bhray_version(void);                      /* major<<16 | minor                   */
int  bhray_device_count(void);                        /* usable gfx950 devices, ≥0           */

/* Row partition used by the multi-GPU modes (and by bhray_config.row_*): frame row r belongs to partition
 * (r / stripe_rows) % world.  Pure host arithmetic (no device): the de-interleave kernel uses the same functions.
 * bhray_partition_rows = rows of `part`; bhray_partition_row_index = frame row of the part's packed row i.        */
uint32_t bhray_partition_rows(uint32_t frame_h, uint32_t world, uint32_t stripe_rows, uint32_t part);
int bhray_partition_row_index(uint32_t frame_h, uint32_t world, uint32_t stripe_rows, uint32_t part, uint32_t i, uint32_t* frame_row);
/* The same for the partition a bhray_config describes (stripes or slabs; world = device_count when >= 2, else row_world). */
uint32_t bhray_config_partition_rows(const bhray_config* cfg, uint32_t part);
int bhray_config_partition_row_index(const bhray_config* cfg, uint32_t part, uint32_t i, uint32_t* frame_row);
/* Slab bounds balanced by measured work.  row_work[l] points to level_h[l] numbers: the work (ray-steps, bhray_get_row_work) of
 * every row of ladder level l of a calibration frame rendered WHOLE with the speculative_levels the partitions will use.  A
 * partition's work is the work of the level rows its frame rows depend on (every level: the ladder arithmetic of cfg, crop
 * included); the bounds minimise the largest partition's work over all contiguous partitions.  Pure host arithmetic.
 * Writes slab_row0[0..world] (slab_row0[0] = 0, slab_row0[world] = frame_h); the caller copies them into bhray_config.slab_row0. */
int bhray_balance_slabs(const bhray_config* cfg, const uint64_t* const* row_work, uint32_t world, uint32_t* slab_row0);
/* Run-time partition: the bounds can follow the scene (the reference's camera moves every frame, src/app.rs:98-102; the rows the
 * hole and the disk project to are where the work is).  None of these re-creates the ctx; frames before and after are the same pixels.
 *
 * bhray_set_partition: from the next bhray_render on, partition p owns the frame rows [slab_row0[p], slab_row0[p + 1]) (contiguous
 *   slabs whatever the ctx was created with; partitions = device_count, or row_world; a partition may own no rows).  Waits for
 *   everything enqueued so far, rewrites the row tables of the engines and the gather tables in place; per-frame queues, send and
 *   staging buffers grow when the new rows need more (by a quarter more than needed, so that bounds that keep moving a few rows do
 *   not reallocate).  One process per GPU: every rank calls it with the same bounds before its next bhray_render. 
 * bhray_rebalance_slabs: pure host arithmetic behind bhray_rebalance, exposed so that a host can balance by its own measurements.
 *   row_weight[frame_h] is the caller's persistent estimate of what every frame row costs (all zero before the first call); the call
 *   rescales the rows of every partition so that their sum is that partition's measured part_cost (the shape inside a partition is
 *   kept: it is what earlier calls learned), then finds the bounds that minimise the largest (sum of weights + extra_cost[p]) over
 *   contiguous partitions.  extra_cost (may be NULL) = work of a partition that does not move with its rows (the root's gather).
 *   shift_rows: the frames to come are expected to show what was measured this many rows further down (a camera that pitches moves
 *   the rows the hole projects to); the learned weights are shifted before the bounds are found.  0 for a scene at rest.
 * bhray_get_work: what the frames still held by the frame slots cost this ctx's partition(s), per frame: the integrator steps its trace
 *   waves ISSUED (counted by the kernels in every build: a wave pays for a step whether 1 or 64 of its lanes march - this is the GPU
 *   time of the march, independent of what else shares the GPU and of how many frames are in flight) and the pixels its classify
 *   launches visit (an HBM-bound pass).  Waits for the frames in flight.
 * bhray_rebalance: new bounds from that measure, per partition, in wave-steps: steps issued + classified pixels at their price in
 *   wave-steps (0.035 for the RK kernel on MI355X, 0.06 for Euler) and, on the root, the pixels it receives and de-interleaves (0.05
 *   each) as work that does not move with its rows; shift_rows = how far the frame row the hole projects to (from the uniforms of
 *   bhray_set_uniforms) moved since the previous call.  No counting build, no calibration frame, no timing flags.  The new bounds are
 *   applied (bhray_set_partition) when they promise at least 2 % less for the slowest partition.  One process per GPU: collective -
 *   every rank calls it at the same point of its frame sequence; the numbers travel over the ctx's communicator (one small all-gather).
 *   Call it every second or so of frames: it waits for the frames in flight.                                                      */
typedef struct bhray_rebalance_info {
    uint32_t partitions;
    uint32_t applied;                                 /* 1: the partition was changed                                              */
    uint32_t slab_row0[BHRAY_MAX_DEVICES + 1];        /* the bounds in force after the call                                         */
    float    part_cost[BHRAY_MAX_DEVICES];            /* measured, wave-steps per frame: steps issued + classified pixels at their price */
    float    extra_cost[BHRAY_MAX_DEVICES];           /* modelled, same unit: work that does not move with the rows (root: receive + de-interleave) */
    float    slowest_before, slowest_predicted;       /* max over partitions of part_cost + extra_cost; what the new bounds promise */
    uint32_t frames;                                  /* frames the measurement covers (partition of this rank / the root)          */
} bhray_rebalance_info;
int bhray_set_partition(bhray_ctx* ctx, const uint32_t* slab_row0 /* partitions + 1 entries */);
int bhray_rebalance_slabs(uint32_t frame_h, uint32_t partitions, const uint32_t* slab_row0, const double* part_cost, const double* extra_cost,
                          double shift_rows, double* row_weight, uint32_t* slab_row0_out, double* slowest_predicted /* may be NULL */);
int bhray_get_work(bhray_ctx* ctx, double* wave_steps_per_frame, double* classify_pixels_per_frame, uint32_t* frames /* may be NULL */);
int bhray_rebalance(bhray_ctx* ctx, bhray_rebalance_info* out /* may be NULL */);
/* The numbers bhray_rebalance works with, for the partitions THIS ctx renders (cost[q] and extra[q] of the others are 0; both arrays hold
 * `partitions` entries): a launcher that has a channel of its own between its ranks (MPI, torch.distributed) sums the ranks' arrays,
 * calls bhray_rebalance_slabs and hands every rank the same bounds for bhray_set_partition - no collective inside the library.     */
int bhray_get_partition_costs(bhray_ctx* ctx, double* cost, double* extra, uint32_t* frames /* may be NULL */);
/* The partition in force: partitions + 1 bounds when it is contiguous slabs; BHRAY_E_STATE for interleaved stripes.              */
int bhray_get_partition(const bhray_ctx* ctx, uint32_t* slab_row0 /* BHRAY_MAX_DEVICES + 1 entries */, uint32_t* partitions);

/* One process per GPU: a fresh communicator id (ncclGetUniqueId); call on ONE rank, hand the bytes to all ranks.  */
int bhray_comm_unique_id(uint8_t id[BHRAY_COMM_ID_BYTES]);
/* What a ctx gathers with.                                                                                          */
typedef struct bhray_gather_info {
    uint32_t partitions;                /* row partitions of the frame (1 = no tiling)                               */
    uint32_t local_partitions;          /* partitions rendered by this ctx                                           */
    uint32_t root;                      /* partition that receives the frame                                         */
    uint32_t root_is_local;             /* 1: the frame is delivered by this ctx                                     */
    uint32_t comm_ranks;                /* ranks of the RCCL communicator (0: no gather)                             */
    uint32_t rccl_version;              /* ncclGetVersion, e.g. 22707 (0: RCCL not loaded)                           */
    uint64_t bytes_sent_per_frame;      /* by this ctx's non-root partitions                                         */
    uint64_t bytes_received_per_frame;  /* by the root partition (0 when it is not local)                            */
} bhray_gather_info;
int bhray_get_gather_info(const bhray_ctx* ctx, bhray_gather_info* out);

/* Static inputs — replaces the include_bytes! textures (ray_pipeline.rs:63-70) and
 * texture.rs:16-69 semantics: RGBA8 unorm, no sRGB decode, bilinear, clamp-to-edge, 1 mip.  */
enum { BHRAY_TEX_TEMP_LUT = 0,   /* binding 7  color.png */
       BHRAY_TEX_DISK     = 1,   /* binding 9  disk.png  */
       BHRAY_TEX_SKY      = 2 }; /* binding 12 sky.png   */
int bhray_set_texture(bhray_ctx* ctx, int slot, const uint8_t* rgba8, uint32_t w, uint32_t h);

/* Model upload — replaces `scene.models.create_buffer/update_buffer` (mod.rs:114,391,
 * array_buffer.rs:71-89).  Either the exact 48 234 572-byte ModelUniform or the compact form. */
int bhray_upload_model_uniform(bhray_ctx* ctx, uint32_t model_index, const void* bytes, size_t size);
int bhray_upload_model(bhray_ctx* ctx, uint32_t model_index, const bhray_model_desc* desc);
/* Per-frame model state without re-uploading 48 MB (the reference re-uploads, mod.rs:391).  */
int bhray_set_model_transform(bhray_ctx* ctx, uint32_t model_index, const float position[3], int32_t visible);

/* Materials — `queue.write_buffer(&self.material_buffer, ..)` (mod.rs:113,389; material.rs:7-38: MaterialUniform
 * {color:[f32;4]} x 8 = 128 B, binding 3).  The shader never reads them (no use of `materials` after ray.wgsl:8), so the
 * bytes are accepted, size-checked and ignored; the call exists so that the Rust call order maps one to one.      */
int bhray_set_materials(bhray_ctx* ctx, const void* material_uniforms_128, size_t size);

/* Per-frame uniforms — replaces queue.write_buffer ×3 (mod.rs:386-388).                      */
int bhray_set_uniforms(bhray_ctx* ctx, const void* camera_uniform_32,
                       const void* black_hole_uniform_132, const void* ray_details_32);

/* Dispatch — replaces `for rp in ray_pipelines { rp.pass() }` (mod.rs:415-417,
 * ray_pipeline.rs:301-309).  Asynchronous on the ctx stream; levels ordered.                 */
int bhray_render(bhray_ctx* ctx);
/* frames_per_batch > 1: enqueue the launches of the frames staged so far (a partial batch).  No-op otherwise.
 * bhray_sync, the bhray_read_* calls, bhray_resolve_sky, bhray_signal_stream and the texture / model uploads flush
 * by themselves.                                                                              */
int bhray_flush(bhray_ctx* ctx);
int bhray_sync(bhray_ctx* ctx);

/* Output — replaces RayPipeline::output_view (ray_pipeline.rs:297-299): RGBA32F,
 * row 0 = top, x fastest (textureStore(screen_pos), ray.wgsl:182).  Rows of this ctx's
 * partition only, packed; local_rows = bhray_local_rows().  Synchronises the stream.        */
int bhray_read_hdr(bhray_ctx* ctx, float* dst_rgba32f, size_t row_pitch_bytes);
/* Any ladder level, full size level_w×level_h (unrendered pixels are NaN-filled at create).  */
int bhray_read_level(bhray_ctx* ctx, uint32_t level, float* dst_rgba32f, size_t row_pitch_bytes);
uint32_t bhray_local_rows(const bhray_ctx* ctx);
/* frame row index of packed row i (0 ≤ i < local_rows).                                      */
int bhray_local_row_index(const bhray_ctx* ctx, uint32_t i, uint32_t* frame_row);

/* Zero-copy consumers (sky pass, RCCL gather).  An output buffer holds local_rows × frame_w × 4
 * f32.  bhray_hdr_device_ptr returns the buffer of the most recently enqueued frame.
 * bhray_bind_output makes the NEXT bhray_render — that one frame only — write into caller-supplied device
 * memory (on the GPU that delivers the frame); later frames go to the ctx-owned buffers again unless bound
 * again.  NULL cancels a pending binding.                                                     */
int bhray_hdr_device_ptr(bhray_ctx* ctx, void** dev_ptr, size_t* bytes);
int bhray_bind_output(bhray_ctx* ctx, void* dev_ptr, size_t bytes);

/* Hand-off to a consumer that is NOT a HIP client of the same GPU (the reference's SkyPipeline samples the ray output as a wgpu
 * texture: ray_pipeline.rs:297-299, mod.rs:215, sky.wgsl:4,17).  Two ways, cheapest first:
 *
 * (1) zero copy: the consumer exports the memory behind its texture / buffer as a file descriptor (Vulkan
 *     VK_KHR_external_memory_fd: an OPAQUE_FD or dma-buf of a linear VkBuffer, which the host then copies to or aliases with its
 *     Rgba32Float texture on its own queue); bhray_import_external_fd maps it into the address space of the GPU that delivers
 *     the frame (hipImportExternalMemory) and returns a device pointer for bhray_bind_output.  The library imports a dup() of the
 *     descriptor: hipImportExternalMemory follows the CUDA rule that an imported fd belongs to the runtime afterwards, so the
 *     CALLER'S fd stays the caller's - usable and closable at any time after the call returns - whatever the runtime does with the
 *     duplicate (ROCm 7.2 does not say whether hipDestroyExternalMemory closes it: bhray_release_external / bhray_destroy close the
 *     duplicate themselves if its number still refers to the file it was duplicated from, so an import leaks no descriptor either way).  The handle type is the opaque-fd one also for dma-buf descriptors; where tests/test_gpu_handoff.py skips (export or
 *     import refused by the driver) this path is unverified on that system.  Ordering: bhray_sync, or an exported semaphore the host
 *     signals from a stream it ordered with bhray_signal_stream.
 * (2) asynchronous read-back: bhray_read_hdr_async enqueues the device->host copy of the most recently enqueued frame (SDMA: no CU
 *     time) in stream order behind that frame's kernels - on the frame's own slot stream, so the other slots' frames render while it
 *     runs - and returns at once with a ticket.  bhray_wait_read(ticket) blocks until that frame has landed.  `dst` must stay valid
 *     until then and should be pinned host memory (bhray_host_alloc, or the host's own hipHostRegister): a pageable destination makes
 *     the runtime stage the copy.  A slot's image is not overwritten before its copy has read it.  At most 64 tickets are outstanding
 *     (the 65th call waits for the oldest).  Multi-GPU ctx: the assembled frame on the root, behind its de-interleave.  Measured,
 *     1920x1080 RGBA32F (33.2 MB): 0.62 ms per frame with 4 frame slots = the link rate (55.7 GB/s), against 0.40 without the hand-off
 *     and 2.8 ms for bhray_read_hdr after every render. */
int bhray_read_hdr_async(bhray_ctx* ctx, float* dst_rgba32f, size_t row_pitch_bytes, uint64_t* ticket);
/* The same for the RGBA16F image of the sky pass (bhray_resolve_sky first): half the bytes - for a host that lets the library run the
 * sky pass too and uploads into the texture its SkyPipeline would have written (sky_pipeline.rs:17-148).  Tickets are shared.      */
int bhray_read_sky_async(bhray_ctx* ctx, uint16_t* dst_rgba16f, size_t row_pitch_bytes, uint64_t* ticket);
int bhray_wait_read(bhray_ctx* ctx, uint64_t ticket);
int bhray_host_alloc(size_t bytes, void** out);               /* pinned host memory (hipHostMalloc) for hosts that do not link HIP */
int bhray_host_free(void* p);
int bhray_import_external_fd(bhray_ctx* ctx, int fd, size_t bytes, void** dev_ptr);
int bhray_release_external(bhray_ctx* ctx, void* dev_ptr);

/* Frames in flight.  The ladder levels of ONE frame are dependent launches, and the coarse levels
 * are far too small to fill 256 CUs (level 0 is ~3 k rays), so a ctx keeps `frames_in_flight`
 * frame slots, each with its own HIP stream and level/queue buffers; consecutive bhray_render
 * calls go to consecutive slots and overlap on the device (the reference's swap chain runs with
 * desired_maximum_frame_latency = 2, mod.rs:101).  ROCm maps HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware
 * queues and aliased streams serialise: a host that keeps more than 2 frames in flight exports
 * GPU_MAX_HW_QUEUES >= frames_in_flight + 2 before its first HIP call; the library never changes the process
 * environment.  It reads that variable: slots beyond GPU_MAX_HW_QUEUES - 2 share the streams of the first ones (frames on
 * one stream run in order), because MORE streams than queues do not merely alias, they collapse (measured: 24 slots on 24
 * queues lose 20 % of the throughput and a 20-frame burst takes 9x as long).  Ordering against the caller's own streams:
 *   bhray_wait_stream(ctx, s)    the NEXT bhray_render starts after everything enqueued on s so far
 *   bhray_signal_stream(ctx, s)  work enqueued on s from now on starts after the LAST bhray_render
 * `s` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the legacy stream.  */
int bhray_wait_stream(bhray_ctx* ctx, void* hip_stream);
int bhray_signal_stream(bhray_ctx* ctx, void* hip_stream);
/* The hipStream_t of the slot the NEXT bhray_render will use.  A caller that enqueues its own work for that frame on
 * this stream (e.g. wraps it as torch.cuda.ExternalStream and issues the RCCL gather there) needs no extra ordering
 * (with frames_per_batch > 1: all frames of a batch share the stream, and the caller flushes before enqueueing). */
int bhray_next_stream(bhray_ctx* ctx, void** hip_stream);

/* Sky resolve — the compute pass that follows the ray pass in the reference (shaders/sky.wgsl:1-38,
 * pipelines/sky_pipeline.rs:17-148, dispatched right after the ray levels at mod.rs:419): alpha == 0 pixels carry
 * an escape direction and become sky^4 (alpha 1), other pixels pass through; target format rgba16float.
 * bhray_resolve_sky enqueues it behind the most recently enqueued frame (same slot, same stream) into that slot's
 * RGBA16F image: local_rows x frame_w x 4 binary16, round-to-nearest-even.  The image belongs to that frame: after the
 * next bhray_render the reads below return BHRAY_E_STATE until bhray_resolve_sky has run for the new frame (a copy
 * enqueued with bhray_read_sky_async BEFORE that render still delivers the earlier frame's image).                  */
int bhray_resolve_sky(bhray_ctx* ctx);
int bhray_read_sky(bhray_ctx* ctx, uint16_t* dst_rgba16f, size_t row_pitch_bytes);
int bhray_sky_device_ptr(bhray_ctx* ctx, void** dev_ptr, size_t* bytes);

/* ------------------------------------------------------------------------------------------
 * Measurement
 * ---------------------------------------------------------------------------------------- */
typedef struct bhray_counters {        /* summed over all levels of the last render          */
    uint64_t pixels;                   /* pixels written (all levels)                        */
    uint64_t copied;                   /* grid: copied from the coarser level (ray.wgsl:193) */
    uint64_t interpolated;             /* grid: bilinear mix of directions (ray.wgsl:217)    */
    uint64_t traced;                   /* pixels that ran trace_ray                          */
    uint64_t steps;                    /* relativity iterations (integrator steps)           */
    uint64_t flat_iters;               /* flat-space iterations                              */
    uint64_t node_pairs;               /* BVH inner-node visits (2 AABB tests each)          */
    uint64_t triangles;                /* hit_triangle calls                                 */
    uint64_t disk_hits;                /* accretion-disk shading events                      */
    uint64_t sky_samples;              /* in-kernel sky taps (ray.wgsl:587)                  */
    /* scheduling of the trace kernel (not a property of the frame: depends on frames in flight, batches, the kernel build)   */
    uint64_t wave_steps;               /* integrator steps issued by waves: `steps` / (64 * wave_steps) = fraction of the lanes
                                          of a stepping wave that hold a live ray                                            */
    uint64_t rays_adopted;             /* rays that changed wave through the drain-merging mailbox (dense build)             */
    uint64_t max_ray_iterations;       /* iterations of the longest ray (a maximum, also over levels): the latency floor of a level
                                          is its longest ray                                                                  */
} bhray_counters;
int bhray_get_counters(bhray_ctx* ctx, bhray_counters* out);   /* needs BHRAY_F_COUNTERS     */
int bhray_get_level_counters(bhray_ctx* ctx, uint32_t level, bhray_counters* out);
/* Where the work of the last render lies: out[y] = iterations of all rays traced for row y of ladder level `level` (n = level_h[level]
 * numbers; rows this ctx did not render are 0; a multi-partition ctx sums its local partitions).  Needs BHRAY_F_COUNTERS.
 * The input of bhray_balance_slabs.                                                                                              */
int bhray_get_row_work(bhray_ctx* ctx, uint32_t level, uint64_t* out, uint32_t n);

/* Device self-test of the properties two exact shortcuts rest on: (i) the integrator computes the correctly
 * rounded 1/x and sqrt(x) with short gfx950 sequences — run against the IEEE lowering on all 2^32 binary32 bit patterns;
 * (ii) the grid classification replaces `acos(c) < threshold` by `c > c*` — the portable acos must be monotone over every
 * binary32 value of [-1, 1].  Returns the number of violating inputs of each (all must be 0).  ~20 ms.              */
int bhray_selftest(bhray_ctx* ctx, uint64_t mismatches[3]);    /* [0] = 1/x and the step-size power, [1] = sqrt, [2] = acos monotonicity */

/* HIP-event timing of every launch (events recorded on the ctx stream).  bhray_get_timing sums
 * over the batches launched since the previous call (at most BHRAY_TIMING_RING of them).     */
#define BHRAY_TIMING_RING 128
typedef struct bhray_timing {
    uint32_t frames;                   /* frames aggregated                                  */
    uint32_t batches;                  /* batches aggregated (= frames unless frames_per_batch > 1) */
    float    total_ms;                 /* Σ (first launch → last launch) per batch           */
    float    trace_ms;                 /* Σ trace kernels                                    */
    float    classify_ms;              /* Σ grid classify kernels                            */
    uint32_t trace_launches;
    uint32_t classify_launches;
    float    level_trace_ms[BHRAY_MAX_LEVELS];
    float    level_classify_ms[BHRAY_MAX_LEVELS];
    float    sky_ms;                   /* Σ sky resolve kernels                              */
    uint32_t sky_launches;
    float    gather_ms;                /* multi-GPU, root: Σ (receive of the row tiles: start → all tiles arrived)   */
    float    deinterleave_ms;          /* multi-GPU, root: Σ de-interleave kernels                                   */
    uint32_t gathers;                  /* batches gathered                                                            */
    float    predicted_trace_ms;       /* BHRAY_F_TEMPORAL: Σ (prediction + the predicted trace launch, all levels): the bulk of such a
                                          frame; trace_ms / level_trace_ms then hold the fix-up launches only          */
    uint32_t predicted_launches;
    float    trace_exec_ms;            /* Σ EXECUTION spans of the trace kernels: first block's start → last block's end on the device's
                                          constant-rate clock, stamped by the kernel itself.  trace_ms (HIP events in the stream) also
                                          contains the time a launch waits for room beside the persistent kernels of the other frames
                                          in flight; this is what `rocprofv3 --kernel-trace` reports as the kernel's duration          */
    uint32_t trace_exec_launches;
} bhray_timing;
int bhray_get_timing(bhray_ctx* ctx, bhray_timing* out);       /* needs BHRAY_F_TIMING       */

/* ------------------------------------------------------------------------------------------
 * Host-side scene helpers (C++ behind this ABI; mirror the Rust host code on the path)
 * ---------------------------------------------------------------------------------------- */

/* CameraUniform::update — camera.rs:84-88.                                                   */
void bhray_camera_uniform_update(bhray_camera_uniform* u, const float position[3],
                                 const float forward[3], float fov);
/* BlackHoleUniform::update — blackhole.rs:68-98 (cgmath Euler→quaternion, rotate (0,-1,0),
 * right = z × up, forward = right × up).                                                     */
typedef struct bhray_black_hole {      /* scene::BlackHole, blackhole.rs:3-13                 */
    float   position[3];
    float   accretion_disk_rotation[3];
    float   accretion_disk_inner, accretion_disk_outer;
    float   rotation_speed;
    float   relativity_sphere_radius;
    int32_t show_disk_texture, show_red_shift;
    float   feather_amount;
} bhray_black_hole;
void bhray_black_hole_default(bhray_black_hole* bh);            /* blackhole.rs:16-28          */
void bhray_black_hole_uniform_update(bhray_black_hole_uniform* u, const bhray_black_hole* bh);
void bhray_details_default(bhray_details* d);                   /* mod.rs:116-121              */

/* Model + BVH builder — triangle.rs:65-259 (`Model::{new,add_*,build_bvh}`), growable storage
 * but the reference's capacity limit (524 288 per array) is enforced.                         */
typedef struct bhray_model bhray_model;
int  bhray_model_new(bhray_model** out);                        /* position (-10,0,30), visible 1 */
void bhray_model_free(bhray_model* m);
int  bhray_model_add_vertex(bhray_model* m, const float p[4]);
int  bhray_model_add_normal(bhray_model* m, const float n[4]);
int  bhray_model_add_triangle(bhray_model* m, const bhray_triangle* t);
int  bhray_model_build_bvh(bhray_model* m);                     /* triangle.rs:143-259         */
int  bhray_model_max_depth(const bhray_model* m);               /* deepest leaf, root = 1      */
/* Alternative builder behind a flag: binned surface-area-heuristic splits on centroid bounds, same
 * node / bvh_lookup format, leaves of <= 4 triangles.  NOT the reference's tree: traversal finds the same closest hit,
 * but rays that hit two triangles at exactly equal t (shared edges) may pick the other one, so frames can differ from the
 * reference-identical builder in isolated pixels.  Use for speed (shallower, tighter trees), never for parity runs.      */
int  bhray_model_build_bvh_sah(bhray_model* m);
int  bhray_model_desc_get(const bhray_model* m, bhray_model_desc* out); /* borrowed pointers   */
int  bhray_model_set_transform(bhray_model* m, const float position[3], int32_t visible);
/* Writes the exact ModelUniform image (BHRAY_MODEL_UNIFORM_BYTES) — triangle.rs:308-325.     */
int  bhray_model_pack_uniform(const bhray_model* m, void* dst, size_t size);
/* load_model — model.rs:7-87: OBJ (v / vn / f, triangles only) → scaled (0.5,-0.5,0.5),
 * flat-normal fallback, then build_bvh.                                                      */
int  bhray_load_model(const char* obj_path, bhray_model** out);

/* Disk-texture generator — the reference's offline asset tool perlin/src/main.rs:1-148 (hash-gradient Perlin noise at
 * densities 4/20/50/100, spiral warp (amount 2, power 0.5), pairwise 0.5 merges), whose shipped output is
 * src/renderer/textures/disk.png (binding 9).  Writes size x size RGBA8 with the value replicated into all four
 * channels, ready for bhray_set_texture(BHRAY_TEX_DISK).  size = 1000 reproduces disk.png up to libm rounding.     */
int bhray_generate_disk_texture(uint32_t size, uint8_t* rgba8_out);

#ifdef __cplusplus
}
#endif
#endif /* BHRAY_H */
