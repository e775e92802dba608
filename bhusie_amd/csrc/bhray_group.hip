// bhray_group.hip — the public bhray_ctx: one or several row partitions of the frame, and the gather.
//
// The reference renders on one device from one thread (src/app.rs:108-114, src/renderer/mod.rs:415-420: one compute pass,
// one submit).  This file keeps that host model and puts the multi-GPU row tiling BEHIND the C ABI:
//
//   bhray_ctx = N row partitions (frame row r -> partition (r / stripe_rows) % N), each rendered by a per-device engine
//   (bhray_dev, bhray_api.hip) — all N by this process (bhray_config.device_count = N), or one of them by this process as
//   rank row_rank of a communicator whose id the launcher distributes (one process per GPU).  Scene and uniforms are
//   replicated; every partition recomputes the coarse ladder rows its stripes depend on, so nothing is exchanged until the
//   frame's last level is done.  Then, per launched batch and on dedicated communication streams:
//       non-root partitions   ncclSend(packed rows of the batch's frames)            -> root rank      \  one RCCL group:
//       root partition        ncclRecv(... from every other partition) into staging                    /  7 links in parallel
//                             deinterleave_kernel: staging rows -> their frame rows (the root's own rows are written
//                             straight into the frame by its trace kernel)
//   so bhray_render stays asynchronous and the gather of batch k overlaps the render of batch k+1.
//
// RCCL is used directly (no PyTorch): loaded with dlopen on first use, because librccl.so is ~570 MB and a single-GPU host
// (the reference's normal case) must not pay for it.  Partitions that live on the same device (functional tests on a
// one-GPU box: devices = {0,0,...}) share one RCCL rank and exchange their tiles as send/recv-to-self inside the same group
// (RCCL refuses a communicator with a duplicated device; self send/recv inside a group is supported and matches in order).
#include <dlfcn.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bhray_dev.h"
#include "bhray_internal.h"

using namespace bhray;

// ------------------------------------------------------------------------------------------
// RCCL, bound at run time: types and prototypes from <rccl/rccl.h>, entry points through dlsym (nothing links librccl)
// ------------------------------------------------------------------------------------------
namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;       // (optional: only a failed ctx is torn down with it)
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    int version = 0;
    std::string error;
};

Rccl g_rccl;
std::mutex g_rccl_mutex;

// nullptr + g_rccl.error when RCCL cannot be loaded
Rccl* rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.so) return &g_rccl;
    void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!so) { g_rccl.error = std::string("dlopen(librccl.so.1): ") + dlerror(); return nullptr; }
    Rccl r; r.so = so;
#define SYM(field, name)                                                                       \
    do {                                                                                       \
        *(void**)(&r.field) = dlsym(so, name);                                                 \
        if (!r.field) { g_rccl.error = std::string("librccl: missing symbol ") + name; dlclose(so); return nullptr; } \
    } while (0)
    SYM(GetVersion, "ncclGetVersion"); SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitAll, "ncclCommInitAll");
    SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void**)(&r.CommAbort) = dlsym(so, "ncclCommAbort");
    if (r.GetVersion(&r.version) != ncclSuccess) r.version = 0;
    g_rccl = r;
    return &g_rccl;
}

// ------------------------------------------------------------------------------------------
// row partition arithmetic (shared by the host tables, the ABI helpers and — through the tables — the kernel)
// ------------------------------------------------------------------------------------------
// frame row of packed row i of partition `part`: stripes of `stripe` rows are dealt round-robin
inline uint64_t part_row(uint32_t world, uint32_t stripe, uint32_t part, uint32_t i) {
    return ((uint64_t)(i / stripe) * world + part) * stripe + (i % stripe);
}
inline uint32_t part_rows(uint32_t frame_h, uint32_t world, uint32_t stripe, uint32_t part) {
    const uint64_t cycle = (uint64_t)world * stripe;              // rows of one round over all partitions
    const uint64_t full = frame_h / cycle, rem = frame_h % cycle;
    uint64_t n = full * stripe;
    const uint64_t lo = (uint64_t)part * stripe;                  // this partition's stripe inside the last, partial round
    if (rem > lo) n += (rem - lo < stripe) ? (rem - lo) : stripe;
    return (uint32_t)n;
}

// One staging row of the root: where it comes from and which frame row it is.
struct RowDesc { uint32_t src_row0; uint32_t part_rows; uint32_t frame_row; uint32_t pad; };
struct FramePtrs { float4* p[BHRAY_MAX_FRAMES_PER_BATCH]; };

// De-interleave: row j of the concatenated non-root tiles of frame k of the batch -> its row of frame k: unpack_kernel below for the RGBA32F
// frame (whose rows travel packed), deinterleave16_kernel for the RGBA16F tiles of BHRAY_F_GATHER_SKY (8 B read + 8 B written per pixel).
// One block per row, consecutive lanes on consecutive pixels.
struct SkyPtrs { uint2* p[BHRAY_MAX_FRAMES_PER_BATCH]; };
__global__ __launch_bounds__(256) void deinterleave16_kernel(const uint2* __restrict__ staging, const SkyPtrs frames,
                                                             const RowDesc* __restrict__ table, const int width) {
    const RowDesc t = table[blockIdx.x];
    const int k = blockIdx.y;
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    const u2v* __restrict__ src = (const u2v*)(staging + ((size_t)t.src_row0 + (size_t)k * t.part_rows) * (size_t)width);
    u2v* __restrict__ dst = (u2v*)(frames.p[k] + (size_t)t.frame_row * (size_t)width);
    if ((width & 1) == 0) {                    // an even row: every row starts on 16 bytes, two pixels per lane and access
        const u4v* __restrict__ s4 = (const u4v*)src; u4v* __restrict__ d4 = (u4v*)dst;
        for (int x = threadIdx.x; x < width / 2; x += 256) d4[x] = __builtin_nontemporal_load(s4 + x);
    } else {
        for (int x = threadIdx.x; x < width; x += 256) dst[x] = __builtin_nontemporal_load(src + x);
    }
}

// The descriptor handed to hipImportExternalMemory is a dup() of the caller's.  ROCm 7.2 does not say whether the runtime closes it
// (CUDA's rule is that an imported fd belongs to the driver).  So the library closes it when the import is released - but only if the
// number still refers to the same open file as when it was duplicated (same device and inode: a dma-buf has an inode of its own); if the
// runtime closed it and the number has been reused by something else, it is left alone.  Either way an import leaks no descriptor.
struct FdId { dev_t dev; ino_t ino; bool ok; };
inline FdId fd_identity(int fd) { struct stat st; FdId r = {0, 0, false}; if (fstat(fd, &st) == 0) { r.dev = st.st_dev; r.ino = st.st_ino; r.ok = true; } return r; }
inline void close_if_still_ours(int fd, const FdId& id) {
    if (fd < 0 || !id.ok) return;
    const FdId now = fd_identity(fd);
    if (now.ok && now.dev == id.dev && now.ino == id.ino) (void)close(fd);
}

// The wire format of the RGBA32F gather: a pixel's alpha is exactly 0 (an escape direction) or 1 (a colour) - ray.wgsl:589-594 and the grid's
// copy / interpolate arms write nothing else - so a row of W pixels travels as 3 W floats (x, y, z) followed by ceil(W / 32) words of alpha
// bits: 12.1 bytes per pixel instead of 16, the same frame bit for bit.  The links into the root are what bounds an 8-GPU 1080p run (every
// frame's tiles over 7 links: profiles/EXPERIMENTS.md R5.7), so a quarter fewer bytes is a quarter more frames.  pack_kernel runs on a
// partition's communication stream behind its render (HBM-bound, 28 bytes per pixel); unpack_kernel replaces the de-interleave on the root
// (reads 12.1, writes 16).  A pixel whose alpha is neither 0 nor 1 (none can exist) raises the ctx's kernel error flag instead of being mangled.
__host__ __device__ inline size_t packed_row_words(size_t W) { return 3 * W + (W + 31) / 32; }
// (one launch packs the rows of ALL the partitions a GPU sends - one partition per GPU on a real node; the seven of a one-GPU test box)
struct PackJobs { const float4* src[BHRAY_MAX_DEVICES]; uint32_t* dst[BHRAY_MAX_DEVICES]; int rows[BHRAY_MAX_DEVICES]; int row0[BHRAY_MAX_DEVICES + 1]; int n; };
__global__ __launch_bounds__(256) void pack_kernel(const PackJobs jobs, const int width, int* __restrict__ err_flag) {
    int j = 0;
    while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.row0[j + 1]) j++;          // which partition this row belongs to (<= 16 entries, scalar)
    const int r = (int)blockIdx.x - jobs.row0[j], rows = jobs.rows[j];
    const size_t wpr = packed_row_words((size_t)width);
    const float4* __restrict__ s = jobs.src[j] + ((size_t)blockIdx.y * (size_t)rows + (size_t)r) * (size_t)width;      // frame k of the batch, row r
    uint32_t* __restrict__ d = jobs.dst[j] + ((size_t)blockIdx.y * (size_t)rows + (size_t)r) * wpr;
    float* __restrict__ xyz = reinterpret_cast<float*>(d);
    uint32_t* __restrict__ mask = d + 3 * (size_t)width;
    bool bad = false;
    for (int x0 = 0; x0 < width; x0 += 256) {
        const int x = x0 + (int)threadIdx.x;
        float4 p = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (x < width) {
            p = s[x];
            xyz[3 * (size_t)x] = p.x; xyz[3 * (size_t)x + 1] = p.y; xyz[3 * (size_t)x + 2] = p.z;
            const uint32_t a = __float_as_uint(p.w);
            bad = bad || (a != 0u && a != 0x3f800000u);
        }
        const unsigned long long m = __ballot(x < width && p.w != 0.0f);
        const int lane = (int)(threadIdx.x & 63), w0 = (x0 + (int)(threadIdx.x & ~63u)) >> 5;     // first mask word of this wave's 64 pixels
        if (lane == 0 && (w0 << 5) < width) mask[w0] = (uint32_t)m;
        if (lane == 0 && ((w0 + 1) << 5) < width) mask[w0 + 1] = (uint32_t)(m >> 32);
    }
    if (bad) *err_flag = BHRAY_E_STATE;
}
__global__ __launch_bounds__(256) void unpack_kernel(const uint32_t* __restrict__ staging, const FramePtrs frames, const RowDesc* __restrict__ table, const int width) {
    const RowDesc t = table[blockIdx.x];
    const int k = blockIdx.y;
    const size_t wpr = packed_row_words((size_t)width);
    const uint32_t* __restrict__ s = staging + ((size_t)t.src_row0 + (size_t)k * t.part_rows) * wpr;
    const float* __restrict__ xyz = reinterpret_cast<const float*>(s);
    const uint32_t* __restrict__ mask = s + 3 * (size_t)width;
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v* __restrict__ dst = (f4v*)(frames.p[k] + (size_t)t.frame_row * (size_t)width);
    for (int x = threadIdx.x; x < width; x += 256) {
        const uint32_t bit = (mask[x >> 5] >> (x & 31)) & 1u;
        f4v o = {xyz[3 * (size_t)x], xyz[3 * (size_t)x + 1], xyz[3 * (size_t)x + 2], bit ? 1.0f : 0.0f};
        dst[x] = o;
    }
}

struct Part {                      // one row partition of the frame
    bhray_dev* dev = nullptr;      // non-null: rendered by this ctx
    int device = -1;
    int rank = -1;                 // rank of the communicator this partition's tiles leave from / arrive at
    uint32_t rows = 0;
    std::vector<uint32_t> row_list; // its frame rows, increasing (bhray_config.partition)
    size_t stage_row0 = 0;         // root staging: first row of this partition's block (units of rows, for batch index 0)
};

struct CommRank {                  // a local rank of the communicator: one GPU of this process
    int device = -1;
    int rank = -1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;  // communication stream of that GPU (sends / receives / de-interleave)
};

struct WaitEvent { hipEvent_t ev = nullptr; int device = -1; bool in_use = false; };

#define BHRAY_WATCH_SCOPES 24         // concurrently armed scopes: the caller's thread + one per issue thread (BHRAY_MAX_DEVICES) + slack

struct GroupSlot {                 // per batch slot (same index as the devices' slots)
    float4* frames = nullptr;                 // root: B assembled frames
    std::vector<float4*> send;                // per partition: packed rows of the batch's frames (local non-root partitions)
    std::vector<uint2*> send16;               // BHRAY_F_GATHER_SKY: the same rows after the partition's own sky pass (what is sent)
    uint2* staging16 = nullptr;               // BHRAY_F_GATHER_SKY, root: RGBA16F tiles of the other partitions (layout of `staging`)
    std::vector<uint32_t*> sendp;             // RGBA32F gather: the packed rows that travel (pack_kernel: send[q] -> sendp[q])
    uint32_t* stagingp = nullptr;             // ... and where the root receives them, [part][frame of batch][row][packed row] (unpack_kernel -> the frames)
    std::vector<hipEvent_t> sent;             // per partition: recorded behind its send
    hipEvent_t frame_done = nullptr;          // root: recorded behind the de-interleave
    float4* dst[BHRAY_MAX_FRAMES_PER_BATCH];  // root: destination of each frame of the batch staged here (own or caller-bound)
    uint2* sky[BHRAY_MAX_FRAMES_PER_BATCH];   // root: RGBA16F images of bhray_resolve_sky (allocated on first use)
    uint64_t frame_no[BHRAY_MAX_FRAMES_PER_BATCH] = {};       // serial of the frame staged in each place (1-based), and of the frame its sky
    uint64_t sky_frame_no[BHRAY_MAX_FRAMES_PER_BATCH] = {};   // image was resolved from: a sky image is current only while the two agree
    hipEvent_t tev[3] = {nullptr, nullptr, nullptr};   // timing: before receive, after receive, after de-interleave
    bool timed = false;
};

// ------------------------------------------------------------------------------------------
// Issue threads (one process, N GPUs).  The reference host is ONE thread (src/renderer/mod.rs:415-420); driving 8 engines from it
// costs 80 us of host time per 1080p frame (8 x ~9 launches + events per batch, one RCCL group: profiles/r05_host_issue_n8.json)
// against the 50 us a GPU needs for its eighth of the frame - the host would be the bottleneck.  So bhray_render of a multi-device
// ctx only RECORDS the frame (uniform bytes, staging position, output binding) and hands it to one issue thread per GPU; each
// thread stages the frame on its GPU's engines, enqueues the launches of a full batch and its own share of the gather (its sends;
// on the root GPU the receives and the de-interleave) - the one-thread-per-device form of RCCL.  Every other call of the ABI first
// waits until the threads are idle and then runs on the caller's thread as before, so the engines are only ever touched by one
// thread at a time.  An error on an issue thread is reported by the next call of the ABI (and the ctx is failed).
// BHRAY_ISSUE_THREADS=0 in the environment: the one-thread path of rounds 2-4.
// ------------------------------------------------------------------------------------------
struct RenderCmd {
    uint8_t cam[32], bh[132], det[32];
    float mpos[BHRAY_MAX_MODELS][3]; int32_t mvis[BHRAY_MAX_MODELS]; bool mset[BHRAY_MAX_MODELS];
    int si; uint32_t sub;                  // where the caller's mirror of the staging position says this frame goes
    float4* bound;                         // bhray_bind_output (root only)
    uint64_t serial;                       // 1-based frame number
};
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv, idle_cv;
    std::deque<RenderCmd> q;
    bool busy = false, stop = false;
    int rc = 0; std::string err;           // first error of this thread
    size_t rank_index = 0;                 // c->ranks[rank_index]
};

}  // namespace

struct bhray_ctx {
    bhray_config cfg{};
    std::vector<Part> parts;
    std::vector<CommRank> ranks;           // local ranks
    // issue threads (see above): one per local rank of a multi-device ctx
    bool threaded = false;
    std::vector<std::unique_ptr<Worker>> workers;
    std::atomic<bool> async_failed{false};
    int async_rc = 0;
    // the caller's mirror of the engines' staging position (the engines themselves are only touched by their issue thread)
    bool mirror_stale = true;
    uint64_t st_counter = 0; uint32_t st_pending = 0; int st_method = 0; bool st_models = false;
    uint8_t u_cam[32] = {0}, u_bh[132] = {0}, u_det[32] = {0}; bool have_uniforms = false;
    float m_pos[BHRAY_MAX_MODELS][3] = {{0}}; int32_t m_vis[BHRAY_MAX_MODELS] = {0}; bool m_dirty[BHRAY_MAX_MODELS] = {false}, m_usable[BHRAY_MAX_MODELS] = {false};
    bool single = true;                    // one partition, no gather: every call goes straight to parts[0].dev
    bool gather = false;
    bool gather_sky = false;               // BHRAY_F_GATHER_SKY: the sky image is what travels
    uint32_t root = 0;
    bool root_local = false;
    uint32_t world = 1;                    // partitions
    uint32_t comm_size = 0;
    uint32_t B = 1, nslots = 1;
    std::vector<GroupSlot> gslots;
    RowDesc* d_table = nullptr; uint32_t table_rows = 0;      // root GPU
    size_t staging_rows = 0;               // rows of one staging buffer (all non-root partitions, B frames)
    size_t staging_alloc = 0, table_alloc = 0;                 // rows / entries allocated (grown by bhray_set_partition)
    std::vector<size_t> send_alloc;        // per partition: rows (x B frames) its send buffers hold
    // bhray_rebalance: what the frame rows cost, as far as the partitions' measured times tell (one weight per frame row; the bounds that
    // equalise their sums are the next partition), and the root's extra work per frame (gather + de-interleave) in the same unit
    std::vector<double> row_weight;
    double root_extra_ms = 0.0;
    uint32_t rebalances = 0, repartitions = 0;
    double hole_row = 0.0, hole_row_prev = 0.0; bool hole_row_valid = false, hole_row_prev_valid = false;   // frame row the hole projects to: at the last render / at the last rebalance
    float* d_xchg = nullptr;               // one process per GPU: 2 + 2 * partitions floats on this rank's GPU (the all-gather of bhray_rebalance)
    float4* bound = nullptr;               // bhray_bind_output: destination of the next frame (one-shot)
    hipEvent_t read_ev[64] = {nullptr};
    uint64_t read_tickets = 0;
    std::vector<void*> external;           // bhray_import_external_fd: hipExternalMemory_t handles, by mapped pointer (pairs: ptr, handle)
    std::vector<int> external_fd;          // ... and the duplicated descriptor each import handed to the runtime, with what it referred to then
    std::vector<FdId> external_fd_id;
    std::vector<WaitEvent> wait_pool;      // bhray_wait_stream: one event per call until the next render has consumed them
    int last_slot = 0; uint32_t last_sub = 0;
    uint64_t frames_staged = 0;
    bool rendered = false;
    std::atomic<bool> failed{false};       // a collective failed half way (on whichever thread): the communicator's state is unknown, every later gather is refused
    // communication watchdog (see "Watchdog" below): armed scopes, the thread, what it did
    std::atomic<int64_t> watch_deadline[BHRAY_WATCH_SCOPES];   // steady-clock ns by which an armed scope must have ended; 0 = free
    const char* watch_what[BHRAY_WATCH_SCOPES] = {nullptr};
    std::thread watch_thread;
    std::mutex watch_m; std::condition_variable watch_cv; bool watch_stop = false;
    int64_t watch_timeout_ms = 0;          // BHRAY_COMM_TIMEOUT_MS (default 30 000; 0 = no watchdog)
    std::atomic<bool> watch_fired{false};
    std::atomic<bool> comms_aborted{false}; // ncclCommAbort has been called on this ctx's communicators (by the watchdog, by drain / stop_workers of a failed ctx): never again, and no ncclCommDestroy
    std::string watch_msg;                  // written once, before watch_fired is set
    int* h_abort = nullptr;                 // pinned, device-visible: set to 1 when the watchdog fires (the test stall of BHRAY_TEST_FAULT leaves on it)
    float gather_ms = 0, deint_ms = 0; uint32_t gathers = 0;
    std::string err;
};

namespace {

// bhray_ctx::err belongs to the caller's thread; an issue thread collects its message in its own string (worker_main)
thread_local std::string* tl_err = nullptr;
int gfail(bhray_ctx* c, int code, const char* fmt, ...) {
    char buf[640];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    std::string msg = buf;
    if (c && c->watch_fired.load(std::memory_order_acquire)) { msg = c->watch_msg + " [then: " + msg + "]"; code = BHRAY_E_COMM; }   // what the aborted call reports is a consequence
    if (tl_err) *tl_err = msg; else if (c) c->err = msg; else dev_set_create_error(msg.c_str());
    return code;
}
// error of a per-device call: carry the device's message
int dfail(bhray_ctx* c, const bhray_dev* d, int rc) { if (rc) { if (tl_err) *tl_err = dev_last_error(d); else c->err = dev_last_error(d); } return rc; }

#define GHIP(c, call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return gfail(c, BHRAY_E_HIP, "%s: %s", #call, hipGetErrorString(e_));    \
    } while (0)
#define GNCCL(c, R, call)                                                                              \
    do {                                                                                               \
        ncclResult_t r_ = (call);                                                                      \
        if (r_ != ncclSuccess) return gfail(c, BHRAY_E_COMM, "%s: %s", #call, (R)->GetErrorString(r_));          \
    } while (0)
#define DEV(c, d, call) do { int rc_ = (call); if (rc_) return dfail(c, d, rc_); } while (0)

// ------------------------------------------------------------------------------------------
// Watchdog.  The gather has only ever run between partitions of ONE device; the first time it meets real links it may
// meet a peer that never posts (a rank that failed, a mis-wired launcher), and then ncclGroupEnd (connection set-up on the first batch)
// or the communication stream (an ncclRecv kernel that spins) waits for ever - and so does every bhray_sync, the bench, and the
// driver's clock.  So: every call of the ABI on a ctx that gathers, and every frame an issue thread enqueues, is an ARMED SCOPE with a
// deadline of BHRAY_COMM_TIMEOUT_MS (environment, default 30 000; 0 = no watchdog).  A thread of the ctx looks at the armed scopes ten
// times a second; when one is overdue it writes the message, marks the ctx failed, raises the device-visible abort word and calls
// ncclCommAbort on the ctx's communicators - which makes RCCL's blocked host calls return an error and its kernels leave their spin
// loops.  The overdue call then returns BHRAY_E_COMM with the watchdog's message (a host call that RCCL failed, or the check behind the
// wait it was blocked in); every later call of the ctx does too; bhray_destroy tears down without ncclCommDestroy.
// What it cannot reach: ncclCommInitRank / ncclCommInitAll inside bhray_create (no communicator to abort yet).
// ------------------------------------------------------------------------------------------
int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void abort_comms(bhray_ctx* c) {
    if (c->comms_aborted.exchange(true)) return;
    Rccl* R = g_rccl.so ? &g_rccl : nullptr;
    if (!R || !R->CommAbort) return;
    for (CommRank& r : c->ranks) if (r.comm) (void)R->CommAbort(r.comm);      // (the handles stay: a thread inside a call still holds them; nothing destroys them later)
}

void watch_main(bhray_ctx* c) {
    std::unique_lock<std::mutex> lk(c->watch_m);
    while (!c->watch_stop) {
        c->watch_cv.wait_for(lk, std::chrono::milliseconds(100));
        if (c->watch_stop || c->watch_fired.load()) continue;
        const int64_t t = now_ns();
        for (int i = 0; i < BHRAY_WATCH_SCOPES; i++) {
            const int64_t d = c->watch_deadline[i].load(std::memory_order_acquire);
            if (d == 0 || t < d) continue;
            char buf[512];
            snprintf(buf, sizeof buf, "watchdog: %s did not finish within BHRAY_COMM_TIMEOUT_MS = %lld ms - a peer of the RCCL gather never posted its share, or the "
                     "links are down; the communicator has been aborted (ncclCommAbort), the ctx is failed: destroy it", c->watch_what[i] ? c->watch_what[i] : "a call", (long long)c->watch_timeout_ms);
            c->watch_msg = buf;
            c->failed = true;
            c->watch_fired.store(true, std::memory_order_release);
            fprintf(stderr, "bhray: %s\n", buf); fflush(stderr);
            if (c->h_abort) __atomic_store_n(c->h_abort, 1, __ATOMIC_RELEASE);
            abort_comms(c);
            break;
        }
    }
}

struct WatchScope {                // arms a scope for the lifetime of the object (a ctx without a watchdog: nothing)
    bhray_ctx* c; int slot = -1;
    WatchScope(bhray_ctx* c_, const char* what) : c(c_) {
        if (!c || c->watch_timeout_ms <= 0 || !c->watch_thread.joinable()) return;
        const int64_t d = now_ns() + c->watch_timeout_ms * 1000000ll;
        for (int i = 0; i < BHRAY_WATCH_SCOPES; i++) {
            int64_t zero = 0;
            if (c->watch_deadline[i].load(std::memory_order_relaxed) == 0) {
                c->watch_what[i] = what;               // (written before the deadline is published; read only behind it)
                if (c->watch_deadline[i].compare_exchange_strong(zero, d, std::memory_order_acq_rel)) { slot = i; return; }
            }
        }
    }
    ~WatchScope() { if (slot >= 0) c->watch_deadline[slot].store(0, std::memory_order_release); }
    WatchScope(const WatchScope&) = delete; WatchScope& operator=(const WatchScope&) = delete;
};
// behind a wait that the watchdog may have cut short: the frames are not what they should be, say so
#define WATCH_CHECK(c) do { if ((c)->watch_fired.load(std::memory_order_acquire)) return gfail(c, BHRAY_E_COMM, "the wait ended because the communicator was aborted"); } while (0)

void watch_start(bhray_ctx* c) {
    const char* e = getenv("BHRAY_COMM_TIMEOUT_MS");
    c->watch_timeout_ms = e ? atoll(e) : 30000;
    if (c->watch_timeout_ms <= 0) return;
    if (hipHostMalloc((void**)&c->h_abort, sizeof(int), hipHostMallocMapped) == hipSuccess && c->h_abort) *c->h_abort = 0; else c->h_abort = nullptr;
    c->watch_thread = std::thread(watch_main, c);
}
void watch_stop(bhray_ctx* c) {
    if (c->watch_thread.joinable()) {
        { std::lock_guard<std::mutex> lk(c->watch_m); c->watch_stop = true; }
        c->watch_cv.notify_all();
        c->watch_thread.join();
    }
    if (c->h_abort) { (void)hipHostFree(c->h_abort); c->h_abort = nullptr; }
}

// BHRAY_TEST_FAULT (tests only; tests/test_gpu_comm_watchdog.py): "stall_gather:<n>" - the n-th gather of the process enqueues, in front of its
// RCCL group, a kernel that spins on the communication stream until the watchdog raises the abort word (at most ~20 s: never a hung GPU) -
// what a receive that is never matched looks like from the host; "fail_render:<n>" - the n-th frame an issue thread enqueues fails.
__global__ void test_stall_kernel(const int* abort_word, long long limit_ticks) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0 && wall_clock64() - t0 < limit_ticks) __builtin_amdgcn_s_sleep(127);
}
std::atomic<int> g_fault_gathers{0}, g_fault_renders{0};
int test_fault(const char* kind) {            // 0: off, n >= 1: at the n-th event
    const char* e = getenv("BHRAY_TEST_FAULT");
    if (!e) return 0;
    const size_t k = strlen(kind);
    if (strncmp(e, kind, k) != 0 || e[k] != ':') return 0;
    return atoi(e + k + 1);
}

Part* root_part(bhray_ctx* c) { return &c->parts[c->root]; }
CommRank* rank_of(bhray_ctx* c, const Part& p) {
    for (CommRank& r : c->ranks) if (r.rank == p.rank) return &r;
    return nullptr;
}
size_t frame_pixels(const bhray_ctx* c) { return (size_t)c->cfg.frame_w * (size_t)c->cfg.frame_h; }

// Gather of one launched batch (nb frames staged in slot si): sends, receives, de-interleave; see the file header.
// only == nullptr: every local rank's share in ONE RCCL group (the caller's thread drives all GPUs: grouped, so that no call waits for
// a peer this thread has not called yet); only != nullptr: the share of that rank alone, in a group of its own (its issue thread).
int group_gather(bhray_ctx* c, int si, uint32_t nb, const CommRank* only = nullptr) {
    Rccl* R = rccl();
    if (!R) return gfail(c, BHRAY_E_COMM, "%s", g_rccl.error.c_str());
    if (c->failed) return gfail(c, BHRAY_E_COMM, "an earlier gather of this ctx failed inside its RCCL group; destroy the ctx");
    GroupSlot& G = c->gslots[(size_t)si];
    const size_t W = c->cfg.frame_w;
    const bool timing = (c->cfg.flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE)) != 0;
    auto mine = [&](const Part& p) { return p.dev && (!only || p.rank == only->rank); };
    // the communication stream of every local rank waits for the renders whose rows it moves
    for (Part& p : c->parts) {
        if (!mine(p)) continue;
        CommRank* cr = rank_of(c, p);
        GHIP(c, hipSetDevice(p.device));
        GHIP(c, hipStreamWaitEvent(cr->stream, dev_slot_done(p.dev, si), 0));
    }
    Part& rp = *root_part(c);
    CommRank* rr = (c->root_local && mine(rp)) ? rank_of(c, rp) : nullptr;
    if (c->gather_sky) {
        // every partition resolves the sky over its own rows (sky.wgsl is per pixel) behind its render, on its communication stream: the
        // other partitions into their 8-byte send buffers, the root from its rows of the RGBA32F frames straight into the sky images
        for (uint32_t q = 0; q < c->world; q++) {
            Part& p = c->parts[q];
            if (!mine(p) || p.rows == 0) continue;
            CommRank* cr = rank_of(c, p);
            if (q != c->root) {
                DEV(c, p.dev, dev_launch_sky(p.dev, G.send[q], G.send16[q], (size_t)nb * p.rows * W, cr->stream));
            } else {
                for (uint32_t k = 0; k < nb; k++) {
                    for (size_t i = 0; i < p.row_list.size();) {            // runs of consecutive frame rows (a slab: one; stripes: one per stripe)
                        size_t j = i + 1;
                        while (j < p.row_list.size() && p.row_list[j] == p.row_list[j - 1] + 1) j++;
                        const size_t off = (size_t)p.row_list[i] * W;
                        DEV(c, p.dev, dev_launch_sky(p.dev, G.dst[k] + off, G.sky[k] + off, (j - i) * W, cr->stream));
                        i = j;
                    }
                }
            }
        }
    }
    const size_t wpr = c->gather_sky ? 2 * W : packed_row_words(W);          // 32-bit words per row on the wire (RGBA16F: 8 bytes per pixel; RGBA32F: packed, 12.1)
    if (!c->gather_sky) {
        // the rows that travel, packed (x, y, z + alpha bits): behind the render, on the communication stream of the GPU that sends them -
        // one launch per GPU for all the partitions it sends
        for (CommRank& cr : c->ranks) {
            if (only && cr.rank != only->rank) continue;
            PackJobs jobs; memset(&jobs, 0, sizeof jobs);
            int* err_flag = nullptr;
            for (uint32_t q = 0; q < c->world; q++) {
                Part& p = c->parts[q];
                if (q == c->root || !p.dev || p.rank != cr.rank || p.rows == 0) continue;
                jobs.src[jobs.n] = G.send[q]; jobs.dst[jobs.n] = G.sendp[q]; jobs.rows[jobs.n] = (int)p.rows;
                jobs.row0[jobs.n + 1] = jobs.row0[jobs.n] + (int)p.rows;
                jobs.n++;
                err_flag = dev_err_flag(p.dev);
            }
            if (jobs.n == 0) continue;
            GHIP(c, hipSetDevice(cr.device));
            (void)hipGetLastError();
            hipLaunchKernelGGL(pack_kernel, dim3((unsigned)jobs.row0[jobs.n], nb), dim3(256), 0, cr.stream, jobs, (int)W, err_flag);
            GHIP(c, hipGetLastError());
        }
    }
    if (rr && timing) { GHIP(c, hipSetDevice(rr->device)); GHIP(c, hipEventRecord(G.tev[0], rr->stream)); }
    if (const int at = test_fault("stall_gather")) {
        if (rr && c->h_abort && ++g_fault_gathers == at) {
            GHIP(c, hipSetDevice(rr->device));
            hipLaunchKernelGGL(test_stall_kernel, dim3(1), dim3(1), 0, rr->stream, (const int*)c->h_abort, 20ll * 100000000ll);
        }
    }
    // ONE group: every tile of the batch.  Sends and receives are issued in partition order, so the messages between
    // a pair of ranks (several partitions may share a rank) match in order.
    GNCCL(c, R, R->GroupStart());
    {
        // an error between GroupStart and GroupEnd must not leave the thread's RCCL group open (later collectives of this or any
        // other ctx would be queued into it, peers of a multi-process run would hang): close it, then report
        int grc = BHRAY_OK;
        auto in_group = [&]() -> int {
            for (uint32_t q = 0; q < c->world; q++) {
                Part& p = c->parts[q];
                if (q == c->root || !mine(p) || p.rows == 0) continue;
                CommRank* cr = rank_of(c, p);
                GHIP(c, hipSetDevice(p.device));
                GNCCL(c, R, R->Send(c->gather_sky ? (const void*)G.send16[q] : (const void*)G.sendp[q], (size_t)nb * p.rows * wpr, ncclFloat32, rp.rank, cr->comm, cr->stream));
            }
            if (rr) {
                GHIP(c, hipSetDevice(rr->device));
                for (uint32_t q = 0; q < c->world; q++) {
                    const Part& p = c->parts[q];
                    if (q == c->root || p.rows == 0) continue;
                    void* into = c->gather_sky ? (void*)(G.staging16 + p.stage_row0 * W) : (void*)(G.stagingp + p.stage_row0 * wpr);
                    GNCCL(c, R, R->Recv(into, (size_t)nb * p.rows * wpr, ncclFloat32, p.rank, rr->comm, rr->stream));
                }
            }
            return BHRAY_OK;
        };
        grc = in_group();
        if (grc != BHRAY_OK) { (void)R->GroupEnd(); c->failed = true; return grc; }      // c->err keeps the first message
    }
    GNCCL(c, R, R->GroupEnd());
    // a slot's next batch may overwrite its send buffer only after the send has read it
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        if (q == c->root || !mine(p)) continue;
        CommRank* cr = rank_of(c, p);
        GHIP(c, hipSetDevice(p.device));
        GHIP(c, hipEventRecord(G.sent[q], cr->stream));
        GHIP(c, hipStreamWaitEvent(dev_slot_stream(p.dev, si), G.sent[q], 0));
    }
    if (rr) {
        GHIP(c, hipSetDevice(rr->device));
        if (timing) GHIP(c, hipEventRecord(G.tev[1], rr->stream));
        if (c->table_rows && c->gather_sky) {
            SkyPtrs sp; memset(&sp, 0, sizeof sp);
            for (uint32_t k = 0; k < nb; k++) sp.p[k] = G.sky[k];
            (void)hipGetLastError();
            hipLaunchKernelGGL(deinterleave16_kernel, dim3(c->table_rows, nb), dim3(256), 0, rr->stream, G.staging16, sp, c->d_table, (int)W);
            GHIP(c, hipGetLastError());
        } else if (c->table_rows) {
            FramePtrs fp; memset(&fp, 0, sizeof fp);
            for (uint32_t k = 0; k < nb; k++) fp.p[k] = G.dst[k];
            (void)hipGetLastError();          // a stale error of an unrelated earlier call must not be blamed on this launch
            hipLaunchKernelGGL(unpack_kernel, dim3(c->table_rows, nb), dim3(256), 0, rr->stream, G.stagingp, fp, c->d_table, (int)W);
            GHIP(c, hipGetLastError());
        }
        if (c->gather_sky) for (uint32_t k = 0; k < nb; k++) G.sky_frame_no[k] = G.frame_no[k];      // the sky image of every frame of the batch is current
        if (timing) { GHIP(c, hipEventRecord(G.tev[2], rr->stream)); G.timed = true; }
        GHIP(c, hipEventRecord(G.frame_done, rr->stream));
        // the root's next render into this slot writes its own rows into the frames the de-interleave is filling: keep them ordered
        GHIP(c, hipStreamWaitEvent(dev_slot_stream(rp.dev, si), G.frame_done, 0));
        c->gathers++;
    } else if (!c->root_local && !only) {
        c->gathers++;
    }
    return BHRAY_OK;
}

// after a call that may have launched a batch on every local partition (only != nullptr: on that rank's partitions): enqueue its gather
int after_launch(bhray_ctx* c, const CommRank* only = nullptr) {
    int slot = -1; uint32_t nb = 0; bool any = false, first = true;
    for (Part& p : c->parts) {
        if (!p.dev || (only && p.rank != only->rank)) continue;
        int s; uint32_t n;
        if (dev_take_launched(p.dev, &s, &n)) {
            if (!first && (!any || s != slot || n != nb)) return gfail(c, BHRAY_E_STATE, "internal: partitions out of step");
            slot = s; nb = n; any = true;
        } else if (any) {
            return gfail(c, BHRAY_E_STATE, "internal: partitions out of step");
        }
        first = false;
    }
    if (!any) return BHRAY_OK;
    // timing of an earlier gather held by this slot is folded in before its events are re-recorded
    GroupSlot& G = c->gslots[(size_t)slot];
    const bool root_here = c->root_local && (!only || root_part(c)->rank == only->rank);
    if (G.timed && root_here) {
        CommRank* rr = rank_of(c, *root_part(c));
        GHIP(c, hipSetDevice(rr->device));
        GHIP(c, hipEventSynchronize(G.tev[2]));
        WATCH_CHECK(c);
        float a = 0, b = 0;
        GHIP(c, hipEventElapsedTime(&a, G.tev[0], G.tev[1])); GHIP(c, hipEventElapsedTime(&b, G.tev[1], G.tev[2]));
        c->gather_ms += a; c->deint_ms += b; G.timed = false;
    }
    return group_gather(c, slot, nb, only);
}

int group_flush(bhray_ctx* c) {
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_flush(p.dev));
    return after_launch(c);
}

int group_sync(bhray_ctx* c) {
    { int rc = group_flush(c); if (rc) return rc; }
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_sync(p.dev));
    for (CommRank& r : c->ranks) { GHIP(c, hipSetDevice(r.device)); GHIP(c, hipStreamSynchronize(r.stream)); }
    WATCH_CHECK(c);
    return BHRAY_OK;
}

void group_free(bhray_ctx* c) {
    Rccl* R = g_rccl.so ? &g_rccl : nullptr;
    // A ctx that failed in the middle of a batch (one rank's issue thread met an error, the others have posted their sends and receives)
    // may hold RCCL operations that will never be matched: they are aborted, or the synchronisations below would wait for them for ever.
    if (c->failed.load() || c->async_failed.load()) abort_comms(c);
    for (CommRank& r : c->ranks) if (r.stream) { (void)hipSetDevice(r.device); (void)hipStreamSynchronize(r.stream); }
    for (Part& p : c->parts) if (p.dev) { dev_destroy(p.dev); p.dev = nullptr; }
    for (uint32_t q = 0; q < c->parts.size(); q++) {
        const Part& p = c->parts[q];
        if (p.device < 0) continue;
        (void)hipSetDevice(p.device);
        for (GroupSlot& G : c->gslots) {
            if (q < G.send.size() && G.send[q]) (void)hipFree(G.send[q]);
            if (q < G.send16.size() && G.send16[q]) (void)hipFree(G.send16[q]);
            if (q < G.sendp.size() && G.sendp[q]) (void)hipFree(G.sendp[q]);
            if (q < G.sent.size() && G.sent[q]) (void)hipEventDestroy(G.sent[q]);
        }
    }
    if (c->root_local) {
        (void)hipSetDevice(c->parts[c->root].device);
        for (GroupSlot& G : c->gslots) {
            if (G.frames) (void)hipFree(G.frames);
            if (G.staging16) (void)hipFree(G.staging16);
            if (G.stagingp) (void)hipFree(G.stagingp);
            if (G.frame_done) (void)hipEventDestroy(G.frame_done);
            for (auto& e : G.tev) if (e) (void)hipEventDestroy(e);
            for (auto& s : G.sky) if (s) (void)hipFree(s);
        }
        if (c->d_table) (void)hipFree(c->d_table);
    }
    for (CommRank& r : c->ranks) {
        (void)hipSetDevice(r.device);
        if (r.comm && R && !c->comms_aborted.load()) (void)R->CommDestroy(r.comm);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    for (auto& e : c->read_ev) if (e) (void)hipEventDestroy(e);
    if (c->d_xchg && !c->ranks.empty()) { (void)hipSetDevice(c->ranks[0].device); (void)hipFree(c->d_xchg); }
    for (size_t i = 0; i + 1 < c->external.size(); i += 2) if (c->external[i + 1]) (void)hipDestroyExternalMemory((hipExternalMemory_t)c->external[i + 1]);
    for (size_t i = 0; i < c->external_fd.size(); i++) close_if_still_ours(c->external_fd[i], c->external_fd_id[i]);
    for (WaitEvent& w : c->wait_pool) if (w.ev) { (void)hipSetDevice(w.device); (void)hipEventDestroy(w.ev); }
}

// destination of the frame about to be staged at (slot, sub): every local partition's output binding (only != nullptr: that rank's
// partitions; `bound` = the caller's one-shot binding of the frame, root only)
int bind_partitions(bhray_ctx* c, int si, uint32_t sub, float4* bound, uint64_t serial, const CommRank* only = nullptr) {
    GroupSlot& G = c->gslots[(size_t)si];
    const size_t W = c->cfg.frame_w;
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        if (!p.dev || (only && p.rank != only->rank)) continue;
        if (q == c->root) {
            float4* dst = bound ? bound : G.frames + (size_t)sub * frame_pixels(c);
            G.dst[sub] = dst;
            G.frame_no[sub] = serial;
            DEV(c, p.dev, dev_bind_output(p.dev, dst, frame_pixels(c) * sizeof(float4)));
        } else if (p.rows) {
            DEV(c, p.dev, dev_bind_output(p.dev, G.send[q] + (size_t)sub * p.rows * W, (size_t)p.rows * W * sizeof(float4)));
        }
    }
    return BHRAY_OK;
}

// Where every partition's tiles land on the root, the de-interleave table, and the send / staging buffers - for the partition the
// ctx holds now.  Called by bhray_create and, with `headroom`, by bhray_set_partition (everything idle): buffers only ever grow.
int layout_gather(bhray_ctx* c, bool headroom) {
    const size_t W = c->cfg.frame_w;
    size_t row0 = 0;
    std::vector<RowDesc> table;
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        if (q == c->root) continue;
        p.stage_row0 = row0;
        for (uint32_t i = 0; i < p.rows; i++) {
            RowDesc d; d.src_row0 = (uint32_t)(row0 + i); d.part_rows = p.rows; d.frame_row = p.row_list[i]; d.pad = 0;
            table.push_back(d);
        }
        row0 += (size_t)c->B * p.rows;
    }
    c->staging_rows = row0;
    if (c->send_alloc.size() != c->world) c->send_alloc.assign(c->world, 0);
    auto grown = [&](size_t need, size_t cap) { size_t n = headroom ? need + need / 4 + 8 : need; return n > cap ? std::max(cap, need) : n; };
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        if (!p.dev || q == c->root || p.rows <= c->send_alloc[q]) continue;
        const size_t rows = grown(p.rows, c->cfg.frame_h);
        GHIP(c, hipSetDevice(p.device));
        for (GroupSlot& G : c->gslots) {
            if (G.send[q]) { GHIP(c, hipFree(G.send[q])); G.send[q] = nullptr; }
            GHIP(c, hipMalloc(&G.send[q], (size_t)c->B * rows * W * sizeof(float4)));
            GHIP(c, hipMemset(G.send[q], 0xFF, (size_t)c->B * rows * W * sizeof(float4)));
            if (c->gather_sky) {
                if (G.send16[q]) { GHIP(c, hipFree(G.send16[q])); G.send16[q] = nullptr; }
                GHIP(c, hipMalloc(&G.send16[q], (size_t)c->B * rows * W * sizeof(uint2)));
            } else {
                if (G.sendp[q]) { GHIP(c, hipFree(G.sendp[q])); G.sendp[q] = nullptr; }
                GHIP(c, hipMalloc(&G.sendp[q], (size_t)c->B * rows * packed_row_words(W) * sizeof(uint32_t)));
            }
        }
        c->send_alloc[q] = rows;
    }
    if (c->root_local) {
        GHIP(c, hipSetDevice(c->parts[c->root].device));
        if (c->staging_rows > c->staging_alloc) {
            const size_t rows = grown(c->staging_rows, (size_t)c->B * c->cfg.frame_h);
            for (GroupSlot& G : c->gslots) {
                if (!c->gather_sky) { if (G.stagingp) { GHIP(c, hipFree(G.stagingp)); G.stagingp = nullptr; } GHIP(c, hipMalloc(&G.stagingp, rows * packed_row_words(W) * sizeof(uint32_t))); }
                else { if (G.staging16) { GHIP(c, hipFree(G.staging16)); G.staging16 = nullptr; } GHIP(c, hipMalloc(&G.staging16, rows * W * sizeof(uint2))); }
            }
            c->staging_alloc = rows;
        }
        c->table_rows = (uint32_t)table.size();
        if (table.size() > c->table_alloc) {
            if (c->d_table) { GHIP(c, hipFree(c->d_table)); c->d_table = nullptr; }
            GHIP(c, hipMalloc(&c->d_table, (size_t)c->cfg.frame_h * sizeof(RowDesc)));      // never more rows than the frame has
            c->table_alloc = c->cfg.frame_h;
        }
        if (!table.empty()) GHIP(c, hipMemcpy(c->d_table, table.data(), table.size() * sizeof(RowDesc), hipMemcpyHostToDevice));
    }
    return BHRAY_OK;
}

// ---- issue threads ---------------------------------------------------------------------------
// One recorded frame on one rank's engines: uniforms, the staging position (checked against the caller's mirror), the gather of a
// batch that a change of kernel variant launched, the output bindings, the render, the gather of the batch it completed.
int worker_render(bhray_ctx* c, const CommRank* cr, const RenderCmd& cmd) {
    for (Part& p : c->parts) {
        if (!p.dev || p.rank != cr->rank) continue;
        DEV(c, p.dev, dev_set_uniforms(p.dev, cmd.cam, cmd.bh, cmd.det));
        for (uint32_t mi = 0; mi < BHRAY_MAX_MODELS; mi++) if (cmd.mset[mi]) DEV(c, p.dev, dev_set_model_transform(p.dev, mi, cmd.mpos[mi], cmd.mvis[mi]));
        int s; uint32_t k;
        DEV(c, p.dev, dev_next_position(p.dev, &s, &k));
        if (s != cmd.si || k != cmd.sub) return gfail(c, BHRAY_E_STATE, "internal: issue thread out of step with the caller (slot %d/%d, position %u/%u)", s, cmd.si, k, cmd.sub);
    }
    { int rc = after_launch(c, cr); if (rc) return rc; }
    { int rc = bind_partitions(c, cmd.si, cmd.sub, cmd.bound, cmd.serial, cr); if (rc) return rc; }
    for (Part& p : c->parts) if (p.dev && p.rank == cr->rank) DEV(c, p.dev, dev_render(p.dev));
    return after_launch(c, cr);
}

void worker_main(bhray_ctx* c, Worker* w) {
    const CommRank* cr = &c->ranks[w->rank_index];
    (void)hipSetDevice(cr->device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv.wait(lk, [&] { return w->stop || !w->q.empty(); });
        if (w->q.empty()) { if (w->stop) return; continue; }
        const RenderCmd cmd = w->q.front();
        w->q.pop_front();
        w->busy = true;
        lk.unlock();
        int rc = BHRAY_OK; std::string msg;
        tl_err = &msg;                 // gfail / dfail of this thread write here
        if (!c->async_failed.load(std::memory_order_acquire)) {
            WatchScope ws(c, "a frame's launches and gather on an issue thread");
            const int at = test_fault("fail_render");
            if (at && ++g_fault_renders == at) rc = gfail(c, BHRAY_E_STATE, "BHRAY_TEST_FAULT: frame %d fails on the issue thread of GPU %d", at, cr->device);
            else rc = worker_render(c, cr, cmd);
        }
        lk.lock();
        if (rc != BHRAY_OK && w->rc == BHRAY_OK) { w->rc = rc; w->err = msg; c->async_failed.store(true, std::memory_order_release); }
        w->busy = false;
        w->idle_cv.notify_all();
        w->cv.notify_all();          // (a caller waiting for room in the queue)
    }
}

// wait until every issue thread is idle; then report the first error any of them met (sticky: the ctx is failed)
int drain(bhray_ctx* c) {
    if (!c->threaded) return BHRAY_OK;
    for (auto& w : c->workers) {
        std::unique_lock<std::mutex> lk(w->m);
        // One thread failed before its sends: the others' ncclGroupEnd (connection set-up on the first batch) would wait for that peer for
        // ever, and this wait with them.  A failed ctx's communicators are aborted as soon as the failure is seen - here, not in
        // bhray_destroy behind the join.
        while (!w->idle_cv.wait_for(lk, std::chrono::milliseconds(50), [&] { return w->q.empty() && !w->busy; }))
            if (c->async_failed.load(std::memory_order_acquire) || c->failed.load()) { lk.unlock(); abort_comms(c); lk.lock(); }
    }
    c->mirror_stale = true;          // whatever the caller does next on its own thread may launch staged frames
    if (c->async_failed.load(std::memory_order_acquire)) {
        for (auto& w : c->workers) {
            std::lock_guard<std::mutex> lk(w->m);
            if (w->rc != BHRAY_OK) { c->err = "issue thread of GPU " + std::to_string(c->ranks[w->rank_index].device) + ": " + w->err; c->async_rc = w->rc; break; }
        }
        c->failed = true;
        return c->async_rc ? c->async_rc : BHRAY_E_STATE;
    }
    return BHRAY_OK;
}
// every call of the ABI on a ctx that gathers: an armed scope (Watchdog above) for as long as the call lasts; a ctx the watchdog failed answers with its message
#define ENTER(c) WatchScope watch_scope_((c)->gather ? (c) : nullptr, __func__);                                        \
    do { if ((c)->watch_fired.load(std::memory_order_acquire)) return gfail(c, BHRAY_E_COMM, "%s refused", __func__);  \
         if ((c)->threaded) { int rc_ = drain(c); if (rc_) return rc_; } } while (0)

void stop_workers(bhray_ctx* c) {
    if (c->async_failed.load() || c->failed.load()) abort_comms(c);     // (a thread blocked in RCCL behind a failed peer would never be joined)
    for (auto& w : c->workers) {
        { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    c->workers.clear();
    c->threaded = false;
}

// Frame row the hole projects to (create_ray, ray.wgsl:269-285, inverted; plain float arithmetic: a heuristic for bhray_rebalance, not pixels)
void track_hole_row(bhray_ctx* c, const void* cam32, const void* bh132) {
    bhray_camera_uniform cam; bhray_black_hole_uniform bh;
    memcpy(&cam, cam32, sizeof cam); memcpy(&bh, bh132, sizeof bh);
    const double f[3] = {cam.forward[0], cam.forward[1], cam.forward[2]};
    // right = normalize(forward x (0,-1,0)), up = normalize(forward x right)
    double r[3] = {f[1] * 0.0 - f[2] * -1.0, f[2] * 0.0 - f[0] * 0.0, f[0] * -1.0 - f[1] * 0.0};
    const double rl = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (!(rl > 0.0)) { c->hole_row_valid = false; return; }
    for (double& v : r) v /= rl;
    double u[3] = {f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0]};
    const double ul = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (!(ul > 0.0)) { c->hole_row_valid = false; return; }
    for (double& v : u) v /= ul;
    const double ff = 1.0 / tan(0.5 * (double)cam.fov);
    const double d[3] = {bh.position[0] - cam.position[0], bh.position[1] - cam.position[1], bh.position[2] - cam.position[2]};
    const double along = (d[0] * f[0] + d[1] * f[1] + d[2] * f[2]) / ((f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) * ff);   // d = along * (posx right + posy up + forward ff)
    if (!(along > 0.0)) { c->hole_row_valid = false; return; }
    const double posy = (d[0] * u[0] + d[1] * u[1] + d[2] * u[2]) / along;
    const uint32_t nl = c->cfg.levels;
    const double lw = c->cfg.level_w[nl - 1], lh = c->cfg.level_h[nl - 1];
    const double sm = std::min(lw - 1.0, lh - 1.0);
    c->hole_row = posy * sm * 0.5 + (lh - 1.0) * 0.5 - (double)c->cfg.crop_y;
    c->hole_row_valid = c->hole_row == c->hole_row;
}

// kernel variant of the frame the current uniforms describe (the rule of the engines: bhray_api.hip frame_variant)
void ctx_variant(const bhray_ctx* c, int& method, bool& models) {
    bhray_details d; memcpy(&d, c->u_det, sizeof d);
    method = d.integration_method != 0 ? 1 : 0;
    int mc = d.model_count; if (mc < 0) mc = 0; if (mc > BHRAY_MAX_MODELS) mc = BHRAY_MAX_MODELS;
    models = false;
    for (int i = 0; i < mc; i++) if (c->m_usable[i] && c->m_vis[i] != 0) models = true;
}

}  // namespace

extern "C" {

uint32_t bhray_partition_rows(uint32_t frame_h, uint32_t world, uint32_t stripe_rows, uint32_t part) {
    if (world < 1 || stripe_rows < 1 || part >= world) return 0;
    return part_rows(frame_h, world, stripe_rows, part);
}

int bhray_partition_row_index(uint32_t frame_h, uint32_t world, uint32_t stripe_rows, uint32_t part, uint32_t i, uint32_t* frame_row) {
    if (!frame_row || world < 1 || stripe_rows < 1 || part >= world) return BHRAY_E_INVALID;
    const uint64_t r = part_row(world, stripe_rows, part, i);
    if (r >= frame_h) return BHRAY_E_INVALID;
    *frame_row = (uint32_t)r;
    return BHRAY_OK;
}

static uint32_t config_world(const bhray_config* cfg) { return cfg->device_count >= 2 ? cfg->device_count : (cfg->row_world ? cfg->row_world : 1); }

uint32_t bhray_config_partition_rows(const bhray_config* cfg, uint32_t part) {
    if (!cfg) return 0;
    bhray_config one = *cfg;
    if (one.partition == BHRAY_PARTITION_STRIPES && one.stripe_rows == 0) one.stripe_rows = 27;
    const uint32_t world = config_world(cfg);
    if (part >= world || partition_error(one, world)) return 0;
    return (uint32_t)partition_row_list(one, world, part).size();
}

int bhray_config_partition_row_index(const bhray_config* cfg, uint32_t part, uint32_t i, uint32_t* frame_row) {
    if (!cfg || !frame_row) return BHRAY_E_INVALID;
    bhray_config one = *cfg;
    if (one.partition == BHRAY_PARTITION_STRIPES && one.stripe_rows == 0) one.stripe_rows = 27;
    const uint32_t world = config_world(cfg);
    if (part >= world || partition_error(one, world)) return BHRAY_E_INVALID;
    const std::vector<uint32_t> rows = partition_row_list(one, world, part);
    if (i >= rows.size()) return BHRAY_E_INVALID;
    *frame_row = rows[i];
    return BHRAY_OK;
}

// Slab bounds that minimise the largest partition's work.  The work of the slab [a, b) is, at every level, the work of the level rows
// its frame rows depend on - for consecutive frame rows a consecutive range of level rows, found with the ladder's own arithmetic
// (coarse_rows_needed on the two end rows) - so with prefix sums it is O(levels); the optimum over contiguous partitions is found
// by bisection on the load a slab may carry (greedy packing decides feasibility: the work of [a, b) is non-decreasing in b).
int bhray_balance_slabs(const bhray_config* cfg, const uint64_t* const* row_work, uint32_t world, uint32_t* slab_row0) {
    if (!cfg || !row_work || !slab_row0 || world < 1 || world > BHRAY_MAX_DEVICES) return BHRAY_E_INVALID;
    const uint32_t nl = cfg->levels;
    if (nl < 1 || nl > BHRAY_MAX_LEVELS || cfg->frame_h < 1 || cfg->crop_y + cfg->frame_h > cfg->level_h[nl - 1]) return BHRAY_E_INVALID;
    std::vector<std::vector<double>> pre(nl);                      // pre[l][y] = work of level rows [0, y)
    for (uint32_t l = 0; l < nl; l++) {
        if (!row_work[l] || cfg->level_h[l] < 2) return BHRAY_E_INVALID;
        pre[l].assign((size_t)cfg->level_h[l] + 1, 0.0);
        for (uint32_t y = 0; y < cfg->level_h[l]; y++) pre[l][y + 1] = pre[l][y] + (double)row_work[l][y];
    }
    auto work = [&](uint32_t a, uint32_t b) -> double {         // frame rows [a, b), b > a
        int lo = (int)(cfg->crop_y + a), hi = (int)(cfg->crop_y + b - 1);
        double w = 0.0;
        for (int l = (int)nl - 1; l >= 0; l--) {
            w += pre[(size_t)l][(size_t)hi + 1] - pre[(size_t)l][(size_t)lo];
            if (l > 0) {
                const std::vector<int32_t> ends = coarse_rows_needed({lo, hi}, (int)cfg->level_h[l], (int)cfg->level_h[l - 1]);
                if (ends.empty()) break;
                lo = ends.front(); hi = ends.back();
            }
        }
        return w;
    };
    const uint32_t H = cfg->frame_h;
    auto pack = [&](double cap, uint32_t* bounds) -> bool {      // greedy: every slab as long as its work stays within cap
        uint32_t a = 0;
        for (uint32_t p = 0; p < world; p++) {
            if (bounds) bounds[p] = a;
            if (a < H) {
                if (work(a, a + 1) > cap) return false;
                uint32_t lo = a + 1, hi = H;                      // largest b with work(a, b) <= cap
                while (lo < hi) { const uint32_t mid = lo + (hi - lo + 1) / 2; if (work(a, mid) <= cap) lo = mid; else hi = mid - 1; }
                a = lo;
            }
        }
        if (bounds) bounds[world] = H;
        return a >= H;
    };
    double lo = 0.0, hi = work(0, H);
    if (!(hi > 0.0)) {                                             // no work measured: equal rows
        for (uint32_t p = 0; p <= world; p++) slab_row0[p] = (uint32_t)((uint64_t)H * p / world);
        return BHRAY_OK;
    }
    for (int it = 0; it < 60 && hi - lo > 1e-9 * hi; it++) { const double mid = 0.5 * (lo + hi); if (pack(mid, nullptr)) hi = mid; else lo = mid; }
    if (!pack(hi, slab_row0)) return BHRAY_E_STATE;
    // the greedy packing front-loads: the last slabs may be short.  Spread the slack by a few rounds of moving each interior bound
    // to the position that evens out its two neighbours (never beyond the cap found above).
    for (int round = 0; round < 8; round++) {
        for (uint32_t p = 1; p < world; p++) {
            const uint32_t a = slab_row0[p - 1], b = slab_row0[p + 1];
            if (b <= a + 1) continue;
            uint32_t best = slab_row0[p]; double bestv = 1e300;
            uint32_t l2 = a + 1, h2 = b - 1;                       // work(a, m) rises, work(m, b) falls with m: bisect to the crossing
            if (h2 < l2) continue;
            while (l2 < h2) { const uint32_t mid = l2 + (h2 - l2) / 2; if (work(a, mid) < work(mid, b)) l2 = mid + 1; else h2 = mid; }
            for (uint32_t m = (l2 > a + 1 ? l2 - 1 : l2); m <= l2 && m < b; m++) {
                const double v = std::max(work(a, m), work(m, b));
                if (v < bestv) { bestv = v; best = m; }
            }
            slab_row0[p] = best;
        }
    }
    return BHRAY_OK;
}

int bhray_comm_unique_id(uint8_t id[BHRAY_COMM_ID_BYTES]) {
    if (!id) return BHRAY_E_INVALID;
    Rccl* R = rccl();
    if (!R) return gfail(nullptr, BHRAY_E_COMM, "%s", g_rccl.error.c_str());
    ncclUniqueId u;
    ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) return gfail(nullptr, BHRAY_E_COMM, "ncclGetUniqueId: %s", R->GetErrorString(r));
    static_assert(sizeof(ncclUniqueId) == BHRAY_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, u.internal, BHRAY_COMM_ID_BYTES);
    return BHRAY_OK;
}

const char* bhray_last_error(const bhray_ctx* c) { return c ? c->err.c_str() : dev_last_error(nullptr); }

void bhray_destroy(bhray_ctx* c) {
    if (!c) return;
    {
        WatchScope ws(c->gather ? c : nullptr, "bhray_destroy");     // (issue threads that finish what they hold may wait for a peer)
        stop_workers(c);             // (the issue threads finish what they hold first)
    }
    watch_stop(c);
    group_free(c);
    delete c;
}

int bhray_create(const bhray_config* cfg_in, bhray_ctx** out) {
    if (!cfg_in || !out) return gfail(nullptr, BHRAY_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg_in->struct_size != sizeof(bhray_config)) return gfail(nullptr, BHRAY_E_INVALID, "bhray_config.struct_size mismatch (%u, library expects %zu)", cfg_in->struct_size, sizeof(bhray_config));
    // More than 22 frame slots put the HIP runtime into an intermittent state in which a 20-frame burst takes 64-94 ms instead of 8.5
    // (measured with 24 and 28 slots, whatever GPU_MAX_HW_QUEUES says; 21-22 slots are the best short-burst setting, 20 the best
    // sustained one): a larger request (up to BHRAY_MAX_FRAMES_IN_FLIGHT) is served with 22.
    bhray_config cfg_norm = *cfg_in;
    if (cfg_norm.frames_in_flight > 22 && cfg_norm.frames_in_flight <= BHRAY_MAX_FRAMES_IN_FLIGHT) cfg_norm.frames_in_flight = 22;
    const bhray_config* cfg = &cfg_norm;
    if (cfg->device_count > BHRAY_MAX_DEVICES) return gfail(nullptr, BHRAY_E_INVALID, "device_count > %d", BHRAY_MAX_DEVICES);
    if (cfg->gather > BHRAY_GATHER_RCCL) return gfail(nullptr, BHRAY_E_INVALID, "unknown gather mode %u", cfg->gather);
    bhray_ctx* c = new (std::nothrow) bhray_ctx();
    if (!c) return gfail(nullptr, BHRAY_E_NOMEM, "host allocation failed");
    c->cfg = *cfg;
    const bool multi_dev = cfg->device_count >= 2;                                   // one process, N GPUs
    const bool multi_proc = !multi_dev && cfg->gather == BHRAY_GATHER_RCCL && cfg->row_world > 1;   // one process per GPU
    c->gather = multi_dev || multi_proc;
    c->single = !c->gather;
    c->gather_sky = c->gather && (cfg->flags & BHRAY_F_GATHER_SKY) != 0;
    if ((cfg->flags & BHRAY_F_GATHER_SKY) && !c->gather) { delete c; return gfail(nullptr, BHRAY_E_INVALID, "BHRAY_F_GATHER_SKY needs a multi-GPU ctx (device_count >= 2 or gather = BHRAY_GATHER_RCCL)"); }
#define FAIL(code, ...) do { int rc_ = gfail(nullptr, code, __VA_ARGS__); bhray_destroy(c); return rc_; } while (0)
    if (c->single) {
        bhray_config one = *cfg;
        if (cfg->device_count == 1) one.device = cfg->devices[0];
        c->parts.resize(1);
        int rc = dev_create(&one, DevOptions(), &c->parts[0].dev);
        if (rc) { bhray_destroy(c); return rc; }
        c->parts[0].device = one.device;
        c->cfg = one; c->cfg.device_count = cfg->device_count;
        c->world = one.row_world ? one.row_world : 1;
        *out = c;
        return BHRAY_OK;
    }
    // ---- partitions
    c->world = multi_dev ? cfg->device_count : cfg->row_world;
    if (multi_dev && cfg->row_world > 1 && cfg->row_world != cfg->device_count) FAIL(BHRAY_E_INVALID, "row_world must be 0, 1 or device_count when device_count >= 2");
    if (multi_proc && cfg->row_rank >= cfg->row_world) FAIL(BHRAY_E_INVALID, "bad row partition");
    if (cfg->gather_root >= c->world) FAIL(BHRAY_E_INVALID, "gather_root %u is not a partition (%u partitions)", cfg->gather_root, c->world);
    if (cfg->frame_w < 1 || cfg->frame_h < 1) FAIL(BHRAY_E_INVALID, "frame window outside the last level");
    c->root = cfg->gather_root;
    c->cfg.row_world = c->world;
    const uint32_t stripe = cfg->stripe_rows ? cfg->stripe_rows : 27;
    c->cfg.stripe_rows = stripe;
    if (const char* why = partition_error(c->cfg, c->world)) FAIL(BHRAY_E_INVALID, "bad row partition: %s", why);
    c->parts.resize(c->world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) FAIL(BHRAY_E_NO_DEVICE, "no HIP device visible (libbhray has no CPU path)");
    // ranks: distinct devices of this process in list order (one process per GPU: the launcher's rank)
    std::vector<int> rank_dev;
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        p.row_list = partition_row_list(c->cfg, c->world, q);
        p.rows = (uint32_t)p.row_list.size();
        if (multi_dev) {
            if (cfg->devices[q] < 0 || cfg->devices[q] >= ndev) FAIL(BHRAY_E_NO_DEVICE, "device %d not present (%d visible)", cfg->devices[q], ndev);
            p.device = cfg->devices[q];
            int r = -1;
            for (size_t k = 0; k < rank_dev.size(); k++) if (rank_dev[k] == p.device) r = (int)k;
            if (r < 0) { r = (int)rank_dev.size(); rank_dev.push_back(p.device); }
            p.rank = r;
        } else {
            p.rank = (int)q;
            if (q == cfg->row_rank) {
                const int d = cfg->device_count == 1 ? cfg->devices[0] : cfg->device;
                if (d < 0 || d >= ndev) FAIL(BHRAY_E_NO_DEVICE, "device %d not present (%d visible)", d, ndev);
                p.device = d;
            }
        }
    }
    c->comm_size = multi_dev ? (uint32_t)rank_dev.size() : c->world;
    c->root_local = multi_dev || cfg->row_rank == c->root;
    c->nslots = cfg->frames_in_flight ? cfg->frames_in_flight : 4;
    c->B = cfg->frames_per_batch ? cfg->frames_per_batch : 1;
    if (c->nslots > BHRAY_MAX_FRAMES_IN_FLIGHT) FAIL(BHRAY_E_INVALID, "frames_in_flight > %d", BHRAY_MAX_FRAMES_IN_FLIGHT);
    if (c->B > BHRAY_MAX_FRAMES_PER_BATCH) FAIL(BHRAY_E_INVALID, "frames_per_batch > %d", BHRAY_MAX_FRAMES_PER_BATCH);
    // ---- per-device engines
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        if (p.device < 0) continue;
        bhray_config one = *cfg;
        one.device = p.device; one.device_count = 0; one.gather = BHRAY_GATHER_NONE;
        one.row_rank = q; one.row_world = c->world; one.stripe_rows = stripe;
        DevOptions opt; opt.external_out = true; opt.frame_rowmap = (q == c->root);
        // Partitions that share a physical GPU (a one-GPU box standing in for N) share its hardware queues: beyond ~24 user queues the
        // device's scheduler serves them a few at a time, milliseconds apart (8 engines x 6 streams: a 20-frame block 44 ms, x 2 streams:
        // 14 ms = the sum of the rank probes; profiles/EXPERIMENTS.md R5.1), so their frame slots together get 22 streams.
        uint32_t same_dev = 0;
        for (uint32_t k = 0; k < c->world; k++) if (c->parts[k].device == p.device) same_dev++;
        if (same_dev > 1) opt.max_streams = std::max(1u, 22u / same_dev);
        int rc = dev_create(&one, opt, &p.dev);
        if (rc) { bhray_destroy(c); return rc; }
        if (dev_local_rows(p.dev) != p.rows) FAIL(BHRAY_E_STATE, "internal: partition %u has %u rows, expected %u", q, dev_local_rows(p.dev), p.rows);
    }
    // ---- communicator
    Rccl* R = rccl();
    if (!R) FAIL(BHRAY_E_COMM, "%s", g_rccl.error.c_str());
#define CH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) FAIL(BHRAY_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); } while (0)
#define CN(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) FAIL(BHRAY_E_COMM, "%s: %s", #call, R->GetErrorString(r_)); } while (0)
    if (multi_dev) {
        std::vector<ncclComm_t> comms(rank_dev.size(), nullptr);
        CN(R->CommInitAll(comms.data(), (int)rank_dev.size(), rank_dev.data()));
        for (size_t k = 0; k < rank_dev.size(); k++) { CommRank r; r.device = rank_dev[k]; r.rank = (int)k; r.comm = comms[k]; c->ranks.push_back(r); }
    } else {
        Part& me = c->parts[cfg->row_rank];
        CH(hipSetDevice(me.device));
        ncclUniqueId id; memcpy(id.internal, cfg->comm_id, BHRAY_COMM_ID_BYTES);
        CommRank r; r.device = me.device; r.rank = (int)cfg->row_rank;
        CN(R->CommInitRank(&r.comm, (int)c->world, id, (int)cfg->row_rank));
        c->ranks.push_back(r);
    }
    for (CommRank& r : c->ranks) { CH(hipSetDevice(r.device)); CH(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking)); }
    watch_start(c);
    // ---- buffers
    c->gslots.resize(c->nslots);
    for (GroupSlot& G : c->gslots) {
        memset(G.dst, 0, sizeof G.dst); memset(G.sky, 0, sizeof G.sky);
        G.send.assign(c->world, nullptr); G.sent.assign(c->world, nullptr); G.send16.assign(c->world, nullptr); G.sendp.assign(c->world, nullptr);
        for (uint32_t q = 0; q < c->world; q++) {
            Part& p = c->parts[q];
            if (!p.dev || q == c->root) continue;
            CH(hipSetDevice(p.device));
            CH(hipEventCreateWithFlags(&G.sent[q], hipEventDisableTiming));
        }
        if (c->root_local) {
            CH(hipSetDevice(c->parts[c->root].device));
            CH(hipMalloc(&G.frames, (size_t)c->B * frame_pixels(c) * sizeof(float4)));
            CH(hipMemset(G.frames, 0xFF, (size_t)c->B * frame_pixels(c) * sizeof(float4)));
            if (c->gather_sky)
                for (uint32_t k = 0; k < c->B; k++) { CH(hipMalloc(&G.sky[k], frame_pixels(c) * sizeof(uint2))); CH(hipMemset(G.sky[k], 0xFF, frame_pixels(c) * sizeof(uint2))); }
            CH(hipEventCreateWithFlags(&G.frame_done, hipEventDisableTiming));
            if (cfg->flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE)) for (auto& e : G.tev) CH(hipEventCreate(&e));
        }
    }
    // send buffers, staging, the de-interleave table: for the partition of the config (bhray_set_partition lays them out again)
    { int rc_ = layout_gather(c, false); if (rc_) { dev_set_create_error(c->err.c_str()); bhray_destroy(c); return rc_; } }
#undef CH
#undef CN
#undef FAIL
    // one process, N GPUs: one issue thread per GPU (see "Issue threads" above)
    {
        const char* e = getenv("BHRAY_ISSUE_THREADS");
        if (multi_dev && !(e && atoi(e) == 0)) {
            for (size_t k = 0; k < c->ranks.size(); k++) {
                std::unique_ptr<Worker> w(new (std::nothrow) Worker());
                if (!w) { bhray_destroy(c); return gfail(nullptr, BHRAY_E_NOMEM, "host allocation failed"); }
                w->rank_index = k;
                c->workers.push_back(std::move(w));
            }
            c->threaded = true;
            for (auto& w : c->workers) w->th = std::thread(worker_main, c, w.get());
        }
    }
    *out = c;
    return BHRAY_OK;
}

// ---- run-time partition ---------------------------------------------------------------------------
int bhray_get_partition(const bhray_ctx* c, uint32_t* slab_row0, uint32_t* partitions) {
    if (!c || !slab_row0) return BHRAY_E_INVALID;
    if (partitions) *partitions = c->world;
    if (c->world > 1 && c->cfg.partition != BHRAY_PARTITION_SLABS) return BHRAY_E_STATE;
    if (c->world <= 1) { slab_row0[0] = 0; slab_row0[1] = c->cfg.frame_h; return BHRAY_OK; }
    for (uint32_t p = 0; p <= c->world; p++) slab_row0[p] = c->cfg.slab_row0[p];
    return BHRAY_OK;
}

int bhray_set_partition(bhray_ctx* c, const uint32_t* slab_row0) {
    if (!c || !slab_row0) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->world > BHRAY_MAX_DEVICES) return gfail(c, BHRAY_E_INVALID, "more partitions than slab_row0 holds");
    bhray_config n = c->cfg;
    n.partition = BHRAY_PARTITION_SLABS;
    for (uint32_t p = 0; p <= c->world; p++) n.slab_row0[p] = slab_row0[p];
    if (const char* why = partition_error(n, c->world)) return gfail(c, BHRAY_E_INVALID, "bad row partition: %s", why);
    if (c->single) {
        bhray_dev* d = c->parts[0].dev;
        DEV(c, d, dev_set_partition(d, BHRAY_PARTITION_SLABS, 0, n.slab_row0, c->cfg.row_rank, c->world));
        c->cfg = n;
        c->repartitions++;
        return BHRAY_OK;
    }
    { int rc = group_sync(c); if (rc) return rc; }          // nothing in flight reads the tables that are rewritten below
    c->cfg = n;
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        p.row_list = partition_row_list(c->cfg, c->world, q);
        p.rows = (uint32_t)p.row_list.size();
        if (!p.dev) continue;
        DEV(c, p.dev, dev_set_partition(p.dev, BHRAY_PARTITION_SLABS, 0, n.slab_row0, q, c->world));
        if (dev_local_rows(p.dev) != p.rows) return gfail(c, BHRAY_E_STATE, "internal: partition %u has %u rows, expected %u", q, dev_local_rows(p.dev), p.rows);
    }
    { int rc = layout_gather(c, true); if (rc) return rc; }
    // a sky image resolved for a frame of the old partition stays valid (it is a whole-frame image on the root); nothing else is cached
    c->repartitions++;
    return BHRAY_OK;
}

int bhray_rebalance_slabs(uint32_t H, uint32_t world, const uint32_t* b_in, const double* part_ms, const double* extra_ms,
                          double shift_rows, double* w, uint32_t* b_out, double* slowest_predicted) {
    if (!b_in || !part_ms || !w || !b_out || H < 1 || world < 1 || world > BHRAY_MAX_DEVICES) return BHRAY_E_INVALID;
    if (b_in[0] != 0 || b_in[world] != H) return BHRAY_E_INVALID;
    for (uint32_t p = 0; p < world; p++) if (b_in[p] > b_in[p + 1] || !(part_ms[p] >= 0.0) || (extra_ms && !(extra_ms[p] >= 0.0))) return BHRAY_E_INVALID;
    bool any = false;
    for (uint32_t r = 0; r < H; r++) { if (!(w[r] >= 0.0)) return BHRAY_E_INVALID; any = any || w[r] > 0.0; }
    if (!any) for (uint32_t r = 0; r < H; r++) w[r] = 1.0;
    // (1) what was measured: the rows of partition p cost part_ms[p] together
    for (uint32_t p = 0; p < world; p++) {
        const uint32_t a = b_in[p], b = b_in[p + 1];
        if (b <= a || !(part_ms[p] > 0.0)) continue;
        double sum = 0.0;
        for (uint32_t r = a; r < b; r++) sum += w[r];
        if (!(sum > 0.0)) { for (uint32_t r = a; r < b; r++) w[r] = part_ms[p] / (double)(b - a); continue; }
        const double f = part_ms[p] / sum;
        for (uint32_t r = a; r < b; r++) w[r] *= f;
    }
    // (1b) the scene moves: what was learned about row r is expected at row r + shift_rows in the frames to come (the rows the hole and
    // the disk project to travel with the camera's pitch); linear interpolation, the frame's edge rows repeat
    if (shift_rows == shift_rows && shift_rows != 0.0 && fabs(shift_rows) < (double)H) {
        std::vector<double> old(w, w + H);
        for (uint32_t r = 0; r < H; r++) {
            double src = (double)r - shift_rows;
            if (src < 0.0) src = 0.0;
            if (src > (double)(H - 1)) src = (double)(H - 1);
            const uint32_t i0 = (uint32_t)src, i1 = i0 + 1 < H ? i0 + 1 : i0;
            const double t = src - (double)i0;
            w[r] = old[i0] * (1.0 - t) + old[i1] * t;
        }
    }
    // rows nobody measured (an empty partition's neighbours own them: none are left out) keep their weight; a row must cost something
    double total = 0.0;
    for (uint32_t r = 0; r < H; r++) total += w[r];
    if (!(total > 0.0)) return BHRAY_E_INVALID;
    const double floor_w = 1e-6 * total / (double)H;
    std::vector<double> pre((size_t)H + 1, 0.0);
    for (uint32_t r = 0; r < H; r++) pre[r + 1] = pre[r] + (w[r] > floor_w ? w[r] : floor_w);
    auto ex = [&](uint32_t p) { return extra_ms ? extra_ms[p] : 0.0; };
    // (2) bounds that minimise the largest (rows' weight + extra): bisection on the load, greedy packing decides feasibility
    auto pack = [&](double cap, uint32_t* bounds) -> bool {
        uint32_t a = 0;
        for (uint32_t p = 0; p < world; p++) {
            if (bounds) bounds[p] = a;
            const double budget = cap - ex(p);
            if (a < H && budget > 0.0) {
                uint32_t lo = a, hi = H;                          // largest b with pre[b] - pre[a] <= budget
                while (lo < hi) { const uint32_t mid = lo + (hi - lo + 1) / 2; if (pre[mid] - pre[a] <= budget) lo = mid; else hi = mid - 1; }
                a = lo;
            }
        }
        if (bounds) bounds[world] = H;
        return a >= H;
    };
    double lo = 0.0, hi = pre[H];
    for (uint32_t p = 0; p < world; p++) hi += ex(p);
    for (int it = 0; it < 80 && hi - lo > 1e-12 * hi; it++) { const double mid = 0.5 * (lo + hi); if (pack(mid, nullptr)) hi = mid; else lo = mid; }
    if (!pack(hi, b_out)) return BHRAY_E_STATE;
    // the greedy packing front-loads: even out neighbours (never beyond the cap found above)
    auto load = [&](uint32_t p, uint32_t a, uint32_t b) { return pre[b] - pre[a] + ex(p); };
    for (int round = 0; round < 8; round++) {
        for (uint32_t p = 1; p < world; p++) {
            const uint32_t a = b_out[p - 1], b = b_out[p + 1];
            if (b <= a) continue;
            uint32_t best = b_out[p]; double bestv = std::max(load(p - 1, a, best), load(p, best, b));
            uint32_t l2 = a, h2 = b;
            while (l2 < h2) { const uint32_t mid = l2 + (h2 - l2) / 2; if (load(p - 1, a, mid) < load(p, mid, b)) l2 = mid + 1; else h2 = mid; }
            for (uint32_t m = (l2 > a ? l2 - 1 : l2); m <= l2 && m <= b; m++) {
                const double v = std::max(load(p - 1, a, m), load(p, m, b));
                if (v < bestv) { bestv = v; best = m; }
            }
            b_out[p] = best;
        }
    }
    if (slowest_predicted) {
        double mx = 0.0;
        for (uint32_t p = 0; p < world; p++) mx = std::max(mx, load(p, b_out[p], b_out[p + 1]));
        *slowest_predicted = mx;
    }
    return BHRAY_OK;
}

// What a partition costs its GPU per frame, in integrator steps issued by a wave ("wave-steps"): the steps its trace waves issued (counted
// by the kernels: FrameLaunch::work - a property of the rays and of how they share waves, not of what else the GPU is doing or of how many
// frames are in flight) plus the pixels its classify launches visit, an HBM-bound pass, at their price in wave-steps; on the root also the
// pixels it receives and de-interleaves.  The prices are ratios of measured rates on MI355X (1920x1080: a wave-step of the RK kernel 207
// VALU instructions at ~0.7 of the issue rate = ~590 SIMD-cycles, of the Euler kernel 92 at ~0.52 = ~350; classify 21 SIMD-cycles of a
// saturated device per pixel; profiles/EXPERIMENTS.md R5.2), fitted against the ranks' wall times in profiles/r05_rebalance_emulated.json.
// BHRAY_REBALANCE_PRICES="classify_rk,classify_euler,gather" overrides (tuning).
struct RebalancePrices { double classify[2]; double gather; };
RebalancePrices rebalance_prices() {
    RebalancePrices p = {{0.035, 0.06}, 0.05};
    if (const char* e = getenv("BHRAY_REBALANCE_PRICES")) { double a, b, g; if (sscanf(e, "%lf,%lf,%lf", &a, &b, &g) == 3 && a >= 0 && b >= 0 && g >= 0) { p.classify[0] = a; p.classify[1] = b; p.gather = g; } }
    return p;
}

// cost[q] / extra[q] / have[q] of the partitions THIS ctx renders (the others stay 0 / false), for the bounds `cur`: see above
static int partition_costs(bhray_ctx* c, const uint32_t* cur, double* cost, double* extra, bool* have, uint32_t* frames) {
    const RebalancePrices price = rebalance_prices();
    const uint32_t N = c->world, H = c->cfg.frame_h;
    int method = -1;
    for (uint32_t q = 0; q < N; q++) {
        Part& p = c->parts[q];
        if (!p.dev) continue;
        double ws = 0.0, px = 0.0; uint32_t n = 0; int m = 0;
        DEV(c, p.dev, dev_get_work(p.dev, &ws, &px, &n, &m));
        if (n == 0) continue;
        have[q] = true;
        method = m;
        cost[q] = ws + price.classify[m ? 0 : 1] * px;
        if (frames && (q == c->root || !c->root_local)) *frames = n;
    }
    if (method < 0) return gfail(c, BHRAY_E_STATE, "nothing rendered since the ctx was created (or since its partition was set): no work has been counted");
    // the root's share of the gather: every pixel of the other partitions is received and de-interleaved there (with BHRAY_F_GATHER_SKY at half the bytes)
    if (c->root_local) extra[c->root] = price.gather * (c->gather_sky ? 0.5 : 1.0) * (double)c->cfg.frame_w * (double)(H - (cur[c->root + 1] - cur[c->root]));
    return BHRAY_OK;
}

int bhray_get_partition_costs(bhray_ctx* c, double* cost, double* extra, uint32_t* frames) {
    if (!c || !cost || !extra) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->world < 2 || c->single) return gfail(c, BHRAY_E_STATE, "bhray_get_partition_costs needs a ctx that gathers (device_count >= 2, or gather = BHRAY_GATHER_RCCL)");
    const uint32_t N = c->world, H = c->cfg.frame_h;
    uint32_t cur[BHRAY_MAX_DEVICES + 1];
    const bool was_slabs = c->cfg.partition == BHRAY_PARTITION_SLABS;
    for (uint32_t p = 0; p <= N; p++) cur[p] = was_slabs ? c->cfg.slab_row0[p] : (uint32_t)((uint64_t)H * p / N);
    { int rc = group_sync(c); if (rc) return rc; }
    bool have[BHRAY_MAX_DEVICES] = {false};
    for (uint32_t q = 0; q < N; q++) { cost[q] = 0.0; extra[q] = 0.0; }
    if (frames) *frames = 0;
    return partition_costs(c, cur, cost, extra, have, frames);
}

int bhray_rebalance(bhray_ctx* c, bhray_rebalance_info* out) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (out) memset(out, 0, sizeof *out);
    if (c->world < 2 || c->single) return gfail(c, BHRAY_E_STATE, "bhray_rebalance needs a ctx that gathers (device_count >= 2, or gather = BHRAY_GATHER_RCCL); a host that moves the tiles itself balances with bhray_get_work + bhray_rebalance_slabs");
    const uint32_t N = c->world, H = c->cfg.frame_h;
    // a ctx created with interleaved stripes starts from equal slabs
    uint32_t cur[BHRAY_MAX_DEVICES + 1];
    const bool was_slabs = c->cfg.partition == BHRAY_PARTITION_SLABS;
    for (uint32_t p = 0; p <= N; p++) cur[p] = was_slabs ? c->cfg.slab_row0[p] : (uint32_t)((uint64_t)H * p / N);
    { int rc = group_sync(c); if (rc) return rc; }
    double cost[BHRAY_MAX_DEVICES] = {0}, extra[BHRAY_MAX_DEVICES] = {0};
    uint32_t frames = 0;
    bool have[BHRAY_MAX_DEVICES] = {false};
    if (!was_slabs) {
        // A ctx created with interleaved stripes has measured STRIPE SETS, and their costs say nothing about contiguous slabs (rescaling
        // the rows of slab p by the cost of stripe set p polluted the learned row weights and made the first bounds arbitrary).  The first call
        // only moves the ctx to equal slabs - every rank takes this branch, there is nothing to exchange - and learning starts with the next.
        c->rebalances++;
        { int rc = bhray_set_partition(c, cur); if (rc) return rc; }
        if (out) { out->partitions = N; out->applied = 1u; for (uint32_t p = 0; p <= N; p++) out->slab_row0[p] = cur[p]; }
        return BHRAY_OK;
    }
    { int rc = partition_costs(c, cur, cost, extra, have, &frames); if (rc) return rc; }
    if (c->ranks.size() == 1 && c->comm_size > 1) {
        // one process per GPU: everybody learns everybody's number over the communicator, on the communication stream
        Rccl* R = rccl();
        if (!R) return gfail(c, BHRAY_E_COMM, "%s", g_rccl.error.c_str());
        CommRank& cr = c->ranks[0];
        GHIP(c, hipSetDevice(cr.device));
        if (!c->d_xchg) GHIP(c, hipMalloc(&c->d_xchg, (2 + 2 * (size_t)N) * sizeof(float)));
        const uint32_t me = c->cfg.row_rank;
        float mine[2] = {have[me] ? (float)cost[me] : -1.0f, (float)extra[me]};
        GHIP(c, hipMemcpyAsync(c->d_xchg, mine, sizeof mine, hipMemcpyHostToDevice, cr.stream));
        GNCCL(c, R, R->AllGather(c->d_xchg, c->d_xchg + 2, 2, ncclFloat32, cr.comm, cr.stream));
        std::vector<float> all(2 * (size_t)N);
        GHIP(c, hipMemcpyAsync(all.data(), c->d_xchg + 2, all.size() * sizeof(float), hipMemcpyDeviceToHost, cr.stream));
        GHIP(c, hipStreamSynchronize(cr.stream));
        WATCH_CHECK(c);
        for (uint32_t q = 0; q < N; q++) { have[q] = all[2 * q] >= 0.0f; cost[q] = have[q] ? all[2 * q] : 0.0; extra[q] = all[2 * q + 1]; }
    }
    // a partition without rows measures nothing and that is fine; a partition WITH rows and no measurement means it has not rendered yet
    for (uint32_t q = 0; q < N; q++) if (cur[q + 1] > cur[q] && !have[q]) return gfail(c, BHRAY_E_STATE, "bhray_rebalance: partition %u has rendered no frame yet", q);
    if (c->row_weight.size() != H) c->row_weight.assign(H, 0.0);
    // the frames to come are expected one period's displacement of the hole's projection further on (a camera that keeps pitching)
    double shift = 0.0;
    if (c->hole_row_valid && c->hole_row_prev_valid) shift = c->hole_row - c->hole_row_prev;
    if (!(fabs(shift) < 0.5 * (double)H)) shift = 0.0;          // a cut, not a motion
    c->hole_row_prev = c->hole_row; c->hole_row_prev_valid = c->hole_row_valid;
    uint32_t next[BHRAY_MAX_DEVICES + 1];
    double predicted = 0.0, before = 0.0;
    for (uint32_t q = 0; q < N; q++) before = std::max(before, cost[q] + extra[q]);
    { int rc = bhray_rebalance_slabs(H, N, cur, cost, extra, shift, c->row_weight.data(), next, &predicted); if (rc) return gfail(c, rc, "bhray_rebalance_slabs failed"); }
    c->rebalances++;
    bool differ = false;
    for (uint32_t p = 0; p <= N; p++) differ = differ || next[p] != cur[p];
    const bool apply = differ && predicted < 0.98 * before;
    if (apply) { int rc = bhray_set_partition(c, next); if (rc) return rc; }
    if (out) {
        out->partitions = N; out->applied = apply ? 1u : 0u; out->frames = frames;
        for (uint32_t p = 0; p <= N; p++) out->slab_row0[p] = apply ? next[p] : cur[p];
        for (uint32_t q = 0; q < N; q++) { out->part_cost[q] = (float)cost[q]; out->extra_cost[q] = (float)extra[q]; }
        out->slowest_before = (float)before; out->slowest_predicted = (float)predicted;
    }
    return BHRAY_OK;
}

// What the frames still held by the frame slots cost the partition(s) of this ctx (see bhray_rebalance): for a host that balances by itself.
int bhray_get_work(bhray_ctx* c, double* wave_steps_per_frame, double* classify_pixels_per_frame, uint32_t* frames) {
    if (!c || !wave_steps_per_frame || !classify_pixels_per_frame) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->gather) { int rc = group_sync(c); if (rc) return rc; }
    double ws = 0.0, px = 0.0; uint32_t n = 0;
    for (Part& p : c->parts) {
        if (!p.dev) continue;
        double a = 0.0, b = 0.0; uint32_t k = 0; int m = 0;
        DEV(c, p.dev, dev_get_work(p.dev, &a, &b, &k, &m));
        ws += a; px += b; n = std::max(n, k);
    }
    *wave_steps_per_frame = ws; *classify_pixels_per_frame = px;
    if (frames) *frames = n;
    return BHRAY_OK;
}

int bhray_get_gather_info(const bhray_ctx* c, bhray_gather_info* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    memset(out, 0, sizeof *out);
    out->partitions = c->world;
    out->root = c->root;
    const uint64_t rowb = c->gather_sky ? (uint64_t)c->cfg.frame_w * 8u : (uint64_t)packed_row_words(c->cfg.frame_w) * 4u;    // bytes of one row on the wire
    for (uint32_t q = 0; q < c->parts.size(); q++) {
        const Part& p = c->parts[q];
        if (p.dev) out->local_partitions++;
        if (!c->gather) continue;
        if (p.dev && q != c->root) out->bytes_sent_per_frame += p.rows * rowb;
        if (c->root_local && q != c->root) out->bytes_received_per_frame += p.rows * rowb;
    }
    out->root_is_local = c->gather ? (c->root_local ? 1u : 0u) : 1u;
    out->comm_ranks = c->gather ? c->comm_size : 0;
    out->rccl_version = g_rccl.so ? (uint32_t)g_rccl.version : 0;
    return BHRAY_OK;
}

// diagnostics, not declared in include/bhray.h (scratch experiments read the temporal fix-up sets through it); single-partition ctx only
int bhray_debug_read_queue(bhray_ctx* c, uint32_t level, uint32_t* out, uint32_t cap, uint32_t* count) {
    if (!c || !c->single) return BHRAY_E_INVALID;
    return dev_debug_read_queue(c->parts[0].dev, level, out, cap, count);
}

// ---- scene state: replicated on every local partition ---------------------------------------
int bhray_set_texture(bhray_ctx* c, int slot, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->gather) { int rc = group_sync(c); if (rc) return rc; }
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_set_texture(p.dev, slot, rgba8, w, h));
    return BHRAY_OK;
}
int bhray_upload_model_uniform(bhray_ctx* c, uint32_t mi, const void* bytes, size_t size) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->gather) { int rc = group_sync(c); if (rc) return rc; }
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_upload_model_uniform(p.dev, mi, bytes, size));
    if (mi < BHRAY_MAX_MODELS && bytes && size == BHRAY_MODEL_UNIFORM_BYTES) {      // the caller's mirror of what decides the kernel variant
        bhray_model_header hd; memcpy(&hd, bytes, sizeof hd);
        c->m_usable[mi] = hd.triangle_count > 0; c->m_vis[mi] = hd.visible; memcpy(c->m_pos[mi], hd.position, 12); c->m_dirty[mi] = false;
    }
    return BHRAY_OK;
}
int bhray_upload_model(bhray_ctx* c, uint32_t mi, const bhray_model_desc* d) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->gather) { int rc = group_sync(c); if (rc) return rc; }
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_upload_model(p.dev, mi, d));
    if (mi < BHRAY_MAX_MODELS && d) { c->m_usable[mi] = d->triangle_count > 0; c->m_vis[mi] = d->visible; memcpy(c->m_pos[mi], d->position, 12); c->m_dirty[mi] = false; }
    return BHRAY_OK;
}
int bhray_set_model_transform(bhray_ctx* c, uint32_t mi, const float position[3], int32_t visible) {
    if (!c) return BHRAY_E_INVALID;
    if (c->threaded) {               // travels with the next frame (the engines belong to their issue threads)
        if (mi >= BHRAY_MAX_MODELS || !position) return gfail(c, BHRAY_E_INVALID, "bad model arguments");
        memcpy(c->m_pos[mi], position, 12); c->m_vis[mi] = visible; c->m_dirty[mi] = true;
        return BHRAY_OK;
    }
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_set_model_transform(p.dev, mi, position, visible));
    return BHRAY_OK;
}
int bhray_set_materials(bhray_ctx* c, const void* bytes, size_t size) {
    if (!c) return BHRAY_E_INVALID;
    if (!bytes || size != 16u * BHRAY_MAX_MATERIALS) return gfail(c, BHRAY_E_INVALID, "materials must be %u bytes (MaterialUniform x %d)", 16u * BHRAY_MAX_MATERIALS, BHRAY_MAX_MATERIALS);
    return BHRAY_OK;                      // bound at binding 3, never read by the shader (ray.wgsl:8)
}
int bhray_set_uniforms(bhray_ctx* c, const void* cam32, const void* bh132, const void* det32) {
    if (!c) return BHRAY_E_INVALID;
    if (c->gather && cam32 && bh132) track_hole_row(c, cam32, bh132);
    if (c->threaded) {               // travel with the next frame
        if (!cam32 || !bh132 || !det32) return gfail(c, BHRAY_E_INVALID, "null uniform block");
        memcpy(c->u_cam, cam32, 32); memcpy(c->u_bh, bh132, 132); memcpy(c->u_det, det32, 32);
        c->have_uniforms = true;
        return BHRAY_OK;
    }
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_set_uniforms(p.dev, cam32, bh132, det32));
    return BHRAY_OK;
}

// ---- dispatch --------------------------------------------------------------------------------
int bhray_render(bhray_ctx* c) {
    if (!c) return BHRAY_E_INVALID;
    if (c->single) {
        bhray_dev* d = c->parts[0].dev;
        if (c->bound) { DEV(c, d, dev_bind_output(d, c->bound, (size_t)-1)); c->bound = nullptr; }
        DEV(c, d, dev_render(d));
        for (WaitEvent& w : c->wait_pool) w.in_use = false;      // the render's stream waits have been enqueued
        c->rendered = true;
        return BHRAY_OK;
    }
    if (c->watch_fired.load(std::memory_order_acquire)) return gfail(c, BHRAY_E_COMM, "bhray_render refused");
    if (c->threaded) {
        // record the frame and hand it to the issue threads ("Issue threads" above); nothing here touches an engine or a GPU
        if (c->async_failed.load(std::memory_order_acquire)) return drain(c);
        if (!c->have_uniforms) return gfail(c, BHRAY_E_STATE, "bhray_set_uniforms has not been called");
        if (c->mirror_stale) {       // the threads are idle (every call but this one waits for them): read the staging position back
            for (Part& p : c->parts) {
                if (!p.dev) continue;
                int m; bool md;
                dev_peek_position(p.dev, &c->st_counter, &c->st_pending, &m, &md);
                c->st_method = m; c->st_models = md;
                break;
            }
            c->mirror_stale = false;
        }
        int method; bool models;
        ctx_variant(c, method, models);
        if (c->st_pending > 0 && (method != c->st_method || models != c->st_models)) { c->st_counter++; c->st_pending = 0; }   // the engines launch the staged frames first
        if (c->st_pending == 0) { c->st_method = method; c->st_models = models; }
        RenderCmd cmd;
        memcpy(cmd.cam, c->u_cam, 32); memcpy(cmd.bh, c->u_bh, 132); memcpy(cmd.det, c->u_det, 32);
        for (uint32_t mi = 0; mi < BHRAY_MAX_MODELS; mi++) { memcpy(cmd.mpos[mi], c->m_pos[mi], 12); cmd.mvis[mi] = c->m_vis[mi]; cmd.mset[mi] = c->m_dirty[mi]; c->m_dirty[mi] = false; }
        cmd.si = (int)(c->st_counter % c->nslots); cmd.sub = c->st_pending;
        cmd.bound = c->bound; c->bound = nullptr;
        cmd.serial = ++c->frames_staged;
        if (++c->st_pending == c->B) { c->st_counter++; c->st_pending = 0; }
        c->last_slot = cmd.si; c->last_sub = cmd.sub; c->rendered = true;
        for (WaitEvent& w : c->wait_pool) w.in_use = false;
        for (auto& w : c->workers) {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->q.size() < 256; });        // (a host far ahead of its GPUs)
            w->q.push_back(cmd);
            lk.unlock();
            w->cv.notify_all();
        }
        return BHRAY_OK;
    }
    // staged frames of another kernel variant are launched (and gathered) first: the staging position is then final
    WatchScope watch_scope_(c, "bhray_render");
    int si = -1; uint32_t sub = 0;
    for (Part& p : c->parts) {
        if (!p.dev) continue;
        int s; uint32_t k;
        DEV(c, p.dev, dev_next_position(p.dev, &s, &k));
        if (si >= 0 && (s != si || k != sub)) return gfail(c, BHRAY_E_STATE, "internal: partitions out of step");
        si = s; sub = k;
    }
    { int rc = after_launch(c); if (rc) return rc; }
    { int rc = bind_partitions(c, si, sub, c->bound, c->frames_staged + 1); if (rc) return rc; }
    c->bound = nullptr;
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_render(p.dev));
    for (WaitEvent& w : c->wait_pool) w.in_use = false;
    c->last_slot = si; c->last_sub = sub; c->rendered = true;
    ++c->frames_staged;
    return after_launch(c);
}

int bhray_flush(bhray_ctx* c) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_flush(c->parts[0].dev)); return BHRAY_OK; }
    return group_flush(c);
}

int bhray_sync(bhray_ctx* c) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_sync(c->parts[0].dev)); return BHRAY_OK; }
    return group_sync(c);
}

// ---- output ----------------------------------------------------------------------------------
uint32_t bhray_local_rows(const bhray_ctx* c) {
    if (!c) return 0;
    if (c->single) return dev_local_rows(c->parts[0].dev);
    return c->root_local ? c->cfg.frame_h : 0;
}

int bhray_local_row_index(const bhray_ctx* c, uint32_t i, uint32_t* frame_row) {
    if (!c || !frame_row) return BHRAY_E_INVALID;
    if (c->single) return dev_local_row_index(c->parts[0].dev, i, frame_row);
    if (!c->root_local || i >= c->cfg.frame_h) return BHRAY_E_INVALID;
    *frame_row = i;
    return BHRAY_OK;
}

int bhray_read_hdr(bhray_ctx* c, float* dst, size_t pitch) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_read_hdr(c->parts[0].dev, dst, pitch)); return BHRAY_OK; }
    if (c->gather_sky) return gfail(c, BHRAY_E_STATE, "BHRAY_F_GATHER_SKY: the RGBA32F frame is not assembled (read the sky image)");
    { int rc = group_sync(c); if (rc) return rc; }
    if (!c->root_local) return BHRAY_OK;                        // the frame lives on another rank
    if (!c->rendered) return gfail(c, BHRAY_E_STATE, "nothing rendered yet");
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(float4);
    if (!dst || pitch < rowb) return gfail(c, BHRAY_E_INVALID, "bad destination / pitch");
    GHIP(c, hipSetDevice(root_part(c)->device));
    GHIP(c, hipMemcpy2D(dst, pitch, c->gslots[(size_t)c->last_slot].dst[c->last_sub], rowb, rowb, c->cfg.frame_h, hipMemcpyDeviceToHost));
    return BHRAY_OK;
}

int bhray_read_level(bhray_ctx* c, uint32_t level, float* dst, size_t pitch) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_read_level(c->parts[0].dev, level, dst, pitch)); return BHRAY_OK; }
    // every partition computed the level rows its stripes depend on: overlay them (unrendered pixels are NaN-filled; pixels
    // computed by several partitions are identical)
    if (level >= c->cfg.levels) return gfail(c, BHRAY_E_INVALID, "level out of range");
    { int rc = group_sync(c); if (rc) return rc; }
    const size_t w = c->cfg.level_w[level], h = c->cfg.level_h[level], rowb = w * sizeof(float4);
    if (!dst || pitch < rowb) return gfail(c, BHRAY_E_INVALID, "bad destination / pitch");
    for (size_t y = 0; y < h; y++) memset((uint8_t*)dst + y * pitch, 0xFF, rowb);
    std::vector<uint32_t> tmp(w * h * 4);
    const bool last = level + 1 == c->cfg.levels;
    for (uint32_t q = 0; q < c->world; q++) {
        Part& p = c->parts[q];
        if (!p.dev) continue;
        if (last) {
            // a partition's last-level image is its output binding: the root's is the assembled frame, read through the ctx
            if (q != c->root) continue;
            std::vector<float> fr(frame_pixels(c) * 4);
            int rc = bhray_read_hdr(c, fr.data(), (size_t)c->cfg.frame_w * 16);
            if (rc) return rc;
            for (size_t y = 0; y < c->cfg.frame_h; y++)
                memcpy((uint8_t*)dst + (y + c->cfg.crop_y) * pitch + (size_t)c->cfg.crop_x * 16, fr.data() + y * c->cfg.frame_w * 4, (size_t)c->cfg.frame_w * 16);
            continue;
        }
        DEV(c, p.dev, dev_read_level(p.dev, level, (float*)tmp.data(), rowb));
        for (size_t y = 0; y < h; y++) {
            const uint32_t* s = tmp.data() + y * w * 4;
            uint32_t* d = (uint32_t*)((uint8_t*)dst + y * pitch);
            for (size_t x = 0; x < w; x++) {
                const uint32_t* px = s + 4 * x;
                if (px[0] == 0xFFFFFFFFu && px[1] == 0xFFFFFFFFu && px[2] == 0xFFFFFFFFu && px[3] == 0xFFFFFFFFu) continue;
                memcpy(d + 4 * x, px, 16);
            }
        }
    }
    return BHRAY_OK;
}

int bhray_hdr_device_ptr(bhray_ctx* c, void** p, size_t* bytes) {
    if (!c || !p) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_hdr_device_ptr(c->parts[0].dev, p, bytes)); return BHRAY_OK; }
    if (c->gather_sky) { *p = nullptr; return gfail(c, BHRAY_E_STATE, "BHRAY_F_GATHER_SKY: the RGBA32F frame is not assembled (read the sky image)"); }
    *p = c->root_local ? (void*)c->gslots[(size_t)c->last_slot].dst[c->last_sub] : nullptr;
    if (bytes) *bytes = c->root_local ? frame_pixels(c) * sizeof(float4) : 0;
    return BHRAY_OK;
}

int bhray_bind_output(bhray_ctx* c, void* p, size_t bytes) {
    if (!c) return BHRAY_E_INVALID;
    if (!p) { c->bound = nullptr; return BHRAY_OK; }
    if (c->gather_sky) return gfail(c, BHRAY_E_STATE, "BHRAY_F_GATHER_SKY: the RGBA32F frame is not assembled (read the sky image)");
    size_t need;
    if (c->single) { void* q; DEV(c, c->parts[0].dev, dev_hdr_device_ptr(c->parts[0].dev, &q, &need)); }
    else need = c->root_local ? frame_pixels(c) * sizeof(float4) : 0;
    if (bytes < need) return gfail(c, BHRAY_E_INVALID, "output binding needs %zu bytes", need);
    if (((uintptr_t)p & 15u) != 0) return gfail(c, BHRAY_E_INVALID, "output binding must be 16-byte aligned");
    c->bound = (float4*)p;
    return BHRAY_OK;
}

// ---- asynchronous hand-off -------------------------------------------------------------------------
int bhray_read_hdr_async(bhray_ctx* c, float* dst, size_t pitch, uint64_t* ticket) {
    if (!c || !ticket) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_read_hdr_async(c->parts[0].dev, dst, pitch, ticket)); return BHRAY_OK; }
    if (c->gather_sky) return gfail(c, BHRAY_E_STATE, "BHRAY_F_GATHER_SKY: the RGBA32F frame is not assembled (read the sky image)");
    if (!c->rendered) return gfail(c, BHRAY_E_STATE, "nothing rendered yet");
    { int rc = group_flush(c); if (rc) return rc; }
    const uint64_t t = c->read_tickets;
    if (!c->root_local) { *ticket = t; c->read_tickets = t + 1; return BHRAY_OK; }   // the frame lives on another rank: nothing to copy here
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(float4);
    if (!dst || pitch < rowb) return gfail(c, BHRAY_E_INVALID, "bad destination / pitch");
    Part& rp = *root_part(c);
    CommRank* rr = rank_of(c, rp);
    GHIP(c, hipSetDevice(rp.device));
    hipEvent_t& ev = c->read_ev[t % 64];
    if (!ev) GHIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (t >= 64) { GHIP(c, hipEventSynchronize(ev)); WATCH_CHECK(c); }
    GroupSlot& G = c->gslots[(size_t)c->last_slot];
    // in stream order on the root's communication stream: behind the de-interleave (and the sky pass, if it was resolved) of this batch,
    // ahead of the next batch's receive.  (Cross-stream event ordering made ROCm 7.2 serialise copies and kernels: see dev_read_hdr_async.)
    if (pitch == rowb) GHIP(c, hipMemcpyAsync(dst, G.dst[c->last_sub], rowb * c->cfg.frame_h, hipMemcpyDeviceToHost, rr->stream));
    else GHIP(c, hipMemcpy2DAsync(dst, pitch, G.dst[c->last_sub], rowb, rowb, c->cfg.frame_h, hipMemcpyDeviceToHost, rr->stream));
    GHIP(c, hipEventRecord(ev, rr->stream));
    *ticket = t; c->read_tickets = t + 1;                       // the ticket exists once its event does (an error above consumes none)
    GHIP(c, hipEventRecord(G.frame_done, rr->stream));
    GHIP(c, hipStreamWaitEvent(dev_slot_stream(rp.dev, c->last_slot), G.frame_done, 0));   // the root's next render into this slot writes its own rows into that frame
    return BHRAY_OK;
}

int bhray_read_sky_async(bhray_ctx* c, uint16_t* dst, size_t pitch, uint64_t* ticket) {
    if (!c || !ticket) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_read_sky_async(c->parts[0].dev, dst, pitch, ticket)); return BHRAY_OK; }
    if (!c->rendered) return gfail(c, BHRAY_E_STATE, "nothing rendered yet");
    if (c->gather_sky) { int rc = group_flush(c); if (rc) return rc; }      // (the batch's gather produces the image)
    const uint64_t t = c->read_tickets;
    if (!c->root_local) { *ticket = t; c->read_tickets = t + 1; return BHRAY_OK; }
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(uint2);
    if (!dst || pitch < rowb) return gfail(c, BHRAY_E_INVALID, "bad destination / pitch");
    const GroupSlot& GS = c->gslots[(size_t)c->last_slot];
    uint2* src = GS.sky[c->last_sub];
    if (!src || GS.sky_frame_no[c->last_sub] != GS.frame_no[c->last_sub]) return gfail(c, BHRAY_E_STATE, "bhray_resolve_sky has not been called for this frame");
    Part& rp = *root_part(c);
    CommRank* rr = rank_of(c, rp);
    GHIP(c, hipSetDevice(rp.device));
    hipEvent_t& ev = c->read_ev[t % 64];
    if (!ev) GHIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else if (t >= 64) { GHIP(c, hipEventSynchronize(ev)); WATCH_CHECK(c); }
    if (pitch == rowb) GHIP(c, hipMemcpyAsync(dst, src, rowb * c->cfg.frame_h, hipMemcpyDeviceToHost, rr->stream));     // behind the sky pass, in stream order
    else GHIP(c, hipMemcpy2DAsync(dst, pitch, src, rowb, rowb, c->cfg.frame_h, hipMemcpyDeviceToHost, rr->stream));
    GHIP(c, hipEventRecord(ev, rr->stream));
    *ticket = t; c->read_tickets = t + 1;
    return BHRAY_OK;
}

int bhray_wait_read(bhray_ctx* c, uint64_t ticket) {
    if (!c) return BHRAY_E_INVALID;
    if (c->single) { DEV(c, c->parts[0].dev, dev_wait_read(c->parts[0].dev, ticket)); return BHRAY_OK; }
    if (ticket >= c->read_tickets) return gfail(c, BHRAY_E_INVALID, "unknown read ticket");
    if (!c->root_local || c->read_tickets - ticket > 64 || !c->read_ev[ticket % 64]) return BHRAY_OK;
    GHIP(c, hipSetDevice(root_part(c)->device));
    GHIP(c, hipEventSynchronize(c->read_ev[ticket % 64]));
    WATCH_CHECK(c);
    return BHRAY_OK;
}

int bhray_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return BHRAY_E_INVALID;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return gfail(nullptr, e == hipErrorOutOfMemory ? BHRAY_E_NOMEM : BHRAY_E_HIP, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return BHRAY_OK;
}

int bhray_host_free(void* p) {
    if (!p) return BHRAY_OK;
    hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return gfail(nullptr, BHRAY_E_HIP, "hipHostFree: %s", hipGetErrorString(e));
    return BHRAY_OK;
}

// ---- zero-copy hand-off: memory exported by the consumer's API ---------------------------------------
int bhray_import_external_fd(bhray_ctx* c, int fd, size_t bytes, void** dev_ptr) {
    if (!c || !dev_ptr || fd < 0 || bytes == 0) return BHRAY_E_INVALID;
    *dev_ptr = nullptr;
    const int dev = c->single ? c->parts[0].device : (c->root_local ? root_part(c)->device : -1);
    if (dev < 0) return gfail(c, BHRAY_E_STATE, "this rank does not deliver the frame");
    GHIP(c, hipSetDevice(dev));
    // the runtime owns an imported descriptor (CUDA semantics): hand it a duplicate, the caller keeps the original (include/bhray.h)
    const int own = dup(fd);
    if (own < 0) return gfail(c, BHRAY_E_INVALID, "dup(fd %d) failed: not an open descriptor", fd);
    const FdId own_id = fd_identity(own);
    hipExternalMemoryHandleDesc hd; memset(&hd, 0, sizeof hd);
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = own;
    hd.size = bytes;
    hipExternalMemory_t em = nullptr;
    hipError_t e = hipImportExternalMemory(&em, &hd);
    if (e != hipSuccess) { (void)close(own); return gfail(c, BHRAY_E_HIP, "hipImportExternalMemory(fd %d, %zu bytes): %s", fd, bytes, hipGetErrorString(e)); }
    hipExternalMemoryBufferDesc bd; memset(&bd, 0, sizeof bd);
    bd.offset = 0; bd.size = bytes;
    void* p = nullptr;
    e = hipExternalMemoryGetMappedBuffer(&p, em, &bd);
    if (e != hipSuccess || !p) { (void)hipDestroyExternalMemory(em); close_if_still_ours(own, own_id); return gfail(c, BHRAY_E_HIP, "hipExternalMemoryGetMappedBuffer: %s", hipGetErrorString(e)); }
    c->external.push_back(p); c->external.push_back((void*)em);
    c->external_fd.push_back(own); c->external_fd_id.push_back(own_id);
    *dev_ptr = p;
    return BHRAY_OK;
}

int bhray_release_external(bhray_ctx* c, void* dev_ptr) {
    if (!c || !dev_ptr) return BHRAY_E_INVALID;
    for (size_t i = 0; i + 1 < c->external.size(); i += 2) {
        if (c->external[i] != dev_ptr) continue;
        { int rc = bhray_sync(c); if (rc) return rc; }
        hipError_t e = hipDestroyExternalMemory((hipExternalMemory_t)c->external[i + 1]);
        c->external.erase(c->external.begin() + (long)i, c->external.begin() + (long)i + 2);
        close_if_still_ours(c->external_fd[i / 2], c->external_fd_id[i / 2]);
        c->external_fd.erase(c->external_fd.begin() + (long)(i / 2)); c->external_fd_id.erase(c->external_fd_id.begin() + (long)(i / 2));
        if (e != hipSuccess) return gfail(c, BHRAY_E_HIP, "hipDestroyExternalMemory: %s", hipGetErrorString(e));
        return BHRAY_OK;
    }
    return gfail(c, BHRAY_E_INVALID, "not a pointer returned by bhray_import_external_fd");
}

// ---- ordering against caller streams ------------------------------------------------------------
int bhray_wait_stream(bhray_ctx* c, void* s) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    // One event per call (two calls with different streams before one render are two dependencies), recorded with the STREAM's
    // device current (a stream of another partition's GPU is fine: the renders wait across devices).  The events are handed
    // to every local partition and recycled once the next bhray_render has consumed them.
    int dev = c->single ? c->parts[0].device : (c->root_local ? root_part(c)->device : c->ranks[0].device);
    if (s) { int sd = -1; if (hipStreamGetDevice((hipStream_t)s, &sd) == hipSuccess && sd >= 0) dev = sd; }
    GHIP(c, hipSetDevice(dev));
    hipEvent_t ev = nullptr;
    for (WaitEvent& w : c->wait_pool) if (!w.in_use && w.device == dev) { w.in_use = true; ev = w.ev; break; }
    if (!ev) {
        GHIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        WaitEvent w; w.ev = ev; w.device = dev; w.in_use = true;
        c->wait_pool.push_back(w);
    }
    GHIP(c, hipEventRecord(ev, (hipStream_t)s));
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_wait_event(p.dev, ev));
    return BHRAY_OK;
}

int bhray_next_stream(bhray_ctx* c, void** s) {
    if (!c || !s) return BHRAY_E_INVALID;
    if (c->single) { DEV(c, c->parts[0].dev, dev_next_stream(c->parts[0].dev, s)); return BHRAY_OK; }
    CommRank* rr = c->root_local ? rank_of(c, *root_part(c)) : &c->ranks[0];
    *s = (void*)rr->stream;               // the gather and the de-interleave of every batch run on this stream
    return BHRAY_OK;
}

int bhray_signal_stream(bhray_ctx* c, void* s) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_signal_stream(c->parts[0].dev, s)); return BHRAY_OK; }
    if (!c->rendered) return BHRAY_OK;
    { int rc = group_flush(c); if (rc) return rc; }
    if (c->root_local) {
        GHIP(c, hipSetDevice(root_part(c)->device));
        GHIP(c, hipStreamWaitEvent((hipStream_t)s, c->gslots[(size_t)c->last_slot].frame_done, 0));
    } else {
        const Part& me = c->parts[c->cfg.row_rank];
        GHIP(c, hipSetDevice(me.device));
        GHIP(c, hipStreamWaitEvent((hipStream_t)s, c->gslots[(size_t)c->last_slot].sent[c->cfg.row_rank], 0));
    }
    return BHRAY_OK;
}

// ---- sky resolve -----------------------------------------------------------------------------------
int bhray_resolve_sky(bhray_ctx* c) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_resolve_sky(c->parts[0].dev)); return BHRAY_OK; }
    if (!c->rendered) return gfail(c, BHRAY_E_STATE, "nothing rendered yet");
    { int rc = group_flush(c); if (rc) return rc; }
    if (c->gather_sky) return BHRAY_OK;              // the partitions resolved their rows behind the render; the gather assembled the image
    if (!c->root_local) return BHRAY_OK;
    Part& rp = *root_part(c);
    CommRank* rr = rank_of(c, rp);
    GroupSlot& G = c->gslots[(size_t)c->last_slot];
    GHIP(c, hipSetDevice(rp.device));
    if (!G.sky[c->last_sub]) GHIP(c, hipMalloc(&G.sky[c->last_sub], frame_pixels(c) * sizeof(uint2)));
    DEV(c, rp.dev, dev_launch_sky(rp.dev, G.dst[c->last_sub], G.sky[c->last_sub], frame_pixels(c), rr->stream));   // behind the de-interleave
    G.sky_frame_no[c->last_sub] = G.frame_no[c->last_sub];
    GHIP(c, hipEventRecord(G.frame_done, rr->stream));
    // the root's next render into this slot writes its own rows straight into the frame the sky pass is reading (the wait that
    // group_gather enqueued captured the EARLIER record of frame_done, behind the de-interleave): order it behind this one
    GHIP(c, hipStreamWaitEvent(dev_slot_stream(rp.dev, c->last_slot), G.frame_done, 0));
    return BHRAY_OK;
}

int bhray_read_sky(bhray_ctx* c, uint16_t* dst, size_t pitch) {
    if (!c) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_read_sky(c->parts[0].dev, dst, pitch)); return BHRAY_OK; }
    { int rc = group_sync(c); if (rc) return rc; }
    if (!c->root_local) return BHRAY_OK;
    const size_t rowb = (size_t)c->cfg.frame_w * sizeof(uint2);
    if (!dst || pitch < rowb) return gfail(c, BHRAY_E_INVALID, "bad destination / pitch");
    const GroupSlot& GS = c->gslots[(size_t)c->last_slot];
    uint2* src = GS.sky[c->last_sub];
    if (!src || GS.sky_frame_no[c->last_sub] != GS.frame_no[c->last_sub]) return gfail(c, BHRAY_E_STATE, "bhray_resolve_sky has not been called for this frame");
    GHIP(c, hipSetDevice(root_part(c)->device));
    GHIP(c, hipMemcpy2D(dst, pitch, src, rowb, rowb, c->cfg.frame_h, hipMemcpyDeviceToHost));
    return BHRAY_OK;
}

int bhray_sky_device_ptr(bhray_ctx* c, void** p, size_t* bytes) {
    if (!c || !p) return BHRAY_E_INVALID;
    if (c->single) { DEV(c, c->parts[0].dev, dev_sky_device_ptr(c->parts[0].dev, p, bytes)); return BHRAY_OK; }
    ENTER(c);
    *p = nullptr;
    if (c->root_local) {                 // the same rule as the reads: an image resolved from an earlier frame of this slot position is stale
        const GroupSlot& GS = c->gslots[(size_t)c->last_slot];
        if (!GS.sky[c->last_sub] || GS.sky_frame_no[c->last_sub] != GS.frame_no[c->last_sub]) return gfail(c, BHRAY_E_STATE, "bhray_resolve_sky has not been called for this frame");
    }
    *p = c->root_local ? (void*)c->gslots[(size_t)c->last_slot].sky[c->last_sub] : nullptr;
    if (bytes) *bytes = c->root_local ? frame_pixels(c) * sizeof(uint2) : 0;
    return BHRAY_OK;
}

// ---- measurement -------------------------------------------------------------------------------------
int bhray_selftest(bhray_ctx* c, uint64_t mismatches[3]) {
    if (!c || !mismatches) return BHRAY_E_INVALID;
    ENTER(c);
    uint64_t sum[3] = {0, 0, 0};
    for (Part& p : c->parts) {
        if (!p.dev) continue;
        uint64_t m[3];
        DEV(c, p.dev, dev_selftest(p.dev, m));
        for (int k = 0; k < 3; k++) sum[k] += m[k];
    }
    for (int k = 0; k < 3; k++) mismatches[k] = sum[k];
    return BHRAY_OK;
}

int bhray_get_level_counters(bhray_ctx* c, uint32_t level, bhray_counters* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->gather) { int rc = group_sync(c); if (rc) return rc; }
    memset(out, 0, sizeof *out);
    for (Part& p : c->parts) {                                   // local partitions only: whole-frame work = sum over all
        if (!p.dev) continue;
        bhray_counters t;
        DEV(c, p.dev, dev_get_level_counters(p.dev, level, &t));
        const uint64_t* a = (const uint64_t*)&t; uint64_t* b = (uint64_t*)out;
        for (size_t k = 0; k < sizeof(bhray_counters) / 8; k++) { if (k == 12) b[k] = a[k] > b[k] ? a[k] : b[k]; else b[k] += a[k]; }   // [12] max_ray_iterations
    }
    return BHRAY_OK;
}

int bhray_get_row_work(bhray_ctx* c, uint32_t level, uint64_t* out, uint32_t n) {
    if (!c || !out) return BHRAY_E_INVALID;
    ENTER(c);
    if (level >= c->cfg.levels || n != c->cfg.level_h[level]) return gfail(c, BHRAY_E_INVALID, "level out of range, or n is not the level's height");
    if (c->gather) { int rc = group_sync(c); if (rc) return rc; }
    memset(out, 0, (size_t)n * sizeof(uint64_t));
    for (Part& p : c->parts) if (p.dev) DEV(c, p.dev, dev_add_row_work(p.dev, level, out, n));
    return BHRAY_OK;
}

int bhray_get_counters(bhray_ctx* c, bhray_counters* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    memset(out, 0, sizeof *out);
    for (uint32_t l = 0; l < c->cfg.levels; l++) {
        bhray_counters t;
        int rc = bhray_get_level_counters(c, l, &t);
        if (rc) return rc;
        const uint64_t* a = (const uint64_t*)&t; uint64_t* b = (uint64_t*)out;
        for (size_t k = 0; k < sizeof(bhray_counters) / 8; k++) { if (k == 12) b[k] = a[k] > b[k] ? a[k] : b[k]; else b[k] += a[k]; }   // [12] max_ray_iterations
    }
    return BHRAY_OK;
}

int bhray_get_timing(bhray_ctx* c, bhray_timing* out) {
    if (!c || !out) return BHRAY_E_INVALID;
    ENTER(c);
    if (c->single) { DEV(c, c->parts[0].dev, dev_get_timing(c->parts[0].dev, out)); return BHRAY_OK; }
    if (!(c->cfg.flags & (BHRAY_F_TIMING | BHRAY_F_TIMING_SPARSE))) return gfail(c, BHRAY_E_STATE, "ctx created without BHRAY_F_TIMING");
    { int rc = group_sync(c); if (rc) return rc; }
    // kernel times: the root partition's (or this rank's) launches; gather: the root's communication stream
    Part* src = c->root_local ? root_part(c) : &c->parts[c->cfg.row_rank];
    for (Part& p : c->parts) {
        if (!p.dev) continue;
        bhray_timing t;
        DEV(c, p.dev, dev_get_timing(p.dev, &t));               // resets every partition's aggregation
        if (&p == src) *out = t;
    }
    if (c->root_local) {
        GHIP(c, hipSetDevice(root_part(c)->device));
        for (GroupSlot& G : c->gslots) {
            if (!G.timed) continue;
            float a = 0, b = 0;
            GHIP(c, hipEventElapsedTime(&a, G.tev[0], G.tev[1])); GHIP(c, hipEventElapsedTime(&b, G.tev[1], G.tev[2]));
            c->gather_ms += a; c->deint_ms += b; G.timed = false;
        }
    }
    out->gather_ms = c->gather_ms; out->deinterleave_ms = c->deint_ms; out->gathers = c->gathers;
    c->gather_ms = 0; c->deint_ms = 0; c->gathers = 0;
    return BHRAY_OK;
}

}  // extern "C"
