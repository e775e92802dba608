// bhray_dev.h — the per-device engine behind the public bhray_ctx.
//
// A bhray_dev renders ONE row partition of the frame on ONE GPU (bhray_api.hip): ladder levels, frame slots, frame batches,
// speculative levels.  The public bhray_ctx (bhray_group.hip) owns one bhray_dev per local partition and, when the frame is
// split over several partitions, the gather of the row tiles to the root partition's GPU (RCCL) and the de-interleave into
// the frame.  Nothing here is exported; the C ABI is include/bhray.h.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "../../include/bhray.h"

#include <vector>

struct bhray_dev;

namespace bhray {
// Row partition arithmetic shared by the engine, the gather tables and the ABI helpers (bhray_config.partition).
// nullptr: the partition described by cfg for `world` partitions is valid; otherwise what is wrong with it.
inline const char* partition_error(const bhray_config& cfg, uint32_t world) {
    if (world < 1) return "no partitions";
    if (cfg.partition == BHRAY_PARTITION_STRIPES) return cfg.stripe_rows < 1 ? "stripe_rows must be >= 1" : nullptr;
    if (cfg.partition != BHRAY_PARTITION_SLABS) return "unknown partition mode";
    if (world > BHRAY_MAX_DEVICES) return "BHRAY_PARTITION_SLABS: more partitions than slab_row0 holds";
    if (cfg.slab_row0[0] != 0 || cfg.slab_row0[world] != cfg.frame_h) return "BHRAY_PARTITION_SLABS: slab_row0 must start at 0 and end at frame_h";
    for (uint32_t p = 0; p < world; p++) if (cfg.slab_row0[p] > cfg.slab_row0[p + 1]) return "BHRAY_PARTITION_SLABS: slab_row0 must be non-decreasing";
    return nullptr;
}
// frame rows of partition `part`, increasing (a valid partition: partition_error)
inline std::vector<uint32_t> partition_row_list(const bhray_config& cfg, uint32_t world, uint32_t part) {
    std::vector<uint32_t> rows;
    if (cfg.partition == BHRAY_PARTITION_SLABS) {
        for (uint32_t r = cfg.slab_row0[part]; r < cfg.slab_row0[part + 1]; r++) rows.push_back(r);
    } else {
        for (uint32_t r = 0; r < cfg.frame_h; r++) if ((r / cfg.stripe_rows) % world == part) rows.push_back(r);
    }
    return rows;
}
// rows of level k-1 that the rows `fine` (sorted) of level k read (ray.wgsl:185-201), same binary32 arithmetic as the kernels' (bhray_api.hip)
std::vector<int32_t> coarse_rows_needed(const std::vector<int32_t>& fine, int h, int ph);

struct DevOptions {
    bool external_out = false;   // the ctx binds every frame's destination (send buffer / assembled frame): no own output buffers
    bool frame_rowmap = false;   // the destination is a whole frame_h x frame_w frame: row r of the frame lands in row r (root partition of a gather)
    uint32_t max_streams = 0;    // > 0: at most this many HIP streams for the frame slots (the slots beyond share them) - several engines on ONE physical GPU
                                 // (functional tests of the multi-GPU path on a one-GPU box) must stay within the device's hardware queues TOGETHER
};
}  // namespace bhray

const char* dev_last_error(const bhray_dev* c);            // c may be NULL: last create error of this thread
void dev_set_create_error(const char* msg);
int  dev_create(const bhray_config* cfg, const bhray::DevOptions& opt, bhray_dev** out);
void dev_destroy(bhray_dev* c);
int  dev_set_texture(bhray_dev* c, int slot, const uint8_t* rgba8, uint32_t w, uint32_t h);
int  dev_upload_model_uniform(bhray_dev* c, uint32_t model_index, const void* bytes, size_t size);
int  dev_upload_model(bhray_dev* c, uint32_t model_index, const bhray_model_desc* desc);
int  dev_set_model_transform(bhray_dev* c, uint32_t model_index, const float position[3], int32_t visible);
int  dev_set_uniforms(bhray_dev* c, const void* cam32, const void* bh132, const void* det32);
// another row partition (bhray_config.partition / stripe_rows / slab_row0 / row_rank / row_world) for the same frame; synchronises the engine
int  dev_set_partition(bhray_dev* c, uint32_t partition, uint32_t stripe_rows, const uint32_t* slab_row0, uint32_t row_rank, uint32_t row_world);
int  dev_render(bhray_dev* c);
int  dev_flush(bhray_dev* c);
int  dev_sync(bhray_dev* c);
uint32_t dev_local_rows(const bhray_dev* c);
int  dev_local_row_index(const bhray_dev* c, uint32_t i, uint32_t* frame_row);
int  dev_read_hdr(bhray_dev* c, float* dst, size_t pitch);
int  dev_read_hdr_async(bhray_dev* c, float* dst, size_t pitch, uint64_t* ticket);   // in stream order behind the last frame's kernels
int  dev_read_sky_async(bhray_dev* c, uint16_t* dst, size_t pitch, uint64_t* ticket);
int  dev_wait_read(bhray_dev* c, uint64_t ticket);
int  dev_read_level(bhray_dev* c, uint32_t level, float* dst, size_t pitch);
int  dev_hdr_device_ptr(bhray_dev* c, void** p, size_t* bytes);
int  dev_bind_output(bhray_dev* c, void* p, size_t bytes);     // destination of the NEXT dev_render only
int  dev_resolve_sky(bhray_dev* c);
int  dev_read_sky(bhray_dev* c, uint16_t* dst, size_t pitch);
int  dev_sky_device_ptr(bhray_dev* c, void** p, size_t* bytes);
int  dev_wait_event(bhray_dev* c, hipEvent_t ev);             // the next render's launches start after ev (ev is owned by the caller)
int  dev_next_stream(bhray_dev* c, void** s);
int  dev_signal_stream(bhray_dev* c, void* s);
int  dev_selftest(bhray_dev* c, uint64_t mismatches[3]);
int  dev_get_level_counters(bhray_dev* c, uint32_t level, bhray_counters* out);
int  dev_add_row_work(bhray_dev* c, uint32_t level, uint64_t* acc, uint32_t n);   // acc[y] += iterations of the last render's rays of level row y
int  dev_debug_read_queue(bhray_dev* c, uint32_t level, uint32_t* out, uint32_t cap, uint32_t* count);   // diagnostics only
int  dev_get_counters(bhray_dev* c, bhray_counters* out);
int  dev_get_timing(bhray_dev* c, bhray_timing* out);
// what the frames still held by the slots cost: integrator steps issued by the trace waves (mean per frame) and pixels visited by the classify launches
int  dev_get_work(bhray_dev* c, double* wave_steps_per_frame, double* classify_pixels_per_frame, uint32_t* frames, int* method);   // method: the integrator of the current uniforms (1 = RK)

// hooks for the gather (bhray_group.hip)
int  dev_device(const bhray_dev* c);
// Position (slot, index in the batch) the next dev_render will stage its frame at.  A staged batch of another kernel variant
// (integrator, mesh) than the current uniforms need is launched first, so the position is final.
int  dev_next_position(bhray_dev* c, int* slot, uint32_t* sub);
// The same state read without side effects: batches launched so far (the staging slot is that number modulo the slots), frames staged
// in that slot and the kernel variant they were staged for.  For the caller's mirror of the staging position (issue threads).
void dev_peek_position(const bhray_dev* c, uint64_t* batch_counter, uint32_t* pending, int* method, bool* models);
// True when launches were enqueued since the last call; reports the slot and the number of frames of the (last) launched batch.
bool dev_take_launched(bhray_dev* c, int* slot, uint32_t* frames);
int* dev_err_flag(bhray_dev* c);                              // the engine's device-side error word (checked by dev_sync)
hipStream_t dev_slot_stream(bhray_dev* c, int slot);
hipEvent_t  dev_slot_done(bhray_dev* c, int slot);            // recorded behind the slot's last launch
// sky.wgsl over an arbitrary RGBA32F image resident on this device, with this partition's sky texture
int  dev_launch_sky(bhray_dev* c, const void* src_rgba32f, void* dst_rgba16f, size_t npix, hipStream_t stream);
