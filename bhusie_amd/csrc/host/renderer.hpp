// renderer.hpp — C++ host-side mirror of the reference's ray-pass surface, over the C ABI (include/bhray.h).
//
// The reference host is Rust; with no Rust toolchain in this image the host side above the C ABI is C++.  Names and
// argument meaning follow the Rust:
//   RayDetails            src/renderer/pipelines/ray_pipeline.rs:3-14   (defaults: src/renderer/mod.rs:116-121)
//   Camera / BlackHole    src/scene/camera.rs:3-16, src/scene/blackhole.rs:3-28
//   Model, load_model     src/renderer/triangle.rs:65-259, src/renderer/model.rs:7-87
//   RayPipeline           src/renderer/pipelines/ray_pipeline.rs:28-310  {new, pass, output_view}
//   Renderer              src/renderer/mod.rs:58-207, 370-420            {new, render} restricted to the ray pass
// Header-only; link with -lbhray.  Errors become std::runtime_error (the reference panics).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/bhray.h"

namespace bhusie {

inline void check(int rc, const bhray_ctx* ctx = nullptr) {
    if (rc != BHRAY_OK) {
        const char* m = bhray_last_error(ctx);
        throw std::runtime_error(std::string("bhray: ") + ((m && *m) ? m : bhray_strerror(rc)));
    }
}

struct RayDetails : bhray_details {
    RayDetails() { bhray_details_default(this); }              // step_size 0.15, max_iterations 2000, threshold 0.02, Euler
};

struct Camera {                                                // camera.rs:10-16
    float position[3] = {0.0f, 0.0f, -19.0f};
    float forward[3] = {0.0f, 0.0f, 1.0f};
    float fov = 1.0f;
    bhray_camera_uniform uniform() const { bhray_camera_uniform u; bhray_camera_uniform_update(&u, position, forward, fov); return u; }
};

struct BlackHole : bhray_black_hole {                          // blackhole.rs:16-28
    BlackHole() { bhray_black_hole_default(this); }
    bhray_black_hole_uniform uniform() const { bhray_black_hole_uniform u; bhray_black_hole_uniform_update(&u, this); return u; }
};

class Model {                                                  // triangle.rs:65-259
public:
    Model() { check(bhray_model_new(&m_)); }
    explicit Model(const std::string& obj_path) { check(bhray_load_model(obj_path.c_str(), &m_)); }     // model::load_model
    Model(Model&& o) noexcept : m_(o.m_) { o.m_ = nullptr; }
    Model(const Model&) = delete;
    ~Model() { bhray_model_free(m_); }
    void add_vertex(const float p[4]) { check(bhray_model_add_vertex(m_, p)); }
    void add_normal(const float n[4]) { check(bhray_model_add_normal(m_, n)); }
    void add_triangle(const bhray_triangle& t) { check(bhray_model_add_triangle(m_, &t)); }
    void build_bvh() { check(bhray_model_build_bvh(m_)); }
    bhray_model_desc desc() const { bhray_model_desc d; check(bhray_model_desc_get(m_, &d)); return d; }
    const bhray_model* handle() const { return m_; }
private:
    bhray_model* m_ = nullptr;
};

// How a frame reaches a consumer that is not a HIP client of this GPU (the reference's SkyPipeline samples the ray output as a wgpu
// texture, ray_pipeline.rs:297-299, mod.rs:215) - the three forms of the drop-in shim of INTEGRATION.md §3:
enum class Handoff {
    Sync,        // finish() = bhray_read_hdr of the frame just rendered: the naive binding, one frame at a time
    AsyncHdr,    // pass() also enqueues the RGBA32F copy into pinned memory (bhray_read_hdr_async); finish() hands over the OLDEST frame in flight
    AsyncSky     // pass() also runs the sky pass in the library (sky.wgsl) and enqueues the copy of its RGBA16F image: half the bytes
};

// The chain of RayPipelines of one frame (mod.rs:181-207) as one object.
class RayPipeline {
public:
    // RayPipeline::new x levels: base resolution, multiplier, iterations as in mod.rs:177-205
    // frames_in_flight / frames_per_batch > 1: a host that renders ahead (offline sequences); pass() then only stages a frame
    // until a batch is full, output()/flush() launch what is staged
    RayPipeline(std::pair<uint32_t, uint32_t> base, uint32_t multiplier, uint32_t levels, int device = 0, uint32_t frames_in_flight = 1,
                uint32_t frames_per_batch = 1) {
        std::memset(&cfg_, 0, sizeof cfg_);
        check(bhray_ladder_from_base(base.first, base.second, multiplier, levels, &cfg_));
        cfg_.device = device; cfg_.frames_in_flight = frames_in_flight; cfg_.frames_per_batch = frames_per_batch;
        check(bhray_create(&cfg_, &ctx_));
    }
    // Row-tiled over several GPUs of the node, still ONE pipeline object driven from one thread (mod.rs:415-420): partition i
    // of the frame is rendered on devices[i]; pass() also enqueues the RCCL gather to devices[0] and the de-interleave, and
    // output() is the whole frame.
    RayPipeline(std::pair<uint32_t, uint32_t> base, uint32_t multiplier, uint32_t levels, const std::vector<int>& devices,
                uint32_t frames_in_flight = 1, uint32_t frames_per_batch = 1) {
        std::memset(&cfg_, 0, sizeof cfg_);
        check(bhray_ladder_from_base(base.first, base.second, multiplier, levels, &cfg_));
        if (devices.empty() || devices.size() > BHRAY_MAX_DEVICES) throw std::runtime_error("bhray: bad device list");
        cfg_.device_count = (uint32_t)devices.size();
        for (size_t i = 0; i < devices.size(); i++) cfg_.devices[i] = devices[i];
        cfg_.frames_in_flight = frames_in_flight; cfg_.frames_per_batch = frames_per_batch;
        check(bhray_create(&cfg_, &ctx_));
    }
    // The reference host's own configuration: the shipped 72x41 x3 x4 ladder, `frames_in_flight` = the swap chain's
    // desired_maximum_frame_latency (2, mod.rs:108), optionally BHRAY_F_TEMPORAL (consecutive frames of an interactive host are similar)
    RayPipeline(std::pair<uint32_t, uint32_t> base, uint32_t multiplier, uint32_t levels, int device, uint32_t frames_in_flight, uint32_t flags,
                uint32_t speculative_levels) {
        std::memset(&cfg_, 0, sizeof cfg_);
        check(bhray_ladder_from_base(base.first, base.second, multiplier, levels, &cfg_));
        cfg_.device = device; cfg_.frames_in_flight = frames_in_flight; cfg_.flags = flags; cfg_.speculative_levels = speculative_levels;
        check(bhray_create(&cfg_, &ctx_));
    }
    RayPipeline(const RayPipeline&) = delete;
    ~RayPipeline() {
        if (ctx_) (void)bhray_sync(ctx_);
        for (void* p : staging_) (void)bhray_host_free(p);
        bhray_destroy(ctx_);
    }
    // ---- the drop-in shim's per-frame pair: pass_handoff() where the reference dispatches its ray pipelines
    // (mod.rs:415-417), finish() where the host needs the pixels (before sky_pipeline.pass(), mod.rs:419)
    void enable_handoff(Handoff h, uint32_t frames_in_flight) {
        handoff_ = h; ring_ = frames_in_flight < 1 ? 1 : frames_in_flight;
        const size_t bytes = (size_t)cfg_.frame_w * cfg_.frame_h * (h == Handoff::AsyncSky ? 8 : 16);
        for (uint32_t i = 0; i < ring_; i++) { void* p = nullptr; check(bhray_host_alloc(bytes, &p)); staging_.push_back(p); }
        tickets_.assign(ring_, 0); pending_.assign(ring_, false);
    }
    void pass_handoff() {
        const uint32_t k = (uint32_t)(frame_ % ring_);
        if (pending_[k]) { check(bhray_wait_read(ctx_, tickets_[k]), ctx_); pending_[k] = false; }   // (finish() already took it unless the host skipped a frame)
        pass();
        if (handoff_ == Handoff::AsyncHdr) {
            check(bhray_read_hdr_async(ctx_, (float*)staging_[k], (size_t)cfg_.frame_w * 16, &tickets_[k]), ctx_); pending_[k] = true;
        } else if (handoff_ == Handoff::AsyncSky) {
            resolve_sky();
            check(bhray_read_sky_async(ctx_, (uint16_t*)staging_[k], (size_t)cfg_.frame_w * 8, &tickets_[k]), ctx_); pending_[k] = true;
        }
        frame_++;
    }
    // The pixels the host uploads into its texture (queue.write_texture): Sync - the frame just rendered (RGBA32F); Async* - the OLDEST
    // frame in flight, i.e. the frame enqueued ring-1 passes ago (nullptr while the pipeline fills): with a ring of 2 the host shows frame
    // k-1 while frame k renders, which is what its swap chain's frame latency of 2 does to every frame anyway.
    const void* finish() {
        if (handoff_ == Handoff::Sync) {
            check(bhray_read_hdr(ctx_, (float*)staging_[0], (size_t)cfg_.frame_w * 16), ctx_);
            return staging_[0];
        }
        if (frame_ < ring_) return nullptr;
        const uint32_t k = (uint32_t)(frame_ % ring_);             // the slot the NEXT pass reuses = the oldest frame in flight
        if (pending_[k]) { check(bhray_wait_read(ctx_, tickets_[k]), ctx_); pending_[k] = false; }
        return staging_[k];
    }
    // after the last pass: the k-th oldest frame still in flight (k = 1 .. ring-1), already drained - what the next finish() calls would hand over
    const void* finish_pending(uint32_t k) {
        if (handoff_ == Handoff::Sync || frame_ < k || k >= ring_) return nullptr;
        return staging_[(size_t)((frame_ + k) % ring_)];
    }
    void drain() { for (uint32_t k = 0; k < ring_; k++) if (pending_.size() > k && pending_[k]) { check(bhray_wait_read(ctx_, tickets_[k]), ctx_); pending_[k] = false; } }
    std::pair<uint32_t, uint32_t> resolution() const { return {cfg_.frame_w, cfg_.frame_h}; }
    void set_texture(int slot, const uint8_t* rgba8, uint32_t w, uint32_t h) { check(bhray_set_texture(ctx_, slot, rgba8, w, h), ctx_); }
    void upload_model(const Model& m) { bhray_model_desc d = m.desc(); check(bhray_upload_model(ctx_, 0, &d), ctx_); }
    void set_materials(const void* material_uniforms_128) { check(bhray_set_materials(ctx_, material_uniforms_128, 128), ctx_); }   // mod.rs:389 (ignored by the shader)
    void set_uniforms(const bhray_camera_uniform& c, const bhray_black_hole_uniform& b, const bhray_details& d) { check(bhray_set_uniforms(ctx_, &c, &b, &d), ctx_); }
    void pass() { check(bhray_render(ctx_), ctx_); }                                            // ray_pipeline.rs:301-309
    void flush() { check(bhray_flush(ctx_), ctx_); }
    void resolve_sky() { check(bhray_resolve_sky(ctx_), ctx_); }                                // sky_pipeline.rs pass
    std::vector<float> output() {                                                               // output_view + read-back
        std::vector<float> out((size_t)cfg_.frame_w * cfg_.frame_h * 4);
        check(bhray_read_hdr(ctx_, out.data(), (size_t)cfg_.frame_w * 16), ctx_);
        return out;
    }
    bhray_ctx* ctx() { return ctx_; }
private:
    bhray_config cfg_;
    bhray_ctx* ctx_ = nullptr;
    Handoff handoff_ = Handoff::Sync;
    uint32_t ring_ = 1;
    uint64_t frame_ = 0;
    std::vector<void*> staging_;
    std::vector<uint64_t> tickets_;
    std::vector<bool> pending_;
};

// Renderer::{new, render} restricted to the ray pass.
class Renderer {
public:
    Camera camera;
    BlackHole black_hole;
    RayDetails ray_details;
    explicit Renderer(int device = 0) : ray_pipeline_({72, 41}, 3, 4, device) {}               // mod.rs:177-179
    Renderer(std::pair<uint32_t, uint32_t> base, uint32_t multiplier, uint32_t levels, int device = 0) : ray_pipeline_(base, multiplier, levels, device) {}
    Renderer(std::pair<uint32_t, uint32_t> base, uint32_t multiplier, uint32_t levels, const std::vector<int>& devices) : ray_pipeline_(base, multiplier, levels, devices) {}
    // the drop-in host: frames_in_flight = the swap chain's frame latency, a hand-off form, optionally temporal speculation
    Renderer(std::pair<uint32_t, uint32_t> base, uint32_t multiplier, uint32_t levels, int device, uint32_t frames_in_flight, Handoff h, bool temporal)
        : ray_pipeline_(base, multiplier, levels, device, frames_in_flight, temporal ? (uint32_t)BHRAY_F_TEMPORAL : 0u, temporal ? 0u : (levels >= 4 ? 3u : (levels >= 3 ? 2u : 0u))) {       // a host that shows every frame wants the shortest chain: 3 speculative levels (1.08 against 1.21 ms with 2)
        ray_pipeline_.enable_handoff(h, frames_in_flight);
    }
    RayPipeline& ray_pipeline() { return ray_pipeline_; }
    void set_model(const Model& m) { ray_pipeline_.upload_model(m); ray_details.model_count = 1; }   // mod.rs:384
    void render(float dt) {                                                                     // mod.rs:378-420
        ray_details.time += dt;                                                                 // mod.rs:382
        ray_pipeline_.set_uniforms(camera.uniform(), black_hole.uniform(), ray_details);        // mod.rs:386-388
        const float materials[32] = {0};
        ray_pipeline_.set_materials(materials);                                                 // mod.rs:389
        ray_pipeline_.pass();
    }
    // the same frame through the drop-in shim: update + pass_handoff (mod.rs:378-417), then finish() where the sky pass would start
    const void* render_handoff(float dt) {
        ray_details.time += dt;
        ray_pipeline_.set_uniforms(camera.uniform(), black_hole.uniform(), ray_details);
        const float materials[32] = {0};
        ray_pipeline_.set_materials(materials);
        ray_pipeline_.pass_handoff();
        return ray_pipeline_.finish();
    }
private:
    RayPipeline ray_pipeline_;
};

}  // namespace bhusie
