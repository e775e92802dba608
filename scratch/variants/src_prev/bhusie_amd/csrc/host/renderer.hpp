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
    RayPipeline(const RayPipeline&) = delete;
    ~RayPipeline() { bhray_destroy(ctx_); }
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
    RayPipeline& ray_pipeline() { return ray_pipeline_; }
    void set_model(const Model& m) { ray_pipeline_.upload_model(m); ray_details.model_count = 1; }   // mod.rs:384
    void render(float dt) {                                                                     // mod.rs:378-420
        ray_details.time += dt;                                                                 // mod.rs:382
        ray_pipeline_.set_uniforms(camera.uniform(), black_hole.uniform(), ray_details);        // mod.rs:386-388
        const float materials[32] = {0};
        ray_pipeline_.set_materials(materials);                                                 // mod.rs:389
        ray_pipeline_.pass();
    }
private:
    RayPipeline ray_pipeline_;
};

}  // namespace bhusie
