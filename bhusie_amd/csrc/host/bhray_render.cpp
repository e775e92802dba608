// bhray_render — minimal C++ host program over renderer.hpp: renders one frame of the reference's default scene
// (camera (0,0,-19), hole at the origin, disk 2..10, R = 20; camera.rs:10-16, blackhole.rs:16-28) and writes the HDR frame
// as raw little-endian f32 RGBA (row 0 = top).  Usage:
//   bhray_render OUT.f32 [--rk] [--base W H] [--levels N] [--disk-size S] [--obj mesh.obj] [--devices 0,1,2,...]
// --devices: row-tile the frame over several GPUs from this one process (RCCL gather to the first one, inside libbhray).
//   bhray_render --handoff sync|hdr|sky|sky1|sky-temporal FRAMES OUT.bin [--rk] [--base W H] [--levels N]
//   bhray_render --dropin FRAMES [--rk] [--base W H] [--levels N]
// --handoff: FRAMES frames of the same scene (time += 1/60 per frame) through Renderer::render_handoff in the given form, every DELIVERED frame
// appended to OUT.bin in delivery order (RGBA32F for sync / hdr, RGBA16F for sky) - the tests compare them with the frames the Python host renders.
// --dropin: times the drop-in shim of INTEGRATION.md §3 the way the reference host drives it - one thread, `time += dt` every frame
// (mod.rs:382), every frame handed to host memory - in its three hand-off forms, and prints one JSON object (bench.py embeds it as `dropin`).
// Textures: the disk texture comes from the reference's own generator (bhray_generate_disk_texture); the LUT and the sky
// are flat grey here (this program demonstrates the host surface, the tests use the seeded assets).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "renderer.hpp"

namespace {
// seeded stand-ins of the image assets (color.png is an image, sky.png is missing from the reference checkout): a smooth LUT and a
// noisy sky of the shipped sizes, so that the texture taps of the timed frames touch real memory
void fill_textures(bhusie::RayPipeline& rp, uint32_t disk) {
    std::vector<uint8_t> d((size_t)disk * disk * 4);
    bhusie::check(bhray_generate_disk_texture(disk, d.data()));
    rp.set_texture(BHRAY_TEX_DISK, d.data(), disk, disk);
    std::vector<uint8_t> lut(256 * 256 * 4);
    for (int y = 0; y < 256; y++) for (int x = 0; x < 256; x++) {
        uint8_t* p = &lut[((size_t)y * 256 + x) * 4];
        p[0] = (uint8_t)(255 - x / 2); p[1] = (uint8_t)(80 + x / 2); p[2] = (uint8_t)x; p[3] = 255;
    }
    rp.set_texture(BHRAY_TEX_TEMP_LUT, lut.data(), 256, 256);
    std::vector<uint8_t> sky((size_t)4096 * 2048 * 4);
    uint32_t h = 2463534242u;
    for (size_t i = 0; i < sky.size(); i += 4) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; sky[i] = (uint8_t)(h & 63); sky[i + 1] = (uint8_t)((h >> 8) & 63); sky[i + 2] = (uint8_t)((h >> 16) & 127); sky[i + 3] = 255; }
    rp.set_texture(BHRAY_TEX_SKY, sky.data(), 4096, 2048);
}

struct Leg { const char* name; bhusie::Handoff h; uint32_t in_flight; bool temporal; bool orbit; };

// one leg: `frames` frames through Renderer::render_handoff, host wall clock around them (after a warm-up of 8 frames)
double run_leg(const Leg& L, uint32_t bw, uint32_t bh, uint32_t levels, bool rk, int frames, uint64_t* checksum, uint32_t* w, uint32_t* h) {
    bhusie::Renderer r({bw, bh}, 3, levels, 0, L.in_flight, L.h, L.temporal);
    fill_textures(r.ray_pipeline(), 1000);
    r.ray_details.integration_method = rk ? 1 : 0;
    auto res = r.ray_pipeline().resolution(); *w = res.first; *h = res.second;
    const size_t words = (size_t)res.first * res.second * (L.h == bhusie::Handoff::AsyncSky ? 2 : 4);
    auto frame = [&](int i) {
        if (L.orbit) {                                   // a camera orbiting the hole, 0.02 rad per frame, looking at it (the UI's drag speed)
            const float a = 0.02f * (float)i;
            r.camera.position[0] = 19.0f * std::sin(a); r.camera.position[1] = 0.0f; r.camera.position[2] = -19.0f * std::cos(a);
            r.camera.forward[0] = -std::sin(a); r.camera.forward[1] = 0.0f; r.camera.forward[2] = std::cos(a);
        }
        const uint32_t* px = (const uint32_t*)r.render_handoff(1.0f / 60.0f);
        if (px) *checksum += px[(size_t)i * 7919 % words];          // the host touches what it was handed
    };
    for (int i = 0; i < 8; i++) frame(i);
    bhusie::check(bhray_sync(r.ray_pipeline().ctx()), r.ray_pipeline().ctx());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; i++) frame(8 + i);
    r.ray_pipeline().drain();
    bhusie::check(bhray_sync(r.ray_pipeline().ctx()), r.ray_pipeline().ctx());
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

int main(int argc, char** argv) {
    if (argc >= 5 && !std::strcmp(argv[1], "--handoff")) {
        const char* form = argv[2]; const int frames = std::atoi(argv[3]); const char* out_path = argv[4];
        uint32_t bw = 24, bh = 14, levels = 3; bool rk = false;
        for (int i = 5; i < argc; i++) {
            if (!std::strcmp(argv[i], "--rk")) rk = true;
            else if (!std::strcmp(argv[i], "--base") && i + 2 < argc) { bw = (uint32_t)std::atoi(argv[++i]); bh = (uint32_t)std::atoi(argv[++i]); }
            else if (!std::strcmp(argv[i], "--levels") && i + 1 < argc) levels = (uint32_t)std::atoi(argv[++i]);
        }
        try {
            const bool sky = !std::strncmp(form, "sky", 3), temporal = !std::strcmp(form, "sky-temporal");
            const bhusie::Handoff h = !std::strcmp(form, "sync") ? bhusie::Handoff::Sync : (sky ? bhusie::Handoff::AsyncSky : bhusie::Handoff::AsyncHdr);
            const uint32_t in_flight = (h == bhusie::Handoff::Sync || !std::strcmp(form, "sky1")) ? 1u : 2u;   // sky1: the sky form, one frame in flight
            bhusie::Renderer r({bw, bh}, 3, levels, 0, in_flight, h, temporal);
            fill_textures(r.ray_pipeline(), 128);
            r.ray_details.integration_method = rk ? 1 : 0;
            auto res = r.ray_pipeline().resolution();
            const size_t bytes = (size_t)res.first * res.second * (sky ? 8 : 16);
            FILE* f = std::fopen(out_path, "wb");
            if (!f) { std::perror(out_path); return 1; }
            int delivered = 0;
            for (int i = 0; i < frames; i++) {
                const void* px = r.render_handoff(1.0f / 60.0f);
                if (px) { std::fwrite(px, 1, bytes, f); delivered++; }
            }
            // the frames still in flight: one more finish() per frame (the host would show them on its next frames)
            for (uint32_t k = 1; k < in_flight; k++) {
                r.ray_pipeline().drain();
                const void* px = r.ray_pipeline().finish_pending(k);
                if (px) { std::fwrite(px, 1, bytes, f); delivered++; }
            }
            std::fclose(f);
            std::printf("%ux%u %d\n", res.first, res.second, delivered);
        } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
        return 0;
    }
    if (argc >= 3 && !std::strcmp(argv[1], "--dropin")) {
        const int frames = std::atoi(argv[2]);
        uint32_t bw = 72, bh = 41, levels = 4; bool rk = false;
        for (int i = 3; i < argc; i++) {
            if (!std::strcmp(argv[i], "--rk")) rk = true;
            else if (!std::strcmp(argv[i], "--base") && i + 2 < argc) { bw = (uint32_t)std::atoi(argv[++i]); bh = (uint32_t)std::atoi(argv[++i]); }
            else if (!std::strcmp(argv[i], "--levels") && i + 1 < argc) levels = (uint32_t)std::atoi(argv[++i]);
        }
        const Leg legs[] = {
            {"sync_read_hdr", bhusie::Handoff::Sync, 1, false, false},
            {"async_rgba32f_2_in_flight", bhusie::Handoff::AsyncHdr, 2, false, false},
            {"async_sky_rgba16f_2_in_flight", bhusie::Handoff::AsyncSky, 2, false, false},
            {"async_sky_rgba16f_2_in_flight_temporal", bhusie::Handoff::AsyncSky, 2, true, false},
            {"async_sky_rgba16f_2_in_flight_temporal_orbit", bhusie::Handoff::AsyncSky, 2, true, true},
            {"async_sky_rgba16f_2_in_flight_orbit", bhusie::Handoff::AsyncSky, 2, false, true},
        };
        try {
            std::printf("{\"frames\": %d, \"integrator\": \"%s\", \"legs\": {", frames, rk ? "rk" : "euler");
            uint64_t checksum = 0; uint32_t w = 0, h = 0; bool first = true;
            for (const Leg& L : legs) {
                if (L.temporal && levels > BHRAY_MAX_SPEC_LEVELS) continue;
                const double s = run_leg(L, bw, bh, levels, rk, frames, &checksum, &w, &h);
                std::printf("%s\"%s\": {\"ms_per_frame\": %.5f, \"mrays_per_s\": %.2f, \"frames_in_flight\": %u}", first ? "" : ", ", L.name, s / frames * 1e3,
                            (double)w * h * frames / s / 1e6, L.in_flight);
                first = false;
            }
            std::printf("}, \"frame\": [%u, %u], \"checksum\": %llu}\n", w, h, (unsigned long long)checksum);
        } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
        return 0;
    }
    if (argc < 2) { std::fprintf(stderr, "usage: %s OUT.f32 [--rk] [--base W H] [--levels N] [--disk-size S] [--obj mesh.obj] [--devices 0,1,...]\n", argv[0]); return 2; }
    uint32_t bw = 72, bh = 41, levels = 4, disk = 256;
    bool rk = false;
    const char* obj = nullptr;
    std::vector<int> devices;
    for (int i = 2; i < argc; i++) {
        if (!std::strcmp(argv[i], "--rk")) rk = true;
        else if (!std::strcmp(argv[i], "--base") && i + 2 < argc) { bw = (uint32_t)std::atoi(argv[++i]); bh = (uint32_t)std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--levels") && i + 1 < argc) levels = (uint32_t)std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--disk-size") && i + 1 < argc) disk = (uint32_t)std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--obj") && i + 1 < argc) obj = argv[++i];
        else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) { for (const char* p = argv[++i]; *p; ) { devices.push_back(std::atoi(p)); while (*p && *p != ',') p++; if (*p) p++; } }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    try {
        std::unique_ptr<bhusie::Renderer> rp(devices.empty() ? new bhusie::Renderer({bw, bh}, 3, levels) : new bhusie::Renderer({bw, bh}, 3, levels, devices));
        bhusie::Renderer& r = *rp;
        std::vector<uint8_t> d((size_t)disk * disk * 4);
        bhusie::check(bhray_generate_disk_texture(disk, d.data()));
        r.ray_pipeline().set_texture(BHRAY_TEX_DISK, d.data(), disk, disk);
        const uint8_t grey[4] = {160, 160, 160, 255};
        r.ray_pipeline().set_texture(BHRAY_TEX_TEMP_LUT, grey, 1, 1);
        r.ray_pipeline().set_texture(BHRAY_TEX_SKY, grey, 1, 1);
        std::unique_ptr<bhusie::Model> model;
        if (obj) { model.reset(new bhusie::Model(obj)); r.set_model(*model); }
        r.ray_details.integration_method = rk ? 1 : 0;
        r.render(0.0f);
        const std::vector<float> out = r.ray_pipeline().output();
        auto res = r.ray_pipeline().resolution();
        FILE* f = std::fopen(argv[1], "wb");
        if (!f) { std::perror(argv[1]); return 1; }
        std::fwrite(out.data(), sizeof(float), out.size(), f);
        std::fclose(f);
        std::printf("%ux%u\n", res.first, res.second);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
