// bhray_render — minimal C++ host program over renderer.hpp: renders one frame of the reference's default scene
// (camera (0,0,-19), hole at the origin, disk 2..10, R = 20; camera.rs:10-16, blackhole.rs:16-28) and writes the HDR frame
// as raw little-endian f32 RGBA (row 0 = top).  Usage:
//   bhray_render OUT.f32 [--rk] [--base W H] [--levels N] [--disk-size S] [--obj mesh.obj] [--devices 0,1,2,...]
// --devices: row-tile the frame over several GPUs from this one process (RCCL gather to the first one, inside libbhray).
// Textures: the disk texture comes from the reference's own generator (bhray_generate_disk_texture); the LUT and the sky
// are flat grey here (this program demonstrates the host surface, the tests use the seeded assets).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "renderer.hpp"

int main(int argc, char** argv) {
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
