// Drives shim/gsplat_bridge.cpp — the Godot-free core of the GDExtension shim, i.e. the state machine of
// util/gaussian_splatting_rasterizer.gd over the C ABI — the way main.gd drives the reference class: construct from the
// .ply rows, init_gpu (loader thread uploads ~1000 chunks), update_camera_matrices, rasterize, get_splat_position,
// texture_size setter, rasterize again; then the pipelined read-back form.  Results go to files the test compares
// with the oracle.   usage: bridge_driver rows.bin n w h w2 h2 out_prefix pick_x pick_y
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../shim/gsplat_bridge.h"

static int dump(const char *prefix, const char *name, const float *data, size_t floats) {
    char path[1024];
    snprintf(path, sizeof path, "%s_%s.bin", prefix, name);
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(data, sizeof(float), floats, f);
    fclose(f);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 10) { fprintf(stderr, "usage: %s rows.bin n w h w2 h2 out_prefix pick_x pick_y\n", argv[0]); return 2; }
    const uint32_t n = (uint32_t)atol(argv[2]), w = atoi(argv[3]), h = atoi(argv[4]), w2 = atoi(argv[5]), h2 = atoi(argv[6]);
    const char *prefix = argv[7];
    const float pick_x = (float)atof(argv[8]), pick_y = (float)atof(argv[9]);
    std::vector<float> rows((size_t)n * GSPLAT_PLY_ROW_FLOATS);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(rows.data(), sizeof(float), rows.size(), f) != rows.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    gsplat_shim::Bridge b(rows.data(), n, w, h);
#define OK(expr) do { int rc_ = (expr); if (rc_ != GSPLAT_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, b.last_error().c_str()); return 1; } } while (0)
    OK(b.init_gpu(0.0));
    gsplat_shim::CameraState cam;
    memset(&cam, 0, sizeof cam);
    cam.xform[0] = cam.xform[4] = cam.xform[8] = 1.0f;          // identity basis, origin (0, 0, 5): SURVEY.md §8(d) camera
    cam.xform[11] = 5.0f;
    cam.basis_override[0] = cam.basis_override[4] = cam.basis_override[8] = 1.0f;
    cam.fovy_degrees = 75.0f; cam.z_near = 0.05f; cam.z_far = 4000.0f;
    b.update_camera_matrices(cam);
    // frames while the loader thread is still uploading (main.gd:146-152 rasterizes while is_loaded is false)
    int frames_while_loading = 0;
    while (!b.is_loaded.load()) {
        OK(b.rasterize(0.5));
        ++frames_while_loading;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (b.num_splats_loaded.load() != n) { fprintf(stderr, "loaded %u of %u\n", b.num_splats_loaded.load(), n); return 1; }
    const double steady = 1000.0;                                // every fade-in long over: tf = tfl = 1
    OK(b.rasterize(steady));
    if (dump(prefix, "frame", b.rgba().data(), b.rgba().size())) return 1;
    float xyz[3] = {0, 0, 0};
    bool hit = false;
    OK(b.get_splat_position(pick_x, pick_y, steady, xyz, &hit));
    gsplat_stats st;
    OK(b.debug_info(&st));
    printf("frames_while_loading %d\nhit %d\npick %.9g %.9g %.9g\nvisible %llu emitted %llu ms_total %.4f\n", frames_while_loading,
           hit ? 1 : 0, xyz[0], xyz[1], xyz[2], (unsigned long long)st.num_visible, (unsigned long long)st.num_emitted, st.ms_total);
    // texture_size setter (:26-48), then the camera again (the aspect changed) and a frame at the new size
    OK(b.set_texture_size(w2, h2));
    b.update_camera_matrices(cam);
    OK(b.rasterize(steady));
    if (dump(prefix, "resized", b.rgba().data(), b.rgba().size())) return 1;
    // the pipelined hand-off: frame k is returned by call k + 1
    const float *prev = nullptr;
    OK(b.rasterize_pipelined(steady, &prev));
    if (prev != nullptr) { fprintf(stderr, "first pipelined call returned a frame\n"); return 1; }
    OK(b.rasterize_pipelined(steady, &prev));
    if (prev == nullptr || dump(prefix, "pipelined", prev, (size_t)b.width() * b.height() * 4)) return 1;
    // ... and a second rasterizer whose pipelined frames are the colour channels alone (readback_rgb: 12 B/px over PCIe)
    {
        gsplat_shim::Bridge c(rows.data(), n, w2, h2);
#define OKC(expr) do { int rc_ = (expr); if (rc_ != GSPLAT_OK) { fprintf(stderr, "%s -> %d: %s\n", #expr, rc_, c.last_error().c_str()); return 1; } } while (0)
        c.readback_rgb = true;
        OKC(c.init_gpu(0.0));
        c.update_camera_matrices(cam);
        while (!c.is_loaded.load()) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        const float *rgb = nullptr;
        OKC(c.rasterize_pipelined(steady, &rgb));
        OKC(c.rasterize_pipelined(steady, &rgb));
        if (rgb == nullptr || c.readback_channels() != 3 || dump(prefix, "pipelined_rgb", rgb, (size_t)c.width() * c.height() * 3)) return 1;
    }
    puts("bridge_driver ok");
    return 0;
}
