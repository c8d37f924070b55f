// Drives shim/gsplat_gdextension.cpp — the GDExtension class GDScript would see instead of
// util/gaussian_splatting_rasterizer.gd — compiled against the STAND-IN godot-cpp declarations of
// tests/native/godot_cpp_standin (godot-cpp itself is not in the image): library entry point, class registration and
// the bound method names, then a session the way main.gd runs one: create from PlyFile.vertices, update_camera_matrices
// from a Camera3D, rasterize while the loader thread uploads, the `loaded` signal, the final frame as the byte array
// RenderingDevice.texture_update takes, get_splat_position, debug_info.
//     usage: gdext_driver rows.bin n w h out_prefix pick_x pick_y
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../../shim/gsplat_gdextension.cpp"

static bool has(const std::vector<std::string> &v, const char *name) {
    for (const std::string &s : v)
        if (s == name) return true;
    return false;
}

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: %s rows.bin n w h out_prefix pick_x pick_y\n", argv[0]); return 2; }
    const int n = atoi(argv[2]), w = atoi(argv[3]), h = atoi(argv[4]);
    const char *prefix = argv[5];
    PackedFloat32Array vertices;
    vertices.resize((int64_t)n * 62);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(vertices.ptrw(), sizeof(float), (size_t)vertices.size(), f) != (size_t)vertices.size()) return 2;
    fclose(f);

    GDExtensionInitialization init;
    if (!gsplat_library_init(nullptr, nullptr, &init)) return 1;
    const StandinRegistry &reg = StandinRegistry::get();
    if (init.minimum_initialization_level != (int)MODULE_INITIALIZATION_LEVEL_SCENE || !has(reg.classes, "GsplatBridge")) return 1;
    // the reference class's surface (gaussian_splatting_rasterizer.gd:26,122,162,175; main.gd:93-119)
    for (const char *m : {"create", "set_texture_size", "update_camera_matrices", "rasterize", "rasterize_pipelined", "set_readback_rgb",
                          "get_splat_position", "num_splats_loaded", "is_loaded", "debug_info"})
        if (!has(reg.methods, m)) { fprintf(stderr, "method %s is not bound\n", m); return 1; }
    if (!has(reg.signals, "loaded")) return 1;

    GsplatBridge bridge;
    bridge.create(vertices, w, h);
    Camera3D camera;                                  // SURVEY.md §8(d) camera: identity basis at (0, 0, 5)
    camera.standin_transform.origin = Vector3(0, 0, 5);
    if (!bridge.update_camera_matrices(&camera, Basis())) return 1;   // first call: everything changed
    Time::get_singleton()->standin_ticks_msec = 500;
    int frames_while_loading = 0;
    while (!bridge.is_loaded()) {
        if (bridge.rasterize(1.0f, false).size() != (int64_t)w * h * 16) return 1;
        ++frames_while_loading;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (bridge.num_splats_loaded() != n) return 1;
    Time::get_singleton()->standin_ticks_msec = 1000 * 1000;          // every fade-in long over
    PackedByteArray bytes = bridge.rasterize(1.0f, false);
    if (bytes.size() != (int64_t)w * h * 16) return 1;
    char path[1024];
    snprintf(path, sizeof path, "%s_frame.bin", prefix);
    f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(bytes.ptr(), 1, (size_t)bytes.size(), f);
    fclose(f);
    // the pipelined form: call k + 1 hands out frame k (INTEGRATION.md route 2)
    if (bridge.rasterize_pipelined(1.0f, false).size() != 0) { fprintf(stderr, "first pipelined call returned a frame\n"); return 1; }
    PackedByteArray lagged = bridge.rasterize_pipelined(1.0f, false);
    if (lagged.size() != (int64_t)w * h * 16) return 1;
    snprintf(path, sizeof path, "%s_pipelined.bin", prefix);
    f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(lagged.ptr(), 1, (size_t)lagged.size(), f);
    fclose(f);
    const Vector3 p = bridge.get_splat_position(Vector2((real_t)atof(argv[6]), (real_t)atof(argv[7])));
    Dictionary info = bridge.debug_info();
    printf("frames_while_loading %d\n", frames_while_loading);
    printf("loaded_signal %d\n", (int)has(reg.emitted, "loaded"));
    printf("pick %.9g %.9g %.9g\n", p.x, p.y, p.z);
    printf("rendered_splats %lld\n", (long long)info["rendered_splats"].i);
    printf("info_keys %lld\n", (long long)info.size());
    return 0;
}
