// gsplat_gdextension.cpp — GDExtension class `GsplatBridge` (godot-cpp 4.3) around gsplat_shim::Bridge: what GDScript
// sees instead of the six compute pipelines of util/gaussian_splatting_rasterizer.gd.  godot-cpp is not in this image:
// the translation unit compiles to nothing without it (the tests compile and run it against stand-in declarations of the
// godot-cpp names used here, tests/native/godot_cpp_standin); build it in a godot-cpp tree with
// scons target=template_release  and link libgsplat_hip.so.
#if __has_include(<godot_cpp/classes/ref_counted.hpp>)
#include <godot_cpp/classes/camera3d.hpp>
#include <godot_cpp/classes/ref_counted.hpp>
#include <godot_cpp/classes/time.hpp>
#include <godot_cpp/core/class_db.hpp>
#include <godot_cpp/godot.hpp>
#include <godot_cpp/variant/packed_byte_array.hpp>
#include <godot_cpp/variant/packed_float32_array.hpp>

#include <cmath>
#include <cstring>
#include <memory>

#include "gsplat_bridge.h"

using namespace godot;

class GsplatBridge : public RefCounted {
    GDCLASS(GsplatBridge, RefCounted)
    std::unique_ptr<gsplat_shim::Bridge> core;
    PackedFloat32Array rows;  // keeps PlyFile.vertices alive for the loader thread

    static double now() { return Time::get_singleton()->get_ticks_msec() * 1e-3; }

protected:
    static void _bind_methods() {
        ClassDB::bind_method(D_METHOD("create", "ply_vertices", "width", "height"), &GsplatBridge::create);
        ClassDB::bind_method(D_METHOD("set_texture_size", "width", "height"), &GsplatBridge::set_texture_size);
        ClassDB::bind_method(D_METHOD("update_camera_matrices", "camera", "basis_override"), &GsplatBridge::update_camera_matrices);
        ClassDB::bind_method(D_METHOD("rasterize", "model_scale", "heatmap"), &GsplatBridge::rasterize);
        ClassDB::bind_method(D_METHOD("rasterize_pipelined", "model_scale", "heatmap"), &GsplatBridge::rasterize_pipelined);
        ClassDB::bind_method(D_METHOD("set_readback_rgb", "enabled"), &GsplatBridge::set_readback_rgb);
        ClassDB::bind_method(D_METHOD("get_splat_position", "screen_pos"), &GsplatBridge::get_splat_position);
        ClassDB::bind_method(D_METHOD("num_splats_loaded"), &GsplatBridge::num_splats_loaded);
        ClassDB::bind_method(D_METHOD("is_loaded"), &GsplatBridge::is_loaded);
        ClassDB::bind_method(D_METHOD("debug_info"), &GsplatBridge::debug_info);
        ADD_SIGNAL(MethodInfo("loaded"));
    }

public:
    void create(const PackedFloat32Array &ply_vertices, int width, int height) {   // _init + init_gpu
        rows = ply_vertices;
        core = std::make_unique<gsplat_shim::Bridge>(rows.ptr(), (uint32_t)(rows.size() / 62), (uint32_t)width, (uint32_t)height);
    }
    void set_texture_size(int w, int h) { core->set_texture_size((uint32_t)w, (uint32_t)h); }
    bool update_camera_matrices(Camera3D *camera, const Basis &basis_override) {
        const Transform3D t = camera->get_global_transform();
        gsplat_shim::CameraState cs;
        const Vector3 cols[4] = {t.basis.get_column(0), t.basis.get_column(1), t.basis.get_column(2), t.origin};
        for (int k = 0; k < 4; ++k) { cs.xform[3 * k] = cols[k].x; cs.xform[3 * k + 1] = cols[k].y; cs.xform[3 * k + 2] = cols[k].z; }
        for (int c = 0; c < 3; ++c) {
            const Vector3 col = basis_override.get_column(c);
            cs.basis_override[3 * c] = col.x; cs.basis_override[3 * c + 1] = col.y; cs.basis_override[3 * c + 2] = col.z;
        }
        cs.fovy_degrees = camera->get_fov(); cs.z_near = camera->get_near(); cs.z_far = camera->get_far();
        return core->update_camera_matrices(cs);
    }
    PackedByteArray rasterize(float model_scale, bool heatmap) {   // -> RenderingDevice.texture_update(render_texture, 0, bytes)
        core->model_scale = model_scale;
        core->should_enable_heatmap = heatmap;
        PackedByteArray out;
        if (core->rasterize(now()) != GSPLAT_OK) return out;
        if (core->is_loaded.exchange(false)) { emit_signal("loaded"); core->is_loaded.store(true); }
        out.resize((int64_t)core->rgba().size() * 4);
        memcpy(out.ptrw(), core->rgba().data(), core->rgba().size() * 4);
        return out;
    }
    // The same frame through the library's pinned read-back ring: returns at once with the PREVIOUS call's frame (empty on
    // the first call) — the copy of frame k overlaps the kernels of frame k + 1 (route 2).  With
    // set_readback_rgb(true) before the first frame the bytes are RGB32F (Image.FORMAT_RGBF, 12 per pixel).
    void set_readback_rgb(bool enabled) { core->readback_rgb = enabled; }
    PackedByteArray rasterize_pipelined(float model_scale, bool heatmap) {
        core->model_scale = model_scale;
        core->should_enable_heatmap = heatmap;
        PackedByteArray out;
        const float *prev = nullptr;
        if (core->rasterize_pipelined(now(), &prev) != GSPLAT_OK) return out;
        if (core->is_loaded.exchange(false)) { emit_signal("loaded"); core->is_loaded.store(true); }
        if (!prev) return out;
        const int64_t bytes = (int64_t)core->width() * core->height() * core->readback_channels() * 4;
        out.resize(bytes);
        memcpy(out.ptrw(), prev, (size_t)bytes);
        return out;
    }
    Vector3 get_splat_position(const Vector2 &p) {
        float xyz[3]; bool hit = false;
        if (core->get_splat_position(p.x, p.y, now(), xyz, &hit) != GSPLAT_OK || !hit) return Vector3(INFINITY, INFINITY, INFINITY);
        return Vector3(xyz[0], xyz[1], xyz[2]);
    }
    int num_splats_loaded() const { return (int)core->num_splats_loaded.load(); }
    bool is_loaded() const { return core->is_loaded.load(); }
    Dictionary debug_info() const {   // main.gd:93-119
        gsplat_stats st; Dictionary d;
        if (core->debug_info(&st) != GSPLAT_OK) return d;
        d["rendered_splats"] = (int64_t)st.num_emitted; d["overflow"] = st.overflow != 0; d["vram_bytes"] = (int64_t)st.bytes_allocated;
        d["ms_projection"] = st.ms_projection; d["ms_sort"] = st.ms_sort; d["ms_boundaries"] = st.ms_boundaries; d["ms_render"] = st.ms_render;
        return d;
    }
};

void initialize_gsplat(ModuleInitializationLevel level) { if (level == MODULE_INITIALIZATION_LEVEL_SCENE) ClassDB::register_class<GsplatBridge>(); }
void uninitialize_gsplat(ModuleInitializationLevel) {}

extern "C" GDExtensionBool GDE_EXPORT gsplat_library_init(GDExtensionInterfaceGetProcAddress get_proc, GDExtensionClassLibraryPtr lib,
                                                          GDExtensionInitialization *init) {
    GDExtensionBinding::InitObject obj(get_proc, lib, init);
    obj.register_initializer(initialize_gsplat);
    obj.register_terminator(uninitialize_gsplat);
    obj.set_minimum_library_initialization_level(MODULE_INITIALIZATION_LEVEL_SCENE);
    return obj.init();
}
#endif
