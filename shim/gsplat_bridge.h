// gsplat_bridge.h — C++ side of the Godot shim: the state machine of util/gaussian_splatting_rasterizer.gd over the C ABI
// of libgsplat_hip.so, free of Godot types so that it compiles (and is compile-checked, tests/test_shim_layout.py)
// without godot-cpp.  gsplat_gdextension.cpp wraps it in a GDExtension class when godot-cpp is on the include path.
#pragma once
#include <atomic>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

#include "../include/gsplat.h"

namespace gsplat_shim {

struct CameraState {          // what update_camera_matrices() reads from Camera3D (gaussian_splatting_rasterizer.gd:175-195)
    float xform[12];          // camera-to-world basis columns X, Y, Z, then origin
    float basis_override[9];  // columns
    float fovy_degrees, z_near, z_far;
};

class Bridge {
public:
    static constexpr int kTileSize = GSPLAT_TILE_SIZE;                      // :4
    Bridge(const float *ply_rows62, uint32_t num_splats, uint32_t width, uint32_t height);   // _init, :59-63
    ~Bridge();                                                              // cleanup_gpu, :116-120
    Bridge(const Bridge &) = delete;
    Bridge &operator=(const Bridge &) = delete;

    // properties of the reference class (:51-57)
    float render_scale = 1.0f, model_scale = 1.0f;
    bool should_enable_heatmap = false;
    bool readback_rgb = false;   // set before init_gpu: rasterize_pipelined delivers RGB32F (Image.FORMAT_RGBF), 12 B/px over PCIe
    std::atomic<uint32_t> num_splats_loaded{0};
    std::atomic<bool> is_loaded{false};

    int init_gpu(double now_seconds);                                       // :65-114 (starts the loader thread)
    int set_texture_size(uint32_t viewport_w, uint32_t viewport_h);         // texture_size setter, :26-48
    bool update_camera_matrices(const CameraState &cam);                    // :175-195; true if anything changed
    int rasterize(double now_seconds);                                      // :122-160; the frame is in rgba()
    // The same frame through the library's pinned read-back ring (gsplat_render_async): returns at once, *rgba_out is
    // the PREVIOUS call's frame (nullptr on the first call) — one frame of latency buys a copy that overlaps the next
    // frame's kernels instead of stalling the render thread (fps_with_d2h in bench.py).  Valid until two calls later.
    int rasterize_pipelined(double now_seconds, const float **rgba_out);
    // Render straight into memory the engine owns: fd = the opaque fd exported for the Vulkan image behind Texture2DRD
    // (gaussian_splatting_rasterizer.gd:92,101; RenderingDevice.get_driver_resource + vkGetMemoryFdKHR).  After this,
    // rasterize_to_bound() leaves the frame there and rgba() is not touched.
    int bind_texture_memory(int fd, uint64_t size_bytes, uint64_t offset_bytes);
    int rasterize_to_bound(double now_seconds);
    int get_splat_position(float screen_x, float screen_y, double now_seconds, float out_xyz[3], bool *hit);  // :162-171
    int debug_info(gsplat_stats *out) const;                                // update_debug_info, main.gd:93-119

    const std::vector<float> &rgba() const { return rgba_; }                // W*H*4 floats for RenderingDevice.texture_update
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    uint32_t readback_channels() const { return readback_rgb ? 3u : 4u; }   // floats per pixel of rasterize_pipelined's image
    uint32_t tile_dims_x() const { return (width_ + kTileSize - 1) / kTileSize; }
    uint32_t tile_dims_y() const { return (height_ + kTileSize - 1) / kTileSize; }
    const std::string &last_error() const { return error_; }

private:
    gsplat_frame make_frame(double now_seconds, uint32_t target_tile) const;
    int fail(int status, const char *where);
    void load_splats(double t0);

    const float *rows_;
    uint32_t num_splats_, width_, height_;
    gsplat_ctx *ctx_ = nullptr;
    std::thread loader_;
    std::atomic<bool> terminate_{false};
    std::vector<float> rgba_;
    float view_proj_[32] = {0};
    float cam_pos_[3] = {0};
    float inv_override_[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::string error_;
    uint64_t pending_ticket_ = 0;
};

}  // namespace gsplat_shim
