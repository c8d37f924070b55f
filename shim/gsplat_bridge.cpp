// gsplat_bridge.cpp — see gsplat_bridge.h.  Line references: util/gaussian_splatting_rasterizer.gd of the reference.
#include "gsplat_bridge.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace gsplat_shim {

Bridge::Bridge(const float *ply_rows62, uint32_t num_splats, uint32_t width, uint32_t height)
    : rows_(ply_rows62), num_splats_(num_splats), width_(std::max(1u, width)), height_(std::max(1u, height)) {}

Bridge::~Bridge() {
    terminate_.store(true);                       // :117-118 should_terminate_thread + wait_to_finish
    if (loader_.joinable()) loader_.join();
    if (ctx_) gsplat_destroy(ctx_);               // :119 context.free()
}

int Bridge::fail(int status, const char *where) {
    error_ = std::string(where) + ": " + gsplat_status_string(status) + " " + gsplat_last_error();
    return status;
}

int Bridge::init_gpu(double now_seconds) {
    if (ctx_) return GSPLAT_OK;
    gsplat_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_splats = num_splats_;                 // :79,83
    cfg.width = width_;
    cfg.height = height_;
    cfg.key_budget_factor = 10;                   // :79
    cfg.device_id = -1;
    cfg.flags = GSPLAT_FLAG_TIMING;               // the capture_timestamp calls of :135-160
    if (readback_rgb) cfg.flags |= GSPLAT_FLAG_READBACK_RGB;
    cfg.sh_degree = -1;
    const int rc = gsplat_create(&cfg, &ctx_);
    if (rc != GSPLAT_OK) return fail(rc, "gsplat_create");
    rgba_.assign((size_t)width_ * height_ * 4, 0.0f);
    loader_ = std::thread(&Bridge::load_splats, this, now_seconds);   // :114
    return GSPLAT_OK;
}

void Bridge::load_splats(double t0) {             // PlyFile.load_gaussian_splats, ply_file.gd:28-77
    const auto start = std::chrono::steady_clock::now();
    const uint32_t stride = std::max(1u, num_splats_ / 1000u);        // :114
    for (uint32_t first = 0; first < num_splats_ && !terminate_.load(); first += stride) {
        const uint32_t count = std::min(stride, num_splats_ - first);
        const double now = t0 + std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        if (gsplat_upload_ply_rows(ctx_, first, count, rows_ + (size_t)first * GSPLAT_PLY_ROW_FLOATS, (float)now) != GSPLAT_OK)
            return;
        num_splats_loaded.fetch_add(count);       // ply_file.gd:72-74
    }
    if (!terminate_.load()) is_loaded.store(true);  // -> the `loaded` signal (:10), emitted by the GDExtension wrapper
}

int Bridge::set_texture_size(uint32_t viewport_w, uint32_t viewport_h) {   // :26-48
    width_ = std::max(1u, (uint32_t)(viewport_w * render_scale));
    height_ = std::max(1u, (uint32_t)(viewport_h * render_scale));
    if (!ctx_) return GSPLAT_OK;
    const int rc = gsplat_resize(ctx_, width_, height_);
    if (rc != GSPLAT_OK) return fail(rc, "gsplat_resize");
    pending_ticket_ = 0;   // (the read-back ring was sized for the old frame)
    rgba_.assign((size_t)width_ * height_ * 4, 0.0f);
    return GSPLAT_OK;
}

bool Bridge::update_camera_matrices(const CameraState &cam) {   // :175-195
    float next[32], pos[3];
    const float aspect = (float)width_ / (float)height_;
    if (gsplat_make_view_proj(cam.xform, cam.basis_override, cam.fovy_degrees, aspect, cam.z_near, cam.z_far, next, pos) != GSPLAT_OK)
        return false;
    // inverse of basis_override (a rotation/scale basis: adjugate / determinant), for get_splat_position :171
    const float *b = cam.basis_override;   // columns
    const float m[3][3] = {{b[0], b[3], b[6]}, {b[1], b[4], b[7]}, {b[2], b[5], b[8]}};
    const float det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                      m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    if (det != 0.0f) {
        const float inv[3][3] = {
            {(m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det, (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det, (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det},
            {(m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det, (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det, (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det},
            {(m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det, (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det, (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det}};
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) inv_override_[c * 3 + r] = inv[r][c];
    }
    const bool changed = std::memcmp(next, view_proj_, sizeof next) != 0 || std::memcmp(pos, cam_pos_, sizeof pos) != 0;
    std::memcpy(view_proj_, next, sizeof next);
    std::memcpy(cam_pos_, pos, sizeof pos);
    return changed;
}

gsplat_frame Bridge::make_frame(double now_seconds, uint32_t target_tile) const {
    gsplat_frame f;
    std::memset(&f, 0, sizeof f);
    std::memcpy(f.view, view_proj_, sizeof f.view);
    std::memcpy(f.proj, view_proj_ + 16, sizeof f.proj);
    std::memcpy(f.cam_pos, cam_pos_, sizeof f.cam_pos);     // :125-126 (-x, -y, z of basis_override * camera origin)
    f.model_scale = model_scale;
    f.time = (float)now_seconds;
    f.heatmap_factor = should_enable_heatmap ? 1.0f : 0.0f;  // :158
    f.target_tile = target_tile;
    return f;
}

int Bridge::rasterize(double now_seconds) {       // :122-160
    if (!ctx_) {
        const int rc = init_gpu(now_seconds);     // :123
        if (rc != GSPLAT_OK) return rc;
    }
    const gsplat_frame f = make_frame(now_seconds, GSPLAT_NO_TARGET_TILE);
    const int rc = gsplat_render(ctx_, &f, rgba_.data());   // projection, sort, tile ranges, compositor: one call
    return rc == GSPLAT_OK ? rc : fail(rc, "gsplat_render");
}

int Bridge::rasterize_pipelined(double now_seconds, const float **rgba_out) {
    if (!ctx_) {
        const int rc = init_gpu(now_seconds);
        if (rc != GSPLAT_OK) return rc;
    }
    const gsplat_frame f = make_frame(now_seconds, GSPLAT_NO_TARGET_TILE);
    uint64_t ticket = 0;
    int rc = gsplat_render_async(ctx_, &f, &ticket);
    if (rc != GSPLAT_OK) return fail(rc, "gsplat_render_async");
    *rgba_out = nullptr;
    if (pending_ticket_) {
        rc = gsplat_readback_wait(ctx_, pending_ticket_, rgba_out);
        if (rc != GSPLAT_OK) return fail(rc, "gsplat_readback_wait");
    }
    pending_ticket_ = ticket;
    return GSPLAT_OK;
}

int Bridge::bind_texture_memory(int fd, uint64_t size_bytes, uint64_t offset_bytes) {
    if (!ctx_) return fail(GSPLAT_ERR_INVALID_ARGUMENT, "bind_texture_memory before init_gpu");
    const int rc = gsplat_bind_external_image(ctx_, fd, size_bytes, offset_bytes);
    return rc == GSPLAT_OK ? rc : fail(rc, "gsplat_bind_external_image");
}

int Bridge::rasterize_to_bound(double now_seconds) {
    const gsplat_frame f = make_frame(now_seconds, GSPLAT_NO_TARGET_TILE);
    int rc = gsplat_render(ctx_, &f, nullptr);
    if (rc == GSPLAT_OK) rc = gsplat_synchronize(ctx_);   // (a shared semaphore on the context's stream in a real engine)
    return rc == GSPLAT_OK ? rc : fail(rc, "gsplat_render");
}

int Bridge::get_splat_position(float screen_x, float screen_y, double now_seconds, float out_xyz[3], bool *hit) {
    const uint32_t tx = (uint32_t)(screen_x * render_scale / kTileSize), ty = (uint32_t)(screen_y * render_scale / kTileSize);   // :163
    const uint32_t tile_id = ty * tile_dims_x() + tx;                                                                          // :164
    const gsplat_frame f = make_frame(now_seconds, tile_id);
    float s[4] = {0, 0, 0, 0};
    const int rc = gsplat_pick(ctx_, &f, tile_id, s);        // :166-170
    if (rc != GSPLAT_OK) return fail(rc, "gsplat_pick");
    *hit = s[3] != 0.0f;                                     // :171 w == 0 -> Vector3.INF
    const float p[3] = {-s[0], -s[1], s[2]};
    for (int r = 0; r < 3; ++r)
        out_xyz[r] = inv_override_[0 * 3 + r] * p[0] + inv_override_[1 * 3 + r] * p[1] + inv_override_[2 * 3 + r] * p[2];
    return GSPLAT_OK;
}

int Bridge::debug_info(gsplat_stats *out) const {
    out->struct_size = sizeof(gsplat_stats);  // (the library never writes past what the caller was built against)
    return gsplat_get_stats(ctx_, out);
}

}  // namespace gsplat_shim
