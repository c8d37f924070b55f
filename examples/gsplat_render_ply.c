/*
 * gsplat_render_ply.c — a plain-C host of libgsplat_hip.so: the same sequence of calls a Godot-side shim makes
 * (INTEGRATION.md), without Python or PyTorch.
 *
 *   gsplat_render_ply scene.ply out.ppm [width height [cam_x cam_y cam_z]] [--batch B]
 *
 * Loads an INRIA-style binary .ply the way util/ply_file.gd:10-19 does (naive header walk, 62 float properties per
 * vertex), uploads the raw rows (the per-vertex swizzle of ply_file.gd:41-69 runs on the GPU), renders one frame
 * with a Godot default camera (fov 75, near 0.05, far 4000) looking at the origin from (cam_x, cam_y, cam_z)
 * (default 0 0 5), prints the stats of main.gd:93-119 and writes the frame as a binary PPM (clamped to [0,1]).
 * --batch B (2..4): afterwards B frames of an orbit (2 degrees per frame) go through ONE launch sequence
 * (gsplat_create_batch_view / gsplat_render_batch) and each is compared, byte for byte, with the same frame rendered alone.
 *
 * Build:  gcc -O2 -Iinclude examples/gsplat_render_ply.c -Lgodotgaussiansplatting_amd -lgsplat_hip \
 *             -Wl,-rpath,'$ORIGIN/../godotgaussiansplatting_amd' -lm -o examples/gsplat_render_ply
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsplat.h"

#define CHECK(call)                                                                                              \
    do {                                                                                                         \
        int _rc = (call);                                                                                        \
        if (_rc != GSPLAT_OK) {                                                                                  \
            fprintf(stderr, "%s -> %d (%s) %s\n", #call, _rc, gsplat_status_string(_rc), gsplat_last_error());   \
            return 2;                                                                                            \
        }                                                                                                        \
    } while (0)

static float *load_ply(const char *path, uint32_t *count_out) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    char line[512];
    long count = 0;
    int props = 0, big_endian = 0;
    while (fgets(line, sizeof line, f)) {
        if (!strncmp(line, "end_header", 10)) break;
        if (!strncmp(line, "format", 6)) big_endian = strstr(line, "binary_big_endian") != NULL;
        else if (!strncmp(line, "element", 7)) sscanf(line, "element %*s %ld", &count);
        else if (!strncmp(line, "property", 8)) ++props;
    }
    if (props != GSPLAT_PLY_ROW_FLOATS || count <= 0) {
        fprintf(stderr, "%s: need %d float properties per vertex, found %d (vertices %ld)\n", path,
                GSPLAT_PLY_ROW_FLOATS, props, count);
        fclose(f);
        return NULL;
    }
    const size_t n = (size_t)count * props;
    float *rows = (float *)malloc(n * sizeof(float));
    if (!rows || fread(rows, sizeof(float), n, f) != n) {
        fprintf(stderr, "%s: short read\n", path);
        fclose(f);
        free(rows);
        return NULL;
    }
    fclose(f);
    if (big_endian) {
        unsigned char *b = (unsigned char *)rows;
        for (size_t i = 0; i < n; ++i, b += 4) {
            unsigned char t = b[0]; b[0] = b[3]; b[3] = t;
            t = b[1]; b[1] = b[2]; b[2] = t;
        }
    }
    *count_out = (uint32_t)count;
    return rows;
}

/* Godot Transform3D.looking_at: camera -Z points at the target, up = +Y */
static void look_at(const float eye[3], float xform12[12]) {
    float z[3] = {eye[0], eye[1], eye[2]};
    float len = sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    if (len == 0.0f) { z[2] = 1.0f; len = 1.0f; }
    for (int i = 0; i < 3; ++i) z[i] /= len;
    float up[3] = {0, 1, 0};
    float x[3] = {up[1] * z[2] - up[2] * z[1], up[2] * z[0] - up[0] * z[2], up[0] * z[1] - up[1] * z[0]};
    len = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (len == 0.0f) { x[0] = 1.0f; x[1] = x[2] = 0.0f; len = 1.0f; }
    for (int i = 0; i < 3; ++i) x[i] /= len;
    float y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    memcpy(xform12 + 0, x, sizeof x);
    memcpy(xform12 + 3, y, sizeof y);
    memcpy(xform12 + 6, z, sizeof z);
    memcpy(xform12 + 9, eye, 3 * sizeof(float));
}

static int make_frame(const float eye[3], uint32_t width, uint32_t height, gsplat_frame *frame) {
    float xform[12], vp[32];
    look_at(eye, xform);
    memset(frame, 0, sizeof *frame);
    CHECK(gsplat_make_view_proj(xform, NULL, 75.0f, (float)width / (float)height, 0.05f, 4000.0f, vp, frame->cam_pos));
    memcpy(frame->view, vp, sizeof frame->view);
    memcpy(frame->proj, vp + 16, sizeof frame->proj);
    frame->model_scale = 1.0f;
    frame->time = 0.0f;
    frame->target_tile = GSPLAT_NO_TARGET_TILE;
    return 0;
}

int main(int argc, char **argv) {
    const int help = argc > 1 && !strcmp(argv[1], "--help");
    uint32_t batch = 0;
    if (argc >= 5 && !strcmp(argv[argc - 2], "--batch")) {
        batch = (uint32_t)atoi(argv[argc - 1]);
        argc -= 2;
    }
    if (argc < 3 || help || batch == 1 || batch > GSPLAT_MAX_BATCH) {
        fprintf(stderr, "usage: %s scene.ply out.ppm [width height [cam_x cam_y cam_z]] [--batch 2..%d]   (libgsplat_hip %u.%u)\n",
                argv[0], GSPLAT_MAX_BATCH, gsplat_version() >> 16, gsplat_version() & 0xFFFFu);
        return help ? 0 : 1;
    }
    const uint32_t width = argc > 4 ? (uint32_t)atoi(argv[3]) : 1280, height = argc > 4 ? (uint32_t)atoi(argv[4]) : 720;
    float eye[3] = {0.0f, 0.0f, 5.0f};
    if (argc > 7) for (int i = 0; i < 3; ++i) eye[i] = (float)atof(argv[5 + i]);

    uint32_t count = 0;
    float *rows = load_ply(argv[1], &count);
    if (!rows) return 1;

    gsplat_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_splats = count;
    cfg.width = width; cfg.height = height;
    cfg.key_budget_factor = 10;          /* gaussian_splatting_rasterizer.gd:79 */
    cfg.device_id = -1;
    cfg.flags = GSPLAT_FLAG_TIMING;
    cfg.sh_degree = -1;
    gsplat_ctx *ctx = NULL;
    CHECK(gsplat_create(&cfg, &ctx));
    /* ply_file.gd:28-77: chunked upload; creation_time well in the past => load animation finished */
    const uint32_t stride = count / 1000u ? count / 1000u : 1u;
    for (uint32_t first = 0; first < count; first += stride * 64u) {
        const uint32_t m = count - first < stride * 64u ? count - first : stride * 64u;
        CHECK(gsplat_upload_ply_rows(ctx, first, m, rows + (size_t)first * GSPLAT_PLY_ROW_FLOATS, -10.0f));
    }

    gsplat_frame frame;
    if (make_frame(eye, width, height, &frame)) return 2;

    float *rgba = (float *)malloc((size_t)width * height * 4 * sizeof(float));
    CHECK(gsplat_render(ctx, &frame, rgba));   /* warm-up (first launch loads the code object) */
    CHECK(gsplat_render(ctx, &frame, rgba));
    gsplat_stats st;
    st.struct_size = sizeof st;
    CHECK(gsplat_get_stats(ctx, &st));
    printf("splats %llu visible %llu pairs %llu%s  sh_degree %d  VRAM %.1f MB\n", (unsigned long long)st.num_splats,
           (unsigned long long)st.num_visible, (unsigned long long)st.num_emitted,
           st.overflow ? " (buffer overflow!)" : "", st.sh_degree, st.bytes_allocated / 1e6);
    printf("projection %.3f ms  sort %.3f ms  boundaries %.3f ms  render %.3f ms  total %.3f ms\n", st.ms_projection,
           st.ms_sort, st.ms_boundaries, st.ms_render, st.ms_total);

    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 1; }
    fprintf(out, "P6\n%u %u\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        unsigned char px[3];
        for (int ch = 0; ch < 3; ++ch) {
            float v = rgba[i * 4 + ch];
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            px[ch] = (unsigned char)(v * 255.0f + 0.5f);
        }
        fwrite(px, 1, 3, out);
    }
    fclose(out);
    if (batch >= 2) {
        /* frames in batches: B cameras of an orbit through ONE launch sequence, each compared with the frame rendered alone */
        gsplat_frame frames[GSPLAT_MAX_BATCH];
        const float r = sqrtf(eye[0] * eye[0] + eye[2] * eye[2]), a0 = atan2f(eye[0], eye[2]);
        for (uint32_t k = 0; k < batch; ++k) {
            const float a = a0 + 0.034906585f * (float)k;
            const float e[3] = {r * sinf(a), eye[1], r * cosf(a)};
            if (make_frame(e, width, height, &frames[k])) return 2;
        }
        gsplat_config bcfg = cfg;
        bcfg.max_splats = 0;                 /* the owner's */
        bcfg.flags = 0;
        gsplat_ctx *bctx = NULL;
        CHECK(gsplat_create_batch_view(ctx, &bcfg, batch, &bctx));
        CHECK(gsplat_render_batch(bctx, frames, batch));
        const size_t frame_bytes = (size_t)width * height * 4 * sizeof(float);
        float *images = (float *)malloc(frame_bytes * batch);
        size_t got = 0;
        CHECK(gsplat_debug_read(bctx, GSPLAT_DEBUG_IMAGE, images, frame_bytes * batch, &got));
        int same = got == frame_bytes * batch;
        for (uint32_t k = 0; k < batch && same; ++k) {
            CHECK(gsplat_render(ctx, &frames[k], rgba));
            same = memcmp(rgba, images + (size_t)k * width * height * 4, frame_bytes) == 0;
        }
        printf("batch of %u frames through one launch sequence: identical to gsplat_render frame by frame: %s\n", batch,
               same ? "yes" : "NO");
        free(images);
        CHECK(gsplat_destroy(bctx));
        if (!same) return 3;
    }
    free(rgba);
    free(rows);
    CHECK(gsplat_destroy(ctx));
    return 0;
}
