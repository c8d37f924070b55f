// GsplatNative.cs — P/Invoke binding of libgsplat_hip.so (include/gsplat.h) for a Godot 4.3 C# project.
//
// One declaration per symbol and one struct per struct of the header, same field order, same C types
// (tests/test_shim_layout.py parses this file and include/gsplat.h and compares names, order, sizes and
// offsets; it also checks that every function of the header is bound here with the same parameter count).
// GaussianSplattingRasterizerNative.cs shows the class that replaces the body of
// util/gaussian_splatting_rasterizer.gd with these calls.
using System;
using System.Runtime.InteropServices;

namespace GsplatHip
{
    public enum GsplatStatus : int
    {
        Ok = 0,
        InvalidArgument = -1,
        OutOfMemory = -2,
        Hip = -3,
        NoDevice = -4,
        OutOfRange = -5,
        Unsupported = -6,
    }

    public static class GsplatFlags
    {
        public const uint Timing = 0x1;
        public const uint FixLastTile = 0x2;
        public const uint FastExp = 0x4;
        public const uint KeepEmitted = 0x8;
        public const uint KernelTiming = 0x10;
        public const uint BlockCull = 0x20;
        public const uint TiesStorageOrder = 0x40;
        public const uint ReadbackRgb = 0x80;
        public const uint NoTargetTile = 0xFFFFFFFF;
        public const uint StripeNone = 0, StripeColumns = 1, StripeRows = 2;
        public const int KernelClasses = 9;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GsplatConfig
    {
        public uint struct_size;
        public uint max_splats;
        public uint width;
        public uint height;
        public uint key_budget_factor;
        public int device_id;
        public uint flags;
        public uint stripe_axis;
        public uint stripe_begin;
        public uint stripe_end;
        public int sh_degree;
        public IntPtr stream;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GsplatFrame
    {
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 16)] public float[] view;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 16)] public float[] proj;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public float[] cam_pos;
        public float model_scale;
        public float time;
        public float heatmap_factor;
        public uint target_tile;
        public uint reserved;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct GsplatStats
    {
        public uint struct_size;   // in: Marshal.SizeOf<GsplatStats>() — the library fills at most that many bytes
        public uint reserved0;
        public ulong num_splats;
        public ulong num_visible;
        public ulong num_emitted;
        public ulong num_sorted;
        public ulong num_composited;
        public ulong capacity;
        public int overflow;
        public int sort_passes;
        public int sh_degree;
        public int lazy_colors;
        public float ms_projection;
        public float ms_sort;
        public float ms_boundaries;
        public float ms_render;
        public float ms_total;
        public int pair_key_bytes;
        public ulong bytes_allocated;
        public ulong scene_bytes;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 4)] public ulong[] algorithmic_bytes;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 9)] public float[] ms_kernel;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 9)] public uint[] launches_kernel;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 2)] public ulong[] pairs_round;
        public float ms_gather;
        public float ms_readback;
    }

    public static class Native
    {
        private const string Lib = "gsplat_hip"; // libgsplat_hip.so next to the project or on LD_LIBRARY_PATH

        [DllImport(Lib)] public static extern int gsplat_create(ref GsplatConfig config, out IntPtr out_ctx);
        [DllImport(Lib)] public static extern int gsplat_create_view(IntPtr scene_owner, ref GsplatConfig config, out IntPtr out_ctx);
        [DllImport(Lib)] public static extern int gsplat_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern int gsplat_upload_splats(IntPtr ctx, uint first, uint count, float[] records60);
        [DllImport(Lib)] public static extern int gsplat_upload_ply_rows(IntPtr ctx, uint first, uint count, float[] rows62, float load_time);
        [DllImport(Lib)] public static extern int gsplat_finalize_scene(IntPtr ctx);
        [DllImport(Lib)] public static extern int gsplat_resize(IntPtr ctx, uint width, uint height);
        [DllImport(Lib)] public static extern int gsplat_set_stripe(IntPtr ctx, uint stripe_axis, uint stripe_begin, uint stripe_end);
        [DllImport(Lib)] public static extern int gsplat_render(IntPtr ctx, ref GsplatFrame frame, byte[] rgba_out);
        [DllImport(Lib)] public static extern int gsplat_render_to(IntPtr ctx, ref GsplatFrame frame, IntPtr device_out, uint pitch_px, uint origin_x, uint origin_y);
        [DllImport(Lib)] public static extern int gsplat_render_begin(IntPtr ctx, ref GsplatFrame frame, IntPtr last_tile_out_device);
        [DllImport(Lib)] public static extern int gsplat_render_end(IntPtr ctx, IntPtr device_out, uint pitch_px, uint origin_x, uint origin_y, IntPtr frame_last_tile_device);
        [DllImport(Lib)] public static extern int gsplat_pick(IntPtr ctx, ref GsplatFrame frame, uint tile_id, float[] out_xyzn);
        [DllImport(Lib)] public static extern int gsplat_get_stats(IntPtr ctx, ref GsplatStats stats);
        [DllImport(Lib)] public static extern int gsplat_set_timing(IntPtr ctx, uint timing_flags);
        [DllImport(Lib)] public static extern int gsplat_debug_read(IntPtr ctx, int which, IntPtr dst, UIntPtr size, out UIntPtr bytes_written);
        [DllImport(Lib)] public static extern int gsplat_debug_pow02(IntPtr ctx, uint first_bits, ulong count, float[] out_host);
        [DllImport(Lib)] public static extern int gsplat_render_async(IntPtr ctx, ref GsplatFrame frame, out ulong ticket_out);
        [DllImport(Lib)] public static extern int gsplat_readback_wait(IntPtr ctx, ulong ticket, out IntPtr host_rgba_out);
        [DllImport(Lib)] public static extern int gsplat_bind_external_image(IntPtr ctx, int fd, ulong size_bytes, ulong offset_bytes);
        [DllImport(Lib)] public static extern int gsplat_export_image_fd(IntPtr ctx, out int fd_out, out ulong size_bytes_out);
        [DllImport(Lib)] public static extern int gsplat_group_unique_id(byte[] id_out);
        [DllImport(Lib)] public static extern int gsplat_group_create(IntPtr ctx, byte[] id, int rank, int world, uint stripe_axis, out IntPtr out_group);
        [DllImport(Lib)] public static extern int gsplat_group_create_local(IntPtr[] ctxs, int n, uint stripe_axis, out IntPtr out_group);
        [DllImport(Lib)] public static extern int gsplat_group_set_cuts(IntPtr group, uint[] cuts);
        [DllImport(Lib)] public static extern int gsplat_group_render(IntPtr group, ref GsplatFrame frame, IntPtr[] outs);
        [DllImport(Lib)] public static extern int gsplat_group_exchanges_last_tile(IntPtr group);
        [DllImport(Lib)] public static extern int gsplat_group_render_batch(IntPtr group, [In] GsplatFrame[] frames, uint count);
        [DllImport(Lib)] public static extern int gsplat_group_destroy(IntPtr group);
        // batched frames: `count` consecutive frames of one context through one launch sequence (gsplat.h)
        [DllImport(Lib)] public static extern int gsplat_create_batch_view(IntPtr scene_owner, ref GsplatConfig config, uint batch, out IntPtr out_ctx);
        [DllImport(Lib)] public static extern int gsplat_render_batch(IntPtr ctx, [In] GsplatFrame[] frames, uint count);
        [DllImport(Lib)] public static extern int gsplat_render_batch_begin(IntPtr ctx, [In] GsplatFrame[] frames, uint count, IntPtr last_tiles_out_device);
        [DllImport(Lib)] public static extern int gsplat_render_batch_end(IntPtr ctx, IntPtr frame_last_tiles_device);
        [DllImport(Lib)] public static extern int gsplat_batch_image_device_ptr(IntPtr ctx, uint index, out IntPtr out_ptr);
        [DllImport(Lib)] public static extern int gsplat_image_device_ptr(IntPtr ctx, out IntPtr out_ptr);
        [DllImport(Lib)] public static extern int gsplat_synchronize(IntPtr ctx);
        [DllImport(Lib)] public static extern int gsplat_make_view_proj(float[] camera_xform, float[] basis_override, float fovy_degrees, float aspect, float z_near, float z_far, float[] out32, float[] out_cam_pos);
        [DllImport(Lib)] public static extern IntPtr gsplat_status_string(int status);
        [DllImport(Lib)] public static extern IntPtr gsplat_last_error();
        [DllImport(Lib)] public static extern uint gsplat_version();

        public static void Check(int status, string where)
        {
            if (status == 0) return;
            string msg = Marshal.PtrToStringAnsi(gsplat_status_string(status)) ?? "?";
            string detail = Marshal.PtrToStringAnsi(gsplat_last_error()) ?? "";
            throw new InvalidOperationException($"{where}: {msg} ({status}) {detail}");
        }
    }
}
