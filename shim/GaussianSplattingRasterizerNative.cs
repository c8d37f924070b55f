// GaussianSplattingRasterizerNative.cs — the reference's util/gaussian_splatting_rasterizer.gd with the compute-shader
// pipeline replaced by libgsplat_hip.so.  Same public surface (texture_size, render_scale, model_scale,
// should_enable_heatmap, basis_override, num_splats_loaded, is_loaded, `loaded` signal, rasterize(),
// get_splat_position(), update_camera_matrices(), cleanup_gpu()), so main.gd keeps working unchanged when it
// constructs this class instead of the GDScript one.  Not compiled in this repository (no Godot / .NET in the
// image); the P/Invoke layer it sits on is checked field by field (tests/test_shim_layout.py) and the same call
// sequence is compiled and run as examples/gsplat_render_ply.c.
using System;
using System.Threading;
using Godot;

namespace GsplatHip
{
    public partial class GaussianSplattingRasterizerNative : Resource
    {
        [Signal] public delegate void LoadedEventHandler();                       // gaussian_splatting_rasterizer.gd:10

        public const int TileSize = 16;                                           // :4
        public float RenderScale = 1.0f, ModelScale = 1.0f;                       // :52-53
        public bool ShouldEnableHeatmap = false, IsLoaded = false;                // :54,56
        public Basis BasisOverride = Basis.Identity;                              // :57
        public int NumSplatsLoaded;                                               // :51
        public Vector2I TileDims { get; private set; }

        private readonly float[] _rows;       // PlyFile.vertices: 62 floats per splat (ply_file.gd:3-5)
        private readonly int _numSplats;
        private readonly Camera3D _camera;
        private readonly Texture2Drd _renderTexture;
        private RenderingDevice _device;
        private Rid _textureRid;
        private IntPtr _ctx = IntPtr.Zero;
        private Thread _loadThread;
        private volatile bool _terminate;
        private byte[] _rgba = Array.Empty<byte>();
        private float[] _viewProj = new float[32];
        private float[] _camPos = new float[3];
        private Vector2I _textureSize;

        public GaussianSplattingRasterizerNative(float[] plyRows, Vector2I outputTextureSize, Texture2Drd renderTexture,
                                                 Camera3D camera)                  // _init, :59-63
        {
            _rows = plyRows;
            _numSplats = plyRows.Length / 62;
            _renderTexture = renderTexture;
            _camera = camera;
            TextureSize = outputTextureSize;
        }

        public Vector2I TextureSize                                                // setter, :26-48
        {
            get => _textureSize;
            set
            {
                _textureSize = new Vector2I(Math.Max(1, (int)(value.X * RenderScale)), Math.Max(1, (int)(value.Y * RenderScale)));
                TileDims = (_textureSize + new Vector2I(TileSize - 1, TileSize - 1)) / TileSize;   // :29
                if (_ctx == IntPtr.Zero) return;
                Native.Check(Native.gsplat_resize(_ctx, (uint)_textureSize.X, (uint)_textureSize.Y), "gsplat_resize");
                CreateTexture();
            }
        }

        private void CreateTexture()                                               // :39-48,92,101
        {
            if (_textureRid.IsValid) _device.FreeRid(_textureRid);
            var fmt = new RDTextureFormat
            {
                Format = RenderingDevice.DataFormat.R32G32B32A32Sfloat,
                Width = (uint)_textureSize.X, Height = (uint)_textureSize.Y,
                UsageBits = RenderingDevice.TextureUsageBits.SamplingBit | RenderingDevice.TextureUsageBits.CanUpdateBit,
            };
            _textureRid = _device.TextureCreate(fmt, new RDTextureView());
            _renderTexture.TextureRdRid = _textureRid;
            _rgba = new byte[_textureSize.X * _textureSize.Y * 16];
        }

        public void InitGpu()                                                      // init_gpu, :65-114
        {
            _device = RenderingServer.GetRenderingDevice();
            var cfg = new GsplatConfig
            {
                struct_size = (uint)System.Runtime.InteropServices.Marshal.SizeOf<GsplatConfig>(),
                max_splats = (uint)_numSplats, width = (uint)_textureSize.X, height = (uint)_textureSize.Y,
                key_budget_factor = 10, device_id = -1, flags = GsplatFlags.Timing, sh_degree = -1, stream = IntPtr.Zero,
            };
            Native.Check(Native.gsplat_create(ref cfg, out _ctx), "gsplat_create");
            CreateTexture();
            _loadThread = new Thread(LoadSplats);                                  // :114, ply_file.gd:28-77
            _loadThread.Start();
        }

        private void LoadSplats()
        {
            int stride = Math.Max(1, _numSplats / 1000);                           // :114
            var chunk = new float[stride * 62];
            for (int first = 0; first < _numSplats && !_terminate; first += stride)
            {
                int count = Math.Min(stride, _numSplats - first);
                Array.Copy(_rows, first * 62, chunk, 0, count * 62);
                float now = Time.GetTicksMsec() * 1e-3f;                           // creation_time, ply_file.gd:39
                Native.Check(Native.gsplat_upload_ply_rows(_ctx, (uint)first, (uint)count, chunk, now), "gsplat_upload_ply_rows");
                Interlocked.Add(ref NumSplatsLoaded, count);                       // ply_file.gd:72-74
            }
            if (_terminate) return;
            IsLoaded = true;
            CallDeferred(GodotObject.MethodName.EmitSignal, SignalName.Loaded);    // ply_file.gd:77
        }

        public bool UpdateCameraMatrices()                                         // :175-195
        {
            Transform3D view = new Transform3D(BasisOverride, Vector3.Zero) * _camera.GlobalTransform;
            var b = view.Basis;
            float[] cam12 = { b.X.X, b.X.Y, b.X.Z, b.Y.X, b.Y.Y, b.Y.Z, b.Z.X, b.Z.Y, b.Z.Z, view.Origin.X, view.Origin.Y, view.Origin.Z };
            var next = new float[32];
            float aspect = (float)_textureSize.X / _textureSize.Y;
            Native.Check(Native.gsplat_make_view_proj(cam12, null, _camera.Fov, aspect, _camera.Near, _camera.Far, next, _camPos),
                         "gsplat_make_view_proj");
            bool changed = !((ReadOnlySpan<float>)next).SequenceEqual(_viewProj);
            _viewProj = next;
            return changed;
        }

        public void Rasterize()                                                    // rasterize, :122-160 (render thread)
        {
            if (_ctx == IntPtr.Zero) InitGpu();
            var frame = new GsplatFrame
            {
                view = _viewProj[..16], proj = _viewProj[16..], cam_pos = _camPos,
                model_scale = ModelScale, time = Time.GetTicksMsec() * 1e-3f,      // :125-126
                heatmap_factor = ShouldEnableHeatmap ? 1.0f : 0.0f, target_tile = GsplatFlags.NoTargetTile,  // :158
            };
            Native.Check(Native.gsplat_render(_ctx, ref frame, _rgba), "gsplat_render");   // 15 dispatches -> one call
            _device.TextureUpdate(_textureRid, 0, _rgba);                          // feeds the same Texture2DRD (:92,101)
        }

        public Vector3 GetSplatPosition(Vector2 screenPos)                         // get_splat_position, :162-171
        {
            Vector2I tile = (Vector2I)(screenPos * RenderScale / TileSize);
            uint tileId = (uint)(tile.Y * TileDims.X + tile.X);                    // :164
            var frame = new GsplatFrame
            {
                view = _viewProj[..16], proj = _viewProj[16..], cam_pos = _camPos, model_scale = ModelScale,
                time = Time.GetTicksMsec() * 1e-3f, heatmap_factor = ShouldEnableHeatmap ? 1.0f : 0.0f, target_tile = tileId,
            };
            var s = new float[4];
            Native.Check(Native.gsplat_pick(_ctx, ref frame, tileId, s), "gsplat_pick");
            if (s[3] == 0.0f) return Vector3.Inf;                                  // :171
            return BasisOverride.Inverse() * new Vector3(-s[0], -s[1], s[2]);
        }

        public GsplatStats DebugInfo()                                             // update_debug_info, main.gd:93-119
        {
            var st = new GsplatStats { struct_size = (uint)System.Runtime.InteropServices.Marshal.SizeOf<GsplatStats>() };
            Native.Check(Native.gsplat_get_stats(_ctx, ref st), "gsplat_get_stats");
            return st;   // num_emitted / overflow -> "rendered splats (buffer overflow!)", ms_* -> the stage timings
        }

        public void CleanupGpu()                                                   // cleanup_gpu, :116-120
        {
            _terminate = true;
            _loadThread?.Join();
            if (_ctx != IntPtr.Zero) Native.gsplat_destroy(_ctx);
            _ctx = IntPtr.Zero;
            if (_textureRid.IsValid) _device.FreeRid(_textureRid);
        }
    }
}
