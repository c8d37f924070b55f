"""Child process of tests/test_gpu_parity.py::test_frames_land_in_memory_exported_by_another_process: it OWNS a device
image (stand-in for the Vulkan texture a Godot process owns, gaussian_splatting_rasterizer.gd:92,101), exports it as a
dma-buf descriptor, passes the descriptor over a UNIX socket (SCM_RIGHTS — how a Vulkan / compositor process hands memory
to another process), waits until the renderer says the frame is there, and saves what ITS memory holds."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from godotgaussiansplatting_amd import capi  # noqa: E402

sock_fd, w, h, out_path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sock = socket.socket(fileno=sock_fd)
owner = capi.Context(1, w, h)
try:
    fd, size = owner.export_image_fd()
except Exception as e:  # noqa: BLE001
    sock.sendall(b"NOEXPORT " + repr(e).encode()[:200])
    sys.exit(0)
socket.send_fds(sock, [b"FD %d" % size], [fd])
os.close(fd)                      # (the receiver holds its own duplicate now)
msg = sock.recv(64)
assert msg.startswith(b"RENDERED"), msg
np.save(out_path, owner.read_image())
sock.sendall(b"SAVED")
owner.close()
