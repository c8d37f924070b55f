"""CPU oracle for the forward Gaussian-splat path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
See the header of oracle/gsplat_oracle.c for the parity status ("parity unpinned") and the
arithmetic contract.
"""
from .cpu import (  # noqa: F401
    Frame, Stats, render_frame, project, sort_pairs, boundaries, render_tiles, records_from_ply_rows,
    pack_camera, pow02, pow02_bits, exp2, num_threads, set_num_threads, lib_path, grid,
)
