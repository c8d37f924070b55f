"""Worker of tests/test_bench_fallback.py: what bench.py's second attempt relies on, without a GPU.  First life: join
the launcher's rendezvous (gloo), then — on rank 1 half a second later than on rank 0, as watchdogs do not fire
together — replace the process by a second life with bench.fallback_env(); second life: a rendezvous of its own must
come up (fresh store on another port, no stale keys) and a collective must complete."""
import importlib.util
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
t = torch.tensor([rank + 1.0])
dist.all_reduce(t)
assert t.item() == world * (world + 1) / 2
if os.environ.get("GSPLAT_BENCH_FELL_BACK") != "1":
    time.sleep(0.5 * rank)
    os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)], bench.fallback_env(os.environ, "simulated", "group1"))
if rank == 0:
    print("SECOND_LIFE_OK", os.environ["GSPLAT_BENCH_DIST_NOTE"], os.environ["MASTER_PORT"], flush=True)
dist.destroy_process_group()
