"""bench.py --gpus N: a first attempt (`--dist group`) that fails or does not finish is replaced, rank by rank and in
place, by a second one with the torch host (bench.restart_with_torch_host).  The part that can be checked without a GPU:
under the driver's launcher, processes that replace themselves find each other again."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_second_attempt_gets_a_rendezvous_of_its_own():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    env = bench.fallback_env({"MASTER_PORT": "29500", "TORCHELASTIC_USE_AGENT_STORE": "True", "X": "1"}, "why")
    assert env["MASTER_PORT"] == "29523" and env["TORCHELASTIC_USE_AGENT_STORE"] == "False"
    assert env["GSPLAT_BENCH_FELL_BACK"] == "1" and env["GSPLAT_BENCH_DIST_NOTE"] == "why" and env["X"] == "1"
    assert env["MASTER_ADDR"] == "127.0.0.1"


def test_ranks_that_replace_themselves_meet_again_under_the_launcher():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GSPLAT_BENCH_FELL_BACK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "tests", "_reexec_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "SECOND_LIFE_OK simulated 29634" in r.stdout, r.stdout[-3000:]
