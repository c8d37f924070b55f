"""bench.py --gpus N degrades in steps: an attempt that fails or does not finish is replaced, rank by rank and in place, by
the next stage of bench.DIST_STAGES — group×3 (three communicators in flight per rank) -> group×1 (one communicator, the
grouped-broadcast gather) -> the torch host (bench.restart_next_stage).  The part that can be checked without a GPU: the
stages' environments, and that under the driver's launcher processes that replace themselves find each other again."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_second_attempt_gets_a_rendezvous_of_its_own():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.DIST_STAGES == ["group", "group1", "torch"]
    env = bench.fallback_env({"MASTER_PORT": "29500", "TORCHELASTIC_USE_AGENT_STORE": "True", "X": "1"}, "why", "group1")
    assert env["MASTER_PORT"] == "29523" and env["TORCHELASTIC_USE_AGENT_STORE"] == "False"
    assert env["GSPLAT_BENCH_FELL_BACK"] == "1" and env["GSPLAT_BENCH_DIST_NOTE"] == "why" and env["X"] == "1"
    assert env["MASTER_ADDR"] == "127.0.0.1"
    # the middle stage: still the product path, with the most conservative use of RCCL it has
    assert env["GSPLAT_BENCH_STAGE"] == "group1" and env["GSPLAT_MULTI_IN_FLIGHT"] == "1" and env["GSPLAT_GROUP_GATHER"] == "broadcast"
    assert bench.current_stage("group") == "group" and bench.current_stage("torch") == "torch"
    # the last stage: another rendezvous again, the reasons of both earlier stages in the note
    env2 = bench.fallback_env(env, "why again", "torch")
    assert env2["MASTER_PORT"] == "29546" and env2["GSPLAT_BENCH_STAGE"] == "torch"
    assert env2["GSPLAT_BENCH_DIST_NOTE"] == "why | why again"


def test_probe_switches_do_not_ship():
    """A switch that changes RESULTS (not speed) must not be readable from the environment by the shipped library: diagnosis
    switches named GSPLAT_PROBE_* live under #ifdef GSPLAT_TEST_HOOKS only (GSPLAT_PROBE_RING_NO_COPY skipped the
    read-back copy and handed the host stale memory)."""
    import glob
    import re
    bad = []
    for path in sorted(glob.glob(os.path.join(ROOT, "godotgaussiansplatting_amd", "csrc", "*"))):
        stack = []     # one entry per open #if: True while inside `#ifdef GSPLAT_TEST_HOOKS` (before its #else)
        for ln, line in enumerate(open(path, errors="replace"), 1):
            t = line.strip()
            if re.match(r"#\s*if", t):
                stack.append(bool(re.match(r"#\s*ifdef\s+GSPLAT_TEST_HOOKS\b", t)))
            elif re.match(r"#\s*else", t) and stack:
                stack[-1] = False
            elif re.match(r"#\s*endif", t) and stack:
                stack.pop()
            if 'getenv("GSPLAT_PROBE' in line and not any(stack):
                bad.append(f"{os.path.basename(path)}:{ln}")
    assert not bad, f"probe switches outside #ifdef GSPLAT_TEST_HOOKS: {bad}"


def test_ranks_that_replace_themselves_meet_again_under_the_launcher():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GSPLAT_BENCH_FELL_BACK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", os.path.join(ROOT, "tests", "_reexec_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "SECOND_LIFE_OK simulated 29634" in r.stdout, r.stdout[-3000:]
