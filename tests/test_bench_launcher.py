"""`python bench.py --gpus N` as the driver calls it, with no torchrun environment: the file has to start its own N
ranks (a child `python -m torch.distributed.run`), let rank 0's JSON line through and return the child's status.
Exercised on the CPU through --rank-probe, which stops after the ranks have met (gloo where no GPU is visible)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SPICE_DIST_BACKEND")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def test_launcher_is_a_no_op_for_one_gpu_and_inside_torchrun(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for k in ("WORLD_SIZE", "RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.launch_ranks_if_needed(["--gpus", "1", "--steps", "3"]) is None
    assert bench.launch_ranks_if_needed(["--steps", "3"]) is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.launch_ranks_if_needed(["--gpus", "4"]) is None    # already a rank: torchrun started it


def test_plain_invocation_with_two_gpus_starts_two_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-probe"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout      # rank 0's line and nothing else on stdout
    d = json.loads(lines[0])["dist"]
    assert d["world_size"] == 2 and d["all_reduce_of_ones"] == 2
    assert d["backend"] in ("gloo", "nccl")
    assert sorted(x["rank"] for x in d["ranks"]) == [0, 1]
    assert len({x["pid"] for x in d["ranks"]}) == 2 and os.getpid() not in {x["pid"] for x in d["ranks"]}
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr


def test_child_failure_is_the_exit_status():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-such-flag"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
