"""`python bench.py --gpus N` typed as is must start N ranks itself (one process per GPU under
torch.distributed.run), form the process group and tear it down.  Checked here without a GPU: gloo backend,
two ranks, rendezvous only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=240, env=env, cwd=ROOT)


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text
    return json.loads(lines[-1])


def test_plain_invocation_spawns_one_rank_per_gpu():
    r = run("--gpus", "2", "--backend", "gloo", "--rendezvous-only")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = last_json(r.stdout)
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["rendezvous"] == "ok"


def test_single_rank_needs_no_launcher():
    r = run("--gpus", "1", "--backend", "gloo", "--rendezvous-only")
    assert r.returncode == 0, r.stderr[-2000:]
    assert last_json(r.stdout)["n_ranks_seen"] == 1


def test_child_failure_is_the_launcher_s_exit_status():
    # ranks started under a launcher whose world size disagrees with --gpus refuse to run; the parent reports it
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
