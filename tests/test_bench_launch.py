"""`python bench.py --gpus N` starts its own ranks: the launch path (children through torch.distributed.run,
rendezvous on 127.0.0.1, the timed region's barriers and reductions, one JSON line from rank 0, the children's
exit status) checked here with two gloo ranks and no GPU work (--launch-check)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra_env, *argv):
    env = dict(os.environ, ISG_DIST_BACKEND="gloo", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_2_self_launches_two_ranks_and_prints_one_json_line():
    p = _run({}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["steps"] == 3 and rec["launch_check"]
    # the slower rank (2 ms per empty step) sets the time: MAX over ranks
    assert rec["ms_per_step"] >= 2.0


def test_single_rank_needs_no_launcher():
    p = _run({}, "--gpus", "1", "--steps", "2", "--launch-check")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["ranks_seen"] == 1


def test_a_failing_rank_fails_the_run():
    p = _run({"ISG_BENCH_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "1", "--launch-check")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "does not match" in p.stderr
