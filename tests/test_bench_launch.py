"""`python bench.py --gpus N` starts its own ranks: the launch path (children through torch.distributed.run,
rendezvous on 127.0.0.1, one JSON line from rank 0, the children's exit status) and the WHOLE control flow of a
run (bench.run_legs: warm-up, timed region, rasteriser / REINFORCE / actor-critic legs, the per-kernel timing
steps, the extra legs) checked here with two gloo ranks and no GPU work (--launch-check): every leg is a stand-in
that issues the collectives the real leg issues, through the same parallel helpers, and the ranks compare what
they are about to issue before each one."""
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


def test_every_leg_ran_and_the_line_carries_the_exchange_facts():
    p = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-check")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["scaling"] == "weak" and rec["config"]["global_batch"] == 128 and rec["config"]["parallelism"] == "dp2"
    assert rec["allreduce_bytes"] > 0
    assert set(rec["extra_runs"]) == {"all_durations_T", "fresh_batches", "f32", "endtokendeep2", "decode",
                                      "recurrence_224_L5"}
    assert rec["iou"]["rollouts"] == 2 * 16384 and rec["reinforce"]["envs"] == 2 * 2048 and rec["a3c"]["envs"] == 2 * 1024
    # MAX over ranks of the stand-in legs' times (rank r reports 1 + r ms)
    assert rec["iou"]["ms_per_launch"] == 2.0 and rec["reinforce"]["ms_per_step"] == 2.0


def test_a_step_that_only_rank_0_takes_is_caught():
    """The fault this check exists for: a training step (two all-reduces) outside the timed region that only rank 0
    runs pairs its collectives with whatever the other ranks issue next."""
    p = _run({"ISG_BENCH_INJECT": "rank0_step"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-check")
    assert p.returncode != 0
    assert "ranks disagree on the next collective" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_strong_scaling_splits_the_global_batch():
    p = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "0", "--launch-check", "--scaling", "strong")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["config"]["global_batch"] == 64
    assert rec["iou"]["rollouts"] == 16384 and rec["reinforce"]["envs"] == 2048 and rec["a3c"]["envs"] == 1024
    p = _run({}, "--gpus", "2", "--launch-check", "--scaling", "strong", "--batch", "63")
    assert p.returncode != 0 and "does not divide" in p.stderr


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
