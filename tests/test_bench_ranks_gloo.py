"""bench.py's rank logic at the world size the driver's scaling run uses (8: one process per GPU of a node, run_gmpi.py:110 /
fid_evaluation.py:86-133 pattern) -- without GPUs: `--dry-run` swaps the library for a launch counter, the tensors live on the CPU and the
process group is gloo.  What this covers: the launcher contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), argument parsing, per-rank
seeding and the view shard of every rank, the barriers around the timed region, all_reduce(MAX) of the clocks, the shapes of the final
all_gather_into_tensor, and that rank 0 -- and only rank 0 -- prints ONE JSON line with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _run(world, workload, port, steps=3, warmup=1):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup),
           "--workload", workload, "--dry-run", "--dry-size", "32"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]     # rank 0 only
    return json.loads(lines[0])


@pytest.mark.skipif(len(os.sched_getaffinity(0)) < 4, reason="needs a few cores")
@pytest.mark.parametrize("workload,views,chan", [("cfg3", 4, 4), ("cfg4", 8, 4), ("cfg5", 4, 5)])
def test_bench_rank_logic_at_world_8(workload, views, chan):
    world = 8
    line = _run(world, workload, 29600 + {"cfg3": 0, "cfg4": 1, "cfg5": 2}[workload])
    assert KEYS <= set(line), sorted(KEYS - set(line))
    assert line["n_gpus"] == world and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == f"views sharded x{world}" and line["config"]["views_per_gpu"] == views
    assert line["data"].startswith("DRY RUN")
    # whole-job aggregate: `value` counts the units of ALL ranks over the max-over-ranks clock
    units = views * 32 * 32 * line["config"]["planes"] * world * line["steps"]
    assert abs(line["value"] - units / (line["ms_per_step"] * 1e-3 * line["steps"]) / 1e6) <= 0.02 * line["value"] + 1.0
    # every rank's frames in the gather: [world, views, 3 + 1 (+ 1 with the transmittance), H, W]; 1 pre-warm-free run = W + K + e2e launches
    assert line["dry_run"]["gathered_shape"] == [world, views, chan, 32, 32]
    assert line["dry_run"]["launches_on_rank0"] >= line["steps"] + line["warmup"]
    assert line["gather_ms"] is not None and line["cpu_baseline"] is None
    # round 5: what an N > 1 line says about its one collective and about stragglers (the headline clock is the max over ranks)
    rc = line["rccl"]
    assert rc["world"] == world and rc["backend"] == "gloo" and rc["gather_ms"] > 0 and rc["gather_gbs"] > 0
    assert rc["gather_bytes"] == world * views * chan * 32 * 32 * 4
    spread = line["ms_per_step_ranks"]
    assert len(spread["all"]) == world and spread["min"] <= spread["max"] <= line["ms_per_step"] * 1.0001 + 1e-3
    assert line["config"]["views_total"] == views * world
    if workload in ("cfg4", "cfg5"):   # the BASELINE 8-GPU configurations, named in the line
        assert "BASELINE configs" in line["config"]["workload"] and f"{views * world} views on {world} GPU" in line["config"]["workload"]


def test_bench_single_process_dry_run():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run", "--dry-size", "32"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["gather_ms"] is None and line["dry_run"]["gathered_shape"] is None
    assert "rccl" not in line and "ms_per_step_ranks" not in line and line["roofline"]["frac_pose_mean"] is None
