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


def _run(world, workload, port, steps=3, warmup=1, extra=(), env_extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup),
           "--workload", workload, "--dry-run", "--dry-size", "32", *extra]
    env = dict(os.environ, OMP_NUM_THREADS="1", **(env_extra or {}))
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
    # the job's views and the bytes of its ONE collective at full size -- configs[3]: 64 views, 8 x (8 x 4 x 512^2 x 4 B) = 8 x 33.5 MB;
    # configs[4]: 32 seeds, 8 x (4 x 5 x 1024^2 x 4 B) = 8 x 84 MB (config 3 sharded the same way: 32 views, 8 x 67 MB)
    want = {"cfg3": (32, 8 * 4 * 4 * 1024 * 1024 * 4), "cfg4": (64, 8 * 33554432), "cfg5": (32, 8 * 83886080)}[workload]
    assert (line["config"]["views_total"], rc["gather_bytes_at_full_size"]) == want


def _fake_sysfs(root, gpus_per_node=4, nodes=2, cpus_per_node=16):
    """A sysfs tree as an 8-GPU MI355X node shows it: PCI devices with a numa_node file, NUMA nodes with a cpulist."""
    pcis = []
    for g in range(gpus_per_node * nodes):
        pci = f"0000:{0x05 + 0x10 * g:02x}:00.0"
        d = os.path.join(root, "bus/pci/devices", pci)
        os.makedirs(d)
        open(os.path.join(d, "numa_node"), "w").write(f"{g // gpus_per_node}\n")
        pcis.append(pci.upper())   # (the runtime prints upper-case hex; sysfs is lower-case)
    for n in range(nodes):
        d = os.path.join(root, f"devices/system/node/node{n}")
        os.makedirs(d)
        lo = n * cpus_per_node
        open(os.path.join(d, "cpulist"), "w").write(f"{lo}-{lo + cpus_per_node // 2 - 1},{lo + 64}-{lo + 64 + cpus_per_node // 2 - 1}\n")
    return pcis


def test_numa_lookup_parses_sysfs(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    pcis = _fake_sysfs(str(tmp_path))
    node, ids, cpus = bench.numa_cpus_of_pci(pcis[5], str(tmp_path))
    assert node == 1 and cpus == "16-23,80-87" and ids == set(range(16, 24)) | set(range(80, 88))
    os.makedirs(tmp_path / "bus/pci/devices/0000:ff:00.0")
    (tmp_path / "bus/pci/devices/0000:ff:00.0/numa_node").write_text("-1\n")
    assert bench.numa_cpus_of_pci("0000:FF:00.0", str(tmp_path)) == (-1, None, None)


@pytest.mark.skipif(len(os.sched_getaffinity(0)) < 4, reason="needs a few cores")
def test_bench_cfg5_at_world_8_with_numa_pin(tmp_path):
    """`bench.py --workload cfg5 --numa-pin` under the launcher at world 8: every rank looks its GPU's NUMA node up (a fake sysfs tree: 8 GPUs on two
    nodes) and the line carries all eight bindings; rank-local state is independent (each rank its own volumes, outputs, status words: nothing shared
    but the final gather)."""
    world = 8
    pcis = _fake_sysfs(str(tmp_path))
    line = _run(world, "cfg5", 29611, extra=("--numa-pin",), env_extra={"GMPI_BENCH_SYSFS": str(tmp_path), "GMPI_BENCH_PCI_IDS": ",".join(pcis)})
    pins = line["numa_pin"]
    assert isinstance(pins, list) and len(pins) == world
    for rank, text in enumerate(pins):
        assert text.startswith(f"gpu {rank} ({pcis[rank]}) -> NUMA node {rank // 4}, 16 cpus"), text
    assert line["config"]["views_total"] == 32 and line["rccl"]["gather_bytes_at_full_size"] == 8 * 83886080


def test_bench_single_process_dry_run():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dry-run", "--dry-size", "32"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["gather_ms"] is None and line["dry_run"]["gathered_shape"] is None
    assert "rccl" not in line and "ms_per_step_ranks" not in line and line["roofline"]["frac_pose_mean"] is None
