"""Multi-process (world_size 2 and 3, gloo, CPU) tests of the view-sharding driver: the partition is
the reference's rank-strided counter (fid_evaluation.py:86,133), every rank renders only its shard,
and ONE all_gather assembles the frames in view order -- bit-identical to a single-process run.
The render function is injected (the CPU oracle stands in for the device kernel here; tests may use
the oracle, the product never does)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ml_gmpi_amd.driver import gather_frames, render_views_sharded, shard_views


def test_shard_views_partition():
    for n in (0, 1, 7, 64):
        for w in (1, 2, 3, 8):
            for mode in ("strided", "block"):
                parts = [shard_views(n, r, w, mode) for r in range(w)]
                flat = sorted(i for p in parts for i in p)
                assert flat == list(range(n)), (n, w, mode)
    assert shard_views(10, 1, 4, "strided") == [1, 5, 9]  # img_counter = rank; += world_size
    assert shard_views(10, 3, 4, "block") == [9]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frames_for(indices):
    """Deterministic 'render': oracle on tiny per-view inputs derived from the view index."""
    import oracle
    outs = []
    for i in indices:
        rgba = oracle.synth_rgba(100 + i, (1, 3, 4, 8, 8))
        dhw = np.array([[[0.95, 0.25, 0.25], [1.0, 0.25, 0.25], [1.12, 0.5, 0.5]]], np.float32)
        ys, xs = np.meshgrid(np.linspace(-0.1, 0.1, 6), np.linspace(-0.1, 0.1, 6), indexing="ij")
        ray = np.stack([xs + 0.01 * i, ys, np.ones_like(xs)]).astype(np.float32)[None]
        ray /= np.linalg.norm(ray, axis=1, keepdims=True)
        o = oracle.render(rgba, dhw, ray, np.zeros((1, 3), np.float32), np.array([[0, 0, 1]], np.float32))
        outs.append(np.concatenate([o["color"], o["depth"]], axis=1))
    if not outs:
        return torch.zeros((0, 4, 6, 6))
    return torch.from_numpy(np.concatenate(outs, 0))


def _worker(rank, world, port, n_views, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = render_views_sharded(_frames_for, n_views, rank, world, mode=mode, gather=True)
        local, idx = render_views_sharded(_frames_for, n_views, rank, world, mode=mode, gather=False)
        q.put((rank, full.numpy(), idx, local.shape[0]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_views,mode", [(2, 6, "strided"), (2, 5, "block"), (3, 7, "strided")])
def test_sharded_render_equals_single_process(world, n_views, mode):
    import sys
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _frames_for(list(range(n_views))).numpy()
    seen = []
    for rank, full, idx, n_local in got:
        assert np.array_equal(full, want), rank           # every rank holds the full, ordered sequence
        assert idx == shard_views(n_views, rank, world, mode) and n_local == len(idx)
        seen += idx
    assert sorted(seen) == list(range(n_views))            # each view rendered exactly once


def test_gather_single_process_path():
    local = _frames_for([0, 2])
    out = gather_frames(local, [0, 1], 2)
    assert torch.equal(out, local)


def test_dump_frames_writes_the_reference_layout(tmp_path):
    """prepare_fake_data.py:184-190, 226-258: <save_dir>/<task>/{rgb,angle,depth}/<idx:06d>[_<j>].{png,npy}."""
    import numpy as np
    from PIL import Image
    import ml_gmpi_amd
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(4, 8, 8, 3), dtype=np.uint8)
    ang = rng.random((4, 2)).astype(np.float32)
    dep = rng.random((4, 1, 8, 8)).astype(np.float32)
    paths = ml_gmpi_amd.dump_frames(str(tmp_path), "fid", 10, img, ang, depth=dep)
    assert [os.path.basename(p) for p in paths] == ["000010.png", "000011.png", "000012.png", "000013.png"]
    assert np.array_equal(np.asarray(Image.open(paths[2])), img[2])
    assert np.array_equal(np.load(tmp_path / "fid" / "angle" / "000012.npy"), ang[2])
    assert np.array_equal(np.load(tmp_path / "fid" / "depth" / "000013.npy"), dep[3, 0])
    paths = ml_gmpi_amd.dump_frames(str(tmp_path), "consistency", 0, torch.from_numpy(img), torch.from_numpy(ang), n_view_per_z=2)
    assert [os.path.basename(p) for p in paths] == ["000000_0.png", "000000_1.png", "000001_0.png", "000001_1.png"]
    assert not (tmp_path / "consistency" / "depth" / "000000_0.npy").exists()
