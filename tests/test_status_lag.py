"""`MPIRenderer.render` asserts in the call that trips the assertion by default (`status_mode="sync"`: the reference's timing, mpi.py:70-72,
103-128, 185-187; mpi_renderer.py:447-449).  With `status_mode="lag"` / `defer_status="lag"` (opt-in) it looks at the status bits of a call one
or a few calls late (hip_mpi._StatusRing): the host never blocks on the render kernel, and every assertion still surfaces -- at a later call on
the stream, at `flush_status()`, or at interpreter exit (exit status 1) -- with the diagnostics of the call that tripped it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(D=6, S=64, B=2, **kw):
    import ml_gmpi_amd
    dev = torch.device("cuda:0")
    kw.setdefault("status_mode", "lag")   # (what this file is about; the default is "sync")
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=dev, on_out_of_plane="raise", **kw)
    rgba = torch.rand((B, D, 4, S, S), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    return ml_gmpi_amd, r, rgba, S


def test_assertion_surfaces_at_flush_not_at_the_call():
    m, r, rgba, S = _setup()
    m.flush_status()
    bad = rgba.clone()
    bad[1, 2, 3] = 1.5
    with torch.no_grad():
        good = r.render(rgba, S, S)[0].clone()
        out = r.render(bad, S, S)              # does not raise: nobody has looked yet
        assert out[0].shape == good.shape
        with pytest.raises(AssertionError, match="alpha to be within"):
            m.flush_status()
        m.flush_status()                       # raised once, then clean
        torch.manual_seed(1)
        a = r.render(rgba, S, S)[0]
        torch.manual_seed(1)
        b = r.render(rgba, S, S, defer_status=False)[0]   # the reference's timing on request; same pixels
        assert torch.equal(a, b)
        with pytest.raises(AssertionError, match="alpha to be within"):
            r.render(bad, S, S, defer_status=False)
    m.flush_status()


def test_a_later_call_raises_what_an_earlier_one_asserted():
    m, r, rgba, S = _setup()
    m.flush_status()
    bad = rgba.clone()
    bad[0, 0, 0] = float("nan")   # (a whole channel image: the check covers the texels the render touches)
    with torch.no_grad():
        r.render(bad, S, S)
        torch.cuda.synchronize()               # the kernel and the copy of its status words are done: the next call finds the event complete
        with pytest.raises(AssertionError):
            r.render(rgba, S, S)
    m.flush_status()


def test_ring_wraps_and_sync_mode_raises_at_once():
    m, r, rgba, S = _setup()
    with torch.no_grad():
        for _ in range(50):                    # more calls than slots: the oldest is waited for, nothing leaks
            r.render(rgba, S, S)
    m.flush_status()
    from ml_gmpi_amd import hip_mpi
    assert all(len(ring.pending) == 0 and len(ring.free) == hip_mpi._RING_SLOTS for ring in hip_mpi._RINGS.values())
    m2, r2, rgba2, S2 = _setup(status_mode="sync")
    bad = rgba2.clone()
    bad[0, 1, 3] = -0.25
    with torch.no_grad(), pytest.raises(AssertionError, match="alpha to be within"):
        r2.render(bad, S2, S2)


def test_per_stream_state_is_bounded_and_thread_safe():
    """The workspace / status caches are LRU-bounded (a program that creates many streams does not leak a workspace per stream), and two host
    threads that render on ONE stream are serialised by that stream's lock (the C header forbids concurrent calls that share a workspace)."""
    import threading
    from ml_gmpi_amd import hip_mpi
    m, r, rgba, S = _setup(D=4, S=512, B=5)            # 5 views of 512^2: AUTO's band path -> a workspace per stream
    rgba = rgba.to(torch.bfloat16)
    with torch.no_grad():
        ref = r.render(rgba, S, S, given_yaws=torch.zeros(5, 1), given_pitches=torch.zeros(5, 1))[0].clone()
        for _ in range(12):                            # more streams than cache entries
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                out = r.render(rgba, S, S, given_yaws=torch.zeros(5, 1), given_pitches=torch.zeros(5, 1))[0]
            st.synchronize()
            assert torch.equal(out, ref)
    m.flush_status()
    assert len(hip_mpi._WORKSPACES) <= hip_mpi._MAX_STREAMS and len(hip_mpi._RINGS) <= hip_mpi._MAX_STREAMS
    errors = []

    def worker():
        try:
            r2 = m.make_renderer("FFHQ", n_planes=4, device=torch.device("cuda:0"), on_out_of_plane="raise")
            with torch.no_grad():
                for _ in range(20):
                    o = r2.render(rgba, S, S, given_yaws=torch.zeros(5, 1), given_pitches=torch.zeros(5, 1))[0]
                    torch.cuda.synchronize()
                    if not torch.equal(o, ref):
                        errors.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker) for _ in range(3)]   # all on the default stream of cuda:0
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    m.flush_status()
    assert not errors, errors[:3]


def test_default_is_the_reference_timing_also_through_install():
    """The default renderer -- and the classes install() hands to the reference's scripts -- raise in the call that trips the assertion."""
    import ml_gmpi_amd
    import importlib
    inst = importlib.import_module("ml_gmpi_amd.install")   # (the package exports the FUNCTION `install` under that name)
    dev = torch.device("cuda:0")
    D, S = 6, 64
    rgba = torch.rand((2, D, 4, S, S), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    bad = rgba.clone()
    bad[1, 2, 3] = 1.5
    kw = dict(ml_gmpi_amd.PRESETS["FFHQ"])
    kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
              mpi_align_corners=True, use_confined_volume=True, device=dev)
    for cls in (ml_gmpi_amd.MPIRenderer, inst._installed_classes("full")[1], inst._installed_classes("touched")[1]):
        r = cls(**kw)
        assert r.status_mode is False
        with torch.no_grad():
            r.render(rgba, S, S)
            with pytest.raises(AssertionError, match="alpha to be within"):
                r.render(bad, S, S)
            r.render(rgba, S, S)     # nothing left behind
    ml_gmpi_amd.flush_status()


def test_lagged_out_of_plane_reports_the_call_that_tripped_it(capsys):
    """A tripped assert_not_out_of_last_plane in call n, then a DIFFERENT pose in call n + 1 (same shapes: the renderer's reused ray buffers
    would have been overwritten): the message and the pos / dir print-out are call n's (mpi.py:105-128)."""
    import re
    m, r, rgba, S = _setup(D=4, S=64, B=1)
    m.flush_status()
    with torch.no_grad():
        # a pose far outside the range the planes were sized for: rays leave the last plane
        r.render(rgba, S, S, given_yaws=torch.full((1, 1), 1.2), given_pitches=torch.zeros(1, 1), random_pose=False)
        torch.cuda.synchronize()
        ref_r = m.make_renderer("FFHQ", n_planes=4, device=rgba.device, on_out_of_plane="raise", status_mode="sync")
        with pytest.raises(RuntimeError) as sync_err:
            ref_r.render(rgba, S, S, given_yaws=torch.full((1, 1), 1.2), given_pitches=torch.zeros(1, 1), random_pose=False)
        sync_out = capsys.readouterr().out
        with pytest.raises(RuntimeError) as lag_err:
            r.render(rgba, S, S, given_yaws=torch.zeros(1, 1), given_pitches=torch.zeros(1, 1), random_pose=False)   # frontal: in range by itself
            m.flush_status()
        lag_out = capsys.readouterr().out
    assert str(lag_err.value) == str(sync_err.value), (str(lag_err.value), str(sync_err.value))
    assert re.search(r"goes out of plane at .*(min|max) val", str(lag_err.value))
    pos = lambda t: t[t.index("pos:"):t.index("yaws:")] if "yaws:" in t else t[t.index("pos:"):]   # noqa: E731
    assert pos(lag_out) == pos(sync_out)          # the tilted call's eye position and corner ray, not the frontal call's
    m.flush_status()


def test_lagged_assertion_at_exit_is_exit_status_1(tmp_path):
    """Python ignores exceptions raised by atexit handlers (exit status 0): the handler reports and leaves with status 1 instead."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    early, logf = str(tmp_path / "early.txt"), str(tmp_path / "host.log")
    code = (
        "import atexit, logging, sys, torch\n"
        # a host program's own exit handler and log file, registered BEFORE this package is imported: they must still run / be flushed
        f"atexit.register(lambda: open({early!r}, 'w').write('written by an exit handler registered before the import'))\n"
        f"logging.basicConfig(filename={logf!r}, level=logging.INFO)\n"
        "logging.getLogger('host').info('host log line')\n"
        f"sys.path.insert(0, {root!r})\n"
        "import ml_gmpi_amd\n"
        "dev = torch.device('cuda:0')\n"
        "r = ml_gmpi_amd.make_renderer('FFHQ', n_planes=4, device=dev, on_out_of_plane='raise', status_mode='lag')\n"
        "rgba = torch.rand((1, 4, 4, 64, 64), device=dev)\n"
        "rgba[0, 1, 3] = 2.0\n"
        "with torch.no_grad():\n"
        "    r.render(rgba, 64, 64)\n"
        "print('script end')\n")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "script end" in res.stdout
    assert res.returncode == 1, (res.returncode, res.stderr[-500:])
    assert "alpha to be within" in res.stderr
    assert open(early).read().startswith("written by an exit handler")      # (os._exit would have skipped it)
    assert "host log line" in open(logf).read()
