"""`MPIRenderer.render` looks at the status bits of a call one or a few calls late (hip_mpi._StatusRing): the host never blocks on the render
kernel, and every assertion of the reference (mpi.py:70-72, 103-128, 185-187; mpi_renderer.py:447-449) still surfaces -- at a later call on
the stream, at `flush_status()`, or at interpreter exit -- with the diagnostics of the call that tripped it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(D=6, S=64, B=2, **kw):
    import ml_gmpi_amd
    dev = torch.device("cuda:0")
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
