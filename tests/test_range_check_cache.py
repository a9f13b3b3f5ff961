"""range_check="full" (what `install()` gives the reference's scripts: the min/max over the WHOLE volume of mpi_renderer.py:447-449 /
mpi.py:185-187) runs its exhaustive pass once per unchanged volume, not once per call -- the reference's video loop (render_video.py:95-130)
renders 100 views of one MPI.  The library is replaced by a recorder (no GPU needed): what is tested is which launches reach the C ABI."""
import pickle

import torch

import ml_gmpi_amd
from ml_gmpi_amd import _lib


class _Recorder:
    records_only = True

    def __init__(self):
        self.calls = []

    def gmpi_mpi_render_launch(self, pref, stream):
        self.calls.append("render")
        return 0

    def gmpi_rgba_range_check_launch(self, ptr, dtype, count, status, stream):
        self.calls.append("range_check")
        return 0


def _setup(monkeypatch, **kw):
    rec = _Recorder()
    monkeypatch.setattr(_lib, "load_library", lambda: rec)
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=4, device=torch.device("cpu"), ray_backend="torch", range_check="full", **kw)
    return rec, r


def _render(r, vol, **kw):
    torch.manual_seed(0)
    return r.render(vol, 16, 16, **kw)


def test_full_check_runs_once_per_unchanged_volume(monkeypatch):
    rec, r = _setup(monkeypatch)
    vol = torch.rand(2, 4, 4, 8, 8)
    for _ in range(3):
        _render(r, vol)
    assert rec.calls == ["range_check", "render", "render", "render"]


def test_in_place_update_rearms_the_full_check(monkeypatch):
    rec, r = _setup(monkeypatch)
    vol = torch.rand(2, 4, 4, 8, 8)
    _render(r, vol)
    vol.add_(0.0)                      # any in-place operation bumps the version counter
    _render(r, vol)
    vol[0, 0, 3].mul_(1.0)             # ... also through a view
    _render(r, vol)
    _render(r, vol)
    assert rec.calls == ["range_check", "render"] * 3 + ["render"]


def test_another_tensor_or_layout_rearms_and_views_of_one_base_do_not(monkeypatch):
    rec, r = _setup(monkeypatch)
    vol = torch.rand(2, 4, 4, 8, 8)
    _render(r, vol)
    _render(r, vol.clone())            # another tensor
    assert rec.calls == ["range_check", "render"] * 2
    del rec.calls[:]
    _render(r, vol[:1])                # a view: another pointer range than the last pass covered
    _render(r, vol[:1])                # the same view again (a NEW view object of the same base): unchanged
    _render(r, vol[1:])                # the other half: not covered by the pass over vol[:1]
    assert rec.calls == ["range_check", "render", "render", "range_check", "render"]


def test_a_new_tensor_at_the_old_address_is_checked(monkeypatch):
    rec, r = _setup(monkeypatch)
    vol = torch.rand(2, 4, 4, 8, 8)
    _render(r, vol)
    hit = r.mpi._full_check_passed
    assert hit is not None and hit[0]() is vol
    ptr = vol.data_ptr()
    del vol
    assert hit[0]() is None            # the anchor is held weakly: the cache keeps no volume alive
    again = torch.rand(2, 4, 4, 8, 8)  # (the allocator may or may not reuse the address; either way it must be checked)
    _render(r, again)
    assert rec.calls == ["range_check", "render"] * 2, (ptr, again.data_ptr())


def test_deferred_calls_never_record_a_pass(monkeypatch):
    rec, r = _setup(monkeypatch)
    vol = torch.rand(2, 4, 4, 8, 8)
    _render(r, vol, defer_status=True)
    _render(r, vol, defer_status=True)
    assert rec.calls == ["range_check", "render"] * 2


def test_touched_mode_has_no_exhaustive_pass_and_module_pickles(monkeypatch):
    rec = _Recorder()
    monkeypatch.setattr(_lib, "load_library", lambda: rec)
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=4, device=torch.device("cpu"), ray_backend="torch")
    vol = torch.rand(2, 4, 4, 8, 8)
    _render(r, vol)
    assert rec.calls == ["render"]
    full = ml_gmpi_amd.MPI(range_check="full")
    full._full_check_record(vol)
    clone = pickle.loads(pickle.dumps(full))
    assert clone._full_check_passed is None and clone.range_check == "full"
