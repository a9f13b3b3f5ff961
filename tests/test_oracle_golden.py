"""Pins the CPU oracle (oracle/mpi_oracle.c) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by oracle/make_golden.py, which imports the real
reference (gmpi/core/mpi_renderer.py:387 `MPIRenderer.render`, mpi.py:308 `MPI.forward`,
mpi.py:218 `MPI.old_forward`) on CPU fp32.  Tolerance: 1e-5 is the north-star bar; the oracle is
held to 2e-6 (observed <= 7.2e-7: the only differences are summation order / FMA use inside
torch.sum, einsum and grid_sampler, SURVEY.md section 7 hard part 1).
"""
import numpy as np
import pytest

import oracle
from _util import load_npz, load_render_fixture, render_fixture_names

TOL = 2e-6


@pytest.mark.parametrize("name", render_fixture_names())
def test_oracle_matches_reference_render(name):
    fx = load_render_fixture(name)
    out = oracle.render(fx["rgba"], fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"], align_corners=fx["meta"]["ac"])
    rgb_pm1 = 2.0 * out["color"] - 1.0  # mpi_renderer.py:467
    assert np.abs(rgb_pm1 - fx["ref_rgb_pm1"]).max() <= 2 * TOL
    assert np.abs(out["depth"] - fx["ref_depth"]).max() <= TOL
    # second formulation of the composite inside the reference (back-to-front loop)
    assert np.abs(out["color"] - fx["ref_old_color01"]).max() <= TOL
    assert np.abs(out["depth"] - fx["ref_old_depth"]).max() <= TOL
    assert out["status"] == 0  # reference rendered these without tripping an assert
    assert np.all(np.abs(out["uv_minmax"]) <= 1.0)


def test_oracle_matches_reference_forward_ragged_views():
    fx = load_npz("forward_ragged_views.npz")
    m = fx["meta"]
    rgba = oracle.synth_rgba(m["seed"], (m["M"], m["D"], 4, *m["tex"]))
    out = oracle.render(rgba, fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"], view_to_mpi=fx["view_to_mpi"],
                        align_corners=m["ac"])
    assert np.abs(out["color"] - fx["ref_color01"]).max() <= TOL
    assert np.abs(out["depth"] - fx["ref_depth"]).max() <= TOL


def test_oracle_threaded_build_is_bit_identical():
    fx = load_render_fixture("ffhq_d8_32_ac1")
    a = oracle.render(fx["rgba"], fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"], threads=False)
    b = oracle.render(fx["rgba"], fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"], threads=True)
    for k in ("color", "depth", "T", "uv_minmax"):
        assert np.array_equal(a[k], b[k])


def test_oracle_flags():
    fx = load_render_fixture("ffhq_d8_32_ac1")
    rgba = fx["rgba"].copy()
    rgba[0, 3, 1] = 1.5  # out of [0,1] on a whole channel plane -> certainly sampled
    out = oracle.render(rgba, fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"])
    assert out["status"] & oracle.STATUS_RGBA_RANGE
    assert oracle.range_check(rgba) == oracle.STATUS_RGBA_RANGE and oracle.range_check(fx["rgba"]) == 0
    dhw = fx["dhw"].copy()
    dhw[:, -1, 1:] *= 0.25  # shrink the last plane: rays must leave it (mpi.py:106-109)
    out = oracle.render(fx["rgba"], dhw, fx["ray_dir"], fx["eye"], fx["zdir"])
    assert out["status"] & oracle.STATUS_OUT_OF_LAST_PLANE
    dhw = fx["dhw"].copy()
    dhw[:, 0, 0] = -5.0  # plane behind the camera (mpi.py:70-72)
    out = oracle.render(fx["rgba"], dhw, fx["ray_dir"], fx["eye"], fx["zdir"])
    assert out["status"] & oracle.STATUS_CAMERA_BEHIND_PLANE


def test_oracle_transmittance_consistency():
    """T_final is the cumprod element the reference slices off (mpi.py:423); sum(weights)=1-T up to 1e-10 terms."""
    fx = load_render_fixture("ffhq_d4_alpha01")
    rgba = fx["rgba"].copy()
    rgba[:, :, :3] = 0.25  # constant colour: C = 0.25 * sum(w)
    out = oracle.render(rgba, fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"])
    assert np.abs(out["color"] - 0.25 * (1.0 - out["T"])).max() <= 1e-6


@pytest.mark.parametrize("name", ["backward_ffhq_d8_32", "backward_ffhq_d8_32_opaque", "backward_ffhq_d6_tex40_img24_ac0",
                                  "backward_forward_ragged_views"])
def test_oracle_matches_the_forward_of_the_backward_fixtures(name):
    """The gradient fixtures (reference autograd, train.py:740-779) also carry the reference's forward values: exactly and
    nearly opaque planes in the middle of the stack, a texture finer than the image with align_corners=False."""
    fx = load_npz(name + ".npz")
    v2m = fx.get("view_to_mpi")
    out = oracle.render(fx["rgba"], fx["dhw"], fx["ray_dir"], fx["eye"], fx["zdir"], view_to_mpi=v2m, align_corners=fx["meta"]["ac"])
    if "ref_rgb_pm1" in fx:
        assert np.abs(2.0 * out["color"] - 1.0 - fx["ref_rgb_pm1"]).max() <= 2 * TOL
    else:
        assert np.abs(out["color"] - fx["ref_color01"]).max() <= TOL
    assert np.abs(out["depth"] - fx["ref_depth"]).max() <= TOL


@pytest.mark.parametrize("name", render_fixture_names())
def test_torch_op_chain_matches_reference_render(name):
    """oracle/torch_ops.py (the reference's op chain as PyTorch calls: bench.py's `reference-ops` CPU baseline) reproduces
    the reference's own outputs."""
    import torch
    import torch_ops
    fx = load_render_fixture(name)
    t = torch.from_numpy
    rgb, depth = torch_ops.renderer_render(t(fx["rgba"]), t(fx["dhw"][0]), t(fx["ray_dir"]), t(fx["eye"]), t(fx["zdir"]),
                                           align_corners=fx["meta"]["ac"])
    assert np.abs(rgb.numpy() - fx["ref_rgb_pm1"]).max() <= 2 * TOL
    assert np.abs(depth.numpy() - fx["ref_depth"]).max() <= TOL
