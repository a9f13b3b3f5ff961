"""The advisory pose hints `MPIRenderer.render` derives from the poses it draws on the host (renderer.py: `_draw_poses`, _COS_FRONTAL / _COS_OBLIQUE /
_COS_TILTED) and passes in GmpiRenderParams.flags: every camera axis within 0.2 rad of the MPI normal -> GMPI_FLAG_HINT_FRONTAL; some axis beyond 0.35 rad ->
GMPI_FLAG_HINT_OBLIQUE (views that share an MPI then stay on the tile kernel: include/gmpi_render.h); beyond 0.53 rad -> GMPI_FLAG_HINT_TILTED.  Results never
depend on the hints (tests/test_hip_band.py checks that on the device); this is the host arithmetic, no GPU."""
import math

import torch


def test_pose_hints_follow_the_camera_axes():
    import ml_gmpi_amd
    from ml_gmpi_amd import renderer as R
    assert abs(R._COS_FRONTAL - math.cos(0.2)) < 1e-7 and abs(R._COS_OBLIQUE - math.cos(0.35)) < 1e-7 and abs(R._COS_TILTED - math.cos(0.53)) < 1e-7
    V, D, S = 8, 4, 32
    r = ml_gmpi_amd.make_renderer("FFHQ", n_planes=D, device=torch.device("cpu"), ray_backend="torch", on_out_of_plane="raise")
    r.set_cam(r.cam_fov, S, S)
    want = {0.15: (True, False, False), 0.3: (False, False, False), 0.4: (False, True, False), 0.6: (False, True, True)}
    for lim, flags in want.items():
        for _ in range(2):   # (the second call answers out of the pose look-ahead / cache: the same hints)
            r._draw_poses(V, 0, 0, 0, 0, False, torch.linspace(lim, -lim, V).view(-1, 1), torch.zeros(V, 1))
            assert (bool(r._frontal), bool(r._oblique), bool(r._tilted)) == flags, (lim, r._frontal, r._oblique, r._tilted)
    # a pitch counts like a yaw: the hint is about the angle between the camera axis and the normal
    r._draw_poses(V, 0, 0, 0, 0, False, torch.zeros(V, 1), torch.linspace(0.4, -0.4, V).view(-1, 1))
    assert (bool(r._frontal), bool(r._oblique), bool(r._tilted)) == (False, True, False)
