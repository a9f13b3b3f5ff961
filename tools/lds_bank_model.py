"""CPU model of the LDS bank conflicts of the band kernel's two-byte taps (profiles/r04_band_variants.txt item 15): for a camera pose, the dword
index of every pixel's taps in the staged box, 64 consecutive pixels of a row per wavefront, conflict cycles per instruction under a bank / pass
model.  32 banks x 32 lanes per pass reproduces the SQ_LDS_BANK_CONFLICT counts of the MI355X.   usage: python tools/lds_bank_model.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ml_gmpi_amd.renderer import MPIRenderer, PRESETS
def coords(S, D, yaw, pitch=0.0, views=1):
    kw = dict(PRESETS["FFHQ"])
    kw.update(n_mpi_planes=D, plan_spatial_enlarge_factor=1.001, plane_distances_sample_method="inverse", cam_sample_method="truncated_gaussian",
              mpi_align_corners=True, use_confined_volume=True, device=torch.device("cpu"))
    r = MPIRenderer(**kw); r.set_cam(r.cam_fov, S, S)
    cam = r.sample_cam_poses(1, 0, 0, 0, 0, False, given_yaws=torch.tensor([[yaw]]), given_pitches=torch.tensor([[pitch]]))
    ray = torch.cat(cam[3])[0].double().numpy(); eye = torch.cat(cam[4])[0].double().numpy()
    dhw = r.static_mpi_plane_dhws.reshape(-1, 3).double().numpy()
    out = []
    for k in range(0, D, max(D // 12, 1)):
        d, h, w = dhw[k]
        sc = (d - eye[2]) / ray[2]
        x = eye[0] + ray[0] * sc; y = eye[1] + ray[1] * sc
        ix = (2 * x / w + 1) * (S - 1) / 2; iy = (2 * y / h + 1) * (S - 1) / 2
        out.append((np.floor(ix).astype(np.int64), np.floor(iy).astype(np.int64)))
    return out
def conflicts(planes, pitch_dw, NB, L, S):
    tot = 0; n = 0
    for ix0, iy0 in planes:
        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
            X = ix0 + dx; Y = iy0 + dy
            D = Y * pitch_dw + (X >> 1)            # dword index (bf16: 2 texels per dword)
            Dw = D.reshape(S, S // 64, 64)          # waves: 64 consecutive pixels of a row
            for p0 in range(0, 64, L):
                g = Dw[:, :, p0:p0 + L]
                bank = g % NB
                # conflict cycles = max over banks of distinct dwords in that bank, minus 1
                srt = np.sort(g, axis=2)
                # count distinct dwords per bank: brute force via loop over banks is slow; use trick: for each lane, count lanes with same bank and different dword
                cyc = np.ones(g.shape[:2], dtype=np.int64)
                for b in range(NB):
                    m = bank == b
                    # distinct dwords among lanes in bank b
                    vals = np.where(m, g, -1)
                    vs = np.sort(vals, axis=2)
                    distinct = ((vs[:, :, 1:] != vs[:, :, :-1]) & (vs[:, :, 1:] >= 0)).sum(axis=2) + (vs[:, :, 0] >= 0)
                    cyc = np.maximum(cyc, distinct)
                tot += (cyc - 1).sum(); n += cyc.size / (64 // L)
    return tot / n
S = 512
for yaw in (0.0, 0.3, 0.45):
    pl = coords(S, 96, yaw)
    for NB, L in ((64, 32), (32, 32), (64, 64), (32, 16), (64, 16)):
        print('yaw', yaw, 'banks', NB, 'lanes/pass', L, 'pitch160 %.3f' % conflicts(pl, 160, NB, L, S), 'pitch176 %.3f' % conflicts(pl, 176, NB, L, S))
print('---- pitch sweep, 32 banks / 32 lanes per pass')
sets = {y: coords(S, 96, y) for y in (0.15, 0.3, 0.45)}
sets['-0.3/p0.1'] = coords(S, 96, -0.3, 0.1)
for pad in range(0, 32, 4):
    print('pitch', 160 + pad, 'P', (160 + pad) % 32, {k: round(conflicts(v, 160 + pad, 32, 32, S), 3) for k, v in sets.items()})


def conflicts_interleaved(planes, pitch_dw, NB, L, S):
    """The same with lane l of a wave on pixel 2 * (l % 32) + l // 32 of its 64: a pass of 32 lanes = every other pixel = 32 dwords = all banks."""
    tot = 0; n = 0
    perm = np.array([2 * (l % 32) + l // 32 for l in range(64)])
    for ix0, iy0 in planes:
        for dx, dy in ((0, 0), (1, 0), (0, 1), (1, 1)):
            D = ((iy0 + dy) * pitch_dw + ((ix0 + dx) >> 1)).reshape(S, S // 64, 64)[:, :, perm]
            for p0 in range(0, 64, L):
                g = D[:, :, p0:p0 + L]
                bank = g % NB
                cyc = np.ones(g.shape[:2], dtype=np.int64)
                for b in range(NB):
                    vs = np.sort(np.where(bank == b, g, -1), axis=2)
                    distinct = ((vs[:, :, 1:] != vs[:, :, :-1]) & (vs[:, :, 1:] >= 0)).sum(axis=2) + (vs[:, :, 0] >= 0)
                    cyc = np.maximum(cyc, distinct)
                tot += (cyc - 1).sum(); n += cyc.size / (64 // L)
    return tot / n


print('---- lanes interleaved over the 64 pixels (32 banks, 32 lanes per pass, pitch 160 dwords)')
print({k: (round(conflicts(v, 160, 32, 32, S), 3), round(conflicts_interleaved(v, 160, 32, 32, S), 3)) for k, v in sets.items()})
