"""Texel-box statistics of pixel tiles (CPU): for the camera sets of tools/kbench_dump.py, the box a TW x TH pixel tile needs on every plane
(corner pixels, as the kernels compute it), aligned to `align` texels.   python tools/box_stats.py [set] [TW] [TH] [align]"""
import sys, numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "bench"
TW = int(sys.argv[2]) if len(sys.argv) > 2 else 32
TH = int(sys.argv[3]) if len(sys.argv) > 3 else 16
AL = int(sys.argv[4]) if len(sys.argv) > 4 else 8
f = open(f"gpurun_in/kb_{name}.bin", "rb")
N, S, D = np.fromfile(f, np.int32, 3)
focal = np.fromfile(f, np.float32, 1)[0]
dhw = np.fromfile(f, np.float32, D * 3).reshape(D, 3)
c2w = np.fromfile(f, np.float32, N * 16).reshape(N, 4, 4)
xs = np.arange(0, S, TW); ys = np.arange(0, S, TH)
res = []
for n in range(N):
    R = c2w[n, :3, :3].astype(np.float64); eye = c2w[n, :3, 3].astype(np.float64)
    def coords(px, py, k):
        d = np.stack([(px + 0.5 - S / 2) / focal, (py + 0.5 - S / 2) / focal, np.ones_like(px, dtype=np.float64)], -1)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        r = d @ R.T
        s = (dhw[k, 0] - eye[2]) / r[..., 2]
        x = eye[0] + r[..., 0] * s; y = eye[1] + r[..., 1] * s
        ix = (2 * x / dhw[k, 2] + 1) * (S - 1) / 2; iy = (2 * y / dhw[k, 1] + 1) * (S - 1) / 2
        return ix, iy
    X0, Y0 = np.meshgrid(xs, ys)
    for k in range(D):
        cx = []; cy = []
        for dx in (0, TW - 1):
            for dy in (0, TH - 1):
                ix, iy = coords((X0 + dx).astype(np.float64), (Y0 + dy).astype(np.float64), k)
                cx.append(ix); cy.append(iy)
        cx = np.stack(cx); cy = np.stack(cy)
        bx0 = np.floor(cx.min(0) - 1 / 64); bx1 = np.floor(cx.max(0) + 1 / 64) + 1
        by0 = np.floor(cy.min(0) - 1 / 64); by1 = np.floor(cy.max(0) + 1 / 64) + 1
        q0 = np.floor(bx0 / AL) * AL
        nq = np.floor((bx1 - q0) / AL) + 1
        rows = by1 - by0 + 1
        res.append((n, k, nq, rows, bx1 - bx0 + 1))
nq = np.stack([r[2] for r in res]).reshape(N, D, -1); rows = np.stack([r[3] for r in res]).reshape(N, D, -1); wid = np.stack([r[4] for r in res]).reshape(N, D, -1)
print(f"set {name}: {N} views {S}^2 x {D}, tile {TW}x{TH}, align {AL}")
for n in range(N):
    print(f" view {n}: width {wid[n].min():.0f}-{wid[n].max():.0f} (mean {wid[n].mean():.1f}) items {nq[n].min():.0f}-{nq[n].max():.0f} (mean {nq[n].mean():.2f}) rows {rows[n].min():.0f}-{rows[n].max():.0f} (mean {rows[n].mean():.2f});"
          f" per-tile max over planes: items mean {nq[n].max(0).mean():.2f} rows mean {rows[n].max(0).mean():.2f}, frac tiles rows>18: {(rows[n].max(0) > 18).mean():.2f}")
print(f" staged texels per pixel (exact boxes): {(wid * rows).mean() / (TW * TH):.3f};  with item alignment: {(nq * AL * rows).mean() / (TW * TH):.3f};  with per-tile max box: {(nq.max(1, keepdims=True) * AL * rows.max(1, keepdims=True)).mean() / (TW * TH):.3f}")
