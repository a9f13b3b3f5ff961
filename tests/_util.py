"""Shared helpers for the test-suite: golden-fixture loading and synthetic inputs."""
import glob
import json
import os

import numpy as np

import oracle  # oracle/oracle.py (tests may use the oracle; the product never does)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def render_fixture_names():
    return sorted(os.path.basename(p)[len("render_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "render_*.npz")))


def load_render_fixture(name):
    z = np.load(os.path.join(GOLDEN, f"render_{name}.npz"))
    meta = json.loads(str(z["meta"]))
    Ht, Wt = meta["tex"]
    rgba = oracle.synth_rgba(meta["seed"], (meta["B"], meta["D"], 4, Ht, Wt),
                             last_alpha_one=meta.get("last_alpha_one", False), bf16_round=meta.get("bf16", False))
    if meta.get("alpha_binary"):
        rgba[:, :, 3] = (rgba[:, :, 3] > 0.5).astype(np.float32)
    d = {k: z[k] for k in z.files if k != "meta"}
    d["rgba"] = rgba
    d["meta"] = meta
    return d


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    d = {k: z[k] for k in z.files if k != "meta"}
    d["meta"] = json.loads(str(z["meta"]))
    return d
