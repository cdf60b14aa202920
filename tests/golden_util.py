"""Loader for tests/golden/*.npz (see tests/golden/make_golden.py for provenance)."""
import ast
import glob
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files():
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(HERE, "golden_*.npz")))


def code_for(name):
    from bp_osd_amd.codes import surface13, h1922, hgp

    if name.startswith("golden_s13"):
        return surface13().hz
    if name.startswith("golden_hgp400"):
        seed = np.loadtxt(os.path.join(HERE, "mkmn_16_4_6.txt")).astype(np.uint8)
        return hgp(seed, compute_logicals=False).hx
    return h1922(compute_logicals=False).hz


def load(name):
    z = np.load(os.path.join(HERE, name))
    H = code_for(name)
    m, n = H.shape
    cfg = ast.literal_eval(str(z["cfg"]))
    out = {"cfg": cfg, "H": H}
    packed = z["syn"].shape[1] != m
    for k in ("syn", "osdw", "osd0", "bp"):
        a = z[k]
        if packed:
            width = m if k == "syn" else n
            a = np.unpackbits(a, axis=1)[:, :width]
        out[k] = a.astype(np.uint8)
    out["converged"] = z["converged"]
    out["iters"] = z["iters"]
    out["llr"] = z["llr"]  # first rows only
    return out
