"""Packs the logical-operator matrices the reference ships for its three example codes
(/root/reference/examples/codes/hgp_codes/hgp_(4,7)-[[N,K,d]]_{lx,lz}.txt -- data, read as text) into one small .npz,
so that tests/test_oracle.py can check this repository's code construction against them on any machine.

    python tests/golden/make_reference_logicals_fixture.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/examples/codes/hgp_codes"
out = {}
for N, K, d in ((400, 16, 6), (625, 25, 8), (900, 36, 10)):
    for op in ("lx", "lz"):
        a = np.loadtxt(os.path.join(SRC, f"hgp_(4,7)-[[{N},{K},{d}]]_{op}.txt")).astype(np.uint8)
        assert a.shape == (K, N), a.shape
        out[f"{op}_{N}"] = np.packbits(a, axis=1)
np.savez_compressed(os.path.join(HERE, "hgp_reference_logicals_fixture.npz"), **out)
print({k: v.shape for k, v in out.items()})
