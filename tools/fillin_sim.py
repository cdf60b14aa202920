"""Fill-in of the L29k OSD elimination under different pivot-ROW policies (CPU model, tools/fillin_sim.c).

    python tools/fillin_sim.py [shots] [policies]      e.g.  python tools/fillin_sim.py 2 0,1,2

Takes the first non-converged syndromes of bench.py's l29k_ms_e15 batch 0 (BP by the CPU oracle, OSD off), writes the
matrix with its columns in reliability order and runs the model once per policy."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import make_batch  # noqa: E402
from bp_osd_amd.codes import l29k  # noqa: E402
from oracle import OracleDecoder  # noqa: E402

shots = int(sys.argv[1]) if len(sys.argv) > 1 else 2
policies = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3").split(",")]
exe = "/tmp/fillin_sim"
subprocess.check_call(["gcc", "-O3", "-march=native", "-o", exe, os.path.join(ROOT, "tools", "fillin_sim.c")])
H = l29k(compute_logicals=False).hz.tocsr()
m, n = H.shape
_, syn = make_batch(H, 0.05, 64, seed=0)
dec = OracleDecoder(H, error_rate=0.05, max_iter=100, bp_method="ms", ms_scaling_factor=0.625, osd_method="osd_off")
r = dec.decode_batch(syn, want_llr=True)
bad = np.flatnonzero(r["converged"] == 0)[:shots]
print("non-converged:", bad, flush=True)
for s in bad:
    order = np.argsort(r["llr"][s], kind="stable")
    inv = np.empty(n, np.int32)
    inv[order] = np.arange(n, dtype=np.int32)
    path = f"/tmp/fillin_{s}.bin"
    with open(path, "wb") as f:
        np.array([m, n], np.int32).tofile(f)
        for row in range(m):
            cols = np.sort(inv[H.indices[H.indptr[row]:H.indptr[row + 1]]]).astype(np.int32)
            np.array([len(cols)], np.int32).tofile(f)
            cols.tofile(f)
    procs = [(p, subprocess.Popen([exe, path, str(p)], stdout=subprocess.PIPE, text=True)) for p in policies]
    for p, pr in procs:
        print(f"shot {s}:", pr.communicate()[0].strip(), flush=True)
