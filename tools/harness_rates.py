"""Rates of the batched Monte-Carlo harness (bp_osd_amd.sim.css_decode_sim) on the [[1922,50]] code, p = 0.05 depolarising:
default numpy engine, torch engine fed numpy's random stream (identical counters), torch engine with the device RNG."""
import os, sys, time
import numpy as np
import torch  # noqa: F401  (before the decoder: INTEGRATION.md)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bp_osd_amd.codes import h1922
from bp_osd_amd.sim import css_decode_sim

c = h1922()
base = dict(hx=c.hx, hz=c.hz, error_rate=0.05, xyz_error_bias=[1, 1, 1], seed=1, bp_method="ms", ms_scaling_factor=0, max_iter=0,
            osd_method="osd_cs", osd_order=7, tqdm_disable=1)
for cu in (None, "x->z"):
    for engine, rng, B, nb in (("numpy", "numpy", 65536, 2), ("torch", "numpy", 65536, 3), ("torch", "torch", 131072, 8)):
        sim = css_decode_sim(target_runs=B, batch_size=B, channel_update=cu, engine=engine, rng=rng, **base)  # warm-up batch
        t0 = time.time()
        sim.target_runs = B * (nb + 1)
        sim.run_decode_sim()
        dt = time.time() - t0
        print(f"channel_update={cu!s:5} engine={engine:5} rng={rng:5}: {nb * B / dt:10.0f} runs/s   "
              f"LER {sim.osdw_logical_error_rate:.2e} +- {sim.osdw_logical_error_rate_eb:.1e} after {sim.run_count} runs", flush=True)
