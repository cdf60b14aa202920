import time, numpy as np, cProfile, pstats, torch
from bp_osd_amd.codes import h1922
from bp_osd_amd.sim import css_decode_sim
c = h1922()
pr = cProfile.Profile()
pr.enable()
r = css_decode_sim(hx=c.hx, hz=c.hz, error_rate=0.05, xyz_error_bias=[1,1,1], target_runs=65536, seed=1, bp_method="ms", ms_scaling_factor=0, max_iter=0, osd_method="osd_cs", osd_order=7, channel_update=None, tqdm_disable=1, batch_size=65536)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
