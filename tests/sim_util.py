"""Test helpers for the Monte-Carlo harness: an oracle-backed decoder adapter and a literal serial
re-statement of the reference's per-shot loop (css_decode_sim.py:163-365).  TEST INFRASTRUCTURE ONLY."""
import numpy as np

from oracle import OracleDecoder


class OracleAdapter:
    """decode_batch interface of bp_osd_amd.BpOsdDecoder on top of the CPU oracle."""

    def __init__(self, pcm, channel_probs=None, **kw):
        self.base = np.array(channel_probs, dtype=float)
        self.dec = OracleDecoder(pcm, channel_probs=self.base, **kw)
        self.n = self.dec.n

    def update_channel_probs(self, p):
        self.base = np.array(p, dtype=float)
        self.dec.update_channel_probs(self.base)

    def decode_batch(self, syn, want_osd0=True, want_bp=True, want_llr=False, prior_select=None, alt_channel_probs=None):
        syn = np.asarray(syn, dtype=np.uint8)
        if prior_select is None:
            r = self.dec.decode_batch(syn, want_llr=False)
        else:
            outs = []
            alt = np.asarray(alt_channel_probs, dtype=float)
            for b in range(len(syn)):
                self.dec.update_channel_probs(np.where(np.asarray(prior_select[b]) != 0, alt, self.base))
                outs.append(self.dec.decode_batch(syn[b:b + 1], want_llr=False))
            self.dec.update_channel_probs(self.base)
            r = {k: np.concatenate([o[k] for o in outs]) for k in ("osdw", "osd0", "bp", "converged", "iters")}
        self.batch_osd0, self.batch_bp = r["osd0"], r["bp"]
        self.batch_converge = r["converged"].astype(bool)
        self.batch_iter = r["iters"]
        return r["osdw"]


def serial_reference_loop(hx, hz, lx, lz, *, error_rate, xyz_error_bias, target_runs, seed, channel_update,
                          bp_method, ms_scaling_factor, max_iter, osd_method, osd_order):
    """Shot-by-shot loop with the reference's control flow and RNG consumption (N scalar draws per shot)."""
    hx = np.asarray(hx.toarray() if hasattr(hx, "toarray") else hx, dtype=np.uint8)
    hz = np.asarray(hz.toarray() if hasattr(hz, "toarray") else hz, dtype=np.uint8)
    N = hx.shape[1]
    px, py, pz = error_rate * np.array(xyz_error_bias, dtype=float) / np.sum(xyz_error_bias)
    cpx, cpy, cpz = np.ones(N) * px, np.ones(N) * py, np.ones(N) * pz
    kw = dict(max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms_scaling_factor, osd_method=osd_method,
              osd_order=osd_order)
    bpd_z = OracleDecoder(hx, channel_probs=cpz + cpy, **kw)
    bpd_x = OracleDecoder(hz, channel_probs=cpx + cpy, **kw)
    np.random.seed(seed)
    c = dict(osdw_success_count=0, osd0_success_count=0, bp_success_count=0, bp_converge_count_x=0,
             bp_converge_count_z=0, min_logical_weight=N)  # css_decode_sim.py:142-145
    for _ in range(target_runs):
        ex, ez = np.zeros(N, dtype=np.uint8), np.zeros(N, dtype=np.uint8)
        for i in range(N):
            rand = np.random.random()
            if rand < cpz[i]:
                ez[i] = 1
            elif cpz[i] <= rand < cpz[i] + cpx[i]:
                ex[i] = 1
            elif cpz[i] + cpx[i] <= rand < cpx[i] + cpy[i] + cpz[i]:
                ez[i] = 1
                ex[i] = 1
        sz, sx = hx @ ez % 2, hz @ ex % 2
        if channel_update is None:
            rz, rx = bpd_z.decode(sz), bpd_x.decode(sx)
        elif channel_update == "x->z":
            rx = bpd_x.decode(sx)
            probs = np.zeros(N)
            for i in range(N):
                if rx["osdw"][i] == 1:
                    probs[i] = 0 if (cpx[i] + cpy[i]) == 0 else cpy[i] / (cpx[i] + cpy[i])
                else:
                    probs[i] = cpz[i] / (1 - cpx[i] - cpy[i])
            bpd_z.update_channel_probs(probs)
            rz = bpd_z.decode(sz)
        else:
            rz = bpd_z.decode(sz)
            probs = np.zeros(N)
            for i in range(N):
                if rz["osdw"][i] == 1:
                    probs[i] = 0 if (cpz[i] + cpy[i]) == 0 else cpy[i] / (cpz[i] + cpy[i])
                else:
                    probs[i] = cpx[i] / (1 - cpz[i] - cpy[i])
            bpd_x.update_channel_probs(probs)
            rx = bpd_x.decode(sx)
        for key in ("osdw", "osd0"):
            res_x, res_z = (ex + rx[key]) % 2, (ez + rz[key]) % 2
            if (lz @ res_x % 2).any():
                c["min_logical_weight"] = min(c["min_logical_weight"], int(res_x.sum()))
            elif (lx @ res_z % 2).any():
                c["min_logical_weight"] = min(c["min_logical_weight"], int(res_z.sum()))
            else:
                c[f"{key}_success_count"] += 1
        c["bp_converge_count_z"] += int(rz["converged"])
        c["bp_converge_count_x"] += int(rx["converged"])
        if rz["converged"] and rx["converged"]:
            res_x, res_z = (ex + rx["bp"]) % 2, (ez + rz["bp"]) % 2
            if not (lz @ res_x % 2).any() and not (lx @ res_z % 2).any():
                c["bp_success_count"] += 1
    return c
