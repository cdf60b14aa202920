"""Batched Monte-Carlo harness: the MI355X counterpart of the reference's ``css_decode_sim``
(/root/reference/src/bposd/css_decode_sim.py:11-567).

Same kwargs, same defaults, same output keys and statistics definitions; the shot loop is replaced by
batches: all errors of a batch are drawn at once, both syndromes are computed as sparse products and the
two decoders are called with ``decode_batch``.  The per-shot Bayesian channel update ("x->z" / "z->x",
css_decode_sim.py:207-248) becomes the per-syndrome two-valued channel of ``decode_batch(prior_select=...)``.

Shot-for-shot reproducibility: the reference draws N numbers per shot from numpy's global legacy stream
after ``np.random.seed(seed)`` (css_decode_sim.py:135-138,471-472); ``np.random.random((B, N))`` consumes the
same stream in the same order, so with equal decoders this harness sees exactly the reference's errors.
What is NOT reproduced: tqdm, wall-clock based saving/early-stop points (the error-bar cut-off is checked
after every batch instead), the hadamard_rotate option's print-outs.

``engine="torch"`` keeps a whole batch on the GPU: errors, syndromes (sparse products), both decoders through their
device-pointer API (the two handles overlap when there is no channel update) and the logical checks, so that the
harness runs at the decoders' rate instead of numpy's (34 k runs/s on the [[1922,50]] code).  With ``rng="numpy"`` the
random numbers still come from numpy's legacy stream (uploaded), which makes the counters identical to the default
engine's; ``rng="torch"`` draws them on the device (seeded, but a different stream).  torch is plumbing here (device
buffers, sparse products); every decode runs in the HIP kernels either way.
"""
from __future__ import annotations

import datetime
import json
import time

import numpy as np
import scipy.sparse as sp

from .codes import CssCode

__all__ = ["css_decode_sim"]

_DEFAULT_INPUT = {  # css_decode_sim.py:65-84
    "error_rate": None,
    "xyz_error_bias": [1, 1, 1],
    "target_runs": 100,
    "seed": 0,
    "bp_method": "minimum_sum",
    "ms_scaling_factor": 0.625,
    "max_iter": 0,
    "osd_method": "osd_cs",
    "osd_order": 2,
    "save_interval": 2,
    "output_file": None,
    "check_code": 1,
    "tqdm_disable": 0,
    "run_sim": 1,
    "channel_update": "x->z",
    "hadamard_rotate": 0,
    "hadamard_rotate_sector1_length": 0,
    "error_bar_precision_cutoff": 1e-3,
}
_OUTPUT_VALUES = {  # css_decode_sim.py:94-115
    "K": None,
    "N": None,
    "start_date": None,
    "runtime": 0.0,
    "runtime_readable": None,
    "run_count": 0,
    "bp_converge_count_x": 0,
    "bp_converge_count_z": 0,
    "bp_success_count": 0,
    "bp_logical_error_rate": 0,
    "bp_logical_error_rate_eb": 0,
    "osd0_success_count": 0,
    "osd0_logical_error_rate": 0.0,
    "osd0_logical_error_rate_eb": 0.0,
    "osdw_success_count": 0,
    "osdw_logical_error_rate": 0.0,
    "osdw_logical_error_rate_eb": 0.0,
    "osdw_word_error_rate": 0.0,
    "osdw_word_error_rate_eb": 0.0,
    "min_logical_weight": 1e9,
}
_NOT_SAVED = ("channel_probs_x", "channel_probs_z", "channel_probs_y", "hx", "hz")


def _default_decoder_factory(pcm, **kw):
    from .decoder import BpOsdDecoder

    return BpOsdDecoder(pcm, **kw)


def _mod2_mul(A_csr, X, chunk=4096):
    """(A @ X^T mod 2)^T for uint8 X [B, n] -> C-contiguous uint8 [B, rows].

    Done in chunks of shots: the sparse product wants the shots as columns and hands the result back transposed, and a
    whole-batch transpose copy of a 65536 x 961 byte array is what used to dominate the harness (5.4 of 9.0 s per batch)."""
    B = X.shape[0]
    out = np.empty((B, A_csr.shape[0]), dtype=np.uint8)
    for lo in range(0, B, chunk):
        blk = np.ascontiguousarray(X[lo:lo + chunk].T, dtype=np.int32)
        out[lo:lo + chunk] = (np.asarray(A_csr @ blk) & 1).T
    return out


class css_decode_sim:
    """See the module docstring; parameters as documented at css_decode_sim.py:19-61, plus

    batch_size : int -- shots per decode_batch call (default 4096)
    decoder_factory : callable(pcm, **decoder_kwargs) -> object with ``decode_batch`` (default: the MI355X
        ``BpOsdDecoder``); tests inject a CPU-oracle adapter here.
    """

    def __init__(self, hx=None, hz=None, batch_size=4096, decoder_factory=None, engine="numpy", rng="numpy", **input_dict):
        if engine not in ("numpy", "torch") or rng not in ("numpy", "torch"):
            raise ValueError("engine and rng must be 'numpy' or 'torch'")
        if engine == "torch" and decoder_factory is not None:
            raise ValueError("engine='torch' drives the MI355X decoders through device pointers; decoder_factory must be None")
        if engine == "numpy" and rng == "torch":
            raise ValueError("rng='torch' needs engine='torch'")
        self._engine, self._rng = engine, rng
        for key, val in input_dict.items():  # css_decode_sim.py:87-91: anything passed overrides
            self.__dict__[key] = val
        for key, val in _DEFAULT_INPUT.items():
            if key not in input_dict:
                self.__dict__[key] = val
        for key, val in _OUTPUT_VALUES.items():  # resume: output keys present in the input win (:117-119)
            if key not in self.__dict__:
                self.__dict__[key] = val
        self.output_keys = [k for k in self.__dict__ if k not in _NOT_SAVED and not k.startswith("_")]
        self._batch_size = int(batch_size)
        self._factory = decoder_factory or _default_decoder_factory

        if self.seed == 0 or self.run_count != 0:  # css_decode_sim.py:135-137
            self.seed = int(np.random.randint(low=1, high=2 ** 32 - 1))
        np.random.seed(self.seed)

        if engine == "torch":
            # torch bundles its own HIP runtime and must initialise it BEFORE libbposd_mi355x.so pulls in the system
            # one: otherwise torch reports "No HIP GPUs are available" on the first batch
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        self.hx = sp.csr_matrix(hx).astype(np.uint8)
        self.hz = sp.csr_matrix(hz).astype(np.uint8)
        self._construct_code()
        self._error_channel_setup()
        self._decoder_setup()
        if self.run_sim:
            self.run_decode_sim()

    # ------------------------------------------------------------------ setup
    def _construct_code(self):
        qcode = CssCode(self.hx, self.hz)
        self.lx, self.lz, self.K, self.N = qcode.lx, qcode.lz, qcode.K, qcode.N
        if self.min_logical_weight == 1e9:  # css_decode_sim.py:142-145: the 1e9 placeholder becomes N before the run
            self.min_logical_weight = self.N
        if self.check_code and not qcode.test():
            raise Exception("Error: invalid CSS code. Check the form of your hx and hz matrices!")
        self._lx = sp.csr_matrix(self.lx)
        self._lz = sp.csr_matrix(self.lz)

    def _error_channel_setup(self):  # css_decode_sim.py:390-434
        bias = np.array(self.xyz_error_bias, dtype=float)
        if bias[0] == np.inf:
            self.px, self.py, self.pz = self.error_rate, 0, 0
        elif bias[1] == np.inf:
            self.px, self.py, self.pz = 0, self.error_rate, 0
        elif bias[2] == np.inf:
            self.px, self.py, self.pz = 0, 0, self.error_rate
        else:
            self.px, self.py, self.pz = self.error_rate * bias / np.sum(bias)
        N = self.N
        if self.hadamard_rotate == 0:
            self.channel_probs_x = np.ones(N) * self.px
            self.channel_probs_z = np.ones(N) * self.pz
            self.channel_probs_y = np.ones(N) * self.py
        elif self.hadamard_rotate == 1:
            n1 = self.hadamard_rotate_sector1_length
            self.channel_probs_x = np.hstack([np.ones(n1) * self.px, np.ones(N - n1) * self.pz])
            self.channel_probs_z = np.hstack([np.ones(n1) * self.pz, np.ones(N - n1) * self.px])
            self.channel_probs_y = np.ones(N) * self.py
        else:
            raise ValueError(f"The hadamard rotate attribute should be set to 0 or 1. Not '{self.hadamard_rotate}")

    def _decoder_setup(self):  # css_decode_sim.py:436-463
        self.ms_scaling_factor = float(self.ms_scaling_factor)
        kw = dict(max_iter=self.max_iter, bp_method=self.bp_method, ms_scaling_factor=self.ms_scaling_factor,
                  osd_method=self.osd_method, osd_order=self.osd_order)
        self.bpd_z = self._factory(self.hx, channel_probs=self.channel_probs_z + self.channel_probs_y, **kw)
        self.bpd_x = self._factory(self.hz, channel_probs=self.channel_probs_x + self.channel_probs_y, **kw)

    # ------------------------------------------------------------------ one batch
    def _generate_errors(self, B):  # css_decode_sim.py:465-498, vectorised over B shots
        rand = np.random.random((B, self.N))
        pz, px, py = self.channel_probs_z, self.channel_probs_x, self.channel_probs_y
        is_z = rand < pz
        is_x = (pz <= rand) & (rand < pz + px)
        is_y = (pz + px <= rand) & (rand < px + py + pz)
        error_z = (is_z | is_y).astype(np.uint8)
        error_x = (is_x | is_y).astype(np.uint8)
        return error_x, error_z

    @staticmethod
    def _decode(dec, syn, select=None, alt=None):
        if select is None:
            osdw = dec.decode_batch(syn, want_osd0=True, want_bp=True)
        else:
            osdw = dec.decode_batch(syn, want_osd0=True, want_bp=True, prior_select=select, alt_channel_probs=alt)
        return dict(osdw=osdw, osd0=dec.batch_osd0, bp=dec.batch_bp, conv=np.asarray(dec.batch_converge, dtype=bool))

    def _updated_channel(self, first_probs, other_probs):
        """css_decode_sim.py:217-227 / 236-246: the second decoder's per-bit probability given the first
        decoder's output bit: py / (p_first + py) where it is 1 (0 if the denominator is 0), and
        p_other / (1 - p_first - py) where it is 0."""
        py = self.channel_probs_y
        denom = first_probs + py
        with np.errstate(divide="ignore", invalid="ignore"):
            p_if_one = np.where(denom == 0, 0.0, py / denom)
        p_if_zero = other_probs / (1 - first_probs - py)
        return p_if_one, p_if_zero

    # ------------------------------------------------------------------ one batch, resident on the device
    def _torch_setup(self):
        import torch

        self._torch = torch
        dev = self._dev = torch.device("cuda", getattr(self.bpd_x, "device", 0))

        def sparse_t(a):
            a = sp.csr_matrix(a)
            return torch.sparse_csr_tensor(torch.from_numpy(a.indptr.astype(np.int64)), torch.from_numpy(a.indices.astype(np.int64)),
                                           torch.ones(a.nnz, dtype=torch.float32), size=a.shape).to(dev)

        self._t_hx, self._t_hz = sparse_t(self.hx), sparse_t(self.hz)
        self._t_lx = torch.from_numpy(np.asarray(self.lx, dtype=np.float32)).to(dev)
        self._t_lz = torch.from_numpy(np.asarray(self.lz, dtype=np.float32)).to(dev)
        self._t_p = [torch.from_numpy(np.asarray(v, dtype=np.float64)).to(dev)
                     for v in (self.channel_probs_z, self.channel_probs_x, self.channel_probs_y)]
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(int(self.seed))
        self._t_out = {}

    def _torch_outputs(self, name, B):
        torch = self._torch
        key = (name, B)
        if key not in self._t_out:
            mk = lambda: torch.empty((B, self.N), dtype=torch.uint8, device=self._dev)
            self._t_out[key] = dict(osdw=mk(), osd0=mk(), bp=mk(), conv=torch.empty(B, dtype=torch.uint8, device=self._dev))
        return self._t_out[key]

    def _torch_launch(self, dec, syn, out, select=None, alt=None):
        dec.decode_batch_device(syn.data_ptr(), syn.shape[0], out["osdw"].data_ptr(), out["osd0"].data_ptr(),
                                out["bp"].data_ptr(), out["conv"].data_ptr(), None, None,
                                d_prior_select=None if select is None else select.data_ptr(), alt_channel_probs=alt)

    def _run_batch_torch(self, B):
        if not hasattr(self, "_dev"):
            self._torch_setup()
        torch = self._torch
        if self._rng == "numpy":
            rand = torch.from_numpy(np.random.random((B, self.N))).to(self._dev)
        else:
            rand = torch.rand((B, self.N), dtype=torch.float64, device=self._dev, generator=self._gen)
        pz, px, py = self._t_p
        is_y = (pz + px <= rand) & (rand < px + py + pz)
        error_z = ((rand < pz) | is_y).to(torch.uint8)
        error_x = (((pz <= rand) & (rand < pz + px)) | is_y).to(torch.uint8)
        del rand, is_y

        def mod2(a, x):  # (A @ X^T mod 2)^T, uint8 [B, rows]; exact in float32 (row weights far below 2^24)
            return (torch.sparse.mm(a, x.to(torch.float32).T) if a.layout != torch.strided else a @ x.to(torch.float32).T
                    ).remainder_(2).T.contiguous().to(torch.uint8)

        synd_z, synd_x = mod2(self._t_hx, error_z), mod2(self._t_hz, error_x)
        torch.cuda.synchronize(self._dev)  # the decoders run on their own streams
        oz, ox = self._torch_outputs("z", B), self._torch_outputs("x", B)
        if self.channel_update is None:
            self._torch_launch(self.bpd_z, synd_z, oz)
            self._torch_launch(self.bpd_x, synd_x, ox)
        elif self.channel_update == "x->z":
            self._torch_launch(self.bpd_x, synd_x, ox)
            self.bpd_x.synchronize()
            p1, p0 = self._updated_channel(self.channel_probs_x, self.channel_probs_z)
            self.bpd_z.update_channel_probs(p0)
            self._torch_launch(self.bpd_z, synd_z, oz, select=ox["osdw"], alt=p1)
        elif self.channel_update == "z->x":
            self._torch_launch(self.bpd_z, synd_z, oz)
            self.bpd_z.synchronize()
            p1, p0 = self._updated_channel(self.channel_probs_z, self.channel_probs_x)
            self.bpd_x.update_channel_probs(p0)
            self._torch_launch(self.bpd_x, synd_x, ox, select=oz["osdw"], alt=p1)
        else:
            raise ValueError(f"channel_update='{self.channel_update}' is invalid")
        self.bpd_z.synchronize()
        self.bpd_x.synchronize()

        def logical_fail(dx, dz):  # css_decode_sim.py:257-272
            rx_, rz_ = error_x ^ dx, error_z ^ dz
            fx = (mod2(self._t_lz, rx_) != 0).any(dim=1)  # bool (torch.any of a uint8 tensor would stay uint8)
            fz = (mod2(self._t_lx, rz_) != 0).any(dim=1)
            weight = torch.where(fx, rx_.sum(dim=1, dtype=torch.int64), rz_.sum(dim=1, dtype=torch.int64))
            return fx, fz, weight

        self.run_count += B
        conv_z, conv_x = oz["conv"] != 0, ox["conv"] != 0
        self.bp_converge_count_z += int(conv_z.sum().item())
        self.bp_converge_count_x += int(conv_x.sum().item())
        fx, fz, _ = logical_fail(ox["bp"], oz["bp"])
        self.bp_success_count += int(((conv_z & conv_x) & ~(fx | fz)).sum().item())
        for key in ("osdw", "osd0"):
            fx, fz, weight = logical_fail(ox[key], oz[key])
            failed = fx | fz
            if bool(failed.any().item()):
                wmin = int(weight[failed].min().item())
                if wmin < self.min_logical_weight:
                    self.min_logical_weight = wmin
            setattr(self, f"{key}_success_count", getattr(self, f"{key}_success_count") + int((~failed).sum().item()))
        self._update_rates()

    def _run_batch(self, B):
        if self._engine == "torch":
            return self._run_batch_torch(B)
        error_x, error_z = self._generate_errors(B)
        synd_z = _mod2_mul(self.hx, error_z)
        synd_x = _mod2_mul(self.hz, error_x)
        if self.channel_update is None:
            rz = self._decode(self.bpd_z, synd_z)
            rx = self._decode(self.bpd_x, synd_x)
        elif self.channel_update == "x->z":
            rx = self._decode(self.bpd_x, synd_x)
            p1, p0 = self._updated_channel(self.channel_probs_x, self.channel_probs_z)
            self.bpd_z.update_channel_probs(p0)
            rz = self._decode(self.bpd_z, synd_z, select=rx["osdw"], alt=p1)
        elif self.channel_update == "z->x":
            rz = self._decode(self.bpd_z, synd_z)
            p1, p0 = self._updated_channel(self.channel_probs_z, self.channel_probs_x)
            self.bpd_x.update_channel_probs(p0)
            rx = self._decode(self.bpd_x, synd_x, select=rz["osdw"], alt=p1)
        else:
            raise ValueError(f"channel_update='{self.channel_update}' is invalid")
        self._encoded_error_rates(error_x, error_z, rx, rz)

    def _logical_fail(self, error_x, error_z, dx, dz):
        """css_decode_sim.py:257-272: X-logical checked first, Z-logical only otherwise."""
        residual_x = (error_x ^ dx).astype(np.uint8)
        residual_z = (error_z ^ dz).astype(np.uint8)
        fail_x = _mod2_mul(self._lz, residual_x).any(axis=1)
        fail_z = _mod2_mul(self._lx, residual_z).any(axis=1)
        weight = np.where(fail_x, residual_x.sum(axis=1), residual_z.sum(axis=1))
        return fail_x, fail_z, weight

    def _encoded_error_rates(self, error_x, error_z, rx, rz):  # css_decode_sim.py:250-365
        B = len(error_x)
        self.run_count += B
        self.bp_converge_count_z += int(rz["conv"].sum())
        self.bp_converge_count_x += int(rx["conv"].sum())
        both = rz["conv"] & rx["conv"]
        fx_bp, fz_bp, w_bp = self._logical_fail(error_x, error_z, rx["bp"], rz["bp"])
        self.bp_success_count += int((both & ~(fx_bp | fz_bp)).sum())
        # where both decoders converged, osdw = osd0 = bp in both sectors (decode contract, README.md:197-202 /
        # SURVEY Appendix A.2), so the logical checks of those shots are the BP ones; only the rest is recomputed
        rest = np.flatnonzero(~both)
        for key in ("osdw", "osd0"):
            fx, fz, weight = fx_bp.copy(), fz_bp.copy(), w_bp.copy()
            if rest.size:
                fx[rest], fz[rest], weight[rest] = self._logical_fail(error_x[rest], error_z[rest], rx[key][rest], rz[key][rest])
            failed = fx | fz
            if failed.any():
                wmin = int(weight[failed].min())
                if wmin < self.min_logical_weight:
                    self.min_logical_weight = wmin
            setattr(self, f"{key}_success_count", getattr(self, f"{key}_success_count") + int((~failed).sum()))
        self._update_rates()

    def _update_rates(self):
        n = self.run_count
        for key in ("osdw", "osd0", "bp"):
            ler = 1 - getattr(self, f"{key}_success_count") / n
            eb = float(np.sqrt((1 - ler) * ler / n))
            setattr(self, f"{key}_logical_error_rate", ler)
            setattr(self, f"{key}_logical_error_rate_eb", eb)
            setattr(self, f"{key}_word_error_rate", 1.0 - (1 - ler) ** (1 / self.K))
            setattr(self, f"{key}_word_error_rate_eb", eb * ((1 - eb) ** (1 / self.K - 1)) / self.K)

    # ------------------------------------------------------------------ main loop
    def run_decode_sim(self):  # css_decode_sim.py:500-555
        self.start_date = datetime.datetime.fromtimestamp(time.time()).strftime("%A, %B %d, %Y %H:%M:%S")
        start = time.time()
        while self.run_count < self.target_runs:
            B = min(self._batch_size, self.target_runs - self.run_count)
            self._run_batch(B)
            self.runtime = self.runtime + (time.time() - start)
            start = time.time()
            self.runtime_readable = time.strftime("%H:%M:%S", time.gmtime(self.runtime))
            if self.output_file is not None:
                with open(self.output_file, "w+") as f:
                    print(self.output_dict(), file=f)
            if (self.osdw_logical_error_rate_eb > 0 and
                    self.osdw_logical_error_rate_eb / self.osdw_logical_error_rate < self.error_bar_precision_cutoff):
                print("\\nTarget error bar precision reached. Stopping simulation...")
                break
        return json.dumps(self.output_dict(), sort_keys=True, indent=4)

    def output_dict(self):  # css_decode_sim.py:557-567 (returns a JSON string, as the reference does)
        out = {}
        for key, value in self.__dict__.items():
            if key in self.output_keys and not key.startswith("_"):
                if isinstance(value, (np.integer,)):
                    value = int(value)
                elif isinstance(value, (np.floating,)):
                    value = float(value)
                elif isinstance(value, np.ndarray) or sp.issparse(value) or hasattr(value, "decode_batch"):
                    continue
                out[key] = value
        return json.dumps(out, sort_keys=True, indent=4)
