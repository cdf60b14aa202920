"""The batched harness reproduces the reference's per-shot loop exactly (same RNG stream, same counters)
when both run on the same decoder -- here the CPU oracle, so no GPU is needed."""
import json

import numpy as np
import pytest

from bp_osd_amd.sim import css_decode_sim
from tests.sim_util import OracleAdapter, serial_reference_loop

KEYS = ("osdw_success_count", "osd0_success_count", "bp_success_count", "bp_converge_count_x",
        "bp_converge_count_z", "min_logical_weight")


@pytest.mark.parametrize("channel_update", [None, "x->z", "z->x"])
@pytest.mark.parametrize("bias", [[1, 1, 1], [0, 0, 1]])
def test_batched_harness_equals_serial_reference_loop(surface13, channel_update, bias):
    opts = dict(error_rate=0.12, xyz_error_bias=bias, target_runs=150, seed=42, channel_update=channel_update,
                bp_method="ms", ms_scaling_factor=0.625, max_iter=2, osd_method="osd_cs", osd_order=3)
    sim = css_decode_sim(hx=surface13.hx, hz=surface13.hz, batch_size=64, decoder_factory=OracleAdapter,
                         tqdm_disable=1, **opts)
    ref = serial_reference_loop(surface13.hx, surface13.hz, sim.lx, sim.lz, **opts)
    for k in KEYS:
        assert getattr(sim, k) == ref[k], (k, getattr(sim, k), ref[k])
    assert sim.run_count == 150 and sim.N == 13 and sim.K == 1
    # rates follow the reference's formulas (css_decode_sim.py:274-290)
    ler = 1 - sim.osdw_success_count / 150
    assert sim.osdw_logical_error_rate == pytest.approx(ler)
    assert sim.osdw_logical_error_rate_eb == pytest.approx(np.sqrt((1 - ler) * ler / 150))
    assert sim.osdw_word_error_rate == pytest.approx(1 - (1 - ler) ** (1 / sim.K))


def test_output_dict_keys_and_resume(surface13, tmp_path):
    out = tmp_path / "sim.json"
    sim = css_decode_sim(hx=surface13.hx, hz=surface13.hz, decoder_factory=OracleAdapter, error_rate=0.05,
                         target_runs=40, seed=7, channel_update=None, output_file=str(out), osd_order=2)
    d = json.loads(sim.output_dict())
    for key in ("K", "N", "run_count", "osdw_success_count", "osdw_logical_error_rate", "osdw_word_error_rate",
                "osd0_logical_error_rate", "bp_logical_error_rate", "bp_converge_count_x", "min_logical_weight",
                "error_rate", "osd_method", "seed", "runtime"):
        assert key in d, key
    assert d["run_count"] == 40 and json.loads(out.read_text())["run_count"] == 40
    # resume: pass the dict back, target_runs raised (css_decode_sim.py:87-91,117-119,135-136)
    d2 = dict(d)
    d2["target_runs"] = 60
    sim2 = css_decode_sim(hx=surface13.hx, hz=surface13.hz, decoder_factory=OracleAdapter, **d2)
    assert sim2.run_count == 60 and sim2.osdw_success_count >= sim.osdw_success_count
    assert sim2.seed != 7  # a fresh seed is drawn when resuming


def test_invalid_code_rejected(surface13):
    with pytest.raises(Exception, match="invalid CSS code"):
        css_decode_sim(hx=surface13.hx, hz=surface13.hx, decoder_factory=OracleAdapter, error_rate=0.05)


def test_engine_arguments_are_validated():
    """engine / rng switches of the batched harness (host-side checks only; the torch engine itself needs a GPU)."""
    from bp_osd_amd.codes import surface13
    from bp_osd_amd.sim import css_decode_sim
    from tests.sim_util import OracleAdapter

    c = surface13()
    base = dict(hx=c.hx, hz=c.hz, error_rate=0.05, target_runs=4, seed=3, run_sim=0, tqdm_disable=1)
    with pytest.raises(ValueError):
        css_decode_sim(engine="cuda", decoder_factory=OracleAdapter, **base)
    with pytest.raises(ValueError):
        css_decode_sim(engine="numpy", rng="torch", decoder_factory=OracleAdapter, **base)
    with pytest.raises(ValueError):
        css_decode_sim(engine="torch", decoder_factory=OracleAdapter, **base)
    sim = css_decode_sim(decoder_factory=OracleAdapter, **base)
    assert "engine" not in sim.output_dict() and "_rng" not in sim.output_dict()
