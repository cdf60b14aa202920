"""The oracle reproduces the committed golden vectors (regression pin of the oracle itself)."""
import numpy as np
import pytest

from oracle import OracleDecoder
from tests.golden_util import golden_files, load


@pytest.mark.parametrize("name", golden_files())
def test_oracle_matches_golden(name):
    g = load(name)
    dec = OracleDecoder(g["H"], **g["cfg"])
    r = dec.decode_batch(g["syn"])
    assert (r["osdw"] == g["osdw"]).all()
    assert (r["osd0"] == g["osd0"]).all()
    assert (r["bp"] == g["bp"]).all()
    assert (r["converged"] == g["converged"]).all()
    assert (r["iters"] == g["iters"]).all()
    k = len(g["llr"])
    assert (r["llr"][:k].view(np.uint64) == g["llr"].view(np.uint64)).all()
    Hd = g["H"].toarray()
    assert ((g["osdw"] @ Hd.T) % 2 == g["syn"]).all()
