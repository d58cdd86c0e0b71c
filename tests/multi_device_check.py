"""The checker of rust_dataframe_amd/selftest.py's multi-device run: the union of the ranks' groups and the distributed aggregate
against the CPU oracle over the unsharded rows (test infrastructure: used by tests/ and by __graft_entry__.smoke())."""
import numpy as np

from rust_dataframe_amd import _abi as A


def check_against_oracle(out, ora, rtol=1e-6):
    """The union of the ranks' groups equals the oracle's GROUP BY over the unsharded rows (keys and counts exactly, sums within
    rtol: north_star's f64 tolerance); every rank got the same distributed aggregate, equal to the oracle's."""
    keys, vals = out["inputs"]
    ok, ov, oc = ora.groupby_agg([[A.HostArray.from_numpy(keys)]], [A.HostArray.from_numpy(vals)], "sum", len(np.unique(keys)) + 8)
    n = oc.length
    ek, es, ec = ok[0].to_numpy()[:n], ov.to_numpy()[:n], oc.to_numpy()[:n]
    o1, o2 = np.argsort(out["keys"]), np.argsort(ek)
    assert len(out["keys"]) == n and np.array_equal(out["keys"][o1], ek[o2]), "group keys differ"
    assert np.array_equal(out["counts"][o1], ec[o2]), "group counts differ"
    assert np.allclose(out["sums"][o1], es[o2], rtol=rtol, atol=0), "group sums differ"
    e = A.Expr()
    exp = ora.pipeline(e, [[A.HostArray.from_numpy(keys)], [A.HostArray.from_numpy(vals)]], [e.col(1)], e.op("gt", e.col(1), e.scalar(0.5)))[0]
    for s, c, mn, mx in out["pipeline"]:
        assert c == exp.count and mn == exp.min and mx == exp.max and abs(s - exp.sum) <= rtol * abs(exp.sum), "distributed aggregate differs"
    return True
