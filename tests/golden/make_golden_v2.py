#!/usr/bin/env python3
"""Generates tests/golden/vectors_v2.npz: seeded input/output vectors for the functions added after vectors_v1 —
ScalarFunctions::hour (src/functions/scalar.rs:267-273) and the ArrayFunctions over List<Int64> columns
(src/functions/array.rs:15-399).

As for v1 the reference cannot run here; the expected outputs come from the CPU oracle and are CROSS-CHECKED before they are
written: hour against numpy's datetime64 calendar and pyarrow.compute.hour, the list functions against a statement of the
same per-row results with plain Python lists.  Run from the repo root:  python tests/golden/make_golden_v2.py
"""
import os
import sys

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from rust_dataframe_amd import _abi as A  # noqa: E402

SEED = 20260927
UNITS = [(A.TIME_SECOND, "s", 1), (A.TIME_MILLISECOND, "ms", 10 ** 3), (A.TIME_MICROSECOND, "us", 10 ** 6), (A.TIME_NANOSECOND, "ns", 10 ** 9)]
NEEDLE = 2


def rows_of(offs, vals):
    o, v = offs.to_numpy().tolist(), vals.to_numpy().tolist()
    return [v[o[i]:o[i + 1]] for i in range(len(o) - 1)]


def py_rows(op, a_rows, b_rows, count):
    out = []
    for i, ra in enumerate(a_rows):
        if ra is None:
            out.append([])
            continue
        rb = (b_rows[i] or []) if b_rows is not None else []
        uniq = list(dict.fromkeys(ra))
        out.append({"remove": [x for x in ra if x != NEEDLE], "distinct": uniq, "except": [x for x in uniq if x not in rb],
                    "intersect": [x for x in uniq if x in rb], "union": list(dict.fromkeys(ra + rb)), "repeat": ra * count}[op])
    return out


def random_rows(rng, nrows, max_len):
    rows = []
    for _ in range(nrows):
        if rng.uniform() < 0.1:
            rows.append(None)
        else:
            rows.append(rng.integers(-4, 5, int(rng.integers(0, max_len + 1))).astype(np.int64).tolist())
    return rows


def store_list(out, name, rows):
    offs, vals, valid = [0], [], []
    for r in rows:
        valid.append(r is not None)
        vals.extend(r or [])
        offs.append(len(vals))
    out[name + "_offsets"] = np.array(offs, dtype=np.int32)
    out[name + "_values"] = np.array(vals, dtype=np.int64)
    out[name + "_valid"] = np.array(valid, dtype=bool)


def main():
    o = oracle.api()
    rng = np.random.default_rng(SEED)
    out = {}
    # ---- hour
    valid = rng.uniform(size=2048) >= 0.1
    out["hour_valid"] = valid
    for unit, code, per_sec in UNITS:
        secs = rng.integers(-2_208_988_800, 4_102_444_800, 2048)
        v = (secs * per_sec + rng.integers(0, per_sec, 2048)).astype(np.int64)
        v[:6] = [0, -1, 86400 * per_sec - 1, 86400 * per_sec, -86400 * per_sec, 3600 * per_sec]
        r = o.hour([A.HostArray.from_numpy(v, valid)], unit)[0]
        dt = v.astype(f"datetime64[{code}]")
        want = (dt.astype("datetime64[h]") - dt.astype("datetime64[D]")).astype(np.int64)
        assert np.array_equal(r.valid_mask(), valid) and np.array_equal(r.to_numpy()[valid], want[valid]), code
        ref = pc.hour(pa.array(v, type=pa.timestamp(code), mask=~valid))
        assert np.array_equal(r.to_numpy()[valid], ref.fill_null(0).to_numpy(zero_copy_only=False)[valid]), code
        out[f"hour_{code}_in"] = v
        out[f"hour_{code}_out"] = r.to_numpy().copy()
    t32 = rng.integers(0, 86400, 1000).astype(np.int32)
    r = o.hour([A.HostArray.from_numpy(t32)], A.TIME_SECOND)[0]
    assert np.array_equal(r.to_numpy(), pc.hour(pa.array(t32, type=pa.time32("s"))).to_numpy())
    out["hour_time32_in"], out["hour_time32_out"] = t32, r.to_numpy().copy()
    # ---- ArrayFunctions over List<Int64>
    a_rows, b_rows = random_rows(rng, 400, 9), random_rows(rng, 400, 6)
    store_list(out, "la", a_rows)
    store_list(out, "lb", b_rows)
    la, lb = A.HostList.from_lists(a_rows, A.I64), A.HostList.from_lists(b_rows, A.I64)
    c = o.list_contains(la, NEEDLE)
    assert c.to_pylist() == [None if r is None else NEEDLE in r for r in a_rows]
    out["contains_values"], out["contains_valid"] = c.to_numpy().copy(), c.valid_mask().copy()
    p = o.list_position(la, NEEDLE)
    assert p.to_pylist() == [0 if r is None or NEEDLE not in r else r.index(NEEDLE) + 1 for r in a_rows]
    out["position_values"] = p.to_numpy().copy()
    for want_max in (True, False):
        m = o.list_extreme(la, want_max)
        f = max if want_max else min
        assert m.to_pylist() == [None if not r else f(r) for r in a_rows]
        nm = "max" if want_max else "min"
        out[nm + "_values"], out[nm + "_valid"] = m.to_numpy().copy(), m.valid_mask().copy()
    srt = o.list_sort(la)
    assert srt.to_numpy().tolist() == [x for r in a_rows for x in sorted(r or [])]
    out["sort_values"] = srt.to_numpy().copy()
    for op in ("remove", "distinct", "except", "intersect", "union", "repeat"):
        two = op in ("except", "intersect", "union")
        offs, vals = o.list_remove(la, NEEDLE) if op == "remove" else o.list_set(op, la, lb if two else None, count=3)
        assert rows_of(offs, vals) == py_rows(op, a_rows, b_rows if two else None, 3), op
        out[op + "_offsets"], out[op + "_values"] = offs.to_numpy().copy(), vals.to_numpy()[:vals.length].copy()
    path = os.path.join(ROOT, "tests", "golden", "vectors_v2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
