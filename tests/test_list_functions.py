"""ArrayFunctions over List<primitive> columns (src/functions/array.rs) — SURVEY.md §8f item 4.

The oracle is pinned by the reference's own tests (array.rs:421-640: the 16-value / 6-row fixture, array_contains over
i32 / i64 / f64, array_position, array_remove, array_sort); the device kernels are then held to the oracle on
randomized lists: empty and NULL rows, row offsets, short rows (one row per lane) and long rows (one row per wave)."""
import numpy as np
import pytest

from rust_dataframe_amd import _abi as A

FIX_VALUES = [0, 0, 0, 1, 2, 1, 3, 4, 5, 1, 3, 2, 3, 2, 8, 3]
FIX_OFFSETS = [0, 3, 6, 8, 12, 14, 16]


def fixture(dtype):
    rows = [FIX_VALUES[FIX_OFFSETS[i]:FIX_OFFSETS[i + 1]] for i in range(6)]
    return A.HostList.from_lists(rows, dtype)


def check_reference_answers(api):
    for dt, needle in ((A.I32, 2), (A.I64, 2), (A.F64, 2.0)):                       # test_array_contains_i32s / i64s / f64s
        assert api.list_contains(fixture(dt), needle).to_pylist() == [False, True, False, True, True, False]
    assert api.list_position(fixture(A.I64), 2).to_pylist() == [0, 2, 0, 4, 2, 0]     # test_array_position
    offs, vals = api.list_remove(fixture(A.I64), 2)                                   # test_array_remove
    assert offs.to_numpy().tolist() == [0, 3, 5, 7, 10, 11, 13] and vals.length == 13
    srt = api.list_sort(fixture(A.I64))                                               # test_array_sort
    assert srt.to_numpy().tolist() == [0, 0, 0, 1, 1, 2, 3, 4, 1, 2, 3, 5, 2, 3, 3, 8]
    assert api.list_extreme(fixture(A.I32), True).to_pylist() == [0, 2, 4, 5, 3, 8]
    assert api.list_extreme(fixture(A.I32), False).to_pylist() == [0, 1, 3, 1, 2, 3]


def test_oracle_reference_known_answers(ora):
    check_reference_answers(ora)


@pytest.mark.gpu
def test_device_reference_known_answers(gpu):
    check_reference_answers(gpu)


def random_lists(rng, dtype, nrows, max_len, null_frac):
    npdt = A.NP_OF[dtype]
    rows = []
    for _ in range(nrows):
        if rng.uniform() < null_frac:
            rows.append(None)
            continue
        n = int(rng.integers(0, max_len + 1))
        if dtype in (A.F32, A.F64):
            v = np.round(rng.uniform(-4, 4, n), 0).astype(npdt)
            if n and rng.uniform() < 0.2:
                v[rng.integers(0, n)] = np.nan
            if n and rng.uniform() < 0.1:
                v[:] = np.nan
        else:
            info = np.iinfo(npdt)
            v = rng.integers(max(info.min, -5), min(info.max, 5) + 1, n).astype(npdt)
            if n and rng.uniform() < 0.1:
                v[rng.integers(0, n)] = info.max if rng.uniform() < 0.5 else info.min
        rows.append(v.tolist())
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [A.I8, A.I32, A.I64, A.U16, A.U64, A.F32, A.F64])
def test_list_functions_parity(gpu, ora, dtype):
    rng = np.random.default_rng(5000 + dtype)
    needle = 2.0 if dtype in (A.F32, A.F64) else 2
    for nrows, max_len, nf, off in [(1, 0, 0.0, 0), (200, 6, 0.1, 0), (3000, 12, 0.05, 3), (70, 400, 0.1, 1), (5, 5000, 0.0, 0)]:
        lst = A.HostList.from_lists(random_lists(rng, dtype, nrows, max_len, nf), dtype, row_offset=off)
        what = f"dtype={dtype} rows={nrows} max_len={max_len}"
        g, o = gpu.list_contains(lst, needle), ora.list_contains(lst, needle)
        assert g.to_pylist() == o.to_pylist() and g.null_count == o.null_count, "contains " + what
        assert gpu.list_position(lst, needle).to_pylist() == ora.list_position(lst, needle).to_pylist(), "position " + what
        for want_max in (True, False):
            g, o = gpu.list_extreme(lst, want_max), ora.list_extreme(lst, want_max)
            assert np.array_equal(g.valid_mask(), o.valid_mask()) and g.null_count == o.null_count, "extreme validity " + what
            gv, ov = g.to_numpy()[o.valid_mask()], o.to_numpy()[o.valid_mask()]
            assert np.array_equal(gv, ov, equal_nan=True), f"{'max' if want_max else 'min'} {what}"
        (go, gv), (oo, ov) = gpu.list_remove(lst, needle), ora.list_remove(lst, needle)
        assert np.array_equal(go.to_numpy(), oo.to_numpy()) and gv.length == ov.length, "remove offsets " + what
        assert np.array_equal(gv.to_numpy(), ov.to_numpy(), equal_nan=True), "remove values " + what
        gs, os_ = gpu.list_sort(lst), ora.list_sort(lst)
        assert gs.length == os_.length
        a, b = gs.to_numpy(), os_.to_numpy()
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "sort (bit-exact incl. NaN payloads and signed zeros) " + what


@pytest.mark.gpu
def test_list_function_errors(gpu, ora):
    lst = fixture(A.I32)
    for api in (gpu, ora):
        bad = A.HostList(lst.offsets.astype(np.int32), A.HostArray.from_numpy(np.array([True, False])), None, 0, 1)
        with pytest.raises(A.RdfError) as ei:
            api.list_contains(bad, 1)
        assert ei.value.status == A.RDF_INVALID_ARGUMENT
