#!/usr/bin/env python3
"""Writes tests/golden/mixed_batches.arrow: an Arrow IPC FILE (pyarrow, uncompressed) with 3 record batches
(1024, 1024, 576 rows) whose values follow closed formulas of the global row number i, so the C++ test of
rdf::DataFrame::from_arrow recomputes every expectation without pyarrow:

  i8   = (i % 200) - 100                      f32 = i / 8
  i32  = 7 i - 1000                            f64 = 0.5 i - 100, NULL when i % 10 == 3
  i64  = 10^9 i                                flag = (i % 3 == 0), NULL when i % 7 == 0
  u16  = (13 i) % 65536                        city = "city<i>"
"""
import os

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
LENS = [1024, 1024, 576]


def batch(first, n):
    i = np.arange(first, first + n, dtype=np.int64)
    return pa.record_batch([
        pa.array(((i % 200) - 100).astype(np.int8)),
        pa.array((7 * i - 1000).astype(np.int32)),
        pa.array(i * 10 ** 9),
        pa.array(((13 * i) % 65536).astype(np.uint16)),
        pa.array((i / 8).astype(np.float32)),
        pa.array(0.5 * i - 100.0, mask=(i % 10 == 3)),
        pa.array(i % 3 == 0, mask=(i % 7 == 0)),
        pa.array([f"city{k}" for k in i]),
    ], names=["i8", "i32", "i64", "u16", "f32", "f64", "flag", "city"])


def main():
    path = os.path.join(HERE, "mixed_batches.arrow")
    first, batches = 0, []
    for n in LENS:
        batches.append(batch(first, n))
        first += n
    with pa.OSFile(path, "wb") as f, pa.ipc.new_file(f, batches[0].schema) as w:
        for b in batches:
            w.write_batch(b)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------------------------
# Dictionary-encoded columns: tests/golden/dict_batches.arrows (IPC STREAM, delta dictionaries) and dict_batches.arrow (IPC FILE).
# Row i of batch b (b = 0, 1, 2 with 700, 300, 524 rows; K(b) = 5, 8, 11 categories known to batch b's dictionary):
#   i64   = i                                   (plain column)
#   code  = "cat<(7 i) % K(b)>", NULL when i % 11 == 5      dictionary<values: utf8, indices: int8>
#   level = 0.25 * ((3 i) % 6),  NULL when i % 13 == 0      dictionary<values: float64, indices: int16>
#   small = ((5 i) % 4) * 1000 - 1500                        dictionary<values: int32, indices: uint8>
# The FILE holds the same rows with one unified 11-entry dictionary (the file format has no replacement dictionaries).
DICT_LENS = [700, 300, 524]
DICT_K = [5, 8, 11]


def dict_batch(first, n, k, k_dict):
    i = np.arange(first, first + n, dtype=np.int64)
    code = pa.DictionaryArray.from_arrays(pa.array(((7 * i) % k).astype(np.int8), mask=(i % 11 == 5)), pa.array([f"cat{j}" for j in range(k_dict)]))
    level = pa.DictionaryArray.from_arrays(pa.array(((3 * i) % 6).astype(np.int16), mask=(i % 13 == 0)), pa.array([0.25 * j for j in range(6)]))
    small = pa.DictionaryArray.from_arrays(pa.array(((5 * i) % 4).astype(np.uint8)), pa.array([j * 1000 - 1500 for j in range(4)], pa.int32()))
    return pa.record_batch([pa.array(i), code, level, small], names=["i64", "code", "level", "small"])


def dict_main():
    first, stream_batches, file_batches = 0, [], []
    for n, k in zip(DICT_LENS, DICT_K):
        stream_batches.append(dict_batch(first, n, k, k))
        file_batches.append(dict_batch(first, n, k, DICT_K[-1]))
        first += n
    spath = os.path.join(HERE, "dict_batches.arrows")
    opts = pa.ipc.IpcWriteOptions(emit_dictionary_deltas=True)
    with pa.OSFile(spath, "wb") as f, pa.ipc.new_stream(f, stream_batches[0].schema, options=opts) as w:
        for b in stream_batches:
            w.write_batch(b)
    fpath = os.path.join(HERE, "dict_batches.arrow")
    with pa.OSFile(fpath, "wb") as f, pa.ipc.new_file(f, file_batches[0].schema) as w:
        for b in file_batches:
            w.write_batch(b)
    # read back with pyarrow: both hold the same logical rows
    st = pa.ipc.open_stream(spath).read_all()
    ft = pa.ipc.open_file(fpath).read_all()
    assert st.num_rows == ft.num_rows == sum(DICT_LENS)
    for name in ("i64", "code", "level", "small"):
        assert st[name].cast(st[name].type.value_type if name != "i64" else pa.int64()).to_pylist() == \
               ft[name].cast(ft[name].type.value_type if name != "i64" else pa.int64()).to_pylist(), name
    print(spath, os.path.getsize(spath), "bytes;", fpath, os.path.getsize(fpath), "bytes")


if __name__ == "__main__":
    dict_main()
