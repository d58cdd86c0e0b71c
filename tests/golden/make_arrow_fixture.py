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
