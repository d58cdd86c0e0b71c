#!/usr/bin/env python3
"""Ingestion benchmark driver (GPU box): writes an Arrow IPC file and a CSV with pyarrow, builds tools/bench_ingest.cpp against
include/rdf_frame.hpp + librdf_mi355x.so, runs it.  Usage: python tools/bench_ingest.py [--rows N] [--batch-rows B] [--csv-rows M]"""
import argparse
import os
import subprocess
import tempfile

import numpy as np
import pyarrow as pa
import pyarrow.csv as pacsv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=64_000_000)
ap.add_argument("--batch-rows", type=int, default=1_000_000)
ap.add_argument("--csv-rows", type=int, default=2_000_000)
args = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="rdf_ingest_")
arrow_path, csv_path = os.path.join(tmp, "t.arrow"), os.path.join(tmp, "t.csv")
rng = np.random.default_rng(1)
schema = pa.schema([("a", pa.float64()), ("b", pa.float64()), ("c", pa.float64()), ("k", pa.int64()), ("n", pa.float64())])
with pa.OSFile(arrow_path, "wb") as sink, pa.ipc.new_file(sink, schema) as w:
    for first in range(0, args.rows, args.batch_rows):
        m = min(args.batch_rows, args.rows - first)
        cols = [pa.array(rng.uniform(size=m)) for _ in range(3)] + [pa.array(rng.integers(0, 1 << 40, m))]
        nn = rng.uniform(size=m)
        cols.append(pa.array(nn, mask=nn < 0.1))       # a nullable column: validity bitmaps travel too
        w.write_batch(pa.record_batch(cols, schema=schema))
m = args.csv_rows
pacsv.write_csv(pa.table({"x": rng.uniform(size=m), "i": rng.integers(0, 1_000_000, m), "y": rng.normal(size=m)}), csv_path)
exe = os.path.join(tmp, "bench_ingest")
pkg = os.path.join(ROOT, "rust_dataframe_amd")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "bench_ingest.cpp"), "-o", exe,
                       "-L", pkg, "-lrdf_mi355x", f"-Wl,-rpath,{pkg}"])
subprocess.check_call([exe, arrow_path, csv_path])
for p in (arrow_path, csv_path, exe):
    os.remove(p)
os.rmdir(tmp)
