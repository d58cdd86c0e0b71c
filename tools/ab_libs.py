#!/usr/bin/env python3
"""Same-box A/B of two builds of librdf_mi355x.so: the boxes gpurun hands out differ by +-1.5 % on the headline and by more on
the instruction-bound kernels, so two builds measured in two calls cannot be told apart below ~5 %.  This runs one of the bench
tools alternately against build A and build B in ONE call (A B A B ...), and reports the median kernel time of every entry and
the B / A ratio.

    cp rust_dataframe_amd/librdf_mi355x.so /tmp/a.so      # build A, then change the source and rebuild -> build B in place
    python tools/ab_libs.py --a /tmp/a.so --b rust_dataframe_amd/librdf_mi355x.so --rounds 3 -- \\
        python tools/bench_shapes.py --dtypes i16,i32,f64 --programs two_level_2col

Both builds must travel to the GPU box: keep A under the repo (e.g. gpurun_out/ is NOT shipped; use a path such as
rust_dataframe_amd/librdf_mi355x_a.so, which *.so keeps out of git).  Entries are keyed by every string / integer field of the
bench line except the measurements."""
import argparse
import json
import os
import statistics
import subprocess
import sys

MEASURED = {"kernel_ms", "wall_ms", "GBps", "frac_of_8TBps"}


def run(cmd, lib_path, dry):
    env = dict(os.environ, RDF_LIB_PATH=os.path.abspath(lib_path))
    if dry:
        print(f"RDF_LIB_PATH={env['RDF_LIB_PATH']} {' '.join(cmd)}")
        return {}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, check=True).stdout
    res = {}
    for line in out.splitlines():
        if not line.startswith("{") or "kernel_ms" not in line:
            continue
        d = json.loads(line)
        key = " ".join(str(v) for k, v in d.items() if k not in MEASURED and not isinstance(v, float))
        res[key] = float(d["kernel_ms"])
    return res


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--a", required=True, help="build A (baseline)")
    ap.add_argument("--b", required=True, help="build B (candidate)")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("cmd", nargs=argparse.REMAINDER, help="-- bench command printing JSON lines with kernel_ms")
    args = ap.parse_args()
    cmd = [c for c in args.cmd if c != "--"]
    assert cmd, "give the bench command after --"
    times = {"a": {}, "b": {}}
    for _ in range(args.rounds):
        for tag, path in (("a", args.a), ("b", args.b)):
            for key, ms in run(cmd, path, args.dry_run).items():
                times[tag].setdefault(key, []).append(ms)
    for key in times["a"]:
        if key not in times["b"]:
            continue
        ma, mb = statistics.median(times["a"][key]), statistics.median(times["b"][key])
        print(json.dumps({"entry": key, "a_ms": round(ma, 4), "b_ms": round(mb, 4), "b_over_a": round(mb / ma, 4),
                          "a_spread": round((max(times["a"][key]) - min(times["a"][key])) / ma, 4),
                          "b_spread": round((max(times["b"][key]) - min(times["b"][key])) / mb, 4), "rounds": args.rounds}))


if __name__ == "__main__":
    sys.exit(main())
