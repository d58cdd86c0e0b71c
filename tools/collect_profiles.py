#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/profile_<round>/ (written by tools/profile_round.sh on the GPU box) into profiles/
and derives the PMC summary: HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE (gfx950 reports half the bytes
of wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both in KB per dispatch."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_mean(path, name, kernel_substr):
    vals, meta = [], {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and kernel_substr in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            meta = {"vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"], "kernel": r["Kernel_Name"]}
    return {"dispatches": len(vals), "mean_KB": sum(vals) / max(len(vals), 1), "min_KB": min(vals, default=0), "max_KB": max(vals, default=0), **meta}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"profile_{rnd}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    bench = json.loads(open(os.path.join(src, "bench_1e9.json")).read().strip().splitlines()[-1])
    kernel = bench["roofline"]["kernel"]
    shutil.copy(os.path.join(src, "bench_1e9.json"), os.path.join(dst, f"{rnd}_bench_1e9.json"))
    shutil.copy(os.path.join(src, "bench_1e9_under_rocprof.json"), os.path.join(dst, f"{rnd}_bench_1e9_under_rocprof.json"))
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{rnd}_bench_1e9_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "kernels_1e9_microbench.jsonl"), os.path.join(dst, f"{rnd}_kernels_1e9_microbench.jsonl"))
    if os.path.exists(os.path.join(src, "workloads.jsonl")):
        shutil.copy(os.path.join(src, "workloads.jsonl"), os.path.join(dst, f"{rnd}_workloads_c3_c4_q1.jsonl"))
    raw = {}
    for cname, d in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        files = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)
        raw[cname] = counter_mean(files[0], cname, "spec_kernel") if files else {}
    rows = bench["config"]["rows_per_gpu"]
    hbm = (2.0 * raw["FETCH_SIZE"].get("mean_KB", 0) + raw["WRITE_SIZE"].get("mean_KB", 0)) * 1024.0
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    summary = {"round": int(rnd[1:]), "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 5 --warmup 1 --cpu-sample 0 (two separate passes, tools/profile_round.sh)",
               "kernel": kernel, "rows": rows, "validity": False, "raw": raw,
               "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": hbm / alg if alg else None,
               "note": "FETCH_SIZE doubled: gfx950 reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM / rocprofv3 section)"}
    json.dump(summary, open(os.path.join(dst, f"{rnd}_bench_1e9_pmc_summary.json"), "w"), indent=1)
    json.dump({"rows": rows, "validity": False, "hbm_bytes_per_launch": hbm, "source": f"profiles/{rnd}_bench_1e9_pmc_summary.json"},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps({"value": bench["value"], "frac": bench["roofline"]["frac"], "avg_kernel_ms": bench["roofline"]["avg_kernel_ms"],
                      "cpu": bench["cpu_baseline"]["value"], "hbm_bytes": hbm, "ratio": hbm / alg}))


if __name__ == "__main__":
    main()
