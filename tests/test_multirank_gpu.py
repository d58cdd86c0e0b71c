"""Two ranks with the HIP library as the per-rank engine (both ranks share the one GPU of the test box; gloo carries the
collectives, staged through host tensors for the collective alone): every bench.py workload at N = 2 equals the N = 1 run
over the same global rows.  The 8-GPU RCCL run is the driver's; this holds the rank-sharded code path itself — row-range
shards, all_gather combine of partials, device-side bucketing + all_to_all + device merge of partial groups — to the
single-rank result with the product kernels, not the oracle, underneath."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(workload, gpus, rows, port, launcher="torchrun", flags=(), **extra_env):
    """launcher = "torchrun": the driver's N > 1 command line; "self": plain `python bench.py --gpus N`, which must start
    its own ranks.  Both ranks share the test box's one GPU, so N > 1 here means gloo + --share-gpu."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
            "--workload", workload] + (["--rows", str(rows)] if rows else []) + list(flags)
    if gpus > 1 and launcher == "torchrun":
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a port nobody listens on (the number passed in is only a label)
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base + ["--backend", "gloo", "--share-gpu"]
    elif gpus > 1:
        cmd = [sys.executable] + base + ["--backend", "gloo", "--share-gpu"]
    else:
        cmd = [sys.executable] + base
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"{cmd}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}"
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_headline_two_ranks_equal_one():
    two = _run("headline", 2, 20_000_000, 29611)
    one = _run("headline", 1, 40_000_000, 0)
    assert two["n_gpus"] == 2 and two["config"]["total_rows"] == one["config"]["total_rows"] == 40_000_000
    assert two["config"]["result_count"] == one["config"]["result_count"]
    assert abs(two["config"]["result_sum"] - one["config"]["result_sum"]) <= 1e-12 * one["config"]["result_sum"]


def test_c4_two_ranks_device_exchange():
    two = _run("c4", 2, 10_000_000, 29621)
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["self_check"] is True and cfg["parity_on_sample"] is True, cfg
    assert cfg["groups_total"] == 1_000_000          # every key owned by exactly one rank, none lost, none twice


def test_c4_four_ranks_device_exchange():
    """Four ranks (the uneven per-owner splits of the driver's N = 4 / 8 runs): partial groups packed on the device,
    exchanged, merged at their owners; and the headline's row ranges at four ranks."""
    four = _run("c4", 4, 5_000_000, 29651)
    cfg = four["config"]
    assert four["n_gpus"] == 4 and cfg["self_check"] is True and cfg["parity_on_sample"] is True, cfg
    assert cfg["groups_total"] == 1_000_000
    h4 = _run("headline", 4, 10_000_000, 29661)
    h1 = _run("headline", 1, 40_000_000, 0)
    assert h4["config"]["result_count"] == h1["config"]["result_count"]
    assert abs(h4["config"]["result_sum"] - h1["config"]["result_sum"]) <= 1e-12 * h1["config"]["result_sum"]


def test_c4_two_ranks_row_shuffle():
    """SURVEY.md 8e's fallback: the rows themselves are bucketed on the device, exchanged and aggregated at their owner."""
    two = _run("c4", 2, 10_000_000, 29641, RDF_C4_SHUFFLE_ROWS="1")
    cfg = two["config"]
    assert two["n_gpus"] == 2 and cfg["self_check"] is True and cfg["parity_on_sample"] is True, cfg
    assert cfg["groups_total"] == 1_000_000 and cfg["result"].get("exchange") == "rows", cfg


def test_eight_ranks_small_shards_through_bench_main():
    """The driver's N = 8 command line end to end through bench.py's own main() at 1e6 rows per rank (eight processes on the one
    GPU, gloo): every rank's share of the result is gathered and must add up to the combined count before the line is printed;
    the line carries every rank's step time, kernel time and roofline fraction.  C4 at eight ranks: every key owned once."""
    h8 = _run("headline", 8, 1_000_000, 0)
    h1 = _run("headline", 1, 8_000_000, 0)
    cfg = h8["config"]
    assert h8["n_gpus"] == 8 and cfg["total_rows"] == 8_000_000 and cfg["result_count"] == h1["config"]["result_count"]
    assert abs(cfg["result_sum"] - h1["config"]["result_sum"]) <= 1e-12 * h1["config"]["result_sum"]
    pr = cfg["per_rank"]
    assert len(pr["result_count"]) == 8 and sum(pr["result_count"]) == cfg["result_count"] and min(pr["result_count"]) > 0
    assert len(pr["roofline_frac"]) == 8 and all(f is not None and 0 < f < 1 for f in pr["roofline_frac"]), pr
    assert h8["roofline"]["frac_wall"] <= h8["roofline"]["frac"] * 1.05
    c8 = _run("c4", 8, 1_000_000, 0)
    cc = c8["config"]
    assert c8["n_gpus"] == 8 and cc["self_check"] is True and cc["parity_on_sample"] is True, cc
    assert cc["check_totals"]["sum_of_group_counts"] == 8_000_000


def test_q1_two_ranks_equal_one():
    two = _run("q1", 2, 10_000_000, 29631)
    one = _run("q1", 1, 20_000_000, 0)
    a, b = two["config"], one["config"]
    assert a["self_check"] is True and b["self_check"] is True
    assert a["result"]["count_star"] == b["result"]["count_star"]
    for name in ("sum_qty", "sum_price", "sum_disc_price", "sum_charge", "sum_disc"):
        for x, y in zip(a["result"][name], b["result"][name]):
            assert abs(x - y) <= 1e-9 * abs(y), name


# ------------------------------------------------------------------ the driver's plain command line, strong scaling, RCCL
def test_self_launch_two_ranks_headline_and_c4():
    """`python bench.py --gpus 2` with no launcher around it (what the driver runs): bench.py starts its own ranks under
    torch.distributed.run and rank 0's single JSON line comes back."""
    two = _run("headline", 2, 20_000_000, 0, launcher="self")
    one = _run("headline", 1, 40_000_000, 0)
    assert two["n_gpus"] == 2 and two["config"]["ranks"] == 2 and two["config"]["launcher"].startswith("self")
    assert two["config"]["result_count"] == one["config"]["result_count"]
    c4 = _run("c4", 2, 10_000_000, 0, launcher="self")
    cfg = c4["config"]
    assert cfg["ranks"] == 2 and cfg["self_check"] is True and cfg["parity_on_sample"] is True and cfg["groups_total"] == 1_000_000, cfg
    assert cfg["result"]["exchange_bytes_sent"] > 0 and cfg["result"]["exchange_ms"] > 0, cfg


def test_strong_scaling_total_rows():
    """--total-rows: one global row range cut over the ranks (BASELINE C4 is stated that way: 1e9 rows over 8 GPUs)."""
    total = 30_000_000 + 1024 * 3 + 17          # ragged: the last rank's shard ends inside a batch
    two = _run("c4", 2, 0, 0, launcher="self", flags=["--total-rows", str(total)])
    one = _run("c4", 1, 0, 0, flags=["--total-rows", str(total)])
    a, b = two["config"], one["config"]
    assert two["scaling"] == "strong" == one["scaling"] and a["total_rows"] == b["total_rows"] == total
    assert a["self_check"] is True and b["self_check"] is True and a["groups_total"] == b["groups_total"]
    h2 = _run("headline", 2, 0, 0, launcher="self", flags=["--total-rows", str(total)])
    h1 = _run("headline", 1, 0, 0, flags=["--total-rows", str(total)])
    assert h2["config"]["result_count"] == h1["config"]["result_count"] and h2["config"]["total_rows"] == total
    assert abs(h2["config"]["result_sum"] - h1["config"]["result_sum"]) <= 1e-12 * h1["config"]["result_sum"]


@pytest.mark.parametrize("workload,env", [("headline", {}), ("q1", {}), ("c3", {}), ("c4", {}), ("c4", {"RDF_C4_SHUFFLE_ROWS": "1"})])
def test_rccl_one_rank_communicator(workload, env):
    """The production backend: init_process_group("nccl", device_id=...) and every collective of the N > 1 path on DEVICE
    tensors — all_gather of the partials, all_to_all_single of the split sizes, all_to_all_single of the packed HBM buffers
    with uneven split lists — on a 1-rank RCCL communicator (the test box has one GPU; 2 ranks on one GPU is not a
    configuration RCCL accepts).  Results equal the run without any process group."""
    rows = 20_000_000 if workload == "c4" else 10_000_000      # 20 rows per key: every one of the 1e6 keys occurs
    nc = _run(workload, 1, rows, 0, flags=["--force-exchange", "--backend", "nccl"], **env)
    plain = _run(workload, 1, rows, 0)
    cfg, ref = nc["config"], plain["config"]
    assert cfg["ranks"] == 1 and cfg["backend"] == "nccl" and cfg["rccl_version"], cfg
    if workload == "headline":
        assert cfg["result_count"] == ref["result_count"] and cfg["result_sum"] == ref["result_sum"]
        return
    assert cfg["self_check"] is True and cfg["parity_on_sample"] is True, cfg
    if workload == "c4":
        assert cfg["groups_total"] == ref["groups_total"] == 1_000_000
        assert cfg["result"]["exchange"] == ("rows" if env else "partial groups")
        width = 16 * rows if env else 24 * 1_000_000
        assert cfg["result"]["exchange_bytes_sent"] == width == cfg["result"]["exchange_bytes_received"], cfg
        assert cfg["result"]["exchange_bytes_sent_remote"] == 0
    else:
        assert cfg["result"] == ref["result"]


def test_exchange_in_rounds():
    """all_to_all_single calls are kept below 256 MiB (a call past ~1 GiB delivers half its buffer on this RCCL: see
    tools/rccl_a2a_probe.py): with the cap shrunk to 1 MiB both exchanges of the group-by run in dozens of rounds — 2 ranks under
    gloo, 1 rank under RCCL — and still produce every group once."""
    for env in ({}, {"RDF_C4_SHUFFLE_ROWS": "1"}):
        two = _run("c4", 2, 6_000_000, 0, launcher="self", RDF_A2A_MAX_BYTES=str(1 << 20), **env)
        cfg = two["config"]
        assert cfg["self_check"] is True and cfg["parity_on_sample"] is True and 999_000 < cfg["groups_total"] <= 1_000_000, cfg
        one = _run("c4", 1, 12_000_000, 0, flags=["--force-exchange", "--backend", "nccl"], RDF_A2A_MAX_BYTES=str(1 << 20), **env)
        assert one["config"]["self_check"] is True and one["config"]["groups_total"] == cfg["groups_total"], one["config"]
