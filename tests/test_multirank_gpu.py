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


def _run(workload, gpus, rows, port, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--rows", str(rows), "--cpu-sample", "0",
            "--workload", workload]
    if gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + base + ["--backend", "gloo", "--share-gpu"]
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


def test_q1_two_ranks_equal_one():
    two = _run("q1", 2, 10_000_000, 29631)
    one = _run("q1", 1, 20_000_000, 0)
    a, b = two["config"], one["config"]
    assert a["self_check"] is True and b["self_check"] is True
    assert a["result"]["count_star"] == b["result"]["count_star"]
    for name in ("sum_qty", "sum_price", "sum_disc_price", "sum_charge", "sum_disc"):
        for x, y in zip(a["result"][name], b["result"][name]):
            assert abs(x - y) <= 1e-9 * abs(y), name
