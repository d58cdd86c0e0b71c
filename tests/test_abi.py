"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/rdf_mi355x.h declares,
its structs match the ctypes mirror byte for byte, argument validation happens before any device work, and
— with no device — every compute entry point fails loudly with RDF_DEVICE_ERROR (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from rust_dataframe_amd import _abi as A
from rust_dataframe_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rdf_mi355x.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rdf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 25
    so = lib.load()
    for n in names:
        assert hasattr(so, n), f"librdf_mi355x.so does not export {n}"
    assert sorted(lib.EXPORTS) == names
    assert "gfx950" in lib.version()


def test_integration_notes_bind_every_declared_symbol():
    """INTEGRATION.md (the reference-side binding) and integration/rdf_shim.rs name every entry point of the header."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(ROOT, "integration", "rdf_shim.rs")).read()
    for n in declared_functions():
        assert n in doc, f"INTEGRATION.md does not mention {n}"
        assert n in shim, f"integration/rdf_shim.rs does not declare {n}"


def test_oracle_exports_the_mirror_symbols(ora):
    for n in ["binary", "unary", "cast", "sum", "min", "max", "count", "avg", "predicate", "filter_count", "filter",
              "filter_columns", "take", "pipeline", "fill_uniform_f64", "fill_uniform_i64", "fill_validity"]:
        assert hasattr(ora.lib, "ora_" + n)


def test_ctypes_structs_match_the_header():
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "rdf_mi355x.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  S(rdf_array); O(rdf_array, validity); O(rdf_array, offset); O(rdf_array, null_count); O(rdf_array, dtype); O(rdf_array, mem);
  S(rdf_out); O(rdf_out, capacity); O(rdf_out, length); O(rdf_out, null_count); O(rdf_out, dtype); O(rdf_out, mem);
  S(rdf_expr_node); O(rdf_expr_node, column); O(rdf_expr_node, f64); O(rdf_expr_node, i64);
  S(rdf_program); O(rdf_program, nnodes); O(rdf_program, filter_root); O(rdf_program, value_roots); O(rdf_program, sink);
  S(rdf_agg_result); O(rdf_agg_result, sum_i64); O(rdf_agg_result, count); O(rdf_agg_result, is_some); O(rdf_agg_result, dtype);
  S(rdf_exchange_stats); O(rdf_exchange_stats, rounds); O(rdf_exchange_stats, local_groups); O(rdf_exchange_stats, bytes_sent); O(rdf_exchange_stats, exchange_ms);
  printf("enum %d %d %d %d %d\n", RDF_OP_TANH, RDF_OP_CAST, RDF_OP_OR, RDF_BOOL, RDF_DEVICE_ERROR);
  return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), src, "-o", exe])  # header is plain C
        out = dict(line.rsplit(" ", 1) for line in subprocess.check_output([exe], text=True).splitlines() if not line.startswith("enum"))
        enums = subprocess.check_output([exe], text=True).splitlines()[-1]
    for cls in (A.rdf_array, A.rdf_out, A.rdf_expr_node, A.rdf_program, A.rdf_agg_result, A.rdf_exchange_stats):
        assert int(out[cls.__name__]) == C.sizeof(cls), cls.__name__
        for key, val in out.items():
            if key.startswith(cls.__name__ + "."):
                assert getattr(cls, key.split(".")[1]).offset == int(val), key
    assert enums == f"enum {A.OP_TANH} {A.OP_CAST} {A.OP_OR} {A.BOOL} {A.RDF_DEVICE_ERROR}"


def test_argument_validation_needs_no_device():
    """Errors that are values in the reference are reported before the device is touched."""
    api = lib.api()
    a = [A.HostArray.from_numpy(np.array([1.0, 2.0, 3.0]))]
    b = [A.HostArray.from_numpy(np.array([1.0, 2.0]))]
    with pytest.raises(A.RdfError) as ei:
        api.binary("add", a, b)
    assert ei.value.status == A.RDF_COMPUTE_ERROR and "different length" in ei.value.message
    with pytest.raises(A.RdfError) as ei:
        api.unary("sin", [A.HostArray.from_numpy(np.array([1, 2], dtype=np.int32))])
    assert ei.value.status == A.RDF_INVALID_ARGUMENT
    with pytest.raises(A.RdfError) as ei:
        api.binary("add", a, [A.HostArray.from_numpy(np.array([1, 2, 3], dtype=np.int64))])
    assert ei.value.status == A.RDF_INVALID_ARGUMENT
    # metadata-only count needs no device either (O(#chunks), like aggregate.rs:70-80)
    assert api.count([A.HostArray.from_numpy(np.arange(5.0), valid=[1, 0, 1, 1, 0])]) == 3
    # empty inputs produce empty outputs
    assert api.unary("sin", [A.HostArray.from_numpy(np.zeros(0))])[0].length == 0


def _build_c_example(d, name="example"):
    exe = os.path.join(d, name)
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-pthread", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "integration", name + ".c"),
                           "-L", libdir, "-lrdf_mi355x", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


@pytest.mark.skipif(lib.device_count() > 0, reason="a GPU is visible")
def test_c_example_links_and_fails_loudly_without_a_gpu():
    """integration/example.c: a plain C caller of the ABI builds against the header and the library; with no device its one
    compute call reports RDF_DEVICE_ERROR (no CPU fallback)."""
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([_build_c_example(d)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "gfx950" in p.stdout and "status 5" in p.stdout and "no CPU fallback" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
def test_c_example_runs_on_the_gpu():
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([_build_c_example(d)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "sum = 3.25 over 4 rows" in p.stdout, p.stdout + p.stderr


@pytest.mark.skipif(lib.device_count() > 0, reason="a GPU is visible")
def test_c_dist_example_links_without_a_gpu():
    """integration/example_dist.c (the N-GPU GROUP BY from plain C) builds against the header and the library
    alone — RCCL is not on its link line — and says so when there is no device."""
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([_build_c_example(d, "example_dist")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "no gfx950 device" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,expect", [(["peer", "4"], "ok: 1000 groups over 4 rank(s)"), (["rccl"], "ok: 1000 groups over 1 rank(s)")])
def test_c_dist_example_runs_on_the_gpu(args, expect):
    """the same program on the GPU box: four ranks sharing the one device over the peer-copy transport, and one rank on a
    real RCCL communicator (ncclCommInitAll through the library)."""
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([_build_c_example(d, "example_dist")] + args, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, RDF_COMM_TIMEOUT_S="60"))
    assert p.returncode == 0 and expect in p.stdout, p.stdout + p.stderr


def test_comm_entry_points_validate_before_the_device():
    """rdf_comm_* argument checks are values, not crashes: NULL communicators, bad rank counts, unknown transports."""
    so = lib.load()
    api = lib.api()
    so.rdf_comm_barrier.restype = C.c_int
    assert so.rdf_comm_barrier(None) == A.RDF_INVALID_ARGUMENT and b"null communicator" in so.rdf_last_error()
    assert so.rdf_comm_destroy(None) == A.RDF_OK
    h = C.c_void_p(0)
    uid = (C.c_uint8 * A.COMM_ID_BYTES)()
    assert so.rdf_comm_init_rank(C.c_int32(0), C.c_int32(0), uid, C.byref(h)) == A.RDF_INVALID_ARGUMENT
    assert so.rdf_comm_init_rank(C.c_int32(2), C.c_int32(2), uid, C.byref(h)) == A.RDF_INVALID_ARGUMENT
    hs = (C.c_void_p * 2)()
    assert so.rdf_comm_init_all(C.c_int32(2), (C.c_int32 * 2)(0, 0), C.c_int32(7), hs) == A.RDF_INVALID_ARGUMENT
    assert so.rdf_comm_init_all(C.c_int32(0), (C.c_int32 * 2)(0, 0), C.c_int32(A.COMM_PEER), hs) == A.RDF_INVALID_ARGUMENT
    if lib.device_count() == 0:
        with pytest.raises(A.RdfError) as ei:
            A.Comm.init_all(api, [0, 0], A.COMM_PEER)
        assert ei.value.status == A.RDF_DEVICE_ERROR


def test_frame_entry_points_validate_before_the_device():
    """rdf_frame_pin refuses host buffers and empty lists, and the _frame calls refuse a NULL handle — all as RDF_INVALID_ARGUMENT
    values, before any device work."""
    so = lib.load()
    api = lib.api()
    with pytest.raises(A.RdfError) as ei:      # host buffers are staged per call: nothing to pin
        A.PinnedFrame(api, [[A.HostArray.from_numpy(np.arange(4.0))]])
    assert ei.value.status == A.RDF_INVALID_ARGUMENT and "device-resident" in ei.value.message
    h = C.c_void_p(0)
    so.rdf_frame_pin.restype = C.c_int
    assert so.rdf_frame_pin(None, C.c_int32(1), C.c_int64(1), C.byref(h)) == A.RDF_INVALID_ARGUMENT and not h.value
    assert so.rdf_frame_pin(None, C.c_int32(1), C.c_int64(1), None) == A.RDF_INVALID_ARGUMENT
    e = A.Expr()
    c0 = e.col(0)
    nodes = e.c_array()
    prog = A.rdf_program(C.cast(nodes, C.POINTER(A.rdf_expr_node)), len(e.nodes), -1, 1, (C.c_int32 * A.MAX_VALUES)(c0), A.SINK_AGG)
    aggs = (A.rdf_agg_result * A.MAX_VALUES)()
    for fn, args in [(so.rdf_pipeline_frame, (C.byref(prog), None, None, aggs)),
                     (so.rdf_predicate_frame, (nodes, C.c_int32(len(e.nodes)), C.c_int32(c0), None, None)),
                     (so.rdf_group_pipeline_frame, (nodes, C.c_int32(len(e.nodes)), C.c_int32(-1), C.c_int32(c0), C.c_int32(4),
                                                    (C.c_int32 * 1)(c0), C.c_int32(1), None, None, None))]:
        fn.restype = C.c_int
        assert fn(*args) == A.RDF_INVALID_ARGUMENT
        assert b"null frame" in so.rdf_last_error()
    so.rdf_frame_release.restype = C.c_int
    assert so.rdf_frame_release(None) == A.RDF_OK      # releasing nothing is not an error (Drop of a frame that was never pinned)


def test_round5_entry_points_validate_before_the_device():
    """rdf_filter_pipeline (host-resident batches only, an expression, outputs) and rdf_pipeline_dist / rdf_pipeline_frame_dist (a live
    communicator, an aggregating program, a frame) answer bad arguments with RDF_INVALID_ARGUMENT and a message — no device needed."""
    so = lib.load()
    e = A.Expr()
    root = e.op("gt", e.col(0), e.scalar(0.5))
    nodes = e.c_array()
    so.rdf_filter_pipeline.restype = C.c_int
    col = A.HostArray.from_numpy(np.arange(8.0))
    arr = (A.rdf_array * 1)(col.c_struct())
    out = (A.rdf_out * 1)()
    # no expression / root out of range / no columns / no outputs
    assert so.rdf_filter_pipeline(None, C.c_int32(0), C.c_int32(0), arr, C.c_int32(1), C.c_int64(1), out) == A.RDF_INVALID_ARGUMENT
    assert so.rdf_filter_pipeline(nodes, C.c_int32(len(e.nodes)), C.c_int32(len(e.nodes)), arr, C.c_int32(1), C.c_int64(1), out) == A.RDF_INVALID_ARGUMENT
    assert so.rdf_filter_pipeline(nodes, C.c_int32(len(e.nodes)), C.c_int32(root), None, C.c_int32(1), C.c_int64(1), out) == A.RDF_INVALID_ARGUMENT
    assert so.rdf_filter_pipeline(nodes, C.c_int32(len(e.nodes)), C.c_int32(root), arr, C.c_int32(1), C.c_int64(1), None) == A.RDF_INVALID_ARGUMENT
    assert so.rdf_filter_pipeline(nodes, C.c_int32(len(e.nodes)), C.c_int32(root), arr, C.c_int32(0), C.c_int64(1), out) == A.RDF_INVALID_ARGUMENT
    # a device-resident batch belongs to rdf_filter_frame
    dev = (A.rdf_array * 1)(col.c_struct())
    dev[0].mem = A.MEM_DEVICE
    assert so.rdf_filter_pipeline(nodes, C.c_int32(len(e.nodes)), C.c_int32(root), dev, C.c_int32(1), C.c_int64(1), out) == A.RDF_INVALID_ARGUMENT
    assert b"rdf_filter_frame" in so.rdf_last_error()
    # zero batches: nothing to do, not an error
    assert so.rdf_filter_pipeline(nodes, C.c_int32(len(e.nodes)), C.c_int32(root), None, C.c_int32(1), C.c_int64(0), None) == A.RDF_OK
    c0 = e.col(0)
    prog = A.rdf_program(C.cast(nodes, C.POINTER(A.rdf_expr_node)), len(e.nodes), -1, 1, (C.c_int32 * A.MAX_VALUES)(c0), A.SINK_AGG)
    aggs = (A.rdf_agg_result * A.MAX_VALUES)()
    so.rdf_pipeline_dist.restype = C.c_int
    so.rdf_pipeline_frame_dist.restype = C.c_int
    assert so.rdf_pipeline_dist(None, C.byref(prog), arr, C.c_int32(1), C.c_int64(1), aggs) == A.RDF_INVALID_ARGUMENT
    assert b"null communicator" in so.rdf_last_error()
    assert so.rdf_pipeline_frame_dist(None, C.byref(prog), None, aggs) == A.RDF_INVALID_ARGUMENT


@pytest.mark.skipif(lib.device_count() > 0, reason="a GPU is visible")
def test_no_gpu_means_loud_device_error_not_a_fallback():
    api = lib.api()
    a = [A.HostArray.from_numpy(np.array([1.0, 2.0, 3.0]))]
    e = A.Expr()
    calls = [lambda: api.binary("add", a, a), lambda: api.unary("sin", a), lambda: api.sum(a),
             lambda: api.pipeline(e, [a], [e.col(0)]), lambda: api.filter(a, [A.HostArray.from_numpy(np.array([1, 0, 1], dtype=bool), dtype=A.BOOL)]),
             lambda: api.take(a, A.HostArray.from_numpy(np.array([0], dtype=np.uint32)))]
    for call in calls:
        with pytest.raises(A.RdfError) as ei:
            call()
        assert ei.value.status == A.RDF_DEVICE_ERROR
        assert "no CPU fallback" in ei.value.message


def test_product_never_references_the_oracle():
    """The shipped library and package must not link, import or call anything under oracle/."""
    so = subprocess.check_output(["nm", "-D", lib.LIB_PATH], text=True)
    assert "ora_" not in so
    pkg = os.path.join(ROOT, "rust_dataframe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f"{f} imports the oracle"
                assert "librdf_oracle" not in text and "ora_" not in text and "rdf_oracle" not in text, f"{f} references the oracle"
