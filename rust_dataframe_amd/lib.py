"""Loader for librdf_mi355x.so — the product library (HIP kernels + C ABI).

There is no Python or CPU fallback: if the shared library is missing this module raises, and every
compute entry point of the library itself returns RDF_DEVICE_ERROR without a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RDF_LIB_PATH: another build of the same library (tools/ab_libs.py alternates two builds on one box); default: the in-tree build
LIB_PATH = os.environ.get("RDF_LIB_PATH") or os.path.join(_HERE, "librdf_mi355x.so")

# Every symbol include/rdf_mi355x.h declares (tests check the library exports all of them).
EXPORTS = [
    "rdf_version", "rdf_last_error", "rdf_device_count", "rdf_set_device", "rdf_set_stream", "rdf_synchronize",
    "rdf_dev_alloc", "rdf_dev_free", "rdf_copy_h2d", "rdf_copy_d2h", "rdf_host_alloc", "rdf_host_free", "rdf_host_register", "rdf_host_unregister", "rdf_copy_h2d_async", "rdf_copy_fence",
    "rdf_binary", "rdf_unary", "rdf_cast", "rdf_hour", "rdf_sum", "rdf_min", "rdf_max", "rdf_count", "rdf_avg",
    "rdf_predicate", "rdf_filter_count", "rdf_filter", "rdf_filter_columns", "rdf_filter_pipeline", "rdf_take", "rdf_list_contains", "rdf_list_position", "rdf_list_max", "rdf_list_min", "rdf_list_remove", "rdf_list_sort", "rdf_list_distinct", "rdf_list_except", "rdf_list_intersect", "rdf_list_union", "rdf_list_repeat", "rdf_pipeline", "rdf_stream_stats", "rdf_frame_pin", "rdf_frame_release", "rdf_pipeline_frame", "rdf_group_pipeline_frame", "rdf_predicate_frame", "rdf_frame_info", "rdf_frame_column", "rdf_filter_frame", "rdf_take_columns", "rdf_take_frame", "rdf_sort_frame", "rdf_groupby_agg_frame", "rdf_group_pipeline", "rdf_groupby_sum", "rdf_groupby_agg", "rdf_groupby_merge", "rdf_group_exchange_pack", "rdf_group_exchange_unpack", "rdf_row_exchange_pack", "rdf_row_exchange_unpack", "rdf_sort_to_indices", "rdf_equijoin_indices", "rdf_equijoin_indices_multi",
    "rdf_comm_unique_id", "rdf_comm_init_rank", "rdf_comm_init_all", "rdf_comm_destroy", "rdf_comm_info", "rdf_comm_barrier", "rdf_comm_allgather",
    "rdf_agg_combine", "rdf_pipeline_dist", "rdf_pipeline_frame_dist", "rdf_group_combine", "rdf_groupby_agg_dist", "rdf_groupby_agg_frame_dist",
    "rdf_fill_uniform_f64", "rdf_fill_uniform_i64", "rdf_fill_validity",
    "rdf_kernel_timing_reset", "rdf_kernel_timing_get", "rdf_probe_stream", "rdf_set_option", "rdf_spec_catalog_size", "rdf_jit_status", "rdf_last_kernel",
]

_lib = None
_api = None


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C rust_dataframe_amd/csrc). There is no fallback path.")
        # torch bundles its own libamdhip64/libhsa-runtime64; when it shares the process it must be loaded
        # FIRST so that both sides bind the same HIP runtime (two runtimes in one process cannot both own
        # the device: whichever comes second reports "no HIP GPUs").  A pure C/C++ host never needs torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.rdf_version.restype = C.c_char_p
        _lib.rdf_last_error.restype = C.c_char_p
        _lib.rdf_dev_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_int64]
        _lib.rdf_dev_free.argtypes = [C.c_void_p]
        _lib.rdf_copy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib.rdf_copy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib.rdf_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_int64]
        _lib.rdf_host_free.argtypes = [C.c_void_p]
        _lib.rdf_host_register.argtypes = [C.c_void_p, C.c_int64]
        _lib.rdf_host_unregister.argtypes = [C.c_void_p]
        _lib.rdf_copy_h2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _lib.rdf_set_stream.argtypes = [C.c_void_p]
        _lib.rdf_set_device.argtypes = [C.c_int32]
        _lib.rdf_kernel_timing_reset.argtypes = [C.c_int32]
    return _lib


def api() -> _abi.Api:
    """The product call layer (prefix rdf_)."""
    global _api
    if _api is None:
        _api = _abi.Api(load(), "rdf_")
    return _api


def _check(status: int):
    if status != _abi.RDF_OK:
        raise _abi.RdfError(status, (load().rdf_last_error() or b"").decode("utf-8", "replace"))


def version() -> str:
    return load().rdf_version().decode()


def device_count() -> int:
    n = C.c_int32(0)
    st = load().rdf_device_count(C.byref(n))
    return n.value if st == _abi.RDF_OK else 0


def set_device(dev: int):
    _check(load().rdf_set_device(dev))


def set_stream(hip_stream_ptr):
    _check(load().rdf_set_stream(C.c_void_p(hip_stream_ptr)))


def synchronize():
    _check(load().rdf_synchronize())


def set_option(name: str, value: int):
    load().rdf_set_option.argtypes = [C.c_char_p, C.c_int64]
    _check(load().rdf_set_option(name.encode(), value))


def probe_stream(kind: int, a: int, b: int = 0, c: int = 0, nbytes: int = 0, reps: int = 5):
    """-> (GB/s of bytes moved, shape): the best bare streaming kernel of `kind` (0 read, 1 copy, 2 two reads + one write) on device buffers."""
    gbps = C.c_double(0.0)
    shape = C.create_string_buffer(256)
    fn = load().rdf_probe_stream
    fn.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.c_char_p, C.c_int32]
    _check(fn(kind, a, b or None, c or None, nbytes, reps, C.byref(gbps), shape, 256))
    return gbps.value, shape.value.decode()


def last_kernel() -> str:
    load().rdf_last_kernel.restype = C.c_char_p
    return (load().rdf_last_kernel() or b"").decode()


def spec_catalog_size() -> int:
    return load().rdf_spec_catalog_size()


def kernel_timing_reset(enable: bool):
    _check(load().rdf_kernel_timing_reset(1 if enable else 0))


def kernel_timing_get():
    ms, n = C.c_double(0), C.c_int64(0)
    _check(load().rdf_kernel_timing_get(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def fill_uniform_f64(dev_ptr: int, n: int, seed: int, column_id: int, first_row: int, lo: float, hi: float):
    _check(load().rdf_fill_uniform_f64(C.c_void_p(dev_ptr), n, seed, column_id, first_row, lo, hi))


def fill_uniform_i64(dev_ptr: int, n: int, seed: int, column_id: int, first_row: int, lo: int, hi: int):
    _check(load().rdf_fill_uniform_i64(C.c_void_p(dev_ptr), n, seed, column_id, first_row, lo, hi))


def fill_validity(dev_ptr: int, nbits: int, seed: int, column_id: int, first_row: int, null_fraction: float):
    _check(load().rdf_fill_validity(C.c_void_p(dev_ptr), nbits, seed, column_id, first_row, null_fraction))


def stream_stats():
    """(slabs, bytes through the staging buffer, bytes copied straight out of page-locked memory) of this thread's last rdf_pipeline call."""
    a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    _check(load().rdf_stream_stats(C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def jit_status() -> str:
    load().rdf_jit_status.restype = C.c_char_p
    return (load().rdf_jit_status() or b"").decode()
