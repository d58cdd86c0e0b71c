"""rust_dataframe_amd — MI355X (gfx950) engine for rust-dataframe's Arrow compute hot path.

The product is librdf_mi355x.so (hand-written HIP kernels behind the C ABI of include/rdf_mi355x.h);
this package is the thin host-side mirror of the reference's interface for that path.
"""
from . import _abi  # noqa: F401
from ._abi import (BOOL, F32, F64, I8, I16, I32, I64, U8, U16, U32, U64, DeviceArray, Expr, HostArray,  # noqa: F401
                   RdfError)
