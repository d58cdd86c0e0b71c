// rdf_gspec.hip — ahead-of-time specialised kernels for the fused grouped sink (rdf_group_pipeline).
//
// Transformation::GroupAggregate over a small dense group domain after Calculate/Filter steps (TPC-H Q1 shape,
// BASELINE config C5; planned by Dataset::try_aggregate, src/expression.rs:114-221): one pass over the columns,
// predicate, group id and value expressions in registers, and per-lane REGISTER accumulators sum[g][v] for a
// compile-time number of groups G — a row adds into the accumulators of its group under the EXEC mask, so no
// atomics and no LDS traffic on the common path.  Rare rows (NULL group id, a NULL value) go through LDS
// atomics on the block's table.  Columns may have different element widths (f64 measures, i8 dictionary
// codes, date32): lane l of a wave-load owns the 2 consecutive rows 2l, 2l+1, so an 8-byte column is read
// with 16-byte loads, a 4-byte column with 8-byte loads, ... all fully coalesced.
//
// gspec_kernel<GProg>: tiles of kEvalTile = kBlock * 4 rows; wave w owns rows [256 w, +256) of the tile.
// Per-block tables go to HBM in the layout of group_words() and are folded by group_final_kernel.
#include "rdf_gspec_kernel.hip.h"

namespace rdfk {

// ------------------------------------------------------------------------------------------------
// the catalog

typedef void (*GSpecLaunch)(const GSpecArgs&, int, hipStream_t);

template <class P>
static void launch_gprog(const GSpecArgs& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)group_words(a.ngroups, P::NV) * 8;
    hipLaunchKernelGGL((gspec_kernel<P>), dim3(grid), dim3(kBlock), lds, s, a);
}
static std::map<std::string, GSpecLaunch>& gregistry() {
    static std::map<std::string, GSpecLaunch> r;
    return r;
}
template <class P>
static void greg() { gregistry()[P::sig()] = &launch_gprog<P>; }

template <int G> static void reg_q1() {
    // TPC-H Q1 (config C5): canonical columns in order of first use — ship date32, returnflag / linestatus i8
    // dictionary codes, quantity, extendedprice, discount, tax f64; literals: cutoff (compared in f64), 2, 1.0
    using SHIP = Col<0, RDF_I32>; using FLAG = Col<1, RDF_I8>; using STAT = Col<2, RDF_I8>;
    using QTY = Col<3, RDF_F64>; using PRICE = Col<4, RDF_F64>; using DISC = Col<5, RDF_F64>; using TAX = Col<6, RDF_F64>;
    using ONE = Imm<2, RDF_F64>;
    using PRED = Bin<RDF_OP_LE, SHIP, Imm<0, RDF_F64>>;
    using GID = Bin<RDF_OP_ADD, Bin<RDF_OP_MUL, Cast<RDF_I32, FLAG>, Imm<1, RDF_I32>>, Cast<RDF_I32, STAT>>;
    using DP = Bin<RDF_OP_MUL, PRICE, Bin<RDF_OP_SUB, ONE, DISC>>;
    using CH = Bin<RDF_OP_MUL, DP, Bin<RDF_OP_ADD, ONE, TAX>>;
    greg<GProg<G, PRED, GID, QTY, PRICE, DP, CH, DISC>>();
}
template <int KDT> static void reg_key_family() {
    // sum(v) [, sum(w)] GROUP BY k for a small-domain key column used directly as the group id
    using K = Col<0, KDT>;
    greg<GProg<8, None, K, Col<1, RDF_F64>>>();
    greg<GProg<8, None, K, Col<1, RDF_I64>>>();
    greg<GProg<8, None, K, Col<1, RDF_F64>, Col<2, RDF_F64>>>();
}

static void build_gregistry() {
    reg_q1<6>();
    reg_q1<8>();
    reg_key_family<RDF_I8>();
    reg_key_family<RDF_I32>();
    reg_key_family<RDF_I64>();
}

static GSpecLaunch gspec_lookup(const char* sig) {
    static bool built = (build_gregistry(), true);
    (void)built;
    auto it = gregistry().find(sig);
    return it == gregistry().end() ? nullptr : it->second;
}
bool gspec_available(const char* sig) { return gspec_lookup(sig) != nullptr; }
hipError_t launch_gspec(const char* sig, const GSpecArgs& a, int grid, hipStream_t s) {
    GSpecLaunch f = gspec_lookup(sig);
    if (!f) {   // not in the catalog: compiled at run time? (rdf_jit.cpp)
        const JitKernel* j = jit_find(sig);
        if (!j) return hipErrorInvalidValue;
        const hipError_t e = jit_launch_grouped(*j, a, grid, s);
        if (e != hipSuccess) jit_mark_failed(sig);
        return e;
    }
    f(a, grid, s);
    return hipGetLastError();
}
int gspec_catalog_size() { gspec_lookup(""); return (int)gregistry().size(); }

}  // namespace rdfk
