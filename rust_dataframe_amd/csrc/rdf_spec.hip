// rdf_spec.hip — the exact-signature catalog of the specialised fused kernels and the basic f64 / i64 shape families;
// the kernel itself lives in rdf_spec_kernel.hip.h, the remaining shape families in rdf_spec_shapes.hip.
#include "rdf_spec_kernel.hip.h"

namespace rdfk {

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// the catalog

using D0 = Col<0, RDF_F64>; using D1 = Col<1, RDF_F64>; using D2 = Col<2, RDF_F64>;
using L0 = Col<0, RDF_I64>; using L1 = Col<1, RDF_I64>; using L3 = Col<3, RDF_I64>;
using W0 = Col<0, RDF_U64>; using W1 = Col<1, RDF_U64>;
using KD0 = Imm<0, RDF_F64>; using KL0 = Imm<0, RDF_I64>;
using F0 = Col<0, RDF_F32>; using F1 = Col<1, RDF_F32>; using I0 = Col<0, RDF_I32>; using I1 = Col<1, RDF_I32>;
using J0 = Col<0, RDF_U32>; using J1 = Col<1, RDF_U32>;
using KF0 = Imm<0, RDF_F32>; using KI0 = Imm<0, RDF_I32>;

template <int OP> static void reg_cmp_family() {
    // filter(x CMP c) -> aggregates of x / of another column (headline family, config C2)
    reg<Prog<Bin<OP, D0, KD0>, D0, None, SINK_AGG>>();
    reg<Prog<Bin<OP, D0, KD0>, D1, None, SINK_AGG>>();
    reg<Prog<Bin<OP, D0, KD0>, L1, None, SINK_AGG>>();
    reg<Prog<Bin<OP, L0, KD0>, L0, None, SINK_AGG>>();
    // BooleanFilter::eval_to_array of column CMP scalar / column CMP column -> mask
    reg<Prog<None, Bin<OP, D0, KD0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, L0, KD0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, D0, D1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, L0, L1>, None, SINK_STORE>>();
    // 4-byte columns: comparisons still happen in f64 (the scalar arrives as an f64 immediate)
    reg<Prog<Bin<OP, F0, KD0>, F0, None, SINK_AGG>>();
    reg<Prog<Bin<OP, I0, KD0>, I0, None, SINK_AGG>>();
    reg<Prog<None, Bin<OP, F0, KD0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, I0, KD0>, None, SINK_STORE>>();
}
template <int OP> static void reg_arith_family() {
    reg<Prog<None, Bin<OP, D0, D1>, None, SINK_STORE>>();   // ScalarFunctions::add/... f64
    reg<Prog<None, Bin<OP, L0, L1>, None, SINK_STORE>>();   // i64
    reg<Prog<None, Bin<OP, W0, W1>, None, SINK_STORE>>();   // u64
    reg<Prog<None, Bin<OP, D0, KD0>, None, SINK_STORE>>();  // column OP scalar ("add_scalar", config C1)
    reg<Prog<None, Bin<OP, L0, KL0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, F0, F1>, None, SINK_STORE>>();   // f32 / i32 / u32 (the reference's type matrix, src/evaluation.rs:107-238)
    reg<Prog<None, Bin<OP, I0, I1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, J0, J1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, F0, KF0>, None, SINK_STORE>>();
    reg<Prog<None, Bin<OP, I0, KI0>, None, SINK_STORE>>();
}
template <int OP> static void reg_unary_f64() {
    reg<Prog<None, Un<OP, D0>, None, SINK_STORE>>();                      // ScalarFunctions::<op>
    reg<Prog<None, Un<OP, Bin<RDF_OP_ADD, D0, KD0>>, None, SINK_AGG>>();  // sum(op(x + c)) — config C1 shape
    reg<Prog<None, Un<OP, D0>, None, SINK_AGG>>();
    reg<Prog<None, Un<OP, F0>, None, SINK_STORE>>();                      // f32
}

std::map<std::string, SpecEntry>& spec_registry() {
    static std::map<std::string, SpecEntry> r;
    return r;
}
static std::map<std::string, SpecEntry>& registry() { return spec_registry(); }

static void build_registry() {
    reg_shape_family_basic<RDF_F64, RDF_F64>();
    reg_shape_family_basic<RDF_I64, RDF_I64>();
    reg_shape_family_basic<RDF_F64, RDF_I64>();   // predicate on an i64 key, f64 measures
    reg_shape_family_basic<RDF_I64, RDF_F64>();
    spec_register_shapes1(); spec_register_shapes2(); spec_register_shapes3();
    spec_register_shapes4(); spec_register_shapes5(); spec_register_shapes6(); spec_register_shapes7(); spec_register_shapes8();
    // aggregates of a plain column (AggregateFunctions::sum/min/max/count/avg)
    reg<Prog<None, D0, None, SINK_AGG>>();
    reg<Prog<None, L0, None, SINK_AGG>>();
    reg<Prog<None, W0, None, SINK_AGG>>();
    reg<Prog<None, Cast<RDF_F64, L0>, None, SINK_AGG>>();  // avg of an i64 column
    reg<Prog<None, Cast<RDF_F64, W0>, None, SINK_AGG>>();
    reg<Prog<None, F0, None, SINK_AGG>>();
    reg<Prog<None, I0, None, SINK_AGG>>();
    reg<Prog<None, J0, None, SINK_AGG>>();
    reg_cmp_family<RDF_OP_GT>(); reg_cmp_family<RDF_OP_GE>(); reg_cmp_family<RDF_OP_EQ>();
    reg_cmp_family<RDF_OP_NE>(); reg_cmp_family<RDF_OP_LT>(); reg_cmp_family<RDF_OP_LE>();
    reg_arith_family<RDF_OP_ADD>(); reg_arith_family<RDF_OP_SUB>(); reg_arith_family<RDF_OP_MUL>(); reg_arith_family<RDF_OP_DIV>();
    reg<Prog<None, Bin<RDF_OP_ATAN2, D0, D1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<RDF_OP_HYPOT, D0, D1>, None, SINK_STORE>>();
    reg<Prog<None, Bin<RDF_OP_LOG, D0, D1>, None, SINK_STORE>>();
    reg_unary_f64<RDF_OP_ABS>(); reg_unary_f64<RDF_OP_ACOS>(); reg_unary_f64<RDF_OP_ASIN>(); reg_unary_f64<RDF_OP_ATAN>();
    reg_unary_f64<RDF_OP_CBRT>(); reg_unary_f64<RDF_OP_CEIL>(); reg_unary_f64<RDF_OP_COS>(); reg_unary_f64<RDF_OP_COSH>();
    reg_unary_f64<RDF_OP_DEGREES>(); reg_unary_f64<RDF_OP_EXP>(); reg_unary_f64<RDF_OP_EXPM1>(); reg_unary_f64<RDF_OP_FLOOR>();
    reg_unary_f64<RDF_OP_LOG10>(); reg_unary_f64<RDF_OP_LOG2>(); reg_unary_f64<RDF_OP_RADIANS>(); reg_unary_f64<RDF_OP_ROUND>();
    reg_unary_f64<RDF_OP_SIN>(); reg_unary_f64<RDF_OP_SINH>(); reg_unary_f64<RDF_OP_SQRT>(); reg_unary_f64<RDF_OP_TAN>();
    reg_unary_f64<RDF_OP_TANH>(); reg_unary_f64<RDF_OP_COT>(); reg_unary_f64<RDF_OP_SEC>(); reg_unary_f64<RDF_OP_CSC>();
    reg<Prog<None, Un<RDF_OP_ABS, L0>, None, SINK_STORE>>();
    reg<Prog<None, Un<RDF_OP_ABS, I0>, None, SINK_STORE>>();
    // casts between the 8-byte types (Function::Cast, src/evaluation.rs:296-315)
    reg<Prog<None, Cast<RDF_F64, L0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_I64, D0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_F64, W0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_U64, D0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_F32, I0>, None, SINK_STORE>>();
    reg<Prog<None, Cast<RDF_I32, F0>, None, SINK_STORE>>();
    // config C3: fused a*b+c -> min/max/count, plus the i64 key column's min/max/count, one pass over 4 columns
    using FMA = Bin<RDF_OP_ADD, Bin<RDF_OP_MUL, D0, D1>, D2>;
    reg<Prog<None, FMA, L3, SINK_AGG>>();
    reg<Prog<None, FMA, None, SINK_AGG>>();
    reg<Prog<None, FMA, None, SINK_STORE>>();
}

const SpecEntry* spec_lookup(const char* sig) {
    static bool built = (build_registry(), true);
    (void)built;
    auto it = registry().find(sig);
    return it == registry().end() ? nullptr : &it->second;
}

// (a signature the catalog does not hold may have been compiled at run time: rdf_jit.cpp)
int spec_rows_per_tile(const char* sig) {   // rows one wave iteration covers: tiles never straddle chunks
    const SpecEntry* e = spec_lookup(sig);
    if (e) return e->rows_per_tile;
    const JitKernel* j = jit_find(sig);
    return j ? j->rows_per_tile : 0;
}
bool spec_available(const char* sig) { return spec_lookup(sig) != nullptr; }   // the catalog proper; the caller asks jit_find itself
hipError_t launch_spec(const char* sig, const SpecArgs& a, int grid, hipStream_t s) {
    const SpecEntry* e = spec_lookup(sig);
    if (!e) {
        const JitKernel* j = jit_find(sig);
        if (!j) return hipErrorInvalidValue;
        const hipError_t e = jit_launch(*j, a, grid, s);
        if (e != hipSuccess) jit_mark_failed(sig);
        return e;
    }
    e->launch(a, grid, s);
    return hipGetLastError();
}
int spec_catalog_size() { spec_lookup(""); return (int)registry().size(); }

}  // namespace rdfk
