// rdf_spec_shapes.hip — shape-level specialised kernels beyond the basic f64 / i64 families of rdf_spec.hip, compiled in
// slices (-DRDF_SHAPE_TU=1..8) so the slices build in parallel:
//   1-3: three-level f64 / i64 trees (((l.l).l).l, (l.l).(l.l), ((l.l).l).(l.l)), plain and behind the two predicate forms;
//   4:   the basic shapes for u64 and the 4-byte types f32 / i32 / u32 (Evaluate::calculate's type matrix,
//        src/evaluation.rs:107-293);
//   5-6: three-level trees of the 4-byte types and u64;
//   7:   i16 / u16, basic and three-level;
//   8:   operands of different types (one cast leaf), plain casts, trig of a cast column: mixed element widths.
#include "rdf_spec_kernel.hip.h"

#ifndef RDF_SHAPE_TU
#error "compile with -DRDF_SHAPE_TU=1..8"
#endif

namespace rdfk {

#define RDF_CAT2(a, b) a##b
#define RDF_CAT(a, b) RDF_CAT2(a, b)
void RDF_CAT(spec_register_shapes, RDF_SHAPE_TU)() {
#if RDF_SHAPE_TU == 1
    reg_shape_family_deep<RDF_F64, RDF_F64>();
#elif RDF_SHAPE_TU == 2
    reg_shape_family_deep<RDF_I64, RDF_I64>();
#elif RDF_SHAPE_TU == 3
    reg_shape_family_deep<RDF_F64, RDF_I64>();   // predicate on an i64 key, f64 measures
    reg_shape_family_deep<RDF_I64, RDF_F64>();
#elif RDF_SHAPE_TU == 4
    reg_shape_family_basic<RDF_U64, RDF_U64>();
    reg_shape_family_basic<RDF_F32, RDF_F32>();
    reg_shape_family_basic<RDF_I32, RDF_I32>();
    reg_shape_family_basic<RDF_U32, RDF_U32>();
    reg_shape_family_basic<RDF_F32, RDF_I32>();
    reg_shape_family_basic<RDF_I32, RDF_F32>();
#elif RDF_SHAPE_TU == 5
    reg_shape_family_deep<RDF_F32, RDF_F32>();
    reg_shape_family_deep<RDF_I32, RDF_I32>();
#elif RDF_SHAPE_TU == 6
    reg_shape_family_deep<RDF_U32, RDF_U32>();
    reg_shape_family_deep<RDF_U64, RDF_U64>();
#elif RDF_SHAPE_TU == 7
    // Int16 / UInt16: the narrowest types of Evaluate::calculate's matrix (src/evaluation.rs:107-238; Int8 / UInt8 are
    // rejected there, :239).  Eight rows per 16-byte vector.
    reg_shape_family_basic<RDF_I16, RDF_I16>();
    reg_shape_family_basic<RDF_U16, RDF_U16>();
    reg_shape_family_deep<RDF_I16, RDF_I16>();
    reg_shape_family_deep<RDF_U16, RDF_U16>();
    // Int8 / UInt8: sixteen rows per 16-byte vector
    reg_byte_family<RDF_I8>();
    reg_byte_family<RDF_U8>();
#else
    // operands of different types: a OP cast(b), cast(b), trig(cast(b)) over every pair of the eight types
    reg_cast_family<RDF_F64>(); reg_cast_family<RDF_I64>(); reg_cast_family<RDF_U64>(); reg_cast_family<RDF_F32>();
    reg_cast_family<RDF_I32>(); reg_cast_family<RDF_U32>(); reg_cast_family<RDF_I16>(); reg_cast_family<RDF_U16>();
#endif
}

}  // namespace rdfk
