// Plan builders (src/operation/scalar.rs) — no device needed.
#include "mini_test.hpp"
#include "rdf_frame.hpp"

using namespace rdf::plan;
using rdf::DataType;
using Column = rdf::plan::Column;

// scalar_operations, src/operation/scalar.rs:324-341: add(Int64 a, Int32 b) = Cast(b -> Int64) then Add
TEST(scalar_operations) {
    Column a{"a", DataType::Int64}, b{"b", DataType::Int32};
    auto add = AddOperation::transform({a, b}, std::nullopt, std::nullopt);
    CHECK_EQ(debug(add),
             std::string("[Calculation { name: \"cast\", inputs: [Column { name: \"b\", column_type: Scalar(Int32) }], output: Column { name: "
                         "\"b\", column_type: Scalar(Int64) }, function: Cast }, Calculation { name: \"add\", inputs: [Column { name: \"a\", "
                         "column_type: Scalar(Int64) }, Column { name: \"b\", column_type: Scalar(Int64) }], output: Column { name: \"add(a, b)\", "
                         "column_type: Scalar(Int64) }, function: Scalar(Add) }]"));
}

TEST(add_same_type_is_one_step_and_named) {
    Column a{"lat", DataType::Float64}, b{"lng", DataType::Float64};
    auto add = AddOperation::transform({a, b}, std::nullopt, std::nullopt);
    CHECK_EQ(add.size(), 1u);
    CHECK_EQ(add[0].output.name, std::string("add(lat, lng)"));
    auto named = AddOperation::transform({a, b}, std::string("sum"), std::nullopt);
    CHECK_EQ(named[0].output.name, std::string("sum"));
    CHECK_THROWS(AddOperation::transform({a}, std::nullopt, std::nullopt));
}

// SubtractOperation's mismatch branch emits Add in the reference: the intent is Subtract
TEST(subtract_mismatch_is_subtract) {
    Column a{"a", DataType::Int64}, b{"b", DataType::Int32};
    auto sub = SubtractOperation::transform({a, b}, std::nullopt, std::nullopt);
    CHECK_EQ(sub.size(), 2u);
    CHECK_EQ(sub[1].function.debug(), std::string("Scalar(Subtract)"));
    CHECK_EQ(sub[1].output.name, std::string("subtract(a, b)"));
}

// SinOperation, src/operation/scalar.rs:227-318: integers are cast to Float64 first; names "sin(x as datatype)"
TEST(sin_operation) {
    auto f = SinOperation::transform({Column{"x", DataType::Float32}}, std::nullopt, std::nullopt);
    CHECK_EQ(f.size(), 1u);
    CHECK_EQ(f[0].output.name, std::string("sin(x as datatype)"));
    CHECK(f[0].output.data_type == DataType::Float32);
    auto i = SinOperation::transform({Column{"k", DataType::Int64}}, std::nullopt, std::nullopt);
    CHECK_EQ(i.size(), 2u);
    CHECK_EQ(i[0].function.debug(), std::string("Cast"));
    CHECK_EQ(i[0].output.name, std::string("cast(k as datatype)"));
    CHECK(i[0].output.data_type == DataType::Float64);
    CHECK_EQ(i[1].inputs[0].name, std::string("cast(k as datatype)"));
    CHECK(i[1].output.data_type == DataType::Float64);
    CHECK_THROWS(SinOperation::transform({Column{"s", DataType::Utf8}}, std::nullopt, std::nullopt));
}

TEST(cast_operation) {
    auto c = CastOperation::transform({Column{"b", DataType::Int32}}, std::nullopt, DataType::Int64);
    CHECK_EQ(c[0].output.name, std::string("cast(b as datatype)"));
    CHECK_THROWS(CastOperation::transform({Column{"b", DataType::Int32}}, std::nullopt, std::nullopt));
}

// Calculation::calculate, src/expression.rs:433-499
TEST(calculation_dispatch) {
    Dataset ds{"d", {Column{"lat", DataType::Float64}, Column{"lng", DataType::Float64}}};
    auto t = calculate(ds, {"lat", "lng"}, Function::Scalar_(ScalarFunction::Add), std::string("sum"), std::nullopt);
    CHECK_EQ(t.size(), 1u);
    CHECK(t[0].kind == Transformation::Calculate);
    CHECK_THROWS(calculate(ds, {"nope"}, Function::Scalar_(ScalarFunction::Add), std::nullopt, std::nullopt));
    // Cosecant / Secant / Cotangent panic in the reference's builder (:487-489); here they plan like the sine
    auto csc = calculate(ds, {"lat"}, Function::Scalar_(ScalarFunction::Cosecant), std::nullopt, std::nullopt);
    CHECK_EQ(csc.size(), 1u);
    CHECK_EQ(csc[0].calc.name, std::string("csc"));
    CHECK(csc[0].calc.output.data_type == DataType::Float64);
    CHECK_THROWS(calculate(ds, {"lat", "lng"}, Function::Scalar_(ScalarFunction::Secant), std::nullopt, std::nullopt));   // one input
    // Dataset::append_column replaces in place (src/expression.rs:95-112)
    auto d2 = ds.append_column(Column{"lat", DataType::Int64});
    CHECK_EQ(d2.columns.size(), 2u);
    CHECK(d2.columns[0].data_type == DataType::Int64);
}

// test_group_aggregate (src/lazyframe.rs:513-540) plans max(lat), max(lng) by city on the cities file and prints the plan:
// the planned output is Dataset::try_aggregate's (src/expression.rs:114-221)
TEST(test_group_aggregate_plan) {
    Dataset cities;
    cities.name = "source";
    cities.columns = {Column{"city", DataType::Utf8}, Column{"lat", DataType::Float64}, Column{"lng", DataType::Float64}};
    const Dataset out = try_aggregate(cities, {"city"}, {Aggregation{AggregateFunction::Max, {"lat", "lng"}}});
    CHECK_EQ(out.name, std::string("aggregated_dataset"));
    CHECK_EQ(out.columns.size(), 3u);
    CHECK_EQ(out.columns[0].debug(), std::string("Column { name: \"city\", column_type: Scalar(Utf8) }"));
    CHECK_EQ(out.columns[1].debug(), std::string("Column { name: \"max(lat)\", column_type: Scalar(Float64) }"));
    CHECK_EQ(out.columns[2].name, std::string("max(lng)"));
    const Dataset counts = try_aggregate(cities, {}, {Aggregation{AggregateFunction::Count, {"city"}}, Aggregation{AggregateFunction::Avg, {"lat"}}});
    CHECK_EQ(counts.columns[0].debug(), std::string("Column { name: \"count(city)\", column_type: Scalar(UInt32) }"));
    CHECK_EQ(counts.columns[1].name, std::string("avg(lat)"));
    CHECK_THROWS(try_aggregate(cities, {"town"}, {}));                                                   // Grouping column "town" does not exist
    CHECK_THROWS(try_aggregate(cities, {}, {Aggregation{AggregateFunction::Sum, {"nope"}}}));            // Aggregating column "nope" does not exist
    CHECK_THROWS(try_aggregate(cities, {}, {Aggregation{AggregateFunction::StdDev, {"lat"}}}));          // Aggregation not yet supported
}

// Dataset::try_join (src/expression.rs:223-285), the planning half of test_lazy_join (src/lazyframe.rs:410-470)
TEST(test_join_plan) {
    Dataset a, b;
    a.columns = {Column{"town", DataType::Utf8}, Column{"lat", DataType::Float64}, Column{"sin_lat", DataType::Float64}};
    b.columns = {Column{"city", DataType::Utf8}, Column{"lat", DataType::Float64}};
    const Dataset j = try_join(a, b, {{"town", "city"}});
    CHECK_EQ(j.name, std::string("joined_dataframe"));
    CHECK_EQ(j.columns.size(), 5u);
    CHECK_EQ(j.columns[1].name, std::string("a.lat"));
    CHECK_EQ(j.columns[2].name, std::string("sin_lat"));
    CHECK_EQ(j.columns[3].name, std::string("city"));
    CHECK_EQ(j.columns[4].name, std::string("b.lat"));
    CHECK_THROWS(try_join(a, b, {{"nope", "city"}}));   // not in table A
    CHECK_THROWS(try_join(a, b, {{"town", "nope"}}));   // not in table B
    CHECK_THROWS(try_join(a, b, {{"x", "y"}}));         // in neither
    CHECK_THROWS(try_join(a, b, {{"town", "lat"}}));    // incompatible types
}

int main() { return run_all(); }
