// C++ mirrors of the reference's in-file #[test] functions for the hot path, run on the device through
// rdf_frame.hpp (-> librdf_mi355x.so), plus fused-vs-oracle parity of the batch loop.  The oracle
// (librdf_oracle.so) is linked here as the checker only.
#include <map>
#include <random>
#include <thread>
#include <tuple>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#include "mini_test.hpp"
#include "rdf_frame.hpp"
#include "rdf_oracle.h"

using namespace rdf;
namespace P = rdf::plan;

static std::string g_csv = "tests/golden/uk_cities_with_headers.csv";
static std::string g_arrow = "tests/golden/mixed_batches.arrow";

template <class T> static std::vector<T> host(const ArrayRef& a) { return a->values_to_host<T>(); }

// ---------------------------------------------------------------- src/functions/scalar.rs:565-602
TEST(test_primitive_array_abs_f64) {
    auto a = Array::from_vec<double>({-5.2, -6.1, 7.3, -8.6, -0.0});
    auto c = host<double>(ScalarFunctions::abs({a})[0]);
    CHECK_EQ(c, (std::vector<double>{5.2, 6.1, 7.3, 8.6, 0.0}));
}
TEST(test_primitive_array_abs_i32) {
    auto a = Array::from_vec<int32_t>({-5, -6, 7, -8, 0});
    CHECK_EQ(host<int32_t>(ScalarFunctions::abs({a})[0]), (std::vector<int32_t>{5, 6, 7, 8, 0}));
}
TEST(test_primitive_array_acos_f64) {
    auto c = host<double>(ScalarFunctions::acos({Array::from_vec<double>({-0.2, 0.25, 0.75})})[0]);
    CHECK_NEAR(c[0], 1.7721542475852274, 1e-15); CHECK_NEAR(c[1], 1.318116071652818, 1e-15); CHECK_NEAR(c[2], 0.7227342478134157, 1e-15);
}
TEST(test_primitive_array_cos_f64) {
    auto c = host<double>(ScalarFunctions::cos({Array::from_vec<double>({-0.2, 0.25, 0.75})})[0]);
    CHECK_NEAR(c[0], 0.9800665778412416, 1e-15); CHECK_NEAR(c[1], 0.9689124217106447, 1e-15); CHECK_NEAR(c[2], 0.7316888688738209, 1e-15);
}

// ---------------------------------------------------------------- src/functions/aggregate.rs:123-146
TEST(test_aggregate_count) {
    auto a = Array::from_vec<int32_t>({5, 6, 7, 8, 9});
    CHECK_EQ(*AggregateFunctions::count(ChunkedArray::from_arrays({a})), 5);
}
TEST(test_aggregate_mean) {
    auto a = Array::from_vec<int32_t>({0, 1, 2, 3, 4});
    auto b = Array::from_vec<int32_t>({5, 6, 7, 8, 9});
    CHECK_EQ(*AggregateFunctions::avg(ChunkedArray::from_arrays({a, b})), 4.5);
    const std::vector<bool> valid{1, 0, 1, 0, 1, 1, 1};
    auto d = Array::from_vec<int32_t>({0, 0, 1, 0, 2, 3, 4}, &valid);
    CHECK_EQ(*AggregateFunctions::avg(ChunkedArray::from_arrays({d, b})), 4.5);
}

// ---------------------------------------------------------------- src/dataframe.rs:782-836
TEST(test_dataframe_ops) {
    DataFrame df = DataFrame::from_csv(g_csv);
    CHECK_EQ(df.num_columns(), 3u);
    CHECK_EQ(df.num_rows(), 37);
    auto lat = df.column_by_name("lat").data().chunks(), lng = df.column_by_name("lng").data().chunks();
    auto add = host<double>(ScalarFunctions::add(lat, lng)[0]);
    CHECK(std::fabs(add[0] - 54.31776) < 1e-4);
    CHECK_EQ(host<double>(ScalarFunctions::abs(lng)[0])[0], 3.335724);
    // with_column appends (an existing name is dropped and re-added last, :97-113)
    df = df.with_column("lat_lng", Column::from_arrays(ScalarFunctions::add(lat, lng), Field{"lat_lng", DataType::Float64, true}));
    CHECK_EQ(df.num_columns(), 4u);
    df = df.with_column("lat", df.column_by_name("lat_lng"));
    CHECK_EQ(df.num_columns(), 4u);
    CHECK_EQ(df.schema().fields.back().name, std::string("lat"));
}

// test_create_empty_dataframe / test_read_csv_to_dataframe / test_increasing_id (src/dataframe.rs:737-750, 945-960)
TEST(test_create_empty_and_read_csv_and_increasing_id) {
    DataFrame empty = DataFrame::empty();
    CHECK_EQ(empty.num_columns(), 0u);
    CHECK_EQ(empty.schema().fields.size(), 0u);
    DataFrame dataframe = DataFrame::from_csv(g_csv);
    CHECK_EQ(dataframe.num_columns(), 3u);
    CHECK_EQ(dataframe.num_rows(), 37);
    dataframe = dataframe.limit(10).with_id("id");
    const Column& id = dataframe.column_by_name("id");
    CHECK(id.data_type() == DataType::UInt64);
    CHECK_EQ(id.name(), std::string("id"));
    CHECK_EQ(host<uint64_t>(id.data().chunk(0)), (std::vector<uint64_t>{1, 2, 3, 4, 5, 6, 7, 8, 9, 10}));
}

// ---------------------------------------------------------------- src/lazyframe.rs:324-408, src/evaluation.rs:359-434
TEST(test_lazy_pipeline) {
    LazyFrame frame = LazyFrame::read(DataFrame::from_csv(g_csv));
    frame = frame.with_column_renamed("city", "town");
    frame = frame.with_column("sin_lat", P::Function::Scalar_(P::ScalarFunction::Sine), {"lat"});
    frame = frame.with_column("sin_lng", P::Function::Scalar_(P::ScalarFunction::Sine), {"lng"});
    DataFrame df = frame.evaluate();
    CHECK_EQ(df.num_columns(), 5u);
    CHECK_EQ(df.num_rows(), 37);
    CHECK(df.has_column("town") && !df.has_column("city"));
    CHECK_NEAR(df.column_by_name("sin_lat").data().chunk(0)->value<double>(0), 0.8933816410476535, 1e-12);
    CHECK_NEAR(df.column_by_name("sin_lng").data().chunk(0)->value<double>(0), 0.1929142713855381, 1e-12);
}
// ScalarFunction::{Cotangent, Secant, Cosecant} (src/expression.rs:670-672) through the lazy plan: the reference's builder
// panics on them (:487-489); here they are planned like the sine and run as RDF_OP_COT / SEC / CSC in the fused batch loop
TEST(test_reciprocal_trig_functions) {
    LazyFrame frame = LazyFrame::read(DataFrame::from_csv(g_csv));
    frame = frame.with_column("cot_lat", P::Function::Scalar_(P::ScalarFunction::Cotangent), {"lat"});
    frame = frame.with_column("sec_lat", P::Function::Scalar_(P::ScalarFunction::Secant), {"lat"});
    frame = frame.with_column("csc_lng", P::Function::Scalar_(P::ScalarFunction::Cosecant), {"lng"});
    CHECK_EQ(frame.output().columns.size(), 6u);
    DataFrame df = frame.evaluate();
    auto lat = host<double>(df.column_by_name("lat").data().chunk(0)), lng = host<double>(df.column_by_name("lng").data().chunk(0));
    auto cot = host<double>(df.column_by_name("cot_lat").data().chunk(0)), sec = host<double>(df.column_by_name("sec_lat").data().chunk(0));
    auto csc = host<double>(df.column_by_name("csc_lng").data().chunk(0));
    for (size_t i = 0; i < lat.size(); ++i) {
        CHECK_NEAR(cot[i], 1.0 / std::tan(lat[i]), 1e-12 * std::fabs(cot[i]));
        CHECK_NEAR(sec[i], 1.0 / std::cos(lat[i]), 1e-12 * std::fabs(sec[i]));
        CHECK_NEAR(csc[i], 1.0 / std::sin(lng[i]), 1e-12 * std::fabs(csc[i]));
    }
    auto a = Array::from_vec<double>({0.5, -1.25, 0.0});
    auto c = host<double>(ScalarFunctions::cot({a})[0]);
    CHECK_NEAR(c[0], 1.0 / std::tan(0.5), 1e-15);
    CHECK(std::isinf(c[2]) && c[2] > 0);
    CHECK_NEAR(host<double>(ScalarFunctions::sec({a})[0])[1], 1.0 / std::cos(-1.25), 1e-14);
    CHECK_NEAR(host<double>(ScalarFunctions::csc({a})[0])[1], 1.0 / std::sin(-1.25), 1e-14);
}
// test_projection (src/lazyframe.rs:472-511): rename, two sine columns, select three, drop one — evaluated here
TEST(test_projection) {
    LazyFrame frame = LazyFrame::read(DataFrame::from_csv(g_csv));
    frame = frame.with_column_renamed("city", "town");
    frame = frame.with_column("sin_lat", P::Function::Scalar_(P::ScalarFunction::Sine), {"lat"});
    frame = frame.with_column("sin_lng", P::Function::Scalar_(P::ScalarFunction::Sine), {"lng"});
    frame = frame.select({"town", "sin_lat", "sin_lng"});
    frame = frame.drop({"town"});
    CHECK_EQ(frame.output().columns.size(), 2u);
    DataFrame df = frame.evaluate();
    CHECK_EQ(df.num_columns(), 2u);
    CHECK_EQ(df.num_rows(), 37);
    CHECK_EQ(df.schema().fields[0].name, std::string("sin_lat"));
    CHECK_EQ(df.schema().fields[1].name, std::string("sin_lng"));
    CHECK_NEAR(df.column(0).data().chunk(0)->value<double>(0), 0.8933816410476535, 1e-12);
    // test_group_aggregate (:513-540) plans max(lat), max(lng) by city: the plan's output schema, and the evaluation error
    using AF = P::AggregateFunction;
    LazyFrame agg = LazyFrame::read(DataFrame::from_csv(g_csv)).aggregate({"city"}, {{AF::Max, {"lat", "lng"}}});
    CHECK_EQ(agg.output().name, std::string("aggregated_dataset"));
    CHECK_EQ(agg.output().columns[1].name, std::string("max(lat)"));
    CHECK_THROWS(agg.evaluate());   // "aggregations not supported" in the reference (evaluation.rs:73); a string grouping column here
    CHECK_THROWS(LazyFrame::read(DataFrame::from_csv(g_csv)).aggregate({"town"}, {{AF::Max, {"lat"}}}));   // Grouping column "town" does not exist
}
// A plan that starts at a Reader (LazyFrame::read(Computation), src/lazyframe.rs:25-38) is optimised before it runs
// (src/optimiser.rs): select + limit end up in the CSV reader's projection / max_records, so only 2 columns x 32 rows are
// parsed and uploaded; the result equals the same steps over the fully loaded frame.
TEST(test_reader_plan_is_optimised_then_evaluated) {
    P::CsvReadOptions o;
    o.batch_size = 1024;
    const P::Computation read = P::Computation::compute_read(P::Reader::Csv_(g_csv, o));
    LazyFrame frame = LazyFrame::read(read).select({"city", "lat"}).limit(32);
    CHECK_EQ(frame.optimised().size(), 1u);
    DataFrame df = frame.evaluate();
    CHECK_EQ(df.num_columns(), 2u);
    CHECK_EQ(df.num_rows(), 32);
    DataFrame full = DataFrame::from_csv(g_csv);
    CHECK_EQ(host<double>(df.column_by_name("lat").data().chunk(0)), host<double>(full.limit(32).column_by_name("lat").data().chunk(0)));
    CHECK_EQ((*df.column_by_name("city").data().chunk(0)->strings)[31], (*full.column_by_name("city").data().chunk(0)->strings)[31]);
    // computed columns and a filter behind the read: the select reaches the reader, the rest runs fused on the device
    LazyFrame g = LazyFrame::read(read)
                      .select({"lat", "lng"})
                      .with_column("sin_lat", P::Function::Scalar_(P::ScalarFunction::Sine), {"lat"})
                      .filter(BooleanFilter::gt(BooleanFilter::column("lng"), BooleanFilter::scalar(Scalar(-2.0))));
    DataFrame got = g.evaluate();
    DataFrame want = LazyFrame::read(full).select({"lat", "lng"})
                         .with_column("sin_lat", P::Function::Scalar_(P::ScalarFunction::Sine), {"lat"})
                         .filter(BooleanFilter::gt(BooleanFilter::column("lng"), BooleanFilter::scalar(Scalar(-2.0)))).evaluate();
    CHECK_EQ(got.num_columns(), 3u);
    CHECK_EQ(got.num_rows(), want.num_rows());
    CHECK(got.num_rows() > 0 && got.num_rows() < 37);
    CHECK_EQ(host<double>(got.column_by_name("sin_lat").data().chunk(0)), host<double>(want.column_by_name("sin_lat").data().chunk(0)));
    // a projection that is not a prefix of the file's columns (the reference's index bug, not copied)
    DataFrame tail = LazyFrame::read(read).select({"lat", "lng"}).evaluate();
    CHECK_EQ(tail.schema().fields[0].name, std::string("lat"));
    CHECK_EQ(tail.schema().fields[1].name, std::string("lng"));
    CHECK_EQ(host<double>(tail.column(1).data().chunk(0)), host<double>(full.column_by_name("lng").data().chunk(0)));
}
TEST(test_with_columns) {
    LazyFrame frame = LazyFrame::read(DataFrame::from_csv(g_csv));
    frame = frame.with_column("sum", P::Function::Scalar_(P::ScalarFunction::Add), {"lat", "lng"});
    DataFrame df = frame.evaluate();
    CHECK_EQ(df.column_by_name("sum").data().chunk(0)->value<double>(0), 57.653484 - 3.335724);
}
TEST(test_lazy_evaluation) {
    LazyFrame frame = LazyFrame::read(DataFrame::from_csv(g_csv));
    frame = frame.with_column_renamed("city", "town");
    frame = frame.with_column("sin_lat", P::Function::Scalar_(P::ScalarFunction::Sine), {"lat"});
    frame = frame.with_column("sin_lng", P::Function::Scalar_(P::ScalarFunction::Sine), {"lng"});
    DataFrame all = frame.evaluate();
    frame = frame.limit(25);
    DataFrame df = frame.evaluate();
    CHECK_EQ(df.num_columns(), 5u);
    CHECK_EQ(df.num_rows(), 25);
    {   // the limit is taken on the source before the pending sines run (limit push-down): same rows, same values
        auto full = host<double>(all.column_by_name("sin_lng").data().chunk(0)), lim = host<double>(df.column_by_name("sin_lng").data().chunk(0));
        CHECK_EQ(lim.size(), 25u);
        CHECK(std::equal(lim.begin(), lim.end(), full.begin()));
        // behind a filter the limit counts FILTERED rows
        DataFrame fl = LazyFrame::read(DataFrame::from_csv(g_csv)).filter(BooleanFilter::gt(BooleanFilter::column("lat"), BooleanFilter::scalar(Scalar(55.0)))).limit(3).evaluate();
        CHECK_EQ(fl.num_rows(), 3);
        for (double v : host<double>(fl.column_by_name("lat").data().chunk(0))) CHECK(v > 55.0);
    }
    // ops after a limit see sliced (offset) arrays
    DataFrame lim = DataFrame::from_csv(g_csv).limit(30).limit(10);
    auto s = host<double>(ScalarFunctions::sin(lim.column_by_name("lat").data().chunks())[0]);
    CHECK_EQ(s.size(), 10u);
    CHECK_NEAR(s[0], 0.8933816410476535, 1e-12);
}

// ---------------------------------------------------------------- src/dataframe.rs:963-1003 (the take half of test_sort)
TEST(test_sort_take) {
    const std::vector<bool> valid{1, 1, 0, 1, 1, 1};
    auto a = Array::from_vec<int32_t>({1, 1, 0, 3, 3, 4}, &valid);
    auto b = Array::from_vec<uint8_t>({9, 5, 6, 7, 4, 8});
    DataFrame frame = DataFrame::from_columns({Column::from_arrays({a}, Field{"a", DataType::Int32, true}), Column::from_arrays({b}, Field{"b", DataType::UInt8, false})});
    // lexsort_to_indices(a desc, b asc, nulls last) = [5, 4, 3, 1, 0, 2]
    DataFrame sorted = frame.take(Array::from_vec<uint32_t>({5, 4, 3, 1, 0, 2}));
    auto ac = sorted.column(0).data().chunks();
    CHECK_EQ(ac.size(), 1u);
    CHECK_EQ(ac[0]->valid_to_host(), (std::vector<bool>{1, 1, 1, 1, 1, 0}));
    auto av = host<int32_t>(ac[0]);
    CHECK_EQ(std::vector<int32_t>(av.begin(), av.begin() + 5), (std::vector<int32_t>{4, 3, 3, 1, 1}));
    CHECK_EQ(host<uint8_t>(sorted.column(1).data().chunk(0)), (std::vector<uint8_t>{8, 4, 7, 5, 9, 6}));
}

// the whole of test_sort: DataFrame::sort = lexsort_to_indices + take (src/dataframe.rs:963-1003)
TEST(test_sort) {
    const std::vector<bool> valid{1, 1, 0, 1, 1, 1};
    auto a = Array::from_vec<int32_t>({1, 1, 0, 3, 3, 4}, &valid);
    auto b = Array::from_vec<uint8_t>({9, 5, 6, 7, 4, 8});
    DataFrame frame = DataFrame::from_columns({Column::from_arrays({a}, Field{"a", DataType::Int32, true}), Column::from_arrays({b}, Field{"b", DataType::UInt8, false})});
    DataFrame sorted = frame.sort({{"a", true, false}, {"b", false, false}});
    auto ac = sorted.column(0).data().chunks();
    CHECK_EQ(ac.size(), 1u);
    CHECK(ac[0]->is_null(5));
    auto av = host<int32_t>(ac[0]);
    CHECK_EQ(std::vector<int32_t>(av.begin(), av.begin() + 5), (std::vector<int32_t>{4, 3, 3, 1, 1}));
    CHECK_EQ(host<uint8_t>(sorted.column(1).data().chunk(0)), (std::vector<uint8_t>{8, 4, 7, 5, 9, 6}));
    CHECK_THROWS(frame.sort({}));  // "Sort criteria cannot be empty" (:195-199)
    // through the lazy plan
    DataFrame lz = LazyFrame::read(frame).sort({"a", "b"}, {true, false}).evaluate();
    CHECK_EQ(host<uint8_t>(lz.column(1).data().chunk(0)), (std::vector<uint8_t>{8, 4, 7, 5, 9, 6}));
}

// test_left_join / test_right_join / test_inner_join (src/dataframe.rs:1006-1060) on the numeric columns of
// join_test_j1 / join_test_j2 (sql/postgresql/002.sql) — no Postgres needed
TEST(test_joins) {
    const std::vector<bool> av{0, 1, 1, 0, 0, 1, 1};
    DataFrame j1 = DataFrame::from_columns({Column::from_arrays({Array::from_vec<int32_t>({0, 2, 3, 0, 0, 6, 6}, &av)}, Field{"a", DataType::Int32, true}),
                                            Column::from_arrays({Array::from_vec<int32_t>({1, 2, 3, 4, 5, 6, 60})}, Field{"b", DataType::Int32, false})});
    const std::vector<bool> fv{1, 1, 1, 1, 0, 1, 1, 1, 1};
    DataFrame j2 = DataFrame::from_columns({Column::from_arrays({Array::from_vec<int32_t>({1, 2, 3, 4, 4, 4, 5, 6, 7})}, Field{"d", DataType::Int32, false}),
                                            Column::from_arrays({Array::from_vec<double>({1.1, 2.2, INFINITY, NAN, 0.0, 4.0, 5.0, 6.0, 7.000000000001}, &fv)}, Field{"f", DataType::Float64, true})});
    using JT = DataFrame::JoinType;
    DataFrame l = j1.join(j2, {JT::LeftJoin, {{"b", "d"}}});
    CHECK_EQ(l.num_columns(), 4u);
    CHECK_EQ(l.num_rows(), 9);
    DataFrame r = j1.join(j2, {JT::RightJoin, {{"a", "d"}}});
    CHECK_EQ(r.num_rows(), 10);
    DataFrame i = j1.join(j2, {JT::InnerJoin, {{"a", "d"}}});
    CHECK_EQ(i.num_rows(), 4);
    CHECK_EQ(i.num_columns(), 4u);
    CHECK_EQ(host<int32_t>(i.column_by_name("a").data().chunk(0)), (std::vector<int32_t>{2, 3, 6, 6}));
    CHECK_EQ(host<int32_t>(i.column_by_name("d").data().chunk(0)), (std::vector<int32_t>{2, 3, 6, 6}));
    CHECK_EQ(host<double>(i.column_by_name("f").data().chunk(0)), (std::vector<double>{2.2, INFINITY, 6.0, 6.0}));
}

// test_lazy_join (src/lazyframe.rs:410-470) builds rename -> two sine columns -> inner join of two reads of the cities
// file (on the city names) and only prints the plan.  Here the same plan shape on the numeric join fixture, evaluated:
// computed columns survive the join, a name present on both sides is prefixed, try_join's errors are raised.
TEST(test_lazy_join) {
    const std::vector<bool> av{0, 1, 1, 0, 0, 1, 1};
    DataFrame j1 = DataFrame::from_columns({Column::from_arrays({Array::from_vec<int32_t>({0, 2, 3, 0, 0, 6, 6}, &av)}, Field{"a", DataType::Int32, true}),
                                            Column::from_arrays({Array::from_vec<double>({1, 2, 3, 4, 5, 6, 60})}, Field{"x", DataType::Float64, false})});
    DataFrame j2 = DataFrame::from_columns({Column::from_arrays({Array::from_vec<int32_t>({1, 2, 3, 4, 4, 4, 5, 6, 7})}, Field{"d", DataType::Int32, false}),
                                            Column::from_arrays({Array::from_vec<double>({1.5, 2.5, 3.5, 4.5, 4.5, 4.5, 5.5, 6.5, 7.5})}, Field{"x", DataType::Float64, false})});
    LazyFrame frame = LazyFrame::read(j1).with_column_renamed("a", "key").with_column("sin_x", P::Function::Scalar_(P::ScalarFunction::Sine), {"x"});
    LazyFrame joined = frame.join(LazyFrame::read(j2), {DataFrame::JoinType::InnerJoin, {{"key", "d"}}});
    CHECK_EQ(joined.output().name, std::string("joined_dataframe"));
    DataFrame r = joined.with_column("sum_x", P::Function::Scalar_(P::ScalarFunction::Add), {"a.x", "b.x"}).evaluate();
    CHECK_EQ(r.num_rows(), 4);
    CHECK_EQ(r.num_columns(), 6u);   // key, a.x, sin_x, d, b.x, sum_x
    CHECK_EQ(host<int32_t>(r.column_by_name("key").data().chunk(0)), (std::vector<int32_t>{2, 3, 6, 6}));
    CHECK_EQ(host<double>(r.column_by_name("a.x").data().chunk(0)), (std::vector<double>{2, 3, 6, 60}));
    CHECK_EQ(host<double>(r.column_by_name("sum_x").data().chunk(0)), (std::vector<double>{4.5, 6.5, 12.5, 66.5}));
    CHECK_NEAR(host<double>(r.column_by_name("sin_x").data().chunk(0))[3], std::sin(60.0), 1e-15);
    CHECK_THROWS(frame.join(LazyFrame::read(j2), {DataFrame::JoinType::InnerJoin, {{"nope", "d"}}}));   // not in table A
    CHECK_THROWS(frame.join(LazyFrame::read(j2), {DataFrame::JoinType::InnerJoin, {{"key", "x"}}}));    // incompatible types
}

// ---------------------------------------------------------------- filter: DataFrame::filter + the fused filter -> aggregate
TEST(test_filter_and_fused_aggregate) {
    DataFrame df = DataFrame::from_csv(g_csv).drop({"city"});
    auto cond = BooleanFilter::gt(BooleanFilter::column("lat"), BooleanFilter::scalar(Scalar(55.0)));
    DataFrame f = df.filter(cond);
    CHECK_EQ(f.num_rows(), 5);
    CHECK_EQ(f.num_columns(), 2u);
    CHECK_NEAR(*AggregateFunctions::sum<double>(f.column_by_name("lat").data()), 282.746235, 1e-12);
    // same through the lazy, fused path: filter -> {sum, count, min, max, avg}(lat) in one pass
    using AF = P::AggregateFunction;
    DataFrame agg = LazyFrame::read(df).filter(cond).aggregate({}, {{AF::Sum, {"lat"}}, {AF::Count, {"lat"}}, {AF::Min, {"lat"}}, {AF::Max, {"lat"}}, {AF::Avg, {"lat"}}}).evaluate();
    CHECK_EQ(agg.num_rows(), 1);
    CHECK_EQ(agg.schema().fields[0].name, std::string("sum(lat)"));
    CHECK_EQ(agg.schema().fields[1].name, std::string("count(lat)"));
    CHECK(agg.schema().fields[1].data_type == DataType::UInt32);
    CHECK_NEAR(agg.column(0).data().chunk(0)->value<double>(0), 282.746235, 1e-12);
    CHECK_EQ(agg.column(1).data().chunk(0)->value<uint32_t>(0), 5u);
    CHECK_EQ(agg.column(3).data().chunk(0)->value<double>(0), 57.653484);
    CHECK_NEAR(agg.column(4).data().chunk(0)->value<double>(0), 282.746235 / 5, 1e-12);
    // unknown column -> ComputeError("Cannot find column ..") like expression.rs:812-815
    CHECK_THROWS(df.filter(BooleanFilter::gt(BooleanFilter::column("nope"), BooleanFilter::scalar(Scalar(1.0)))));
    // GroupAggregate with grouping columns: "aggregations not supported" (evaluation.rs:73)
    CHECK_THROWS(LazyFrame::read(df).aggregate({"lat"}, {{AF::Sum, {"lng"}}}).evaluate());
}

TEST(test_evaluate_type_rules) {
    auto k = Array::from_vec<int64_t>({1, 2, 3, 4});
    auto s = Array::from_vec<int8_t>({1, 2, 3, 4});
    DataFrame df = DataFrame::from_columns({Column::from_arrays({k}, Field{"k", DataType::Int64, true}), Column::from_arrays({s}, Field{"s", DataType::Int8, true})});
    // sin(Int64) inserts Cast -> Float64 (SinOperation)
    DataFrame r = LazyFrame::read(df).with_column("sk", P::Function::Scalar_(P::ScalarFunction::Sine), {"k"}).evaluate();
    CHECK(r.column_by_name("sk").data_type() == DataType::Float64);
    CHECK_NEAR(r.column_by_name("sk").data().chunk(0)->value<double>(2), std::sin(3.0), 1e-12);
    // Int8 arithmetic: panic!("Unsupported operation") in the reference (evaluation.rs:239) -> an error value here
    CHECK_THROWS(LazyFrame::read(df).with_column("ss", P::Function::Scalar_(P::ScalarFunction::Add), {"s", "s"}).evaluate());
    // add(Int64, Int8): the builder casts the right side to Int64 first
    DataFrame m = LazyFrame::read(df).with_column("ks", P::Function::Scalar_(P::ScalarFunction::Add), {"k", "s"}).evaluate();
    CHECK_EQ(host<int64_t>(m.column_by_name("ks").data().chunk(0)), (std::vector<int64_t>{2, 4, 6, 8}));
}

// ---------------------------------------------------------------- the fused batch loop vs the unfused oracle
struct HostCol { std::vector<double> v; std::vector<bool> valid; std::vector<uint8_t> bits; };

TEST(test_fused_pipeline_matches_unfused_oracle) {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const std::vector<size_t> lens{1024, 1024, 300};
    std::vector<std::vector<HostCol>> cols(2);  // a, b per chunk
    std::vector<Column> dev;
    for (int c = 0; c < 2; ++c) {
        std::vector<ArrayRef> chunks;
        for (size_t n : lens) {
            HostCol h;
            for (size_t i = 0; i < n; ++i) { h.v.push_back(U(rng)); h.valid.push_back(U(rng) > -0.8); }
            h.bits = pack_bits(h.valid);
            chunks.push_back(Array::from_vec(h.v, &h.valid));
            cols[c].push_back(std::move(h));
        }
        dev.push_back(Column::from_arrays(chunks, Field{c == 0 ? "a" : "b", DataType::Float64, true}));
    }
    DataFrame df = DataFrame::from_columns(dev);
    using AF = P::AggregateFunction;
    LazyFrame lf = LazyFrame::read(df)
                       .with_column("s", P::Function::Scalar_(P::ScalarFunction::Add), {"a", "b"})
                       .with_column("t", P::Function::Scalar_(P::ScalarFunction::Sine), {"s"})
                       .filter(BooleanFilter::gt(BooleanFilter::column("s"), BooleanFilter::scalar(Scalar(0.2))));
    DataFrame agg = lf.aggregate({}, {{AF::Sum, {"t"}}, {AF::Min, {"t"}}, {AF::Max, {"t"}}, {AF::Count, {"t"}}, {AF::Sum, {"s"}}}).evaluate();
    DataFrame mat = lf.evaluate();  // materialised: a, b, s, t filtered

    // oracle: sin(a + b) where a + b > 0.2, unfused
    rdf_expr_node nodes[6];
    std::memset(nodes, 0, sizeof nodes);
    nodes[0].kind = RDF_NODE_COLUMN; nodes[0].column = 0; nodes[0].lhs = nodes[0].rhs = -1;
    nodes[1].kind = RDF_NODE_COLUMN; nodes[1].column = 1; nodes[1].lhs = nodes[1].rhs = -1;
    nodes[2].kind = RDF_NODE_OP; nodes[2].op = RDF_OP_ADD; nodes[2].lhs = 0; nodes[2].rhs = 1;
    nodes[3].kind = RDF_NODE_OP; nodes[3].op = RDF_OP_SIN; nodes[3].lhs = 2; nodes[3].rhs = -1;
    nodes[4].kind = RDF_NODE_SCALAR; nodes[4].dtype = RDF_F64; nodes[4].f64 = 0.2; nodes[4].lhs = nodes[4].rhs = -1;
    nodes[5].kind = RDF_NODE_OP; nodes[5].op = RDF_OP_GT; nodes[5].lhs = 2; nodes[5].rhs = 4;
    std::vector<rdf_array> hv;
    for (int c = 0; c < 2; ++c)
        for (size_t i = 0; i < lens.size(); ++i) {
            rdf_array a; a.values = cols[c][i].v.data(); a.validity = cols[c][i].bits.data(); a.offset = 0; a.length = (int64_t)lens[i];
            a.null_count = -1; a.dtype = RDF_F64; a.mem = RDF_MEM_HOST;
            hv.push_back(a);
        }
    rdf_program prog;
    std::memset(&prog, 0, sizeof prog);
    prog.nodes = nodes; prog.nnodes = 6; prog.filter_root = 5; prog.nvalues = 2; prog.value_roots[0] = 3; prog.value_roots[1] = 2; prog.sink = RDF_SINK_AGG;
    rdf_agg_result exp[RDF_MAX_VALUES];
    CHECK_EQ(ora_pipeline(&prog, hv.data(), 2, (int64_t)lens.size(), nullptr, exp), RDF_OK);
    CHECK_NEAR(agg.column(0).data().chunk(0)->value<double>(0), exp[0].sum_f64, 1e-9);
    CHECK_NEAR(agg.column(1).data().chunk(0)->value<double>(0), exp[0].min_f64, 1e-9);
    CHECK_NEAR(agg.column(2).data().chunk(0)->value<double>(0), exp[0].max_f64, 1e-9);
    CHECK_EQ((int64_t)agg.column(3).data().chunk(0)->value<uint32_t>(0), exp[0].count);
    CHECK_NEAR(agg.column(4).data().chunk(0)->value<double>(0), exp[1].sum_f64, 1e-9);
    // the materialised frame agrees with the aggregates and keeps the chunking
    CHECK_EQ(mat.num_columns(), 4u);
    CHECK_EQ(mat.num_chunks(), lens.size());
    CHECK_EQ(mat.num_rows() - mat.column_by_name("t").null_count(), exp[0].count);
    CHECK_NEAR(*AggregateFunctions::sum<double>(mat.column_by_name("t").data()), exp[0].sum_f64, 1e-9);
    // eager Evaluate::calculate of one step == the lazy result for that column
    P::Calculation add = P::AddOperation::transform({P::Column{"a", DataType::Float64}, P::Column{"b", DataType::Float64}}, std::string("s"), std::nullopt)[0];
    DataFrame eager = Evaluate::calculate(df, add);
    CHECK_EQ(eager.num_columns(), 3u);
    auto e0 = host<double>(eager.column_by_name("s").data().chunk(0));
    for (size_t i = 0; i < 50; ++i)
        if (cols[0][0].valid[i] && cols[1][0].valid[i]) CHECK_EQ(e0[i], cols[0][0].v[i] + cols[1][0].v[i]);
}

// DataFrame::join with two criteria (JoinCriteria.criteria is a Vec of column pairs, src/expression.rs:332-337)
TEST(test_join_two_key_columns) {
    auto L = DataFrame::from_columns({Column::from_arrays({Array::from_vec(std::vector<int32_t>{1, 1, 2, 2, 3})}, Field{"a", DataType::Int32, true}),
                                      Column::from_arrays({Array::from_vec(std::vector<int64_t>{10, 20, 10, 20, 10})}, Field{"b", DataType::Int64, true}),
                                      Column::from_arrays({Array::from_vec(std::vector<double>{0.1, 0.2, 0.3, 0.4, 0.5})}, Field{"x", DataType::Float64, true})});
    auto R = DataFrame::from_columns({Column::from_arrays({Array::from_vec(std::vector<int32_t>{2, 1, 2, 4})}, Field{"c", DataType::Int32, true}),
                                      Column::from_arrays({Array::from_vec(std::vector<int64_t>{20, 10, 20, 10})}, Field{"d", DataType::Int64, true}),
                                      Column::from_arrays({Array::from_vec(std::vector<double>{7.0, 8.0, 9.0, 6.0})}, Field{"y", DataType::Float64, true})});
    DataFrame::JoinCriteria jc{DataFrame::JoinType::InnerJoin, {{"a", "c"}, {"b", "d"}}};
    DataFrame j = L.join(R, jc);
    CHECK_EQ(j.num_rows(), (int64_t)3);   // (1,10)-(1,10), (2,20)-(2,20) twice
    auto x = host<double>(j.column_by_name("x").data().chunk(0));
    auto y = host<double>(j.column_by_name("y").data().chunk(0));
    double sx = 0, sy = 0;
    for (double v : x) sx += v;
    for (double v : y) sy += v;
    CHECK_NEAR(sx, 0.1 + 0.4 + 0.4, 1e-12);
    CHECK_NEAR(sy, 8.0 + 7.0 + 9.0, 1e-12);
    DataFrame::JoinCriteria lj{DataFrame::JoinType::LeftJoin, {{"a", "c"}, {"b", "d"}}};
    CHECK_EQ(L.join(R, lj).num_rows(), (int64_t)6);   // 3 matches + 3 unmatched left rows
}

// GroupAggregate with a grouping column: the reference has only the schema (Dataset::try_aggregate) and panics on
// execution (src/evaluation.rs:73); expectations are SQL semantics computed on the host.
// `stride` spreads the 37 keys: 1 = a small dense domain (one fused rdf_group_pipeline pass for all aggregations),
// 100003 = a sparse one (rdf_groupby_sum per aggregation); both must give the same frame
static void group_aggregate_by_key(int32_t stride) {
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const std::vector<size_t> lens{1024, 500, 2000};
    std::vector<ArrayRef> kch, vch, wch;
    std::map<int32_t, double> sum_v; std::map<int32_t, int64_t> cnt_v, sum_w; double null_sum = 0; int64_t null_cnt = 0, null_w = 0; bool has_null = false;
    for (size_t n : lens) {
        std::vector<int32_t> k(n); std::vector<double> v(n); std::vector<int64_t> w(n);
        std::vector<bool> kvalid(n), vvalid(n);
        for (size_t i = 0; i < n; ++i) {
            k[i] = ((int32_t)(rng() % 37) - 5) * stride; v[i] = U(rng); w[i] = (int64_t)(rng() % 1000) - 500;
            kvalid[i] = U(rng) > 0.02; vvalid[i] = U(rng) > 0.1;
        }
        kch.push_back(Array::from_vec(k, &kvalid)); vch.push_back(Array::from_vec(v, &vvalid)); wch.push_back(Array::from_vec(w));
        for (size_t i = 0; i < n; ++i) {
            if (!(v[i] > 0.25) || !vvalid[i]) continue;       // rows dropped by the filter v > 0.25 (NULL -> dropped)
            if (!kvalid[i]) { has_null = true; null_sum += v[i]; ++null_cnt; null_w += w[i]; continue; }
            sum_v[k[i]] += v[i]; ++cnt_v[k[i]]; sum_w[k[i]] += w[i];
        }
    }
    DataFrame df = DataFrame::from_columns({Column::from_arrays(kch, Field{"k", DataType::Int32, true}),
                                            Column::from_arrays(vch, Field{"v", DataType::Float64, true}),
                                            Column::from_arrays(wch, Field{"w", DataType::Int64, false})});
    using AF = P::AggregateFunction;
    DataFrame g = LazyFrame::read(df)
                      .filter(BooleanFilter::gt(BooleanFilter::column("v"), BooleanFilter::scalar(Scalar(0.25))))
                      .aggregate({"k"}, {{AF::Sum, {"v", "w"}}, {AF::Count, {"v"}}, {AF::Avg, {"v"}}})
                      .evaluate();
    CHECK_EQ(g.num_columns(), 5u);
    CHECK_EQ(g.schema().fields[0].name, std::string("k"));
    CHECK_EQ(g.schema().fields[1].name, std::string("sum(v)"));
    CHECK_EQ(g.schema().fields[2].name, std::string("sum(w)"));
    CHECK_EQ(g.schema().fields[3].name, std::string("count(v)"));
    CHECK_EQ(g.schema().fields[4].name, std::string("avg(v)"));
    CHECK_EQ((size_t)g.num_rows(), sum_v.size() + (has_null ? 1 : 0));
    auto gk = host<int32_t>(g.column(0).data().chunk(0));
    auto gs = host<double>(g.column(1).data().chunk(0));
    auto gw = host<int64_t>(g.column(2).data().chunk(0));
    auto gc = host<uint32_t>(g.column(3).data().chunk(0));
    auto ga = host<double>(g.column(4).data().chunk(0));
    size_t r = 0;
    for (auto& kv : sum_v) {   // std::map iterates in key order == the sorted result
        CHECK_EQ(gk[r], kv.first);
        CHECK_NEAR(gs[r], kv.second, 1e-9);
        CHECK_EQ(gw[r], sum_w[kv.first]);
        CHECK_EQ((int64_t)gc[r], cnt_v[kv.first]);
        CHECK_NEAR(ga[r], kv.second / (double)cnt_v[kv.first], 1e-9);
        ++r;
    }
    if (has_null) {            // the NULL group sorts last
        CHECK(g.column(0).data().chunk(0)->is_null((int64_t)r));
        CHECK_NEAR(gs[r], null_sum, 1e-9);
        CHECK_EQ(gw[r], null_w);
        CHECK_EQ((int64_t)gc[r], null_cnt);
    }
    // unsupported shapes are errors, not silent fallbacks
    CHECK_THROWS(LazyFrame::read(df).aggregate({"v"}, {{AF::Sum, {"w"}}}).evaluate());   // a Float64 grouping column
}
// TPC-H Q1's shape through the lazy API (BASELINE.json config C5): filter -> two computed columns -> GROUP BY two
// dictionary-coded columns with sums / average / count.  Everything runs as ONE fused pass (rdf_group_pipeline).
TEST(test_group_aggregate_q1_shape_two_keys) {
    std::mt19937_64 rng(23);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const std::vector<size_t> lens{1024, 1024, 700};
    std::vector<ArrayRef> fch, sch, qch, pch, dch, och, hch;
    struct G { double qty = 0, price = 0, dp = 0; int64_t n = 0; };
    std::map<std::pair<int, int>, G> exp;
    for (size_t n : lens) {
        std::vector<int8_t> fl(n), st(n); std::vector<double> q(n), pr(n), di(n), one(n, 1.0); std::vector<int32_t> sh(n);
        std::vector<bool> qvalid(n);
        for (size_t i = 0; i < n; ++i) {
            fl[i] = (int8_t)(rng() % 3); st[i] = (int8_t)(rng() % 2);
            q[i] = 1.0 + (double)(rng() % 50); pr[i] = 900.0 + 1000.0 * U(rng); di[i] = (double)(rng() % 11) / 100.0;
            sh[i] = 8036 + (int32_t)(rng() % 2526); qvalid[i] = U(rng) > 0.05;
            if (sh[i] > 10471) continue;
            G& g = exp[{fl[i], st[i]}];
            g.price += pr[i]; g.dp += pr[i] * (1.0 - di[i]);
            if (qvalid[i]) { g.qty += q[i]; ++g.n; }
        }
        fch.push_back(Array::from_vec(fl)); sch.push_back(Array::from_vec(st)); qch.push_back(Array::from_vec(q, &qvalid)); pch.push_back(Array::from_vec(pr));
        dch.push_back(Array::from_vec(di)); och.push_back(Array::from_vec(one)); hch.push_back(Array::from_vec(sh));
    }
    DataFrame df = DataFrame::from_columns({Column::from_arrays(fch, Field{"returnflag", DataType::Int8, false}), Column::from_arrays(sch, Field{"linestatus", DataType::Int8, false}),
                                            Column::from_arrays(qch, Field{"quantity", DataType::Float64, true}), Column::from_arrays(pch, Field{"extendedprice", DataType::Float64, false}),
                                            Column::from_arrays(dch, Field{"discount", DataType::Float64, false}), Column::from_arrays(och, Field{"one", DataType::Float64, false}),
                                            Column::from_arrays(hch, Field{"shipdate", DataType::Int32, false})});
    using AF = P::AggregateFunction;
    DataFrame g = LazyFrame::read(df)
                      .filter(BooleanFilter::le(BooleanFilter::column("shipdate"), BooleanFilter::scalar(Scalar((int32_t)10471))))
                      .with_column("one_minus_discount", P::Function::Scalar_(P::ScalarFunction::Subtract), {"one", "discount"})
                      .with_column("disc_price", P::Function::Scalar_(P::ScalarFunction::Multiply), {"extendedprice", "one_minus_discount"})
                      .aggregate({"returnflag", "linestatus"}, {{AF::Sum, {"quantity", "extendedprice", "disc_price"}}, {AF::Avg, {"quantity"}}, {AF::Count, {"quantity"}}})
                      .evaluate();
    CHECK_EQ(g.num_columns(), 7u);
    CHECK_EQ(g.schema().fields[0].name, std::string("returnflag"));
    CHECK_EQ(g.schema().fields[1].name, std::string("linestatus"));
    CHECK_EQ(g.schema().fields[4].name, std::string("sum(disc_price)"));
    CHECK_EQ(g.schema().fields[5].name, std::string("avg(quantity)"));
    CHECK_EQ(g.schema().fields[6].name, std::string("count(quantity)"));
    CHECK_EQ((size_t)g.num_rows(), exp.size());
    auto kf = host<int8_t>(g.column(0).data().chunk(0)), ks = host<int8_t>(g.column(1).data().chunk(0));
    auto sq = host<double>(g.column(2).data().chunk(0)), sp = host<double>(g.column(3).data().chunk(0)), sd = host<double>(g.column(4).data().chunk(0));
    auto aq = host<double>(g.column(5).data().chunk(0));
    auto cq = host<uint32_t>(g.column(6).data().chunk(0));
    size_t r = 0;
    for (auto& kv : exp) {   // std::map<pair> iterates in (returnflag, linestatus) order == the result's row order
        CHECK_EQ((int)kf[r], kv.first.first);
        CHECK_EQ((int)ks[r], kv.first.second);
        CHECK_NEAR(sq[r], kv.second.qty, 1e-9 * kv.second.qty);
        CHECK_NEAR(sp[r], kv.second.price, 1e-9 * kv.second.price);
        CHECK_NEAR(sd[r], kv.second.dp, 1e-9 * kv.second.dp);
        CHECK_NEAR(aq[r], kv.second.qty / (double)kv.second.n, 1e-9);
        CHECK_EQ((int64_t)cq[r], kv.second.n);
        ++r;
    }
}
TEST(test_group_aggregate_by_key) { group_aggregate_by_key(1); }
TEST(test_group_aggregate_by_sparse_key) { group_aggregate_by_key(100003); }

// ---------------------------------------------------------------- typed CSV (src/dataframe.rs:349-389: arrow's csv reader infers the schema)
static std::string write_temp_csv(const std::string& body) {
    char name[] = "/tmp/rdf_csv_XXXXXX";
    const int fd = mkstemp(name);
    CHECK(fd >= 0);
    FILE* f = fdopen(fd, "w");
    fputs(body.c_str(), f);
    fclose(f);
    return name;
}
TEST(test_read_csv_infers_types_and_nulls) {
    std::string body = "id,score,flag,name,qty\n";
    const size_t n = 2500;   // three batches of 1024 rows
    int64_t want_qty = 0, qty_nulls = 0;
    for (size_t i = 0; i < n; ++i) {
        body += std::to_string((int64_t)i - 7) + "," + std::to_string(0.5 * (double)i) + "," + (i % 3 ? "true" : "false") + ",\"row, " + std::to_string(i) + "\",";
        if (i % 10 == 3) { ++qty_nulls; } else { body += std::to_string(i * 3); want_qty += (int64_t)i * 3; }
        body += "\n";
    }
    const std::string path = write_temp_csv(body);
    DataFrame df = DataFrame::from_csv(path);
    std::remove(path.c_str());
    CHECK_EQ(df.num_columns(), 5u);
    CHECK_EQ(df.num_rows(), (int64_t)n);
    CHECK_EQ(df.num_chunks(), 3u);
    CHECK(df.column_by_name("id").data_type() == DataType::Int64);
    CHECK(df.column_by_name("score").data_type() == DataType::Float64);
    CHECK(df.column_by_name("flag").data_type() == DataType::Boolean);
    CHECK(df.column_by_name("name").data_type() == DataType::Utf8);
    CHECK(df.column_by_name("qty").data_type() == DataType::Int64);
    CHECK_EQ(df.column_by_name("qty").null_count(), qty_nulls);
    CHECK_EQ(df.column_by_name("id").null_count(), (int64_t)0);
    CHECK_EQ(host<int64_t>(df.column_by_name("id").data().chunk(1))[0], (int64_t)1024 - 7);
    CHECK_EQ((*df.column_by_name("name").data().chunk(2)->strings)[0], std::string("row, 2048"));
    CHECK(df.column_by_name("qty").data().chunk(0)->is_null(3));
    CHECK(!df.column_by_name("qty").data().chunk(0)->is_null(4));
    CHECK_EQ(*AggregateFunctions::sum<int64_t>(df.column_by_name("qty").data()), want_qty);
    CHECK_EQ(*AggregateFunctions::count(df.column_by_name("qty").data()), (int64_t)n - qty_nulls);
    // the typed columns are on the compute path: a filter on the Boolean column, then a sum of an Int64 one
    DataFrame kept = df.filter_by_mask(df.column_by_name("flag"));
    int64_t want_kept = 0, want_id = 0;
    for (size_t i = 0; i < n; ++i) if (i % 3) { ++want_kept; want_id += (int64_t)i - 7; }
    CHECK_EQ(kept.num_rows(), want_kept);
    CHECK_EQ(*AggregateFunctions::sum<int64_t>(kept.column_by_name("id").data()), want_id);
    CHECK_EQ(kept.column_by_name("name").num_rows(), want_kept);
    // select("*") keeps every column (:272)
    CHECK_EQ(df.select({"*"}).num_columns(), 5u);
    CHECK_EQ(df.select({"id", "nope"}).num_columns(), 1u);
}

// A long file: the types come from the first 16 384 records and every cell is parsed once against them; what only a late record
// shows — a fraction in a column of integers, text in a column of booleans, the first value of a column that was empty so far —
// must give the types inference over the WHOLE file gives, and the values with them.
TEST(test_read_long_csv_types_hold_for_late_records) {
    const size_t n = 70000;
    for (int variant = 0; variant < 2; ++variant) {     // 0: every column keeps its sampled type; 1: late records change three of them
        std::string body = "a,b,c,d\n";
        int64_t want_b = 0;
        double want_a = 0;
        for (size_t i = 0; i < n; ++i) {
            const bool late = variant == 1 && i == n - 10;
            body += late ? "2.5" : std::to_string((int64_t)i - 100);
            want_a += late ? 2.5 : (double)((int64_t)i - 100);
            body += "," + std::to_string((int64_t)i * 2) + ",";
            want_b += (int64_t)i * 2;
            if (variant == 1 && i >= 20000) body += std::to_string((int64_t)i % 7);      // column c: empty in the sampled records
            body += std::string(",") + (variant == 1 && i == n - 5 ? "maybe" : (i % 2 ? "true" : "false")) + "\n";
        }
        const std::string path = write_temp_csv(body);
        DataFrame df = DataFrame::from_csv(path);
        std::remove(path.c_str());
        CHECK_EQ(df.num_rows(), (int64_t)n);
        CHECK(df.column_by_name("b").data_type() == DataType::Int64);
        CHECK_EQ(*AggregateFunctions::sum<int64_t>(df.column_by_name("b").data()), want_b);
        if (variant == 0) {
            CHECK(df.column_by_name("a").data_type() == DataType::Int64);
            CHECK(df.column_by_name("c").data_type() == DataType::Utf8);           // no value anywhere
            CHECK(df.column_by_name("d").data_type() == DataType::Boolean);
            CHECK_EQ(*AggregateFunctions::sum<int64_t>(df.column_by_name("a").data()), (int64_t)want_a);
        } else {
            CHECK(df.column_by_name("a").data_type() == DataType::Float64);
            CHECK_NEAR(*AggregateFunctions::sum<double>(df.column_by_name("a").data()), want_a, 1e-12);
            CHECK(df.column_by_name("c").data_type() == DataType::Int64);
            CHECK_EQ(df.column_by_name("c").null_count(), (int64_t)20000);
            CHECK(df.column_by_name("d").data_type() == DataType::Utf8);
            CHECK_EQ((*df.column_by_name("d").data().chunk((n - 5) / 1024)->strings)[(n - 5) % 1024], std::string("maybe"));
        }
    }
}

// sort / join / take of a frame holding text and Boolean columns (Column::take, src/table.rs:218-241, is type-generic)
TEST(test_sort_and_join_carry_text_and_boolean_columns) {
    DataFrame df = DataFrame::from_csv(g_csv);   // city (Utf8), lat, lng
    std::vector<std::string> city;
    std::vector<double> lat;
    for (auto& c : df.column_by_name("city").data().chunks()) city.insert(city.end(), c->strings->begin(), c->strings->end());
    for (auto& c : df.column_by_name("lat").data().chunks()) { auto v = host<double>(c); lat.insert(lat.end(), v.begin(), v.end()); }
    std::vector<size_t> order(lat.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lat[a] < lat[b]; });
    DataFrame sorted = df.sort({{"lat", false, false}});
    CHECK_EQ(sorted.num_columns(), 3u);
    CHECK_EQ(sorted.num_rows(), 37);
    const auto& sc = *sorted.column_by_name("city").data().chunk(0)->strings;
    auto sl = host<double>(sorted.column_by_name("lat").data().chunk(0));
    for (size_t i = 0; i < order.size(); ++i) { CHECK_EQ(sc[i], city[order[i]]); CHECK_EQ(sl[i], lat[order[i]]); }
    // a Boolean column and a NULL index
    std::vector<bool> north(lat.size());
    for (size_t i = 0; i < lat.size(); ++i) north[i] = lat[i] > 53.0;
    DataFrame with_flag = df.with_column("north", Column::from_arrays({Array::from_bools(north)}, Field{"north", DataType::Boolean, false}));
    const std::vector<bool> iv{1, 0, 1, 1};
    DataFrame t = with_flag.take(Array::from_vec<uint32_t>({36, 0, 5, 5}, &iv));
    CHECK_EQ(t.num_rows(), 4);
    const auto tn = t.column_by_name("north").data().chunk(0);
    CHECK(tn->dtype == DataType::Boolean);
    CHECK_EQ(tn->bools_to_host()[0], (bool)north[36]);
    CHECK_EQ(tn->bools_to_host()[2], (bool)north[5]);
    CHECK(tn->is_null(1));
    const auto tc = t.column_by_name("city").data().chunk(0);
    CHECK_EQ((*tc->strings)[0], city[36]);
    CHECK_EQ((*tc->strings)[3], city[5]);
    CHECK(tc->is_null(1) && !tc->is_null(0));
    // join of the frame with a numeric one on an Int64 id: the text column rides through the gather
    DataFrame left = df.with_id("id");
    DataFrame right = DataFrame::from_columns({Column::from_arrays({Array::from_vec<uint64_t>({3, 1, 37, 99})}, Field{"rid", DataType::UInt64, false}),
                                               Column::from_arrays({Array::from_vec<double>({30.0, 10.0, 370.0, 990.0})}, Field{"w", DataType::Float64, false})});
    DataFrame j = left.join(right, {DataFrame::JoinType::InnerJoin, {{"id", "rid"}}});
    CHECK_EQ(j.num_rows(), 3);
    auto jid = host<uint64_t>(j.column_by_name("id").data().chunk(0));
    const auto& jc = *j.column_by_name("city").data().chunk(0)->strings;
    auto jw = host<double>(j.column_by_name("w").data().chunk(0));
    for (size_t r = 0; r < 3; ++r) { CHECK_EQ(jc[r], city[(size_t)jid[r] - 1]); CHECK_EQ(jw[r], 10.0 * (double)jid[r]); }
}

// GroupAggregate with min / max and with several sparse grouping columns (what the reference's plan can express,
// src/lazyframe.rs:200-258, evaluated on the device; the reference's evaluator stops at "aggregations not supported")
TEST(test_group_aggregate_min_max_and_sparse_key_columns) {
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(-100.0, 100.0);
    const size_t nchunks = 4, n = 6000;
    struct G { double sum = 0, mn = INFINITY, mx = -INFINITY; int64_t imn = INT64_MAX, imx = INT64_MIN; int64_t cnt = 0; };
    std::map<std::tuple<int64_t, int32_t, int16_t>, G> exp;
    std::vector<ArrayRef> ach, bch, cch, xch, ych;
    for (size_t c = 0; c < nchunks; ++c) {
        std::vector<int64_t> a(n), y(n);
        std::vector<int32_t> b(n);
        std::vector<int16_t> k3(n);
        std::vector<double> x(n);
        std::vector<bool> xv(n);
        for (size_t i = 0; i < n; ++i) {
            a[i] = (int64_t)(rng() % 7) * 1000000007LL - 3000000000LL;   // 7 values spread over 6e9
            b[i] = (int32_t)(rng() % 5) * 400000 - 800000;               // 5 values spread over 1.6e6
            k3[i] = (int16_t)((int)(rng() % 3) * 9000 - 9000);           // 3 values spread over 18000
            x[i] = U(rng); xv[i] = rng() % 17 != 0; y[i] = (int64_t)(rng() % 2000001) - 1000000;
            G& g = exp[{a[i], b[i], k3[i]}];
            if (xv[i]) { g.sum += x[i]; g.mn = std::min(g.mn, x[i]); g.mx = std::max(g.mx, x[i]); ++g.cnt; }
            g.imn = std::min(g.imn, y[i]); g.imx = std::max(g.imx, y[i]);
        }
        ach.push_back(Array::from_vec(a)); bch.push_back(Array::from_vec(b)); cch.push_back(Array::from_vec(k3));
        xch.push_back(Array::from_vec(x, &xv)); ych.push_back(Array::from_vec(y));
    }
    DataFrame df = DataFrame::from_columns({Column::from_arrays(ach, Field{"a", DataType::Int64, false}), Column::from_arrays(bch, Field{"b", DataType::Int32, false}),
                                            Column::from_arrays(cch, Field{"c", DataType::Int16, false}), Column::from_arrays(xch, Field{"x", DataType::Float64, true}),
                                            Column::from_arrays(ych, Field{"y", DataType::Int64, false})});
    using AF = P::AggregateFunction;
    DataFrame g = LazyFrame::read(df).aggregate({"a", "b", "c"}, {{AF::Min, {"x", "y"}}, {AF::Max, {"x", "y"}}, {AF::Sum, {"x"}}, {AF::Count, {"x"}}}).evaluate();
    CHECK_EQ(g.num_columns(), 9u);
    CHECK_EQ((size_t)g.num_rows(), exp.size());
    CHECK_EQ(g.schema().fields[3].name, std::string("min(x)"));
    CHECK_EQ(g.schema().fields[4].name, std::string("min(y)"));
    CHECK_EQ(g.schema().fields[5].name, std::string("max(x)"));
    CHECK_EQ(g.schema().fields[6].name, std::string("max(y)"));
    CHECK(g.schema().fields[4].data_type == DataType::Int64);   // min / max keep the input type
    CHECK(g.schema().fields[3].data_type == DataType::Float64);
    auto ka = host<int64_t>(g.column(0).data().chunk(0));
    auto kb = host<int32_t>(g.column(1).data().chunk(0));
    auto kc = host<int16_t>(g.column(2).data().chunk(0));
    auto mnx = host<double>(g.column(3).data().chunk(0)), mxx = host<double>(g.column(5).data().chunk(0)), sx = host<double>(g.column(7).data().chunk(0));
    auto mny = host<int64_t>(g.column(4).data().chunk(0)), mxy = host<int64_t>(g.column(6).data().chunk(0));
    auto cx = host<uint32_t>(g.column(8).data().chunk(0));
    size_t r = 0;
    for (auto& kv : exp) {   // std::map<tuple> iterates in (a, b, c) order == the result's row order
        CHECK_EQ(ka[r], std::get<0>(kv.first)); CHECK_EQ(kb[r], std::get<1>(kv.first)); CHECK_EQ(kc[r], std::get<2>(kv.first));
        CHECK_EQ(mnx[r], kv.second.mn); CHECK_EQ(mxx[r], kv.second.mx);
        CHECK_EQ(mny[r], kv.second.imn); CHECK_EQ(mxy[r], kv.second.imx);
        CHECK_NEAR(sx[r], kv.second.sum, 1e-9 * (1.0 + std::fabs(kv.second.sum)));
        CHECK_EQ((int64_t)cx[r], kv.second.cnt);
        ++r;
    }
}

// Per-group Min / Max over a SMALL DENSE key domain (keys 0..9): the dense one-pass kernel folds only sums and counts, so
// the step must take the hash path instead of failing (round-2 review: dense_groups threw "Aggregation not yet supported").
TEST(test_group_aggregate_min_max_dense_keys) {
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> U(-50.0, 50.0);
    const size_t nchunks = 3, n = 4000;
    struct G { double sum = 0, mn = INFINITY, mx = -INFINITY; int64_t cnt = 0; };
    std::map<int64_t, G> exp;
    std::vector<ArrayRef> kch, xch;
    for (size_t c = 0; c < nchunks; ++c) {
        std::vector<int64_t> k(n);
        std::vector<double> x(n);
        for (size_t i = 0; i < n; ++i) {
            k[i] = (int64_t)(rng() % 10); x[i] = U(rng);
            G& g = exp[k[i]];
            g.sum += x[i]; g.mn = std::min(g.mn, x[i]); g.mx = std::max(g.mx, x[i]); ++g.cnt;
        }
        kch.push_back(Array::from_vec(k)); xch.push_back(Array::from_vec(x));
    }
    DataFrame df = DataFrame::from_columns({Column::from_arrays(kch, Field{"k", DataType::Int64, false}), Column::from_arrays(xch, Field{"x", DataType::Float64, false})});
    using AF = P::AggregateFunction;
    DataFrame g = LazyFrame::read(df).aggregate({"k"}, {{AF::Min, {"x"}}, {AF::Max, {"x"}}, {AF::Sum, {"x"}}, {AF::Count, {"x"}}}).evaluate();
    CHECK_EQ(g.num_columns(), 5u);
    CHECK_EQ((size_t)g.num_rows(), exp.size());
    auto kk = host<int64_t>(g.column(0).data().chunk(0));
    auto mn = host<double>(g.column(1).data().chunk(0)), mx = host<double>(g.column(2).data().chunk(0)), sx = host<double>(g.column(3).data().chunk(0));
    auto cx = host<uint32_t>(g.column(4).data().chunk(0));
    size_t r = 0;
    for (auto& kv : exp) {
        CHECK_EQ(kk[r], kv.first); CHECK_EQ(mn[r], kv.second.mn); CHECK_EQ(mx[r], kv.second.mx);
        CHECK_NEAR(sx[r], kv.second.sum, 1e-9 * (1.0 + std::fabs(kv.second.sum)));
        CHECK_EQ((int64_t)cx[r], kv.second.cnt);
        ++r;
    }
    // and the sums-only form of the same frame still takes the dense kernel with the same answers
    DataFrame d = LazyFrame::read(df).aggregate({"k"}, {{AF::Sum, {"x"}}, {AF::Count, {"x"}}}).evaluate();
    auto ds = host<double>(d.column(1).data().chunk(0));
    r = 0;
    for (auto& kv : exp) { CHECK_NEAR(ds[r], kv.second.sum, 1e-9 * (1.0 + std::fabs(kv.second.sum))); ++r; }
}

// GpuFrame: the frame-producing operators as handle-in / handle-out calls give what the per-column / per-batch paths of the
// mirror give — DataFrame::filter (src/dataframe.rs:178-189), sort (:194-222), take, GroupAggregate — and chain on the device.
TEST(test_gpu_frame_operators_match_the_dataframe_paths) {
    std::mt19937_64 rng(2024);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    const std::vector<size_t> lens{1024, 1024, 1024, 300, 0, 77};
    std::vector<ArrayRef> xc, kc, yc;
    for (size_t n : lens) {
        std::vector<double> x(n), y(n);
        std::vector<int64_t> k(n);
        std::vector<bool> yv(n);
        for (size_t i = 0; i < n; ++i) { x[i] = U(rng); y[i] = U(rng); yv[i] = rng() % 9 != 0; k[i] = (int64_t)(rng() % 23) - 11; }
        xc.push_back(Array::from_vec(x)); kc.push_back(Array::from_vec(k)); yc.push_back(Array::from_vec(y, &yv));
    }
    DataFrame df = DataFrame::from_columns({Column::from_arrays(xc, Field{"x", DataType::Float64, false}), Column::from_arrays(kc, Field{"k", DataType::Int64, false}),
                                            Column::from_arrays(yc, Field{"y", DataType::Float64, true})});
    GpuFrame g = GpuFrame::pin(df);
    CHECK_EQ(g.num_rows(), df.num_rows());
    CHECK_EQ((size_t)g.num_chunks(), lens.size());
    auto same = [&](const DataFrame& a, const DataFrame& b) {
        CHECK_EQ(a.num_columns(), b.num_columns());
        CHECK_EQ(a.num_rows(), b.num_rows());
        CHECK_EQ(a.num_chunks(), b.num_chunks());
        for (size_t c = 0; c < a.num_columns() && c < b.num_columns(); ++c)
            for (size_t i = 0; i < a.num_chunks() && i < b.num_chunks(); ++i) {
                const ArrayRef p = a.column(c).data().chunk(i), q = b.column(c).data().chunk(i);
                CHECK_EQ(p->length, q->length);
                CHECK_EQ(p->null_count, q->null_count);
                if (a.column(c).data_type() == DataType::Int64) CHECK_EQ(host<int64_t>(p), host<int64_t>(q));
                else {
                    auto hv = host<double>(p), hw = host<double>(q);
                    auto vp = p->valid_to_host(), vq = q->valid_to_host();
                    CHECK_EQ(hv.size(), hw.size());
                    for (size_t r = 0; r < hv.size() && r < hw.size(); ++r) { CHECK_EQ(vp[r], vq[r]); if (vp[r]) CHECK_EQ(hv[r], hw[r]); }
                }
            }
    };
    const FilterRef cond = BooleanFilter::and_(BooleanFilter::gt(BooleanFilter::column("x"), BooleanFilter::scalar(Scalar(0.1))),
                                               BooleanFilter::lt(BooleanFilter::column("k"), BooleanFilter::scalar(Scalar((int64_t)7))));
    same(g.filter(cond).to_dataframe(), df.filter(cond));
    const std::vector<DataFrame::SortCriteria> crit{{"k", true, false}, {"x", false, false}};
    same(g.sort(crit).to_dataframe(), df.sort(crit));
    // a chain that never leaves the device, against the same chain through the per-batch paths
    same(g.filter(cond).sort(crit).to_dataframe(), df.filter(cond).sort(crit));
    // aggregates over the pinned and over a derived frame
    const rdf_agg_result r = g.filter(cond).aggregate("x");
    const Column fx = df.filter(cond).column_by_name("x");
    CHECK_EQ(r.count, fx.num_rows());
    CHECK_NEAR(r.sum_f64, AggregateFunctions::sum<double>(fx.data()).value_or(0.0), 1e-9);
    // GroupAggregate on the handle: sum(y) by k, NULL values skipped
    DataFrame ga = g.group_aggregate({"k"}, "y", P::AggregateFunction::Sum, 64).sort({{"k", false, false}}).to_dataframe();
    std::map<int64_t, std::pair<double, int64_t>> exp;
    for (size_t c = 0; c < lens.size(); ++c) {
        auto kk = host<int64_t>(kc[c]); auto yy = host<double>(yc[c]); auto vv = yc[c]->valid_to_host();
        for (size_t i = 0; i < kk.size(); ++i) { auto& e = exp[kk[i]]; if (vv[i]) { e.first += yy[i]; ++e.second; } }
    }
    CHECK_EQ((size_t)ga.num_rows(), exp.size());
    auto gk = host<int64_t>(ga.column(0).data().chunk(0)); auto gs = host<double>(ga.column(1).data().chunk(0)); auto gc = host<int64_t>(ga.column(2).data().chunk(0));
    size_t row = 0;
    for (auto& kv : exp) { CHECK_EQ(gk[row], kv.first); CHECK_NEAR(gs[row], kv.second.first, 1e-9); CHECK_EQ(gc[row], kv.second.second); ++row; }
    CHECK_THROWS(g.filter(BooleanFilter::gt(BooleanFilter::column("nope"), BooleanFilter::scalar(Scalar(0.0)))));
}

// ShardedFrame: a DataFrame's RecordBatches sharded over ranks by row ranges — what stands where the reference
// panics for GroupAggregate (src/evaluation.rs:73) when N > 1.  One thread per rank; the test box has one GPU, so the ranks share
// device 0 over the peer-copy transport (RDF_COMM_PEER), and a 1-rank RCCL communicator runs the same calls through librccl.
static void sharded_frame_case(rdf_comm_kind kind, int world) {
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const size_t total = 50000;
    std::vector<int64_t> k(total);
    std::vector<double> v(total);
    std::map<int64_t, std::pair<double, int64_t>> exp;
    double vsum = 0.0;
    for (size_t i = 0; i < total; ++i) { k[i] = (int64_t)(rng() % 1500) * 7919 - 3000000; v[i] = U(rng); auto& e = exp[k[i]]; e.first += v[i]; ++e.second; vsum += v[i]; }
    std::vector<int32_t> devices((size_t)world, 0);
    auto comms = Communicator::init_all(devices, kind);
    std::vector<std::map<int64_t, std::pair<double, int64_t>>> owned((size_t)world);
    std::vector<std::string> errors((size_t)world);
    std::vector<rdf_agg_result> totals((size_t)world);
    std::vector<int64_t> rows((size_t)world, 0), big((size_t)world, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            try {
                check(rdf_set_device(0));
                // contiguous row ranges of whole 1024-row batches, ragged on purpose
                const size_t lo = total * (size_t)r / (size_t)world / 1024 * 1024, hi = r + 1 == world ? total : total * (size_t)(r + 1) / (size_t)world / 1024 * 1024;
                std::vector<ArrayRef> kc, vc;
                for (size_t a = lo; a < hi; a += 1024) {
                    const size_t b = std::min(hi, a + 1024);
                    kc.push_back(Array::from_vec(std::vector<int64_t>(k.begin() + (long)a, k.begin() + (long)b)));
                    vc.push_back(Array::from_vec(std::vector<double>(v.begin() + (long)a, v.begin() + (long)b)));
                }
                DataFrame df = DataFrame::from_columns({Column::from_arrays(kc, Field{"k", DataType::Int64, false}), Column::from_arrays(vc, Field{"v", DataType::Float64, false})});
                ShardedFrame sf(GpuFrame::pin(df), comms[(size_t)r]);
                rows[(size_t)r] = sf.num_rows();
                totals[(size_t)r] = sf.aggregate("v");
                rdf_exchange_stats st;
                DataFrame g = sf.group_aggregate("k", "v", P::AggregateFunction::Sum, 2000, RDF_EXCHANGE_AUTO, &st).to_dataframe();
                auto gk = host<int64_t>(g.column(0).data().chunk(0)); auto gs = host<double>(g.column(1).data().chunk(0)); auto gc = host<int64_t>(g.column(2).data().chunk(0));
                for (size_t i = 0; i < gk.size(); ++i) owned[(size_t)r][gk[i]] = {gs[i], gc[i]};
                if (st.exchange != RDF_EXCHANGE_GROUPS) errors[(size_t)r] = "expected the partial-group exchange";
                // shard-local filter, then the aggregates of what is left over all ranks
                ShardedFrame f2 = sf.filter(BooleanFilter::gt(BooleanFilter::column("v"), BooleanFilter::scalar(Scalar(0.5))));
                big[(size_t)r] = f2.aggregate("v").count;
                sf.comm().barrier();
            } catch (const std::exception& e) { errors[(size_t)r] = e.what(); }
        });
    for (auto& t : th) t.join();
    for (int r = 0; r < world; ++r) { if (!errors[(size_t)r].empty()) std::printf("rank %d: %s\n", r, errors[(size_t)r].c_str()); CHECK(errors[(size_t)r].empty()); }
    std::map<int64_t, std::pair<double, int64_t>> got;
    for (auto& m : owned) for (auto& kv : m) { CHECK(got.find(kv.first) == got.end()); got[kv.first] = kv.second; }
    CHECK_EQ(got.size(), exp.size());
    for (auto& kv : exp) { CHECK(got.count(kv.first) == 1); CHECK_EQ(got[kv.first].second, kv.second.second); CHECK_NEAR(got[kv.first].first, kv.second.first, 1e-9); }
    int64_t above = 0;
    for (double x : v) above += x > 0.5;
    for (int r = 0; r < world; ++r) {
        CHECK_EQ(rows[(size_t)r], (int64_t)total);
        CHECK_EQ(totals[(size_t)r].count, (int64_t)total);
        CHECK_NEAR(totals[(size_t)r].sum_f64, vsum, 1e-9);
        CHECK_EQ(totals[(size_t)r].sum_f64, totals[0].sum_f64);     // rank-order fold: the same bits everywhere
        CHECK_EQ(big[(size_t)r], above);
    }
}
TEST(test_sharded_frame_four_ranks_peer_transport) { sharded_frame_case(RDF_COMM_PEER, 4); }
TEST(test_sharded_frame_one_rank_rccl) { sharded_frame_case(RDF_COMM_RCCL, 1); }

// DataFrame::from_arrow (src/dataframe.rs:391-407) on the committed pyarrow-written fixture: schema, chunking (one chunk
// per record batch), every value and validity bit, then the device path over the loaded columns.
TEST(test_from_arrow_ipc_file) {
    DataFrame df = DataFrame::from_arrow(g_arrow);
    const std::vector<std::string> names{"i8", "i32", "i64", "u16", "f32", "f64", "flag", "city"};
    const std::vector<DataType> types{DataType::Int8, DataType::Int32, DataType::Int64, DataType::UInt16, DataType::Float32, DataType::Float64, DataType::Boolean, DataType::Utf8};
    CHECK_EQ(df.num_columns(), names.size());
    for (size_t c = 0; c < names.size(); ++c) { CHECK_EQ(df.schema().fields[c].name, names[c]); CHECK(df.schema().fields[c].data_type == types[c]); }
    const std::vector<int64_t> lens{1024, 1024, 576};
    CHECK_EQ(df.num_chunks(), lens.size());
    CHECK_EQ(df.num_rows(), (int64_t)2624);
    int64_t first = 0, f64_nulls = 0;
    double f64_sum = 0;
    for (size_t b = 0; b < lens.size(); ++b) {
        const int64_t n = lens[b];
        CHECK_EQ(df.column(0).data().chunk(b)->length, n);
        auto v8 = host<int8_t>(df.column(0).data().chunk(b));
        auto v32 = host<int32_t>(df.column(1).data().chunk(b));
        auto v64 = host<int64_t>(df.column(2).data().chunk(b));
        auto v16 = host<uint16_t>(df.column(3).data().chunk(b));
        auto vf = host<float>(df.column(4).data().chunk(b));
        auto vd = host<double>(df.column(5).data().chunk(b));
        auto dvalid = df.column(5).data().chunk(b)->valid_to_host();
        auto flag = df.column(6).data().chunk(b)->bools_to_host();
        auto fvalid = df.column(6).data().chunk(b)->valid_to_host();
        CHECK(df.column(1).data().chunk(b)->validity == nullptr);   // no nulls -> no bitmap uploaded
        for (int64_t r = 0; r < n; ++r) {
            const int64_t i = first + r;
            CHECK_EQ((int)v8[(size_t)r], (int)((i % 200) - 100));
            CHECK_EQ(v32[(size_t)r], (int32_t)(7 * i - 1000));
            CHECK_EQ(v64[(size_t)r], i * 1000000000ll);
            CHECK_EQ((int)v16[(size_t)r], (int)((13 * i) % 65536));
            CHECK_EQ(vf[(size_t)r], (float)i / 8.0f);
            CHECK_EQ((bool)dvalid[(size_t)r], i % 10 != 3);
            if (i % 10 != 3) { CHECK_EQ(vd[(size_t)r], 0.5 * (double)i - 100.0); f64_sum += vd[(size_t)r]; } else ++f64_nulls;
            CHECK_EQ((bool)fvalid[(size_t)r], i % 7 != 0);
            if (i % 7 != 0) CHECK_EQ((bool)flag[(size_t)r], i % 3 == 0);
        }
        CHECK_EQ(df.column(7).data().chunk(b)->strings->at(5), "city" + std::to_string(first + 5));
        first += n;
    }
    CHECK_EQ(df.column_by_name("f64").null_count(), f64_nulls);
    // the loaded columns are ordinary device columns: aggregates and a fused filter run on them
    CHECK_NEAR(*AggregateFunctions::sum<double>(df.column_by_name("f64").data()), f64_sum, 1e-12);
    CHECK_EQ(*AggregateFunctions::sum<int64_t>(df.column_by_name("i64").data()), (int64_t)(2623ll * 2624 / 2) * 1000000000ll);
    DataFrame kept = df.filter(BooleanFilter::gt(BooleanFilter::column("i32"), BooleanFilter::scalar(Scalar((int64_t)6000))));
    CHECK_EQ(kept.num_rows(), (int64_t)(2624 - 1001));   // 7 i - 1000 > 6000  <=>  i >= 1001
    CHECK_EQ(kept.num_chunks(), lens.size());            // chunking preserved; batch 0 (rows 0..1023) keeps 23 rows
    CHECK_EQ(kept.column_by_name("city").data().chunk(0)->strings->at(0), std::string("city1001"));
    {   // Boolean and Utf8 columns ride along: row r of the kept batch 1 is global row 1024 + r
        auto flag = kept.column_by_name("flag").data().chunk(1)->bools_to_host();
        auto fvalid = kept.column_by_name("flag").data().chunk(1)->valid_to_host();
        CHECK_EQ(flag.size(), (size_t)1024);
        for (size_t r = 0; r < flag.size(); ++r) {
            const int64_t i = 1024 + (int64_t)r;
            CHECK_EQ((bool)fvalid[r], i % 7 != 0);
            if (i % 7 != 0) CHECK_EQ((bool)flag[r], i % 3 == 0);
        }
        CHECK_EQ(kept.column_by_name("city").data().chunk(2)->strings->back(), std::string("city2623"));
    }
    CHECK_THROWS(DataFrame::from_arrow("tests/golden/uk_cities_with_headers.csv"));   // not an IPC file
}

// src/functions/scalar.rs:267-273 (no reference test): 2020-10-17T13:45:10Z in four units, a pre-epoch instant, a NULL
// Dictionary-encoded columns and the IPC STREAM format (tests/golden/make_arrow_fixture.py: closed formulas of the global
// row number): utf8 / float64 / int32 dictionaries under int8 / int16 / uint8 indices, NULL indices, delta dictionaries
// between the batches of the stream; the FILE holds the same rows under one unified dictionary.  Decoded while loading.
static void check_dictionary_frame(const DataFrame& df) {
    const std::vector<int64_t> lens{700, 300, 524};
    const std::vector<int64_t> ks{5, 8, 11};
    CHECK_EQ(df.num_columns(), 4u);
    CHECK_EQ(df.num_chunks(), lens.size());
    CHECK_EQ(df.num_rows(), (int64_t)1524);
    CHECK(df.column_by_name("i64").data_type() == DataType::Int64);
    CHECK(df.column_by_name("code").data_type() == DataType::Utf8);
    CHECK(df.column_by_name("level").data_type() == DataType::Float64);
    CHECK(df.column_by_name("small").data_type() == DataType::Int32);
    int64_t first = 0, level_nulls = 0, code_nulls = 0;
    for (size_t b = 0; b < lens.size(); ++b) {
        const auto ic = df.column_by_name("i64").data().chunk(b);
        const auto cc = df.column_by_name("code").data().chunk(b);
        const auto lc = df.column_by_name("level").data().chunk(b);
        const auto sc = df.column_by_name("small").data().chunk(b);
        CHECK_EQ(ic->length, lens[b]); CHECK_EQ(cc->length, lens[b]); CHECK_EQ(lc->length, lens[b]); CHECK_EQ(sc->length, lens[b]);
        const auto iv = host<int64_t>(ic);
        const auto lv = host<double>(lc);
        const auto sv = host<int32_t>(sc);
        const auto lvalid = lc->valid_to_host(), cvalid = cc->valid_to_host();
        for (int64_t r = 0; r < lens[b]; ++r) {
            const int64_t i = first + r;
            CHECK_EQ(iv[(size_t)r], i);
            CHECK_EQ((int64_t)sv[(size_t)r], ((5 * i) % 4) * 1000 - 1500);
            if (i % 13 == 0) { CHECK(!lvalid[(size_t)r]); ++level_nulls; } else { CHECK(lvalid[(size_t)r]); CHECK_EQ(lv[(size_t)r], 0.25 * (double)((3 * i) % 6)); }
            if (i % 11 == 5) { CHECK(!cvalid[(size_t)r]); ++code_nulls; } else { CHECK(cvalid[(size_t)r]); CHECK_EQ((*cc->strings)[(size_t)r], "cat" + std::to_string((7 * i) % ks[b])); }
        }
        first += lens[b];
    }
    CHECK_EQ(df.column_by_name("level").null_count(), level_nulls);
    CHECK_EQ(df.column_by_name("code").null_count(), code_nulls);
    // the decoded columns are ordinary compute-path columns
    int64_t want = 0;
    for (int64_t i = 0; i < 1524; ++i) want += ((5 * i) % 4) * 1000 - 1500;
    CHECK_EQ((int64_t)*AggregateFunctions::sum<int32_t>(df.column_by_name("small").data()), (int64_t)(int32_t)want);
    CHECK_EQ(*AggregateFunctions::count(df.column_by_name("level").data()), (int64_t)1524 - level_nulls);
}
// DataFrame::from_arrow_host: the same file as views into its mapping — nothing uploaded —, aggregated by a LazyFrame whose fused
// filter -> aggregate pass the library streams out of host memory (rdf_pipeline over RDF_MEM_HOST, slab by slab when the frame is
// larger than a slab).  Results equal the device-resident load's; operators that produce device columns want to_device().
TEST(test_from_arrow_host_resident_frame_is_streamed) {
    using AF = P::AggregateFunction;
    DataFrame h = DataFrame::from_arrow_host(g_arrow);
    DataFrame d = DataFrame::from_arrow(g_arrow);
    CHECK_EQ(h.num_rows(), d.num_rows());
    CHECK_EQ(h.num_chunks(), d.num_chunks());
    CHECK(h.column_by_name("f64").data().chunk(0)->host && !d.column_by_name("f64").data().chunk(0)->host);
    CHECK_EQ(h.column_by_name("f64").null_count(), d.column_by_name("f64").null_count());
    CHECK_EQ(h.column_by_name("city").data().chunk(1)->strings->at(3), d.column_by_name("city").data().chunk(1)->strings->at(3));
    auto cond = BooleanFilter::gt(BooleanFilter::column("i32"), BooleanFilter::scalar(Scalar((int64_t)6000)));
    const std::vector<P::Aggregation> aggs{{AF::Sum, {"f64", "i64"}}, {AF::Count, {"f64"}}, {AF::Min, {"f32"}}, {AF::Max, {"u16"}}, {AF::Avg, {"i8"}}};
    auto run = [&](const DataFrame& f) { return LazyFrame::read(f).filter(cond).aggregate({}, aggs).evaluate(); };
    const DataFrame want = run(d);
    auto same = [&](const DataFrame& got) {
        CHECK_EQ(got.num_columns(), want.num_columns());
        for (size_t c = 0; c < want.num_columns(); ++c) {
            CHECK_EQ(got.schema().fields[c].name, want.schema().fields[c].name);
            CHECK(got.schema().fields[c].data_type == want.schema().fields[c].data_type);
        }
        CHECK_NEAR(got.column(0).data().chunk(0)->value<double>(0), want.column(0).data().chunk(0)->value<double>(0), 1e-12);
        CHECK_EQ(got.column(1).data().chunk(0)->value<int64_t>(0), want.column(1).data().chunk(0)->value<int64_t>(0));
        CHECK_EQ(got.column(2).data().chunk(0)->value<uint32_t>(0), want.column(2).data().chunk(0)->value<uint32_t>(0));
        CHECK_EQ(got.column(3).data().chunk(0)->value<float>(0), want.column(3).data().chunk(0)->value<float>(0));
        CHECK_EQ((int)got.column(4).data().chunk(0)->value<uint16_t>(0), (int)want.column(4).data().chunk(0)->value<uint16_t>(0));
        CHECK_NEAR(got.column(5).data().chunk(0)->value<double>(0), want.column(5).data().chunk(0)->value<double>(0), 1e-12);
    };
    int64_t slabs = -1, staged = 0, direct = 0;
    same(run(h));                                     // below one slab: staged whole, one fused pass
    check(rdf_stream_stats(&slabs, &staged, &direct));
    CHECK_EQ(slabs, (int64_t)0);
    check(rdf_set_option("stream_slab_bytes", 8192));  // now the 2624 rows are many slabs: uploader thread, partials folded in slab order
    try {
        same(run(h));
        check(rdf_stream_stats(&slabs, &staged, &direct));
        CHECK(slabs >= 2);
        CHECK(staged > 0);
    } catch (...) { (void)rdf_set_option("stream_slab_bytes", 0); throw; }
    check(rdf_set_option("stream_slab_bytes", 0));
    // column aggregates straight from the mapping
    CHECK_NEAR(*AggregateFunctions::sum<double>(h.column_by_name("f64").data()), *AggregateFunctions::sum<double>(d.column_by_name("f64").data()), 1e-12);
    // round 5: operators over a host-resident frame return host-resident frames — the library streams the batches through HBM and
    // the results come back (the values are read straight out of host memory: no device copy behind value<T>()).
    DataFrame up = h.to_device();
    CHECK(!up.column_by_name("f64").data().chunk(0)->host);
    CHECK_EQ(up.filter(cond).num_rows(), (int64_t)(2624 - 1001));
    same(run(up));
    {
        // (a) DataFrame::filter with text columns aboard: streamed predicate -> host mask, every column compacted by it
        const DataFrame hf = h.filter(cond), df = up.filter(cond);
        CHECK_EQ(hf.num_rows(), df.num_rows());
        CHECK(hf.column_by_name("f64").data().chunk(0)->host);
        CHECK_EQ(hf.column_by_name("city").data().chunk(2)->strings->size(), df.column_by_name("city").data().chunk(2)->strings->size());
        // (b) the numeric columns alone: ONE rdf_filter_pipeline call, also across many slabs
        const DataFrame hn = h.select({"f64", "i64", "i32", "f32", "u16", "i8"}), dn = up.select({"f64", "i64", "i32", "f32", "u16", "i8"});
        auto cmp_frames = [&](const DataFrame& a, const DataFrame& b) {
            CHECK_EQ(a.num_rows(), b.num_rows());
            CHECK_EQ(a.num_chunks(), b.num_chunks());
            for (size_t c = 0; c < b.num_columns(); ++c)
                for (size_t i = 0; i < b.num_chunks(); ++i) {
                    const auto &x = a.column(c).data().chunk(i), &y = b.column(c).data().chunk(i);
                    CHECK_EQ(x->length, y->length);
                    CHECK_EQ(x->count_nulls(), y->count_nulls());
                    CHECK(x->valid_to_host() == y->valid_to_host());
                }
            CHECK(a.column_by_name("i64").data().chunk(1)->values_to_host<int64_t>() == b.column_by_name("i64").data().chunk(1)->values_to_host<int64_t>());
            CHECK(a.column_by_name("u16").data().chunk(2)->values_to_host<uint16_t>() == b.column_by_name("u16").data().chunk(2)->values_to_host<uint16_t>());
        };
        const DataFrame want_n = dn.filter(cond);
        cmp_frames(hn.filter(cond), want_n);
        check(rdf_set_option("stream_slab_bytes", 8192));
        try {
            const DataFrame got = hn.filter(cond);
            check(rdf_stream_stats(&slabs, &staged, &direct));
            CHECK(slabs >= 2);
            CHECK(got.column(0).data().chunk(0)->host);
            cmp_frames(got, want_n);
            // (c) a Calculate step (new column) and a hash GROUP BY over the host-resident frame: streamed, host-resident results
            auto calc = [&](const DataFrame& f) {
                return LazyFrame::read(f).with_column("y", P::Function::Scalar_(P::ScalarFunction::Sine), {"f64"}).evaluate();
            };
            const DataFrame hy = calc(hn), dy = calc(dn);
            CHECK(hy.column_by_name("y").data().chunk(0)->host);
            const auto a = hy.column_by_name("y").data().chunk(1)->values_to_host<double>(), b = dy.column_by_name("y").data().chunk(1)->values_to_host<double>();
            CHECK_EQ(a.size(), b.size());
            for (size_t i = 0; i < a.size(); ++i) CHECK(a[i] == b[i] || (a[i] != a[i] && b[i] != b[i]));
        } catch (...) { (void)rdf_set_option("stream_slab_bytes", 0); throw; }
        check(rdf_set_option("stream_slab_bytes", 0));
    }
    // dictionary-encoded columns are decoded on the device: refused here
    CHECK_THROWS(DataFrame::from_arrow_host("tests/golden/dict_batches.arrow"));
}
TEST(test_from_arrow_stream_with_delta_dictionaries) { check_dictionary_frame(DataFrame::from_arrow("tests/golden/dict_batches.arrows")); }
TEST(test_from_arrow_file_with_dictionaries) { check_dictionary_frame(DataFrame::from_arrow("tests/golden/dict_batches.arrow")); }
TEST(test_from_arrow_rejects_garbage) {
    const std::vector<uint8_t> junk(64, 0x5A);
    CHECK_THROWS(DataFrame::from_arrow_image(junk.data(), junk.size()));
    CHECK_THROWS(DataFrame::from_arrow("tests/golden/uk_cities_with_headers.csv"));
}

TEST(test_hour_of_timestamps) {
    const int64_t t = 1602942310;
    std::vector<bool> valid{true, true, false};
    CHECK(host<int32_t>(ScalarFunctions::hour({Array::from_vec<int64_t>({t, -1, 0}, &valid)}, RDF_TIME_SECOND)[0]) == std::vector<int32_t>({13, 23, 0}));
    CHECK(host<int32_t>(ScalarFunctions::hour({Array::from_vec<int64_t>({t * 1000 + 999, -1})}, RDF_TIME_MILLISECOND)[0]) == std::vector<int32_t>({13, 23}));
    CHECK(host<int32_t>(ScalarFunctions::hour({Array::from_vec<int64_t>({t * 1000000000})}, RDF_TIME_NANOSECOND)[0]) == std::vector<int32_t>({13}));
    CHECK(host<int32_t>(ScalarFunctions::hour({Array::from_vec<int32_t>({49510})}, RDF_TIME_SECOND)[0]) == std::vector<int32_t>({13}));   // Time32(Second)
    CHECK_EQ(ScalarFunctions::hour({Array::from_vec<int64_t>({t, -1, 0}, &valid)}, RDF_TIME_SECOND)[0]->null_count, 1);
}

// src/functions/array.rs:421-640 — the reference's own ArrayFunctions tests on its 16-value / 6-row fixture
template <class T> static ListArray array_fixture() {
    std::vector<T> v;
    for (int x : {0, 0, 0, 1, 2, 1, 3, 4, 5, 1, 3, 2, 3, 2, 8, 3}) v.push_back((T)x);
    return ListArray::from_parts({0, 3, 6, 8, 12, 14, 16}, Array::from_vec<T>(v));
}
TEST(test_array_contains_i32s) {
    auto bools = ArrayFunctions::array_contains<int32_t>(array_fixture<int32_t>(), 2)->bools_to_host();
    CHECK(bools == std::vector<bool>({false, true, false, true, true, false}));
}
TEST(test_array_contains_i64s) {
    auto bools = ArrayFunctions::array_contains<int64_t>(array_fixture<int64_t>(), 2)->bools_to_host();
    CHECK(bools == std::vector<bool>({false, true, false, true, true, false}));
}
TEST(test_array_contains_f64s) {
    auto bools = ArrayFunctions::array_contains<double>(array_fixture<double>(), 2.0)->bools_to_host();
    CHECK(bools == std::vector<bool>({false, true, false, true, true, false}));
}
TEST(test_array_position) {
    auto pos = ArrayFunctions::array_position<int64_t>(array_fixture<int64_t>(), 2)->values_to_host<int32_t>();
    CHECK(pos == std::vector<int32_t>({0, 2, 0, 4, 2, 0}));
}
TEST(test_array_remove) {
    auto b = ArrayFunctions::array_remove<int64_t>(array_fixture<int64_t>(), 2);
    CHECK_EQ(b.len(), 6);
    CHECK_EQ(b.values()->length, 13);
    CHECK(b.value_offsets() == std::vector<int32_t>({0, 3, 5, 7, 10, 11, 13}));
}
TEST(test_array_sort) {
    auto b = ArrayFunctions::array_sort<int64_t>(array_fixture<int64_t>());
    CHECK_EQ(b.len(), 6);
    CHECK_EQ(b.values()->length, 16);
    CHECK(b.value_offsets() == std::vector<int32_t>({0, 3, 6, 8, 12, 14, 16}));
    CHECK(b.values()->values_to_host<int64_t>() == std::vector<int64_t>({0, 0, 0, 1, 1, 2, 3, 4, 1, 2, 3, 5, 2, 3, 3, 8}));
}
// the set-valued functions on the array_tool crate's documented examples; a NULL row becomes an empty valid list
TEST(test_array_set_functions) {
    using Rows = std::vector<std::optional<std::vector<int64_t>>>;
    using Out = std::vector<std::vector<int64_t>>;
    auto L = [](const Rows& r) { return ListArray::from_rows<int64_t>(r); };
    CHECK(ArrayFunctions::array_distinct<int64_t>(L({{{1, 2, 1, 3, 2, 3, 4, 5, 6}}, std::nullopt})).rows_to_host<int64_t>() == Out({{1, 2, 3, 4, 5, 6}, {}}));
    CHECK(ArrayFunctions::array_except<int64_t>(L({{{1, 2, 3, 4, 5, 6}}}), L({{{1, 2, 5, 7, 9}}})).rows_to_host<int64_t>() == Out({{3, 4, 6}}));
    CHECK(ArrayFunctions::array_intersect<int64_t>(L({{{1, 1, 3, 5}}}), L({{{1, 2, 3}}})).rows_to_host<int64_t>() == Out({{1, 3}}));
    CHECK(ArrayFunctions::array_union<int64_t>(L({{{1, 2, 3, 4, 5, 6}}}), L({{{5, 6, 7, 8, 9}}})).rows_to_host<int64_t>() == Out({{1, 2, 3, 4, 5, 6, 7, 8, 9}}));
    CHECK(ArrayFunctions::array_repeat<int64_t>(L({{{1, 2, 3}}, std::nullopt}), 3).rows_to_host<int64_t>() == Out({{1, 2, 3, 1, 2, 3, 1, 2, 3}, {}}));
    CHECK(ArrayFunctions::array_max<int32_t>(array_fixture<int32_t>())->values_to_host<int32_t>() == std::vector<int32_t>({0, 2, 4, 5, 3, 8}));
    CHECK(ArrayFunctions::array_min<int32_t>(array_fixture<int32_t>())->values_to_host<int32_t>() == std::vector<int32_t>({0, 1, 3, 1, 2, 3}));
    bool threw = false;   // array.rs:72-76
    try { ArrayFunctions::array_union<int64_t>(L({{{1}}}), L({{{1}}, {{2}}})); } catch (const DataFrameError& e) { threw = std::string(e.what()).find("same length") != std::string::npos; }
    CHECK(threw);
}

int main(int argc, char** argv) {
    if (argc > 1) g_csv = argv[1];
    if (argc > 2) g_arrow = argv[2];
    return run_all();
}
