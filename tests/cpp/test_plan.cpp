// Plan builders (src/operation/scalar.rs) — no device needed.
#include "mini_test.hpp"
#include "rdf_frame.hpp"
#include <unistd.h>

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

// ---------------------------------------------------------------- src/optimiser.rs:237-420 (the module's own #[test]s)
static const char* kCities = "tests/golden/uk_cities_with_headers.csv";
static Reader cities_reader(std::optional<size_t> max_records = std::nullopt) {
    CsvReadOptions o;
    o.has_headers = true; o.delimiter = (uint8_t)','; o.max_records = max_records; o.batch_size = 1024;
    return Reader::Csv_(kCities, o);
}
TEST(test_read_project) {   // :241-274
    const Computation computation = Computation::compute_read(cities_reader());
    CHECK_EQ(computation.output.columns.size(), 3u);
    CHECK(computation.output.columns[0].data_type == DataType::Utf8 && computation.output.columns[1].data_type == DataType::Float64);
    Computation select;
    select.input = {computation.output};
    select.transformations = {Transformation::Select_({"city", "lat"})};
    select.output = computation.output;
    const std::vector<Computation> optimised = optimise({select, computation});   // read comes last
    CHECK_EQ(optimised.size(), 1u);
    CHECK_EQ(optimised[0].output.columns.size(), 2u);
    CHECK(optimised[0].is_single(Transformation::Read));
    CHECK(optimised[0].transformations[0].reader.csv.projection == std::optional<std::vector<size_t>>(std::vector<size_t>{0, 1}));
}
TEST(test_read_limit_project) {   // :276-305
    rdf::LazyFrame frame = rdf::LazyFrame::read(Computation::compute_read(cities_reader()));
    frame = frame.select({"city", "lat"});
    frame = frame.limit(32);
    const std::vector<Computation> computations = frame.unroll();
    CHECK_EQ(computations.size(), 3u);
    std::vector<Computation> optimised = optimise(computations);
    CHECK_EQ(optimised.size(), 2u);
    CHECK_EQ(optimised[0].output.columns.size(), 2u);
    optimised = optimise(optimised);   // optimise again to join the select with the limit
    CHECK_EQ(optimised.size(), 1u);
    CHECK_EQ(optimised[0].output.columns.size(), 2u);
    const CsvReadOptions& o = optimised[0].transformations[0].reader.csv;
    CHECK(o.max_records == std::optional<size_t>(32));
    CHECK(o.projection == std::optional<std::vector<size_t>>(std::vector<size_t>{0, 1}));
    CHECK_EQ(frame.optimised().size(), 1u);   // the fixed point the evaluator uses
}
TEST(test_optimise_filter_and_limits) {   // :380-404 (test_filter only prints the plan) + the Limit rules (:58-75)
    rdf::LazyFrame frame = rdf::LazyFrame::read(Computation::compute_read(cities_reader()));
    frame = frame.select({"city", "lat", "lng"});
    frame = frame.filter(rdf::BooleanFilter::gt(rdf::BooleanFilter::column("lat"), rdf::BooleanFilter::scalar(rdf::Scalar((int64_t)0))));
    const std::vector<Computation> computations = frame.unroll();
    CHECK_EQ(computations.size(), 3u);
    const std::vector<Computation> optimised = optimise(computations);   // the filter is a barrier: the select still reaches the read
    CHECK_EQ(optimised.size(), 2u);
    CHECK(optimised[0].is_single(Transformation::Filter));
    CHECK(optimised[1].is_single(Transformation::Read) && optimised[1].transformations[0].reader.csv.projection.has_value());
    // two limits merge into the smaller one, and a limit next to the read becomes max_records
    rdf::LazyFrame two = rdf::LazyFrame::read(Computation::compute_read(cities_reader())).limit(20).limit(7);
    const std::vector<Computation> merged = two.optimised();
    CHECK(merged.back().is_single(Transformation::Read));
    CHECK(merged.back().transformations[0].reader.csv.max_records == std::optional<size_t>(7));
    // an existing max_records is only ever lowered
    rdf::LazyFrame capped = rdf::LazyFrame::read(Computation::compute_read(cities_reader(5))).limit(20);
    CHECK(capped.optimised().back().transformations[0].reader.csv.max_records == std::optional<size_t>(5));
}
TEST(test_optimise_select_and_calculate) {   // optimise_project_calc (:183-235) and the pending-computation fix
    rdf::LazyFrame frame = rdf::LazyFrame::read(Computation::compute_read(cities_reader()));
    // the selected columns include the computed one and its input: the select moves above the calculation
    rdf::LazyFrame a = frame.with_column("sin_lat", Function::Scalar_(ScalarFunction::Sine), {"lat"}).select({"lat", "sin_lat"});
    std::vector<Computation> o = optimise(a.unroll());
    CHECK(!o.empty() && o.back().is_single(Transformation::Read));   // nothing is lost: the read is still the plan's last step
    // the computed column is not selected: the calculation is dropped
    rdf::LazyFrame b = frame.with_column("sin_lat", Function::Scalar_(ScalarFunction::Sine), {"lat"}).select({"city"});
    o = b.optimised();
    bool has_calc = false;
    for (auto& c : o) for (auto& t : c.transformations) has_calc |= t.kind == Transformation::Calculate;
    CHECK(!has_calc);
    CHECK(o.back().transformations[0].reader.csv.projection == std::optional<std::vector<size_t>>(std::vector<size_t>{0}));
    // [Calculate, Read]: the reference's optimise returns [Calculate] and loses the read; here the pending read is emitted
    rdf::LazyFrame c = frame.with_column("sin_lat", Function::Scalar_(ScalarFunction::Sine), {"lat"});
    o = optimise(c.unroll());
    CHECK_EQ(o.size(), 2u);
    CHECK(o[1].is_single(Transformation::Read));
}

// ---- the CSV text layer of DataFrame::from_csv / Reader::get_dataset (host only): records, cells, inferred types, parsed values
static std::string write_tmp(const std::string& name, const std::string& text) {
    const std::string path = std::string("/tmp/rdf_csv_") + name + "_" + std::to_string((long)getpid()) + ".csv";
    FILE* f = std::fopen(path.c_str(), "wb");
    std::fwrite(text.data(), 1, text.size(), f);
    std::fclose(f);
    return path;
}
struct ParsedCsv {
    rdf::csv::Text text;
    std::vector<rdf::csv::Inferred> types;
    std::vector<std::vector<uint8_t>> values, validity;
    std::vector<rdf::csv::Filled> filled;
    int64_t i64(size_t c, size_t r) const { int64_t v; std::memcpy(&v, values[c].data() + 8 * r, 8); return v; }
    double f64(size_t c, size_t r) const { double v; std::memcpy(&v, values[c].data() + 8 * r, 8); return v; }
    bool bit(const std::vector<uint8_t>& b, size_t r) const { return (b[r >> 3] >> (r & 7)) & 1; }
};
static ParsedCsv parse_csv(const std::string& path, bool headers = true, char delim = ',', std::optional<size_t> max_records = std::nullopt, int threads = 0) {
    ParsedCsv p;
    p.text = rdf::csv::load(path, headers, delim, max_records);
    std::vector<size_t> cols;
    for (size_t i = 0; i < p.text.header.size(); ++i) cols.push_back(i);
    p.types = rdf::csv::infer(p.text, cols, threads);
    const size_t n = p.text.records.size();
    std::vector<uint8_t*> pv, pb;
    p.values.resize(cols.size()); p.validity.resize(cols.size());
    for (size_t k = 0; k < cols.size(); ++k) {
        p.values[k].assign(n * 8 + 64, 0xAB);
        p.validity[k].assign(n / 8 + 64, 0xAB);
        pv.push_back(p.types[k].dtype == DataType::Utf8 ? nullptr : p.values[k].data());
        pb.push_back(p.types[k].dtype == DataType::Utf8 ? nullptr : p.validity[k].data());
    }
    p.filled = rdf::csv::fill(p.text, cols, p.types, pv, pb, threads);
    return p;
}

TEST(csv_cells_types_and_values) {
    const std::string path = write_tmp("mixed",
        "id,price,flag,name,big,\"quoted, header\"\r\n"
        "1,1.5,true,alpha,1,\"x, y\"\r\n"
        "\n"                                             // a zero-length line is skipped
        "-2,,False,\"be\"\"ta\",99999999999999999999,z\r\n"
        "+3,1e3,TRUE,,3,\n"
        "4,-0.0,false,delta\n"                           // a short record: the missing cells are NULL
        " 5,inf,true,eps,5,a,extra,cells\n");            // a long one is cut; strtoll / strtod take leading blanks
    const ParsedCsv p = parse_csv(path);
    CHECK_EQ(p.text.header.size(), 6u);
    CHECK_EQ(p.text.header[5], std::string("quoted, header"));
    CHECK_EQ(p.text.records.size(), 5u);
    CHECK(p.types[0].dtype == DataType::Int64 && !p.types[0].any_null);
    CHECK(p.types[1].dtype == DataType::Float64 && p.types[1].any_null);
    CHECK(p.types[2].dtype == DataType::Boolean);
    CHECK(p.types[3].dtype == DataType::Utf8);
    CHECK(p.types[4].dtype == DataType::Float64);        // one cell overflows Int64: the column is numbers
    CHECK(p.types[5].dtype == DataType::Utf8);
    const int64_t ids[5] = {1, -2, 3, 4, 5};
    for (size_t r = 0; r < 5; ++r) { CHECK_EQ(p.i64(0, r), ids[r]); CHECK(p.bit(p.validity[0], r)); }
    CHECK_EQ(p.f64(1, 0), 1.5); CHECK(!p.bit(p.validity[1], 1)); CHECK_EQ(p.f64(1, 1), 0.0); CHECK_EQ(p.f64(1, 2), 1000.0);
    CHECK(std::signbit(p.f64(1, 3)) && p.f64(1, 3) == 0.0); CHECK(std::isinf(p.f64(1, 4)));
    CHECK_EQ(p.filled[1].nulls, (int64_t)1);
    const bool flags[5] = {true, false, true, false, true};
    for (size_t r = 0; r < 5; ++r) CHECK_EQ(p.bit(p.values[2], r), flags[r]);
    CHECK_EQ(p.filled[3].strings[1], std::string("beta"));   // quotes toggle and are dropped wherever they stand
    CHECK_EQ(p.filled[3].strings[2], std::string(""));
    CHECK_EQ(p.f64(4, 1), 1e20); CHECK(!p.bit(p.validity[4], 3)); CHECK_EQ(p.filled[4].nulls, (int64_t)1);
    CHECK_EQ(p.filled[5].strings[0], std::string("x, y")); CHECK_EQ(p.filled[5].strings[3], std::string("")); CHECK_EQ(p.filled[5].strings[4], std::string("a"));
    // no header row: arrow's column_N names, the first line is data; another delimiter; max_records
    const std::string p2 = write_tmp("nohdr", "1;2.5\n3;4.5\n5;6.5\n");
    const ParsedCsv q = parse_csv(p2, false, ';', (size_t)2);
    CHECK_EQ(q.text.header[0], std::string("column_1")); CHECK_EQ(q.text.header[1], std::string("column_2"));
    CHECK_EQ(q.text.records.size(), 2u);
    CHECK(q.types[0].dtype == DataType::Int64 && q.types[1].dtype == DataType::Float64);
    CHECK_EQ(q.i64(0, 1), (int64_t)3); CHECK_EQ(q.f64(1, 1), 4.5);
    // a column of empty cells only is Utf8; the planner's reader sees the same schema
    const std::string p3 = write_tmp("empty", "a,b\n1,\n2,\n");
    const ParsedCsv e = parse_csv(p3);
    CHECK(e.types[1].dtype == DataType::Utf8);
    const Dataset d = Reader::Csv_(path).get_dataset();
    CHECK_EQ(d.columns.size(), 6u);
    CHECK(d.columns[1].data_type == DataType::Float64 && d.columns[2].data_type == DataType::Boolean && d.columns[3].data_type == DataType::Utf8);
    std::remove(path.c_str()); std::remove(p2.c_str()); std::remove(p3.c_str());
}

TEST(csv_worker_threads_agree_with_one_thread) {
    // 300 007 records (several ranges of records, the last one ragged): integers, doubles printed with 17 significant digits
    // (they must come back bit for bit), a NULL every 7th / 11th row, a type that only the LAST range breaks
    std::string text = "k,x,late\n";
    const size_t n = 300007;     // ~11 MB of text: csv::load indexes the records with several threads too
    std::vector<double> xs(n);
    uint64_t st = 88172645463325252ull;
    for (size_t r = 0; r < n; ++r) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        xs[r] = (double)(int64_t)(st >> 11) / 9007199254740992.0 * 2000.0 - 1000.0;
        char buf[96];
        std::snprintf(buf, sizeof buf, "%lld,%s%.17g,%s\n", (long long)r * 7 - 1000, "", xs[r], r + 1 == n ? "1.25" : "7");
        std::string line = buf;
        if (r % 7 == 3) line = std::to_string((long long)r * 7 - 1000) + ",," + (r + 1 == n ? "1.25" : "7") + "\n";
        if (r % 11 == 5) line = "," + line.substr(line.find(',') + 1);
        text += line;
    }
    const std::string path = write_tmp("big", text);
    const ParsedCsv one = parse_csv(path, true, ',', std::nullopt, 1), many = parse_csv(path, true, ',', std::nullopt, 5);
    CHECK_EQ(one.text.records.size(), n);
    for (const ParsedCsv* p : {&one, &many}) {
        CHECK(p->types[0].dtype == DataType::Int64 && p->types[0].any_null);
        CHECK(p->types[1].dtype == DataType::Float64 && p->types[1].any_null);
        CHECK(p->types[2].dtype == DataType::Float64 && !p->types[2].any_null);   // "1.25" in the last record only
        int64_t nulls0 = 0, nulls1 = 0;
        for (size_t r = 0; r < n; ++r) {
            const bool k_null = r % 11 == 5, x_null = r % 7 == 3;
            CHECK_EQ(p->bit(p->validity[0], r), !k_null);
            CHECK_EQ(p->bit(p->validity[1], r), !x_null);
            if (!k_null) CHECK_EQ(p->i64(0, r), (int64_t)r * 7 - 1000); else { ++nulls0; CHECK_EQ(p->i64(0, r), (int64_t)0); }
            if (!x_null) CHECK_EQ(p->f64(1, r), xs[r]); else ++nulls1;
            CHECK_EQ(p->f64(2, r), r + 1 == n ? 1.25 : 7.0);
        }
        CHECK_EQ(p->filled[0].nulls, nulls0); CHECK_EQ(p->filled[1].nulls, nulls1);
    }
    std::remove(path.c_str());
}

int main() { return run_all(); }
