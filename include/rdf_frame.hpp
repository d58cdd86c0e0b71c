// rdf_frame.hpp — C++17 host-side mirror of rust-dataframe's interface for the hot path, header-only
// over the C ABI of rdf_mi355x.h.  Columns live in HBM (RDF_MEM_DEVICE); every operation below is a
// call into librdf_mi355x.so — there is no host compute path here.
//
// Mirrors (paths relative to the reference root):
//   ChunkedArray / Column            src/table.rs:13-344        (from_arrays, slice, filter, take)
//   DataFrame                        src/dataframe.rs:30-337    (with_column, with_column_renamed, limit,
//                                                                filter, select, drop, to_record_batches)
//   Scalar / BooleanFilter           src/expression.rs:718-870  (eval_to_array)
//   Column / Calculation / Function / ScalarFunction / Transformation / Computation / Aggregation
//                                    src/expression.rs:286-712
//   Add/Subtract/Cast/Sin operations src/operation/scalar.rs:18-318 (plan builders: names, cast insertion)
//   Evaluate::{evaluate, calculate}  src/evaluation.rs:54-323
//   LazyFrame                        src/lazyframe.rs:15-315
//   AggregateFunctions               src/functions/aggregate.rs:12-93
// Where the reference materialises one DataFrame per plan step, Evaluate::evaluate here FUSES maximal
// runs of Calculate / Filter / aggregate steps into single passes over HBM (rdf_pipeline).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iterator>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <charconv>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <utility>
#include <variant>
#include <vector>

#include "rdf_mi355x.h"

namespace rdf {

// ------------------------------------------------------------------------------------------------
// errors (src/error.rs:6-15)

struct DataFrameError : std::runtime_error {
    enum Kind { MemoryError, ParseError, ComputeError, DivideByZero, IoError, NoneError, ArrowError, SqlError, DeviceError } kind;
    DataFrameError(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};
inline void check(rdf_status s) {
    if (s == RDF_OK) return;
    const std::string msg = rdf_last_error();
    switch (s) {
        case RDF_COMPUTE_ERROR: throw DataFrameError(DataFrameError::ComputeError, msg);
        case RDF_DIVIDE_BY_ZERO: throw DataFrameError(DataFrameError::DivideByZero, msg);
        case RDF_INVALID_ARGUMENT: throw DataFrameError(DataFrameError::ArrowError, msg);
        case RDF_MEMORY_ERROR: throw DataFrameError(DataFrameError::MemoryError, msg);
        default: throw DataFrameError(DataFrameError::DeviceError, msg);
    }
}

// ------------------------------------------------------------------------------------------------
// data types

enum class DataType : int32_t {
    Int8 = RDF_I8, Int16 = RDF_I16, Int32 = RDF_I32, Int64 = RDF_I64, UInt8 = RDF_U8, UInt16 = RDF_U16,
    UInt32 = RDF_U32, UInt64 = RDF_U64, Float32 = RDF_F32, Float64 = RDF_F64, Boolean = RDF_BOOL, Utf8 = 100
};
inline const char* type_name(DataType t) {
    switch (t) {
        case DataType::Int8: return "Int8"; case DataType::Int16: return "Int16"; case DataType::Int32: return "Int32";
        case DataType::Int64: return "Int64"; case DataType::UInt8: return "UInt8"; case DataType::UInt16: return "UInt16";
        case DataType::UInt32: return "UInt32"; case DataType::UInt64: return "UInt64"; case DataType::Float32: return "Float32";
        case DataType::Float64: return "Float64"; case DataType::Boolean: return "Boolean"; default: return "Utf8";
    }
}
inline int type_size(DataType t) {
    switch (t) {
        case DataType::Int8: case DataType::UInt8: return 1;
        case DataType::Int16: case DataType::UInt16: return 2;
        case DataType::Int32: case DataType::UInt32: case DataType::Float32: return 4;
        case DataType::Int64: case DataType::UInt64: case DataType::Float64: return 8;
        default: return 0;
    }
}
inline bool is_integer(DataType t) { return (int)t <= RDF_U64; }
inline bool is_float(DataType t) { return t == DataType::Float32 || t == DataType::Float64; }
template <class T> struct TypeOf;
template <> struct TypeOf<int8_t> { static constexpr DataType value = DataType::Int8; };
template <> struct TypeOf<int16_t> { static constexpr DataType value = DataType::Int16; };
template <> struct TypeOf<int32_t> { static constexpr DataType value = DataType::Int32; };
template <> struct TypeOf<int64_t> { static constexpr DataType value = DataType::Int64; };
template <> struct TypeOf<uint8_t> { static constexpr DataType value = DataType::UInt8; };
template <> struct TypeOf<uint16_t> { static constexpr DataType value = DataType::UInt16; };
template <> struct TypeOf<uint32_t> { static constexpr DataType value = DataType::UInt32; };
template <> struct TypeOf<uint64_t> { static constexpr DataType value = DataType::UInt64; };
template <> struct TypeOf<float> { static constexpr DataType value = DataType::Float32; };
template <> struct TypeOf<double> { static constexpr DataType value = DataType::Float64; };

struct Field {
    std::string name;
    DataType data_type;
    bool nullable = true;
};
struct Schema {
    std::vector<Field> fields;
    std::optional<std::pair<size_t, Field>> column_with_name(const std::string& n) const {
        for (size_t i = 0; i < fields.size(); ++i) if (fields[i].name == n) return std::make_pair(i, fields[i]);
        return std::nullopt;
    }
};

// ------------------------------------------------------------------------------------------------
// device buffers and arrays (Arc<dyn Array> resident in HBM)

class DeviceBuffer {
  public:
    explicit DeviceBuffer(int64_t bytes) : bytes_(bytes) { check(rdf_dev_alloc(&ptr_, bytes + 64)); }
    // memory that belongs to something else (a frame the library returned): `keep` holds the owner alive
    DeviceBuffer(void* borrowed, int64_t bytes, std::shared_ptr<void> keep) : ptr_(borrowed), bytes_(bytes), keep_(std::move(keep)) {}
    ~DeviceBuffer() { if (ptr_ && !keep_) (void)rdf_dev_free(ptr_); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void* data() const { return ptr_; }
    int64_t bytes() const { return bytes_; }
  private:
    void* ptr_ = nullptr;
    int64_t bytes_;
    std::shared_ptr<void> keep_;
};
using BufferRef = std::shared_ptr<DeviceBuffer>;

// Page-locked host memory (rdf_host_alloc): what a reader parses into / reads a file into, so that its uploads are plain DMA
// (no runtime staging copy) and can run on the copy stream while the reader goes on (rdf_copy_h2d_async ... rdf_copy_fence).
class PinnedBuffer {
  public:
    explicit PinnedBuffer(int64_t bytes) : bytes_(bytes) { check(rdf_host_alloc(&ptr_, bytes + 64)); }
    ~PinnedBuffer() { if (ptr_) (void)rdf_host_free(ptr_); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    uint8_t* data() const { return (uint8_t*)ptr_; }
    int64_t bytes() const { return bytes_; }
  private:
    void* ptr_ = nullptr;
    int64_t bytes_;
};
// One load (from_csv / from_arrow): uploads from pinned memory are queued, ONE fence at the end.  `bytes` counts what went up.
struct IngestStats { int64_t bytes = 0, async_copies = 0, blocking_copies = 0; double seconds = 0, parse_seconds = 0; };
inline IngestStats& last_ingest() { static thread_local IngestStats s; return s; }
inline void upload(void* dst, const void* src, int64_t bytes, bool src_pinned) {
    if (bytes <= 0) return;
    IngestStats& st = last_ingest();
    st.bytes += bytes;
    if (src_pinned) { check(rdf_copy_h2d_async(dst, src, bytes)); ++st.async_copies; }
    else { check(rdf_copy_h2d(dst, src, bytes)); ++st.blocking_copies; }
}

// Uploads out of PAGEABLE memory (a file mapped into the address space): page-locking a whole multi-GB image costs more than moving
// it (hipHostMalloc of 2.57 GB: 0.44 s; the link moves it in 0.05 s), so the bytes go through two page-locked staging buffers of
// kGroup bytes that live as long as the thread: worker threads copy the pieces of one group out of the page cache while the
// copy stream drains the other.  Never pins more than 2 x kGroup, whatever the file's size.
class StagedUploader {
  public:
    static StagedUploader& instance() { static thread_local StagedUploader u; return u; }
    void push(void* dst, const void* src, int64_t bytes) {
        IngestStats& st = last_ingest();
        st.bytes += bytes > 0 ? bytes : 0;
        const uint8_t* p = (const uint8_t*)src;
        uint8_t* d = (uint8_t*)dst;
        while (bytes > 0) {
            if (!grp_[0]) { grp_[0] = std::make_unique<PinnedBuffer>(kGroup); grp_[1] = std::make_unique<PinnedBuffer>(kGroup); }
            const int64_t take = std::min(bytes, kGroup - used_);
            pieces_.push_back(Piece{d, p, take, used_});
            used_ += (take + 255) / 256 * 256;
            d += take; p += take; bytes -= take;
            if (used_ >= kGroup) flush();
        }
    }
    void flush() {      // the current group: pages -> staging buffer (threads), staging buffer -> HBM (queued on the copy stream)
        if (pieces_.empty()) return;
        uint8_t* base = grp_[cur_]->data();
        {   // long pieces are cut so that every thread has work: a group usually holds a few dozen column buffers
            std::vector<Piece> cut;
            for (const Piece& q : pieces_)
                for (int64_t o = 0; o < q.bytes; o += kSlice) cut.push_back(Piece{q.dst + o, q.src + o, std::min(kSlice, q.bytes - o), q.off + o});
            pieces_.swap(cut);
        }
        const int T = (int)std::max<size_t>(1, std::min<size_t>({pieces_.size(), (size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)12}));
        auto work = [&](int t) { for (size_t k = (size_t)t; k < pieces_.size(); k += (size_t)T) std::memcpy(base + pieces_[k].off, pieces_[k].src, (size_t)pieces_[k].bytes); };
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
        IngestStats& st = last_ingest();
        for (const Piece& q : pieces_) { check(rdf_copy_h2d_async(q.dst, base + q.off, q.bytes)); ++st.async_copies; }
        pieces_.clear();
        used_ = 0;
        inflight_[cur_] = true;
        cur_ ^= 1;
        if (inflight_[cur_]) { check(rdf_copy_fence()); inflight_[0] = inflight_[1] = false; }   // the buffer we turn to must have left
    }
    void finish() { flush(); check(rdf_copy_fence()); inflight_[0] = inflight_[1] = false; }
  private:
    struct Piece { uint8_t* dst; const uint8_t* src; int64_t bytes, off; };
    static constexpr int64_t kGroup = (int64_t)128 << 20, kSlice = (int64_t)4 << 20;
    std::unique_ptr<PinnedBuffer> grp_[2];
    std::vector<Piece> pieces_;
    int64_t used_ = 0;
    int cur_ = 0;
    bool inflight_[2] = {false, false};
};

inline std::vector<uint8_t> pack_bits(const std::vector<bool>& bits) {
    std::vector<uint8_t> out((bits.size() + 63) / 64 * 8 + 8, 0);
    for (size_t i = 0; i < bits.size(); ++i) if (bits[i]) out[i >> 3] |= (uint8_t)(1u << (i & 7));
    return out;
}

struct Array;
using ArrayRef = std::shared_ptr<const Array>;

struct Array {
    BufferRef values, validity;  // validity null = no nulls
    int64_t offset = 0, length = 0, null_count = 0;
    DataType dtype = DataType::Float64;
    std::shared_ptr<const std::vector<std::string>> strings;  // Utf8 columns are carried opaquely on the host
    // The buffers are HOST memory (DataFrame::from_arrow_host: views into the mapped file): the array can feed aggregates —
    // AggregateFunctions::*, a LazyFrame that ends in aggregate() — which the library then streams through HBM slab by slab
    // (rdf_pipeline over RDF_MEM_HOST, rdf_capi_stream.inc); operators that produce device columns want to_device() first.
    bool host = false;

    size_t len() const { return (size_t)length; }
    DataType data_type() const { return dtype; }

    rdf_array view() const {
        if (dtype == DataType::Utf8) throw DataFrameError(DataFrameError::ComputeError, "Utf8 arrays are not on the compute path");
        rdf_array a;
        a.values = values ? values->data() : nullptr;
        a.validity = validity ? (const uint8_t*)validity->data() : nullptr;
        a.offset = offset; a.length = length; a.null_count = null_count; a.dtype = (int32_t)dtype; a.mem = host ? RDF_MEM_HOST : RDF_MEM_DEVICE;
        return a;
    }
    rdf_array view_unknown_nulls() const { rdf_array a = view(); a.null_count = -1; return a; }
    rdf_out out_view(int64_t capacity) const {
        rdf_out o;
        o.values = values ? values->data() : nullptr;
        o.validity = validity ? (uint8_t*)validity->data() : nullptr;
        o.capacity = capacity; o.length = 0; o.null_count = 0; o.dtype = (int32_t)dtype; o.mem = host ? RDF_MEM_HOST : RDF_MEM_DEVICE;
        return o;
    }
    // freshly allocated output array (capacity rounded up to 64 elements as the ABI asks)
    // `on_host`: the output of an operator over a host-resident frame (DataFrame::from_arrow_host) lives in page-locked host
    // memory — the library streams such calls and its copies back are plain DMA into these buffers
    static std::shared_ptr<Array> make_out(DataType t, int64_t capacity, bool with_validity, bool on_host = false) {
        auto a = std::make_shared<Array>();
        const int64_t cap = (capacity + 63) / 64 * 64;
        a->dtype = t;
        a->host = on_host;
        auto buf = [&](int64_t bytes) -> BufferRef {
            if (!on_host) return std::make_shared<DeviceBuffer>(bytes);
            auto pin = std::make_shared<PinnedBuffer>(bytes);
            return std::make_shared<DeviceBuffer>(pin->data(), bytes, pin);
        };
        a->values = buf(t == DataType::Boolean ? cap / 8 + 8 : cap * type_size(t) + 8);
        if (with_validity) a->validity = buf(cap / 8 + 8);
        return a;
    }
    // bytes of this array's buffers on the caller's side of the link: a copy off the device, or straight out of host memory
    void fetch(void* dst, const void* src, int64_t bytes) const {
        if (bytes <= 0) return;
        if (host) std::memcpy(dst, src, (size_t)bytes);
        else check(rdf_copy_d2h(dst, src, bytes));
    }
    template <class T>
    static ArrayRef from_vec(const std::vector<T>& v, const std::vector<bool>* valid = nullptr) {
        auto a = std::make_shared<Array>();
        a->dtype = TypeOf<T>::value;
        a->length = (int64_t)v.size();
        a->values = std::make_shared<DeviceBuffer>((int64_t)(v.size() * sizeof(T)) + 8);
        if (!v.empty()) check(rdf_copy_h2d(a->values->data(), v.data(), (int64_t)(v.size() * sizeof(T))));
        if (valid) {
            const auto bits = pack_bits(*valid);
            a->validity = std::make_shared<DeviceBuffer>((int64_t)bits.size());
            check(rdf_copy_h2d(a->validity->data(), bits.data(), (int64_t)bits.size()));
            for (bool b : *valid) a->null_count += !b;
        }
        return a;
    }
    static ArrayRef from_bools(const std::vector<bool>& v, const std::vector<bool>* valid = nullptr) {
        auto a = std::make_shared<Array>();
        a->dtype = DataType::Boolean;
        a->length = (int64_t)v.size();
        const auto bits = pack_bits(v);
        a->values = std::make_shared<DeviceBuffer>((int64_t)bits.size());
        check(rdf_copy_h2d(a->values->data(), bits.data(), (int64_t)bits.size()));
        if (valid) {
            const auto vb = pack_bits(*valid);
            a->validity = std::make_shared<DeviceBuffer>((int64_t)vb.size());
            check(rdf_copy_h2d(a->validity->data(), vb.data(), (int64_t)vb.size()));
            for (bool b : *valid) a->null_count += !b;
        }
        return a;
    }
    static ArrayRef from_strings(std::vector<std::string> s) {
        auto a = std::make_shared<Array>();
        a->dtype = DataType::Utf8;
        a->length = (int64_t)s.size();
        a->strings = std::make_shared<const std::vector<std::string>>(std::move(s));
        return a;
    }
    // Array::slice: zero-copy (src/table.rs:88)
    ArrayRef slice(int64_t off, int64_t len) const {
        auto a = std::make_shared<Array>(*this);
        if (off > length) off = length;
        if (len > length - off) len = length - off;
        a->offset = offset + off;
        a->length = len;
        a->null_count = validity ? -1 : 0;
        return a;
    }
    std::vector<bool> bits_to_host(const BufferRef& buf) const {
        std::vector<bool> out((size_t)length, true);
        if (!buf || length == 0) return out;
        const int64_t b0 = offset >> 3, b1 = (offset + length + 7) >> 3;
        std::vector<uint8_t> raw((size_t)(b1 - b0));
        fetch(raw.data(), (const uint8_t*)buf->data() + b0, b1 - b0);
        for (int64_t i = 0; i < length; ++i) { const int64_t k = (offset & 7) + i; out[(size_t)i] = (raw[(size_t)(k >> 3)] >> (k & 7)) & 1; }
        return out;
    }
    std::vector<bool> valid_to_host() const { return bits_to_host(validity); }
    std::vector<bool> bools_to_host() const { return bits_to_host(values); }
    template <class T>
    std::vector<T> values_to_host() const {
        if (TypeOf<T>::value != dtype) throw DataFrameError(DataFrameError::ComputeError, "values_to_host: type mismatch");
        std::vector<T> out((size_t)length);
        if (length) fetch(out.data(), (const T*)values->data() + offset, length * (int64_t)sizeof(T));
        return out;
    }
    bool is_null(int64_t i) const { return validity && !valid_to_host()[(size_t)i]; }
    template <class T> T value(int64_t i) const {
        T v;
        fetch(&v, (const T*)values->data() + offset + i, (int64_t)sizeof(T));
        return v;
    }
    int64_t count_nulls() const {  // resolves an unknown null_count on the device
        if (!validity) return 0;
        if (null_count >= 0) return null_count;
        rdf_array v = view();
        int64_t c = 0; int32_t some = 0;
        check(rdf_count(&v, 1, &c, &some));
        return length - c;
    }
};

inline std::vector<ArrayRef> cast_arrays(const std::vector<ArrayRef>& arr, DataType to);   // arrow::compute::cast per chunk, below

// ------------------------------------------------------------------------------------------------
// ChunkedArray (src/table.rs:13-112)

class ChunkedArray {
  public:
    ChunkedArray() = default;
    static ChunkedArray from_arrays(std::vector<ArrayRef> arrays) {
        if (arrays.empty()) throw DataFrameError(DataFrameError::ComputeError, "ChunkedArray needs at least 1 array");  // assert!, :25
        ChunkedArray c;
        for (auto& a : arrays) {
            if (a->dtype != arrays[0]->dtype) throw DataFrameError(DataFrameError::ComputeError, "arrays of a ChunkedArray share one data type");
            c.num_rows_ += a->length;
        }
        c.chunks_ = std::move(arrays);
        return c;
    }
    int64_t num_rows() const { return num_rows_; }
    int64_t null_count() const { int64_t n = 0; for (auto& a : chunks_) n += a->count_nulls(); return n; }
    size_t num_chunks() const { return chunks_.size(); }
    const ArrayRef& chunk(size_t i) const { return chunks_[i]; }
    const std::vector<ArrayRef>& chunks() const { return chunks_; }
    DataType data_type() const { return chunks_[0]->dtype; }
    std::vector<int64_t> chunk_counts() const { std::vector<int64_t> v; for (auto& a : chunks_) v.push_back(a->length); return v; }

    // zero-copy slice, src/table.rs:77-95 (same chunk walk)
    ChunkedArray slice(int64_t offset, std::optional<int64_t> length = std::nullopt) const {
        int64_t len = std::min(length.value_or(INT64_MAX), num_rows_);
        size_t cur = 0;
        std::vector<ArrayRef> out;
        while (cur < chunks_.size() && offset >= chunks_[cur]->length) { offset -= chunks_[cur]->length; ++cur; }
        while (cur < chunks_.size() && len > 0) {
            out.push_back(chunks_[cur]->slice(offset, len));
            len -= std::min(len, chunks_[cur]->length - offset);
            offset = 0;
            ++cur;
        }
        if (out.empty()) out.push_back(chunks_[0]->slice(0, 0));
        return from_arrays(std::move(out));
    }

    std::vector<rdf_array> views() const { std::vector<rdf_array> v; for (auto& a : chunks_) v.push_back(a->view()); return v; }

    // src/table.rs:97-107: zip(chunks, condition chunks) -> arrow::compute::filter
    ChunkedArray filter(const ChunkedArray& condition) const {
        if (condition.num_chunks() != num_chunks()) throw DataFrameError(DataFrameError::ComputeError, "filter: chunk counts differ");
        const auto cv = views(), mv = condition.views();
        std::vector<int64_t> counts(cv.size());
        check(rdf_filter_count(mv.data(), (int64_t)mv.size(), counts.data()));
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        for (size_t i = 0; i < cv.size(); ++i) {
            outs.push_back(Array::make_out(data_type(), counts[i], chunks_[i]->validity != nullptr));
            ov.push_back(outs.back()->out_view(counts[i]));
        }
        check(rdf_filter(cv.data(), mv.data(), (int64_t)cv.size(), ov.data()));
        std::vector<ArrayRef> res;
        for (size_t i = 0; i < outs.size(); ++i) { outs[i]->length = ov[i].length; outs[i]->null_count = ov[i].null_count; res.push_back(outs[i]); }
        return from_arrays(std::move(res));
    }

  private:
    std::vector<ArrayRef> chunks_;
    int64_t num_rows_ = 0;
};

// ------------------------------------------------------------------------------------------------
// Column (src/table.rs:134-344)

class Column {
  public:
    Column() = default;
    Column(ChunkedArray data, Field field) : data_(std::move(data)), field_(std::move(field)) {}
    static Column from_arrays(std::vector<ArrayRef> arrays, Field field) {
        for (auto& a : arrays)
            if (a->dtype != field.data_type) throw DataFrameError(DataFrameError::ComputeError, "array type differs from the field's");
        return Column(ChunkedArray::from_arrays(std::move(arrays)), std::move(field));
    }
    const std::string& name() const { return field_.name; }
    DataType data_type() const { return field_.data_type; }
    const ChunkedArray& data() const { return data_; }
    const Field& field() const { return field_; }
    int64_t num_rows() const { return data_.num_rows(); }
    int64_t null_count() const { return data_.null_count(); }
    Column slice(int64_t offset, std::optional<int64_t> length = std::nullopt) const {
        if (data_type() == DataType::Utf8) {  // opaque host column: slice the strings
            std::vector<ArrayRef> out;
            int64_t len = std::min(length.value_or(INT64_MAX), num_rows());
            for (auto& c : data_.chunks()) {
                if (offset >= c->length) { offset -= c->length; continue; }
                if (len <= 0) break;
                const int64_t n = std::min(len, c->length - offset);
                out.push_back(Array::from_strings(std::vector<std::string>(c->strings->begin() + offset, c->strings->begin() + offset + n)));
                len -= n; offset = 0;
            }
            if (out.empty()) out.push_back(Array::from_strings({}));
            return Column(ChunkedArray::from_arrays(out), field_);
        }
        return Column(data_.slice(offset, length), field_);
    }
    Column filter(const Column& condition) const { return Column(data_.filter(condition.data()), field_); }  // :213-215
    Column renamed(const std::string& n) const { Column c = *this; c.field_.name = n; return c; }

    // src/table.rs:218-241: gather over the concatenation of the chunks; the result is ONE chunk whatever
    // chunk_size says.  indices: UInt32 (drop-in) or UInt64.
    Column take(const ArrayRef& indices, size_t chunk_size) const {
        if (data_type() == DataType::Utf8) {
            // text columns are carried on the host: gather there (a NULL index gives a NULL, i.e. an empty slot marked invalid)
            std::vector<const std::string*> flat;
            std::vector<bool> flat_valid;
            for (auto& c : data_.chunks()) {
                const std::vector<bool> v = c->valid_to_host();
                for (int64_t r = 0; r < c->length; ++r) { flat.push_back(&(*c->strings)[(size_t)(c->offset + r)]); flat_valid.push_back(v[(size_t)r]); }
            }
            const std::vector<bool> iv_valid = indices->valid_to_host();
            std::vector<uint64_t> idx;
            if (indices->dtype == DataType::UInt32) { for (uint32_t x : indices->values_to_host<uint32_t>()) idx.push_back(x); }
            else if (indices->dtype == DataType::UInt64) { for (uint64_t x : indices->values_to_host<uint64_t>()) idx.push_back(x); }
            else throw DataFrameError(DataFrameError::ComputeError, "take: indices must be UInt32 / UInt64");
            std::vector<std::string> out(idx.size());
            std::vector<bool> valid(idx.size(), true);
            bool any_null = false;
            for (size_t j = 0; j < idx.size(); ++j) {
                if (!iv_valid[j]) { valid[j] = false; any_null = true; continue; }
                if (idx[j] >= flat.size()) throw DataFrameError(DataFrameError::ComputeError, "take: index out of bounds");
                out[j] = *flat[(size_t)idx[j]];
                if (!flat_valid[(size_t)idx[j]]) { valid[j] = false; any_null = true; }
            }
            auto a = std::const_pointer_cast<Array>(Array::from_strings(std::move(out)));
            if (any_null) {
                const auto bits = pack_bits(valid);
                a->validity = std::make_shared<DeviceBuffer>((int64_t)bits.size());
                check(rdf_copy_h2d(a->validity->data(), bits.data(), (int64_t)bits.size()));
                for (bool b : valid) a->null_count += !b;
            }
            return Column(ChunkedArray::from_arrays({ArrayRef(a)}), field_);
        }
        if (data_type() == DataType::Boolean) {   // Boolean columns travel through the gather as UInt8 (cast there and back on the device)
            const Column wide(ChunkedArray::from_arrays(cast_arrays(data_.chunks(), DataType::UInt8)), Field{field_.name, DataType::UInt8, field_.nullable});
            const Column taken = wide.take(indices, chunk_size);
            return Column(ChunkedArray::from_arrays(cast_arrays(taken.data().chunks(), DataType::Boolean)), field_);
        }
        const auto cv = data_.views();
        const rdf_array iv = indices->view();
        bool nullable = indices->validity != nullptr;
        for (auto& c : data_.chunks()) nullable |= c->validity != nullptr;
        auto out = Array::make_out(data_type(), indices->length, nullable);
        rdf_out ov = out->out_view(indices->length);
        check(rdf_take(cv.data(), (int64_t)cv.size(), &iv, &ov));
        out->length = ov.length;
        out->null_count = ov.null_count;
        return Column(ChunkedArray::from_arrays({out}), field_);
    }
  private:
    ChunkedArray data_;
    Field field_;
};

// ------------------------------------------------------------------------------------------------
// Scalar / BooleanFilter (src/expression.rs:718-870)

struct Scalar {
    enum Kind { Null, Int32, Int64, Float32, Float64, Boolean } kind = Null;
    double f = 0;
    int64_t i = 0;
    Scalar() = default;
    Scalar(int32_t v) : kind(Int32), i(v) {}
    Scalar(int64_t v) : kind(Int64), i(v) {}
    Scalar(float v) : kind(Float32), f(v) {}
    Scalar(double v) : kind(Float64), f(v) {}
    Scalar(bool v) : kind(Boolean), i(v) {}
    int32_t rdf_type() const {
        switch (kind) { case Int32: return RDF_I32; case Int64: return RDF_I64; case Float32: return RDF_F32;
                        case Float64: return RDF_F64; case Boolean: return RDF_BOOL; default: return RDF_NULLTYPE; }
    }
};

struct BooleanFilter;
using FilterRef = std::shared_ptr<const BooleanFilter>;
struct BooleanFilter {
    enum Kind { InputScalar, InputColumn, Not, And, Or, Gt, Ge, Eq, Ne, Lt, Le } kind;
    Scalar scalar_v;
    std::string column_name;
    FilterRef l, r;
    static FilterRef scalar(Scalar s) { auto f = std::make_shared<BooleanFilter>(); f->kind = InputScalar; f->scalar_v = s; return f; }
    static FilterRef column(const std::string& name) { auto f = std::make_shared<BooleanFilter>(); f->kind = InputColumn; f->column_name = name; return f; }
    static FilterRef make(Kind k, FilterRef a, FilterRef b = nullptr) { auto f = std::make_shared<BooleanFilter>(); f->kind = k; f->l = std::move(a); f->r = std::move(b); return f; }
    static FilterRef gt(FilterRef a, FilterRef b) { return make(Gt, a, b); }
    static FilterRef ge(FilterRef a, FilterRef b) { return make(Ge, a, b); }
    static FilterRef eq(FilterRef a, FilterRef b) { return make(Eq, a, b); }
    static FilterRef ne(FilterRef a, FilterRef b) { return make(Ne, a, b); }
    static FilterRef lt(FilterRef a, FilterRef b) { return make(Lt, a, b); }
    static FilterRef le(FilterRef a, FilterRef b) { return make(Le, a, b); }
    static FilterRef and_(FilterRef a, FilterRef b) { return make(And, a, b); }
    static FilterRef or_(FilterRef a, FilterRef b) { return make(Or, a, b); }
    static FilterRef not_(FilterRef a) { return make(Not, a); }
};

// ------------------------------------------------------------------------------------------------
// expression trees over named columns, lowered to rdf_expr_node arrays

struct Expr;
using ExprRef = std::shared_ptr<const Expr>;
struct Expr {
    enum Kind { Col, Lit, Op } kind = Col;
    std::string column;  // Col
    Scalar lit;          // Lit
    int32_t lit_type = RDF_F64;
    int32_t op = 0;      // Op (rdf_op)
    int32_t cast_to = 0;
    ExprRef l, r;
    static ExprRef col(const std::string& n) { auto e = std::make_shared<Expr>(); e->kind = Col; e->column = n; return e; }
    static ExprRef literal(Scalar s, int32_t as_type) { auto e = std::make_shared<Expr>(); e->kind = Lit; e->lit = s; e->lit_type = as_type; return e; }
    static ExprRef make(int32_t op, ExprRef a, ExprRef b = nullptr, int32_t cast_to = 0) {
        auto e = std::make_shared<Expr>(); e->kind = Op; e->op = op; e->l = std::move(a); e->r = std::move(b); e->cast_to = cast_to; return e;
    }
    bool has_divide() const { return kind == Op && (op == RDF_OP_DIV || (l && l->has_divide()) || (r && r->has_divide())); }
};

inline ExprRef filter_to_expr(const FilterRef& f, const std::function<ExprRef(const std::string&)>& resolve) {
    switch (f->kind) {
        case BooleanFilter::InputScalar: return Expr::literal(f->scalar_v, f->scalar_v.rdf_type());
        case BooleanFilter::InputColumn: return resolve(f->column_name);
        case BooleanFilter::Not: return Expr::make(RDF_OP_NOT, filter_to_expr(f->l, resolve));
        default: break;
    }
    static const std::map<int, int> ops = {{BooleanFilter::And, RDF_OP_AND}, {BooleanFilter::Or, RDF_OP_OR}, {BooleanFilter::Gt, RDF_OP_GT},
                                            {BooleanFilter::Ge, RDF_OP_GE}, {BooleanFilter::Eq, RDF_OP_EQ}, {BooleanFilter::Ne, RDF_OP_NE},
                                            {BooleanFilter::Lt, RDF_OP_LT}, {BooleanFilter::Le, RDF_OP_LE}};
    return Expr::make(ops.at(f->kind), filter_to_expr(f->l, resolve), filter_to_expr(f->r, resolve));
}

// Lowered program: node array + the distinct columns it reads (in first-use order).
struct Lowered {
    std::vector<rdf_expr_node> nodes;
    std::vector<std::string> columns;
    int add(const ExprRef& e) {
        rdf_expr_node n;
        std::memset(&n, 0, sizeof n);
        n.lhs = n.rhs = -1;
        if (e->kind == Expr::Col) {
            int idx = -1;
            for (size_t i = 0; i < columns.size(); ++i) if (columns[i] == e->column) idx = (int)i;
            if (idx < 0) { idx = (int)columns.size(); columns.push_back(e->column); }
            n.kind = RDF_NODE_COLUMN; n.column = idx;
        } else if (e->kind == Expr::Lit) {
            n.kind = RDF_NODE_SCALAR; n.dtype = e->lit.kind == Scalar::Null ? RDF_NULLTYPE : e->lit_type;
            const bool lit_is_float = e->lit.kind == Scalar::Float32 || e->lit.kind == Scalar::Float64;
            n.f64 = lit_is_float ? e->lit.f : (double)e->lit.i;
            n.i64 = lit_is_float ? (int64_t)e->lit.f : e->lit.i;
        } else {
            const int l = add(e->l);
            const int r = e->r ? add(e->r) : -1;
            n.kind = RDF_NODE_OP; n.op = e->op; n.lhs = l; n.rhs = r; n.dtype = e->cast_to;
        }
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }
};

// ------------------------------------------------------------------------------------------------
// plan types (src/expression.rs:286-712) and operation builders (src/operation/scalar.rs)

// ------------------------------------------------------------------------------------------------
// CSV text -> typed column buffers, on the host (DataFrame::from_csv, plan::Reader::get_dataset): arrow::csv::Reader with an
// inferred schema (src/dataframe.rs:349-389).  The file is read ONCE into memory and indexed by record; the types are
// inferred and the cells parsed by worker threads over ranges of records (std::from_chars: exact, locale-free), numbers land
// straight in the caller's (page-locked) column buffers.  A column whose non-empty cells all parse as integers is Int64, as
// numbers Float64, as true / false Boolean, anything else Utf8; an empty cell is a NULL.  Quotes toggle quoting and are
// dropped wherever they stand, a delimiter inside quotes belongs to the cell, '\r' is dropped, zero-length lines are skipped,
// short records are padded with NULLs and long ones cut — what the line-by-line reader this replaces did (0.055 GB/s: one
// std::string per cell, every cell parsed three times with strtod).
namespace csv {

struct Text {
    std::string buf;                                   // the file image
    std::vector<std::pair<size_t, size_t>> records;    // [begin, end) of every data record
    std::vector<std::string> header;
    char delimiter = ',';
};
struct Cell { const char* b; const char* e; bool dirty; };   // dirty: holds quotes or '\r' that are not part of the value

// the cells of one record, at most `ncols` of them (missing ones come back empty)
inline void split(const char* b, const char* e, char delimiter, size_t ncols, Cell* out) {
    size_t c = 0;
    bool q = false, dirty = false;
    const char* start = b;
    for (const char* p = b; p < e && c < ncols; ++p) {
        if (*p == '"') { q = !q; dirty = true; }
        else if (*p == '\r') dirty = true;
        else if (*p == delimiter && !q) { out[c++] = Cell{start, p, dirty}; start = p + 1; dirty = false; }
    }
    if (c < ncols) {
        const char* end = e;
        bool d = false;
        for (const char* p = start; p < e; ++p) {
            if (*p == '"') { q = !q; d = true; } else if (*p == '\r') d = true; else if (*p == delimiter && !q) { end = p; break; }
        }
        out[c++] = Cell{start, end, d};
    }
    for (; c < ncols; ++c) out[c] = Cell{e, e, false};
}
inline std::string cell_text(const Cell& c) {
    std::string t;
    t.reserve((size_t)(c.e - c.b));
    for (const char* p = c.b; p < c.e; ++p) if (*p != '"' && *p != '\r') t += *p;
    return t;
}
inline const char* skip_lead(const char* b, const char* e, bool* ok) {   // strtoll / strtod take leading blanks and one '+'
    while (b < e && (*b == ' ' || *b == '\t')) ++b;
    if (b < e && *b == '+') { ++b; if (b < e && (*b == '+' || *b == '-')) *ok = false; }
    if (b == e) *ok = false;
    return b;
}
inline bool parse_i64(const char* b, const char* e, int64_t& v) {
    bool ok = true;
    b = skip_lead(b, e, &ok);
    if (!ok) return false;
    const auto r = std::from_chars(b, e, v, 10);
    return r.ec == std::errc() && r.ptr == e;
}
inline bool parse_f64(const char* b, const char* e, double& v) {
    bool ok = true;
    b = skip_lead(b, e, &ok);
    if (!ok) return false;
    const auto r = std::from_chars(b, e, v);
    if (r.ec == std::errc() && r.ptr == e) return true;
    const std::string t(b, e);                          // what strtod takes and from_chars does not (hex floats, overflow to inf)
    char* end = nullptr;
    v = std::strtod(t.c_str(), &end);
    return end != t.c_str() && !*end;
}
inline bool is_bool(const char* b, const char* e, bool* value = nullptr) {
    const size_t n = (size_t)(e - b);
    auto eq = [&](const char* w) { return std::strlen(w) == n && std::memcmp(b, w, n) == 0; };
    if (eq("true") || eq("True") || eq("TRUE")) { if (value) *value = true; return true; }
    if (eq("false") || eq("False") || eq("FALSE")) { if (value) *value = false; return true; }
    return false;
}

inline Text load(const std::string& path, bool has_headers = true, char delimiter = ',', std::optional<size_t> max_records = std::nullopt) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw DataFrameError(DataFrameError::IoError, "cannot open " + path);
    Text t;
    t.delimiter = delimiter;
    const std::streamoff size = f.tellg();
    f.seekg(0);
    t.buf.resize((size_t)size);
    if (size > 0 && !f.read(&t.buf[0], size)) throw DataFrameError(DataFrameError::IoError, "cannot read " + path);
    const char* base = t.buf.data();
    const char* end = base + t.buf.size();
    // the header: the first line that is not empty (or, without a header row, just its number of cells)
    const char* p = base;
    while (p < end) {
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (le > p) {
            size_t ncols = 1;
            { bool q = false; for (const char* c = p; c < le; ++c) { if (*c == '"') q = !q; else if (*c == delimiter && !q) ++ncols; } }
            std::vector<Cell> cells(ncols);
            split(p, le, delimiter, ncols, cells.data());
            if (has_headers) { for (auto& c : cells) t.header.push_back(cell_text(c)); p = nl ? nl + 1 : end; }
            else for (size_t i = 0; i < ncols; ++i) t.header.push_back("column_" + std::to_string(i + 1));   // arrow's csv reader names
            break;
        }
        p = nl ? nl + 1 : end;
    }
    // the records: [begin, end) of every line that is not empty.  A long file is cut into byte ranges, one thread each; a thread
    // takes the lines that START in its range (it finds its first line start behind the first '\n' at or after the range's
    // begin - 1) and follows its last line past the range's end.
    const char* body = p;
    auto scan = [&](const char* from, const char* upto, std::vector<std::pair<size_t, size_t>>& out, size_t stop_after) {
        const char* q = from;
        while (q < upto && out.size() < stop_after) {
            const char* nl = (const char*)std::memchr(q, '\n', (size_t)(end - q));
            const char* le = nl ? nl : end;
            if (le > q) out.emplace_back((size_t)(q - base), (size_t)(le - base));
            q = nl ? nl + 1 : end;
        }
    };
    const size_t limit = max_records ? *max_records : (size_t)-1;
    const size_t bytes = (size_t)(end - body);
    const int T = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16, bytes / ((size_t)4 << 20) + 1});
    if (T <= 1 || max_records) scan(body, end, t.records, limit);
    else {
        std::vector<std::vector<std::pair<size_t, size_t>>> part((size_t)T);
        std::vector<std::thread> pool;
        for (int w = 0; w < T; ++w)
            pool.emplace_back([&, w]() {
                const char* b = body + bytes * (size_t)w / (size_t)T;
                const char* e = body + bytes * (size_t)(w + 1) / (size_t)T;
                if (w > 0) {   // the first line start at or behind b: one past the first '\n' at or behind b - 1
                    const char* nl = (const char*)std::memchr(b - 1, '\n', (size_t)(end - (b - 1)));
                    b = nl ? nl + 1 : end;
                }
                scan(b, e, part[(size_t)w], (size_t)-1);
            });
        for (auto& th : pool) th.join();
        size_t total = 0;
        for (auto& v : part) total += v.size();
        t.records.reserve(total);
        for (auto& v : part) t.records.insert(t.records.end(), v.begin(), v.end());
    }
    return t;
}

// worker threads over ranges of records that start on multiples of 64 (validity bytes and words are never shared)
template <class F>
inline void for_ranges(size_t n, int threads, F&& body) {
    int T = threads > 0 ? threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    const size_t per = ((n + (size_t)T - 1) / (size_t)T + 63) / 64 * 64;
    if (n < 32768 || T == 1) { body(0, (size_t)0, n); return; }
    std::vector<std::thread> pool;
    std::vector<std::exception_ptr> errs((size_t)T);
    int t = 0;
    for (size_t r0 = 0; r0 < n; r0 += per, ++t)
        pool.emplace_back([&, t, r0]() { try { body(t, r0, std::min(n, r0 + per)); } catch (...) { errs[(size_t)t] = std::current_exception(); } });
    for (auto& th : pool) th.join();
    for (auto& e : errs) if (e) std::rethrow_exception(e);
}
inline int range_count(size_t n, int threads) {
    int T = threads > 0 ? threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    if (n < 32768 || T == 1) return 1;
    const size_t per = ((n + (size_t)T - 1) / (size_t)T + 63) / 64 * 64;
    return (int)((n + per - 1) / per);
}

struct Inferred { DataType dtype = DataType::Utf8; bool any_null = false, any = false; };   // any: some cell of the column is not empty
// the types of the columns `cols` (indices into the header), from the first `limit` records
inline std::vector<Inferred> infer(const Text& t, const std::vector<size_t>& cols, int threads = 0, size_t limit = (size_t)-1) {
    struct Flags { bool all_int = true, all_num = true, all_bool = true, any = false, nulls = false; };
    const size_t nh = t.header.size(), n = std::min(t.records.size(), limit);
    const int nr = range_count(n, threads);
    std::vector<std::vector<Flags>> part((size_t)nr, std::vector<Flags>(cols.size()));
    for_ranges(n, threads, [&](int w, size_t r0, size_t r1) {
        std::vector<Cell> cells(nh);
        std::vector<Flags>& fl = part[(size_t)w];
        for (size_t r = r0; r < r1; ++r) {
            split(t.buf.data() + t.records[r].first, t.buf.data() + t.records[r].second, t.delimiter, nh, cells.data());
            for (size_t k = 0; k < cols.size(); ++k) {
                Cell c = cells[cols[k]];
                std::string clean;
                if (c.dirty) { clean = cell_text(c); c = Cell{clean.data(), clean.data() + clean.size(), false}; }
                Flags& f = fl[k];
                if (c.b == c.e) { f.nulls = true; continue; }
                f.any = true;
                int64_t iv; double dv;
                if (f.all_int && !parse_i64(c.b, c.e, iv)) f.all_int = false;
                if (!f.all_int && f.all_num && !parse_f64(c.b, c.e, dv)) f.all_num = false;
                if (f.all_bool && !is_bool(c.b, c.e)) f.all_bool = false;
            }
        }
    });
    std::vector<Inferred> out(cols.size());
    for (size_t k = 0; k < cols.size(); ++k) {
        Flags m;
        for (auto& p : part) { m.all_int &= p[k].all_int; m.all_num &= p[k].all_num; m.all_bool &= p[k].all_bool; m.any |= p[k].any; m.nulls |= p[k].nulls; }
        // (a range that met no cell that breaks "integer" never tried "number": an integer is a number)
        out[k].any_null = m.nulls;
        out[k].any = m.any;
        out[k].dtype = !m.any ? DataType::Utf8 : m.all_int ? DataType::Int64 : m.all_num ? DataType::Float64 : m.all_bool ? DataType::Boolean : DataType::Utf8;
    }
    return out;
}

struct Filled { int64_t nulls = 0; std::vector<std::string> strings; };   // strings: the cells of a Utf8 column
// Parse column cols[k] of every record into values[k] (8 bytes per row; Boolean: one bit per row) and validity[k] (one bit per
// row, 1 = valid); both zeroed here.  Utf8 columns (values[k] == nullptr) come back as strings.
// `violated` (optional, one flag per column): a non-empty cell did not parse as the column's type — the types came from a sample
// of the records and the caller has to infer them from all of them after all.
inline std::vector<Filled> fill(const Text& t, const std::vector<size_t>& cols, const std::vector<Inferred>& types,
                                const std::vector<uint8_t*>& values, const std::vector<uint8_t*>& validity, int threads = 0,
                                std::vector<char>* violated = nullptr) {
    const size_t nh = t.header.size(), n = t.records.size();
    std::vector<Filled> out(cols.size());
    for (size_t k = 0; k < cols.size(); ++k) {
        if (types[k].dtype == DataType::Utf8) { out[k].strings.resize(n); continue; }
        std::memset(values[k], 0, types[k].dtype == DataType::Boolean ? (n + 7) / 8 : n * 8);
        std::memset(validity[k], 0, (n + 7) / 8);
    }
    const int nr = range_count(n, threads);
    std::vector<std::vector<int64_t>> nulls((size_t)nr, std::vector<int64_t>(cols.size(), 0));
    for_ranges(n, threads, [&](int w, size_t r0, size_t r1) {
        std::vector<Cell> cells(nh);
        for (size_t r = r0; r < r1; ++r) {
            split(t.buf.data() + t.records[r].first, t.buf.data() + t.records[r].second, t.delimiter, nh, cells.data());
            for (size_t k = 0; k < cols.size(); ++k) {
                Cell c = cells[cols[k]];
                std::string clean;
                if (c.dirty) { clean = cell_text(c); c = Cell{clean.data(), clean.data() + clean.size(), false}; }
                const DataType dt = types[k].dtype;
                if (dt == DataType::Utf8) { out[k].strings[r].assign(c.b, c.e); continue; }
                if (c.b == c.e) { ++nulls[(size_t)w][k]; continue; }
                validity[k][r >> 3] |= (uint8_t)(1u << (r & 7));
                bool ok;
                if (dt == DataType::Int64) { int64_t v = 0; ok = parse_i64(c.b, c.e, v); std::memcpy(values[k] + 8 * r, &v, 8); }
                else if (dt == DataType::Float64) { double v = 0; ok = parse_f64(c.b, c.e, v); std::memcpy(values[k] + 8 * r, &v, 8); }
                else { bool v = false; ok = is_bool(c.b, c.e, &v); if (v) values[k][r >> 3] |= (uint8_t)(1u << (r & 7)); }
                if (!ok && violated) (*violated)[k] = 1;     // (a racing store of the same value from several threads)
            }
        }
    });
    for (size_t k = 0; k < cols.size(); ++k) for (auto& p : nulls) out[k].nulls += p[k];
    return out;
}

}  // namespace csv

namespace plan {

struct Column {  // expression::Column {name, column_type: Scalar(dtype)}
    std::string name;
    DataType data_type;
    std::string debug() const { return "Column { name: \"" + name + "\", column_type: Scalar(" + type_name(data_type) + ") }"; }
};
struct Dataset {
    std::string name;
    std::vector<Column> columns;
    std::optional<std::pair<size_t, Column>> get_column(const std::string& n) const {
        for (size_t i = 0; i < columns.size(); ++i) if (columns[i].name == n) return std::make_pair(i, columns[i]);
        return std::nullopt;
    }
    Dataset append_column(const Column& c) const {  // :95-112: replace in place or push
        Dataset d = *this;
        for (auto& e : d.columns) if (e.name == c.name) { e = c; return d; }
        d.columns.push_back(c);
        return d;
    }
};
enum class ScalarFunction { Add, Subtract, Divide, Multiply, Abs, Sine, Cosine, Tangent, Cotangent, Secant, Cosecant };
inline const char* scalar_function_name(ScalarFunction f) {
    static const char* n[] = {"Add", "Subtract", "Divide", "Multiply", "Abs", "Sine", "Cosine", "Tangent", "Cotangent", "Secant", "Cosecant"};
    return n[(int)f];
}
enum class AggregateFunction { Avg, Count, CountDistinct, First, Kurtosis, Last, Max, Min, Skewness, StdDev, Sum, SumDistinct, Variance };
struct Function {
    enum Kind { Scalar, Cast, Rename, Filter } kind = Scalar;
    ScalarFunction scalar = ScalarFunction::Add;
    FilterRef filter;
    static Function Scalar_(ScalarFunction f) { Function x; x.kind = Scalar; x.scalar = f; return x; }
    static Function Cast_() { Function x; x.kind = Cast; return x; }
    static Function Rename_() { Function x; x.kind = Rename; return x; }
    static Function Filter_(FilterRef f) { Function x; x.kind = Filter; x.filter = std::move(f); return x; }
    std::string debug() const {
        switch (kind) { case Cast: return "Cast"; case Rename: return "Rename"; case Filter: return "Filter(..)";
                        default: return std::string("Scalar(") + scalar_function_name(scalar) + ")"; }
    }
};
struct Calculation {
    std::string name;
    std::vector<Column> inputs;
    Column output;
    Function function;
    std::string debug() const {  // mirrors #[derive(Debug)] so the reference's plan-text test can be restated
        std::string s = "Calculation { name: \"" + name + "\", inputs: [";
        for (size_t i = 0; i < inputs.size(); ++i) s += (i ? ", " : "") + inputs[i].debug();
        return s + "], output: " + output.debug() + ", function: " + function.debug() + " }";
    }
};
inline std::string debug(const std::vector<Calculation>& v) {
    std::string s = "[";
    for (size_t i = 0; i < v.size(); ++i) s += (i ? ", " : "") + v[i].debug();
    return s + "]";
}
struct Aggregation { AggregateFunction function; std::vector<std::string> columns; };

// Dataset::try_aggregate (src/expression.rs:114-221): the planned output of GroupAggregate — the grouping columns, then one
// column per (aggregation, input column) named "avg(x)" / "sum(x)" / "max(x)" / "min(x)" / "count(x)" / "count_distinct(x)"
// / "first(x)" / "last(x)", typed like the input except the counts (UInt32); the remaining functions are
// "Aggregation not yet supported".  (The reference PLANS avg with the input's type; evaluated, an average is Float64 —
// AggregateFunctions::avg, src/functions/aggregate.rs:32.)
inline Dataset try_aggregate(const Dataset& self, const std::vector<std::string>& groups, const std::vector<Aggregation>& aggr) {
    Dataset out;
    out.name = "aggregated_dataset";
    for (auto& g : groups) {
        const auto c = self.get_column(g);
        if (!c) throw DataFrameError(DataFrameError::ComputeError, "Grouping column \"" + g + "\" does not exist");
        out.columns.push_back(c->second);
    }
    for (auto& a : aggr)
        for (auto& name : a.columns) {
            const auto c = self.get_column(name);
            if (!c) throw DataFrameError(DataFrameError::ComputeError, "Aggregating column \"" + name + "\" does not exist");
            const DataType t = c->second.data_type;
            switch (a.function) {
                case AggregateFunction::Avg: out.columns.push_back(Column{"avg(" + name + ")", t}); break;
                case AggregateFunction::Sum: out.columns.push_back(Column{"sum(" + name + ")", t}); break;
                case AggregateFunction::Max: out.columns.push_back(Column{"max(" + name + ")", t}); break;
                case AggregateFunction::Min: out.columns.push_back(Column{"min(" + name + ")", t}); break;
                case AggregateFunction::Count: out.columns.push_back(Column{"count(" + name + ")", DataType::UInt32}); break;
                case AggregateFunction::CountDistinct: out.columns.push_back(Column{"count_distinct(" + name + ")", DataType::UInt32}); break;
                case AggregateFunction::First: out.columns.push_back(Column{"first(" + name + ")", t}); break;
                case AggregateFunction::Last: out.columns.push_back(Column{"last(" + name + ")", t}); break;
                default: throw DataFrameError(DataFrameError::ComputeError, "Aggregation not yet supported");
            }
        }
    return out;
}

// Dataset::try_join (src/expression.rs:223-285): both key columns must exist and have the same type; the output holds the
// columns of both sides, a name present on both sides as "a.<name>" / "b.<name>"
inline Dataset try_join(const Dataset& a, const Dataset& b, const std::vector<std::pair<std::string, std::string>>& on) {
    for (auto& c : on) {
        const auto ca = a.get_column(c.first), cb = b.get_column(c.second);
        if (ca && cb) { if (ca->second.data_type != cb->second.data_type) throw DataFrameError(DataFrameError::ComputeError, "Join columns must have compatible types"); }
        else if (!ca && cb) throw DataFrameError(DataFrameError::ComputeError, "Join column does not exist in table A");
        else if (ca && !cb) throw DataFrameError(DataFrameError::ComputeError, "Join column does not exist in table B");
        else throw DataFrameError(DataFrameError::ComputeError, "Join columns do not exist in tables");
    }
    Dataset out;
    out.name = "joined_dataframe";
    for (auto& c : a.columns) out.columns.push_back(b.get_column(c.name) ? Column{"a." + c.name, c.data_type} : c);
    for (auto& c : b.columns) out.columns.push_back(a.get_column(c.name) ? Column{"b." + c.name, c.data_type} : c);
    return out;
}

struct ArrowError : DataFrameError { using DataFrameError::DataFrameError; };

// CastOperation (src/operation/scalar.rs:95-137)
struct CastOperation {
    static const char* name() { return "cast"; }
    static std::vector<Calculation> transform(const std::vector<Column>& inputs, std::optional<std::string> out_name, std::optional<DataType> to_type) {
        if (inputs.size() != 1) throw DataFrameError(DataFrameError::ComputeError, "Cast operation expects 1 input");
        if (!to_type) throw DataFrameError(DataFrameError::ArrowError, "Cast requires a target output datatype");
        const Column& a = inputs[0];
        return {Calculation{name(), inputs, Column{out_name.value_or(std::string(name()) + "(" + a.name + " as datatype)"), *to_type}, Function::Cast_()}};
    }
};
// AddOperation / SubtractOperation (:18-93, :139-214) + the same shape for multiply / divide
template <ScalarFunction F>
struct BinaryOperation {
    static const char* name() {
        return F == ScalarFunction::Add ? "add" : F == ScalarFunction::Subtract ? "subtract" : F == ScalarFunction::Multiply ? "multiply" : "divide";
    }
    static std::vector<Calculation> transform(const std::vector<Column>& inputs, std::optional<std::string> out_name, std::optional<DataType>) {
        if (inputs.size() != 2) throw DataFrameError(DataFrameError::ComputeError, std::string(name()) + " operation expects 2 inputs");
        const Column &a = inputs[0], &b = inputs[1];
        const std::string oname = out_name.value_or(std::string(name()) + "(" + a.name + ", " + b.name + ")");
        if (a.data_type != b.data_type) {
            // cast b to a's type first (:47-72).  The reference's subtract emits Add here: not copied.
            auto cast = CastOperation::transform({b}, b.name, a.data_type)[0];
            return {cast, Calculation{name(), {a, cast.output}, Column{oname, a.data_type}, Function::Scalar_(F)}};
        }
        return {Calculation{name(), inputs, Column{oname, a.data_type}, Function::Scalar_(F)}};
    }
};
using AddOperation = BinaryOperation<ScalarFunction::Add>;
using SubtractOperation = BinaryOperation<ScalarFunction::Subtract>;
using MultiplyOperation = BinaryOperation<ScalarFunction::Multiply>;
using DivideOperation = BinaryOperation<ScalarFunction::Divide>;
// SinOperation (:227-318) + the same shape for cosine / tangent: integers are cast to Float64 first
template <ScalarFunction F>
struct TrigOperation {
    static const char* name() {
        return F == ScalarFunction::Sine ? "sin" : F == ScalarFunction::Cosine ? "cos" : F == ScalarFunction::Tangent ? "tan"
             : F == ScalarFunction::Cotangent ? "cot" : F == ScalarFunction::Secant ? "sec" : "csc";
    }
    static std::vector<Calculation> transform(const std::vector<Column>& inputs, std::optional<std::string> out_name, std::optional<DataType>) {
        if (inputs.size() != 1) throw DataFrameError(DataFrameError::ComputeError, "Sine operation expects 2 inputs");  // sic, :242
        const Column& a = inputs[0];
        const std::string oname = out_name.value_or(std::string(name()) + "(" + a.name + " as datatype)");
        if (is_integer(a.data_type)) {
            const Column cast_out{out_name.value_or(std::string(CastOperation::name()) + "(" + a.name + " as datatype)"), DataType::Float64};
            return {Calculation{CastOperation::name(), inputs, cast_out, Function::Cast_()},
                    Calculation{name(), {cast_out}, Column{oname, DataType::Float64}, Function::Scalar_(F)}};
        }
        if (is_float(a.data_type)) return {Calculation{name(), inputs, Column{oname, a.data_type}, Function::Scalar_(F)}};
        throw DataFrameError(DataFrameError::ComputeError, std::string("Cannot perform ") + name() + " operation from " + type_name(a.data_type) + " data type");
    }
};
using SinOperation = TrigOperation<ScalarFunction::Sine>;
using CosOperation = TrigOperation<ScalarFunction::Cosine>;
using TanOperation = TrigOperation<ScalarFunction::Tangent>;
// ScalarFunction::{Cotangent, Secant, Cosecant} (:670-672): the reference's builder panics on them (:487-489); planned
// here with the sine's shape and evaluated as RDF_OP_COT / SEC / CSC
using CotOperation = TrigOperation<ScalarFunction::Cotangent>;
using SecOperation = TrigOperation<ScalarFunction::Secant>;
using CscOperation = TrigOperation<ScalarFunction::Cosecant>;

// Reader / DataSourceType / CsvReadOptions (src/expression.rs:344-378): where a plan's rows come from.  CSV and Arrow IPC are the
// sources this mirror loads (the others are IO outside the hot path); the CSV options are what the optimiser pushes into.
struct CsvReadOptions {
    bool has_headers = true;
    std::optional<uint8_t> delimiter;
    std::optional<size_t> max_records;
    size_t batch_size = 1024;
    std::optional<std::vector<size_t>> projection;
};
struct Reader {
    enum Source { Csv, Json, Arrow, Sql, Parquet } source = Csv;
    std::string path;
    CsvReadOptions csv;
    static Reader Csv_(std::string p, CsvReadOptions o = CsvReadOptions()) { Reader r; r.source = Csv; r.path = std::move(p); r.csv = std::move(o); return r; }
    static Reader Arrow_(std::string p) { Reader r; r.source = Arrow; r.path = std::move(p); return r; }
    // Reader::get_dataset: the schema the source will produce (CSV: inferred from the text, host only)
    Dataset get_dataset() const {
        if (source != Csv) throw DataFrameError(DataFrameError::ComputeError, "get_dataset: only CSV sources are planned here");
        const csv::Text t = csv::load(path, csv.has_headers, (char)csv.delimiter.value_or((uint8_t)','), csv.max_records);
        std::vector<size_t> cols;
        for (size_t i = 0; i < t.header.size(); ++i)
            if (!csv.projection || std::find(csv.projection->begin(), csv.projection->end(), i) != csv.projection->end()) cols.push_back(i);
        const std::vector<csv::Inferred> types = csv::infer(t, cols);
        Dataset d;
        d.name = "csv_source";
        for (size_t k = 0; k < cols.size(); ++k) d.columns.push_back(Column{t.header[cols[k]], types[k].dtype});
        return d;
    }
};

struct Transformation {
    enum Kind { GroupAggregate, Calculate, Select, Drop, Limit, Filter, Sort, Join, Read } kind = Calculate;
    Reader reader;                          // Read
    Calculation calc;                       // Calculate
    std::vector<std::string> names;         // Select / Drop / group columns
    std::vector<Aggregation> aggregations;  // GroupAggregate
    size_t limit = 0;
    FilterRef filter;
    std::vector<bool> sort_descending;      // Sort: per criterion (names holds the columns)
    static Transformation Read_(Reader r) { Transformation t; t.kind = Read; t.reader = std::move(r); return t; }
    static Transformation Calculate_(Calculation c) { Transformation t; t.kind = Calculate; t.calc = std::move(c); return t; }
    static Transformation Filter_(FilterRef f) { Transformation t; t.kind = Filter; t.filter = std::move(f); return t; }
    static Transformation Limit_(size_t n) { Transformation t; t.kind = Limit; t.limit = n; return t; }
    static Transformation Select_(std::vector<std::string> n) { Transformation t; t.kind = Select; t.names = std::move(n); return t; }
    static Transformation Drop_(std::vector<std::string> n) { Transformation t; t.kind = Drop; t.names = std::move(n); return t; }
    static Transformation Sort_(std::vector<std::string> cols, std::vector<bool> descending) {
        Transformation t; t.kind = Sort; t.names = std::move(cols); t.sort_descending = std::move(descending); return t;
    }
    static Transformation GroupAggregate_(std::vector<std::string> groups, std::vector<Aggregation> a) {
        Transformation t; t.kind = GroupAggregate; t.names = std::move(groups); t.aggregations = std::move(a); return t;
    }
};
struct Computation {
    std::vector<Dataset> input;
    std::vector<Transformation> transformations;
    Dataset output;
    static Computation empty() { return Computation(); }
    // Computation::compute_read (src/expression.rs:571-578)
    static Computation compute_read(const Reader& read) {
        Computation c;
        c.transformations = {Transformation::Read_(read)};
        c.output = read.get_dataset();
        return c;
    }
    bool is_single(Transformation::Kind k) const { return transformations.size() == 1 && transformations[0].kind == k; }
};

// optimise (src/optimiser.rs:5-101) over an unrolled plan (newest computation first, the read last), with its two helpers
// (:103-181 optimise_read, :183-235 optimise_project_calc).  The same rules: two Limits merge into the smaller; a Limit sinks
// below the computation that follows it; a Select / Limit in front of a CSV Read becomes the reader's projection /
// max_records; a Select that names a computed column and all of its inputs is pushed above the Calculate, a Calculate whose
// output is not selected is dropped.  Deliberate differences (intent reproduced, bugs not copied — this plan is evaluated, the
// reference's is only printed): the projection indices are taken against the READ's columns (the reference enumerates the
// Select's own output, right only while the selected columns are a prefix of the file's); the computation still pending when
// the list ends is emitted (the reference returns without it unless optimise_read happened to emit it: [Calculate, Read] comes
// back as [Calculate]); and where the reference's optimise_project_calc says "drop calculation" it returns the calculation and
// drops the SELECT (its `project` argument is the calculate computation) — here the calculation goes and the select stays.
inline std::pair<std::vector<Computation>, Computation> optimise_read(const Computation& input, const Computation& read, const Transformation& x, bool* emitted) {
    std::vector<Computation> output;
    Computation mutated = read;
    const Reader& reader = read.transformations[0].reader;
    *emitted = false;
    if (reader.source != Reader::Csv) { output.push_back(input); return {output, mutated}; }   // no projection support (Arrow / Json / Parquet / Sql)
    CsvReadOptions options = reader.csv;
    if (x.kind == Transformation::Select && !options.projection) {
        std::vector<size_t> proj;
        Dataset out_ds;
        out_ds.name = read.output.name;
        for (size_t i = 0; i < read.output.columns.size(); ++i)
            if (std::find(x.names.begin(), x.names.end(), read.output.columns[i].name) != x.names.end()) { proj.push_back(i); out_ds.columns.push_back(read.output.columns[i]); }
        options.projection = proj;
        mutated.transformations = {Transformation::Read_(Reader::Csv_(reader.path, options))};
        mutated.output = out_ds;
        output.push_back(mutated);
        *emitted = true;
    } else if (x.kind == Transformation::Limit) {
        options.max_records = options.max_records ? std::min(*options.max_records, x.limit) : x.limit;
        mutated.transformations = {Transformation::Read_(Reader::Csv_(reader.path, options))};
        output.push_back(mutated);
        *emitted = true;
    } else {
        output.push_back(input);   // Drop, a second projection, anything else: kept as it is
    }
    return {output, mutated};
}
inline std::pair<std::vector<Computation>, Computation> optimise_project_calc(const Computation& input, const Computation& project, const Transformation& x, const Calculation& calc) {
    std::vector<Computation> output;
    Computation mutated = project;
    if (x.kind == Transformation::Select) {
        const bool selected_output = std::find(x.names.begin(), x.names.end(), calc.output.name) != x.names.end();
        if (!selected_output) return {output, input};     // the calculated column is not selected: the calculation is dropped, the select stays pending
        size_t selected_inputs = 0;
        for (auto& in : calc.inputs) selected_inputs += std::find(x.names.begin(), x.names.end(), in.name) != x.names.end();
        if (selected_inputs == calc.inputs.size()) {     // every input is selected: the select moves above the calculation
            output.push_back(input);
            Dataset d;
            d.name = mutated.output.name;
            for (auto& c : mutated.output.columns) if (c.name != calc.output.name) d.columns.push_back(c);
            mutated.output = d;
            return {output, mutated};
        }
        output.push_back(input);   // an input of the calculation is not selected: both steps stay as they are
    } else {
        output.push_back(input);   // Drop
    }
    return {output, mutated};
}
inline std::vector<Computation> optimise(const std::vector<Computation>& computations) {
    std::vector<Computation> output;
    Computation input = Computation::empty();
    bool pending = false;   // `input` holds a computation that is not in `output` yet
    auto emit = [&](const Computation& c) { output.push_back(c); };
    for (const Computation& c : computations) {
        if (input.transformations.empty()) { input = c; pending = true; continue; }
        const size_t ci = c.input.size(), ii = input.input.size();
        if (ci == 0 && ii == 1 && c.is_single(Transformation::Read) && input.transformations.size() == 1) {
            bool emitted = false;
            auto r = optimise_read(input, c, input.transformations[0], &emitted);
            for (auto& o : r.first) emit(o);
            input = r.second;
            pending = !emitted;
        } else if (ci == 1 && ii == 1 && input.is_single(Transformation::Limit) && c.is_single(Transformation::Limit)) {
            input.transformations = {Transformation::Limit_(std::min(input.transformations[0].limit, c.transformations[0].limit))};
            emit(input);
            pending = false;
        } else if (ci == 1 && ii == 1 && input.is_single(Transformation::Limit) && !c.is_single(Transformation::Limit)) {
            emit(c);               // the limit sinks below the computation that follows it; it stays the pending input
        } else if (ci == 1 && ii == 1 && (input.is_single(Transformation::Select) || input.is_single(Transformation::Drop)) && c.is_single(Transformation::Calculate)) {
            auto r = optimise_project_calc(input, c, input.transformations[0], c.transformations[0].calc);
            for (auto& o : r.first) emit(o);
            input = r.second;
            pending = true;
        } else {
            if (pending) emit(input);
            input = c;
            pending = true;
        }
    }
    if (pending && !input.transformations.empty()) emit(input);
    return output;
}

// Calculation::calculate (src/expression.rs:433-499): name lookup + dispatch to the operation builders.
inline std::vector<Transformation> calculate(const Dataset& ds, const std::vector<std::string>& in_col_names, const Function& function,
                                             std::optional<std::string> out_col_name, std::optional<DataType> out_col_type) {
    std::vector<Column> inputs;
    for (auto& n : in_col_names) {
        auto c = ds.get_column(n);
        if (!c) throw DataFrameError(DataFrameError::ParseError, "Column " + n + " not found");
        inputs.push_back(c->second);
    }
    std::vector<Calculation> ops;
    switch (function.kind) {
        case Function::Filter: return {Transformation::Filter_(function.filter)};
        case Function::Cast: ops = CastOperation::transform(inputs, out_col_name, out_col_type); break;
        case Function::Rename: throw DataFrameError(DataFrameError::ComputeError, "Please use rename function directly for now");
        default:
            switch (function.scalar) {
                case ScalarFunction::Add: ops = AddOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Subtract: ops = SubtractOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Multiply: ops = MultiplyOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Divide: ops = DivideOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Sine: ops = SinOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Cosine: ops = CosOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Tangent: ops = TanOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Cotangent: ops = CotOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Secant: ops = SecOperation::transform(inputs, out_col_name, out_col_type); break;
                case ScalarFunction::Cosecant: ops = CscOperation::transform(inputs, out_col_name, out_col_type); break;
                default: throw DataFrameError(DataFrameError::ComputeError, std::string("Scalar Function ") + scalar_function_name(function.scalar) + " not supported");
            }
    }
    std::vector<Transformation> out;
    for (auto& c : ops) out.push_back(Transformation::Calculate_(c));
    return out;
}

}  // namespace plan

// ------------------------------------------------------------------------------------------------
// DataFrame (src/dataframe.rs:30-337)

struct RecordBatch {
    Schema schema;
    std::vector<ArrayRef> columns;
    int64_t num_rows() const { return columns.empty() ? 0 : columns[0]->length; }
};

// arrow::compute::cast per chunk (Function::Cast, src/evaluation.rs:296-315) on device arrays
inline std::vector<ArrayRef> cast_arrays(const std::vector<ArrayRef>& arr, DataType to) {
    std::vector<rdf_array> a;
    std::vector<std::shared_ptr<Array>> outs;
    std::vector<rdf_out> ov;
    for (auto& x : arr) { a.push_back(x->view()); outs.push_back(Array::make_out(to, x->length, true, x->host)); ov.push_back(outs.back()->out_view(x->length)); }   // (a value the target type cannot hold becomes NULL: always a bitmap)
    check(rdf_cast(a.data(), (int64_t)a.size(), ov.data()));
    std::vector<ArrayRef> res;
    for (size_t i = 0; i < outs.size(); ++i) { outs[i]->length = ov[i].length; outs[i]->null_count = ov[i].null_count; res.push_back(outs[i]); }
    return res;
}

// Minimal in-place flatbuffer reader for the Arrow IPC metadata (tables via vtables, vectors, strings, structs);
// every access is bounds-checked against the file image.
struct FlatBuf {
    const uint8_t* base; size_t size;
    [[noreturn]] static void oob() { throw DataFrameError(DataFrameError::IoError, "Arrow IPC: metadata out of bounds"); }
    template <class T> T rd(size_t pos) const { if (pos + sizeof(T) > size) oob(); T v; std::memcpy(&v, base + pos, sizeof(T)); return v; }
    size_t root(size_t pos) const { return pos + rd<uint32_t>(pos); }                 // position of the root table
    // position of field `id` inside table `t`, or 0 when absent (default value)
    size_t field_pos(size_t t, int id) const {
        if (!t) return 0;
        const int32_t so = rd<int32_t>(t);
        const size_t vt = (size_t)((int64_t)t - so);
        const uint16_t vsize = rd<uint16_t>(vt);
        const size_t slot = 4 + 2 * (size_t)id;
        if (slot + 2 > vsize) return 0;
        const uint16_t off = rd<uint16_t>(vt + slot);
        return off ? t + off : 0;
    }
    template <class T> T scalar(size_t t, int id, T def) const { const size_t p = field_pos(t, id); return p ? rd<T>(p) : def; }
    size_t table_field(size_t t, int id) const { const size_t p = field_pos(t, id); return p ? p + rd<uint32_t>(p) : 0; }
    std::string string(size_t t, int id) const {
        const size_t p = field_pos(t, id);
        if (!p) return "";
        const size_t s = p + rd<uint32_t>(p);
        const uint32_t n = rd<uint32_t>(s);
        if (s + 4 + n > size) oob();
        return std::string((const char*)base + s + 4, n);
    }
    // `fp` = position of a vector FIELD (from field_pos), 0 = absent
    size_t vec_len(size_t fp) const { return fp ? rd<uint32_t>(fp + rd<uint32_t>(fp)) : 0; }
    size_t vec_table(size_t fp, size_t i) const { const size_t e = fp + rd<uint32_t>(fp) + 4 + 4 * i; return e + rd<uint32_t>(e); }
    const uint8_t* vec_struct(size_t fp, size_t i, size_t stride) const {
        const size_t e = fp + rd<uint32_t>(fp) + 4 + stride * i;
        if (e + stride > size) oob();
        return base + e;
    }
};

class DataFrame {
  public:
    DataFrame() = default;
    DataFrame(Schema schema, std::vector<Column> columns) : schema_(std::move(schema)), columns_(std::move(columns)) {}
    static DataFrame empty() { return DataFrame(); }
    static DataFrame from_columns(std::vector<Column> cols) {
        Schema s;
        for (auto& c : cols) s.fields.push_back(c.field());
        for (auto& c : cols)
            if (c.num_rows() != cols[0].num_rows()) throw DataFrameError(DataFrameError::ComputeError, "columns differ in length");
        return DataFrame(std::move(s), std::move(cols));
    }
    // DataFrame::from_csv (src/dataframe.rs:349-389: arrow::csv::Reader with an inferred schema, batch_size 1024).
    // Schema inference as the Arrow CSV reader does it — a column whose non-empty cells all parse as integers is Int64,
    // as numbers Float64, as true / false Boolean, anything else Utf8; an empty cell is a NULL.  A typed column is parsed
    // into ONE host buffer (+ validity bitmap) and reaches HBM with ONE copy; its 1024-row RecordBatches are zero-copy
    // slices of that buffer (chunk i = offset 1024 i), so the chunk structure the reference's reader produces is kept
    // without one small copy per batch.  Quoted text columns are carried as opaque Utf8 on the host.
    static DataFrame from_csv(const std::string& path, size_t batch_size = 1024) {
        plan::CsvReadOptions o;
        o.batch_size = batch_size;
        return from_csv(path, o);
    }
    // ... with the reader's options (src/expression.rs:372-378): header row, delimiter, max_records, batch_size, projection —
    // what plan::optimise pushes a Limit / Select into
    static DataFrame from_csv(const std::string& path, const plan::CsvReadOptions& options) {
        const size_t batch_size = options.batch_size ? options.batch_size : 1024;
        const auto t_start = std::chrono::steady_clock::now();
        last_ingest() = IngestStats();
        std::vector<std::unique_ptr<PinnedBuffer>> staged;   // the parsed columns: alive until the fence
        const csv::Text t = csv::load(path, options.has_headers, (char)options.delimiter.value_or((uint8_t)','), options.max_records);
        std::vector<size_t> sel;
        if (options.projection) {
            for (size_t i : *options.projection) {
                if (i >= t.header.size()) throw DataFrameError(DataFrameError::ComputeError, "csv projection index out of range");
                sel.push_back(i);
            }
        } else for (size_t i = 0; i < t.header.size(); ++i) sel.push_back(i);
        const size_t n = t.records.size();
        // The types of a long file are taken from its first records and every cell is parsed ONCE against them; a cell that does
        // not fit (or a column whose sampled cells were all empty) sends the file through inference over all records and a second
        // parse — the result is always what inference over the whole file gives.
        const size_t kSample = 16384;
        bool sampled = n > 4 * kSample;
        std::vector<csv::Inferred> types = csv::infer(t, sel, 0, sampled ? kSample : n);
        if (sampled) for (auto& ty : types) if (!ty.any) { sampled = false; break; }
        if (!sampled && n > 4 * kSample) types = csv::infer(t, sel);
        // typed values are parsed STRAIGHT into page-locked column buffers (+ bitmaps) by the worker threads, then every column
        // is queued for upload; one fence after the last one
        const int64_t bbytes = (int64_t)((n + 63) / 64 * 8 + 8);
        std::vector<uint8_t*> pvs, pbs;
        std::vector<int64_t> vbytes_of;
        auto buffers = [&]() {
            staged.clear();
            pvs.assign(sel.size(), nullptr); pbs.assign(sel.size(), nullptr); vbytes_of.assign(sel.size(), 0);
            for (size_t k = 0; k < sel.size(); ++k) {
                if (types[k].dtype == DataType::Utf8) continue;
                vbytes_of[k] = types[k].dtype == DataType::Boolean ? bbytes : (int64_t)(n * 8);
                staged.push_back(std::make_unique<PinnedBuffer>(vbytes_of[k] + bbytes));
                pvs[k] = staged.back()->data();
                pbs[k] = pvs[k] + vbytes_of[k];
                std::memset(pvs[k], 0, (size_t)(vbytes_of[k] + bbytes));
            }
        };
        buffers();
        std::vector<char> violated(sel.size(), 0);
        std::vector<csv::Filled> filled = csv::fill(t, sel, types, pvs, pbs, 0, sampled ? &violated : nullptr);
        if (sampled && std::find(violated.begin(), violated.end(), (char)1) != violated.end()) {
            types = csv::infer(t, sel);
            buffers();
            filled = csv::fill(t, sel, types, pvs, pbs);
        }
        for (size_t k = 0; k < sel.size(); ++k) types[k].any_null = filled[k].nulls > 0;
        std::vector<Column> cols;
        for (size_t k = 0; k < sel.size(); ++k) {
            const DataType dt = types[k].dtype;
            const bool any_null = types[k].any_null;
            std::vector<ArrayRef> chunks;
            if (dt == DataType::Utf8) {
                std::vector<std::string>& cells = filled[k].strings;
                for (size_t b = 0; b < n || chunks.empty(); b += batch_size) {
                    const size_t e = std::min(n, b + batch_size);
                    chunks.push_back(Array::from_strings(std::vector<std::string>(cells.begin() + b, cells.begin() + e)));
                    if (n == 0) break;
                }
            } else {
                const uint8_t* pv = pvs[k];
                const uint8_t* pb = pbs[k];
                const int64_t vbytes = vbytes_of[k];
                auto w = std::make_shared<Array>();
                w->dtype = dt;
                w->length = (int64_t)n;
                w->null_count = filled[k].nulls;
                w->values = std::make_shared<DeviceBuffer>(vbytes + 8);
                upload(w->values->data(), pv, dt == DataType::Boolean ? (int64_t)((n + 7) / 8) : vbytes, true);
                if (any_null) {
                    w->validity = std::make_shared<DeviceBuffer>(bbytes);
                    upload(w->validity->data(), pb, bbytes - 8, true);
                }
                ArrayRef whole = w;
                for (size_t b = 0; b < n || chunks.empty(); b += batch_size) {   // RecordBatches = zero-copy slices of the one buffer
                    auto c = std::const_pointer_cast<Array>(whole->slice((int64_t)b, (int64_t)std::min(batch_size, n - b)));
                    if (any_null) { c->null_count = 0; for (size_t r = b; r < std::min(n, b + batch_size); ++r) c->null_count += !((pb[r >> 3] >> (r & 7)) & 1); }
                    chunks.push_back(c);
                    if (n == 0) break;
                }
            }
            cols.push_back(Column::from_arrays(chunks, Field{t.header[sel[k]], dt, true}));
        }
        check(rdf_copy_fence());
        last_ingest().seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        return from_columns(std::move(cols));
    }

    // DataFrame::from_arrow (src/dataframe.rs:391-407: arrow::ipc::reader::FileReader -> Table::from_record_batches):
    // reads an Arrow IPC FILE or an IPC STREAM (told apart by the magic).  The flatbuffer metadata (Footer -> Schema,
    // Blocks; Message -> Schema / DictionaryBatch / RecordBatch) is decoded in place with a minimal reader; every column
    // buffer goes from the file image straight to HBM with one H2D copy (no per-value parsing, no intermediate arrays):
    // one chunk per record batch.  Primitive numeric and Boolean columns are on the compute path, Utf8 is carried
    // opaquely.  Dictionary-encoded columns (numeric or Utf8 values, any integer index type; delta and replacement
    // dictionaries of the stream format) are decoded while loading — the frame holds plain columns, as the reference's
    // kernels expect.  Nested types and compressed bodies are rejected with an error.
    // The file is read into ONE page-locked buffer; every record batch's column buffers are queued for upload straight out of
    // it (rdf_copy_h2d_async, the copy stream) while the next batch's metadata is decoded, and one fence ends the load:
    // no intermediate arrays, no runtime staging copy, no synchronisation per buffer.  last_ingest() tells what went up.
    static DataFrame from_arrow(const std::string& path) {
        const auto t0 = std::chrono::steady_clock::now();
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw DataFrameError(DataFrameError::IoError, "cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw DataFrameError(DataFrameError::IoError, "cannot stat " + path); }
        const size_t size = (size_t)st.st_size;
        // the file is MAPPED, not read into a page-locked copy: the metadata is decoded from the mapping and the column buffers
        // travel through the staging buffers of StagedUploader (page-locking 2.57 GB took 0.44 of the load's 0.49 s)
        void* map = size ? ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        ::close(fd);
        if (size && map == MAP_FAILED) throw DataFrameError(DataFrameError::IoError, "cannot map " + path);
        last_ingest() = IngestStats();
        DataFrame df;
        try {
            IpcReader r((const uint8_t*)map, size);
            r.staged = true;
            if (size >= 20 && std::memcmp(map, "ARROW1", 6) == 0) r.read_file(); else r.read_stream();
            df = r.finish();
            StagedUploader::instance().finish();
        } catch (...) {
            try { StagedUploader::instance().finish(); } catch (...) {}
            if (map) ::munmap(map, size);
            throw;
        }
        if (map) ::munmap(map, size);
        last_ingest().seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return df;
    }
    // The same file as a HOST-resident frame: the file is mapped, its metadata decoded, and every primitive / Boolean column chunk
    // becomes a view into the mapping (nothing is uploaded, nothing is copied; the frame keeps the mapping alive).  What such a
    // frame is for: `LazyFrame::read(DataFrame::from_arrow_host(path)).filter(..).aggregate({}, ..).evaluate()` — Evaluate fuses
    // the steps into one rdf_pipeline call over RDF_MEM_HOST batches, which the library streams (slab k + 1 crosses the link while
    // the kernel runs over slab k): compute starts with the first slab instead of after the file is in HBM, and a file larger
    // than free HBM is aggregated in two slabs of it.  Utf8 columns ride along on the host as in from_arrow; dictionary-encoded
    // columns are refused (they are decoded on the device).  to_device() turns the frame into an ordinary device-resident one.
    static DataFrame from_arrow_host(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw DataFrameError(DataFrameError::IoError, "cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw DataFrameError(DataFrameError::IoError, "cannot stat " + path); }
        const size_t size = (size_t)st.st_size;
        void* map = size ? ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        ::close(fd);
        if (size && map == MAP_FAILED) throw DataFrameError(DataFrameError::IoError, "cannot map " + path);
        if (!map) throw DataFrameError(DataFrameError::IoError, "Arrow IPC: empty file " + path);
        std::shared_ptr<void> keep(map, [size](void* p) { ::munmap(p, size); });
        IpcReader r((const uint8_t*)map, size);
        r.host_keep = keep;
        if (size >= 20 && std::memcmp(map, "ARROW1", 6) == 0) r.read_file(); else r.read_stream();
        return r.finish();
    }
    // every host-resident chunk uploaded (blocking copies; an already device-resident frame is returned as it is)
    DataFrame to_device() const {
        std::vector<Column> out;
        for (const Column& c : columns_) {
            std::vector<ArrayRef> arrs;
            for (const ArrayRef& a : c.data().chunks()) {
                if (!a->host) { arrs.push_back(a); continue; }
                auto d = std::make_shared<Array>(*a);
                d->host = false;
                const int64_t vb = a->dtype == DataType::Boolean ? (a->offset + a->length + 7) / 8 : (a->offset + a->length) * (int64_t)type_size(a->dtype);
                d->values = std::make_shared<DeviceBuffer>(vb + 8);
                if (vb > 0) check(rdf_copy_h2d(d->values->data(), a->values->data(), vb));
                if (a->validity) {
                    const int64_t bb = (a->offset + a->length + 7) / 8;
                    d->validity = std::make_shared<DeviceBuffer>(bb + 8);
                    if (bb > 0) check(rdf_copy_h2d(d->validity->data(), a->validity->data(), bb));
                }
                arrs.push_back(d);
            }
            out.push_back(Column::from_arrays(arrs, c.field()));
        }
        return from_columns(std::move(out));
    }
    // An image the caller holds: pinned in place (rdf_host_register) for the duration of the load when the platform allows it.
    static DataFrame from_arrow_image(const uint8_t* img, size_t size) {
        const auto t0 = std::chrono::steady_clock::now();
        last_ingest() = IngestStats();
        const bool reg = size > 0 && rdf_host_register(const_cast<uint8_t*>(img), (int64_t)size) == RDF_OK;
        DataFrame df;
        try { df = load_arrow_image(img, size, reg); }
        catch (...) { if (reg) { (void)rdf_copy_fence(); (void)rdf_host_unregister(const_cast<uint8_t*>(img)); } throw; }
        if (reg) check(rdf_host_unregister(const_cast<uint8_t*>(img)));
        last_ingest().seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return df;
    }
  private:
    static DataFrame load_arrow_image(const uint8_t* img, size_t size, bool pinned) {
        IpcReader r(img, size);
        r.pinned = pinned;
        try {
            if (size >= 20 && std::memcmp(img, "ARROW1", 6) == 0) r.read_file(); else r.read_stream();
            DataFrame df = r.finish();
            check(rdf_copy_fence());     // the image may go away (and kernels may run) after this
            return df;
        } catch (...) { (void)rdf_copy_fence(); throw; }
    }
  public:

  private:
    struct IpcReader {
        struct Dict { bool utf8 = false; DataType dt = DataType::Int64; std::vector<std::string> strs; std::vector<uint8_t> values; std::vector<bool> valid; int64_t n = 0; };
        struct Col { Field field; bool dict = false; int64_t dict_id = 0; DataType index_type = DataType::Int32; };
        const uint8_t* img; size_t size; FlatBuf fb;
        bool pinned = false;     // the image is page-locked: column buffers go up asynchronously, straight out of it
        bool staged = false;     // the image is pageable (a mapped file): column buffers go through StagedUploader
        std::shared_ptr<void> host_keep;   // set: nothing is uploaded — the arrays are views into the image, which this keeps alive
        std::vector<Col> cols;
        std::map<int64_t, Dict> dicts;
        std::vector<std::vector<ArrayRef>> chunks;
        IpcReader(const uint8_t* i, size_t n) : img(i), size(n), fb{i, n} {}
        static DataFrameError bad(const std::string& m) { return DataFrameError(DataFrameError::IoError, "Arrow IPC: " + m); }

        static DataType int_type(const FlatBuf& fb, size_t ty, const std::string& name) {   // Int { 0 bitWidth, 1 is_signed }
            const int bw = fb.scalar<int32_t>(ty, 0, 0);
            const bool sg = fb.scalar<uint8_t>(ty, 1, 0) != 0;
            switch (bw) {
                case 8: return sg ? DataType::Int8 : DataType::UInt8;
                case 16: return sg ? DataType::Int16 : DataType::UInt16;
                case 32: return sg ? DataType::Int32 : DataType::UInt32;
                case 64: return sg ? DataType::Int64 : DataType::UInt64;
                default: throw bad("integer width of column " + name);
            }
        }
        void read_schema(size_t schema_t) {   // Schema: 0 endianness, 1 fields
            if (!schema_t) throw bad("no schema");
            if (fb.scalar<int16_t>(schema_t, 0, 0) != 0) throw bad("big-endian data is not supported");
            const size_t fields_v = fb.field_pos(schema_t, 1);
            const size_t nfields = fb.vec_len(fields_v);
            for (size_t i = 0; i < nfields; ++i) {
                // Field: 0 name, 1 nullable, 2 type_type, 3 type, 4 dictionary, 5 children
                const size_t ft = fb.vec_table(fields_v, i);
                Col c;
                c.field.name = fb.string(ft, 0);
                c.field.nullable = fb.scalar<uint8_t>(ft, 1, 0) != 0;
                const int tt = fb.scalar<uint8_t>(ft, 2, 0);
                const size_t ty = fb.table_field(ft, 3);
                if (tt == 2) c.field.data_type = int_type(fb, ty, c.field.name);
                else if (tt == 3) {  // FloatingPoint { 0 precision: HALF, SINGLE, DOUBLE }
                    const int pr = fb.scalar<int16_t>(ty, 0, 0);
                    if (pr == 1) c.field.data_type = DataType::Float32; else if (pr == 2) c.field.data_type = DataType::Float64; else throw bad("half floats (column " + c.field.name + ")");
                } else if (tt == 6) c.field.data_type = DataType::Boolean;
                else if (tt == 5) c.field.data_type = DataType::Utf8;
                else throw bad("column " + c.field.name + ": unsupported type id " + std::to_string(tt));
                if (const size_t de = fb.table_field(ft, 4)) {   // DictionaryEncoding { 0 id, 1 indexType: Int (default int32), 2 isOrdered, 3 dictionaryKind }
                    c.dict = true;
                    c.dict_id = fb.scalar<int64_t>(de, 0, 0);
                    const size_t it = fb.table_field(de, 1);
                    c.index_type = it ? int_type(fb, it, c.field.name) : DataType::Int32;
                    if (c.field.data_type == DataType::Boolean) throw bad("dictionary of Boolean values (column " + c.field.name + ")");
                }
                cols.push_back(c);
            }
            chunks.assign(cols.size(), {});
        }
        struct Body { const uint8_t* p; int64_t len; };
        // one column's buffers out of a RecordBatch table: validity, values (Utf8: offsets, data)
        struct ColBufs { int64_t len, nulls, vo, vl, d0, dl, so, sl; };
        struct BatchCursor { size_t nodes_v, bufs_v, ni = 0, bi = 0; };
        ColBufs next_col(BatchCursor& cur, const Body& body, bool utf8) {
            if (cur.ni >= fb.vec_len(cur.nodes_v)) throw bad("field node list too short");
            const uint8_t* node = fb.vec_struct(cur.nodes_v, cur.ni++, 16);
            ColBufs b{};
            std::memcpy(&b.len, node, 8); std::memcpy(&b.nulls, node + 8, 8);
            auto next_buf = [&](int64_t& bo, int64_t& bl) {
                if (cur.bi >= fb.vec_len(cur.bufs_v)) throw bad("buffer list too short");
                const uint8_t* p = fb.vec_struct(cur.bufs_v, cur.bi++, 16);
                std::memcpy(&bo, p, 8); std::memcpy(&bl, p + 8, 8);
                if (bo < 0 || bl < 0 || bo + bl > body.len) throw bad("buffer out of bounds");
            };
            next_buf(b.vo, b.vl);
            next_buf(b.d0, b.dl);
            if (utf8) next_buf(b.so, b.sl);
            if (b.len < 0 || (b.nulls > 0 && b.vl < (b.len + 7) / 8)) throw bad("validity buffer too short");
            return b;
        }
        static bool bit(const uint8_t* p, int64_t i) { return (p[i >> 3] >> (i & 7)) & 1; }
        std::vector<std::string> read_strings(const ColBufs& b, const Body& body) {
            if (b.dl < 4 * (b.len + 1) && b.len > 0) throw bad("string offsets buffer too short");
            std::vector<std::string> strs((size_t)b.len);
            for (int64_t r = 0; r < b.len; ++r) {
                int32_t a0, a1;
                std::memcpy(&a0, body.p + b.d0 + 4 * r, 4); std::memcpy(&a1, body.p + b.d0 + 4 * (r + 1), 4);
                if (a0 < 0 || a1 < a0 || a1 > b.sl) throw bad("string offsets out of bounds");
                strs[(size_t)r].assign((const char*)body.p + b.so + a0, (size_t)(a1 - a0));
            }
            return strs;
        }
        static ArrayRef strings_with_validity(std::vector<std::string> strs, const std::vector<bool>* valid) {
            auto a = std::const_pointer_cast<Array>(Array::from_strings(std::move(strs)));
            if (valid) {
                const auto bits = pack_bits(*valid);
                a->validity = std::make_shared<DeviceBuffer>((int64_t)bits.size() + 8);
                check(rdf_copy_h2d(a->validity->data(), bits.data(), (int64_t)bits.size()));
                for (bool v : *valid) a->null_count += !v;
            }
            return a;
        }
        // RecordBatch: 0 length, 1 nodes [FieldNode{int64 length, int64 null_count}], 2 buffers [Buffer{int64 offset, int64 length}], 3 compression
        void read_record_batch(size_t rb, const Body& body) {
            if (fb.field_pos(rb, 3)) throw bad("compressed record batches are not supported");
            const int64_t nrows = fb.scalar<int64_t>(rb, 0, 0);
            BatchCursor cur{fb.field_pos(rb, 1), fb.field_pos(rb, 2)};
            if (fb.vec_len(cur.nodes_v) != cols.size()) throw bad("nested columns are not supported");
            for (size_t c = 0; c < cols.size(); ++c) {
                const Col& col = cols[c];
                const DataType dt = col.field.data_type;
                const ColBufs b = next_col(cur, body, dt == DataType::Utf8 && !col.dict);
                if (b.len != nrows) throw bad("column length differs from the batch length");
                if (col.dict && host_keep) throw bad("dictionary-encoded column " + col.field.name + " is decoded on the device: load the file with from_arrow()");
                if (col.dict) { chunks[c].push_back(decode_dictionary_column(col, b, body)); continue; }
                if (dt == DataType::Utf8) {
                    std::vector<bool> valid;
                    if (b.nulls > 0) { valid.resize((size_t)b.len); for (int64_t r = 0; r < b.len; ++r) valid[(size_t)r] = bit(body.p + b.vo, r); }
                    chunks[c].push_back(strings_with_validity(read_strings(b, body), b.nulls > 0 ? &valid : nullptr));
                    continue;
                }
                const int64_t need = dt == DataType::Boolean ? (b.len + 7) / 8 : b.len * (int64_t)type_size(dt);
                if (b.dl < need) throw bad("values buffer of column " + col.field.name + " is too short");
                auto a = std::make_shared<Array>();
                a->dtype = dt;
                a->length = b.len;
                if (host_keep) {     // a view: the values and the bitmap stay where the file's image has them
                    a->host = true;
                    a->values = std::make_shared<DeviceBuffer>(const_cast<uint8_t*>(body.p + b.d0), need, host_keep);
                    if (b.nulls > 0) {
                        a->validity = std::make_shared<DeviceBuffer>(const_cast<uint8_t*>(body.p + b.vo), (b.len + 7) / 8, host_keep);
                        a->null_count = b.nulls;
                    }
                    chunks[c].push_back(a);
                    continue;
                }
                a->values = std::make_shared<DeviceBuffer>(need + 8);
                if (staged) StagedUploader::instance().push(a->values->data(), body.p + b.d0, need); else upload(a->values->data(), body.p + b.d0, need, pinned);
                if (b.nulls > 0) {
                    a->validity = std::make_shared<DeviceBuffer>((b.len + 7) / 8 + 8);
                    if (staged) StagedUploader::instance().push(a->validity->data(), body.p + b.vo, (b.len + 7) / 8); else upload(a->validity->data(), body.p + b.vo, (b.len + 7) / 8, pinned);
                    a->null_count = b.nulls;
                }
                chunks[c].push_back(a);
            }
        }
        // DictionaryBatch { 0 id, 1 data: RecordBatch (one column = the dictionary's values), 2 isDelta }
        void read_dictionary_batch(size_t db, const Body& body) {
            const int64_t id = fb.scalar<int64_t>(db, 0, 0);
            const bool delta = fb.scalar<uint8_t>(db, 2, 0) != 0;
            const size_t rb = fb.table_field(db, 1);
            if (!rb) throw bad("dictionary batch without data");
            if (fb.field_pos(rb, 3)) throw bad("compressed dictionary batches are not supported");
            const Col* owner = nullptr;
            for (auto& c : cols) if (c.dict && c.dict_id == id) owner = &c;
            if (!owner) throw bad("dictionary " + std::to_string(id) + " belongs to no column");
            BatchCursor cur{fb.field_pos(rb, 1), fb.field_pos(rb, 2)};
            const bool utf8 = owner->field.data_type == DataType::Utf8;
            const ColBufs b = next_col(cur, body, utf8);
            Dict fresh;
            Dict& d = delta && dicts.count(id) ? dicts[id] : fresh;
            d.utf8 = utf8; d.dt = owner->field.data_type;
            for (int64_t r = 0; r < b.len; ++r) d.valid.push_back(b.nulls > 0 ? bit(body.p + b.vo, r) : true);
            if (utf8) { auto s = read_strings(b, body); d.strs.insert(d.strs.end(), s.begin(), s.end()); }
            else {
                const int64_t es = (int64_t)type_size(d.dt);
                if (b.dl < b.len * es) throw bad("dictionary values buffer too short");
                d.values.insert(d.values.end(), body.p + b.d0, body.p + b.d0 + b.len * es);
            }
            d.n += b.len;
            if (&d == &fresh) dicts[id] = std::move(fresh);   // first sight of the id, or a replacement dictionary (stream format)
        }
        ArrayRef decode_dictionary_column(const Col& col, const ColBufs& b, const Body& body) {
            auto it = dicts.find(col.dict_id);
            if (it == dicts.end()) throw bad("record batch before the dictionary of column " + col.field.name);
            const Dict& d = it->second;
            const int64_t is = (int64_t)type_size(col.index_type);
            if (b.dl < b.len * is) throw bad("index buffer of column " + col.field.name + " is too short");
            const bool sg = col.index_type == DataType::Int8 || col.index_type == DataType::Int16 || col.index_type == DataType::Int32 || col.index_type == DataType::Int64;
            std::vector<bool> valid((size_t)b.len, true);
            bool any_null = false;
            std::vector<int64_t> idx((size_t)b.len, 0);
            for (int64_t r = 0; r < b.len; ++r) {
                if (b.nulls > 0 && !bit(body.p + b.vo, r)) { valid[(size_t)r] = false; any_null = true; continue; }
                uint64_t raw = 0;
                std::memcpy(&raw, body.p + b.d0 + r * is, (size_t)is);
                int64_t v = (int64_t)raw;
                if (sg && is < 8) { const int sh = 64 - 8 * (int)is; v = (int64_t)(raw << sh) >> sh; }
                if (v < 0 || v >= d.n) throw bad("dictionary index out of range in column " + col.field.name);
                if (!d.valid[(size_t)v]) { valid[(size_t)r] = false; any_null = true; continue; }
                idx[(size_t)r] = v;
            }
            if (d.utf8) {
                std::vector<std::string> strs((size_t)b.len);
                for (int64_t r = 0; r < b.len; ++r) if (valid[(size_t)r]) strs[(size_t)r] = d.strs[(size_t)idx[(size_t)r]];
                return strings_with_validity(std::move(strs), any_null ? &valid : nullptr);
            }
            const int64_t es = (int64_t)type_size(d.dt);
            std::vector<uint8_t> vals((size_t)(b.len * es) + 8, 0);
            for (int64_t r = 0; r < b.len; ++r) if (valid[(size_t)r]) std::memcpy(vals.data() + r * es, d.values.data() + idx[(size_t)r] * es, (size_t)es);
            auto a = std::make_shared<Array>();
            a->dtype = d.dt;
            a->length = b.len;
            a->values = std::make_shared<DeviceBuffer>(b.len * es + 8);
            if (b.len) check(rdf_copy_h2d(a->values->data(), vals.data(), b.len * es));
            if (any_null) {
                const auto bits = pack_bits(valid);
                a->validity = std::make_shared<DeviceBuffer>((int64_t)bits.size() + 8);
                check(rdf_copy_h2d(a->validity->data(), bits.data(), (int64_t)bits.size()));
                for (bool v : valid) a->null_count += !v;
            }
            return a;
        }
        // one encapsulated message at `pos`: [0xFFFFFFFF] int32 metadata length, flatbuffer Message, body.  Returns the position
        // behind the body, 0 at the end-of-stream marker.  Message: 0 version, 1 header_type, 2 header, 3 bodyLength
        size_t read_message(size_t pos, int64_t known_body_len, bool want_batches_only) {
            if (pos + 4 > size) return 0;
            uint32_t first;
            std::memcpy(&first, img + pos, 4);
            size_t mpos = pos + 4;
            uint32_t meta_len = first;
            if (first == 0xFFFFFFFFu) { if (pos + 8 > size) return 0; std::memcpy(&meta_len, img + pos + 4, 4); mpos = pos + 8; }
            if (meta_len == 0) return 0;   // end of stream
            if (mpos + meta_len > size) throw bad("message metadata out of bounds");
            const size_t msg = fb.root(mpos);
            const int kind = fb.scalar<uint8_t>(msg, 1, 0);
            const int64_t body_len = known_body_len >= 0 ? known_body_len : fb.scalar<int64_t>(msg, 3, 0);
            const size_t body_pos = mpos + meta_len;
            if (body_len < 0 || body_pos + (uint64_t)body_len > size) throw bad("message body out of bounds");
            const Body body{img + body_pos, body_len};
            const size_t hdr = fb.table_field(msg, 2);
            if (kind == 1) { if (!want_batches_only && cols.empty()) read_schema(hdr); }
            else if (kind == 2) read_dictionary_batch(hdr, body);
            else if (kind == 3) read_record_batch(hdr, body);
            else throw bad("unsupported message type " + std::to_string(kind));
            return body_pos + (size_t)body_len;
        }
        void read_stream() {
            size_t pos = 0;
            while (pos < size) {
                const size_t next = read_message(pos, -1, false);
                if (!next) break;
                if (cols.empty()) throw bad("stream does not start with a schema");
                pos = (next + 7) & ~(size_t)7;
            }
            if (cols.empty()) throw bad("not an Arrow IPC file or stream");
        }
        void read_file() {
            if (std::memcmp(img + size - 6, "ARROW1", 6) != 0) throw bad("not an Arrow IPC file (magic)");
            int32_t flen;
            std::memcpy(&flen, img + size - 10, 4);
            if (flen <= 0 || (size_t)flen + 18 > size) throw bad("bad footer length");
            const size_t footer = fb.root(size - 10 - (size_t)flen);
            // Footer: 0 version, 1 schema, 2 dictionaries, 3 recordBatches; Block { int64 offset; int32 metaDataLength; (pad) int64 bodyLength } = 24 bytes
            read_schema(fb.table_field(footer, 1));
            for (int list : {2, 3}) {
                const size_t blocks_v = fb.field_pos(footer, list);
                for (size_t b = 0; b < fb.vec_len(blocks_v); ++b) {
                    const uint8_t* blk = fb.vec_struct(blocks_v, b, 24);
                    int64_t off, body_len; int32_t meta_len;
                    std::memcpy(&off, blk, 8); std::memcpy(&meta_len, blk + 8, 4); std::memcpy(&body_len, blk + 16, 8);
                    if (off < 0 || meta_len < 8 || body_len < 0 || (uint64_t)off + (uint64_t)meta_len + (uint64_t)body_len > size) throw bad("block out of bounds");
                    read_message((size_t)off, body_len, true);
                }
            }
        }
        DataFrame finish() {
            std::vector<Column> out;
            for (size_t c = 0; c < cols.size(); ++c) {
                if (chunks[c].empty()) {   // no record batches: the schema alone, one empty chunk per column
                    if (cols[c].field.data_type == DataType::Utf8) chunks[c].push_back(Array::from_strings({}));
                    else chunks[c].push_back(Array::make_out(cols[c].field.data_type, 0, false));
                }
                out.push_back(Column::from_arrays(chunks[c], cols[c].field));
            }
            return from_columns(std::move(out));
        }
    };

  public:
    const Schema& schema() const { return schema_; }
    size_t num_columns() const { return columns_.size(); }
    size_t num_chunks() const { return columns_.empty() ? 0 : columns_[0].data().num_chunks(); }
    int64_t num_rows() const { return columns_.empty() ? 0 : columns_[0].num_rows(); }
    const Column& column(size_t i) const { return columns_[i]; }
    const std::vector<Column>& columns() const { return columns_; }
    const Column& column_by_name(const std::string& name) const {
        auto c = schema_.column_with_name(name);
        if (!c) throw DataFrameError(DataFrameError::ComputeError, "Column not found by name: " + name);
        return columns_[c->first];
    }
    bool has_column(const std::string& name) const { return schema_.column_with_name(name).has_value(); }

    // :97-113 — an existing column of that name is dropped, the new one goes last
    DataFrame with_column(const std::string& name, Column column) const {
        DataFrame d = has_column(name) ? drop({name}) : *this;
        column = column.renamed(name);
        d.schema_.fields.push_back(column.field());
        d.columns_.push_back(std::move(column));
        return d;
    }
    DataFrame with_column_renamed(const std::string& old_name, const std::string& new_name) const {  // :116-124
        DataFrame d = *this;
        auto c = schema_.column_with_name(old_name);
        if (!c) throw DataFrameError(DataFrameError::NoneError, "column " + old_name + " not found");
        d.schema_.fields[c->first].name = new_name;
        d.columns_[c->first] = d.columns_[c->first].renamed(new_name);
        return d;
    }
    // :128-163 — chunk i of every column = RecordBatch i
    std::vector<RecordBatch> to_record_batches() const {
        std::vector<RecordBatch> out;
        for (size_t i = 0; i < num_chunks(); ++i) {
            RecordBatch b;
            b.schema = schema_;
            for (auto& c : columns_) b.columns.push_back(c.data().chunk(i));
            out.push_back(std::move(b));
        }
        return out;
    }
    // DataFrame::with_id (:234-249): a UInt64 column counting from 100 000 * chunk index + 1 within every chunk (the
    // reference's "no record batch has 100k rows" assumption included).  The ids are built on the host and uploaded.
    DataFrame with_id(const std::string& name) const {
        std::vector<ArrayRef> arrays;
        if (!columns_.empty()) {
            size_t index = 0;
            for (auto& ch : columns_[0].data().chunks()) {
                std::vector<uint64_t> ids((size_t)ch->length);
                for (size_t i = 0; i < ids.size(); ++i) ids[i] = 100000ull * index + 1 + i;
                arrays.push_back(Array::from_vec<uint64_t>(ids));
                ++index;
            }
        }
        return with_column(name, Column::from_arrays(arrays, Field{name, DataType::UInt64, false}));
    }
    DataFrame limit(size_t count) const {  // :166-175, zero-copy slices
        std::vector<Column> cols;
        for (auto& c : columns_) cols.push_back(c.slice(0, (int64_t)count));
        return DataFrame(schema_, std::move(cols));
    }
    DataFrame select(const std::vector<std::string>& names) const {  // :258-297 — unknown names are omitted; "*" keeps every column (:272)
        for (auto& n : names) if (n == "*") return *this;
        Schema s; std::vector<Column> cols;
        for (size_t i = 0; i < columns_.size(); ++i)
            for (auto& n : names) if (schema_.fields[i].name == n) { s.fields.push_back(schema_.fields[i]); cols.push_back(columns_[i]); break; }
        return DataFrame(std::move(s), std::move(cols));
    }
    DataFrame drop(const std::vector<std::string>& names) const {  // :302-337
        Schema s; std::vector<Column> cols;
        for (size_t i = 0; i < columns_.size(); ++i) {
            bool dropped = false;
            for (auto& n : names) dropped |= schema_.fields[i].name == n;
            if (!dropped) { s.fields.push_back(schema_.fields[i]); cols.push_back(columns_[i]); }
        }
        return DataFrame(std::move(s), std::move(cols));
    }

    // evaluate_boolean_filter (:612-624): one BooleanArray mask per RecordBatch, computed in ONE launch
    Column evaluate_boolean_filter(const FilterRef& filter) const {
        Lowered low;
        const int root = low.add(filter_to_expr(filter, [this](const std::string& n) {
            if (!has_column(n)) throw DataFrameError(DataFrameError::ComputeError, "Cannot find column " + n);  // expression.rs:812-815
            return Expr::col(n);
        }));
        return run_predicate(low, root);
    }
    Column run_predicate(const Lowered& low, int root) const {
        const size_t nch = num_chunks();
        std::vector<rdf_array> cols;
        bool nullable = false;
        for (auto& n : low.columns)
            for (auto& a : column_by_name(n).data().chunks()) { cols.push_back(a->view()); nullable |= a->validity != nullptr; }
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        const auto counts = columns_[0].data().chunk_counts();
        const bool on_host = is_host();
        for (size_t i = 0; i < nch; ++i) { outs.push_back(Array::make_out(DataType::Boolean, counts[i], nullable, on_host)); ov.push_back(outs.back()->out_view(counts[i])); }
        if (low.columns.empty()) {  // constant predicate: the batch length comes from column 0
            for (auto& a : columns_[0].data().chunks()) cols.push_back(a->view());
            check(rdf_predicate(low.nodes.data(), (int32_t)low.nodes.size(), root, cols.data(), 1, (int64_t)nch, ov.data()));
        } else
            check(rdf_predicate(low.nodes.data(), (int32_t)low.nodes.size(), root, cols.data(), (int32_t)low.columns.size(), (int64_t)nch, ov.data()));
        std::vector<ArrayRef> res;
        for (size_t i = 0; i < nch; ++i) { outs[i]->length = ov[i].length; outs[i]->null_count = ov[i].null_count; res.push_back(outs[i]); }
        return Column::from_arrays(res, Field{"bool_filter", DataType::Boolean, true});
    }
    // the frame's numeric / Boolean column chunks are views into host memory (from_arrow_host, or the result of an operator over such a frame)
    bool is_host() const {
        for (auto& c : columns_) for (auto& a : c.data().chunks()) if (a->dtype != DataType::Utf8 && a->host) return true;
        return false;
    }
    // ... ALL of them are (a host frame with a device-computed column added is not: the streamed calls take host chunks only)
    bool all_host() const {
        bool any = false;
        for (auto& c : columns_) for (auto& a : c.data().chunks()) if (a->dtype != DataType::Utf8) { if (!a->host) return false; any = true; }
        return any;
    }
    // DataFrame::filter (:178-189): the mask, then EVERY column compacted by it in one pass (rdf_filter_columns).
    // A host-resident frame of numeric columns takes ONE call instead (rdf_filter_pipeline): predicate and compaction run on the
    // device slab by slab while the next slab comes in and the kept rows of the previous one go back — the result is a
    // host-resident frame again, and a frame larger than free HBM is filtered all the same.
    DataFrame filter(const FilterRef& condition) const {
        bool all_numeric = !columns_.empty();
        for (auto& c : columns_) all_numeric = all_numeric && (is_integer(c.data_type()) || is_float(c.data_type()));
        if (!(all_host() && all_numeric && columns_.size() <= 64)) return filter_by_mask(evaluate_boolean_filter(condition));
        Lowered low;
        const int root = low.add(filter_to_expr(condition, [this](const std::string& n) {
            if (!has_column(n)) throw DataFrameError(DataFrameError::ComputeError, "Cannot find column " + n);
            return Expr::col(n);
        }));
        // the call's column order: the predicate's columns first (the expression indexes them), then the rest of the frame
        std::vector<size_t> order;
        for (auto& n : low.columns) order.push_back(schema_.column_with_name(n)->first);
        for (size_t k = 0; k < columns_.size(); ++k) if (std::find(order.begin(), order.end(), k) == order.end()) order.push_back(k);
        if (order.size() != columns_.size()) return filter_by_mask(evaluate_boolean_filter(condition));   // (a column named twice by the predicate's lowering)
        const size_t nch = num_chunks();
        std::vector<rdf_array> cv;
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        for (size_t k : order)
            for (auto& a : columns_[k].data().chunks()) {
                cv.push_back(a->view());
                outs.push_back(Array::make_out(a->dtype, a->length, a->validity != nullptr, true));
                ov.push_back(outs.back()->out_view(a->length));
            }
        check(rdf_filter_pipeline(low.nodes.data(), (int32_t)low.nodes.size(), root, cv.data(), (int32_t)order.size(), (int64_t)nch, ov.data()));
        std::vector<Column> result(columns_.size());
        for (size_t j = 0; j < order.size(); ++j) {
            std::vector<ArrayRef> chunks;
            for (size_t i = 0; i < nch; ++i) {
                auto& o = outs[j * nch + i];
                o->length = ov[j * nch + i].length;
                o->null_count = ov[j * nch + i].null_count;
                chunks.push_back(o);
            }
            result[order[j]] = Column::from_arrays(chunks, columns_[order[j]].field());
        }
        return DataFrame(schema_, std::move(result));
    }
    DataFrame filter_by_mask(const Column& mask) const {
        const size_t nch = num_chunks();
        const auto mv = mask.data().views();
        std::vector<int64_t> counts(nch);
        check(rdf_filter_count(mv.data(), (int64_t)nch, counts.data()));
        std::vector<Column> result(columns_.size());
        for (size_t base = 0; base < columns_.size(); base += 256) {  // rdf_filter_columns takes up to 256 columns per call
            const size_t n = std::min<size_t>(256, columns_.size() - base);
            std::vector<rdf_array> cv;
            std::vector<std::shared_ptr<Array>> outs;
            std::vector<rdf_out> ov;
            std::vector<std::vector<ArrayRef>> widened(n);   // Boolean columns travel through the compaction as UInt8 (cast there and back on the device)
            for (size_t k = 0; k < n; ++k) {
                const Column& c = columns_[base + k];
                const bool is_bool = c.data_type() == DataType::Boolean;
                if (is_bool) widened[k] = cast_arrays(c.data().chunks(), DataType::UInt8);
                for (size_t i = 0; i < nch; ++i) {
                    const ArrayRef& src = is_bool ? widened[k][i] : c.data().chunk(i);
                    if (c.data_type() == DataType::Utf8) {   // opaque host strings: a placeholder keeps the column count of the call
                        cv.push_back(rdf_array{}); outs.push_back(nullptr); ov.push_back(rdf_out{});
                        continue;
                    }
                    cv.push_back(src->view());
                    outs.push_back(Array::make_out(is_bool ? DataType::UInt8 : c.data_type(), counts[i], src->validity != nullptr, src->host));
                    ov.push_back(outs.back()->out_view(counts[i]));
                }
            }
            // device columns of this group, packed without the Utf8 placeholders
            std::vector<rdf_array> dcv; std::vector<rdf_out> dov; std::vector<size_t> dmap;
            for (size_t k = 0; k < n; ++k)
                if (columns_[base + k].data_type() != DataType::Utf8) { dmap.push_back(k); for (size_t i = 0; i < nch; ++i) { dcv.push_back(cv[k * nch + i]); dov.push_back(ov[k * nch + i]); } }
            if (!dmap.empty()) check(rdf_filter_columns(dcv.data(), (int32_t)dmap.size(), mv.data(), (int64_t)nch, dov.data()));
            for (size_t d = 0; d < dmap.size(); ++d) {
                const size_t k = dmap[d];
                std::vector<ArrayRef> chunks;
                for (size_t i = 0; i < nch; ++i) { auto& o = outs[k * nch + i]; o->length = dov[d * nch + i].length; o->null_count = dov[d * nch + i].null_count; chunks.push_back(o); }
                if (columns_[base + k].data_type() == DataType::Boolean) chunks = cast_arrays(chunks, DataType::Boolean);
                result[base + k] = Column::from_arrays(chunks, columns_[base + k].field());
            }
        }
        // Utf8 columns are carried on the host: filtered there with the mask's bits (value AND validity, like Column::filter)
        bool any_text = false;
        for (auto& c : columns_) any_text |= c.data_type() == DataType::Utf8;
        if (any_text) {
            std::vector<std::vector<bool>> keep(nch);
            for (size_t i = 0; i < nch; ++i) {
                const auto bits = mask.data().chunk(i)->bools_to_host(), valid = mask.data().chunk(i)->valid_to_host();
                keep[i].resize(bits.size());
                for (size_t r = 0; r < bits.size(); ++r) keep[i][r] = bits[r] && valid[r];
            }
            for (size_t k = 0; k < columns_.size(); ++k) {
                if (columns_[k].data_type() != DataType::Utf8) continue;
                std::vector<ArrayRef> chunks;
                for (size_t i = 0; i < nch; ++i) {
                    const auto& src = *columns_[k].data().chunk(i)->strings;
                    std::vector<std::string> out;
                    const size_t first = (size_t)columns_[k].data().chunk(i)->offset;
                    for (size_t r = 0; r < keep[i].size(); ++r) if (keep[i][r]) out.push_back(src[first + r]);
                    chunks.push_back(Array::from_strings(std::move(out)));
                }
                result[k] = Column::from_arrays(chunks, columns_[k].field());
            }
        }
        return DataFrame(schema_, std::move(result));
    }
    // DataFrame::join (:626-719): equi-join indices on 1..4 key column pairs, then Column::take of every column
    // of both frames (left columns first).  JoinType as src/expression.rs:339-345.
    enum class JoinType { LeftJoin = RDF_JOIN_LEFT, RightJoin = RDF_JOIN_RIGHT, InnerJoin = RDF_JOIN_INNER, FullJoin = RDF_JOIN_FULL };
    struct JoinCriteria { JoinType join_type; std::vector<std::pair<std::string, std::string>> criteria; };
    DataFrame join(const DataFrame& other, const JoinCriteria& jc) const {
        if (jc.criteria.empty() || jc.criteria.size() > 4) throw DataFrameError(DataFrameError::ComputeError, "join: 1 to 4 key column pairs");
        // key pair k: this[criteria[k].first] against other[criteria[k].second], laid out [k * nchunks + chunk]
        std::vector<rdf_array> lk, rk;
        for (auto& c : jc.criteria) {
            for (auto& v : column_by_name(c.first).data().views()) lk.push_back(v);
            for (auto& v : other.column_by_name(c.second).data().views()) rk.push_back(v);
        }
        const int32_t nk = (int32_t)jc.criteria.size();
        const int64_t lnc = (int64_t)num_chunks(), rnc = (int64_t)other.num_chunks();
        int64_t rows = 0;
        check(rdf_equijoin_indices_multi(lk.data(), lnc, rk.data(), rnc, nk, (int32_t)jc.join_type, nullptr, nullptr, &rows));
        auto li = Array::make_out(DataType::UInt32, rows, true), ri = Array::make_out(DataType::UInt32, rows, true);
        rdf_out lo = li->out_view(rows), ro = ri->out_view(rows);
        check(rdf_equijoin_indices_multi(lk.data(), lnc, rk.data(), rnc, nk, (int32_t)jc.join_type, &lo, &ro, &rows));
        li->length = ri->length = rows;
        li->null_count = lo.null_count; ri->null_count = ro.null_count;
        std::vector<Column> cols;
        for (auto& c : columns_) cols.push_back(c.take(li, 4096));
        for (auto& c : other.columns_) cols.push_back(c.take(ri, 4096));
        return DataFrame::from_columns(cols);
    }
    // DataFrame::sort (:194-214): lexsort_to_indices over the criteria columns, then sort_by_indices.
    // nulls_first is accepted and ignored exactly like the reference does (:208).
    struct SortCriteria { std::string column; bool descending = false; bool nulls_first = false; };
    DataFrame sort(const std::vector<SortCriteria>& criteria) const {
        if (criteria.empty()) throw DataFrameError(DataFrameError::ComputeError, "Sort criteria cannot be empty");
        std::vector<rdf_array> cols;
        std::vector<rdf_sort_options> opts;
        for (auto& c : criteria) {
            for (auto& a : column_by_name(c.column).data().chunks()) cols.push_back(a->view());
            opts.push_back(rdf_sort_options{c.descending ? 1 : 0, 0});
        }
        auto idx = Array::make_out(DataType::UInt32, num_rows(), false);
        rdf_out ov = idx->out_view(num_rows());
        check(rdf_sort_to_indices(cols.data(), (int32_t)criteria.size(), (int64_t)num_chunks(), opts.data(), &ov));
        idx->length = ov.length;
        return take(idx);
    }
    // sort_by_indices (:216-222): Column::take of every column (chunk size 4096 as in the reference)
    DataFrame take(const ArrayRef& indices) const {
        std::vector<Column> cols;
        for (auto& c : columns_) cols.push_back(c.take(indices, 4096));
        return DataFrame(schema_, std::move(cols));
    }

  private:
    Schema schema_;
    std::vector<Column> columns_;
};

// ------------------------------------------------------------------------------------------------
// GpuFrame: a DataFrame of numeric columns pinned in HBM behind a frame handle (rdf_frame_pin), and the reference's
// frame-producing operators — DataFrame::filter (src/dataframe.rs:178-189), DataFrame::take / sort (:194-222), GroupAggregate —
// as handle-in / handle-out calls: nothing per RecordBatch is marshalled, so a frame held in the readers' 1024-row batches
// costs what its kernels cost, and a chain df.pin().filter(..).sort(..) never leaves the device.  to_dataframe() materialises
// the columns (one descriptor walk) when a caller wants Arrow arrays back; they alias the frame's buffers and keep it alive.
class GpuFrame {
  public:
    static GpuFrame pin(const DataFrame& df) {
        if (df.num_columns() == 0 || df.num_chunks() == 0) throw DataFrameError(DataFrameError::ComputeError, "cannot pin an empty frame");
        std::vector<rdf_array> views;
        for (size_t c = 0; c < df.num_columns(); ++c) {
            const DataType dt = df.column(c).data_type();
            if (dt == DataType::Utf8 || dt == DataType::Boolean) throw DataFrameError(DataFrameError::ComputeError, "GpuFrame holds numeric columns (column " + df.column(c).name() + ")");
            for (auto& a : df.column(c).data().chunks()) views.push_back(a->view());
        }
        rdf_frame* h = nullptr;
        check(rdf_frame_pin(views.data(), (int32_t)df.num_columns(), (int64_t)df.num_chunks(), &h));
        GpuFrame g;
        g.schema_ = df.schema();
        g.h_ = std::shared_ptr<Handle>(new Handle{h, std::make_shared<DataFrame>(df)});   // the pinned buffers live as long as the handle
        return g;
    }
    const Schema& schema() const { return schema_; }
    size_t num_columns() const { return schema_.fields.size(); }
    int64_t num_rows() const { int64_t r = 0; check(rdf_frame_info(h_->h, nullptr, nullptr, &r)); return r; }
    int64_t num_chunks() const { int64_t c = 0; check(rdf_frame_info(h_->h, nullptr, &c, nullptr)); return c; }

    // DataFrame::filter: the predicate over every batch, every column compacted in one pass, batch boundaries kept
    GpuFrame filter(const FilterRef& condition) const {
        Lowered low;
        for (auto& f : schema_.fields) low.columns.push_back(f.name);     // expression column c = frame column c
        const int root = low.add(filter_to_expr(condition, [this](const std::string& n) {
            if (!schema_.column_with_name(n)) throw DataFrameError(DataFrameError::ComputeError, "Cannot find column " + n);
            return Expr::col(n);
        }));
        rdf_frame* out = nullptr;
        check(rdf_filter_frame(h_->h, low.nodes.data(), (int32_t)low.nodes.size(), root, &out));
        return derived(out, schema_);
    }
    // DataFrame::take / sort_by_indices: every column gathered by one index list in one pass -> one batch per column
    GpuFrame take(const ArrayRef& indices) const {
        const rdf_array idx = indices->view();
        rdf_frame* out = nullptr;
        check(rdf_take_frame(h_->h, &idx, &out));
        return derived(out, nullable_schema());
    }
    // DataFrame::sort: lexsort_to_indices over the criteria columns, then the take of every column by that order
    GpuFrame sort(const std::vector<DataFrame::SortCriteria>& criteria) const {
        if (criteria.empty()) throw DataFrameError(DataFrameError::ComputeError, "Sort criteria cannot be empty");
        std::vector<int32_t> cols;
        std::vector<rdf_sort_options> opts;
        for (auto& c : criteria) {
            const auto f = schema_.column_with_name(c.column);
            if (!f) throw DataFrameError(DataFrameError::ComputeError, "Cannot find column " + c.column);
            cols.push_back((int32_t)f->first);
            opts.push_back(rdf_sort_options{c.descending ? 1 : 0, 0});
        }
        rdf_frame* out = nullptr;
        check(rdf_sort_frame(h_->h, cols.data(), (int32_t)cols.size(), opts.data(), nullptr, &out));
        return derived(out, schema_);
    }
    // GroupAggregate(groups, [one aggregation]) -> columns: the grouping columns, "<fn>(<value>)", "count"
    GpuFrame group_aggregate(const std::vector<std::string>& groups, const std::string& value, plan::AggregateFunction fn, int64_t max_groups) const {
        using AF = plan::AggregateFunction;
        std::vector<int32_t> keys;
        Schema s;
        for (auto& g : groups) {
            const auto f = schema_.column_with_name(g);
            if (!f) throw DataFrameError(DataFrameError::ComputeError, "Grouping column " + g + " does not exist");
            keys.push_back((int32_t)f->first);
            s.fields.push_back(f->second);
        }
        int32_t vcol = -1;
        DataType vdt = DataType::Int64;
        if (fn != AF::Count) {
            const auto f = schema_.column_with_name(value);
            if (!f) throw DataFrameError(DataFrameError::ComputeError, "Aggregating column " + value + " does not exist");
            vcol = (int32_t)f->first; vdt = f->second.data_type;
        }
        if (fn == AF::Avg) throw DataFrameError(DataFrameError::ComputeError, "avg = sum / count on the caller's side");
        const int32_t agg = fn == AF::Sum ? RDF_AGG_SUM : fn == AF::Min ? RDF_AGG_MIN : fn == AF::Max ? RDF_AGG_MAX : RDF_AGG_COUNT;
        const bool fl = vdt == DataType::Float32 || vdt == DataType::Float64;
        const DataType odt = fn == AF::Count ? DataType::Int64 : fl ? DataType::Float64 : (agg != RDF_AGG_SUM && vdt == DataType::UInt64) ? DataType::UInt64 : DataType::Int64;
        const char* names[] = {"sum", "min", "max", "count"};
        s.fields.push_back(Field{std::string(names[agg]) + "(" + (fn == AF::Count ? std::string("*") : value) + ")", odt, true});
        s.fields.push_back(Field{"count", DataType::Int64, false});
        rdf_frame* out = nullptr;
        check(rdf_groupby_agg_frame(h_->h, keys.data(), (int32_t)keys.size(), vcol, agg, max_groups, &out));
        return derived(out, s);
    }
    // AggregateFunctions over a column of the pinned frame (sum / min / max / count in one fused pass)
    rdf_agg_result aggregate(const std::string& column) const {
        const auto f = schema_.column_with_name(column);
        if (!f) throw DataFrameError(DataFrameError::ComputeError, "Cannot find column " + column);
        if (num_columns() > 8) throw DataFrameError(DataFrameError::ComputeError, "programs run over frames of at most 8 columns: select first");
        rdf_expr_node n;
        std::memset(&n, 0, sizeof n);
        n.kind = RDF_NODE_COLUMN; n.column = (int32_t)f->first; n.lhs = n.rhs = -1;
        rdf_program p;
        std::memset(&p, 0, sizeof p);
        p.nodes = &n; p.nnodes = 1; p.filter_root = -1; p.nvalues = 1; p.value_roots[0] = 0; p.sink = RDF_SINK_AGG;
        rdf_agg_result r;
        check(rdf_pipeline_frame(&p, h_->h, nullptr, &r));
        return r;
    }
    // Arrow arrays back: one descriptor walk; the arrays alias the frame's buffers and keep the handle alive
    DataFrame to_dataframe() const {
        int32_t nc = 0; int64_t nch = 0, rows = 0;
        check(rdf_frame_info(h_->h, &nc, &nch, &rows));
        std::vector<Column> cols;
        std::vector<rdf_array> views((size_t)std::max<int64_t>(nch, 1));
        for (int32_t c = 0; c < nc; ++c) {
            check(rdf_frame_column(h_->h, c, views.data()));
            std::vector<ArrayRef> chunks;
            for (int64_t i = 0; i < nch; ++i) {
                auto a = std::make_shared<Array>();
                a->dtype = schema_.fields[(size_t)c].data_type;
                a->offset = views[(size_t)i].offset;
                a->length = views[(size_t)i].length;
                a->values = std::make_shared<DeviceBuffer>(const_cast<void*>(views[(size_t)i].values), a->length * (int64_t)type_size(a->dtype), h_);
                if (views[(size_t)i].validity) {
                    a->validity = std::make_shared<DeviceBuffer>(const_cast<uint8_t*>(views[(size_t)i].validity), (a->offset + a->length + 7) / 8, h_);
                    int64_t valid = 0; int32_t some = 0;
                    const rdf_array one = a->view_unknown_nulls();
                    check(rdf_count(&one, 1, &valid, &some));
                    a->null_count = a->length - valid;
                }
                chunks.push_back(a);
            }
            cols.push_back(Column::from_arrays(chunks, schema_.fields[(size_t)c]));
        }
        return DataFrame::from_columns(std::move(cols));
    }

  private:
    friend class ShardedFrame;
    struct Handle { rdf_frame* h; std::shared_ptr<DataFrame> pinned; ~Handle() { if (h) (void)rdf_frame_release(h); } };
    Schema schema_;
    std::shared_ptr<Handle> h_;
    GpuFrame derived(rdf_frame* out, const Schema& s) const {
        GpuFrame g;
        g.schema_ = s;
        g.h_ = std::shared_ptr<Handle>(new Handle{out, nullptr});      // the new frame owns its buffers
        return g;
    }
    Schema nullable_schema() const { Schema s = schema_; for (auto& f : s.fields) f.nullable = true; return s; }
};

// ------------------------------------------------------------------------------------------------
// ShardedFrame: one rank's shard of a DataFrame whose RecordBatches are spread over the GPUs of a node by row ranges
//  — the N-GPU form of what LazyFrame::evaluate (src/lazyframe.rs:311-315) drives.  Every reference kernel is
// per-chunk independent, so filter / take-within-shard / elementwise steps are the GpuFrame's own (no communication: the
// outputs stay sharded in rank order); AggregateFunctions fold the ranks' partials (rdf_agg_combine) and GroupAggregate — the
// panic! at src/evaluation.rs:73 — runs the library's exchange (rdf_groupby_agg_frame_dist: local aggregate, partial groups or
// rows to their owners over RCCL / peer copies, merge).  One ShardedFrame per rank, used by the thread that drives its GPU.
class Communicator {
  public:
    // one process (or thread) per GPU: rank 0 makes the id (unique_id()) and hands it to the others
    static std::array<uint8_t, RDF_COMM_ID_BYTES> unique_id() { std::array<uint8_t, RDF_COMM_ID_BYTES> id{}; check(rdf_comm_unique_id(id.data())); return id; }
    static std::shared_ptr<Communicator> init_rank(int world, int rank, const std::array<uint8_t, RDF_COMM_ID_BYTES>& id) {
        rdf_comm* c = nullptr;
        check(rdf_comm_init_rank(world, rank, id.data(), &c));
        return std::shared_ptr<Communicator>(new Communicator(c));
    }
    // a single process driving every GPU: communicator i belongs to the thread that called rdf_set_device(devices[i])
    static std::vector<std::shared_ptr<Communicator>> init_all(const std::vector<int32_t>& devices, rdf_comm_kind kind = RDF_COMM_RCCL) {
        std::vector<rdf_comm*> raw(devices.size(), nullptr);
        check(rdf_comm_init_all((int32_t)devices.size(), devices.data(), kind, raw.data()));
        std::vector<std::shared_ptr<Communicator>> out;
        for (rdf_comm* c : raw) out.push_back(std::shared_ptr<Communicator>(new Communicator(c)));
        return out;
    }
    ~Communicator() { if (c_) (void)rdf_comm_destroy(c_); }
    Communicator(const Communicator&) = delete;
    Communicator& operator=(const Communicator&) = delete;
    int world() const { int32_t w = 1; check(rdf_comm_info(c_, &w, nullptr, nullptr, nullptr, nullptr)); return w; }
    int rank() const { int32_t r = 0; check(rdf_comm_info(c_, nullptr, &r, nullptr, nullptr, nullptr)); return r; }
    void barrier() const { check(rdf_comm_barrier(c_)); }
    rdf_comm* raw() const { return c_; }

  private:
    explicit Communicator(rdf_comm* c) : c_(c) {}
    rdf_comm* c_;
};

class ShardedFrame {
  public:
    ShardedFrame(GpuFrame local, std::shared_ptr<Communicator> comm) : local_(std::move(local)), comm_(std::move(comm)) {}
    const GpuFrame& local() const { return local_; }
    const Communicator& comm() const { return *comm_; }
    // shard-local operators: row order across ranks = rank order
    ShardedFrame filter(const FilterRef& condition) const { return ShardedFrame(local_.filter(condition), comm_); }
    // rows of the whole frame
    int64_t num_rows() const {
        int64_t mine = local_.num_rows(), total = 0;
        std::vector<int64_t> all((size_t)comm_->world());
        check(rdf_comm_allgather(comm_->raw(), &mine, 8, all.data()));
        for (int64_t x : all) total += x;
        return total;
    }
    // AggregateFunctions::{sum, min, max, count} of a column over ALL shards: identical bits on every rank
    rdf_agg_result aggregate(const std::string& column) const {
        rdf_agg_result r = local_.aggregate(column);
        check(rdf_agg_combine(comm_->raw(), &r, 1));
        return r;
    }
    // GroupAggregate(groups = [one Int64 / UInt64 column], [one aggregation]) over all shards -> the groups THIS rank owns
    // (the union over the ranks is the result; owner = hash(key) % world); columns: key, "<fn>(<value>)", "count"
    GpuFrame group_aggregate(const std::string& group, const std::string& value, plan::AggregateFunction fn, int64_t max_groups,
                             rdf_exchange_mode exchange = RDF_EXCHANGE_AUTO, rdf_exchange_stats* stats = nullptr) const {
        using AF = plan::AggregateFunction;
        const Schema& sc = local_.schema();
        const auto kf = sc.column_with_name(group);
        if (!kf) throw DataFrameError(DataFrameError::ComputeError, "Grouping column " + group + " does not exist");
        if (fn == AF::Avg) throw DataFrameError(DataFrameError::ComputeError, "avg = sum / count on the caller's side");
        int32_t vcol = -1;
        DataType vdt = DataType::Int64;
        if (fn != AF::Count) {
            const auto f = sc.column_with_name(value);
            if (!f) throw DataFrameError(DataFrameError::ComputeError, "Aggregating column " + value + " does not exist");
            vcol = (int32_t)f->first; vdt = f->second.data_type;
        }
        const int32_t agg = fn == AF::Sum ? RDF_AGG_SUM : fn == AF::Min ? RDF_AGG_MIN : fn == AF::Max ? RDF_AGG_MAX : RDF_AGG_COUNT;
        const bool fl = vdt == DataType::Float32 || vdt == DataType::Float64;
        const DataType odt = fn == AF::Count ? DataType::Int64 : fl ? DataType::Float64 : (agg != RDF_AGG_SUM && vdt == DataType::UInt64) ? DataType::UInt64 : DataType::Int64;
        const char* names[] = {"sum", "min", "max", "count"};
        Schema s;
        s.fields.push_back(kf->second);
        s.fields.push_back(Field{std::string(names[agg]) + "(" + (fn == AF::Count ? std::string("*") : value) + ")", odt, true});
        s.fields.push_back(Field{"count", DataType::Int64, false});
        rdf_frame* out = nullptr;
        check(rdf_groupby_agg_frame_dist(comm_->raw(), local_.h_->h, (int32_t)kf->first, vcol, agg, max_groups, exchange, &out, stats));
        return local_.derived(out, s);
    }

  private:
    GpuFrame local_;
    std::shared_ptr<Communicator> comm_;
};

// ------------------------------------------------------------------------------------------------
// ScalarFunctions / AggregateFunctions over device columns (chunk list in, chunk list out)

struct ScalarFunctions {
    static std::vector<ArrayRef> binary(int32_t op, const std::vector<ArrayRef>& left, const std::vector<ArrayRef>& right) {
        if (left.size() != right.size()) throw DataFrameError(DataFrameError::ComputeError, "chunk lists differ in length");
        std::vector<rdf_array> a, b;
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        for (size_t i = 0; i < left.size(); ++i) {
            a.push_back(left[i]->view()); b.push_back(right[i]->view());
            outs.push_back(Array::make_out(left[i]->dtype, left[i]->length, left[i]->validity || right[i]->validity, left[i]->host));
            ov.push_back(outs.back()->out_view(left[i]->length));
        }
        check(rdf_binary(op, a.data(), b.data(), (int64_t)a.size(), ov.data()));
        return finish(outs, ov);
    }
    static std::vector<ArrayRef> unary(int32_t op, const std::vector<ArrayRef>& arr) {
        std::vector<rdf_array> a;
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        for (auto& x : arr) { a.push_back(x->view()); outs.push_back(Array::make_out(x->dtype, x->length, x->validity != nullptr, x->host)); ov.push_back(outs.back()->out_view(x->length)); }
        check(rdf_unary(op, a.data(), (int64_t)a.size(), ov.data()));
        return finish(outs, ov);
    }
    static std::vector<ArrayRef> cast(const std::vector<ArrayRef>& arr, DataType to) {
        std::vector<rdf_array> a;
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        for (auto& x : arr) { a.push_back(x->view()); outs.push_back(Array::make_out(to, x->length, true, x->host)); ov.push_back(outs.back()->out_view(x->length)); }   // (a value the target type cannot hold becomes NULL: always a bitmap)
        check(rdf_cast(a.data(), (int64_t)a.size(), ov.data()));
        return finish(outs, ov);
    }
    // src/functions/scalar.rs:267-273: hour of day of a temporal column, given as its Int32 / Int64 storage + time unit
    static std::vector<ArrayRef> hour(const std::vector<ArrayRef>& arr, rdf_time_unit unit) {
        std::vector<rdf_array> a;
        std::vector<std::shared_ptr<Array>> outs;
        std::vector<rdf_out> ov;
        for (auto& x : arr) { a.push_back(x->view()); outs.push_back(Array::make_out(DataType::Int32, x->length, x->validity != nullptr, x->host)); ov.push_back(outs.back()->out_view(x->length)); }
        check(rdf_hour(a.data(), (int64_t)a.size(), (int32_t)unit, ov.data()));
        return finish(outs, ov);
    }
    // src/functions/scalar.rs:16-103
    static std::vector<ArrayRef> add(const std::vector<ArrayRef>& l, const std::vector<ArrayRef>& r) { return binary(RDF_OP_ADD, l, r); }
    static std::vector<ArrayRef> subtract(const std::vector<ArrayRef>& l, const std::vector<ArrayRef>& r) { return binary(RDF_OP_SUB, l, r); }
    static std::vector<ArrayRef> multiply(const std::vector<ArrayRef>& l, const std::vector<ArrayRef>& r) { return binary(RDF_OP_MUL, l, r); }
    static std::vector<ArrayRef> divide(const std::vector<ArrayRef>& l, const std::vector<ArrayRef>& r) { return binary(RDF_OP_DIV, l, r); }
    // :106-452
    static std::vector<ArrayRef> abs(const std::vector<ArrayRef>& a) { return unary(RDF_OP_ABS, a); }
    static std::vector<ArrayRef> sin(const std::vector<ArrayRef>& a) { return unary(RDF_OP_SIN, a); }
    static std::vector<ArrayRef> cos(const std::vector<ArrayRef>& a) { return unary(RDF_OP_COS, a); }
    static std::vector<ArrayRef> tan(const std::vector<ArrayRef>& a) { return unary(RDF_OP_TAN, a); }
    static std::vector<ArrayRef> cot(const std::vector<ArrayRef>& a) { return unary(RDF_OP_COT, a); }
    static std::vector<ArrayRef> sec(const std::vector<ArrayRef>& a) { return unary(RDF_OP_SEC, a); }
    static std::vector<ArrayRef> csc(const std::vector<ArrayRef>& a) { return unary(RDF_OP_CSC, a); }
    static std::vector<ArrayRef> acos(const std::vector<ArrayRef>& a) { return unary(RDF_OP_ACOS, a); }

  private:
    static std::vector<ArrayRef> finish(std::vector<std::shared_ptr<Array>>& outs, std::vector<rdf_out>& ov) {
        std::vector<ArrayRef> res;
        for (size_t i = 0; i < outs.size(); ++i) { outs[i]->length = ov[i].length; outs[i]->null_count = ov[i].null_count; res.push_back(outs[i]); }
        return res;
    }
};

struct AggregateFunctions {  // src/functions/aggregate.rs:12-93
    template <class T> static std::optional<T> call(rdf_status (*fn)(const rdf_array*, int64_t, void*, int32_t*), const ChunkedArray& c) {
        const auto v = c.views();
        T out{}; int32_t some = 0;
        check(fn(v.data(), (int64_t)v.size(), &out, &some));
        return some ? std::optional<T>(out) : std::nullopt;
    }
    template <class T> static std::optional<T> sum(const ChunkedArray& c) { return call<T>(rdf_sum, c); }
    template <class T> static std::optional<T> min(const ChunkedArray& c) { return call<T>(rdf_min, c); }
    template <class T> static std::optional<T> max(const ChunkedArray& c) { return call<T>(rdf_max, c); }
    static std::optional<int64_t> count(const ChunkedArray& c) {
        const auto v = c.views(); int64_t out = 0; int32_t some = 0;
        check(rdf_count(v.data(), (int64_t)v.size(), &out, &some));
        return some ? std::optional<int64_t>(out) : std::nullopt;
    }
    static std::optional<double> avg(const ChunkedArray& c) {
        const auto v = c.views(); double out = 0; int32_t some = 0;
        check(rdf_avg(v.data(), (int64_t)v.size(), &out, &some));
        return some ? std::optional<double>(out) : std::nullopt;
    }
};

// ------------------------------------------------------------------------------------------------
// ListArray over a primitive child + ArrayFunctions (src/functions/array.rs:15-399)

struct ListArray {
    ArrayRef offsets;  // Int32 value_offsets, len() + 1 entries; its validity buffer is the LIST validity (len() bits)
    ArrayRef child;

    int64_t len() const { return offsets->length - 1; }
    ArrayRef values() const { return child; }
    DataType value_type() const { return child->dtype; }
    rdf_list_array view() const { rdf_list_array l; l.offsets = offsets->view(); l.values = child->view(); return l; }
    std::vector<int32_t> value_offsets() const { return offsets->values_to_host<int32_t>(); }
    int32_t value_offset(int64_t i) const { return offsets->value<int32_t>(i); }
    int32_t value_length(int64_t i) const { return offsets->value<int32_t>(i + 1) - offsets->value<int32_t>(i); }
    bool is_null(int64_t i) const { return offsets->validity && !offsets->bits_to_host(offsets->validity)[(size_t)i]; }

    // ArrayData::builder(List(..)).add_buffer(value_offsets).add_child_data(values) of the reference's tests (array.rs:433-440)
    static ListArray from_parts(const std::vector<int32_t>& value_offsets, ArrayRef values, const std::vector<bool>* valid = nullptr) {
        return ListArray{Array::from_vec<int32_t>(value_offsets, valid), std::move(values)};
    }
    template <class T>
    static ListArray from_rows(const std::vector<std::optional<std::vector<T>>>& rows) {
        std::vector<int32_t> off{0};
        std::vector<T> vals;
        std::vector<bool> valid;
        bool any_null = false;
        for (auto& r : rows) {
            valid.push_back(r.has_value());
            any_null |= !r.has_value();
            if (r) vals.insert(vals.end(), r->begin(), r->end());
            off.push_back((int32_t)vals.size());
        }
        return from_parts(off, Array::from_vec<T>(vals), any_null ? &valid : nullptr);
    }
    template <class T>
    std::vector<std::vector<T>> rows_to_host() const {
        const auto off = value_offsets();
        const auto vals = child->values_to_host<T>();
        std::vector<std::vector<T>> out;
        for (size_t i = 0; i + 1 < off.size(); ++i) out.emplace_back(vals.begin() + off[i], vals.begin() + off[i + 1]);
        return out;
    }
};

struct ArrayFunctions {
    // array.rs:15-37: NULL list -> NULL, else whether the row's slice holds `val`
    template <class T> static ArrayRef array_contains(const ListArray& array, T val) {
        expect_child<T>(array, "array_contains");
        auto out = Array::make_out(DataType::Boolean, array.len(), true);
        return one(array, out, [&](const rdf_list_array* l, rdf_out* o) { return rdf_list_contains(l, &val, o); });
    }
    // array.rs:233-260: 1-based position of the first match, 0 when absent or the list is NULL
    template <class T> static ArrayRef array_position(const ListArray& array, T val) {
        expect_child<T>(array, "array_position");
        auto out = Array::make_out(DataType::Int32, array.len(), false);
        return one(array, out, [&](const rdf_list_array* l, rdf_out* o) { return rdf_list_position(l, &val, o); });
    }
    // array.rs:182-231
    template <class T> static ArrayRef array_max(const ListArray& array) {
        expect_child<T>(array, "array_max");
        auto out = Array::make_out(array.value_type(), array.len(), true);
        return one(array, out, [&](const rdf_list_array* l, rdf_out* o) { return rdf_list_max(l, o); });
    }
    template <class T> static ArrayRef array_min(const ListArray& array) {
        expect_child<T>(array, "array_min");
        auto out = Array::make_out(array.value_type(), array.len(), true);
        return one(array, out, [&](const rdf_list_array* l, rdf_out* o) { return rdf_list_min(l, o); });
    }
    // array.rs:262-292
    template <class T> static ListArray array_remove(const ListArray& array, T val) {
        expect_child<T>(array, "array_remove");
        return rebuild(array, array.child->length, [&](const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return rdf_list_remove(l, &val, oo, ov); });
    }
    // array.rs:39-65
    template <class T> static ListArray array_distinct(const ListArray& array) {
        expect_child<T>(array, "array_distinct");
        return rebuild(array, array.child->length, [&](const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return rdf_list_distinct(l, oo, ov); });
    }
    // array.rs:66-153,356-399
    template <class T> static ListArray array_except(const ListArray& a, const ListArray& b) {
        expect_child<T>(a, "array_except");
        const rdf_list_array lb = b.view();
        return rebuild(a, a.child->length, [&](const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return rdf_list_except(l, &lb, oo, ov); });
    }
    template <class T> static ListArray array_intersect(const ListArray& a, const ListArray& b) {
        expect_child<T>(a, "array_intersect");
        const rdf_list_array lb = b.view();
        return rebuild(a, a.child->length, [&](const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return rdf_list_intersect(l, &lb, oo, ov); });
    }
    template <class T> static ListArray array_union(const ListArray& a, const ListArray& b) {
        expect_child<T>(a, "array_union");
        const rdf_list_array lb = b.view();
        return rebuild(a, a.child->length + b.child->length, [&](const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return rdf_list_union(l, &lb, oo, ov); });
    }
    // array.rs:294-326
    template <class T> static ListArray array_repeat(const ListArray& array, int32_t count) {
        expect_child<T>(array, "array_repeat");
        return rebuild(array, array.child->length * (int64_t)(count > 0 ? count : 0), [&](const rdf_list_array* l, rdf_out* oo, rdf_out* ov) { return rdf_list_repeat(l, count, oo, ov); });
    }
    // array.rs:328-354: the value_offsets stay, every slice ascending
    template <class T> static ListArray array_sort(const ListArray& array) {
        expect_child<T>(array, "array_sort");
        auto off = array.value_offsets();
        const int64_t total = off.empty() ? 0 : (int64_t)off.back() - off.front();
        auto vals = Array::make_out(array.value_type(), total, false);
        const rdf_list_array l = array.view();
        rdf_out ov = vals->out_view(total);
        check(rdf_list_sort(&l, &ov));
        vals->length = ov.length;
        const int32_t first = off.empty() ? 0 : off.front();
        for (auto& o : off) o -= first;   // the sorted values start at the first row's slice
        std::vector<bool> valid;
        if (array.offsets->validity) valid = array.offsets->bits_to_host(array.offsets->validity), valid.resize((size_t)array.len());
        return ListArray::from_parts(off, vals, array.offsets->validity ? &valid : nullptr);
    }

  private:
    template <class T> static void expect_child(const ListArray& a, const char* fn) {
        if (a.value_type() != TypeOf<T>::value) throw DataFrameError(DataFrameError::ComputeError, std::string(fn) + ": the list's value type is not the requested one");
    }
    template <class F> static ArrayRef one(const ListArray& array, std::shared_ptr<Array> out, F call) {
        const rdf_list_array l = array.view();
        rdf_out o = out->out_view(array.len());
        check(call(&l, &o));
        out->length = o.length;
        out->null_count = o.null_count;
        return out;
    }
    template <class F> static ListArray rebuild(const ListArray& array, int64_t capacity, F call) {
        auto off = Array::make_out(DataType::Int32, array.len() + 1, false);
        auto vals = Array::make_out(array.value_type(), capacity, false);
        const rdf_list_array l = array.view();
        rdf_out oo = off->out_view(array.len() + 1), ov = vals->out_view(capacity);
        check(call(&l, &oo, &ov));
        off->length = oo.length;
        vals->length = ov.length;
        return ListArray{off, vals};
    }
};

// ------------------------------------------------------------------------------------------------
// Evaluate (src/evaluation.rs:54-323), fusing

inline int32_t scalar_function_op(plan::ScalarFunction f) {
    using SF = plan::ScalarFunction;
    switch (f) {
        case SF::Add: return RDF_OP_ADD; case SF::Subtract: return RDF_OP_SUB; case SF::Multiply: return RDF_OP_MUL;
        case SF::Divide: return RDF_OP_DIV; case SF::Abs: return RDF_OP_ABS; case SF::Sine: return RDF_OP_SIN;
        case SF::Cosine: return RDF_OP_COS; case SF::Tangent: return RDF_OP_TAN;
        case SF::Cotangent: return RDF_OP_COT; case SF::Secant: return RDF_OP_SEC; case SF::Cosecant: return RDF_OP_CSC;
        default: throw DataFrameError(DataFrameError::ComputeError, std::string("Scalar Function ") + plan::scalar_function_name(f) + " not supported");
    }
}

class Evaluate {
  public:
    // Evaluate::calculate, eager: ONE Calculation materialised as a new column (drop-in for :97-323)
    static DataFrame calculate(const DataFrame& frame, const plan::Calculation& calc) {
        Evaluate ev(frame);
        ev.step_calculate(calc);
        return ev.flush();
    }
    // Evaluate::evaluate (:66-96): computations newest-first (Expression::unroll order), each applied to
    // the running frame.  Calculate / Filter / Select / Drop / Rename stay lazy and are flushed as fused
    // passes; Limit and the end of the plan materialise; GroupAggregate with no groups is the fused
    // filter -> aggregate pass.
    static DataFrame evaluate(const DataFrame& frame, const std::vector<plan::Computation>& comps) {
        Evaluate ev(frame);
        for (auto c = comps.rbegin(); c != comps.rend(); ++c)
            for (auto& t : c->transformations) ev.step(t);
        return ev.flush();
    }

  private:
    struct Lazy { std::string name; DataType dtype; ExprRef def; };  // def == nullptr: column `source` of base_
    struct Entry { std::string name; DataType dtype; ExprRef def; std::string source; };

    explicit Evaluate(const DataFrame& f) : base_(f) {
        for (auto& fld : f.schema().fields) cols_.push_back(Entry{fld.name, fld.data_type, nullptr, fld.name});
    }
    Entry* find(const std::string& n) { for (auto& e : cols_) if (e.name == n) return &e; return nullptr; }
    ExprRef ref(const std::string& n) {
        Entry* e = find(n);
        if (!e) throw DataFrameError(DataFrameError::ComputeError, "Column not found by name: " + n);
        return e->def ? e->def : Expr::col(e->source);
    }
    void put(const std::string& name, DataType dt, ExprRef def) {  // with_column: replace-at-end semantics
        for (size_t i = 0; i < cols_.size(); ++i) if (cols_[i].name == name) { cols_.erase(cols_.begin() + i); break; }
        cols_.push_back(Entry{name, dt, std::move(def), ""});
    }

    void step(const plan::Transformation& t) {
        using T = plan::Transformation;
        switch (t.kind) {
            case T::Calculate: step_calculate(t.calc); break;
            case T::Filter: step_filter(t.filter); break;
            case T::Limit:
                // the optimiser's limit push-down (src/optimiser.rs:56-75): pending Calculate steps are row-wise, so the limit
                // is taken on the source columns (zero-copy slices) and the lazy columns are then computed for those rows only;
                // a pending filter changes which rows come first, so it is applied before the limit
                if (!pending_) base_ = base_.limit(t.limit);
                else { DataFrame f = flush(); reset(f.limit(t.limit)); }
                break;
            case T::Select: {
                std::vector<Entry> keep;
                for (auto& e : cols_) for (auto& n : t.names) if (e.name == n) { keep.push_back(e); break; }
                cols_ = keep;
            } break;
            case T::Drop: {
                std::vector<Entry> keep;
                for (auto& e : cols_) { bool d = false; for (auto& n : t.names) d |= e.name == n; if (!d) keep.push_back(e); }
                cols_ = keep;
            } break;
            case T::GroupAggregate: step_aggregate(t); break;
            case T::Sort: {
                DataFrame f = flush();
                std::vector<DataFrame::SortCriteria> cr;
                for (auto& n : t.names) cr.push_back(DataFrame::SortCriteria{n, false, false});
                for (size_t i = 0; i < t.sort_descending.size() && i < cr.size(); ++i) cr[i].descending = t.sort_descending[i];
                reset(f.sort(cr));
            } break;
            case T::Join: throw DataFrameError(DataFrameError::ComputeError, "Join inside a computation list: LazyFrame::join evaluates both sides and joins the frames");
            default: throw DataFrameError(DataFrameError::ComputeError, "Read inside evaluate: pass the frame in");
        }
    }

    void step_calculate(const plan::Calculation& calc) {
        using F = plan::Function;
        std::vector<ExprRef> in;
        for (auto& c : calc.inputs) in.push_back(ref(c.name));
        switch (calc.function.kind) {
            case F::Scalar: {
                const DataType dt = calc.output.data_type;
                const auto sf = calc.function.scalar;
                const bool binary = sf == plan::ScalarFunction::Add || sf == plan::ScalarFunction::Subtract || sf == plan::ScalarFunction::Multiply || sf == plan::ScalarFunction::Divide;
                if (binary) {
                    if (dt == DataType::Int8 || dt == DataType::UInt8 || dt == DataType::Boolean || dt == DataType::Utf8)
                        throw DataFrameError(DataFrameError::ComputeError, "Unsupported operation");  // evaluation.rs:239
                    if (in.size() != 2) throw DataFrameError(DataFrameError::ComputeError, "binary scalar function expects 2 inputs");
                    if (sf == plan::ScalarFunction::Divide && pending_) { DataFrame f = flush(); reset(f); in = {ref(calc.inputs[0].name), ref(calc.inputs[1].name)}; }
                    put(calc.output.name, dt, Expr::make(scalar_function_op(sf), in[0], in[1]));
                } else {
                    if (!is_float(dt)) throw DataFrameError(DataFrameError::ComputeError, std::string("Expecting float datatype for operation, found ") + type_name(dt));  // :282-284
                    put(calc.output.name, dt, Expr::make(scalar_function_op(sf), in[0]));
                }
            } break;
            case F::Cast: put(calc.output.name, calc.output.data_type, Expr::make(RDF_OP_CAST, in[0], nullptr, (int32_t)calc.output.data_type)); break;
            case F::Rename: {
                Entry* e = find(calc.inputs[0].name);
                if (!e) throw DataFrameError(DataFrameError::NoneError, "column " + calc.inputs[0].name + " not found");
                e->name = calc.output.name;
            } break;
            case F::Filter: step_filter(calc.function.filter); break;
        }
    }
    void step_filter(const FilterRef& f) {
        ExprRef p = filter_to_expr(f, [this](const std::string& n) { return ref(n); });
        pending_ = pending_ ? Expr::make(RDF_OP_AND, pending_, p) : p;
    }

    // fused filter -> aggregates (GroupAggregate with no grouping columns)
    // GroupAggregate WITH grouping columns: the reference plans it (Dataset::try_aggregate, src/expression.rs:114-221)
    // and panics on execution (src/evaluation.rs:73).  Here: one integer grouping column, Sum / Count / Avg per group
    // through rdf_groupby_sum; rows of the result are ordered by key (NULL group last) so every aggregate column lines up.
    void step_group_aggregate(const plan::Transformation& t) {
        using AF = plan::AggregateFunction;
        if (fused_dense_group_aggregate(t)) return;   // small dense key domain(s): filter + expressions + grouping in one pass
        DataFrame f = flush();   // lazy columns materialised, pending filters applied
        const size_t nk = t.names.size();
        if (nk < 1 || nk > RDF_MAX_GROUP_KEYS) throw DataFrameError(DataFrameError::ComputeError, "GroupAggregate: 1.." + std::to_string(RDF_MAX_GROUP_KEYS) + " grouping columns");
        std::vector<const Column*> kcs;
        for (auto& n : t.names) {
            kcs.push_back(&f.column_by_name(n));
            if (!is_integer(kcs.back()->data_type())) throw DataFrameError(DataFrameError::ComputeError, "GroupAggregate: the grouping columns must be integer columns");
        }
        const size_t nch = kcs[0]->data().num_chunks();
        std::vector<rdf_array> keys;          // [k * nch + i]
        std::vector<bool> key_nulls(nk, false);
        for (size_t k = 0; k < nk; ++k)
            for (auto& a : kcs[k]->data().chunks()) { keys.push_back(a->view()); if (a->validity != nullptr) key_nulls[k] = true; }
        const int64_t nrows = (int64_t)f.num_rows();
        std::vector<Column> out_cols;
        if (nk == 1 && dense_group_aggregate(f, *kcs[0], t, out_cols)) { reset(DataFrame::from_columns(out_cols)); return; }
        // Sparse or NULL-holding keys: one rdf_groupby_agg per aggregation (hash GROUP BY; several grouping columns are
        // range-compressed into one 64-bit key on the device), results ordered by the grouping columns
        auto one = [&](AF fn, const std::string& col) {
            if (fn != AF::Sum && fn != AF::Count && fn != AF::Avg && fn != AF::Min && fn != AF::Max) throw DataFrameError(DataFrameError::ComputeError, "Aggregation not yet supported");
            const Column& vc = f.column_by_name(col);
            const DataType vdt = vc.data_type();
            if (!(is_integer(vdt) || is_float(vdt))) throw DataFrameError(DataFrameError::ComputeError, "Aggregating column must be numeric");
            std::vector<rdf_array> vals;
            bool val_nulls = false;
            for (auto& a : vc.data().chunks()) { vals.push_back(a->view()); val_nulls |= a->validity != nullptr; }
            const int32_t agg = fn == AF::Min ? RDF_AGG_MIN : fn == AF::Max ? RDF_AGG_MAX : RDF_AGG_SUM;
            const bool extremum = agg != RDF_AGG_SUM;
            const DataType sdt = is_float(vdt) ? DataType::Float64 : (extremum && vdt == DataType::UInt64) ? DataType::UInt64 : DataType::Int64;
            int64_t mg = std::max<int64_t>(1, std::min<int64_t>(nrows, (int64_t)1 << 20));
            std::vector<std::shared_ptr<Array>> ok(nk);
            std::shared_ptr<Array> os, oc;
            for (;;) {   // the number of groups is not known in advance: grow the promise until it holds
                std::vector<rdf_out> vk(nk);
                const bool on_host = f.is_host();      // host-resident batches: the library streams them, the groups come back to the host
                for (size_t k = 0; k < nk; ++k) { ok[k] = Array::make_out(kcs[k]->data_type(), mg + 2, key_nulls[k], on_host); vk[k] = ok[k]->out_view(mg + 2); }
                os = Array::make_out(sdt, mg + 2, extremum && val_nulls, on_host);
                oc = Array::make_out(DataType::Int64, mg + 2, false, on_host);
                rdf_out vs = os->out_view(mg + 2), vcn = oc->out_view(mg + 2);
                const rdf_status st = rdf_groupby_agg(keys.data(), (int32_t)nk, vals.data(), (int64_t)nch, agg, mg, vk.data(), &vs, &vcn);
                if (st == RDF_MEMORY_ERROR && mg < nrows) { mg = std::min<int64_t>(nrows, mg * 16); continue; }
                check(st);
                for (size_t k = 0; k < nk; ++k) { ok[k]->length = vk[k].length; ok[k]->null_count = vk[k].null_count; }
                os->length = oc->length = vs.length;
                os->null_count = vs.null_count;
                break;
            }
            std::vector<Column> gc;
            std::vector<DataFrame::SortCriteria> order;
            for (size_t k = 0; k < nk; ++k) {
                gc.push_back(Column::from_arrays({ok[k]}, Field{"k" + std::to_string(k), kcs[k]->data_type(), true}));
                order.push_back(DataFrame::SortCriteria{"k" + std::to_string(k), false, false});
            }
            gc.push_back(Column::from_arrays({os}, Field{"s", sdt, true}));
            gc.push_back(Column::from_arrays({oc}, Field{"c", DataType::Int64, false}));
            DataFrame g = DataFrame::from_columns(gc).sort(order);
            if (out_cols.empty())
                for (size_t k = 0; k < nk; ++k) out_cols.push_back(Column::from_arrays(g.column_by_name("k" + std::to_string(k)).data().chunks(), Field{t.names[k], kcs[k]->data_type(), true}));
            const std::vector<ArrayRef> sums = g.column_by_name("s").data().chunks(), counts = g.column_by_name("c").data().chunks();
            if (fn == AF::Sum) {
                out_cols.push_back(Column::from_arrays(sdt == vdt ? sums : ScalarFunctions::cast(sums, vdt), Field{"sum(" + col + ")", vdt, true}));
            } else if (fn == AF::Min || fn == AF::Max) {   // typed like the input (try_aggregate): the extremum of a group always fits
                out_cols.push_back(Column::from_arrays(sdt == vdt ? sums : ScalarFunctions::cast(sums, vdt), Field{std::string(fn == AF::Min ? "min(" : "max(") + col + ")", vdt, true}));
            } else if (fn == AF::Count) {
                out_cols.push_back(Column::from_arrays(ScalarFunctions::cast(counts, DataType::UInt32), Field{"count(" + col + ")", DataType::UInt32, true}));
            } else {   // avg = sum / count, NULL for a group without a non-null value (AggregateFunctions::avg, src/functions/aggregate.rs:32-65)
                std::vector<ArrayRef> out;
                const std::vector<ArrayRef> fs = sdt == DataType::Float64 ? sums : ScalarFunctions::cast(sums, DataType::Float64);
                for (size_t i = 0; i < fs.size(); ++i) {
                    const std::vector<double> sv = fs[i]->values_to_host<double>();
                    const std::vector<int64_t> cv = counts[i]->values_to_host<int64_t>();
                    std::vector<double> m(sv.size());
                    std::vector<bool> valid(sv.size());
                    for (size_t r = 0; r < sv.size(); ++r) { valid[r] = cv[r] > 0; m[r] = cv[r] > 0 ? sv[r] / (double)cv[r] : 0.0; }
                    out.push_back(Array::from_vec(m, &valid));
                }
                out_cols.push_back(Column::from_arrays(out, Field{"avg(" + col + ")", DataType::Float64, true}));
            }
        };
        for (auto& a : t.aggregations)
            for (auto& c : a.columns) one(a.function, c);
        if (out_cols.empty()) throw DataFrameError(DataFrameError::ComputeError, "GroupAggregate without aggregations");
        reset(DataFrame::from_columns(out_cols));
    }

    // GroupAggregate over a SMALL DENSE key domain (dictionary codes, flags: TPC-H Q1's returnflag x linestatus): the pending
    // filter, the lazy value expressions, the grouping and every aggregation of the step in ONE pass over the rows through
    // rdf_group_pipeline (group id = mixed-radix number of the keys' offsets from their minima), instead of materialising the
    // computed columns and running one hash GROUP BY per aggregation.  Same output as the hash path: groups in ascending key
    // order, the NULL key (single grouping column only) last.  -> false when the shape does not fit the dense kernel
    // (RDF_MAX_GROUP_SLOTS / RDF_MAX_GROUP_VALUES, NULLs in one of several grouping columns, a non-integer key).
    struct DenseKey { std::string column, out_name; DataType dtype; };               // `column`: its name in `frame`
    struct DenseVal { plan::AggregateFunction fn; std::string name; DataType dtype; ExprRef e; };
    static bool dense_groups(const DataFrame& frame, const std::vector<DenseKey>& keys, const ExprRef& filter, const std::vector<DenseVal>& vals,
                             std::vector<Column>& out_cols) {
        using AF = plan::AggregateFunction;
        if (keys.empty() || vals.empty()) return false;
        // the dense kernel folds sums and counts; per-group Min / Max go through the hash path (rdf_groupby_agg)
        for (auto& v : vals) if (v.fn != AF::Sum && v.fn != AF::Count && v.fn != AF::Avg) return false;
        std::vector<int64_t> kmin(keys.size()), dom(keys.size());
        uint64_t total = 1;
        for (size_t i = 0; i < keys.size(); ++i) {
            if (!is_integer(keys[i].dtype)) return false;
            const Column& kc = frame.column_by_name(keys[i].column);
            const std::vector<rdf_array> kv = kc.data().views();
            if (keys.size() > 1) for (auto& a : kv) if (a.validity && a.null_count != 0) return false;   // NULLs would need one extra code per column
            unsigned char lo[8] = {0}, hi[8] = {0};
            int32_t some_lo = 0, some_hi = 0;
            check(rdf_min(kv.data(), (int64_t)kv.size(), lo, &some_lo));
            check(rdf_max(kv.data(), (int64_t)kv.size(), hi, &some_hi));
            if (!some_lo || !some_hi) return false;   // no non-NULL key at all
            auto as_i64 = [&](const unsigned char* p, bool& ok) -> int64_t {
                ok = true;
                switch (keys[i].dtype) {
                    case DataType::Int8: { int8_t v; std::memcpy(&v, p, 1); return v; }
                    case DataType::Int16: { int16_t v; std::memcpy(&v, p, 2); return v; }
                    case DataType::Int32: { int32_t v; std::memcpy(&v, p, 4); return v; }
                    case DataType::Int64: { int64_t v; std::memcpy(&v, p, 8); return v; }
                    case DataType::UInt8: { uint8_t v; std::memcpy(&v, p, 1); return v; }
                    case DataType::UInt16: { uint16_t v; std::memcpy(&v, p, 2); return v; }
                    case DataType::UInt32: { uint32_t v; std::memcpy(&v, p, 4); return v; }
                    default: { uint64_t v; std::memcpy(&v, p, 8); ok = v <= (uint64_t)INT64_MAX; return (int64_t)v; }
                }
            };
            bool ok_lo = false, ok_hi = false;
            const int64_t mn = as_i64(lo, ok_lo), mx = as_i64(hi, ok_hi);
            if (!ok_lo || !ok_hi || (uint64_t)mx - (uint64_t)mn >= (uint64_t)RDF_MAX_GROUP_SLOTS) return false;
            kmin[i] = mn;
            dom[i] = mx - mn + 1;
            total *= (uint64_t)dom[i];
            if (total > (uint64_t)RDF_MAX_GROUP_SLOTS) return false;
        }
        // the distinct value expressions of the step (Sum / Count / Avg of one column share its per-group sum and count)
        std::vector<std::string> vnames;
        std::vector<ExprRef> vexprs;
        for (auto& v : vals) {
            if (!(is_integer(v.dtype) || is_float(v.dtype))) throw DataFrameError(DataFrameError::ComputeError, "Aggregating column must be numeric");
            if (std::find(vnames.begin(), vnames.end(), v.name) == vnames.end()) { vnames.push_back(v.name); vexprs.push_back(v.e); }
        }
        if (vnames.size() > (size_t)RDF_MAX_GROUP_VALUES || (size_t)(total + 1) * vnames.size() > (size_t)RDF_MAX_GROUP_SLOTS) return false;
        Lowered low;
        const int filter_root = filter ? low.add(filter) : -1;
        ExprRef gid;
        for (size_t i = 0; i < keys.size(); ++i) {
            ExprRef d = Expr::make(RDF_OP_SUB, Expr::make(RDF_OP_CAST, Expr::col(keys[i].column), nullptr, RDF_I64), Expr::literal(Scalar((int64_t)kmin[i]), RDF_I64));
            gid = i == 0 ? d : Expr::make(RDF_OP_ADD, Expr::make(RDF_OP_MUL, gid, Expr::literal(Scalar((int64_t)dom[i]), RDF_I64)), d);
        }
        const int group_root = low.add(gid);
        int32_t value_roots[RDF_MAX_GROUP_VALUES] = {0};
        for (size_t v = 0; v < vexprs.size(); ++v) value_roots[v] = low.add(vexprs[v]);
        std::vector<rdf_array> cols;
        for (auto& cn : low.columns) for (auto& a : frame.column_by_name(cn).data().chunks()) cols.push_back(a->view());
        const size_t S = (size_t)total + 1;
        std::vector<rdf_group_result> res(S * vnames.size());
        std::vector<int64_t> rows(S, 0);
        check(rdf_group_pipeline(low.nodes.data(), (int32_t)low.nodes.size(), filter_root, group_root, (int32_t)total, value_roots, (int32_t)vnames.size(),
                                 cols.data(), (int32_t)low.columns.size(), (int64_t)frame.num_chunks(), res.data(), rows.data()));
        // groups that hold rows, ascending (key 1, key 2, ..) = ascending slot, the NULL key (slot `total`) last
        std::vector<size_t> slots;
        for (size_t g = 0; g < S; ++g) if (rows[g] > 0) slots.push_back(g);
        const bool null_group = !slots.empty() && slots.back() == (size_t)total;
        for (size_t i = 0; i < keys.size(); ++i) {
            uint64_t below = 1;   // product of the domains of the keys after i
            for (size_t j = i + 1; j < keys.size(); ++j) below *= (uint64_t)dom[j];
            std::vector<int64_t> kvv;
            std::vector<bool> kvalid;
            for (size_t g : slots) {
                const bool isnull = g == (size_t)total;
                kvv.push_back(isnull ? 0 : kmin[i] + (int64_t)(((uint64_t)g / below) % (uint64_t)dom[i]));
                kvalid.push_back(!isnull);
            }
            const ArrayRef k64 = Array::from_vec<int64_t>(kvv, null_group ? &kvalid : nullptr);
            out_cols.push_back(Column::from_arrays(keys[i].dtype == DataType::Int64 ? std::vector<ArrayRef>{k64} : ScalarFunctions::cast({k64}, keys[i].dtype),
                                                   Field{keys[i].out_name, keys[i].dtype, true}));
        }
        for (auto& val : vals) {
            const size_t v = (size_t)(std::find(vnames.begin(), vnames.end(), val.name) - vnames.begin());
            const DataType vdt = val.dtype;
            std::vector<double> fs;
            std::vector<int64_t> is, cs;
            for (size_t g : slots) { const rdf_group_result& r = res[v * S + g]; fs.push_back(r.sum_f64); is.push_back(r.sum_i64); cs.push_back(r.count); }
            if (val.fn == AF::Sum) {
                const ArrayRef sums = is_float(vdt) ? Array::from_vec<double>(fs) : Array::from_vec<int64_t>(is);
                const DataType sdt = is_float(vdt) ? DataType::Float64 : DataType::Int64;
                out_cols.push_back(Column::from_arrays(sdt == vdt ? std::vector<ArrayRef>{sums} : ScalarFunctions::cast({sums}, vdt), Field{"sum(" + val.name + ")", vdt, true}));
            } else if (val.fn == AF::Count) {
                out_cols.push_back(Column::from_arrays(ScalarFunctions::cast({Array::from_vec<int64_t>(cs)}, DataType::UInt32), Field{"count(" + val.name + ")", DataType::UInt32, true}));
            } else {
                std::vector<double> m(slots.size());
                std::vector<bool> valid(slots.size());
                for (size_t r = 0; r < slots.size(); ++r) { valid[r] = cs[r] > 0; m[r] = cs[r] > 0 ? (is_float(vdt) ? fs[r] : (double)is[r]) / (double)cs[r] : 0.0; }
                out_cols.push_back(Column::from_arrays({Array::from_vec(m, &valid)}, Field{"avg(" + val.name + ")", DataType::Float64, true}));
            }
        }
        return true;
    }
    // the step as it stands in the lazy state: grouping columns that are columns of base_, the pending filter, lazy value expressions
    bool fused_dense_group_aggregate(const plan::Transformation& t) {
        if (t.names.empty() || t.names.size() > 4) return false;
        std::vector<DenseKey> keys;
        for (auto& n : t.names) {
            Entry* e = find(n);
            if (!e || e->def) return false;   // unknown (the caller reports it) or computed: not a column of base_
            keys.push_back(DenseKey{e->source, n, e->dtype});
        }
        std::vector<DenseVal> vals;
        for (auto& a : t.aggregations)
            for (auto& c : a.columns) {
                Entry* e = find(c);
                if (!e) return false;
                vals.push_back(DenseVal{a.function, c, e->dtype, ref(c)});
            }
        std::vector<Column> out_cols;
        if (!dense_groups(base_, keys, pending_, vals, out_cols)) return false;
        reset(DataFrame::from_columns(out_cols));
        return true;
    }
    static bool dense_group_aggregate(const DataFrame& f, const Column& kc, const plan::Transformation& t, std::vector<Column>& out_cols) {
        std::vector<DenseVal> vals;
        for (auto& a : t.aggregations)
            for (auto& c : a.columns) vals.push_back(DenseVal{a.function, c, f.column_by_name(c).data_type(), Expr::col(c)});
        return dense_groups(f, {DenseKey{kc.name(), kc.name(), kc.data_type()}}, nullptr, vals, out_cols);
    }

    void step_aggregate(const plan::Transformation& t) {
        if (!t.names.empty()) { step_group_aggregate(t); return; }
        struct Want { plan::AggregateFunction fn; std::string col; DataType dt; ExprRef e; };
        std::vector<Want> wants;
        for (auto& a : t.aggregations)
            for (auto& c : a.columns) {
                Entry* e = find(c);
                if (!e) throw DataFrameError(DataFrameError::ComputeError, "Aggregating column \"" + c + "\" does not exist");
                wants.push_back(Want{a.function, c, e->dtype, a.function == plan::AggregateFunction::Avg && !is_float(e->dtype)
                                                                   ? Expr::make(RDF_OP_CAST, ref(c), nullptr, RDF_F64) : ref(c)});
            }
        std::vector<Column> out_cols;
        for (size_t b = 0; b < wants.size(); b += RDF_MAX_VALUES) {
            const size_t n = std::min<size_t>(RDF_MAX_VALUES, wants.size() - b);
            Lowered low;
            rdf_program prog;
            std::memset(&prog, 0, sizeof prog);
            prog.filter_root = pending_ ? low.add(pending_) : -1;
            for (size_t k = 0; k < n; ++k) prog.value_roots[k] = low.add(wants[b + k].e);
            prog.nvalues = (int32_t)n;
            prog.sink = RDF_SINK_AGG;
            prog.nodes = low.nodes.data();
            prog.nnodes = (int32_t)low.nodes.size();
            std::vector<rdf_array> cols;
            for (auto& cn : low.columns) for (auto& a : base_.column_by_name(cn).data().chunks()) cols.push_back(a->view());
            rdf_agg_result res[RDF_MAX_VALUES];
            check(rdf_pipeline(&prog, cols.data(), (int32_t)low.columns.size(), (int64_t)base_.num_chunks(), nullptr, res));
            for (size_t k = 0; k < n; ++k) out_cols.push_back(agg_column(wants[b + k].fn, wants[b + k].col, wants[b + k].dt, res[k]));
        }
        reset(DataFrame::from_columns(out_cols));
    }
    // output naming / typing of Dataset::try_aggregate (src/expression.rs:114-221)
    static Column agg_column(plan::AggregateFunction fn, const std::string& col, DataType dt, const rdf_agg_result& r) {
        using AF = plan::AggregateFunction;
        const std::vector<bool> some{r.is_some != 0};
        auto mk = [&](const std::string& name, DataType t, double f, int64_t i, bool always) -> Column {
            const std::vector<bool>* valid = always ? nullptr : &some;
            ArrayRef a;
            switch (t) {
                case DataType::Float64: a = Array::from_vec(std::vector<double>{f}, valid); break;
                case DataType::Float32: a = Array::from_vec(std::vector<float>{(float)f}, valid); break;
                case DataType::Int64: a = Array::from_vec(std::vector<int64_t>{i}, valid); break;
                case DataType::UInt64: a = Array::from_vec(std::vector<uint64_t>{(uint64_t)i}, valid); break;
                case DataType::Int32: a = Array::from_vec(std::vector<int32_t>{(int32_t)i}, valid); break;
                case DataType::UInt32: a = Array::from_vec(std::vector<uint32_t>{(uint32_t)i}, valid); break;
                case DataType::Int16: a = Array::from_vec(std::vector<int16_t>{(int16_t)i}, valid); break;
                case DataType::UInt16: a = Array::from_vec(std::vector<uint16_t>{(uint16_t)i}, valid); break;
                case DataType::Int8: a = Array::from_vec(std::vector<int8_t>{(int8_t)i}, valid); break;
                default: a = Array::from_vec(std::vector<uint8_t>{(uint8_t)i}, valid); break;
            }
            return Column::from_arrays({a}, Field{name, t, true});
        };
        const bool fl = is_float(dt);
        switch (fn) {
            case AF::Sum: return mk("sum(" + col + ")", dt, r.sum_f64, r.sum_i64, true);
            case AF::Min: return mk("min(" + col + ")", dt, r.min_f64, r.min_i64, false);
            case AF::Max: return mk("max(" + col + ")", dt, r.max_f64, r.max_i64, false);
            case AF::Count:
                if (r.count > (int64_t)UINT32_MAX) throw DataFrameError(DataFrameError::ComputeError, "count does not fit the UInt32 the reference's schema declares");
                return mk("count(" + col + ")", DataType::UInt32, 0, r.count, true);
            case AF::Avg: return mk("avg(" + col + ")", DataType::Float64, r.count ? r.sum_f64 / (double)r.count : 0.0, 0, false);
            default: (void)fl; throw DataFrameError(DataFrameError::ComputeError, "Aggregation not yet supported");
        }
    }

    void reset(const DataFrame& f) {
        base_ = f;
        cols_.clear();
        pending_ = nullptr;
        for (auto& fld : f.schema().fields) cols_.push_back(Entry{fld.name, fld.data_type, nullptr, fld.name});
    }

    // materialise: every lazy column through a fused SINK_STORE pass, then
    // the pending predicate as ONE mask + ONE compaction of all columns.
    DataFrame flush() {
        const size_t nch = base_.num_chunks();
        std::vector<Column> out(cols_.size());
        std::vector<size_t> lazy;
        for (size_t i = 0; i < cols_.size(); ++i) {
            if (cols_[i].def) lazy.push_back(i);
            else out[i] = base_.column_by_name(cols_[i].source).renamed(cols_[i].name);
        }
        if (!lazy.empty() && base_.num_columns() == 0) throw DataFrameError(DataFrameError::ComputeError, "calculation on an empty frame");
        const auto counts = base_.num_columns() ? base_.column(0).data().chunk_counts() : std::vector<int64_t>{};
        // One pass per lazy column: a single-value program is what the two kernel catalogs (exact shapes, then tree
        // shapes with runtime operators) are keyed on — 0.6-0.7 of the HBM peak against 0.36-0.44 for a multi-value
        // pass on the general evaluator, which more than pays for re-reading a shared input column.
        constexpr size_t kValuesPerPass = 1;
        for (size_t b = 0; b < lazy.size(); b += kValuesPerPass) {
            const size_t n = std::min<size_t>(kValuesPerPass, lazy.size() - b);
            Lowered low;
            rdf_program prog;
            std::memset(&prog, 0, sizeof prog);
            prog.filter_root = -1;
            for (size_t k = 0; k < n; ++k) prog.value_roots[k] = low.add(cols_[lazy[b + k]].def);
            prog.nvalues = (int32_t)n;
            prog.sink = RDF_SINK_STORE;
            prog.nodes = low.nodes.data();
            prog.nnodes = (int32_t)low.nodes.size();
            std::vector<rdf_array> cv;
            for (auto& cn : low.columns) for (auto& a : base_.column_by_name(cn).data().chunks()) cv.push_back(a->view());
            std::vector<std::shared_ptr<Array>> outs;
            std::vector<rdf_out> ov;
            for (size_t k = 0; k < n; ++k)
                for (size_t i = 0; i < nch; ++i) { outs.push_back(Array::make_out(cols_[lazy[b + k]].dtype, counts[i], true, base_.is_host())); ov.push_back(outs.back()->out_view(counts[i])); }
            check(rdf_pipeline(&prog, cv.data(), (int32_t)low.columns.size(), (int64_t)nch, ov.data(), nullptr));
            for (size_t k = 0; k < n; ++k) {
                std::vector<ArrayRef> chunks;
                for (size_t i = 0; i < nch; ++i) {
                    auto& o = outs[k * nch + i];
                    o->length = ov[k * nch + i].length;
                    o->null_count = ov[k * nch + i].null_count;
                    if (o->null_count == 0) o->validity.reset();  // no nulls resulted: drop the bitmap like Arrow builders do
                    chunks.push_back(o);
                }
                out[lazy[b + k]] = Column::from_arrays(chunks, Field{cols_[lazy[b + k]].name, cols_[lazy[b + k]].dtype, true});
            }
        }
        DataFrame result = DataFrame::from_columns(out);
        if (pending_) {
            Lowered low;
            const int root = low.add(pending_);
            const Column mask = base_.run_predicate(low, root);
            result = result.filter_by_mask(mask);
        }
        reset(result);
        return result;
    }

    DataFrame base_;
    std::vector<Entry> cols_;
    ExprRef pending_;
};

// ------------------------------------------------------------------------------------------------
// LazyFrame (src/lazyframe.rs:15-315): a plan builder over an in-memory source frame

class LazyFrame {
  public:
    static LazyFrame read(const DataFrame& source) {
        LazyFrame f;
        f.source_ = std::make_shared<DataFrame>(source);
        f.output_.name = "source";
        for (auto& fld : source.schema().fields) f.output_.columns.push_back(plan::Column{fld.name, fld.data_type});
        return f;
    }
    // LazyFrame::read(Computation) (src/lazyframe.rs:25-38): the plan starts at a Reader; nothing is loaded until evaluate(),
    // which first runs plan::optimise over the unrolled plan — a Limit / Select next to a CSV read becomes the reader's
    // max_records / projection, so the rows and columns that are not wanted are never parsed, uploaded or computed on.
    static LazyFrame read(const plan::Computation& read_computation) {
        if (!read_computation.is_single(plan::Transformation::Read)) throw DataFrameError(DataFrameError::ComputeError, "LazyFrame::read expects a read computation");
        LazyFrame f;
        f.read_ = std::make_shared<plan::Computation>(read_computation);
        f.output_ = read_computation.output;
        return f;
    }
    std::optional<std::pair<size_t, plan::Column>> column(const std::string& name) const { return output_.get_column(name); }
    const plan::Dataset& output() const { return output_; }
    // Expression::unroll (src/expression.rs:516-553): newest computation first, the read (if the plan has one) last
    std::vector<plan::Computation> unroll() const {
        std::vector<plan::Computation> u(comps_.rbegin(), comps_.rend());
        if (read_) u.push_back(*read_);
        return u;
    }
    // the plan after plan::optimise to a fixed point (the reference's tests apply it twice to merge a Select and a Limit into the read)
    std::vector<plan::Computation> optimised() const {
        std::vector<plan::Computation> u = unroll();
        for (int pass = 0; pass < 4; ++pass) {
            std::vector<plan::Computation> o = plan::optimise(u);
            const bool same = o.size() == u.size();
            u = std::move(o);
            if (same && pass > 0) break;
        }
        return u;
    }

    LazyFrame with_column(const std::string& col_name, const plan::Function& function, const std::vector<std::string>& input_col_names,
                          std::optional<DataType> as_type = std::nullopt) const {  // :58-96
        auto ops = plan::calculate(output_, input_col_names, function, col_name, as_type);
        LazyFrame f = *this;
        for (auto& t : ops) {
            if (t.kind != plan::Transformation::Calculate) throw DataFrameError(DataFrameError::ComputeError, "can't create column from a non-calculation transformation");
            f.output_ = f.output_.append_column(t.calc.output);
        }
        f.push(ops, output_);
        return f;
    }
    LazyFrame with_column_renamed(const std::string& old_name, const std::string& new_name) const {  // :98-131
        auto c = output_.get_column(old_name);
        if (!c) return *this;
        LazyFrame f = *this;
        f.output_.columns[c->first].name = new_name;
        f.output_.name = "renamed_dataset";
        f.push({plan::Transformation::Calculate_(plan::Calculation{"rename", {c->second}, plan::Column{new_name, c->second.data_type}, plan::Function::Rename_()})}, output_);
        return f;
    }
    LazyFrame filter(const FilterRef& cond) const { LazyFrame f = *this; f.push({plan::Transformation::Filter_(cond)}, output_); return f; }
    LazyFrame limit(size_t n) const { LazyFrame f = *this; f.push({plan::Transformation::Limit_(n)}, output_); return f; }
    LazyFrame select(const std::vector<std::string>& names) const {
        LazyFrame f = *this;
        plan::Dataset d; d.name = output_.name;
        for (auto& c : output_.columns) for (auto& n : names) if (c.name == n) { d.columns.push_back(c); break; }
        f.output_ = d;
        f.push({plan::Transformation::Select_(names)}, output_);
        return f;
    }
    LazyFrame drop(const std::vector<std::string>& names) const {
        LazyFrame f = *this;
        plan::Dataset d; d.name = output_.name;
        for (auto& c : output_.columns) { bool x = false; for (auto& n : names) x |= c.name == n; if (!x) d.columns.push_back(c); }
        f.output_ = d;
        f.push({plan::Transformation::Drop_(names)}, output_);
        return f;
    }
    LazyFrame sort(const std::vector<std::string>& cols, const std::vector<bool>& descending) const {
        LazyFrame f = *this;
        f.push({plan::Transformation::Sort_(cols, descending)}, output_);
        return f;
    }
    // LazyFrame::join (:225-251) with Dataset::try_join's checks and output naming (src/expression.rs:223-285): both key
    // columns must exist and have the same type; a column name present on both sides comes out as "a.<name>" / "b.<name>".
    // Evaluation is the reference's (src/evaluation.rs:75-84): both sub-plans are evaluated, then DataFrame::join (the
    // index pairs come from rdf_equijoin_indices_multi, the columns from rdf_take); the joined frame is the new source.
    LazyFrame join(const LazyFrame& other, const DataFrame::JoinCriteria& jc) const {
        const plan::Dataset planned = plan::try_join(output_, other.output_, jc.criteria);
        const DataFrame j = evaluate().join(other.evaluate(), jc);
        std::vector<Column> cols;
        for (size_t k = 0; k < planned.columns.size(); ++k) cols.push_back(j.column(k).renamed(planned.columns[k].name));
        LazyFrame f = LazyFrame::read(DataFrame::from_columns(cols));
        f.output_.name = planned.name;
        return f;
    }
    LazyFrame aggregate(const std::vector<std::string>& groups, const std::vector<plan::Aggregation>& aggr) const {   // :285-309
        LazyFrame f = *this;
        f.output_ = plan::try_aggregate(output_, groups, aggr);
        f.push({plan::Transformation::GroupAggregate_(groups, aggr)}, output_);
        return f;
    }
    // LazyFrame::evaluate (:311-315): unroll (newest first) and hand to Evaluate
    DataFrame evaluate() const {
        if (!read_) {
            std::vector<plan::Computation> unrolled(comps_.rbegin(), comps_.rend());
            return Evaluate::evaluate(*source_, unrolled);
        }
        std::vector<plan::Computation> plan_ = optimised();
        if (plan_.empty() || !plan_.back().is_single(plan::Transformation::Read)) throw DataFrameError(DataFrameError::ComputeError, "optimised plan does not end in a read");
        const plan::Reader& r = plan_.back().transformations[0].reader;
        DataFrame source = r.source == plan::Reader::Csv ? DataFrame::from_csv(r.path, r.csv)
                         : r.source == plan::Reader::Arrow ? DataFrame::from_arrow(r.path)
                         : throw DataFrameError(DataFrameError::ComputeError, "only CSV and Arrow IPC sources are loaded here");
        plan_.pop_back();
        return Evaluate::evaluate(source, plan_);
    }

  private:
    // one Computation per pushed step: input = the dataset before it (what `this` planned), output = the dataset after it
    void push(std::vector<plan::Transformation> t, const plan::Dataset& before) {
        plan::Computation c;
        c.input = {before};
        c.transformations = std::move(t);
        c.output = output_;
        comps_.push_back(std::move(c));
    }
    std::shared_ptr<DataFrame> source_;
    std::shared_ptr<plan::Computation> read_;   // set when the plan starts at a Reader instead of a loaded frame
    plan::Dataset output_;
    std::vector<plan::Computation> comps_;  // oldest first
};

}  // namespace rdf
