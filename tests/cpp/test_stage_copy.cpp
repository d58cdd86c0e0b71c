// packed_copy (rust_dataframe_amd/csrc/rdf_stage_copy.h) on the CPU: host chunks -> staging buffer and back, cut by bytes over the
// threads.  Every byte of every staged item must arrive exactly once, nothing else may be touched — for item lists of every
// shape: a handful of huge pieces (a streamed slab), a million tiny ones (the readers' 8 KiB batches), empty items, items that
// are not staged at all (large page-locked chunks travel by their own copy) in between.
#include <cstdint>
#include <random>

#include "mini_test.hpp"
#include "../../rust_dataframe_amd/csrc/rdf_stage_copy.h"

namespace {

struct Case { std::vector<std::vector<uint8_t>> src; std::vector<StageItem> items; size_t staged = 0; };

Case make(std::mt19937_64& g, size_t nitems, size_t max_bytes, double p_small, double p_empty) {
    Case c;
    c.src.resize(nitems);
    size_t off = 0;
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (size_t i = 0; i < nitems; ++i) {
        size_t b = u(g) < p_empty ? 0 : (size_t)(g() % max_bytes) + 1;
        if (u(g) < 0.1) b = (b + 7) & ~(size_t)7;
        c.src[i].resize(b);
        for (size_t k = 0; k < b; k += 61) c.src[i][k] = (uint8_t)(g() & 255);
        if (b) { c.src[i][0] = (uint8_t)(i * 7 + 1); c.src[i][b - 1] = (uint8_t)(i * 13 + 5); }
        const bool small = u(g) < p_small;
        c.items.push_back(StageItem{c.src[i].data(), b, 0, small});
        if (small) { c.items.back().off = off; off += (b + 16 + 63) & ~(size_t)63; }     // Region::layout's padding
    }
    c.staged = off;
    return c;
}

void round_trip(Case& c) {
    std::vector<uint8_t> pin(c.staged + 64, 0xEE);
    packed_copy(c.items, (char*)pin.data(), true, c.staged);
    std::vector<uint8_t> expect(c.staged + 64, 0xEE);
    for (size_t i = 0; i < c.items.size(); ++i)
        if (c.items[i].small && c.items[i].bytes) memcpy(expect.data() + c.items[i].off, c.src[i].data(), c.items[i].bytes);
    CHECK(pin == expect);                                   // every staged byte in place, padding and the rest untouched
    // and back: into fresh destinations
    std::vector<std::vector<uint8_t>> dst(c.items.size());
    std::vector<StageItem> back = c.items;
    for (size_t i = 0; i < back.size(); ++i) { dst[i].assign(back[i].bytes + 8, 0xAB); back[i].src = dst[i].data(); }
    packed_copy(back, (char*)pin.data(), false, c.staged);
    for (size_t i = 0; i < back.size(); ++i) {
        if (back[i].small) CHECK(memcmp(dst[i].data(), c.src[i].data(), back[i].bytes) == 0);
        else for (size_t k = 0; k < back[i].bytes; ++k) CHECK(dst[i][k] == 0xAB);
        for (size_t k = back[i].bytes; k < back[i].bytes + 8; ++k) CHECK(dst[i][k] == 0xAB);     // nothing past an item's end
    }
}

}  // namespace

TEST(a_handful_of_huge_pieces) {
    std::mt19937_64 g(1);
    for (int rep = 0; rep < 4; ++rep) { Case c = make(g, 1 + rep * 2, (size_t)48 << 20, 1.0, 0.0); round_trip(c); }
}
TEST(thousands_of_tiny_batches) {
    std::mt19937_64 g(2);
    Case c = make(g, 40000, 8192, 1.0, 0.02);
    round_trip(c);
    Case d = make(g, 3000, 70000, 0.7, 0.1);                 // some items travel on their own
    round_trip(d);
}
TEST(mixed_sizes_and_the_single_thread_path) {
    std::mt19937_64 g(3);
    for (int rep = 0; rep < 20; ++rep) {
        Case c = make(g, (size_t)(g() % 200) + 1, (size_t)1 << (10 + g() % 13), 0.8, 0.15);
        round_trip(c);
    }
    Case e = make(g, 5, 100, 1.0, 1.0);                      // nothing but empty items
    round_trip(e);
    Case one = make(g, 1, (size_t)20 << 20, 1.0, 0.0);       // one item, many threads inside it
    round_trip(one);
}

int main() { return run_all(); }
