// The scan that lays out the equi-join's table of distinct build keys (rust_dataframe_amd/csrc/rdf_join_place.h) on the CPU: the
// summaries {c, m} of runs of keys and their combination, held to the rule they replace — linear probing over keys that arrive in
// home-slot order, slot(r) = max(home(r), slot(r - 1) + 1) — for every way of cutting the key list into tiles and the tiles into
// threads' runs, the way join_place_kernel / join_place_scan_kernel do.
#include <algorithm>
#include <cstdint>
#include <random>
#include <vector>

#include "mini_test.hpp"
#include "../../rust_dataframe_amd/csrc/rdf_join_place.h"

using namespace rdfk;

namespace {

std::vector<long long> sequential(const std::vector<long long>& home) {
    std::vector<long long> pos(home.size());
    long long last = -1;
    for (size_t r = 0; r < home.size(); ++r) { pos[r] = std::max(home[r], last + 1); last = pos[r]; }
    return pos;
}

// the kernels' way: tiles of `tile` keys, cut into runs of `per` keys; summaries per run, exclusive prefixes over the runs of a tile
// and over the tiles, then every run places its keys from the prefix in front of it
std::vector<long long> tiled(const std::vector<long long>& home, size_t tile, size_t per) {
    const size_t n = home.size(), ntiles = (n + tile - 1) / tile;
    std::vector<PlaceCM> tsum(ntiles, place_empty());
    for (size_t t = 0; t < ntiles; ++t)
        for (size_t r0 = t * tile; r0 < std::min(n, (t + 1) * tile); r0 += per) {
            PlaceCM loc = place_empty();
            for (size_t r = r0; r < std::min({n, (t + 1) * tile, r0 + per}); ++r) (void)place_push(loc, home[r]);
            tsum[t] = place_join(tsum[t], loc);
        }
    std::vector<PlaceCM> carry(ntiles);
    PlaceCM run = place_empty();
    for (size_t t = 0; t < ntiles; ++t) { carry[t] = run; run = place_join(run, tsum[t]); }
    std::vector<long long> pos(n);
    for (size_t t = 0; t < ntiles; ++t) {
        PlaceCM ex = place_empty();                     // the runs of this tile in front of the current one
        for (size_t r0 = t * tile; r0 < std::min(n, (t + 1) * tile); r0 += per) {
            PlaceCM here = place_join(carry[t], ex), loc = place_empty();
            long long last = place_last(here);
            for (size_t r = r0; r < std::min({n, (t + 1) * tile, r0 + per}); ++r) {
                pos[r] = place_push(here, home[r]);
                CHECK(pos[r] > last);                   // (what the kernel's gap fill relies on)
                last = pos[r];
                (void)place_push(loc, home[r]);
            }
            ex = place_join(ex, loc);
        }
    }
    return pos;
}

std::vector<long long> homes(std::mt19937_64& g, size_t n, long long slots, int shape) {
    std::vector<long long> h(n);
    for (size_t i = 0; i < n; ++i) {
        switch (shape) {
            case 0: h[i] = (long long)(g() % (uint64_t)slots); break;                        // uniform hashes: load n / slots
            case 1: h[i] = (long long)(g() % 7); break;                                      // everything crowds on a few slots
            case 2: h[i] = slots - 1 - (long long)(g() % 3); break;                          // ... on the LAST slots (past the end: the margin's case)
            case 3: h[i] = (long long)((g() % (uint64_t)slots) & ~(uint64_t)0xFFF); break;   // clusters every 4096 slots
            default: h[i] = (long long)i * 3; break;                                         // no collisions at all
        }
    }
    std::sort(h.begin(), h.end());
    return h;
}

}  // namespace

TEST(join_of_runs_is_associative_and_has_an_identity) {
    std::mt19937_64 g(1);
    for (int it = 0; it < 20000; ++it) {
        PlaceCM a{(long long)(g() % 50), (long long)(g() % 1000) - 300}, b{(long long)(g() % 50), (long long)(g() % 1000) - 300}, c{(long long)(g() % 50), (long long)(g() % 1000) - 300};
        if (a.c == 0) a = place_empty();
        if (b.c == 0) b = place_empty();
        const PlaceCM l = place_join(place_join(a, b), c), r = place_join(a, place_join(b, c));
        CHECK(l.c == r.c);
        if (l.c > 0) CHECK(l.m == r.m);
        const PlaceCM e = place_join(place_empty(), a), f = place_join(a, place_empty());
        CHECK(e.c == a.c && f.c == a.c);
        if (a.c > 0) CHECK(e.m == a.m && f.m == a.m);
    }
}

TEST(tiles_place_what_linear_probing_places) {
    std::mt19937_64 g(2);
    for (int shape = 0; shape < 5; ++shape)
        for (size_t n : {(size_t)1, (size_t)2, (size_t)63, (size_t)2048, (size_t)2049, (size_t)40000}) {
            const long long slots = (long long)std::max<size_t>(4, 2 * n);
            const std::vector<long long> h = homes(g, n, slots, shape);
            const std::vector<long long> want = sequential(h);
            for (auto cut : {std::pair<size_t, size_t>{2048, 8}, {64, 4}, {7, 3}, {1, 1}, {100000, 100000}})
                CHECK(tiled(h, cut.first, cut.second) == want);
            // slots are distinct, in key order, never in front of a key's home
            for (size_t r = 0; r < n; ++r) { CHECK(want[r] >= h[r]); if (r) CHECK(want[r] > want[r - 1]); }
        }
}

TEST(a_probe_finds_every_key_by_walking_from_its_home) {
    std::mt19937_64 g(3);
    const size_t n = 5000;
    const long long slots = 16384;
    const std::vector<long long> h = homes(g, n, slots, 0);
    const std::vector<long long> pos = tiled(h, 2048, 8);
    std::vector<long long> table((size_t)slots + n + 1, -1);
    for (size_t r = 0; r < n; ++r) table[(size_t)pos[r]] = (long long)r;
    for (size_t r = 0; r < n; ++r) {
        long long s = h[r];
        while (table[(size_t)s] != (long long)r) { CHECK(table[(size_t)s] >= 0); ++s; }     // no empty slot between a key's home and the key
    }
}

int main() { return run_all(); }
