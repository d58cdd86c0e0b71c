// rdf_stage_copy.h — packing host chunks into / out of a page-locked staging buffer with several threads (rdf_capi.cpp's host
// path and the streamed batch loop).  No HIP in here: tests/cpp/test_stage_copy.cpp runs it on the CPU.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <thread>
#include <vector>

struct StageItem {
    const void* src;   // host source (H2D) — or destination (D2H)
    size_t bytes;
    size_t off;        // offset inside the region
    bool small;
};

// Packing many small host chunks (the reference's readers emit 1024-row batches = 8 KiB per f64 column) into the pinned
// buffer is a CPU memcpy; above a few MB it is spread over a handful of threads, or it — not PCIe — bounds the call
// (measured: 4.8 GB/s single-threaded against 52-55 GB/s for chunks that are DMA'd directly).
inline void packed_copy(const std::vector<StageItem>& items, char* pin, bool to_pinned, size_t small_bytes) {
    auto run = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const StageItem& it = items[i];
            if (!it.small || !it.bytes) continue;
            if (to_pinned) memcpy(pin + it.off, it.src, it.bytes); else memcpy((void*)it.src, pin + it.off, it.bytes);
        }
    };
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt > 16 ? 16 : nt;
    if (small_bytes < ((size_t)4 << 20) || nt < 2) { run(0, items.size()); return; }
    // cut by BYTES, not by items: a slab of the streamed batch loop is a handful of pieces of tens of MB (by items, one thread
    // moved them at ~10 GB/s, a fifth of the link), the readers' 8 KiB batches are a million items.  Thread t moves the bytes
    // [t, t + 1) * total / nt of the concatenation of the items, wherever in an item that range starts and ends.
    size_t total = 0;
    for (const StageItem& it : items) if (it.small) total += it.bytes;
    if (total == 0) return;
    nt = (unsigned)std::min<size_t>(nt, (total + ((size_t)1 << 20) - 1) >> 20);        // at least 1 MiB per thread
    if (nt < 2) { run(0, items.size()); return; }
    struct Start { size_t item, off; };
    std::vector<Start> start((size_t)nt + 1, Start{items.size(), 0});
    {
        size_t seen = 0;
        unsigned t = 0;
        for (size_t i = 0; i < items.size() && t < nt; ++i) {
            const StageItem& it = items[i];
            if (!it.small || !it.bytes) continue;
            while (t < nt && (size_t)t * total / nt < seen + it.bytes) {
                const size_t target = (size_t)t * total / nt;
                start[t] = Start{i, target > seen ? target - seen : 0};
                ++t;
            }
            seen += it.bytes;
        }
    }
    auto part = [&](unsigned t) {
        const Start a = start[t], b = start[(size_t)t + 1];
        for (size_t i = a.item; i < items.size() && i <= b.item; ++i) {
            const StageItem& it = items[i];
            if (!it.small || !it.bytes) continue;
            const size_t lo = i == a.item ? a.off : 0, hi = i == b.item ? b.off : it.bytes;
            if (hi <= lo) continue;
            if (to_pinned) memcpy(pin + it.off + lo, (const char*)it.src + lo, hi - lo); else memcpy((char*)it.src + lo, pin + it.off + lo, hi - lo);
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(part, t);
    part(0);
    for (auto& x : th) x.join();
}

