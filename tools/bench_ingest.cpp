// Ingestion rates of DataFrame::from_arrow / from_csv (src/dataframe.rs:349-407): bytes that reached HBM per second of wall time,
// next to the raw link rate of one pinned upload of the same size.  Build + run: tools/bench_ingest.py (writes the files with
// pyarrow, compiles this against include/rdf_frame.hpp, prints one JSON line per case).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "rdf_frame.hpp"

using clk = std::chrono::steady_clock;
static double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: bench_ingest <file.arrow> <file.csv>\n"); return 2; }
    const std::string arrow = argv[1], csv = argv[2];
    try {
        // the link: one pinned buffer, one async copy, one fence
        const int64_t link_bytes = (int64_t)1 << 30;
        {
            rdf::PinnedBuffer pin(link_bytes);
            std::memset(pin.data(), 1, (size_t)link_bytes);
            rdf::DeviceBuffer dev(link_bytes);
            for (int rep = 0; rep < 3; ++rep) {
                const auto t0 = clk::now();
                rdf::check(rdf_copy_h2d_async(dev.data(), pin.data(), link_bytes));
                rdf::check(rdf_copy_fence());
                const double s = since(t0);
                if (rep == 2) std::printf("{\"case\": \"link_pinned_1GiB\", \"bytes\": %lld, \"seconds\": %.6f, \"GBps\": %.2f}\n", (long long)link_bytes, s, link_bytes / s / 1e9);
            }
        }
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = clk::now();
            rdf::DataFrame df = rdf::DataFrame::from_arrow(arrow);
            const double s = since(t0);
            const rdf::IngestStats st = rdf::last_ingest();
            // a query over the loaded frame proves the data is there: sum of column 0
            if (rep == 2) {
                std::printf("{\"case\": \"from_arrow_file\", \"rows\": %lld, \"batches\": %zu, \"columns\": %zu, \"bytes_to_hbm\": %lld, \"seconds\": %.6f, \"GBps\": %.2f, "
                            "\"async_copies\": %lld, \"blocking_copies\": %lld, \"note\": \"the file mapped, its buffers through two 128 MiB page-locked staging buffers\"}\n",
                            (long long)df.num_rows(), df.num_chunks(), df.num_columns(), (long long)st.bytes, s, st.bytes / s / 1e9,
                            (long long)st.async_copies, (long long)st.blocking_copies);
            }
        }
        {   // the same image held by the caller (pinned in place by rdf_host_register)
            std::ifstream f(arrow, std::ios::binary | std::ios::ate);
            const int64_t size = (int64_t)f.tellg();
            f.seekg(0);
            std::vector<uint8_t> img((size_t)size);
            f.read((char*)img.data(), size);
            for (int rep = 0; rep < 3; ++rep) {
                const auto t0 = clk::now();
                rdf::DataFrame df = rdf::DataFrame::from_arrow_image(img.data(), img.size());
                const double s = since(t0);
                const rdf::IngestStats st = rdf::last_ingest();
                if (rep == 2) std::printf("{\"case\": \"from_arrow_image_registered\", \"rows\": %lld, \"bytes_to_hbm\": %lld, \"seconds\": %.6f, \"GBps\": %.2f, \"async_copies\": %lld, \"blocking_copies\": %lld}\n",
                                          (long long)df.num_rows(), (long long)st.bytes, s, st.bytes / s / 1e9, (long long)st.async_copies, (long long)st.blocking_copies);
            }
        }
        for (int rep = 0; rep < 2; ++rep) {
            const auto t0 = clk::now();
            rdf::DataFrame df = rdf::DataFrame::from_csv(csv);
            const double s = since(t0);
            const rdf::IngestStats st = rdf::last_ingest();
            if (rep == 1) std::printf("{\"case\": \"from_csv\", \"rows\": %lld, \"batches\": %zu, \"bytes_to_hbm\": %lld, \"seconds\": %.6f, \"GBps\": %.3f, \"async_copies\": %lld, \"blocking_copies\": %lld, "
                                      "\"note\": \"bound by the host text parser (std::from_chars on worker threads over ranges of records), not by the link\"}\n",
                                      (long long)df.num_rows(), df.num_chunks(), (long long)st.bytes, s, st.bytes / s / 1e9, (long long)st.async_copies, (long long)st.blocking_copies);
        }
    } catch (const std::exception& e) { std::fprintf(stderr, "bench_ingest: %s\n", e.what()); return 1; }
    return 0;
}
