// rdf_join_place.h — where the distinct build keys of an equi-join land in the probe table when the build side arrives sorted by
// hash (rdf_kernels.hip: join_place_kernel / join_place_scan_kernel; rdf_capi.cpp: rdf_equijoin_indices_multi).  Linear probing
// over keys that come in home-slot order is a scan: slot(r) = max(home(r), slot(r - 1) + 1) = r + max over q <= r of (home(q) - q).
// A run of consecutive distinct keys is summarised by {c = how many, m = max(home(q) - q) with q counted from the run's start};
// two adjacent runs combine associatively, so tiles are summarised independently, one block scans the tiles' summaries, and every
// tile places its keys from the prefix in front of it.  Shared by the kernels and a CPU test (tests/cpp/test_join_place.cpp) that
// holds the combination to the sequential rule it replaces.  No HIP types in here.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define RDF_PLACE_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RDF_PLACE_HD inline
#endif

namespace rdfk {

constexpr long long kPlaceNone = -(1ll << 62);             // m of an empty run
struct PlaceCM { long long c, m; };
RDF_PLACE_HD PlaceCM place_empty() { PlaceCM r; r.c = 0; r.m = kPlaceNone; return r; }
// run a, then run b: b's keys are counted from a.c on, so their (home - index) values drop by a.c
RDF_PLACE_HD PlaceCM place_join(PlaceCM a, PlaceCM b) { PlaceCM r; r.c = a.c + b.c; const long long bm = b.m - a.c; r.m = a.m > bm ? a.m : bm; return r; }
// one more distinct key with home slot `home` behind run a; returns the slot it takes
RDF_PLACE_HD long long place_push(PlaceCM& a, long long home) {
    const long long g = home - a.c;
    a.m = g > a.m ? g : a.m;
    return a.c++ + a.m;
}
// slot of the last key of a non-empty prefix a (absolute: a counted from the first key of all)
RDF_PLACE_HD long long place_last(PlaceCM a) { return a.c > 0 ? a.c - 1 + a.m : -1; }

}  // namespace rdfk
