/* example_dist.c — the N-GPU path from plain C, nothing but librdf_mi355x.so: GROUP BY key -> sum(value) over row-sharded
 * device-resident columns, one host thread per rank.  It replaces the panic! of Evaluate::evaluate's GroupAggregate arm
 * (src/evaluation.rs:73) for RecordBatches sharded over the GPUs of a node.
 *
 *   gcc -std=c11 -pthread -I include integration/example_dist.c -L rust_dataframe_amd -lrdf_mi355x -Wl,-rpath,$PWD/rust_dataframe_amd -o example_dist
 *   ./example_dist            one rank per visible GPU over RCCL (ncclCommInitAll inside the library; RCCL is never linked here)
 *   ./example_dist peer 4     4 ranks on the peer-copy transport; with fewer GPUs than ranks they share devices (test boxes)
 *
 * Every rank holds ROWS rows of key = (global row) % GROUPS, value = 1.0, so group g must end with count = sum =
 * (rows of all ranks with that key); each rank checks the groups it owns and the ranks' group counts must add up to GROUPS.
 * Without a GPU every call returns RDF_DEVICE_ERROR (there is no CPU fallback) and the program says so. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rdf_mi355x.h"

enum { ROWS = 200000, GROUPS = 1000 };

typedef struct { rdf_comm* comm; int rank, world, device; long long groups_owned; int ok; char err[256]; } rank_state;

#define TRY(call) do { rdf_status st_ = (call); if (st_ != RDF_OK) { snprintf(s->err, sizeof s->err, "%s: status %d: %s", #call, (int)st_, rdf_last_error()); return NULL; } } while (0)

static void* rank_main(void* arg) {
    rank_state* s = (rank_state*)arg;
    TRY(rdf_set_device(s->device));
    int64_t* hk = malloc(ROWS * sizeof *hk);
    double* hv = malloc(ROWS * sizeof *hv);
    for (int64_t i = 0; i < ROWS; ++i) { hk[i] = ((int64_t)s->rank * ROWS + i) % GROUPS; hv[i] = 1.0; }
    void *dk, *dv, *ok, *ov, *oc;
    TRY(rdf_dev_alloc(&dk, ROWS * 8)); TRY(rdf_dev_alloc(&dv, ROWS * 8));
    TRY(rdf_copy_h2d(dk, hk, ROWS * 8)); TRY(rdf_copy_h2d(dv, hv, ROWS * 8));
    const int64_t cap = GROUPS + 2;
    TRY(rdf_dev_alloc(&ok, (cap + 64) * 8)); TRY(rdf_dev_alloc(&ov, (cap + 64) * 8)); TRY(rdf_dev_alloc(&oc, (cap + 64) * 8));
    rdf_array keys = {dk, NULL, 0, ROWS, 0, RDF_I64, RDF_MEM_DEVICE}, vals = {dv, NULL, 0, ROWS, 0, RDF_F64, RDF_MEM_DEVICE};
    rdf_out out_k = {ok, NULL, cap, 0, 0, RDF_I64, RDF_MEM_DEVICE}, out_v = {ov, NULL, cap, 0, 0, RDF_F64, RDF_MEM_DEVICE}, out_c = {oc, NULL, cap, 0, 0, RDF_I64, RDF_MEM_DEVICE};
    rdf_exchange_stats st;
    TRY(rdf_groupby_agg_dist(s->comm, &keys, &vals, 1, RDF_AGG_SUM, GROUPS, RDF_EXCHANGE_AUTO, &out_k, &out_v, &out_c, &st));
    const int64_t n = out_k.length;
    int64_t* rk = malloc((n + 1) * 8); double* rs = malloc((n + 1) * 8); int64_t* rc = malloc((n + 1) * 8);
    if (n) { TRY(rdf_copy_d2h(rk, ok, n * 8)); TRY(rdf_copy_d2h(rs, ov, n * 8)); TRY(rdf_copy_d2h(rc, oc, n * 8)); }
    s->ok = 1;
    for (int64_t i = 0; i < n; ++i) {
        /* rows with key g over all ranks: (world * ROWS) / GROUPS, plus one for the first (world * ROWS) % GROUPS keys */
        const int64_t total = (int64_t)s->world * ROWS, want = total / GROUPS + (rk[i] < total % GROUPS ? 1 : 0);
        if (rc[i] != want || rs[i] != (double)want) { s->ok = 0; snprintf(s->err, sizeof s->err, "group %lld: count %lld sum %g, expected %lld", (long long)rk[i], (long long)rc[i], rs[i], (long long)want); }
    }
    s->groups_owned = n;
    /* the column's aggregates over all ranks: the shard's partial in, the total out (identical on every rank) */
    rdf_agg_result agg[RDF_MAX_VALUES];
    memset(agg, 0, sizeof agg);
    rdf_expr_node node = {RDF_NODE_COLUMN, 0, 0, -1, -1, 0, 0.0, 0};
    rdf_program prog = {&node, 1, -1, 1, {0, 0, 0, 0}, RDF_SINK_AGG};
    TRY(rdf_pipeline(&prog, &vals, 1, 1, NULL, agg));
    TRY(rdf_agg_combine(s->comm, agg, 1));
    if (agg[0].count != (int64_t)s->world * ROWS || agg[0].sum_f64 != (double)s->world * ROWS) { s->ok = 0; snprintf(s->err, sizeof s->err, "agg_combine: count %lld sum %g", (long long)agg[0].count, agg[0].sum_f64); }
    printf("rank %d of %d on device %d: %lld groups owned, exchange of %s in %d round(s), %lld bytes sent (%lld to other ranks), %.3f ms\n", s->rank, s->world, s->device,
           (long long)n, st.exchange == RDF_EXCHANGE_ROWS ? "rows" : "partial groups", st.rounds, (long long)st.bytes_sent, (long long)st.bytes_sent_remote, st.exchange_ms);
    TRY(rdf_comm_barrier(s->comm));
    rdf_dev_free(dk); rdf_dev_free(dv); rdf_dev_free(ok); rdf_dev_free(ov); rdf_dev_free(oc);
    free(hk); free(hv); free(rk); free(rs); free(rc);
    return NULL;
}

int main(int argc, char** argv) {
    const int kind = argc > 1 && strcmp(argv[1], "peer") == 0 ? RDF_COMM_PEER : RDF_COMM_RCCL;
    int32_t ngpu = 0;
    printf("%s\n", rdf_version());
    if (rdf_device_count(&ngpu) != RDF_OK || ngpu < 1) { printf("no gfx950 device: %s\n", rdf_last_error()); return 0; }
    int world = argc > 2 ? atoi(argv[2]) : ngpu;
    if (world < 1 || world > RDF_COMM_MAX_RANKS) world = ngpu;
    if (kind == RDF_COMM_RCCL && world > ngpu) world = ngpu;     /* RCCL wants one rank per GPU */
    int32_t devices[RDF_COMM_MAX_RANKS];
    rdf_comm* comms[RDF_COMM_MAX_RANKS];
    for (int r = 0; r < world; ++r) devices[r] = r % ngpu;
    rdf_status st = rdf_comm_init_all(world, devices, kind, comms);
    if (st != RDF_OK) { printf("rdf_comm_init_all: status %d: %s\n", (int)st, rdf_last_error()); return 1; }
    int32_t version = 0;
    rdf_comm_info(comms[0], NULL, NULL, NULL, NULL, &version);
    printf("%d rank(s), transport %s%s", world, kind == RDF_COMM_PEER ? "peer copies" : "RCCL ", "");
    if (kind == RDF_COMM_RCCL) printf("%d.%d.%d", version / 10000, version / 100 % 100, version % 100);
    printf("\n");
    rank_state state[RDF_COMM_MAX_RANKS];
    pthread_t th[RDF_COMM_MAX_RANKS];
    memset(state, 0, sizeof state);
    for (int r = 0; r < world; ++r) { state[r].comm = comms[r]; state[r].rank = r; state[r].world = world; state[r].device = devices[r]; pthread_create(&th[r], NULL, rank_main, &state[r]); }
    long long groups = 0;
    int ok = 1;
    for (int r = 0; r < world; ++r) {
        pthread_join(th[r], NULL);
        if (!state[r].ok) { ok = 0; printf("rank %d FAILED: %s\n", r, state[r].err); }
        groups += state[r].groups_owned;
    }
    for (int r = 0; r < world; ++r) rdf_comm_destroy(comms[r]);
    if (ok && groups != GROUPS) { ok = 0; printf("the ranks own %lld groups together, expected %d\n", groups, GROUPS); }
    printf(ok ? "ok: %lld groups over %d rank(s)\n" : "FAILED (%lld groups, %d ranks)\n", groups, world);
    return ok ? 0 : 1;
}
