/* example.c — the C ABI from plain C: sum(x | x > 0.5) over two host-resident chunks, the way the Rust shim's
 * `Evaluate::evaluate` replacement calls it.  Build and run on a box with an MI355X:
 *
 *   gcc -std=c11 -I include integration/example.c -L rust_dataframe_amd -lrdf_mi355x -Wl,-rpath,$PWD/rust_dataframe_amd -o example
 *   ./example
 *
 * Prints "sum = 3.25 over 4 rows".  Without a GPU every compute call returns RDF_DEVICE_ERROR (there is no CPU fallback) and the program says so. */
#include <stdio.h>
#include <string.h>

#include "rdf_mi355x.h"

int main(void) {
    double c0[] = {0.25, 0.75, 0.5, 1.0};
    double c1[] = {0.9, 0.1, 0.6};
    rdf_array chunks[2];
    memset(chunks, 0, sizeof chunks);
    chunks[0].values = c0; chunks[0].length = 4; chunks[0].dtype = RDF_F64; chunks[0].mem = RDF_MEM_HOST;
    chunks[1].values = c1; chunks[1].length = 3; chunks[1].dtype = RDF_F64; chunks[1].mem = RDF_MEM_HOST;

    /* expression nodes: 0 = column 0, 1 = scalar 0.5, 2 = (col0 > 0.5) */
    rdf_expr_node nodes[3];
    memset(nodes, 0, sizeof nodes);
    nodes[0].kind = RDF_NODE_COLUMN; nodes[0].column = 0; nodes[0].lhs = nodes[0].rhs = -1;
    nodes[1].kind = RDF_NODE_SCALAR; nodes[1].dtype = RDF_F64; nodes[1].f64 = 0.5; nodes[1].lhs = nodes[1].rhs = -1;
    nodes[2].kind = RDF_NODE_OP; nodes[2].op = RDF_OP_GT; nodes[2].lhs = 0; nodes[2].rhs = 1;

    rdf_program prog;
    memset(&prog, 0, sizeof prog);
    prog.nodes = nodes; prog.nnodes = 3; prog.filter_root = 2; prog.nvalues = 1; prog.value_roots[0] = 0; prog.sink = RDF_SINK_AGG;

    rdf_agg_result agg[RDF_MAX_VALUES];
    memset(agg, 0, sizeof agg);
    printf("%s\n", rdf_version());
    rdf_status st = rdf_pipeline(&prog, chunks, 1, 2, NULL, agg);
    if (st != RDF_OK) {
        printf("rdf_pipeline: status %d: %s\n", (int)st, rdf_last_error());
        return st == RDF_DEVICE_ERROR ? 0 : 1;
    }
    printf("sum = %.17g over %lld rows (expected 3.25 over 4)\n", agg[0].sum_f64, (long long)agg[0].count);
    return agg[0].count == 4 && agg[0].sum_f64 == 3.25 ? 0 : 1;
}
