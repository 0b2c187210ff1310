/*
 * gfs_oracle.h — CPU ORACLE for the path-guided SGD hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the algorithm in pangenome/gfasort `src/sgd.rs`
 * (reference @ 2025-12-12).  It exists only so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg can check / time-beside the HIP product path.  Nothing
 * under gfasort_amd/ may include, link or call it.
 *
 * PARITY STATUS: **parity unpinned** at the RNG boundary.  The reference draws its random
 * numbers from the third-party crates rand 0.9 / rand_xoshiro 0.7 / rand_distr 0.5, whose
 * sources are not under /root/reference and whose versions are not pinned (Cargo.lock is
 * git-ignored there); no reference test fixes an RNG output, a position or an order
 * (SURVEY.md §8c).  The crate semantics restated here (Xoshiro256+, SplitMix64 seeding,
 * rand-0.9 Lemire uniform with precomputed threshold, 53-bit f64) follow their published
 * algorithms; the xoshiro256+ / splitmix64 streams are pinned to the public reference
 * vectors of those generators in tests/test_oracle_kat.py.  Everything that does NOT
 * depend on an RNG stream (fast_precise_pow, zeta table, eta schedule, PathIndex, the
 * derived parameters, Layout, apply_ordering) is pinned against the reference's own
 * unit-test values and fixture files (tests/golden/).
 *
 * The reference itself (Rust) cannot be built here: no cargo/rustc in the image.
 */
#ifndef GFS_ORACLE_H
#define GFS_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFO_NO_NODE 0xFFFFFFFFu

/* Flattened read-only view of what sgd.rs reads from BidirectedGraph
 * (graph.nodes[*].sequence.len(), graph.paths[*].steps, graph.node_order). */
typedef struct gfo_graph {
    uint64_t n_nodes;               /* graph.node_count()                                   */
    uint64_t n_steps;               /* PathIndex::get_total_steps()                          */
    uint64_t n_paths;
    const uint32_t *node_len;       /* [n_nodes] sequence length by dense idx (node_order)   */
    const uint32_t *step_node;      /* [n_steps] dense idx of the step's node, GFO_NO_NODE   */
    const uint8_t  *step_is_rev;    /* [n_steps] Handle::is_reverse()                        */
    const uint64_t *path_first_step;/* [n_paths+1]                                           */
} gfo_graph;

/* PathSGDParams (sgd.rs:196-212) / LayoutSGDParams (sgd.rs:676-707), field for field. */
typedef struct gfo_params {
    uint64_t iter_max;
    uint64_t iter_with_max_learning_rate;
    uint64_t min_term_updates;
    double   delta;
    double   eps;
    double   eta_max;
    double   theta;
    uint64_t space;
    uint64_t space_max;
    uint64_t space_quantization_step;
    double   cooling_start;
    uint64_t nthreads;
    uint64_t seed;
    uint64_t dimensions;            /* layout only */
} gfo_params;

typedef struct gfo_stats {
    uint64_t term_updates;          /* successful updates (counted where sgd.rs:579 counts)  */
    uint64_t attempts;              /* loop trips incl. `continue`s                           */
    uint64_t iterations;            /* batches run (iter_max+1)                               */
    double   seconds;               /* wall time inside the SGD loops                         */
} gfo_stats;

/* One sampled term, for stream-level parity checks of the sampler (x-independent). */
typedef struct gfo_term {
    uint32_t i, j;                  /* dense node idx (1D) or 2*idx+end (nD)                  */
    double   d_ij;                  /* term_dist                                              */
} gfo_term;

/* ---- scalar pieces (KAT surface) ---- */
double   gfo_fast_precise_pow(double a, double b);                    /* sgd.rs:155-182 */
uint64_t gfo_dirty_zipfian(uint64_t min, uint64_t max, double theta,
                           double zeta, double zeta2theta, double u); /* sgd.rs:128-150 with u given */
uint64_t gfo_splitmix64_next(uint64_t *state);
void     gfo_xoshiro_seed(uint64_t seed, uint64_t s[4]);              /* seed_from_u64 */
uint64_t gfo_xoshiro_next(uint64_t s[4]);                             /* xoshiro256+ next_u64 */
uint64_t gfo_uniform_usize(uint64_t s[4], uint64_t n);                /* rand 0.9 Uniform<usize>(0,n) */
void     gfo_set_force_wide_steps(int on);                            /* test hook, see gfs_oracle.c */
uint32_t gfo_flip(uint64_t s[4]);                                     /* Uniform<i32>(0,2) */
double   gfo_random_f64(uint64_t s[4]);                               /* rng.random::<f64>() */

/* ---- tables ---- */
void     gfo_schedule(const gfo_params *p, double *etas /* iter_max+1 */);   /* sgd.rs:617-638 */
uint64_t gfo_zeta_size(const gfo_params *p);                                  /* sgd.rs:311-315 */
void     gfo_zetas(const gfo_params *p, double *zetas);                       /* sgd.rs:317-331 */

/* ---- PathIndex (sgd.rs:34-71): step_pos[S], step_path[S], step_rank[S], path_len[P] ---- */
void     gfo_path_index(const gfo_graph *g, uint64_t *step_pos, uint32_t *step_path,
                        uint64_t *step_rank, uint64_t *path_len);
/* initial positions (sgd.rs:271-294): x[idx] = prefix sum of node_len */
void     gfo_init_positions(const gfo_graph *g, double *x);
/* layout init dim 0 (sgd.rs:832-853); dims>=1 are caller-supplied (rand_distr ziggurat not restated) */
void     gfo_init_layout_dim0(const gfo_graph *g, uint64_t dims, double *coords /* Layout order */);
/* rand_distr StandardNormal (ziggurat) restated — PARITY UNPINNED, see gfs_oracle.c — and the full layout start */
void     gfo_ziggurat_tables(double *x257, double *f257);
double   gfo_standard_normal(uint64_t rng[4]);
void     gfo_init_layout(const gfo_graph *g, uint64_t dims, uint64_t seed, double *coords /* Layout order */); /* sgd.rs:829-853 */

/* ---- deterministic mode: n_streams Xoshiro streams (seed+t, = reference worker tid),
 * advanced round-robin one attempt each; stream t performs exactly its quota of successful
 * updates per iteration (quota = min_term_updates split evenly, remainder to low t);
 * attempts per stream per iteration are bounded by attempt_factor*quota+1024.
 * trace (nullable): first trace_per_stream successful terms of every stream, over the whole run. */
int gfo_sgd_1d(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
               uint64_t n_streams, uint64_t attempt_factor, double *x,
               gfo_term *trace, uint64_t trace_per_stream, gfo_stats *st);
/* coords in Layout order: coords[node*2*D + end*D + dim] (layout.rs:14) */
int gfo_sgd_nd(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
               uint64_t n_streams, uint64_t attempt_factor, double *coords,
               gfo_term *trace, uint64_t trace_per_stream, gfo_stats *st);

/* ---- the same, resumable: one call per iteration, streams keep their RNG in between.
 * dims = 0 for 1D.  stream t is seeded seed + stream_base + t; quota_total (0 = min_term_updates)
 * is this state's share of an iteration's updates (multi-rank tests). */
typedef struct gfo_state gfo_state;
int  gfo_state_create(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
                      uint64_t dims, uint64_t n_streams, uint64_t stream_base, uint64_t quota_total,
                      uint64_t attempt_factor, gfo_term *trace, uint64_t trace_per_stream, gfo_state **out);
/* bundle > 1: mirror of the product's bundled ("run") sampler — NOT a reference mode; 1D only,
 * n_streams % 64 == 0, bundle in {4,8,16,32,64}.  See gfs_oracle.c. */
int  gfo_state_set_bundle(gfo_state *s, uint64_t bundle);
int  gfo_state_set_chain(gfo_state *s, uint64_t chain);  /* mirror of GFS_F_CHAIN: longest run in trips (B = 64, 1D) */
int  gfo_state_set_one_colour(gfo_state *s, int on);   /* mirror of GFS_F_DBG_ONE_COLOUR */
int  gfo_state_set_no_fused_trip(gfo_state *s, int on); /* mirror of GFS_F_DBG_NO_FUSED_TRIP (1D) */
int  gfo_state_set_chunk(gfo_state *s, uint64_t chunk); /* updates per chunk of a team wave's work (product: KArgs.chunk) */
int  gfo_state_set_partners(gfo_state *s, int partners, int no_twin);   /* partner draws per leader (product default at B = 64
                                                                           in 1D: 2); no_twin: mirror of GFS_F_DBG_NO_TWIN_TRIP */
/* bundled mode: the product's internal node layout (slot of dense node k), to mirror its line-aligned runs; NULL = identity */
int  gfo_state_set_node_slots(gfo_state *s, const uint32_t *slot);
int  gfo_state_run_iteration(gfo_state *s, uint64_t k, double *x);
void gfo_state_stats(const gfo_state *s, gfo_stats *st);
void gfo_state_destroy(gfo_state *s);

/* ---- reference-like mode (timed CPU baseline): p->nthreads Hogwild workers + 1 ms checker
 * thread (sgd.rs:366-407, 413-593).  flat=0: node lookup through a hash map and 8-byte step
 * arrays like the reference; flat=1: dense arrays, no hash map (optimised CPU variant).
 * max_seconds>0 stops early (bounded sample); st->term_updates counts exactly. */
int gfo_sgd_1d_threads(const gfo_graph *g, const gfo_params *p, const double *etas,
                       const double *zetas, int flat, double max_seconds, double *x, gfo_stats *st);
int gfo_sgd_nd_threads(const gfo_graph *g, const gfo_params *p, const double *etas,
                       const double *zetas, int flat, double max_seconds, double *coords, gfo_stats *st);

/* ---- quality metrics ---- */
/* calculate_layout_stress (sgd.rs:1196-1283), seed 12345; coords in Layout order, dims>=1 */
double gfo_layout_stress(const gfo_graph *g, uint64_t dims, const double *coords, uint64_t sample_count);
/* same formula on a 1D position vector x[n_nodes] */
double gfo_stress_1d(const gfo_graph *g, const double *x, uint64_t sample_count);

#ifdef __cplusplus
}
#endif
#endif
