/*
 * gfasort_hip.h — C ABI of libgfasort_hip.so: the MI355X (gfx950) path-guided SGD engine that
 * replaces the inner loops of pangenome/gfasort's `Y` (1D sort) and `L` (nD layout) steps.
 *
 * The reference has no FFI; the seam is the pair of Rust functions
 *     pub fn path_linear_sgd(graph: Arc<BidirectedGraph>, params: PathSGDParams) -> HashMap<usize,f64>
 *                                                                        (src/sgd.rs:237-240)
 *     pub fn path_linear_sgd_layout(graph: Arc<BidirectedGraph>, params: LayoutSGDParams) -> Layout
 *                                                                        (src/sgd.rs:773-776)
 * and this header is what a Rust `extern "C"` block for that seam binds (INTEGRATION.md shows
 * the shim).  Plain pointers and sizes only; caller owns every buffer; nothing is retained
 * after a call returns except inside an explicit gfs_ctx.
 *
 * Return codes: 0 ok; 1 nothing to do (empty graph or no path with >1 step: the reference
 * returns an empty map / zero Layout, sgd.rs:242-244,258-261,780-782,795-798 — leave the graph
 * untouched); <0 error, text in gfs_last_error().  Never throws or aborts across the boundary.
 * There is NO CPU fallback: without a HIP device every compute entry point returns GFS_E_HIP.
 */
#ifndef GFASORT_HIP_H
#define GFASORT_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFS_OK            0
#define GFS_NOTHING_TO_DO 1
#define GFS_E_ARG        (-1)
#define GFS_E_HIP        (-2)
#define GFS_E_NOMEM      (-3)
#define GFS_E_STATE      (-4)
#define GFS_E_UNSUPPORTED (-5)

#define GFS_NO_NODE 0xFFFFFFFFu
#define GFS_MAX_DIMS 8

/* Read-only, caller-owned SoA view of the three things sgd.rs reads from BidirectedGraph
 * (src/graph_ops.rs:10-16): graph.nodes[*].sequence.len(), graph.paths[*].steps and
 * graph.node_order.  Dense node index k = k-th existing node of node_order, i.e. the value
 * the reference stores in handle_to_idx[Handle::forward(id)] (sgd.rs:286-294, 811-814). */
typedef struct gfs_graph_view {
    uint64_t        n_nodes;          /* graph.node_count()                       sgd.rs:241 */
    uint64_t        n_steps;          /* PathIndex::get_total_steps()             sgd.rs:73  */
    uint64_t        n_paths;          /* PathIndex::num_paths()                   sgd.rs:101 */
                                       /* limits of the device mirror: n_nodes < 2^31, n_paths < 2^31,
                                          n_steps <= 2^40, < 2^32 steps and < 2^55 bp per path
                                          (GFS_E_UNSUPPORTED beyond) */
    const uint32_t *node_len;         /* [n_nodes]  sequence.len() by dense index            */
    const uint32_t *step_node;        /* [n_steps]  dense index of the step's node, or
                                         GFS_NO_NODE when the id is absent from the graph
                                         (the reference warns and skips, sgd.rs:525-538)     */
    const uint8_t  *step_is_rev;      /* [n_steps]  Handle::is_reverse()      graph.rs:37    */
    const uint64_t *path_first_step;  /* [n_paths+1] PathInfo.first_step; counts by diff     */
} gfs_graph_view;

/* PathSGDParams, src/sgd.rs:196-212, same field names and order (usize->u64, bool->u8). */
typedef struct gfs_sgd_params {
    uint64_t iter_max;
    uint64_t iter_with_max_learning_rate;
    uint64_t min_term_updates;
    double   delta;                   /* carried, unused — as in the reference               */
    double   eps;
    double   eta_max;
    double   theta;
    uint64_t space;
    uint64_t space_max;
    uint64_t space_quantization_step;
    double   cooling_start;
    uint64_t nthreads;                /* CPU concept; ignored by the HIP engine              */
    uint8_t  progress;
    uint8_t  _pad[7];
    uint64_t seed;
} gfs_sgd_params;

/* LayoutSGDParams, src/sgd.rs:676-707. */
typedef struct gfs_layout_params {
    uint64_t dimensions;              /* 1..GFS_MAX_DIMS                                     */
    gfs_sgd_params sgd;               /* the remaining 14 fields are identical               */
} gfs_layout_params;

/* Device-side launch shape (no reference equivalent; the reference's analogue is nthreads).
 * A "stream" is one Xoshiro256+ generator seeded seed + stream_base + t, exactly what
 * reference worker thread tid = stream_base + t uses (sgd.rs:431-432); one GPU lane owns one
 * stream.  All-zero = defaults. */
typedef struct gfs_launch_config {
    uint64_t n_streams;               /* 0 = auto (fill the chip, >= 1 update per stream)    */
    uint64_t stream_base;             /* first global stream id on this device (multi-GPU)   */
    uint64_t term_updates_per_iteration; /* 0 = params.min_term_updates (this device's share) */
    uint64_t attempt_factor;          /* 0 = 64: a stream gives up an iteration after
                                         attempt_factor*quota+1024 sampled-and-rejected trips */
    uint32_t block_size;              /* 0 = 256                                             */
    uint32_t flags;                   /* GFS_F_*                                             */
    uint64_t trace_per_stream;        /* debug: keep the first k successful terms per stream */
} gfs_launch_config;

#define GFS_F_PLAIN_LOADS   1u        /* read positions with plain (L2-cacheable) loads instead
                                         of agent-scope relaxed atomic loads                 */
#define GFS_F_NO_LDS_TABLES 2u        /* keep zeta/path tables in global memory              */
/* Sampling bundle: n adjacent lanes share ONE sampled (step a, jump) drawn by an ordinary reference
 * stream and take n consecutive steps of the path with the same signed jump (nD, D <= 3: and the same
 * pair of end flips), so that record loads, position loads and atomics of a bundle coalesce into a
 * few 64-B requests (gfasort_amd/csrc/sgd_device.h).  n = 1: reference streams, every lane is a
 * reference worker thread.  n = 0 (default): the library picks — 1 for graphs of < 16384 nodes, else the widest
 * bundle (64, 32, ...) for which >= 95 % of the steps lie in paths of at least 4 n steps (measured basis:
 * profiles/r03/policy_sweep.log).  On graphs whose haplotypes disagree by kilobases the bundled sampler needs a longer
 * schedule than the reference's to reach the same quality (tests/test_gpu_quality.py). */
#define GFS_F_BUNDLE(n) (((uint32_t)(n) & 0xFFu) << 16)  /* n in {0 = auto, 1, 4, 8, 16, 32, 64} */
/* Long runs: with bundles of 64 a sampled (step a, jump) is expanded over k consecutive trips of its wave, i.e. over
 * 64*k consecutive steps (k adapts downwards on short paths).  k = 0 (default): 64 for the 1D sort, 16 for the layout
 * kernels, less where an iteration would otherwise draw fewer than 64 leaders (a leader stands for up to 64 * k * partners
 * terms).  k = 1: a run is one trip. */
#define GFS_F_CHAIN(k) (((uint32_t)(k) & 0xFFu) << 24)   /* k in {0 = auto, 1, 2, 4, 8, 16, 32, 64} */
#define GFS_F_NO_FUSE       4u        /* gfs_ctx_run / gfs_ctx_run_range: one launch per iteration even where
                                         a fused persistent launch is possible                        */
#define GFS_F_ONE_PARTNER   0x10u     /* 1D team kernel at B = 64: one partner draw per leader (default: two — a sampled step a
                                         with two independent draws of its partner b; where both are line-aligned long jumps
                                         the two terms of a lane share the loads and the add of their a-side) */
#define GFS_F_DBG_NO_TWIN_TRIP 0x400u /* diagnostic: the two partners of a leader as two trips even where one would do   */
#define GFS_F_DBG_FREE_RUNNING 0x20u   /* diagnostic: fused launches with a fixed quota per wave and iteration and no pacing
                                         (round 1's launch; the waves drift apart in the schedule) instead of work pools */
#define GFS_F_DBG_NO_ATOMICS 0x100u   /* diagnostic ablation (wrong results): skip the atomic adds */
#define GFS_F_DBG_NO_XLOADS  0x200u   /* diagnostic ablation (wrong results): skip position loads  */
#define GFS_F_DBG_ONE_COLOUR   0x800u  /* diagnostic: short-jump runs execute their first colour only (round-1 behaviour:
                                         half of the run's terms, i.e. short jumps under-sampled)                  */
#define GFS_F_DBG_NO_ALIGN      0x1000u /* diagnostic: bundled sampler without line-aligned runs               */
#define GFS_F_DBG_ALIGN_FIRST   0x2000u /* diagnostic: line-align only the first run of a bundle               */
#define GFS_F_DBG_NO_FUSED_TRIP 0x8000u /* diagnostic: the two colours of a short-jump run as two trips even where one
                                         fused trip is possible (same terms, same order)                        */
#define GFS_F_DBG_WIDE_INDEX 0x4000u  /* test hook: draw step indices with the u64 sampler that graphs of
                                         more than 2^32-1 steps use (rand's usize sampler switches there) */

typedef struct gfs_stats {
    uint64_t term_updates;            /* successful updates, counted where sgd.rs:579 counts */
    uint64_t attempts;                /* loop trips including `continue`s                    */
    uint64_t iterations;              /* batches launched                                    */
    uint64_t n_streams;               /* streams actually used                               */
    uint64_t bundle;                  /* lanes per sampling bundle actually used (1 = reference streams) */
    double   kernel_ms;               /* sum of SGD kernel durations (HIP events)            */
    double   total_ms;                /* wall time inside the call (one-shot) / run          */
    uint64_t launches;                /* SGD kernel launches (a fused launch covers many iterations) */
    uint64_t run_trips;               /* longest run in trips (GFS_F_CHAIN) actually used; 1 = one trip per run */
} gfs_stats;

/* One sampled term (debug trace): 1D i,j = dense node index; nD = 2*idx+end. */
typedef struct gfs_term {
    uint32_t i, j;
    double   d_ij;
} gfs_term;

typedef struct gfs_ctx gfs_ctx;

/* ---- library ---- */
const char *gfs_version(void);
const char *gfs_last_error(void);
int         gfs_device_count(void);
/* optional: create the HIP context now (e.g. on a side thread while the caller parses its input) */
int         gfs_warmup(int device);

/* ---- host tables, bit-exact restatements (no device needed) ---- */
/* fast_precise_pow                                                        sgd.rs:155-182 */
double   gfs_fast_precise_pow(double a, double b);
/* path_linear_sgd_schedule with w_min=1/eta_max, w_max=1: etas[iter_max+1]  sgd.rs:300-308,617-638 */
int      gfs_sgd_schedule(const gfs_sgd_params *p, double *etas);
/* zeta table length and contents                                          sgd.rs:311-331 */
uint64_t gfs_zeta_table_len(const gfs_sgd_params *p);
int      gfs_zeta_table(const gfs_sgd_params *p, double *zetas);
/* initial positions x[idx] = bp prefix in node_order                      sgd.rs:271-294 */
int      gfs_init_positions(const gfs_graph_view *g, double *x);
/* layout init, dimension 0 only (+end = prefix, -end = prefix+len), Layout order; the
 * Gaussian dimensions >=1 (rand_distr StandardNormal, sgd.rs:829-850): gfs_init_layout, or the caller's.   */
int      gfs_init_layout_dim0(const gfs_graph_view *g, uint64_t dims, double *coords);
/* the reference's whole layout start (sgd.rs:829-853): dimension 0 as above, dimensions >= 1 = StandardNormal * sqrt(2N)
 * from ONE Xoshiro256+ seeded `seed`, drawn node by node (+ end, then - end).  rand_distr's ziggurat is restated from
 * its published algorithm with rebuilt tables: parity unpinned.                                       */
int      gfs_init_layout(const gfs_graph_view *g, uint64_t dims, uint64_t seed, double *coords);
/* positions -> rank order (ascending, ties by dense index)                sgd.rs:665-671 */
int      gfs_sort_order(const double *x, uint64_t n, uint64_t *order);

/* ---- one-shot entry points: what the Rust seam functions bind ---- */
/* replaces path_linear_sgd (src/sgd.rs:237).  x_inout[n_nodes]: in = initial positions
 * (NULL-initialised by gfs_init_positions if init_x != 0), out = final positions.
 * etas / zetas may be NULL (computed as the reference does).                               */
int gfs_path_linear_sgd(const gfs_graph_view *g, const gfs_sgd_params *p,
                        const gfs_launch_config *cfg, const double *etas, const double *zetas,
                        int init_x, double *x_inout, gfs_stats *stats);
/* replaces path_sgd_sort (src/sgd.rs:641): the same run, plus order_out[r] = dense index of the node
 * of rank r (ascending position, ties by dense index; sorted on the device).  x_inout as above.   */
int gfs_path_sgd_sort(const gfs_graph_view *g, const gfs_sgd_params *p,
                      const gfs_launch_config *cfg, const double *etas, const double *zetas,
                      int init_x, double *x_inout, uint64_t *order_out, gfs_stats *stats);
/* replaces path_linear_sgd_layout (src/sgd.rs:773).  coords_inout[n_nodes*2*D] in the order of
 * Layout.coords: coords[node*2*D + end*D + dim] (src/layout.rs:14,73-78).                   */
int gfs_path_linear_sgd_layout(const gfs_graph_view *g, const gfs_layout_params *p,
                               const gfs_launch_config *cfg, const double *etas, const double *zetas,
                               double *coords_inout, gfs_stats *stats);

/* ---- resident API: graph and positions stay in HBM between calls ---- */
int   gfs_ctx_create(const gfs_graph_view *g, int device, gfs_ctx **out);
/* Same, with an explicit internal node layout: node_perm[k] = slot of dense node k in the device
 * position vector (a permutation of 0..n_nodes-1).  NULL = first-visit path order with branches placed where they branch off (the default of
 * gfs_ctx_create).  Ranks of a multi-GPU run that all-reduce the device buffer in place must share
 * one layout.  Upload / download / trace always speak the ABI's dense indices and Layout.coords order;
 * only the raw device pointer (gfs_ctx_positions_device / gfs_ctx_bind_positions) is in device order:
 * 1D x[slot]; nD the end x dimension planes coords[end][dim][slot] (element-wise collectives do not care). */
int   gfs_ctx_create_with_layout(const gfs_graph_view *g, int device, const uint32_t *node_perm, gfs_ctx **out);
int   gfs_ctx_node_layout(const gfs_ctx *ctx, uint32_t *perm_out, uint64_t n_nodes);
void  gfs_ctx_destroy(gfs_ctx *ctx);
int   gfs_ctx_setup_1d(gfs_ctx *ctx, const gfs_sgd_params *p, const gfs_launch_config *cfg,
                       const double *etas, const double *zetas);
int   gfs_ctx_setup_nd(gfs_ctx *ctx, const gfs_layout_params *p, const gfs_launch_config *cfg,
                       const double *etas, const double *zetas);
uint64_t gfs_ctx_positions_len(const gfs_ctx *ctx);              /* n_nodes (1D) or n_nodes*2*D */
int   gfs_ctx_upload_positions(gfs_ctx *ctx, const double *host, uint64_t n);
int   gfs_ctx_init_positions(gfs_ctx *ctx);                      /* 1D: the reference's start (sgd.rs:286-294),
                                                                    prefix sum of node lengths, on the device */
int   gfs_ctx_download_positions(gfs_ctx *ctx, double *host, uint64_t n);
void *gfs_ctx_positions_device(gfs_ctx *ctx);                    /* device pointer (for RCCL)   */
int   gfs_ctx_bind_positions(gfs_ctx *ctx, void *device_ptr);    /* use a caller-owned buffer   */
int   gfs_ctx_reset_streams(gfs_ctx *ctx);                       /* re-seed RNG streams, zero stats */
/* one SGD batch: iteration k in 0..=iter_max uses etas[k]; cooling for
 * k > floor(cooling_start*iter_max) (sgd.rs:297,383-396).  Asynchronous on hip_stream
 * (a hipStream_t, NULL = the default stream).                                               */
int   gfs_ctx_run_iteration(gfs_ctx *ctx, uint64_t k, void *hip_stream);
/* iterations ks[0..n): ONE fused persistent launch (the team kernels at their widest bundle: sorts, layouts of 2 and 3
 * dimensions; reference streams in any dimension) in which the waves walk the schedule and draw every iteration's exact
 * number of updates from a shared pool; otherwise (narrower bundles, traces, GFS_F_NO_FUSE, more streams than fit on the
 * device at once, an iteration of >= 2^31 updates per pool counter) n launches.  A range of ONE layout iteration is pooled
 * as well where it is many chunks per wave (gfs_ctx_run_iteration keeps a fixed quota per wave).  Asynchronous.         */
int   gfs_ctx_run_range(gfs_ctx *ctx, const uint64_t *ks, uint64_t n, void *hip_stream);
int   gfs_ctx_run(gfs_ctx *ctx, void *hip_stream);               /* k = 0..=iter_max (run_range), then sync */
int   gfs_ctx_synchronize(gfs_ctx *ctx, void *hip_stream);
int   gfs_ctx_stats(gfs_ctx *ctx, gfs_stats *out);               /* synchronises                */
/* rank order of the context's current 1D positions, sorted on the device (rocPRIM radix sort)   */
int   gfs_ctx_sort_order(gfs_ctx *ctx, uint64_t *order, uint64_t n_nodes);
int   gfs_ctx_trace(gfs_ctx *ctx, gfs_term *out, uint64_t n_terms, uint64_t *counts, uint64_t n_streams);

/* ---- multi-device runs (no reference equivalent: the reference is one process, src/sgd.rs:413-593) ----
 * Paths are sharded over `world` ranks, one rank per GPU (one process per GPU, or one host thread per GPU); every rank
 * performs its share of an iteration's term updates on its own replica of the positions; after every window of
 * iterations the replicas are merged.  The collective itself is the CALLER's (RCCL over xGMI: torch.distributed, or
 * ncclAllReduce from the Rust host): the library hands out the device buffer to sum over the ranks and runs the kernels
 * on either side of it.  Only the slots that more than one rank's paths can move are exchanged (all ranks store the
 * positions in one node layout, the default layout's rule applied to the whole graph; a rank's paths touch one span of it);
 * gfs_rank_finish_* completes every replica at the end with one full-length f64 sum.                                  */
#define GFS_MAX_WORLD 64

typedef struct gfs_rank gfs_rank;

typedef struct gfs_rank_config {
    uint32_t rank, world;
    int32_t  device;                  /* HIP device of this rank                                                  */
    uint32_t sharding;                /* 0 auto, 1 consecutive blocks of paths, 2 longest-first bin packing       */
    uint32_t merge_every;             /* iterations per merge window (0 = 1); the last iteration always merges    */
    uint32_t merge_rule;              /* how a window's moves of a slot are merged over the c ranks that moved it:
                                         0 (default): summed move / max(1, c * min(1, window length * eta / mean node length)) —
                                         the mean of the ranks' proposals while the learning rate makes every term a full
                                         correction, their sum once the moves are small steps (multi.hip gfs_rank_window_end);
                                         1 sum; 2 mean over all ranks; 3 summed move / c at every learning rate            */
    uint32_t payload;                 /* exchange buffer element type: 0 f32, 1 f64                               */
    uint32_t exchange;                /* 0: only the slots two or more ranks can move; 1: the whole vector        */
    gfs_launch_config launch;         /* per-rank launch shape; term_updates_per_iteration and stream_base are set
                                         by the library (quota of the rank; rank * n_streams)                     */
} gfs_rank_config;

typedef struct gfs_rank_info {
    uint64_t quota;                   /* this rank's term updates per iteration                                   */
    uint64_t shard_steps;             /* steps of its multi-step paths                                            */
    uint64_t span_lo, span_hi;        /* slots [lo, hi) its paths touch in the shared node layout                 */
    uint64_t shared_slots;            /* slots two or more ranks can move (what a window exchanges)               */
    uint64_t exchange_count;          /* elements of the payload type in the exchange buffer: [delta | touched]   */
    uint64_t positions_len;           /* n_nodes (1D) or n_nodes*2*D                                              */
    uint64_t windows;                 /* merge windows completed                                                  */
    double   last_merge_kernels_ms;   /* prepare + apply kernels of the last window (HIP events)                  */
    uint32_t idle;                    /* 1: no term updates here (no multi-step path in the shard, or quota 0)    */
    uint32_t _pad;
} gfs_rank_info;

/* host-only planning (no device): also what a caller needs to drive gfs_ctx ranks itself */
int gfs_shard_paths(const gfs_graph_view *g, uint32_t world, uint32_t sharding, uint32_t *path_owner /*[n_paths]*/,
                    uint64_t *rank_steps /*[world]*/);
int gfs_shard_quotas(uint64_t term_updates, const uint64_t *rank_steps, uint32_t world, uint64_t *rank_quota /*[world]*/);
int gfs_shared_node_layout(const gfs_graph_view *g, uint32_t *perm /*[n_nodes]*/);
int gfs_exchange_plan(const gfs_graph_view *g, const uint32_t *perm, const uint32_t *path_owner, uint32_t world,
                      uint64_t *span_lo, uint64_t *span_hi /*[world]*/,
                      uint64_t *seg_lo, uint64_t *seg_hi /*[2*world]*/, uint32_t *n_seg,
                      uint64_t *own_lo, uint64_t *own_hi, uint32_t *own_rank /*[2*world+2] or NULL*/, uint32_t *n_own);
/* (own_*: every slot has exactly one designated owner — the lowest rank whose span covers it, rank 0 for slots no span covers —
 * so the intervals tile [0, n_nodes) and gfs_rank_finish_* leaves untouched nodes at their start positions, as sgd.rs:286-294 does) */

/* one rank.  g is the WHOLE graph (every rank derives the same plan from it and keeps only its shard on the device);
 * dims = 0: path_linear_sgd, else path_linear_sgd_layout with that many dimensions.                                  */
int      gfs_rank_create(const gfs_graph_view *g, const gfs_sgd_params *p, uint64_t dims, const gfs_rank_config *cfg,
                         gfs_rank **out);
void     gfs_rank_destroy(gfs_rank *r);
gfs_ctx *gfs_rank_ctx(gfs_rank *r);                               /* the rank's context (stats, raw device pointer)  */
int      gfs_rank_get_info(const gfs_rank *r, gfs_rank_info *out);
int      gfs_rank_set_positions(gfs_rank *r, const double *host, uint64_t n);  /* ABI order; NULL = the reference's 1D start */
int      gfs_rank_positions_changed(gfs_rank *r, void *hip_stream);            /* after writing the device buffer directly */
int      gfs_rank_get_positions(gfs_rank *r, double *host, uint64_t n);        /* complete after gfs_rank_finish_*    */
uint64_t gfs_rank_exchange_count(const gfs_rank *r);
void    *gfs_rank_exchange_buffer(gfs_rank *r);                   /* device pointer (allocated on first use)         */
int      gfs_rank_bind_exchange_buffer(gfs_rank *r, void *device_ptr);         /* use the caller's (a torch tensor)   */
/* a merge window: [this rank's share of iterations ks[0..n), moves -> buffer] | caller: all-reduce(sum) | [apply]    */
int      gfs_rank_window_begin(gfs_rank *r, const uint64_t *ks, uint64_t n, void *hip_stream);
int      gfs_rank_window_end(gfs_rank *r, void *hip_stream);
/* completion: [owned slots -> full (NULL: library scratch, gfs_rank_finish_buffer)] | all-reduce(sum, f64,
 * positions_len) | [full -> positions]                                                                               */
int      gfs_rank_finish_begin(gfs_rank *r, double *full_device, void *hip_stream);
double  *gfs_rank_finish_buffer(gfs_rank *r);
int      gfs_rank_finish_end(gfs_rank *r, const double *full_device, void *hip_stream);
/* the whole schedule.  allreduce must sum device_buf[0..count) over all ranks in place, ordered on hip_stream, and
 * return 0.  Never called when world == 1.                                                                           */
typedef int (*gfs_allreduce_fn)(void *user, void *device_buf, uint64_t count, int is_f64, void *hip_stream);
int      gfs_rank_run(gfs_rank *r, gfs_allreduce_fn allreduce, void *user, void *hip_stream);

/* ---- round-1 whole-vector merge kernels (device pointers), kept for callers that drive gfs_ctx ranks themselves ----
 *   gfs_merge_prepare: buf[0..n) = (float)(x - x_prev), buf[n..2n) = (delta != 0)
 *   all-reduce(sum) of buf over the ranks
 *   gfs_merge_apply  : x_prev += sum_delta / max(1, sum_touched)  (divide_all_by = 0), or
 *                      x_prev += sum_delta / divide_all_by         (1 = plain sum, R = mean); x = x_prev */
int gfs_merge_prepare(const double *x, const double *x_prev, float *buf2n, uint64_t n, void *hip_stream);
int gfs_merge_apply(double *x, double *x_prev, const float *buf2n, uint64_t n, double divide_all_by, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
