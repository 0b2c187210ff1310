/*
 * gfs_oracle.c — CPU ORACLE (test infrastructure only; see gfs_oracle.h for the rules and
 * the "parity unpinned" statement).  Plain C11 restatement of pangenome/gfasort src/sgd.rs.
 * Every function cites the reference lines it follows as `sgd.rs:NNN`.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: Rust never contracts a*b+c into an
 * fma, so neither may we; x86-64 SSE2 doubles are IEEE binary64 like Rust's f64).
 */
#define _GNU_SOURCE
#include "gfs_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------
 * Rust `as` casts saturate and map NaN to 0 (used at sgd.rs:149,157,164).
 * ---------------------------------------------------------------------------------------- */
static inline int32_t sat_i32(double v) {
    if (v != v) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int32_t)v;
}
static inline uint64_t sat_u64(double v) {
    if (!(v > 0.0)) return 0;                       /* NaN, negatives, zeros */
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)v;
}
static inline double bits_f64(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t f64_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static double now_s(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------
 * fast_precise_pow — sgd.rs:155-182.  Integer part of the exponent by square-and-multiply,
 * fractional part by a linear map on the high 32 bits of the IEEE-754 word; the literal
 * 1072632447 is authoritative (the source comment calling it 0x3FF00000 is wrong).
 * `high - 1072632447` wraps like release-mode Rust i32 arithmetic.
 * ---------------------------------------------------------------------------------------- */
double gfo_fast_precise_pow(double a, double b) {
    int32_t e = sat_i32(b);                                           /* :157 */
    int32_t high = (int32_t)(f64_bits(a) >> 32);                      /* :162-163 */
    int32_t diff = (int32_t)((uint32_t)high - 1072632447u);
    int32_t new_high = sat_i32((b - (double)e) * (double)diff + 1072632447.0);  /* :164 */
    double frac_result = bits_f64(((uint64_t)(uint32_t)new_high) << 32);        /* :166-167 */
    double base = a, r = 1.0;
    int32_t ex = e;
    while (ex != 0) {                                                 /* :173-179 */
        if (ex & 1) r *= base;
        base *= base;
        ex >>= 1;                 /* arithmetic shift, as Rust i32 >>; callers keep b >= 0 */
    }
    return r * frac_result;                                           /* :181 */
}
#define fpp gfo_fast_precise_pow

/* ------------------------------------------------------------------------------------------
 * DirtyZipfian::sample — sgd.rs:128-150, with the uniform draw `u` passed in.
 * NB the second fast path returns min+1 without clamping to max (inherited).
 * ---------------------------------------------------------------------------------------- */
uint64_t gfo_dirty_zipfian(uint64_t min, uint64_t max, double theta,
                           double zeta, double zeta2theta, double u) {
    uint64_t n = max - min + 1;                                       /* :129 */
    double alpha = 1.0 / (1.0 - theta);                               /* :132 */
    double eta = (1.0 - fpp(2.0 / (double)n, 1.0 - theta))
               / (1.0 - zeta2theta / zeta);                           /* :133-134 */
    double uz = u * zeta;                                             /* :137 */
    if (uz < 1.0) return min;                                         /* :140 */
    if (uz < 1.0 + fpp(0.5, theta)) return min + 1;                   /* :143 */
    double result = (double)min + ((double)n * fpp(eta * u - eta + 1.0, alpha));  /* :148 */
    uint64_t r = sat_u64(result);
    return r < max ? r : max;                                         /* :149 */
}

/* ------------------------------------------------------------------------------------------
 * RNG — rand_xoshiro 0.7 Xoshiro256Plus, seeded by SplitMix64 (seed_from_u64), as used at
 * sgd.rs:432,829,980,1218.  Third-party crate, source not in /root/reference: restated from
 * the published generators (Blackman & Vigna).  UNPINNED by the reference.
 * ---------------------------------------------------------------------------------------- */
uint64_t gfo_splitmix64_next(uint64_t *state) {
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
void gfo_xoshiro_seed(uint64_t seed, uint64_t s[4]) {
    uint64_t sm = seed;
    for (int k = 0; k < 4; k++) s[k] = gfo_splitmix64_next(&sm);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t xo_next(uint64_t s[4]) {
    uint64_t result = s[0] + s[3];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}
uint64_t gfo_xoshiro_next(uint64_t s[4]) { return xo_next(s); }

/* rand 0.9 `Uniform::new(0, n).sample()` for usize: u32 draws (next_u64 >> 32) when
 * n <= u32::MAX, else u64 draws; widening multiply; accept iff lo >= (2^w - n) mod n. */
static inline uint64_t uniform_usize(uint64_t s[4], uint64_t n) {
    if (n <= 0xFFFFFFFFull) {
        uint32_t range = (uint32_t)n;
        uint32_t thresh = (uint32_t)(0u - range) % range;
        for (;;) {
            uint64_t m = (uint64_t)(uint32_t)(xo_next(s) >> 32) * (uint64_t)range;
            if ((uint32_t)m >= thresh) return m >> 32;
        }
    } else {
        uint64_t thresh = (0ull - n) % n;
        for (;;) {
            unsigned __int128 m = (unsigned __int128)xo_next(s) * n;
            if ((uint64_t)m >= thresh) return (uint64_t)(m >> 64);
        }
    }
}
uint64_t gfo_uniform_usize(uint64_t s[4], uint64_t n) { return uniform_usize(s, n); }
/* TEST HOOK: draw step indices with the u64 branch even when n <= u32::MAX, to exercise on small
 * graphs the sampler that graphs of more than 2^32-1 steps use (the product's GFS_F_DBG_WIDE_INDEX). */
static int g_force_wide_steps = 0;
void gfo_set_force_wide_steps(int on) { g_force_wide_steps = on; }
static inline uint64_t uniform_steps(uint64_t s[4], uint64_t n) {
    if (!g_force_wide_steps) return uniform_usize(s, n);
    uint64_t thresh = (0ull - n) % n;
    for (;;) {
        unsigned __int128 m = (unsigned __int128)xo_next(s) * n;
        if ((uint64_t)m >= thresh) return (uint64_t)(m >> 64);
    }
}
/* `Uniform::new(0, 2)` on default-i32: one u32 draw, wmul by 2, threshold 0 => top bit. */
static inline uint32_t flip(uint64_t s[4]) { return (uint32_t)(xo_next(s) >> 63); }
uint32_t gfo_flip(uint64_t s[4]) { return flip(s); }
/* `rng.random::<f64>()`: 53 high bits * 2^-53 */
static inline double random_f64(uint64_t s[4]) {
    return (double)(xo_next(s) >> 11) * (1.0 / 9007199254740992.0);
}
double gfo_random_f64(uint64_t s[4]) { return random_f64(s); }

/* ------------------------------------------------------------------------------------------
 * path_linear_sgd_schedule — sgd.rs:617-638 (w_min = 1/eta_max, w_max = 1 at :300-301).
 * ---------------------------------------------------------------------------------------- */
void gfo_schedule(const gfo_params *p, double *etas) {
    double w_min = 1.0 / p->eta_max, w_max = 1.0;
    double eta_max = 1.0 / w_min;                                     /* :627 */
    double eta_min = p->eps / w_max;                                  /* :628 */
    double lambda = log(eta_max / eta_min) / ((double)p->iter_max - 1.0);  /* :629 */
    for (uint64_t t = 0; t <= p->iter_max; t++) {                     /* :632 */
        int64_t d = (int64_t)t - (int64_t)p->iter_with_max_learning_rate;
        if (d < 0) d = -d;
        etas[t] = eta_max * exp(-lambda * (double)d);                 /* :633 */
    }
}

/* zeta table — sgd.rs:311-331 (identical copy at :871-890) */
uint64_t gfo_zeta_size(const gfo_params *p) {
    uint64_t n = (p->space <= p->space_max)
        ? p->space
        : p->space_max + (p->space - p->space_max) / p->space_quantization_step + 1;
    return n + 1;
}
void gfo_zetas(const gfo_params *p, double *zetas) {
    uint64_t len = gfo_zeta_size(p);
    for (uint64_t k = 0; k < len; k++) zetas[k] = 0.0;
    double zeta_tmp = 0.0;
    for (uint64_t i = 1; i <= p->space; i++) {
        zeta_tmp += fpp(1.0 / (double)i, p->theta);                   /* :321 */
        if (i <= p->space_max) zetas[i] = zeta_tmp;
        if (i >= p->space_max && (i - p->space_max) % p->space_quantization_step == 0) {
            uint64_t idx = p->space_max + 1 + (i - p->space_max) / p->space_quantization_step;
            if (idx < len) zetas[idx] = zeta_tmp;
        }
    }
}

/* PathIndex::from_graph — sgd.rs:34-71.  A step on a node absent from the graph adds 0 bp. */
void gfo_path_index(const gfo_graph *g, uint64_t *step_pos, uint32_t *step_path,
                    uint64_t *step_rank, uint64_t *path_len) {
    for (uint64_t pth = 0; pth < g->n_paths; pth++) {
        uint64_t position = 0;
        uint64_t b = g->path_first_step[pth], e = g->path_first_step[pth + 1];
        for (uint64_t s = b; s < e; s++) {
            if (step_pos) step_pos[s] = position;
            if (step_path) step_path[s] = (uint32_t)pth;
            if (step_rank) step_rank[s] = s - b;
            uint32_t n = g->step_node[s];
            if (n != GFO_NO_NODE) position += g->node_len[n];
        }
        if (path_len) path_len[pth] = position;
    }
}

void gfo_init_positions(const gfo_graph *g, double *x) {              /* sgd.rs:271-294 */
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; i++) { x[i] = (double)len; len += g->node_len[i]; }
}
void gfo_init_layout_dim0(const gfo_graph *g, uint64_t D, double *c) {   /* sgd.rs:832-853 */
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; i++) {
        c[i * 2 * D + 0 * D + 0] = (double)len;
        c[i * 2 * D + 1 * D + 0] = (double)(len + g->node_len[i]);
        len += g->node_len[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * rand_distr 0.5 `StandardNormal` for f64 (sgd.rs:830,841,848) — an un-vendored crate, restated from its
 * published algorithm: the 256-layer ziggurat of Marsaglia & Tsang as rand_distr implements it.  PARITY UNPINNED: the
 * crate's source and its table literals are not in the container; the tables are rebuilt here by the construction the
 * crate's generator script uses (R = 3.6541528853610088, V = 0.00492867323399; x[0] = V/f(R), x[1] = R,
 * x[i+1] = f_inv(V/x[i] + f(x[i])), x[256] = 0; f = exp(-x^2/2)), which reproduces the literals if libm agrees with the
 * Python that printed them.
 *   loop: bits = next_u64; i = bits & 0xff; u = float_with_exponent(bits >> 12, 1) - 3.0  in [-1, 1);  x = u * X[i];
 *         |x| < X[i+1] -> x;  i == 0 -> tail;  F[i+1] + (F[i] - F[i+1]) * random::<f64>() < exp(-x^2/2) -> x
 *   tail: do { x = ln(Open01) / R; y = ln(Open01) } while (-2y < x^2);  u < 0 ? x - R : R - x
 *   Open01: float_with_exponent(next_u64 >> 12, 0) - (1 - EPSILON/2)
 * ---------------------------------------------------------------------------------------- */
#define ZIG_R 3.6541528853610088
#define ZIG_V 0.00492867323399
static double zig_x[257], zig_f[257];
static int zig_ready = 0;
static void zig_build(void) {
    if (zig_ready) return;
    zig_x[0] = ZIG_V / exp(-ZIG_R * ZIG_R / 2.0);
    zig_x[1] = ZIG_R;
    for (int i = 1; i < 256; i++) zig_x[i + 1] = sqrt(-2.0 * log(ZIG_V / zig_x[i] + exp(-zig_x[i] * zig_x[i] / 2.0)));
    zig_x[256] = 0.0;
    for (int i = 0; i <= 256; i++) zig_f[i] = exp(-zig_x[i] * zig_x[i] / 2.0);
    zig_ready = 1;
}
static inline double float_with_exponent(uint64_t fraction52, int e) { return bits_f64(fraction52 | ((uint64_t)(1023 + e) << 52)); }
static inline double open01(uint64_t rng[4]) {
    return float_with_exponent(gfo_xoshiro_next(rng) >> 12, 0) - (1.0 - 2.220446049250313e-16 / 2.0);
}
void gfo_ziggurat_tables(double *x257, double *f257) {
    zig_build();
    memcpy(x257, zig_x, sizeof zig_x); memcpy(f257, zig_f, sizeof zig_f);
}
double gfo_standard_normal(uint64_t rng[4]) {
    zig_build();
    for (;;) {
        const uint64_t bits = gfo_xoshiro_next(rng);
        const unsigned i = (unsigned)(bits & 0xff);
        const double u = float_with_exponent(bits >> 12, 1) - 3.0;
        const double x = u * zig_x[i];
        if (fabs(x) < zig_x[i + 1]) return x;
        if (i == 0) {
            double tx = 1.0, ty = 0.0;
            while (-2.0 * ty < tx * tx) {
                const double x_ = open01(rng), y_ = open01(rng);
                tx = log(x_) / ZIG_R; ty = log(y_);
            }
            return u < 0.0 ? tx - ZIG_R : ZIG_R - tx;
        }
        if (zig_f[i + 1] + (zig_f[i] - zig_f[i + 1]) * gfo_random_f64(rng) < exp(-x * x / 2.0)) return x;
    }
}
/* the whole start of path_linear_sgd_layout (sgd.rs:829-853): ONE generator seeded `seed`; per node: + end dims 1..D-1,
 * then - end dims 1..D-1; dim 0 = bp prefix / prefix + length.  coords in Layout order [node][end][dim]. */
void gfo_init_layout(const gfo_graph *g, uint64_t D, uint64_t seed, double *c) {
    uint64_t rng[4]; gfo_xoshiro_seed(seed, rng);                     /* :829 */
    const double sqrt_n = sqrt((double)g->n_nodes * 2.0);             /* :836 */
    uint64_t len = 0;
    for (uint64_t i = 0; i < g->n_nodes; i++) {
        c[i * 2 * D + 0] = (double)len;                               /* :839 */
        for (uint64_t d = 1; d < D; d++) c[i * 2 * D + d] = gfo_standard_normal(rng) * sqrt_n;          /* :840-843 */
        c[i * 2 * D + D] = (double)(len + g->node_len[i]);            /* :846 */
        for (uint64_t d = 1; d < D; d++) c[i * 2 * D + D + d] = gfo_standard_normal(rng) * sqrt_n;      /* :847-850 */
        len += g->node_len[i];
    }
}

/* ------------------------------------------------------------------------------------------
 * Internal PathIndex in two representations.
 *  reference-like: four 8-byte arrays + PathInfo + a hash map handle->idx (sgd.rs:14-31,272)
 *  flat          : 16-byte step records + 16-byte path records, dense idx stored in the step
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t step_count, length, first_step; } path_info;
typedef struct { uint32_t node; uint32_t path_rev; uint64_t pos; } step_rec;   /* flat */
typedef struct { uint64_t key; uint64_t val; } hslot;

typedef struct pidx {
    uint64_t n_steps, n_paths, n_nodes;
    /* reference-like */
    uint64_t *step_to_handle, *step_to_position, *step_to_path, *step_to_rank;
    path_info *paths;
    hslot *hmap; uint64_t hmask;          /* Handle::forward(node_id) -> idx */
    uint64_t *node_seq_len;               /* graph.nodes[id].sequence.len() by "node id" */
    /* flat */
    step_rec *rec;
    const uint32_t *node_len;
} pidx;

static inline uint64_t hmix(uint64_t k) {  /* stand-in for SipHash: any decent 64-bit mixer */
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
static inline int64_t hmap_get(const pidx *pi, uint64_t key) {
    uint64_t h = hmix(key) & pi->hmask;
    for (;;) {
        if (pi->hmap[h].key == key) return (int64_t)pi->hmap[h].val;
        if (pi->hmap[h].key == UINT64_MAX) return -1;
        h = (h + 1) & pi->hmask;
    }
}

static void pidx_free(pidx *pi) {
    free(pi->step_to_handle); free(pi->step_to_position); free(pi->step_to_path);
    free(pi->step_to_rank); free(pi->paths); free(pi->hmap); free(pi->node_seq_len); free(pi->rec);
    memset(pi, 0, sizeof *pi);
}

/* In the reference-like form the "node id" of dense idx k is k+1 (ids are only labels here;
 * what matters is that the lookup goes handle -> hash map -> idx as at sgd.rs:525-538). */
static int pidx_build(const gfo_graph *g, int flat, pidx *pi) {
    memset(pi, 0, sizeof *pi);
    pi->n_steps = g->n_steps; pi->n_paths = g->n_paths; pi->n_nodes = g->n_nodes;
    pi->node_len = g->node_len;
    uint64_t S = g->n_steps, P = g->n_paths;
    pi->paths = (path_info *)calloc(P ? P : 1, sizeof(path_info));
    uint64_t *pos = (uint64_t *)malloc((S ? S : 1) * 8);
    uint32_t *pth = (uint32_t *)malloc((S ? S : 1) * 4);
    uint64_t *plen = (uint64_t *)malloc((P ? P : 1) * 8);
    if (!pi->paths || !pos || !pth || !plen) return -1;
    gfo_path_index(g, pos, pth, NULL, plen);
    for (uint64_t p = 0; p < P; p++) {
        pi->paths[p].first_step = g->path_first_step[p];
        pi->paths[p].step_count = g->path_first_step[p + 1] - g->path_first_step[p];
        pi->paths[p].length = plen[p];
    }
    if (flat) {
        pi->rec = (step_rec *)malloc((S ? S : 1) * sizeof(step_rec));
        if (!pi->rec) return -1;
        for (uint64_t s = 0; s < S; s++) {
            pi->rec[s].node = g->step_node[s];
            pi->rec[s].path_rev = pth[s] | ((uint32_t)(g->step_is_rev[s] & 1) << 31);
            pi->rec[s].pos = pos[s];
        }
    } else {
        pi->step_to_handle = (uint64_t *)malloc((S ? S : 1) * 8);
        pi->step_to_position = pos; pos = NULL;
        pi->step_to_path = (uint64_t *)malloc((S ? S : 1) * 8);
        pi->step_to_rank = (uint64_t *)malloc((S ? S : 1) * 8);
        uint64_t cap = 16; while (cap < 2 * g->n_nodes + 2) cap <<= 1;
        pi->hmap = (hslot *)malloc(cap * sizeof(hslot)); pi->hmask = cap - 1;
        pi->node_seq_len = (uint64_t *)calloc(g->n_nodes + 2, 8);
        if (!pi->step_to_handle || !pi->step_to_path || !pi->step_to_rank || !pi->hmap || !pi->node_seq_len) return -1;
        for (uint64_t k = 0; k < cap; k++) pi->hmap[k].key = UINT64_MAX;
        for (uint64_t k = 0; k < g->n_nodes; k++) {
            uint64_t key = (k + 1) << 1;            /* Handle::forward(id) = id<<1 (graph.rs:13-23) */
            uint64_t h = hmix(key) & pi->hmask;
            while (pi->hmap[h].key != UINT64_MAX) h = (h + 1) & pi->hmask;
            pi->hmap[h].key = key; pi->hmap[h].val = k;
            pi->node_seq_len[k + 1] = g->node_len[k];
        }
        for (uint64_t s = 0; s < S; s++) {
            uint32_t n = g->step_node[s];
            uint64_t id = (n == GFO_NO_NODE) ? (g->n_nodes + 1) : ((uint64_t)n + 1);  /* absent id */
            pi->step_to_handle[s] = (id << 1) | (g->step_is_rev[s] & 1);
            pi->step_to_path[s] = pth[s];
            pi->step_to_rank[s] = s - g->path_first_step[pth[s]];
        }
    }
    free(pos); free(pth); free(plen);
    return 0;
}

/* per-iteration shared state (the reference's atomics eta / adj_theta / cooling) */
typedef struct iter_state {
    double eta, theta;
    int cooling;
} iter_state;

typedef struct zipf_env {
    const double *zetas; uint64_t zlen;
    uint64_t space, space_max, q;
} zipf_env;

/* zeta index rule — sgd.rs:463-469 */
static inline uint64_t space_index(const zipf_env *z, uint64_t jump_space) {
    uint64_t idx = (jump_space > z->space_max)
        ? z->space_max + (jump_space - z->space_max) / z->q + 1
        : jump_space;
    return idx < z->zlen - 1 ? idx : z->zlen - 1;
}

/* One trip of the sampler, sgd.rs:444-499 (1D) == :990-1037 (nD): draws step a, decides
 * Zipf/uniform, draws rank b.  Returns 0 on `continue`, 1 with *sa,*sb = global step indices. */
static inline __attribute__((always_inline)) int
sample_pair(const pidx *pi, const int flat, const zipf_env *z, const iter_state *it,
            uint64_t rng[4], uint64_t *sa, uint64_t *sb) {
    uint64_t step_idx = uniform_steps(rng, pi->n_steps);              /* :444 */
    uint64_t path_idx, rank_a;
    if (flat) { path_idx = pi->rec[step_idx].path_rev & 0x7FFFFFFFu; }
    else      { path_idx = pi->step_to_path[step_idx]; }              /* :445 */
    uint64_t cnt = pi->paths[path_idx].step_count;                    /* :446 */
    if (cnt == 1) return 0;                                           /* :448 */
    if (flat) rank_a = step_idx - pi->paths[path_idx].first_step;
    else      rank_a = pi->step_to_rank[step_idx];                    /* :452 */
    uint64_t rank_b = rank_a;
    if (it->cooling || flip(rng) == 1) {                              /* :456 short-circuit */
        double theta = it->theta;                                     /* :458 */
        if (rank_a > 0 && (flip(rng) == 1 || rank_a == cnt - 1)) {    /* :460 */
            uint64_t jump = z->space < rank_a ? z->space : rank_a;    /* :462 */
            uint64_t si = space_index(z, jump);
            double zeta2theta = 1.0 + fpp(0.5, theta);                /* :471 */
            uint64_t zi = gfo_dirty_zipfian(1, jump, theta, z->zetas[si], zeta2theta, random_f64(rng));
            rank_b = rank_a >= zi ? rank_a - zi : 0;                  /* :474 saturating_sub */
        } else if (rank_a < cnt - 1) {                                /* :475 */
            uint64_t rem = cnt - rank_a - 1;
            uint64_t jump = z->space < rem ? z->space : rem;          /* :477 */
            uint64_t si = space_index(z, jump);
            double zeta2theta = 1.0 + fpp(0.5, theta);                /* :486 */
            uint64_t zi = gfo_dirty_zipfian(1, jump, theta, z->zetas[si], zeta2theta, random_f64(rng));
            uint64_t rb = rank_a + zi;
            rank_b = rb < cnt - 1 ? rb : cnt - 1;                     /* :489 */
        }
    } else {
        rank_b = uniform_usize(rng, cnt);                             /* :493-494 */
    }
    if (rank_a == rank_b) return 0;                                   /* :497 */
    *sa = pi->paths[path_idx].first_step + rank_a;                    /* :502 */
    *sb = pi->paths[path_idx].first_step + rank_b;                    /* :503 */
    return 1;
}

/* 1D term — sgd.rs:505-576.  xa/xb are the position slots; `atomic`: relaxed atomics like the
 * reference's AtomicU64 (threaded mode) or plain doubles (deterministic mode). */
static inline __attribute__((always_inline)) int
term_1d(const pidx *pi, const int flat, const iter_state *it, uint64_t sa, uint64_t sb,
        double *x, const int atomic, _Atomic uint64_t *delta_max, gfo_term *tr) {
    double pos_a, pos_b; uint64_t i, j;
    if (flat) {
        pos_a = (double)pi->rec[sa].pos; pos_b = (double)pi->rec[sb].pos;
    } else {
        pos_a = (double)pi->step_to_position[sa]; pos_b = (double)pi->step_to_position[sb];  /* :509-510 */
    }
    double term_dist = fabs(pos_a - pos_b);                           /* :513 */
    if (term_dist == 0.0) return 0;                                   /* :514 */
    double term_weight = 1.0 / term_dist;                             /* :518 */
    double mu = it->eta * term_weight;                                /* :519 */
    mu = fmin(mu, 1.0);                                               /* :520 f64::min ignores a NaN operand, like fmin */
    if (flat) {
        uint32_t ni = pi->rec[sa].node, nj = pi->rec[sb].node;
        if (ni == GFO_NO_NODE || nj == GFO_NO_NODE) return 0;
        i = ni; j = nj;
    } else {
        int64_t li = hmap_get(pi, pi->step_to_handle[sa] & ~1ull);    /* :525 forward handle */
        if (li < 0) return 0;
        int64_t lj = hmap_get(pi, pi->step_to_handle[sb] & ~1ull);    /* :532 */
        if (lj < 0) return 0;
        i = (uint64_t)li; j = (uint64_t)lj;
    }
    _Atomic uint64_t *ax = (_Atomic uint64_t *)x;
    double x_i, x_j;
    if (atomic) {
        x_i = bits_f64(atomic_load_explicit(&ax[i], memory_order_relaxed));
        x_j = bits_f64(atomic_load_explicit(&ax[j], memory_order_relaxed));
    } else { x_i = x[i]; x_j = x[j]; }                                /* :541-542 */
    double dx = x_i - x_j;                                            /* :543 */
    if (dx == 0.0) dx = 1e-9;                                         /* :546-548 */
    double mag = fabs(dx);                                            /* :551 */
    double delta_update = mu * (mag - term_dist) / 2.0;               /* :552 */
    if (delta_max) {                                                  /* :555-567 dead value, kept for cost parity */
        double delta_abs = fabs(delta_update);
        uint64_t cur = atomic_load_explicit(delta_max, memory_order_relaxed);
        while (delta_abs > bits_f64(cur)) {
            if (atomic_compare_exchange_weak_explicit(delta_max, &cur, f64_bits(delta_abs),
                                                      memory_order_relaxed, memory_order_relaxed)) break;
        }
    }
    double r = delta_update / mag;                                    /* :570 */
    double r_x = r * dx;                                              /* :571 */
    if (atomic) {                                                     /* :575-576 load then store, racy by design */
        double vi = bits_f64(atomic_load_explicit(&ax[i], memory_order_relaxed));
        atomic_store_explicit(&ax[i], f64_bits(vi - r_x), memory_order_relaxed);
        double vj = bits_f64(atomic_load_explicit(&ax[j], memory_order_relaxed));
        atomic_store_explicit(&ax[j], f64_bits(vj + r_x), memory_order_relaxed);
    } else {
        x[i] = x[i] - r_x;
        x[j] = x[j] + r_x;
    }
    if (tr) { tr->i = (uint32_t)i; tr->j = (uint32_t)j; tr->d_ij = term_dist; }
    return 1;
}

/* nD term — sgd.rs:1043-1149.  coords in Layout order [node][end][dim]. */
#define GFO_MAX_DIMS 16
static inline __attribute__((always_inline)) int
term_nd(const pidx *pi, const int flat, const iter_state *it, uint64_t rng[4], uint64_t sa, uint64_t sb,
        double *c, const uint64_t D, const int atomic, _Atomic uint64_t *delta_max, gfo_term *tr) {
    double pos_a, pos_b, len_i, len_j; int rev_i, rev_j; int64_t li, lj;
    if (flat) {
        pos_a = (double)pi->rec[sa].pos; pos_b = (double)pi->rec[sb].pos;
        uint32_t ni = pi->rec[sa].node, nj = pi->rec[sb].node;
        len_i = ni == GFO_NO_NODE ? 0.0 : (double)pi->node_len[ni];
        len_j = nj == GFO_NO_NODE ? 0.0 : (double)pi->node_len[nj];
        rev_i = (int)(pi->rec[sa].path_rev >> 31); rev_j = (int)(pi->rec[sb].path_rev >> 31);
        li = ni == GFO_NO_NODE ? -1 : (int64_t)ni; lj = nj == GFO_NO_NODE ? -1 : (int64_t)nj;
    } else {
        uint64_t hi = pi->step_to_handle[sa], hj = pi->step_to_handle[sb];      /* :1043-1044 */
        pos_a = (double)pi->step_to_position[sa]; pos_b = (double)pi->step_to_position[sb];
        len_i = (double)pi->node_seq_len[hi >> 1];                   /* :1051-1058 (0 when absent) */
        len_j = (double)pi->node_seq_len[hj >> 1];
        rev_i = (int)(hi & 1); rev_j = (int)(hj & 1);
        li = lj = -2;
    }
    int use_other_end_a = flip(rng) == 1;                             /* :1062 */
    if (use_other_end_a) { pos_a += len_i; use_other_end_a = !rev_i; }
    else                 { use_other_end_a = rev_i; }                 /* :1063-1068 */
    int use_other_end_b = flip(rng) == 1;                             /* :1071 */
    if (use_other_end_b) { pos_b += len_j; use_other_end_b = !rev_j; }
    else                 { use_other_end_b = rev_j; }
    double term_dist = fabs(pos_a - pos_b);                           /* :1080 */
    if (term_dist == 0.0) return 0;
    double mu = fmin(it->eta * (1.0 / term_dist), 1.0);               /* :1085-1086 */
    if (!flat) {
        li = hmap_get(pi, pi->step_to_handle[sa] & ~1ull);            /* :1089 */
        if (li < 0) return 0;
        lj = hmap_get(pi, pi->step_to_handle[sb] & ~1ull);
        if (lj < 0) return 0;
    } else if (li < 0 || lj < 0) return 0;
    uint64_t idx_i = (uint64_t)li * 2 + (use_other_end_a ? 1 : 0);    /* :1099-1103 */
    uint64_t idx_j = (uint64_t)lj * 2 + (use_other_end_b ? 1 : 0);
    double *ci = c + idx_i * D, *cj = c + idx_j * D;
    _Atomic uint64_t *ai = (_Atomic uint64_t *)ci, *aj = (_Atomic uint64_t *)cj;
    double deltas[GFO_MAX_DIMS];
    double mag_sq = 0.0;
    for (uint64_t d = 0; d < D; d++) {                                /* :1108-1113 */
        double vi = atomic ? bits_f64(atomic_load_explicit(&ai[d], memory_order_relaxed)) : ci[d];
        double vj = atomic ? bits_f64(atomic_load_explicit(&aj[d], memory_order_relaxed)) : cj[d];
        deltas[d] = vi - vj;
        mag_sq += deltas[d] * deltas[d];
    }
    if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }         /* :1116-1119 */
    double mag = sqrt(mag_sq);                                        /* :1121 */
    double delta_update = mu * (mag - term_dist) / 2.0;               /* :1125 */
    if (delta_max) {
        double delta_abs = fabs(delta_update);
        uint64_t cur = atomic_load_explicit(delta_max, memory_order_relaxed);
        while (delta_abs > bits_f64(cur)) {
            if (atomic_compare_exchange_weak_explicit(delta_max, &cur, f64_bits(delta_abs),
                                                      memory_order_relaxed, memory_order_relaxed)) break;
        }
    }
    double r = delta_update / mag;                                    /* :1142 */
    for (uint64_t d = 0; d < D; d++) {                                /* :1143-1149 */
        double r_d = r * deltas[d];
        if (atomic) {
            double vi = bits_f64(atomic_load_explicit(&ai[d], memory_order_relaxed));
            double vj = bits_f64(atomic_load_explicit(&aj[d], memory_order_relaxed));
            atomic_store_explicit(&ai[d], f64_bits(vi - r_d), memory_order_relaxed);
            atomic_store_explicit(&aj[d], f64_bits(vj + r_d), memory_order_relaxed);
        } else {
            double vi = ci[d], vj = cj[d];       /* both loaded before either store (matters when idx_i==idx_j) */
            ci[d] = vi - r_d;
            cj[d] = vj + r_d;
        }
    }
    if (tr) { tr->i = (uint32_t)idx_i; tr->j = (uint32_t)idx_j; tr->d_ij = term_dist; }
    return 1;
}

/* iteration constants: batch k uses etas[k]; cooling for k > floor(cooling_start*iter_max)
 * (sgd.rs:297, 383-396): eta/theta/cooling are switched when the checker moves to new_iter=k. */
static void iter_consts(const gfo_params *p, const double *etas, uint64_t k, iter_state *it) {
    uint64_t first_cooling = sat_u64(floor(p->cooling_start * (double)p->iter_max));
    it->eta = etas[k];
    it->cooling = (k > first_cooling);
    it->theta = it->cooling ? 0.001 : p->theta;
}

static int has_valid_paths(const gfo_graph *g) {                      /* sgd.rs:250-261 */
    for (uint64_t p = 0; p < g->n_paths; p++)
        if (g->path_first_step[p + 1] - g->path_first_step[p] > 1) return 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Deterministic mode, as a resumable state: streams keep their RNG across iterations.
 * ---------------------------------------------------------------------------------------- */
struct gfo_state {
    pidx pi;
    gfo_params p;
    double *etas, *zetas;
    zipf_env z;
    uint64_t T, D, quota_total, attempt_factor, bundle;
    uint64_t *rng, *done, *att, *ntr;
    struct leader_s *lead; uint8_t *lead_left, *lead_cool, *lead_colour, *lead_seg;   /* 1D bundled mode: each wave's partly expanded pass */
    uint64_t chain;                                           /* longest run in trips (B = 64); mirror of GFS_F_CHAIN */
    int one_colour;                                           /* mirror of GFS_F_DBG_ONE_COLOUR */
    int no_fused_trip;                                        /* mirror of GFS_F_DBG_NO_FUSED_TRIP */
    uint64_t chunk;                                           /* updates per chunk of a team wave's work (product: KArgs.chunk; default GFO_TEAM_CHUNK) */
    int partners, no_twin_trip;                               /* partner draws per leader (1, 2); mirror of GFS_F_DBG_NO_TWIN_TRIP */
    uint8_t *lead_p;                                          /* the partner each wave's next trip belongs to */
    uint8_t *lead_flips;                                      /* nD: the end flips every leader of the pass drew (bit 0: a, 1: b, 2: the second partner's b) */
    uint32_t *node_slot;                                     /* bundled mode: the product's internal node layout (line-aligned runs) */
    gfo_term *trace; uint64_t trace_per_stream;
    uint64_t total_upd, total_att, iterations;
    double seconds;
};

void gfo_state_destroy(gfo_state *s) {
    if (!s) return;
    pidx_free(&s->pi);
    free(s->etas); free(s->zetas); free(s->rng); free(s->done); free(s->att); free(s->ntr);
    free(s->lead); free(s->lead_left); free(s->lead_cool); free(s->lead_colour); free(s->lead_seg); free(s->lead_p); free(s->lead_flips); free(s->node_slot);
    free(s);
}

int gfo_state_create(const gfo_graph *g, const gfo_params *p, const double *etas_in, const double *zetas_in,
                     uint64_t dims, uint64_t n_streams, uint64_t stream_base, uint64_t quota_total,
                     uint64_t attempt_factor, gfo_term *trace, uint64_t trace_per_stream, gfo_state **out) {
    *out = NULL;
    if (g->n_nodes == 0 || !has_valid_paths(g)) return 1;             /* sgd.rs:242-244, 258-261 */
    if (n_streams == 0 || dims > GFO_MAX_DIMS) return -1;
    gfo_state *s = (gfo_state *)calloc(1, sizeof *s);
    if (!s) return -2;
    if (pidx_build(g, 1, &s->pi)) { gfo_state_destroy(s); return -2; }
    s->p = *p; s->T = n_streams; s->D = dims; s->bundle = 1; s->chain = 1; s->partners = 1; s->chunk = 0;
    s->quota_total = quota_total ? quota_total : p->min_term_updates;
    s->attempt_factor = attempt_factor ? attempt_factor : 64;
    s->etas = (double *)malloc((p->iter_max + 1) * 8);
    s->zetas = (double *)malloc(gfo_zeta_size(p) * 8);
    if (etas_in) memcpy(s->etas, etas_in, (p->iter_max + 1) * 8); else gfo_schedule(p, s->etas);
    if (zetas_in) memcpy(s->zetas, zetas_in, gfo_zeta_size(p) * 8); else gfo_zetas(p, s->zetas);
    s->z = (zipf_env){ s->zetas, gfo_zeta_size(p), p->space, p->space_max, p->space_quantization_step };
    s->rng = (uint64_t *)malloc(n_streams * 32);
    s->done = (uint64_t *)malloc(n_streams * 8); s->att = (uint64_t *)malloc(n_streams * 8);
    s->ntr = (uint64_t *)calloc(n_streams, 8);
    for (uint64_t t = 0; t < n_streams; t++)
        gfo_xoshiro_seed(p->seed + stream_base + t, s->rng + 4 * t);  /* sgd.rs:431-432 */
    s->trace = trace; s->trace_per_stream = trace_per_stream;
    *out = s;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Bundled ("run") sampling — NOT in the reference.  This mirrors, decision for decision, the
 * product's coalesced sampler (gfasort_amd/csrc/sgd_device.h sample_leader / expand_run and the
 * pass/trip structure and wave-level quota of sgd1d_team_kernel) so that its random-number
 * consumption and the terms it emits can be checked bit for bit.  Every stream samples leader
 * terms exactly like a reference worker (sgd.rs:444-497); a leader is expanded into a run of B
 * lanes taking consecutive steps with the leader's signed jump (nD: and the leader's two end
 * flips).  Updates of one trip are applied here in lane order; the GPU applies them concurrently
 * (and one trip late).
 * ---------------------------------------------------------------------------------------- */
/* The product aligns runs to the 64-B lines of ITS position vector, i.e. by its internal node layout
 * (gfs_ctx_node_layout): slot[k] for dense node k.  NULL = identity. */
int gfo_state_set_node_slots(gfo_state *s, const uint32_t *slot) {
    if (!s) return -1;
    free(s->node_slot); s->node_slot = NULL;
    if (slot) {
        s->node_slot = (uint32_t *)malloc(s->pi.n_nodes * sizeof(uint32_t));
        if (!s->node_slot) return -2;
        memcpy(s->node_slot, slot, s->pi.n_nodes * sizeof(uint32_t));
    }
    return 0;
}

int gfo_state_set_bundle(gfo_state *s, uint64_t bundle) {
    if (!s) return -1;
    if (bundle == 0) bundle = 1;
    if (bundle != 1 && (s->T % 64 != 0 || (bundle != 4 && bundle != 8 && bundle != 16 && bundle != 32 && bundle != 64)))
        return -1;
    s->bundle = bundle;
    return 0;
}

/* Long runs (product: sgd_device.h run_trips): with B = 64 in 1D a leader is expanded over K consecutive trips of its
 * wave, K = the largest power of two <= chain with K*B <= cnt/4.  chain = 1: a run is one trip.  1D and nD alike. */
int gfo_state_set_chain(gfo_state *s, uint64_t chain) {
    if (!s || chain == 0 || chain > 64 || (chain & (chain - 1))) return -1;
    s->chain = chain;
    return 0;
}
static uint64_t run_trips(const gfo_state *s, uint64_t cnt) {
    if (s->bundle != 64) return 1;
    const uint64_t room = cnt / (4 * s->bundle);
    if (room < 2 || s->chain < 2) return 1;
    uint64_t p2 = 1; while (p2 * 2 <= room) p2 *= 2;
    return p2 < s->chain ? p2 : s->chain;
}
/* where trip seg of a run starts, in steps after the run's first step: long-jump runs are contiguous, a leader whose
 * jump is shorter than a trip spreads its trips evenly over the path (product: sgd_device.h run_offset) */
static uint64_t run_offset(const gfo_state *s, uint64_t cnt, uint64_t k, uint64_t ra0, uint64_t rb0, uint64_t seg) {
    if (s->bundle != 64) return 0;
    const uint64_t z = ra0 < rb0 ? rb0 - ra0 : ra0 - rb0;
    return seg * (z < s->bundle ? cnt / k : s->bundle);
}

/* mirror of GFS_F_DBG_ONE_COLOUR: short-jump runs execute their first colour only (the round-1 sampler) */
int gfo_state_set_one_colour(gfo_state *s, int on) {
    if (!s) return -1;
    s->one_colour = on ? 1 : 0;
    return 0;
}

/* Two partners per leader (product: 1D team kernel at B = 64 unless GFS_F_ONE_PARTNER); no_twin: mirror of
 * GFS_F_DBG_NO_TWIN_TRIP — the two partners of an aligned leader as two trips. */
int gfo_state_set_partners(gfo_state *s, int partners, int no_twin) {
    if (!s || (partners != 1 && partners != 2)) return -1;
    s->partners = partners; s->no_twin_trip = no_twin ? 1 : 0;
    return 0;
}

/* updates per chunk (product: a pooled launch of ONE small iteration works in chunks of 256...2048, capi.hip gfs_ctx_run_range) */
int gfo_state_set_chunk(gfo_state *s, uint64_t chunk) {
    if (!s || chunk == 0) return -1;
    s->chunk = chunk;
    return 0;
}

/* mirror of GFS_F_DBG_NO_FUSED_TRIP: the two colours of a short-jump trip as two trips (1D) */
int gfo_state_set_no_fused_trip(gfo_state *s, int on) {
    if (!s) return -1;
    s->no_fused_trip = on ? 1 : 0;
    return 0;
}

/* nD term with the two end flips already drawn (team mode): same arithmetic as term_nd. */
static int nd_prepare(const pidx *pi, uint64_t sa, uint64_t sb, int fa, int fb,
                      double *term_dist, uint64_t *idx_i, uint64_t *idx_j) {
    double pos_a = (double)pi->rec[sa].pos, pos_b = (double)pi->rec[sb].pos;
    uint32_t ni = pi->rec[sa].node, nj = pi->rec[sb].node;
    double len_i = ni == GFO_NO_NODE ? 0.0 : (double)pi->node_len[ni];
    double len_j = nj == GFO_NO_NODE ? 0.0 : (double)pi->node_len[nj];
    int rev_i = (int)(pi->rec[sa].path_rev >> 31), rev_j = (int)(pi->rec[sb].path_rev >> 31);
    int oa = fa; if (oa) { pos_a += len_i; oa = !rev_i; } else oa = rev_i;             /* :1062-1068 */
    int ob = fb; if (ob) { pos_b += len_j; ob = !rev_j; } else ob = rev_j;             /* :1071-1077 */
    *term_dist = fabs(pos_a - pos_b);                                                  /* :1080 */
    if (*term_dist == 0.0 || ni == GFO_NO_NODE || nj == GFO_NO_NODE) return 0;
    *idx_i = (uint64_t)ni * 2 + (oa ? 1 : 0); *idx_j = (uint64_t)nj * 2 + (ob ? 1 : 0);
    return 1;
}
static int nd_term_ok(const pidx *pi, uint64_t sa, uint64_t sb, int fa, int fb) {
    double td; uint64_t i, j;
    return nd_prepare(pi, sa, sb, fa, fb, &td, &i, &j);
}
static int term_nd_flips(const pidx *pi, const iter_state *it, int fa, int fb, uint64_t sa, uint64_t sb,
                         double *c, uint64_t D, gfo_term *tr) {
    double term_dist; uint64_t idx_i, idx_j;
    if (!nd_prepare(pi, sa, sb, fa, fb, &term_dist, &idx_i, &idx_j)) return 0;
    double mu = fmin(it->eta * (1.0 / term_dist), 1.0);
    double *ci = c + idx_i * D, *cj = c + idx_j * D;
    double deltas[GFO_MAX_DIMS], mag_sq = 0.0;
    for (uint64_t d = 0; d < D; d++) { deltas[d] = ci[d] - cj[d]; mag_sq += deltas[d] * deltas[d]; }
    if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }
    double mag = sqrt(mag_sq);
    double r = (mu * (mag - term_dist) / 2.0) / mag;
    for (uint64_t d = 0; d < D; d++) {
        double r_d = r * deltas[d], vi = ci[d], vj = cj[d];
        ci[d] = vi - r_d; cj[d] = vj + r_d;
    }
    if (tr) { tr->i = (uint32_t)idx_i; tr->j = (uint32_t)idx_j; tr->d_ij = term_dist; }
    return 1;
}

/* A leader: step a and one partner b — or, with two partners (product: sgd_device.h Leader, KArgs.partners = 2), the same
 * step a with two independent draws of b.  (ra0, rb0, ok, aligned, rot) is the partner a trip works on: partner_view. */
#define GFO_TEAM_CHUNK 2048u
#define GFO_ND_TEAM_CHUNK 4096u          /* product: sgd_device.h ND_TEAM_CHUNK (layout kernels) */
typedef struct leader_s { uint64_t first, cnt, ra0, rb0; int ok, aligned, rot; uint64_t ra1, rb1; int ok1, aligned1, rot1; } leader_t;
static leader_t partner_view(const leader_t *L, uint64_t p) {
    leader_t v = *L;
    if (p) { v.ra0 = L->ra1; v.rb0 = L->rb1; v.ok = L->ok1; v.aligned = L->aligned1; v.rot = L->rot1; }
    return v;
}

/* one draw of the partner of step a (rank *ra in its path): sgd.rs:456-497 without applying it, then the product's line
 * alignment of the two runs (sgd_device.h draw_partner, which explains the rule) */
static void draw_partner(const gfo_state *s, const iter_state *it, uint64_t *rng, uint64_t cnt, uint32_t node0,
                         uint64_t *ra, uint64_t *rb, int *ok, int *aligned, int *rot) {
    const uint64_t a0 = *ra;
    *rb = a0; *aligned = 0; *rot = 0;
    if (it->cooling || flip(rng) == 1) {                                               /* :456 */
        if (a0 > 0 && (flip(rng) == 1 || a0 == cnt - 1)) {                            /* :460 */
            uint64_t jump = s->z.space < a0 ? s->z.space : a0;
            double z2 = 1.0 + fpp(0.5, it->theta);
            uint64_t zi = gfo_dirty_zipfian(1, jump, it->theta, s->z.zetas[space_index(&s->z, jump)], z2, random_f64(rng));
            *rb = a0 >= zi ? a0 - zi : 0;
        } else if (a0 < cnt - 1) {
            uint64_t room = cnt - a0 - 1;
            uint64_t jump = s->z.space < room ? s->z.space : room;
            double z2 = 1.0 + fpp(0.5, it->theta);
            uint64_t zi = gfo_dirty_zipfian(1, jump, it->theta, s->z.zetas[space_index(&s->z, jump)], z2, random_f64(rng));
            uint64_t b = a0 + zi;
            *rb = b < cnt - 1 ? b : cnt - 1;
        }
    } else {
        *rb = uniform_usize(rng, cnt);                                                 /* :493-494 */
    }
    *ok = *rb != a0;                                                                   /* :497 */
    /* line-aligned runs: for |jump| >= B + 8 the run starts (slot of the leader's node) mod 8 steps before the leader; the
     * partner run is the B-step block jump - r steps further on, r = jump mod 8, with lane l paired to its step (l + r) mod B */
    if (*ok && cnt >= 2 * s->bundle && node0 != GFO_NO_NODE) {             /* only where a run will be expanded */
        const int64_t jump = (int64_t)*rb - (int64_t)a0, Bn = (int64_t)s->bundle;
        const int64_t Rn = Bn * (int64_t)run_trips(s, cnt);                /* steps of the whole run */
        if (jump >= Bn + 8 || jump <= -(Bn + 8)) {                     /* shorter jumps are left alone */
            const int64_t A = Bn < 8 ? Bn : 8;                            /* runs shorter than a line: align to the run length */
            const uint64_t sh = (s->node_slot ? s->node_slot[node0] : node0) & (uint64_t)(A - 1);
            if (a0 >= sh) {
                const int64_t na = (int64_t)a0 - (int64_t)sh;
                const int64_t r = ((jump % A) + A) % A, zp = jump - r, nb = na + zp;
                if (na + Rn <= (int64_t)cnt && nb >= 0 && nb + Rn <= (int64_t)cnt && (zp >= Bn || zp <= -Bn)) {
                    *ra = (uint64_t)na; *rb = (uint64_t)nb; *aligned = 1; *rot = (int)r;
                } else if (*rb >= sh) { *ra -= sh; *rb -= sh; }
            }
        }
    }
}

/* one leader from one reference stream: sgd.rs:444-453, then one or two partner draws */
static leader_t sample_leader(const gfo_state *s, const iter_state *it, uint64_t *rng) {
    const pidx *pi = &s->pi;
    leader_t L;
    memset(&L, 0, sizeof L);
    uint64_t s0 = uniform_steps(rng, pi->n_steps);                                     /* :444 */
    uint64_t path = pi->rec[s0].path_rev & 0x7FFFFFFFu;
    L.first = pi->paths[path].first_step; L.cnt = pi->paths[path].step_count;
    L.ra0 = s0 - L.first; L.rb0 = L.ra0; L.ra1 = L.ra0; L.rb1 = L.ra0;
    if (L.cnt == 1) return L;                                                          /* :448 */
    draw_partner(s, it, rng, L.cnt, pi->rec[s0].node, &L.ra0, &L.rb0, &L.ok, &L.aligned, &L.rot);
    if (s->partners == 2) draw_partner(s, it, rng, L.cnt, pi->rec[s0].node, &L.ra1, &L.rb1, &L.ok1, &L.aligned1, &L.rot1);
    return L;
}

/* Product: sgd_kernel_common.h merged_trip_shift — a short-jump trip (B = 64, 0 < |jump| < 64) whose 64 steps and all
 * their partners lie inside the path without wrapping.  Returns the signed jump, or 0. */
static int64_t merged_trip_shift(const leader_t *ld, uint64_t off) {
    if (!ld->ok || ld->aligned || ld->cnt < 128) return 0;
    const int64_t sh = (int64_t)ld->rb0 - (int64_t)ld->ra0;
    if (sh == 0 || sh >= 64 || sh <= -64) return 0;
    uint64_t base = ld->ra0 + off;
    if (base >= ld->cnt) base -= ld->cnt;
    if (base + 64 > ld->cnt) return 0;
    if (sh > 0 ? base + 63 + (uint64_t)sh > ld->cnt - 1 : (int64_t)base + sh < 0) return 0;
    return sh;
}

/* Product: sgd_kernels_1d.hip fused_trip — both colours of such a trip in one go, as the wave executes them: every lane
 * holds the position of its own step's node in a register (and its partner's when the partner lies beyond the trip's 64
 * steps), colour 1 computes on what colour 0 left in the registers, and a lane's node receives ONE add, the sum of its
 * two contributions: x + (-r + r'), not (x - r) + r'.  Partners beyond the trip are added per colour.  Lane order here;
 * the wave applies them concurrently.  Returns 0 when the wave's quota filled between the colours. */
static int fused_trip_1d(gfo_state *s, const iter_state *it, const leader_t *ld, uint64_t off, int64_t sh,
                         uint64_t wave_first, uint64_t wave_quota, uint64_t *wave_done, double *x) {
    const pidx *pi = &s->pi;
    const uint64_t z = (uint64_t)(sh < 0 ? -sh : sh);
    uint64_t base = ld->ra0 + off;
    if (base >= ld->cnt) base -= ld->cnt;
    uint32_t node[64], pnode[64]; int out[64], term_ok[64], touched[64] = {0}; uint64_t grp[64];
    double xo[64], xp[64], td[64], acc[64] = {0.0};
    for (int l = 0; l < 64; l++) {
        const uint64_t sa = ld->first + base + (uint64_t)l, sb = (uint64_t)((int64_t)sa + sh);
        node[l] = pi->rec[sa].node; pnode[l] = pi->rec[sb].node;
        out[l] = l + sh < 0 || l + sh > 63;
        xo[l] = node[l] != GFO_NO_NODE ? x[node[l]] : 0.0;
        xp[l] = out[l] && pnode[l] != GFO_NO_NODE ? x[pnode[l]] : 0.0;
        td[l] = fabs((double)pi->rec[sa].pos - (double)pi->rec[sb].pos);               /* sgd.rs:513 */
        term_ok[l] = td[l] != 0.0 && node[l] != GFO_NO_NODE && pnode[l] != GFO_NO_NODE; /* :514, :525-538 */
        grp[l] = ((off + (uint64_t)l) / z) & 1;
    }
    int second = 1;
    for (uint64_t colour = 0; colour < 2; colour++) {
        int valid[64]; double r[64];
        uint64_t nvalid = 0, rank = 0;
        for (int l = 0; l < 64; l++) { valid[l] = term_ok[l] && grp[l] == colour; nvalid += (uint64_t)valid[l]; }
        const uint64_t remaining = wave_quota - *wave_done;
        for (int l = 0; l < 64; l++) {
            s->att[wave_first + l]++;
            if (valid[l] && rank++ >= remaining) valid[l] = 0;
        }
        *wave_done += nvalid < remaining ? nvalid : remaining;
        for (int l = 0; l < 64; l++) {                                 /* every term of a colour reads the same snapshot */
            r[l] = 0.0;
            if (!valid[l]) continue;
            const double xj = out[l] ? xp[l] : xo[l + sh];
            const double mu = fmin(it->eta * (1.0 / td[l]), 1.0);                      /* :518-520 */
            double dx = xo[l] - xj;                                                    /* :543 */
            if (dx == 0.0) dx = 1e-9;                                                  /* :546-548 */
            const double mag = fabs(dx);                                               /* :551 */
            const double delta = mu * (mag - td[l]) / 2.0;                             /* :552 */
            r[l] = (delta / mag) * dx;                                                 /* :570-571 */
            const uint64_t tg = wave_first + (uint64_t)l;
            s->done[tg]++;                                                             /* :579 */
            if (s->trace && s->ntr[tg] < s->trace_per_stream) {
                gfo_term *tr = &s->trace[tg * s->trace_per_stream + s->ntr[tg]++];
                tr->i = node[l]; tr->j = pnode[l]; tr->d_ij = td[l];
            }
        }
        for (int l = 0; l < 64; l++) {
            const int src = l - (int)sh;
            const int recv = src >= 0 && src <= 63 && valid[src];
            if (valid[l]) { xo[l] = xo[l] - r[l]; acc[l] = touched[l] ? acc[l] - r[l] : -r[l]; touched[l] = 1; }   /* :575 */
            if (recv) { xo[l] = xo[l] + r[src]; acc[l] = touched[l] ? acc[l] + r[src] : r[src]; touched[l] = 1; }  /* :576 */
            if (valid[l] && out[l]) x[pnode[l]] = x[pnode[l]] + r[l];
        }
        if (colour == 0 && *wave_done >= wave_quota) { second = 0; break; }
    }
    for (int l = 0; l < 64; l++) if (touched[l]) x[node[l]] = x[node[l]] + acc[l];
    return second;
}

/* Product: sgd_kernels_1d.hip twin_trip — both partners of a leader whose two runs are line-aligned long jumps, in one
 * trip, as the wave executes it: every lane loads the positions of its step a and of its two partners b, c once; term
 * (a, c) computes on what term (a, b) left in a's register; a's node receives ONE add, -(r + r'), b and c one each.
 * `off`: the trip starts this many steps after the run's first step.  Returns 0 when the quota filled between the terms. */
static int twin_trip_1d(gfo_state *s, const iter_state *it, const leader_t *ld, uint64_t off,
                        uint64_t wave_first, uint64_t wave_quota, uint64_t *wave_done, double *x) {
    const pidx *pi = &s->pi;
    uint32_t na[64], np[2][64]; double xa[64], xp[2][64], pa[64], pp[2][64], acc[64] = {0.0}; int touched[64] = {0};
    for (uint64_t l = 0; l < 64; l++) {
        const uint64_t sa = ld->first + ld->ra0 + off + l;
        const uint64_t sb = ld->first + ld->rb0 + off + ((l + (uint64_t)ld->rot) & 63);
        const uint64_t sc = ld->first + ld->rb1 + off + ((l + (uint64_t)ld->rot1) & 63);
        na[l] = pi->rec[sa].node; np[0][l] = pi->rec[sb].node; np[1][l] = pi->rec[sc].node;
        pa[l] = (double)pi->rec[sa].pos; pp[0][l] = (double)pi->rec[sb].pos; pp[1][l] = (double)pi->rec[sc].pos;
        xa[l] = na[l] != GFO_NO_NODE ? x[na[l]] : 0.0;
        xp[0][l] = np[0][l] != GFO_NO_NODE ? x[np[0][l]] : 0.0;
        xp[1][l] = np[1][l] != GFO_NO_NODE ? x[np[1][l]] : 0.0;
    }
    int second = 1;
    for (int q = 0; q < 2; q++) {
        int valid[64]; double td[64];
        uint64_t nvalid = 0, rank = 0;
        for (int l = 0; l < 64; l++) {
            td[l] = fabs(pa[l] - pp[q][l]);                                            /* sgd.rs:513 */
            valid[l] = td[l] != 0.0 && na[l] != GFO_NO_NODE && np[q][l] != GFO_NO_NODE; /* :514, :525-538 */
            nvalid += (uint64_t)valid[l];
        }
        const uint64_t remaining = wave_quota - *wave_done;
        for (int l = 0; l < 64; l++) {
            s->att[wave_first + l]++;
            if (valid[l] && rank++ >= remaining) valid[l] = 0;
        }
        *wave_done += nvalid < remaining ? nvalid : remaining;
        for (int l = 0; l < 64; l++) {
            if (!valid[l]) continue;
            const double mu = fmin(it->eta * (1.0 / td[l]), 1.0);                      /* :518-520 */
            double dx = xa[l] - xp[q][l];                                              /* :543 */
            if (dx == 0.0) dx = 1e-9;                                                  /* :546-548 */
            const double mag = fabs(dx);                                               /* :551 */
            const double delta = mu * (mag - td[l]) / 2.0;                             /* :552 */
            const double r = (delta / mag) * dx;                                       /* :570-571 */
            const uint64_t tg = wave_first + (uint64_t)l;
            s->done[tg]++;                                                             /* :579 */
            if (s->trace && s->ntr[tg] < s->trace_per_stream) {
                gfo_term *tr = &s->trace[tg * s->trace_per_stream + s->ntr[tg]++];
                tr->i = na[l]; tr->j = np[q][l]; tr->d_ij = td[l];
            }
            xa[l] = xa[l] - r;                                                         /* :575 */
            acc[l] = touched[l] ? acc[l] - r : -r; touched[l] = 1;
            x[np[q][l]] = x[np[q][l]] + r;                                             /* :576 */
        }
        if (q == 0 && *wave_done >= wave_quota) { second = 0; break; }
    }
    for (int l = 0; l < 64; l++) if (touched[l]) x[na[l]] = x[na[l]] + acc[l];
    return second;
}

/* Product: sgd_kernels_nd_team.hip twin_trip_nd — the layout form of the twin trip: step a takes the end its run's a-flip
 * selects in both terms, b and c the ends their own flips select; all coordinates are loaded before any add; the second term
 * computes on what the first left of a's coordinates; a's end receives ONE add, b's and c's one each.  Coordinates in Layout
 * order [node][end][dim].  Returns 0 when the quota filled between the terms. */
static int twin_trip_nd(gfo_state *s, const iter_state *it, const leader_t *ld, uint64_t off, int fa, int fb, int fc,
                        uint64_t wave_first, uint64_t wave_quota, uint64_t *wave_done, double *c) {
    const pidx *pi = &s->pi;
    const uint64_t D = s->D;
    int ok[2][64], touched[64] = {0}; double td[2][64]; uint64_t ia[64], ip[2][64];
    double ca[64][GFO_MAX_DIMS], cp[2][64][GFO_MAX_DIMS], acc[64][GFO_MAX_DIMS];
    for (uint64_t l = 0; l < 64; l++) {
        const uint64_t sa = ld->first + ld->ra0 + off + l;
        const uint64_t sp[2] = { ld->first + ld->rb0 + off + ((l + (uint64_t)ld->rot) & 63), ld->first + ld->rb1 + off + ((l + (uint64_t)ld->rot1) & 63) };
        for (int q = 0; q < 2; q++) {
            uint64_t i = 0, j = 0;
            ok[q][l] = nd_prepare(pi, sa, sp[q], fa, q ? fc : fb, &td[q][l], &i, &j);
            if (ok[q][l]) {
                ia[l] = i; ip[q][l] = j;
                for (uint64_t d = 0; d < D; d++) { ca[l][d] = c[i * D + d]; cp[q][l][d] = c[j * D + d]; }
            }
        }
        for (uint64_t d = 0; d < D; d++) acc[l][d] = 0.0;
    }
    int second = 1;
    for (int q = 0; q < 2; q++) {
        int valid[64]; uint64_t nvalid = 0, rank = 0;
        for (int l = 0; l < 64; l++) { valid[l] = ok[q][l]; nvalid += (uint64_t)valid[l]; }
        const uint64_t remaining = wave_quota - *wave_done;
        for (int l = 0; l < 64; l++) {
            s->att[wave_first + l]++;
            if (valid[l] && rank++ >= remaining) valid[l] = 0;
        }
        *wave_done += nvalid < remaining ? nvalid : remaining;
        for (int l = 0; l < 64; l++) {
            if (!valid[l]) continue;
            const double mu = fmin(it->eta * (1.0 / td[q][l]), 1.0);                   /* sgd.rs:1085-1086 */
            double deltas[GFO_MAX_DIMS], mag_sq = 0.0;
            for (uint64_t d = 0; d < D; d++) { deltas[d] = ca[l][d] - cp[q][l][d]; mag_sq += deltas[d] * deltas[d]; }   /* :1108-1113 */
            if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }                   /* :1116-1119 */
            const double mag = sqrt(mag_sq);                                           /* :1121 */
            const double r = (mu * (mag - td[q][l]) / 2.0) / mag;                      /* :1125, :1142 */
            const int same = ia[l] == ip[q][l];
            for (uint64_t d = 0; d < D; d++) {
                const double r_d = r * deltas[d];
                if (!same) { ca[l][d] = ca[l][d] - r_d; acc[l][d] = touched[l] ? acc[l][d] - r_d : -r_d; }   /* :1143-1146 */
                c[ip[q][l] * D + d] = c[ip[q][l] * D + d] + r_d;                       /* :1147-1148 */
            }
            if (!same) touched[l] = 1;
            const uint64_t tg = wave_first + (uint64_t)l;
            s->done[tg]++;                                                             /* :1151 */
            if (s->trace && s->ntr[tg] < s->trace_per_stream) {
                gfo_term *tr = &s->trace[tg * s->trace_per_stream + s->ntr[tg]++];
                tr->i = (uint32_t)ia[l]; tr->j = (uint32_t)ip[q][l]; tr->d_ij = td[q][l];
            }
        }
        if (q == 0 && *wave_done >= wave_quota) { second = 0; break; }
    }
    for (int l = 0; l < 64; l++)
        if (touched[l]) for (uint64_t d = 0; d < D; d++) c[ia[l] * D + d] = c[ia[l] * D + d] + acc[l][d];
    return second;
}

/* Product: sgd_kernels_nd_team.hip fused_trip_nd — the layout form of the fused short-jump trip (D >= 2): a lane's step is
 * the a-side of its own term in one colour and the b-side of its neighbour's in the other; it keeps the coordinates of its
 * a-end and (when the run's two flips differ) of its b-end in registers, colour 1 computes on what colour 0 left there, and
 * each of a lane's OWN ends receives ONE add for both colours, c + (-r + r'), not (c - r) + r'.  Partners beyond the trip
 * are added per colour.  Coordinates in Layout order [node][end][dim].  Returns 0 when the quota filled between the colours. */
static int fused_trip_nd(gfo_state *s, const iter_state *it, const leader_t *ld, uint64_t off, int64_t sh, int fa, int fb,
                         uint64_t wave_first, uint64_t wave_quota, uint64_t *wave_done, double *c) {
    const pidx *pi = &s->pi;
    const uint64_t D = s->D;
    const uint64_t z = (uint64_t)(sh < 0 ? -sh : sh);
    uint64_t base = ld->ra0 + off;
    if (base >= ld->cnt) base -= ld->cnt;
    int out[64], term_ok[64], has_node[64], t_a[64] = {0}, t_b[64] = {0}; uint64_t grp[64], ia[64], ibo[64], ij[64];
    double td[64], ca[64][GFO_MAX_DIMS], cb[64][GFO_MAX_DIMS], cp[64][GFO_MAX_DIMS], acc_a[64][GFO_MAX_DIMS], acc_b[64][GFO_MAX_DIMS];
    for (int l = 0; l < 64; l++) {
        const uint64_t sa = ld->first + base + (uint64_t)l, sb = (uint64_t)((int64_t)sa + sh);
        const uint32_t node = pi->rec[sa].node, pnode = pi->rec[sb].node;
        const int rev_own = (int)(pi->rec[sa].path_rev >> 31), rev_p = (int)(pi->rec[sb].path_rev >> 31);
        out[l] = l + sh < 0 || l + sh > 63;
        grp[l] = ((off + (uint64_t)l) / z) & 1;
        has_node[l] = node != GFO_NO_NODE;
        ia[l] = (uint64_t)(has_node[l] ? node : 0) * 2 + (uint64_t)(fa ? !rev_own : rev_own);      /* my a-end, my b-end */
        ibo[l] = (uint64_t)(has_node[l] ? node : 0) * 2 + (uint64_t)(fb ? !rev_own : rev_own);
        ij[l] = (uint64_t)(pnode != GFO_NO_NODE ? pnode : 0) * 2 + (uint64_t)(fb ? !rev_p : rev_p);  /* my partner's b-end */
        uint64_t i2 = 0, j2 = 0;
        term_ok[l] = nd_prepare(pi, sa, sb, fa, fb, &td[l], &i2, &j2);                 /* sgd.rs:1047-1103 */
        for (uint64_t d = 0; d < D; d++) {
            ca[l][d] = has_node[l] ? c[ia[l] * D + d] : 0.0;
            cb[l][d] = has_node[l] && fa != fb ? c[ibo[l] * D + d] : 0.0;
            cp[l][d] = out[l] && pnode != GFO_NO_NODE ? c[ij[l] * D + d] : 0.0;
            acc_a[l][d] = 0.0; acc_b[l][d] = 0.0;
        }
    }
    int second = 1;
    for (uint64_t colour = 0; colour < 2; colour++) {
        int valid[64]; double r[64][GFO_MAX_DIMS];
        uint64_t nvalid = 0, rank = 0;
        for (int l = 0; l < 64; l++) { valid[l] = term_ok[l] && grp[l] == colour; nvalid += (uint64_t)valid[l]; }
        const uint64_t remaining = wave_quota - *wave_done;
        for (int l = 0; l < 64; l++) {
            s->att[wave_first + l]++;
            if (valid[l] && rank++ >= remaining) valid[l] = 0;
        }
        *wave_done += nvalid < remaining ? nvalid : remaining;
        for (int l = 0; l < 64; l++) {                                 /* every term of a colour reads the same snapshot */
            for (uint64_t d = 0; d < D; d++) r[l][d] = 0.0;
            if (!valid[l]) continue;
            const double *cj = out[l] ? cp[l] : (fa != fb ? cb[l + sh] : ca[l + sh]);
            const double mu = fmin(it->eta * (1.0 / td[l]), 1.0);                      /* :1085-1086 */
            double deltas[GFO_MAX_DIMS], mag_sq = 0.0;
            for (uint64_t d = 0; d < D; d++) { deltas[d] = ca[l][d] - cj[d]; mag_sq += deltas[d] * deltas[d]; }   /* :1108-1113 */
            if (mag_sq == 0.0) { deltas[0] = 1e-9; mag_sq = 1e-18; }                   /* :1116-1119 */
            const double mag = sqrt(mag_sq);                                           /* :1121 */
            const double rr = (mu * (mag - td[l]) / 2.0) / mag;                        /* :1125, :1142 */
            for (uint64_t d = 0; d < D; d++) r[l][d] = rr * deltas[d];
            const uint64_t tg = wave_first + (uint64_t)l;
            s->done[tg]++;                                                             /* :1151 */
            if (s->trace && s->ntr[tg] < s->trace_per_stream) {
                gfo_term *tr = &s->trace[tg * s->trace_per_stream + s->ntr[tg]++];
                tr->i = (uint32_t)ia[l]; tr->j = (uint32_t)ij[l]; tr->d_ij = td[l];
            }
        }
        for (int l = 0; l < 64; l++) {
            const int src = l - (int)sh;
            const int recv = src >= 0 && src <= 63 && valid[src];
            if (valid[l] && ia[l] != ij[l]) {                                          /* :1143-1146 */
                for (uint64_t d = 0; d < D; d++) { ca[l][d] = ca[l][d] - r[l][d]; acc_a[l][d] = t_a[l] ? acc_a[l][d] - r[l][d] : -r[l][d]; }
                t_a[l] = 1;
            }
            if (recv) {                                                                /* :1147-1148 */
                if (fa != fb) {
                    for (uint64_t d = 0; d < D; d++) { cb[l][d] = cb[l][d] + r[src][d]; acc_b[l][d] = t_b[l] ? acc_b[l][d] + r[src][d] : r[src][d]; }
                    t_b[l] = 1;
                } else {
                    for (uint64_t d = 0; d < D; d++) { ca[l][d] = ca[l][d] + r[src][d]; acc_a[l][d] = t_a[l] ? acc_a[l][d] + r[src][d] : r[src][d]; }
                    t_a[l] = 1;
                }
            }
            if (valid[l] && out[l]) for (uint64_t d = 0; d < D; d++) c[ij[l] * D + d] = c[ij[l] * D + d] + r[l][d];
        }
        if (colour == 0 && *wave_done >= wave_quota) { second = 0; break; }
    }
    for (int l = 0; l < 64; l++) {
        if (t_a[l]) for (uint64_t d = 0; d < D; d++) c[ia[l] * D + d] = c[ia[l] * D + d] + acc_a[l][d];
        if (t_b[l]) for (uint64_t d = 0; d < D; d++) c[ibo[l] * D + d] = c[ibo[l] * D + d] + acc_b[l][d];
    }
    return second;
}

/* Team semantics of the product (sgd1d_team_kernel): per wave of 64 streams, a PASS samples one
 * leader per stream; B TRIPS then expand the 64 leaders as 64/B runs of B lanes (trip t, run q
 * uses leader t*(64/B)+q).  Wave-level quota with a rank cut-off.  1D: the trips of a pass left over when
 * the quota fills serve the next iteration (dropped only when the cooling phase, which the sampler depends
 * on, changes); nD (sgdnd_team_kernel): leftover leaders are dropped. */
static int run_iteration_bundled(gfo_state *s, uint64_t k, double *x) {
    const uint64_t T = s->T, B = s->bundle, RUNS = 64 / B;
    const pidx *pi = &s->pi;
    iter_state it; iter_consts(&s->p, s->etas, k, &it);
    const uint64_t base = s->quota_total / T, rem = s->quota_total % T;
    for (uint64_t w = 0; w < T / 64; w++) {
        const uint64_t wave_first = w * 64;
        uint64_t wave_quota = base * 64;
        if (wave_first < rem) wave_quota += (rem - wave_first) < 64 ? (rem - wave_first) : 64;
        const int carry = 1;        /* (round 2's layout kernel dropped what was left of a pass at the end of an iteration: the layout
                                     * kernels now work in chunks and keep their passes exactly as the 1D ones do) */
        /* 1D: the wave works through its quota in chunks of GFO_TEAM_CHUNK updates, each with its own rank cut-off and pass
         * budget (product: sgd_device.h TEAM_CHUNK; its fused launch draws such chunks from a pool) */
        const uint64_t wave_quota_all = wave_quota;
        const uint64_t CH = s->chunk ? s->chunk : (s->D ? GFO_ND_TEAM_CHUNK : GFO_TEAM_CHUNK);
        for (uint64_t chunk0 = 0; chunk0 < wave_quota_all; chunk0 += (carry ? CH : wave_quota_all)) {
        if (carry) wave_quota = wave_quota_all - chunk0 < CH ? wave_quota_all - chunk0 : CH;
        const uint64_t max_passes = s->attempt_factor * (wave_quota / (64 * B) + 1) + 16;
        uint64_t wave_done = 0, passes = 0;
        if (!s->lead) {
            s->lead = (leader_t *)calloc(T, sizeof(leader_t));
            s->lead_left = (uint8_t *)calloc(T / 64, 1); s->lead_cool = (uint8_t *)calloc(T / 64, 1);
            s->lead_colour = (uint8_t *)calloc(T / 64, 1); s->lead_seg = (uint8_t *)calloc(T / 64, 1);
            s->lead_p = (uint8_t *)calloc(T / 64, 1);
            s->lead_flips = (uint8_t *)calloc(T, 1);
        }
        leader_t *L = s->lead + wave_first;
        int lead_fa[64] = {0}, lead_fb[64] = {0}, lead_fc[64] = {0};
        for (int l = 0; l < 64; l++) {                                  /* (a pass outlives the chunk it was sampled in) */
            const uint8_t f = s->lead_flips[wave_first + l];
            lead_fa[l] = f & 1; lead_fb[l] = (f >> 1) & 1; lead_fc[l] = (f >> 2) & 1;
        }
        while (wave_done < wave_quota && passes < max_passes) {
            if (!carry || s->lead_left[w] == 0 || s->lead_cool[w] != (uint8_t)it.cooling) {
                passes++;
                for (int l = 0; l < 64; l++) {
                    L[l] = sample_leader(s, &it, s->rng + 4 * (wave_first + l));
                    if (s->D) {                                        /* nD: the run's two end flips, drawn by the leader's stream */
                        lead_fa[l] = (int)flip(s->rng + 4 * (wave_first + l));
                        lead_fb[l] = (int)flip(s->rng + 4 * (wave_first + l));
                        if (s->partners == 2) lead_fc[l] = (int)flip(s->rng + 4 * (wave_first + l));   /* the second partner's b */
                        s->lead_flips[wave_first + l] = (uint8_t)(lead_fa[l] | (lead_fb[l] << 1) | (lead_fc[l] << 2));
                    }
                }
                s->lead_left[w] = (uint8_t)B; s->lead_cool[w] = (uint8_t)it.cooling; s->lead_colour[w] = 0; s->lead_seg[w] = 0;
                s->lead_p[w] = 0;
            }
            /* trips: slot t of the pass, colour 0, then — when some run of the slot has a jump shorter than the run
             * (sgd_device.h two_colour) — the same slot with colour 1: the run's terms chain through shared
             * nodes, each colour is node-disjoint, together they are every term of the run */
            while (s->lead_left[w] > 0 && wave_done < wave_quota) {
                const uint64_t t = B - s->lead_left[w];
                const uint64_t colour = s->lead_colour[w], seg = s->lead_seg[w], pp = s->lead_p[w];
                leader_t V[16];                                         /* the slot's leaders as partner pp sees them */
                for (uint64_t qq = 0; qq < RUNS; qq++) V[qq] = partner_view(&L[t * RUNS + qq], pp);
                int two = 0;
                for (uint64_t qq = 0; qq < RUNS && !s->one_colour; qq++) {
                    const leader_t *ld = &V[qq];
                    const int64_t shift = (int64_t)ld->rb0 - (int64_t)ld->ra0;
                    if (ld->ok && !ld->aligned && ld->cnt >= 2 * B && shift < (int64_t)B && shift > -(int64_t)B) two = 1;
                }
                /* trips of this slot's run (B = 64: one leader per trip); the same for both partners, whose trips alternate */
                const uint64_t ktrips = (RUNS == 1 && (L[t].ok || L[t].ok1) && L[t].cnt >= 2 * B) ? run_trips(s, L[t].cnt) : 1;
                const int more_partners = pp == 0 && s->partners == 2;
                /* after the trip(s) of (seg, pp): the other partner's trip of this seg, else the run's next trip, else the next slot */
#define GFO_ADVANCE(skip_partner) do { \
                    s->lead_colour[w] = 0; \
                    if (more_partners && !(skip_partner)) s->lead_p[w] = 1; \
                    else if (seg + 1 < ktrips) { s->lead_seg[w] = (uint8_t)(seg + 1); s->lead_p[w] = 0; } \
                    else { s->lead_seg[w] = 0; s->lead_p[w] = 0; s->lead_left[w]--; } } while (0)
                /* (not when a block of one partner run lies within two trips of the other's: the wave would read, as one
                 * partner's positions, what it has only just added as the other's) */
                const int64_t pgap = (int64_t)L[t].rb0 - (int64_t)L[t].rb1, plim = 192;
                if (s->D != 1 && RUNS == 1 && more_partners && !s->no_twin_trip && L[t].ok && L[t].aligned && L[t].ok1 && L[t].aligned1 &&
                    (pgap >= plim || pgap <= -plim)) {
                    /* both partners in one trip (product: twin_trip / twin_trip_nd) */
                    const int second = s->D ? twin_trip_nd(s, &it, &L[t], seg * B, lead_fa[t], lead_fb[t], lead_fc[t], wave_first, wave_quota, &wave_done, x)
                                            : twin_trip_1d(s, &it, &L[t], seg * B, wave_first, wave_quota, &wave_done, x);
                    if (!second) s->lead_p[w] = 1;                      /* quota filled between the partners */
                    else GFO_ADVANCE(1);
                    continue;
                }
                if ((s->D == 0 || s->D >= 2) && RUNS == 1 && colour == 0 && two && !s->no_fused_trip) {
                    const uint64_t off0 = run_offset(s, V[0].cnt, ktrips, V[0].ra0, V[0].rb0, seg);
                    const int64_t ms = merged_trip_shift(&V[0], off0);
                    if (ms != 0) {                                      /* both colours in one trip (product: fused_trip / fused_trip_nd) */
                        const int second = s->D ? fused_trip_nd(s, &it, &V[0], off0, ms, lead_fa[t], pp ? lead_fc[t] : lead_fb[t],
                                                                wave_first, wave_quota, &wave_done, x)
                                                : fused_trip_1d(s, &it, &V[0], off0, ms, wave_first, wave_quota, &wave_done, x);
                        if (!second)
                            s->lead_colour[w] = 1;                      /* quota filled between the colours */
                        else GFO_ADVANCE(0);
                        continue;
                    }
                }
                /* what comes next: this trip's second colour, else as above */
                if (colour == 0 && two) s->lead_colour[w] = 1;
                else GFO_ADVANCE(0);
#undef GFO_ADVANCE
                int valid[64], flips_a[64], flips_b[64]; uint64_t sa[64], sb[64];
                uint64_t nvalid = 0;
                for (uint64_t qq = 0; qq < RUNS; qq++) {
                    const leader_t *ld = &V[qq];
                    const int64_t shift = (int64_t)ld->rb0 - (int64_t)ld->ra0;
                    const uint64_t zabs = (uint64_t)(shift < 0 ? -shift : shift);
                    const uint64_t off = run_offset(s, ld->cnt, ktrips, ld->ra0, ld->rb0, seg);
                    for (uint64_t sub = 0; sub < B; sub++) {
                        const uint64_t l = qq * B + sub;
                        const uint64_t pl = off + sub;                  /* place in the run */
                        valid[l] = 0;
                        if (!ld->ok) continue;
                        uint64_t ra = ld->ra0, rb = ld->rb0;
                        if (ld->aligned) {                              /* both runs are aligned blocks inside the path */
                            if (colour) continue;
                            ra = ld->ra0 + pl; rb = ld->rb0 + off + ((sub + (uint64_t)ld->rot) % B);
                        } else if (pl != 0 || colour) {
                            if (ld->cnt < 2 * B) continue;
                            if (zabs < B) { if (((pl / zabs) & 1) != colour) continue; }
                            else if (colour) continue;
                            ra = ld->ra0 + pl;
                            if (ra >= ld->cnt) ra -= ld->cnt;
                            if (ra >= ld->cnt) ra -= ld->cnt;
                            int64_t tt = (int64_t)ra + shift;
                            if (tt < 0 || tt > (int64_t)ld->cnt - 1) {          /* mirror the jump (|jump| >= B only) */
                                if (zabs < B) continue;
                                tt = (int64_t)ra - shift;
                                if (tt < 0 || tt > (int64_t)ld->cnt - 1) continue;
                            }
                            rb = (uint64_t)tt;
                        }
                        sa[l] = ld->first + ra; sb[l] = ld->first + rb;
                        if (s->D) {
                            /* nD: every lane of a run uses the end flips its leader drew (sgd.rs:1062,1071);
                             * probe the term without applying it (:1080) */
                            flips_a[l] = lead_fa[t * RUNS + qq];
                            flips_b[l] = pp ? lead_fc[t * RUNS + qq] : lead_fb[t * RUNS + qq];
                            if (!nd_term_ok(pi, sa[l], sb[l], flips_a[l], flips_b[l])) continue;
                        } else {
                            double td = fabs((double)pi->rec[sa[l]].pos - (double)pi->rec[sb[l]].pos);
                            if (td == 0.0 || pi->rec[sa[l]].node == GFO_NO_NODE || pi->rec[sb[l]].node == GFO_NO_NODE) continue;
                        }
                        valid[l] = 1; nvalid++;
                    }
                }
                const uint64_t remaining = wave_quota - wave_done;
                uint64_t rank = 0;
                for (int l = 0; l < 64; l++) {
                    s->att[wave_first + l]++;
                    if (!valid[l]) continue;
                    if (rank++ >= remaining) continue;
                    gfo_term tr;
                    if (s->D) (void)term_nd_flips(pi, &it, flips_a[l], flips_b[l], sa[l], sb[l], x, s->D, &tr);
                    else (void)term_1d(pi, 1, &it, sa[l], sb[l], x, 0, NULL, &tr);
                    uint64_t tg = wave_first + l;
                    s->done[tg]++;
                    if (s->trace && s->ntr[tg] < s->trace_per_stream) s->trace[tg * s->trace_per_stream + s->ntr[tg]++] = tr;
                }
                wave_done += nvalid < remaining ? nvalid : remaining;
            }
        }
        }   /* chunks */
    }
    return 0;
}

/* One batch: iteration k uses etas[k]; streams advance round-robin, one attempt each. */
int gfo_state_run_iteration(gfo_state *s, uint64_t k, double *x) {
    if (!s || k > s->p.iter_max) return -1;
    const uint64_t T = s->T, D = s->D;
    double t0 = now_s();
    iter_state it; iter_consts(&s->p, s->etas, k, &it);
    uint64_t base = s->quota_total / T, rem = s->quota_total % T;
    memset(s->done, 0, T * 8); memset(s->att, 0, T * 8);
    if (s->bundle > 1) {
        run_iteration_bundled(s, k, x);
        for (uint64_t t = 0; t < T; t++) { s->total_upd += s->done[t]; s->total_att += s->att[t]; }
        s->iterations++;
        s->seconds += now_s() - t0;
        return 0;
    }
    for (;;) {
        int active = 0;
        for (uint64_t t = 0; t < T; t++) {
            uint64_t quota = base + (t < rem ? 1 : 0);
            if (s->done[t] >= quota || s->att[t] >= s->attempt_factor * quota + 1024) continue;
            active = 1;
            s->att[t]++;
            uint64_t sa, sb;
            if (!sample_pair(&s->pi, 1, &s->z, &it, s->rng + 4 * t, &sa, &sb)) continue;
            gfo_term tr;
            int ok = D ? term_nd(&s->pi, 1, &it, s->rng + 4 * t, sa, sb, x, D, 0, NULL, &tr)
                       : term_1d(&s->pi, 1, &it, sa, sb, x, 0, NULL, &tr);
            if (!ok) continue;
            s->done[t]++;
            if (s->trace && s->ntr[t] < s->trace_per_stream) s->trace[t * s->trace_per_stream + s->ntr[t]++] = tr;
        }
        if (!active) break;
    }
    for (uint64_t t = 0; t < T; t++) { s->total_upd += s->done[t]; s->total_att += s->att[t]; }
    s->iterations++;
    s->seconds += now_s() - t0;
    return 0;
}

void gfo_state_stats(const gfo_state *s, gfo_stats *st) {
    st->term_updates = s->total_upd; st->attempts = s->total_att;
    st->iterations = s->iterations; st->seconds = s->seconds;
}

static int sgd_det(const gfo_graph *g, const gfo_params *p, const double *etas_in, const double *zetas_in,
                   uint64_t T, uint64_t attempt_factor, double *x, uint64_t D /*0 = 1D*/,
                   gfo_term *trace, uint64_t trace_per_stream, gfo_stats *st) {
    if (st) memset(st, 0, sizeof *st);
    gfo_state *s;
    int rc = gfo_state_create(g, p, etas_in, zetas_in, D, T, 0, 0, attempt_factor, trace, trace_per_stream, &s);
    if (rc) return rc;
    for (uint64_t k = 0; k <= p->iter_max; k++) gfo_state_run_iteration(s, k, x);
    if (st) gfo_state_stats(s, st);
    gfo_state_destroy(s);
    return 0;
}

int gfo_sgd_1d(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
               uint64_t n_streams, uint64_t attempt_factor, double *x,
               gfo_term *trace, uint64_t trace_per_stream, gfo_stats *st) {
    return sgd_det(g, p, etas, zetas, n_streams, attempt_factor, x, 0, trace, trace_per_stream, st);
}
int gfo_sgd_nd(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
               uint64_t n_streams, uint64_t attempt_factor, double *coords,
               gfo_term *trace, uint64_t trace_per_stream, gfo_stats *st) {
    if (p->dimensions == 0) return -1;
    return sgd_det(g, p, etas, zetas, n_streams, attempt_factor, coords, p->dimensions, trace, trace_per_stream, st);
}

/* ------------------------------------------------------------------------------------------
 * Reference-like threaded mode: Hogwild workers + 1 ms checker (sgd.rs:340-407, 413-601).
 * ---------------------------------------------------------------------------------------- */
typedef struct shared {
    const pidx *pi; int flat; const gfo_params *p; const double *etas; zipf_env z;
    double *x; uint64_t D;
    _Atomic uint64_t term_updates, iteration, eta_bits, theta_bits, delta_max, total_updates, total_attempts;
    _Atomic int cooling, work_todo;
    double deadline;
} shared;
typedef struct wargs { shared *sh; uint64_t tid; } wargs;

static void *checker_main(void *arg) {                                /* sgd.rs:366-407 */
    shared *sh = (shared *)arg;
    const gfo_params *p = sh->p;
    uint64_t first_cooling = sat_u64(floor(p->cooling_start * (double)p->iter_max));
    while (atomic_load_explicit(&sh->work_todo, memory_order_relaxed)) {
        uint64_t cur = atomic_load_explicit(&sh->term_updates, memory_order_relaxed);
        if (cur >= p->min_term_updates) {
            uint64_t new_iter = atomic_fetch_add_explicit(&sh->iteration, 1, memory_order_relaxed) + 1;
            if (new_iter > p->iter_max) {
                atomic_store_explicit(&sh->work_todo, 0, memory_order_relaxed);
            } else {
                atomic_store_explicit(&sh->eta_bits, f64_bits(sh->etas[new_iter]), memory_order_relaxed);
                if (new_iter > first_cooling) {
                    atomic_store_explicit(&sh->theta_bits, f64_bits(0.001), memory_order_relaxed);
                    atomic_store_explicit(&sh->cooling, 1, memory_order_relaxed);
                }
            }
            atomic_fetch_add_explicit(&sh->total_updates, cur, memory_order_relaxed);
            atomic_store_explicit(&sh->term_updates, 0, memory_order_relaxed);   /* :400 */
        }
        if (sh->deadline > 0.0 && now_s() > sh->deadline)
            atomic_store_explicit(&sh->work_todo, 0, memory_order_relaxed);       /* bounded sample */
        struct timespec ts = { 0, 1000000 }; nanosleep(&ts, NULL);               /* :403 */
    }
    return NULL;
}

static void *worker_main(void *arg) {                                 /* sgd.rs:429-590 */
    wargs *wa = (wargs *)arg; shared *sh = wa->sh;
    uint64_t rng[4]; gfo_xoshiro_seed(sh->p->seed + wa->tid, rng);
    uint64_t local = 0, attempts = 0;
    const int flat = sh->flat;
    while (atomic_load_explicit(&sh->work_todo, memory_order_relaxed)) {
        attempts++;
        iter_state it;
        it.cooling = atomic_load_explicit(&sh->cooling, memory_order_relaxed);
        it.theta = bits_f64(atomic_load_explicit(&sh->theta_bits, memory_order_relaxed));
        it.eta = bits_f64(atomic_load_explicit(&sh->eta_bits, memory_order_relaxed));
        uint64_t sa, sb; int ok;
        if (flat) {
            if (!sample_pair(sh->pi, 1, &sh->z, &it, rng, &sa, &sb)) continue;
            ok = sh->D ? term_nd(sh->pi, 1, &it, rng, sa, sb, sh->x, sh->D, 1, NULL, NULL)
                       : term_1d(sh->pi, 1, &it, sa, sb, sh->x, 1, NULL, NULL);
        } else {
            if (!sample_pair(sh->pi, 0, &sh->z, &it, rng, &sa, &sb)) continue;
            ok = sh->D ? term_nd(sh->pi, 0, &it, rng, sa, sb, sh->x, sh->D, 1, &sh->delta_max, NULL)
                       : term_1d(sh->pi, 0, &it, sa, sb, sh->x, 1, &sh->delta_max, NULL);
        }
        if (!ok) continue;
        if (++local >= 1000) {                                        /* :579-583 */
            atomic_fetch_add_explicit(&sh->term_updates, local, memory_order_relaxed);
            local = 0;
        }
    }
    if (local) atomic_fetch_add_explicit(&sh->term_updates, local, memory_order_relaxed);
    atomic_fetch_add_explicit(&sh->total_attempts, attempts, memory_order_relaxed);
    return NULL;
}

static int sgd_threads(const gfo_graph *g, const gfo_params *p, const double *etas_in, const double *zetas_in,
                       int flat, double max_seconds, double *x, uint64_t D, gfo_stats *st) {
    if (st) memset(st, 0, sizeof *st);
    if (g->n_nodes == 0 || !has_valid_paths(g)) return 1;
    if (D > GFO_MAX_DIMS) return -1;
    pidx pi;
    if (pidx_build(g, flat, &pi)) { pidx_free(&pi); return -2; }
    double *etas = NULL, *zetas = NULL;
    if (!etas_in) { etas = (double *)malloc((p->iter_max + 1) * 8); gfo_schedule(p, etas); etas_in = etas; }
    if (!zetas_in) { zetas = (double *)malloc(gfo_zeta_size(p) * 8); gfo_zetas(p, zetas); zetas_in = zetas; }
    shared sh; memset(&sh, 0, sizeof sh);
    sh.pi = &pi; sh.flat = flat; sh.p = p; sh.etas = etas_in; sh.x = x; sh.D = D;
    sh.z = (zipf_env){ zetas_in, gfo_zeta_size(p), p->space, p->space_max, p->space_quantization_step };
    atomic_store(&sh.eta_bits, f64_bits(etas_in[0]));
    atomic_store(&sh.theta_bits, f64_bits(p->theta));
    atomic_store(&sh.work_todo, 1);
    double t0 = now_s();
    sh.deadline = max_seconds > 0.0 ? t0 + max_seconds : 0.0;
    uint64_t T = p->nthreads;
    pthread_t chk; pthread_t *th = (pthread_t *)malloc((T ? T : 1) * sizeof(pthread_t));
    wargs *wa = (wargs *)malloc((T ? T : 1) * sizeof(wargs));
    pthread_create(&chk, NULL, checker_main, &sh);
    for (uint64_t t = 0; t < T; t++) { wa[t].sh = &sh; wa[t].tid = t; pthread_create(&th[t], NULL, worker_main, &wa[t]); }
    if (T == 0) atomic_store(&sh.work_todo, 0);      /* no workers: reference would spin forever; stop */
    for (uint64_t t = 0; t < T; t++) pthread_join(th[t], NULL);
    atomic_store(&sh.work_todo, 0);
    pthread_join(chk, NULL);
    double t1 = now_s();
    if (st) {
        st->term_updates = atomic_load(&sh.total_updates) + atomic_load(&sh.term_updates);
        st->attempts = atomic_load(&sh.total_attempts);
        st->iterations = atomic_load(&sh.iteration);
        st->seconds = t1 - t0;
    }
    free(th); free(wa); free(etas); free(zetas);
    pidx_free(&pi);
    return 0;
}
int gfo_sgd_1d_threads(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
                       int flat, double max_seconds, double *x, gfo_stats *st) {
    return sgd_threads(g, p, etas, zetas, flat, max_seconds, x, 0, st);
}
int gfo_sgd_nd_threads(const gfo_graph *g, const gfo_params *p, const double *etas, const double *zetas,
                       int flat, double max_seconds, double *coords, gfo_stats *st) {
    if (p->dimensions == 0) return -1;
    return sgd_threads(g, p, etas, zetas, flat, max_seconds, coords, p->dimensions, st);
}

/* ------------------------------------------------------------------------------------------
 * calculate_layout_stress — sgd.rs:1196-1283 (+ end of both nodes, seed 12345).
 * ---------------------------------------------------------------------------------------- */
static double stress_impl(const gfo_graph *g, uint64_t D, uint64_t stride, const double *c, uint64_t samples) {
    if (g->n_steps < 2) return 0.0;                                   /* :1220 */
    uint64_t *pos = (uint64_t *)malloc(g->n_steps * 8);
    uint32_t *pth = (uint32_t *)malloc(g->n_steps * 4);
    gfo_path_index(g, pos, pth, NULL, NULL);
    uint64_t rng[4]; gfo_xoshiro_seed(12345, rng);                    /* :1218 */
    double stress_sum = 0.0; uint64_t count = 0;
    for (uint64_t k = 0; k < samples; k++) {
        uint64_t step_a = uniform_usize(rng, g->n_steps);             /* :1230 */
        uint64_t p = pth[step_a];
        uint64_t first = g->path_first_step[p], cnt = g->path_first_step[p + 1] - first;
        if (cnt < 2) continue;
        uint64_t rank_a = step_a - first;
        uint64_t rank_b = uniform_usize(rng, cnt);                    /* :1239-1240 */
        if (rank_a == rank_b) continue;
        uint64_t sa = first + rank_a, sb = first + rank_b;
        double path_dist = fabs((double)pos[sa] - (double)pos[sb]);   /* :1252-1254 */
        if (path_dist == 0.0) continue;
        uint32_t ia = g->step_node[sa], ib = g->step_node[sb];
        if (ia == GFO_NO_NODE || ib == GFO_NO_NODE) continue;
        double sum_sq = 0.0;                                          /* layout.rs:126-133 */
        for (uint64_t d = 0; d < D; d++) {
            double delta = c[(uint64_t)ia * stride + d] - c[(uint64_t)ib * stride + d];
            sum_sq += delta * delta;
        }
        double err = sqrt(sum_sq) - path_dist;                        /* :1273 */
        stress_sum += (err * err) / (path_dist * path_dist);          /* :1274 */
        count++;
    }
    free(pos); free(pth);
    return count ? sqrt(stress_sum / (double)count) : 0.0;            /* :1278-1282 */
}
double gfo_layout_stress(const gfo_graph *g, uint64_t dims, const double *coords, uint64_t sample_count) {
    return stress_impl(g, dims, 2 * dims, coords, sample_count);
}
double gfo_stress_1d(const gfo_graph *g, const double *x, uint64_t sample_count) {
    return stress_impl(g, 1, 1, x, sample_count);
}
