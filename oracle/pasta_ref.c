/* CPU restatement of the Pasta MSM hot path -- TEST INFRASTRUCTURE ONLY (see pasta_ref.h).
 * PARITY UNPINNED BY THE REFERENCE (no MSM known-answer vector exists in /root/reference);
 * pinned instead against oracle/pasta_oracle.py (big-int definition) and SURVEY.md 8c anchors.
 *
 * Plain C11 + unsigned __int128, pthreads: a chunk-per-thread Pippenger (halo2's cpu_best_multiexp shape) and a
 * window-parallel one on a persistent thread pool (pasta-msm's shape; the timed cpu_baseline).
 */
#include "pasta_ref.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct {
    u64 m[4];   /* modulus */
    u64 ninv;   /* -m^-1 mod 2^64 */
    u64 r1[4];  /* R mod m  (Montgomery one) */
    u64 r2[4];  /* R^2 mod m */
} field_t;

/* Constants verified by big-int arithmetic; q == src/backend/r1cs_helper.rs:37-38 */
static const field_t FP = {
    {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0000000000000000ULL, 0x4000000000000000ULL},
    0x992d30ecffffffffULL,
    {0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL},
    {0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL}};
static const field_t FQ = {
    {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0000000000000000ULL, 0x4000000000000000ULL},
    0x8c46eb20ffffffffULL,
    {0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL},
    {0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL}};

static const field_t *field_of(int f) { return f == 0 ? &FP : &FQ; }
/* coordinate field / scalar field of a curve */
static const field_t *coord_field(int curve) { return curve == PASTA_PALLAS ? &FP : &FQ; }
static const field_t *scalar_field(int curve) { return curve == PASTA_PALLAS ? &FQ : &FP; }

typedef struct { u64 v[4]; } fe;

static inline int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) {
    return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
static inline int ge_mod(const u64 *a, const u64 *m) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > m[i]) return 1;
        if (a[i] < m[i]) return 0;
    }
    return 1;
}
static inline u64 sub4(u64 *r, const u64 *a, const u64 *b) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)t;
        borrow = (u64)(t >> 64) & 1;
    }
    return borrow;
}
static inline u64 add4(u64 *r, const u64 *a, const u64 *b) {
    u64 carry = 0;
    for (int i = 0; i < 4; ++i) {
        u128 t = (u128)a[i] + b[i] + carry;
        r[i] = (u64)t;
        carry = (u64)(t >> 64);
    }
    return carry;
}

static inline void fe_add(const field_t *F, fe *r, const fe *a, const fe *b) {
    u64 t[4];
    add4(t, a->v, b->v); /* both < m < 2^255: no carry out */
    if (ge_mod(t, F->m)) sub4(t, t, F->m);
    memcpy(r->v, t, 32);
}
static inline void fe_sub(const field_t *F, fe *r, const fe *a, const fe *b) {
    u64 t[4];
    if (sub4(t, a->v, b->v)) add4(t, t, F->m);
    memcpy(r->v, t, 32);
}
static inline void fe_neg(const field_t *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { memset(r, 0, 32); return; }
    sub4(r->v, F->m, a->v);
}
static inline void fe_dbl(const field_t *F, fe *r, const fe *a) { fe_add(F, r, a, a); }

/* Montgomery product a*b*R^-1 mod m (CIOS, 4 x 64-bit limbs). */
static inline void fe_mul(const field_t *F, fe *r, const fe *a, const fe *b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u64 carry = 0;
        for (int j = 0; j < 4; ++j) {
            u128 x = (u128)a->v[j] * b->v[i] + t[j] + carry;
            t[j] = (u64)x;
            carry = (u64)(x >> 64);
        }
        u128 s = (u128)t[4] + carry;
        t[4] = (u64)s;
        t[5] = (u64)(s >> 64);
        u64 mq = t[0] * F->ninv;
        u128 x = (u128)mq * F->m[0] + t[0];
        carry = (u64)(x >> 64);
        for (int j = 1; j < 4; ++j) {
            x = (u128)mq * F->m[j] + t[j] + carry;
            t[j - 1] = (u64)x;
            carry = (u64)(x >> 64);
        }
        s = (u128)t[4] + carry;
        t[3] = (u64)s;
        t[4] = t[5] + (u64)(s >> 64);
    }
    if (t[4] || ge_mod(t, F->m)) sub4(t, t, F->m);
    memcpy(r->v, t, 32);
}
static inline void fe_sqr(const field_t *F, fe *r, const fe *a) { fe_mul(F, r, a, a); }

static void fe_pow(const field_t *F, fe *r, const fe *a, const u64 *e) {
    fe acc;
    memcpy(acc.v, F->r1, 32);
    for (int i = 255; i >= 0; --i) {
        fe_sqr(F, &acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) fe_mul(F, &acc, &acc, a);
    }
    *r = acc;
}
static void fe_inv(const field_t *F, fe *r, const fe *a) { /* a^(m-2); inv(0) = 0 */
    u64 e[4], two[4] = {2, 0, 0, 0};
    sub4(e, F->m, two);
    fe_pow(F, r, a, e);
}
static void fe_to_mont(const field_t *F, fe *r, const fe *a) {
    fe r2;
    memcpy(r2.v, F->r2, 32);
    fe_mul(F, r, a, &r2);
}
static void fe_from_mont(const field_t *F, fe *r, const fe *a) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(F, r, a, &one);
}

/* ---------------------------------------------------------------- group ---- */
typedef struct { fe x, y; } aff;
typedef struct { fe x, y, z; } jac;

static inline int aff_is_inf(const aff *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline int jac_is_inf(const jac *p) { return fe_is_zero(&p->z); }
static inline void jac_set_inf(const field_t *F, jac *p) {
    memset(p, 0, sizeof *p);
    memcpy(p->y.v, F->r1, 32);
}

/* dbl-2009-l, a = 0 */
static void jac_dbl(const field_t *F, jac *r, const jac *p) {
    if (jac_is_inf(p) || fe_is_zero(&p->y)) { jac_set_inf(F, r); return; }
    fe a, b, c, d, e, f, t;
    fe_sqr(F, &a, &p->x);
    fe_sqr(F, &b, &p->y);
    fe_sqr(F, &c, &b);
    fe_add(F, &t, &p->x, &b);
    fe_sqr(F, &t, &t);
    fe_sub(F, &t, &t, &a);
    fe_sub(F, &t, &t, &c);
    fe_dbl(F, &d, &t);
    fe_dbl(F, &e, &a);
    fe_add(F, &e, &e, &a);
    fe_sqr(F, &f, &e);
    fe z3;
    fe_mul(F, &z3, &p->y, &p->z);
    fe_dbl(F, &z3, &z3);
    fe x3;
    fe_dbl(F, &t, &d);
    fe_sub(F, &x3, &f, &t);
    fe y3;
    fe_sub(F, &t, &d, &x3);
    fe_mul(F, &y3, &e, &t);
    fe_dbl(F, &c, &c);
    fe_dbl(F, &c, &c);
    fe_dbl(F, &c, &c);
    fe_sub(F, &y3, &y3, &c);
    r->x = x3; r->y = y3; r->z = z3;
}

/* general Jacobian addition, all special cases handled */
static void jac_add(const field_t *F, jac *r, const jac *p, const jac *q) {
    if (jac_is_inf(p)) { *r = *q; return; }
    if (jac_is_inf(q)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t;
    fe_sqr(F, &z1z1, &p->z);
    fe_sqr(F, &z2z2, &q->z);
    fe_mul(F, &u1, &p->x, &z2z2);
    fe_mul(F, &u2, &q->x, &z1z1);
    fe_mul(F, &s1, &p->y, &q->z);
    fe_mul(F, &s1, &s1, &z2z2);
    fe_mul(F, &s2, &q->y, &p->z);
    fe_mul(F, &s2, &s2, &z1z1);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { jac_dbl(F, r, p); return; }
        jac_set_inf(F, r);
        return;
    }
    fe_sub(F, &h, &u2, &u1);
    fe_sub(F, &rr, &s2, &s1);
    fe_sqr(F, &hh, &h);
    fe_mul(F, &hhh, &h, &hh);
    fe_mul(F, &v, &u1, &hh);
    fe x3, y3, z3;
    fe_sqr(F, &x3, &rr);
    fe_sub(F, &x3, &x3, &hhh);
    fe_dbl(F, &t, &v);
    fe_sub(F, &x3, &x3, &t);
    fe_sub(F, &t, &v, &x3);
    fe_mul(F, &y3, &rr, &t);
    fe_mul(F, &t, &s1, &hhh);
    fe_sub(F, &y3, &y3, &t);
    fe_mul(F, &z3, &p->z, &q->z);
    fe_mul(F, &z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_add_affine(const field_t *F, jac *r, const jac *p, const aff *q) {
    if (aff_is_inf(q)) { *r = *p; return; }
    if (jac_is_inf(p)) {
        r->x = q->x; r->y = q->y;
        memcpy(r->z.v, F->r1, 32);
        return;
    }
    fe z1z1, u2, s2, h, rr, hh, hhh, v, t;
    fe_sqr(F, &z1z1, &p->z);
    fe_mul(F, &u2, &q->x, &z1z1);
    fe_mul(F, &s2, &q->y, &p->z);
    fe_mul(F, &s2, &s2, &z1z1);
    if (fe_eq(&p->x, &u2)) {
        if (fe_eq(&p->y, &s2)) { jac_dbl(F, r, p); return; }
        jac_set_inf(F, r);
        return;
    }
    fe_sub(F, &h, &u2, &p->x);
    fe_sub(F, &rr, &s2, &p->y);
    fe_sqr(F, &hh, &h);
    fe_mul(F, &hhh, &h, &hh);
    fe_mul(F, &v, &p->x, &hh);
    fe x3, y3, z3;
    fe_sqr(F, &x3, &rr);
    fe_sub(F, &x3, &x3, &hhh);
    fe_dbl(F, &t, &v);
    fe_sub(F, &x3, &x3, &t);
    fe_sub(F, &t, &v, &x3);
    fe_mul(F, &y3, &rr, &t);
    fe_mul(F, &t, &p->y, &hhh);
    fe_sub(F, &y3, &y3, &t);
    fe_mul(F, &z3, &p->z, &h);
    r->x = x3; r->y = y3; r->z = z3;
}

static void jac_to_aff(const field_t *F, aff *r, const jac *p) {
    if (jac_is_inf(p)) { memset(r, 0, sizeof *r); return; }
    fe zi, zi2, zi3;
    fe_inv(F, &zi, &p->z);
    fe_sqr(F, &zi2, &zi);
    fe_mul(F, &zi3, &zi2, &zi);
    fe_mul(F, &r->x, &p->x, &zi2);
    fe_mul(F, &r->y, &p->y, &zi3);
}

/* scalars handed over the ABI -> canonical little-endian 256-bit integers */
static void scalar_canon(int curve, const u64 *s, int is_mont, u64 *out) {
    if (is_mont) {
        fe t;
        memcpy(t.v, s, 32);
        fe_from_mont(scalar_field(curve), &t, &t);
        memcpy(out, t.v, 32);
    } else {
        memcpy(out, s, 32);
    }
}

static void jac_mul(const field_t *F, jac *r, const aff *b, const u64 *k) {
    jac acc;
    jac_set_inf(F, &acc);
    for (int i = 255; i >= 0; --i) {
        jac_dbl(F, &acc, &acc);
        if ((k[i >> 6] >> (i & 63)) & 1) jac_add_affine(F, &acc, &acc, b);
    }
    *r = acc;
}

/* ------------------------------------------------------------- exported ---- */
void pasta_ref_msm_naive(int curve, const u64 *bases, const u64 *scalars, size_t n, int is_mont,
                         u64 *out) {
    const field_t *F = coord_field(curve);
    jac acc, t;
    jac_set_inf(F, &acc);
    for (size_t i = 0; i < n; ++i) {
        u64 k[4];
        scalar_canon(curve, scalars + 4 * i, is_mont, k);
        jac_mul(F, &t, (const aff *)(bases + 8 * i), k);
        jac_add(F, &acc, &acc, &t);
    }
    memcpy(out, &acc, 96);
}

static inline unsigned get_window(const u64 *k, unsigned seg, unsigned c) {
    unsigned skip_bits = seg * c;
    if (skip_bits >= 256) return 0;
    unsigned limb = skip_bits >> 6, off = skip_bits & 63;
    u64 v = k[limb] >> off;
    if (off + c > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
    return (unsigned)(v & ((1ULL << c) - 1));
}

/* restates halo2-style multiexp_serial: returns sum over one chunk */
static void multiexp_serial(int curve, const aff *bases, const u64 *canon, size_t n, jac *acc) {
    const field_t *F = coord_field(curve);
    unsigned c;
    if (n < 4) c = 1;
    else if (n < 32) c = 3;
    else c = (unsigned)ceil(log((double)n));
    unsigned segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    jac *buckets = (jac *)malloc(nb * sizeof(jac));
    jac_set_inf(F, acc);
    for (int seg = (int)segments - 1; seg >= 0; --seg) {
        for (unsigned k = 0; k < c; ++k) jac_dbl(F, acc, acc);
        for (size_t b = 0; b < nb; ++b) jac_set_inf(F, &buckets[b]);
        for (size_t i = 0; i < n; ++i) {
            unsigned d = get_window(canon + 4 * i, (unsigned)seg, c);
            if (d) jac_add_affine(F, &buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        jac run;
        jac_set_inf(F, &run);
        for (size_t b = nb; b-- > 0;) {
            jac_add(F, &run, &run, &buckets[b]);
            jac_add(F, acc, acc, &run);
        }
    }
    free(buckets);
}

typedef struct {
    int curve;
    const aff *bases;
    const u64 *canon;
    size_t n;
    jac out;
} chunk_job;

static void *chunk_main(void *arg) {
    chunk_job *j = (chunk_job *)arg;
    multiexp_serial(j->curve, j->bases, j->canon, j->n, &j->out);
    return NULL;
}

static void msm_threads(int curve, const aff *bases, const u64 *canon, size_t n, int threads,
                        jac *out) {
    const field_t *F = coord_field(curve);
    if (threads < 1) threads = 1;
    if (n <= (size_t)threads || threads == 1) {
        multiexp_serial(curve, bases, canon, n, out);
        return;
    }
    size_t chunk = n / (size_t)threads;
    size_t nchunks = (n + chunk - 1) / chunk;
    chunk_job *jobs = (chunk_job *)calloc(nchunks, sizeof(chunk_job));
    pthread_t *tids = (pthread_t *)calloc(nchunks, sizeof(pthread_t));
    for (size_t t = 0; t < nchunks; ++t) {
        size_t lo = t * chunk, hi = lo + chunk > n ? n : lo + chunk;
        jobs[t].curve = curve;
        jobs[t].bases = bases + lo;
        jobs[t].canon = canon + 4 * lo;
        jobs[t].n = hi - lo;
        pthread_create(&tids[t], NULL, chunk_main, &jobs[t]);
    }
    jac_set_inf(F, out);
    for (size_t t = 0; t < nchunks; ++t) {
        pthread_join(tids[t], NULL);
        jac_add(F, out, out, &jobs[t].out);
    }
    free(jobs);
    free(tids);
}

void pasta_ref_msm_pippenger(int curve, const u64 *bases, const u64 *scalars, size_t n,
                             int is_mont, int threads, u64 *out) {
    u64 *canon = (u64 *)malloc(32 * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) scalar_canon(curve, scalars + 4 * i, is_mont, canon + 4 * i);
    jac r;
    msm_threads(curve, (const aff *)bases, canon, n, threads, &r);
    memcpy(out, &r, 96);
    free(canon);
}

/* ---------------------------------------------------------------- thread pool ----
 * A persistent pool (workers are created once and parked on a condition variable), so that a mid-size MSM is not
 * billed the creation of a few hundred threads: pasta-msm keeps a pool of its own.  Units of one job are handed out
 * through an atomic counter; the calling thread works too. */
typedef void (*unit_fn)(void *ctx, size_t unit);
static struct {
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    pthread_t *tids;
    int nworkers;          /* threads parked in the pool */
    unsigned long gen;     /* job generation */
    int want;              /* workers that may join the current job */
    int active;            /* workers still inside the current job */
    unit_fn fn;
    void *ctx;
    size_t units;
    size_t next;           /* next unit to hand out (atomic) */
} POOL = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, 0, 0, 0, NULL, NULL, 0, 0};
static pthread_mutex_t POOL_JOB = PTHREAD_MUTEX_INITIALIZER;   /* one job at a time */

static void pool_drain(void) {
    for (;;) {
        size_t u = __atomic_fetch_add(&POOL.next, 1, __ATOMIC_RELAXED);
        if (u >= POOL.units) break;
        POOL.fn(POOL.ctx, u);
    }
}
static void *pool_worker(void *arg) {
    const int id = (int)(size_t)arg;
    unsigned long seen = 0;
    pthread_mutex_lock(&POOL.mu);
    for (;;) {
        while (POOL.gen == seen) pthread_cond_wait(&POOL.go, &POOL.mu);
        seen = POOL.gen;
        if (id >= POOL.want) continue;
        pthread_mutex_unlock(&POOL.mu);
        pool_drain();
        pthread_mutex_lock(&POOL.mu);
        if (--POOL.active == 0) pthread_cond_signal(&POOL.done);
    }
    return NULL;
}
static void pool_run(int threads, size_t units, unit_fn fn, void *ctx) {
    if (units == 0) return;
    if (threads <= 1 || units == 1) {
        for (size_t u = 0; u < units; ++u) fn(ctx, u);
        return;
    }
    pthread_mutex_lock(&POOL_JOB);
    const int helpers = threads - 1;
    pthread_mutex_lock(&POOL.mu);
    if (POOL.nworkers < helpers) {
        POOL.tids = (pthread_t *)realloc(POOL.tids, (size_t)helpers * sizeof(pthread_t));
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        while (POOL.nworkers < helpers) {
            if (pthread_create(&POOL.tids[POOL.nworkers], &at, pool_worker, (void *)(size_t)POOL.nworkers) != 0) break;
            ++POOL.nworkers;
        }
        pthread_attr_destroy(&at);
    }
    POOL.fn = fn; POOL.ctx = ctx; POOL.units = units;
    __atomic_store_n(&POOL.next, 0, __ATOMIC_RELAXED);
    POOL.want = helpers < POOL.nworkers ? helpers : POOL.nworkers;
    POOL.active = POOL.want;
    ++POOL.gen;
    pthread_cond_broadcast(&POOL.go);
    pthread_mutex_unlock(&POOL.mu);
    pool_drain();
    pthread_mutex_lock(&POOL.mu);
    while (POOL.active) pthread_cond_wait(&POOL.done, &POOL.mu);
    pthread_mutex_unlock(&POOL.mu);
    pthread_mutex_unlock(&POOL_JOB);
}

/* threads parked in the pool (what pthread_create granted so far) */
int pasta_ref_pool_size(void) { return POOL.nworkers; }

/* ------------------------------------------------- window-parallel Pippenger ----
 * The shape of pasta-msm's own CPU path [recalled, the crate is not in /root/reference]: ONE window size for the whole
 * input, Booth-recoded signed digits (2^(c-1) buckets per window), the (window, slice of the points) tiles dealt out to
 * a thread pool, running-sum bucket reduction per tile, Horner over the windows at the end.  Unlike the chunk-per-thread
 * form above, the window does not shrink with the thread count, which is what made that form a poor baseline. */
typedef struct {
    int curve;
    const aff *bases;
    const u64 *canon;
    size_t n;
    unsigned c, W, S;
    jac *tile;     /* W * S tile sums */
} wmsm_job;

static inline int booth_digit(const u64 *k, unsigned w, unsigned c) {
    /* c + 1 bits starting at bit w*c - 1 (bit -1 is zero): d = b_{-1} + sum_{j < c-1} b_j 2^j - b_{c-1} 2^(c-1) */
    const unsigned start = w * c;
    u64 v;
    if (start == 0) {
        v = (k[0] << 1) & (((u64)1 << (c + 1)) - 1);
    } else {
        const unsigned pos = start - 1, limb = pos >> 6, off = pos & 63;
        v = limb < 4 ? k[limb] >> off : 0;
        if (off + c + 1 > 64 && limb + 1 < 4) v |= k[limb + 1] << (64 - off);
        v &= ((u64)1 << (c + 1)) - 1;
    }
    const int lo = (int)(v & 1) + (int)((v >> 1) & (((u64)1 << (c - 1)) - 1));
    return lo - (int)(((v >> c) & 1) << (c - 1));
}

static void wmsm_unit(void *ctx, size_t unit) {
    wmsm_job *j = (wmsm_job *)ctx;
    const field_t *F = coord_field(j->curve);
    const unsigned w = (unsigned)(unit / j->S), s = (unsigned)(unit % j->S);
    const size_t lo = j->n * s / j->S, hi = j->n * (s + 1) / j->S;
    const size_t nb = (size_t)1 << (j->c - 1);
    jac *buckets = (jac *)malloc(nb * sizeof(jac));
    unsigned char *used = (unsigned char *)calloc(nb, 1);
    for (size_t i = lo; i < hi; ++i) {
        const int d = booth_digit(j->canon + 4 * i, w, j->c);
        if (d == 0 || aff_is_inf(&j->bases[i])) continue;
        const size_t b = (size_t)(d < 0 ? -d : d) - 1;
        aff p = j->bases[i];
        if (d < 0) fe_neg(F, &p.y, &p.y);
        if (!used[b]) { jac_set_inf(F, &buckets[b]); used[b] = 1; }
        jac_add_affine(F, &buckets[b], &buckets[b], &p);
    }
    jac run, acc;
    jac_set_inf(F, &run);
    jac_set_inf(F, &acc);
    for (size_t b = nb; b-- > 0;) {
        if (used[b]) jac_add(F, &run, &run, &buckets[b]);
        jac_add(F, &acc, &acc, &run);
    }
    j->tile[unit] = acc;
    free(buckets);
    free(used);
}

typedef struct { int curve; const u64 *in; int is_mont; u64 *out; size_t n; size_t per; } canon_job;
static void canon_unit(void *ctx, size_t unit) {
    canon_job *j = (canon_job *)ctx;
    const size_t lo = unit * j->per, hi = lo + j->per > j->n ? j->n : lo + j->per;
    for (size_t i = lo; i < hi; ++i) scalar_canon(j->curve, j->in + 4 * i, j->is_mont, j->out + 4 * i);
}

unsigned pasta_ref_window_plan(size_t n, int threads, unsigned *slices_out) {
    /* cost of the slowest thread in bucket additions: rounds of tiles, each n/S point additions + 2 * 2^(c-1) reduction */
    unsigned best_c = 2, best_s = 1;
    double best = 1e300;
    if (threads < 1) threads = 1;
    for (unsigned c = 2; c <= 20; ++c) {
        const unsigned W = (256 + c - 1) / c;
        for (unsigned S = 1; S <= (unsigned)threads; S = S < 4 ? S + 1 : S + S / 4) {
            const double units = (double)W * S, rounds = ceil(units / threads);
            const double cost = rounds * ((double)n / S + 2.0 * (double)((size_t)1 << (c - 1)));
            if (cost < best) { best = cost; best_c = c; best_s = S; }
        }
    }
    if (slices_out) *slices_out = best_s;
    return best_c;
}

void pasta_ref_msm_pippenger_windows(int curve, const u64 *bases, const u64 *scalars, size_t n, int is_mont, int threads,
                                     u64 *out) {
    const field_t *F = coord_field(curve);
    if (threads < 1) threads = 1;
    jac res;
    jac_set_inf(F, &res);
    if (n == 0) { memcpy(out, &res, 96); return; }
    u64 *canon = (u64 *)malloc(32 * n);
    canon_job cj = {curve, scalars, is_mont, canon, n, 4096};
    pool_run(threads, (n + cj.per - 1) / cj.per, canon_unit, &cj);
    wmsm_job j;
    j.curve = curve; j.bases = (const aff *)bases; j.canon = canon; j.n = n;
    j.c = pasta_ref_window_plan(n, threads, &j.S);
    if (j.S > n) j.S = (unsigned)n;
    j.W = (256 + j.c - 1) / j.c;
    j.tile = (jac *)malloc((size_t)j.W * j.S * sizeof(jac));
    pool_run(threads, (size_t)j.W * j.S, wmsm_unit, &j);
    for (int w = (int)j.W - 1; w >= 0; --w) {
        for (unsigned k = 0; k < j.c; ++k) jac_dbl(F, &res, &res);
        for (unsigned s = 0; s < j.S; ++s) jac_add(F, &res, &res, &j.tile[(size_t)w * j.S + s]);
    }
    memcpy(out, &res, 96);
    free(j.tile);
    free(canon);
}

void pasta_ref_to_affine(int curve, const u64 *jacs, size_t n, u64 *out) {
    const field_t *F = coord_field(curve);
    for (size_t i = 0; i < n; ++i) jac_to_aff(F, (aff *)(out + 8 * i), (const jac *)(jacs + 12 * i));
}

void pasta_ref_compress(int curve, const u64 *jacs, size_t n, uint8_t *out32) {
    const field_t *F = coord_field(curve);
    for (size_t i = 0; i < n; ++i) {
        aff a;
        jac_to_aff(F, &a, (const jac *)(jacs + 12 * i));
        if (aff_is_inf(&a)) { memset(out32 + 32 * i, 0, 32); continue; }
        fe x, y;
        fe_from_mont(F, &x, &a.x);
        fe_from_mont(F, &y, &a.y);
        memcpy(out32 + 32 * i, x.v, 32); /* little-endian host assumed */
        out32[32 * i + 31] |= (uint8_t)((y.v[0] & 1) << 7);
    }
}

void pasta_ref_scalar_mul(int curve, const u64 *base, const u64 *k, u64 *out) {
    jac r;
    jac_mul(coord_field(curve), &r, (const aff *)base, k);
    memcpy(out, &r, 96);
}

static void generator(int curve, aff *g) {
    const field_t *F = coord_field(curve);
    fe one, two = {{2, 0, 0, 0}};
    memcpy(one.v, F->r1, 32);
    fe_neg(F, &g->x, &one); /* x = -1 */
    fe_to_mont(F, &g->y, &two);
}

void pasta_ref_gen_bases_ap(int curve, u64 k0, u64 d, size_t n, u64 *out) {
    const field_t *F = coord_field(curve);
    aff g, step;
    generator(curve, &g);
    u64 kk[4] = {k0, 0, 0, 0}, dd[4] = {d, 0, 0, 0};
    jac cur, st;
    jac_mul(F, &cur, &g, kk);
    jac_mul(F, &st, &g, dd);
    jac_to_aff(F, &step, &st);
    /* Jacobian running sum, then batch inversion (Montgomery's trick) for the affine forms */
    jac *pts = (jac *)malloc(sizeof(jac) * (n ? n : 1));
    fe *pre = (fe *)malloc(sizeof(fe) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) {
        pts[i] = cur;
        jac_add_affine(F, &cur, &cur, &step);
    }
    fe acc;
    memcpy(acc.v, F->r1, 32);
    for (size_t i = 0; i < n; ++i) {
        pre[i] = acc;
        if (!jac_is_inf(&pts[i])) fe_mul(F, &acc, &acc, &pts[i].z);
    }
    fe inv;
    fe_inv(F, &inv, &acc);
    for (size_t i = n; i-- > 0;) {
        aff *o = (aff *)(out + 8 * i);
        if (jac_is_inf(&pts[i])) { memset(o, 0, 64); continue; }
        fe zi, zi2, zi3;
        fe_mul(F, &zi, &inv, &pre[i]);
        fe_mul(F, &inv, &inv, &pts[i].z);
        fe_sqr(F, &zi2, &zi);
        fe_mul(F, &zi3, &zi2, &zi);
        fe_mul(F, &o->x, &pts[i].x, &zi2);
        fe_mul(F, &o->y, &pts[i].y, &zi3);
    }
    free(pts);
    free(pre);
}

static inline u64 splitmix(u64 *s) {
    u64 z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void pasta_ref_gen_scalars(int curve, u64 seed, int kind, u64 small_bound, size_t n, int to_mont,
                           u64 *out) {
    const field_t *S = scalar_field(curve);
    u64 st = seed;
    for (size_t i = 0; i < n; ++i) {
        u64 w[4];
        for (int j = 0; j < 4; ++j) w[j] = splitmix(&st); /* always 4 draws per scalar */
        fe v;
        if (kind == 2) {
            v.v[0] = small_bound ? w[0] % small_bound : 0;
            v.v[1] = v.v[2] = v.v[3] = 0;
        } else {
            int cls = 2;
            if (kind == 1) {
                unsigned sel = (unsigned)(w[3] >> 56) % 10; /* top byte picks the class */
                cls = sel < 7 ? 0 : (sel < 9 ? 1 : 2);
            }
            if (cls == 0) { v.v[0] = w[0] & 1; v.v[1] = v.v[2] = v.v[3] = 0; }
            else if (cls == 1) { v.v[0] = w[0] & 0xffff; v.v[1] = v.v[2] = v.v[3] = 0; }
            else {
                memcpy(v.v, w, 32);
                v.v[3] &= 0x7fffffffffffffffULL;
                if (ge_mod(v.v, S->m)) sub4(v.v, v.v, S->m);
            }
        }
        if (to_mont) fe_to_mont(S, &v, &v);
        memcpy(out + 4 * i, v.v, 32);
    }
}

void pasta_ref_fmul(int f, const u64 *a, const u64 *b, u64 *o) { fe_mul(field_of(f), (fe *)o, (const fe *)a, (const fe *)b); }
void pasta_ref_fadd(int f, const u64 *a, const u64 *b, u64 *o) { fe_add(field_of(f), (fe *)o, (const fe *)a, (const fe *)b); }
void pasta_ref_fsub(int f, const u64 *a, const u64 *b, u64 *o) { fe_sub(field_of(f), (fe *)o, (const fe *)a, (const fe *)b); }
void pasta_ref_finv(int f, const u64 *a, u64 *o) { fe_inv(field_of(f), (fe *)o, (const fe *)a); }
void pasta_ref_to_mont(int f, const u64 *a, u64 *o) { fe_to_mont(field_of(f), (fe *)o, (const fe *)a); }
void pasta_ref_from_mont(int f, const u64 *a, u64 *o) { fe_from_mont(field_of(f), (fe *)o, (const fe *)a); }

void pasta_ref_fold(int curve, const u64 *gens, size_t half, const u64 *w1, const u64 *w2, u64 *out) {
    const field_t *F = coord_field(curve);
    const aff *L = (const aff *)gens, *R = L + half;
    for (size_t i = 0; i < half; ++i) {
        jac a, b;
        jac_mul(F, &a, &L[i], w1);
        jac_mul(F, &b, &R[i], w2);
        jac_add(F, &a, &a, &b);
        jac_to_aff(F, (aff *)(out + 8 * i), &a);
    }
}

typedef struct { int curve; const aff *L, *R; size_t half; const u64 *w1, *w2; u64 *out; size_t per; } fold_job;
static void fold_unit(void *ctx, size_t unit) {
    fold_job *j = (fold_job *)ctx;
    const field_t *F = coord_field(j->curve);
    const size_t lo = unit * j->per, hi = lo + j->per > j->half ? j->half : lo + j->per;
    for (size_t i = lo; i < hi; ++i) {
        /* one joint double-and-add chain per pair: what nova's 2-term vartime_multiscalar_mul (cpu_best_multiexp with
         * c = 1 for n < 4 [recalled]) amounts to -- 256 doublings shared by both scalars */
        jac a;
        jac_set_inf(F, &a);
        for (int bit = 255; bit >= 0; --bit) {
            jac_dbl(F, &a, &a);
            if ((j->w1[bit >> 6] >> (bit & 63)) & 1) jac_add_affine(F, &a, &a, &j->L[i]);
            if ((j->w2[bit >> 6] >> (bit & 63)) & 1) jac_add_affine(F, &a, &a, &j->R[i]);
        }
        jac_to_aff(F, (aff *)(j->out + 8 * i), &a);
    }
}
/* the same fold dealt out to the thread pool (nova folds its generators with rayon) */
void pasta_ref_fold_mt(int curve, const u64 *gens, size_t half, const u64 *w1, const u64 *w2, int threads, u64 *out) {
    fold_job j = {curve, (const aff *)gens, (const aff *)gens + half, half, w1, w2, out, 8};
    pool_run(threads, (half + j.per - 1) / j.per, fold_unit, &j);
}

typedef struct {
    int curve;
    const aff *bases;
    const aff *h;
    const u64 *scalars;
    const u64 *blinds;
    size_t row_lo, row_hi, row_len;
    int is_mont;
    jac *out;
} row_job;

static void *row_main(void *arg) {
    row_job *j = (row_job *)arg;
    const field_t *F = coord_field(j->curve);
    u64 *canon = (u64 *)malloc(32 * j->row_len);
    for (size_t r = j->row_lo; r < j->row_hi; ++r) {
        for (size_t i = 0; i < j->row_len; ++i)
            scalar_canon(j->curve, j->scalars + 4 * (r * j->row_len + i), j->is_mont, canon + 4 * i);
        multiexp_serial(j->curve, j->bases, canon, j->row_len, &j->out[r]);
        if (j->blinds) {
            u64 k[4];
            jac t;
            scalar_canon(j->curve, j->blinds + 4 * r, j->is_mont, k);
            jac_mul(F, &t, j->h, k);
            jac_add(F, &j->out[r], &j->out[r], &t);
        }
    }
    free(canon);
    return NULL;
}

void pasta_ref_row_msm(int curve, const u64 *bases, const u64 *h, const u64 *scalars,
                       const u64 *blinds, size_t rows, size_t row_len, int is_mont, int threads,
                       u64 *out) {
    if (threads < 1) threads = 1;
    if ((size_t)threads > rows) threads = rows ? (int)rows : 1;
    row_job *jobs = (row_job *)calloc((size_t)threads, sizeof(row_job));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    size_t per = (rows + (size_t)threads - 1) / (size_t)threads;
    for (int t = 0; t < threads; ++t) {
        size_t lo = (size_t)t * per, hi = lo + per > rows ? rows : lo + per;
        if (lo > rows) lo = rows;
        jobs[t] = (row_job){curve, (const aff *)bases, (const aff *)h, scalars, blinds,
                            lo, hi, row_len, is_mont, (jac *)out};
        pthread_create(&tids[t], NULL, row_main, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
    free(jobs);
    free(tids);
}

/* ---- row N2: sum-check round on the CPU (restates linear_mle_product, r1cs_helper.rs:441-506,
 * with the challenge supplied by the caller).  Tables hold canonical integers; they are moved to
 * Montgomery form once (pasta_ref_sc_to_mont) so that a round costs what the reference's
 * rug::Integer multiply + rem_floor costs: one modular product per term. */
void pasta_ref_sc_to_mont(int field, u64 *table, size_t n) {
    const field_t *F = field_of(field);
    for (size_t i = 0; i < n; ++i) fe_to_mont(F, (fe *)(table + 4 * i), (const fe *)(table + 4 * i));
}
void pasta_ref_sc_from_mont(int field, u64 *table, size_t n) {
    const field_t *F = field_of(field);
    for (size_t i = 0; i < n; ++i) fe_from_mont(F, (fe *)(table + 4 * i), (const fe *)(table + 4 * i));
}
/* out3 = (xsq, x, con) canonical; then both tables are folded with r (canonical). */
void pasta_ref_sc_round(int field, u64 *T, u64 *E, size_t pow, const u64 *r_canon, u64 *out3) {
    const field_t *F = field_of(field);
    fe xsq, x, con, r, t;
    memset(&xsq, 0, 32); memset(&x, 0, 32); memset(&con, 0, 32);
    for (size_t b = 0; b < pow; ++b) {
        const fe *t0 = (const fe *)(T + 4 * b), *t1 = (const fe *)(T + 4 * (b + pow));
        const fe *e0 = (const fe *)(E + 4 * b), *e1 = (const fe *)(E + 4 * (b + pow));
        fe ts, es;
        fe_sub(F, &ts, t1, t0);
        fe_sub(F, &es, e1, e0);
        fe_mul(F, &t, &ts, &es); fe_add(F, &xsq, &xsq, &t);
        fe_mul(F, &t, &es, t0);  fe_add(F, &x, &x, &t);
        fe_mul(F, &t, &ts, e0);  fe_add(F, &x, &x, &t);
        fe_mul(F, &t, t0, e0);   fe_add(F, &con, &con, &t);
    }
    fe_from_mont(F, (fe *)out3, &xsq);
    fe_from_mont(F, (fe *)(out3 + 4), &x);
    fe_from_mont(F, (fe *)(out3 + 8), &con);
    fe_to_mont(F, &r, (const fe *)r_canon);
    for (size_t b = 0; b < pow; ++b) {
        fe *t0 = (fe *)(T + 4 * b), *e0 = (fe *)(E + 4 * b);
        const fe *t1 = (const fe *)(T + 4 * (b + pow)), *e1 = (const fe *)(E + 4 * (b + pow));
        fe d;
        fe_sub(F, &d, t1, t0); fe_mul(F, &d, &d, &r); fe_add(F, t0, t0, &d);
        fe_sub(F, &d, e1, e0); fe_mul(F, &d, &d, &r); fe_add(F, e0, e0, &d);
    }
}
