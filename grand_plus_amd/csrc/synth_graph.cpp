// synth_graph.cpp -- deterministic synthetic power-law CSR generator (host, OpenMP).
//
// Bench/test workload generator for the shapes named in BASELINE.json (Reddit-,
// Amazon2M- and MAG-Scholar-C-shape): the reference ships no such graphs and there is
// no network.  Everything is INTEGER arithmetic (splitmix64 counter RNG, 32.32 fixed
// point, integer square roots), so the same (n, samples, seed, offset) regenerates the
// same CSR bit for bit on any host -- checksums are committed in tests/golden/.
//
// Model: Chung-Lu-style.  Endpoint weights w_i ~ (i + i0)^(-3/4), i.e. degree
// exponent gamma = 1 + 4/3 ~ 2.33.  An endpoint is drawn by inverse CDF:
//      t = A + u (B - A),  A = i0^(1/4), B = (n + i0)^(1/4),  i = floor(t^4) - i0
// then relabelled by an affine bijection so that node id is uncorrelated with degree.
// Undirected samples are symmetrised, duplicates merged, sampled self-pairs dropped,
// then a self-loop is added to every node (the caller-side `adj + I` of the
// reference, model.py:243) and columns are sorted per row.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

typedef unsigned __int128 u128;

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static uint64_t isqrt128(u128 x) {           // floor(sqrt(x)), x < 2^128
    if (x == 0) return 0;
    u128 lo = 0, hi = ((u128)1 << 64) - 1;
    while (lo < hi) {
        u128 mid = lo + (hi - lo + 1) / 2;
        if (mid * mid <= x) lo = mid; else hi = mid - 1;   // mid < 2^64 so mid*mid fits
    }
    return (uint64_t)lo;
}

// floor(x^(1/4) * 2^32) up to the two nested floors (deterministic, which is all we need)
static uint64_t root4_fx32(uint64_t x) {
    uint64_t s = isqrt128((u128)x << 64);          // sqrt(x) * 2^32
    return isqrt128((u128)s << 32);                // sqrt(sqrt(x)*2^32 * 2^32) = x^(1/4) * 2^32
}

struct Sampler {
    uint64_t n, i0, A, span, mult, add, seed;
    inline uint32_t node(uint64_t ctr) const {
        const uint64_t u = splitmix64(seed ^ (ctr * 0xD1342543DE82EF95ull)) >> 32;   // 32-bit fraction
        const uint64_t t = A + (uint64_t)(((u128)u * span) >> 32);                   // 32.32
        const uint64_t t2 = (uint64_t)(((u128)t * t) >> 32);                         // (t/2^32)^2 * 2^32
        uint64_t i = (uint64_t)(((u128)t2 * t2) >> 64);                              // floor((t/2^32)^4)
        i = i > i0 ? i - i0 : 0;
        if (i >= n) i = n - 1;
        return (uint32_t)(((u128)i * mult + add) % n);                               // affine relabel
    }
};

static Sampler make_sampler(int64_t n, uint64_t seed, int64_t i0) {
    Sampler s;
    s.n = (uint64_t)n; s.i0 = (uint64_t)i0; s.seed = splitmix64(seed);
    s.A = root4_fx32(s.i0);
    s.span = root4_fx32(s.n + s.i0) - s.A;
    s.mult = 2654435761ull;                       // prime; coprime with n unless n is a multiple
    while (s.n % s.mult == 0 || std::__gcd(s.mult % s.n ? s.mult % s.n : s.n, s.n) != 1) s.mult += 2;
    s.add = splitmix64(seed ^ 0xA5A5A5A5ull) % s.n;
    return s;
}

extern "C" {

// Generates the CSR.  *indptr_out (n+1 int32) and *indices_out (nnz int32) are malloc'd;
// release both with gp_synth_free.  Returns 0 on success.
int gp_synth_powerlaw_csr(int64_t n_nodes, int64_t n_samples, uint64_t seed, int64_t offset_i0,
                          int32_t** indptr_out, int32_t** indices_out, int64_t* nnz_out)
{
    if (n_nodes < 1 || n_nodes > 2000000000ll || n_samples < 0 || offset_i0 < 0 ||
        !indptr_out || !indices_out || !nnz_out) return -1;
    if (2 * n_samples + n_nodes >= 2147483647ll) return -2;          // int32 CSR offsets
    const Sampler smp = make_sampler(n_nodes, seed, offset_i0);
    const int64_t n = n_nodes;

    std::vector<int64_t> cnt((size_t)n + 1, 0);
    // pass 1: degrees of the symmetrised multigraph (+1 self-loop per node)
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n_samples; ++k) {
        const uint32_t a = smp.node(2 * (uint64_t)k), b = smp.node(2 * (uint64_t)k + 1);
        if (a == b) continue;
        __atomic_fetch_add(&cnt[a], 1, __ATOMIC_RELAXED);
        __atomic_fetch_add(&cnt[b], 1, __ATOMIC_RELAXED);
    }
    std::vector<int64_t> off((size_t)n + 1);
    off[0] = 0;
    for (int64_t i = 0; i < n; ++i) off[i + 1] = off[i] + cnt[i] + 1;
    const int64_t raw = off[n];
    int32_t* buf = (int32_t*)malloc((size_t)raw * sizeof(int32_t));
    if (!buf) return -3;
    std::vector<int64_t> cur(off.begin(), off.begin() + n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) buf[cur[i]++] = (int32_t)i;      // the self-loop
    // pass 2: scatter (slot order inside a row is racy; rows are sorted next)
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < n_samples; ++k) {
        const uint32_t a = smp.node(2 * (uint64_t)k), b = smp.node(2 * (uint64_t)k + 1);
        if (a == b) continue;
        buf[__atomic_fetch_add(&cur[a], 1, __ATOMIC_RELAXED)] = (int32_t)b;
        buf[__atomic_fetch_add(&cur[b], 1, __ATOMIC_RELAXED)] = (int32_t)a;
    }
    // sort + unique per row
    std::vector<int64_t> ucnt((size_t)n);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
        int32_t* b = buf + off[i];
        int32_t* e = buf + off[i + 1];
        std::sort(b, e);
        ucnt[i] = std::unique(b, e) - b;
    }
    int32_t* indptr = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    if (!indptr) { free(buf); return -3; }
    int64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) { indptr[i] = (int32_t)acc; acc += ucnt[i]; }
    indptr[n] = (int32_t)acc;
    int32_t* indices = (int32_t*)malloc((size_t)(acc > 0 ? acc : 1) * sizeof(int32_t));
    if (!indices) { free(buf); free(indptr); return -3; }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i)
        memcpy(indices + indptr[i], buf + off[i], (size_t)ucnt[i] * sizeof(int32_t));
    free(buf);
    *indptr_out = indptr; *indices_out = indices; *nnz_out = acc;
    return 0;
}

void gp_synth_free(void* p) { free(p); }

// Threads used by gp_synth_powerlaw_csr (0 = OpenMP default).  Explicit because several ranks of one
// node generate the same graph at once and OMP_NUM_THREADS is read before Python can set it.
void gp_synth_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

// S distinct seed nodes: an affine walk over 0..n-1 (a seeded permutation prefix).
int gp_synth_seeds(int64_t n_nodes, int64_t n_seeds, uint64_t seed, int32_t* out)
{
    if (n_nodes < 1 || n_seeds < 0 || n_seeds > n_nodes || !out) return -1;
    uint64_t n = (uint64_t)n_nodes, mult = 0x9E3779B1ull | 1ull;
    while (std::__gcd(mult % n ? mult % n : n, n) != 1) mult += 2;
    const uint64_t add = splitmix64(seed ^ 0x5EED5EEDull) % n;
    for (int64_t k = 0; k < n_seeds; ++k)
        out[k] = (int32_t)(((u128)(uint64_t)k * mult + add) % n);
    return 0;
}

// Order-sensitive 64-bit checksum of a byte buffer (8-byte words, FNV-like over splitmix).
uint64_t gp_checksum64(const void* data, int64_t nbytes)
{
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h = 0xCBF29CE484222325ull;
    int64_t i = 0;
    for (; i + 8 <= nbytes; i += 8) {
        uint64_t w; memcpy(&w, p + i, 8);
        h = splitmix64(h ^ w);
    }
    uint64_t tail = 0;
    if (i < nbytes) { memcpy(&tail, p + i, (size_t)(nbytes - i)); h = splitmix64(h ^ tail ^ 0xFFull); }
    return h;
}

}  // extern "C"
