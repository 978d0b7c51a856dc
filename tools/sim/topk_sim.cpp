// topk_sim.cpp -- CPU experiment for the round-4 TOP-K: a CUMULATIVE keyless fixed-point table ("reserve sketch") R[h(v) % M]
// += ceil(coef*share*2^31) over every pushed edge of every level is an UPPER bound on a node's final reserve
// (graph.h:90,109), so nodes in light cells cannot be among the K largest.  Procedure emulated per row:
//   A  t_c = cell value of rank `target` (rounded DOWN to a histogram bin: top 8 bits below the leading one)
//   B  exact totals of every node whose cell >= t_c (one sweep over the per-edge log)
//   C  tau = K-th largest exact total; if tau*2^31 >= t_c every unswept node is below tau: done.  Else sweep again with t_c = tau.
// Reports nodes / records aggregated per sweep, how often a second sweep is needed, and checks the result against the
// true top-K.  Row walk as in oracle/gfpush_oracle.cpp (graph.h:73-110).  Build: g++ -O3 -fopenmp -shared -fPIC.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <omp.h>
static inline uint32_t h1(uint32_t k) { k *= 0x9E3779B1u; k ^= k >> 15; k *= 0x85EBCA77u; return k; }
static inline uint32_t slot_of(uint32_t h, uint32_t cap) { return (uint32_t)(((uint64_t)h * cap) >> 32); }
static inline uint32_t bin_floor(uint32_t x) {           // keep the leading one + 5 bits
    if (x < 64) return x;
    const int lz = __builtin_clz(x); const int sh = 32 - lz - 6; return (x >> sh) << sh;
}
extern "C" int topk_sim(const int32_t* indptr, const int32_t* indices, int64_t n, const int32_t* seeds, int64_t n_seeds,
                        const double* coef, int n_coef, double rmax, int K, uint32_t M, uint32_t target,
                        double* out /* [16] sums */, uint32_t* per_row /* [n_seeds][4]: nodes sweep1, records sweep1, nodes sweep2 (0 = none), support */)
{
    const int L = n_coef - 1;
    double acc[16] = {0};
#pragma omp parallel
    {
        std::vector<double> nxt(n, 0.0), tot(n, 0.0);
        std::vector<int32_t> touched, supp;
        std::vector<std::pair<int32_t, double>> fr, fr2, log;
        std::vector<uint32_t> R(M);
        double a[16] = {0};
#pragma omp for schedule(dynamic, 4)
        for (int64_t it = 0; it < n_seeds; ++it) {
            const int32_t src = seeds[it];
            fr.clear(); fr.push_back({src, 1.0}); log.clear(); supp.clear();
            log.push_back({src, coef[0] * 1.0});
            for (int lvl = 0; lvl < L; ++lvl) {
                touched.clear(); double dangling = 0.0; bool any_d = false;
                for (auto& ur : fr) {
                    const int32_t u = ur.first; const double r = ur.second;
                    const uint32_t deg = (uint32_t)(indptr[u + 1] - indptr[u]);
                    if (deg == 0) { dangling += r; any_d = true; }
                    else if (r >= rmax * deg) {
                        const double sh = r / deg;
                        for (int32_t j = indptr[u]; j < indptr[u + 1]; ++j) {
                            const int32_t v = indices[j];
                            if (nxt[v] == 0.0) touched.push_back(v);
                            nxt[v] += sh; log.push_back({v, coef[lvl + 1] * sh});
                        }
                    }
                }
                if (any_d) { if (nxt[src] == 0.0) touched.push_back(src); nxt[src] += dangling; log.push_back({src, coef[lvl + 1] * dangling}); }
                fr2.clear();
                for (int32_t v : touched) { fr2.push_back({v, nxt[v]}); nxt[v] = 0.0; }
                fr.swap(fr2);
            }
            std::fill(R.begin(), R.end(), 0u);
            for (auto& e : log) {
                if (tot[e.first] == 0.0 && e.second > 0.0) supp.push_back(e.first);
                tot[e.first] += e.second;
                R[slot_of(h1((uint32_t)e.first), M)] += (uint32_t)std::ceil(e.second * 2147483648.0);
            }
            // truth
            std::vector<double> vals; vals.reserve(supp.size());
            for (int32_t v : supp) vals.push_back(tot[v]);
            const size_t k = std::min<size_t>((size_t)K, vals.size());
            double kth = 0.0;
            if (k > 0) { std::nth_element(vals.begin(), vals.begin() + (k - 1), vals.end(), std::greater<double>()); kth = vals[k - 1]; }
            // A
            std::vector<uint32_t> cells(R); std::sort(cells.begin(), cells.end(), std::greater<uint32_t>());
            uint32_t t_c = bin_floor(cells[std::min<uint32_t>(target, M) - 1]);
            if (t_c == 0) t_c = 1;
            // B
            uint32_t n1 = 0, r1 = 0; std::vector<double> agg;
            for (int32_t v : supp) if (R[slot_of(h1((uint32_t)v), M)] >= t_c) { ++n1; agg.push_back(tot[v]); }
            for (auto& e : log) if (R[slot_of(h1((uint32_t)e.first), M)] >= t_c) ++r1;
            // C
            double tau = 0.0; uint32_t n2 = 0;
            if (agg.size() >= (size_t)K) { std::nth_element(agg.begin(), agg.begin() + (K - 1), agg.end(), std::greater<double>()); tau = agg[K - 1]; }
            const double tau_fx = std::floor(tau * 2147483648.0 * (1.0 - 1.0 / 1048576.0));
            bool second = !(tau_fx >= (double)t_c) && agg.size() < supp.size();
            if (second) {
                const uint32_t t2 = tau_fx >= 1.0 ? (uint32_t)tau_fx : 1u;
                agg.clear();
                for (int32_t v : supp) if (R[slot_of(h1((uint32_t)v), M)] >= t2) { ++n2; agg.push_back(tot[v]); }
            }
            // check: K-th of aggregated == true K-th
            double kth2 = 0.0; const size_t k2 = std::min<size_t>((size_t)K, agg.size());
            if (k2 > 0) { std::nth_element(agg.begin(), agg.begin() + (k2 - 1), agg.end(), std::greater<double>()); kth2 = agg[k2 - 1]; }
            if (k2 != k || kth2 != kth) a[5] += 1;
            a[0] += n1; a[1] += r1; a[2] += second ? 1 : 0; a[3] += n2; a[4] += (double)supp.size(); a[6] += (double)log.size();
            a[7] = std::max(a[7], (double)n1); a[8] = std::max(a[8], (double)n2); a[9] += kth; a[10] += (double)t_c / 2147483648.0;
            per_row[it * 4 + 0] = n1; per_row[it * 4 + 1] = r1; per_row[it * 4 + 2] = n2; per_row[it * 4 + 3] = (uint32_t)supp.size();
            for (int32_t v : supp) tot[v] = 0.0;
        }
#pragma omp critical
        { for (int i = 0; i < 16; ++i) { if (i == 7 || i == 8) acc[i] = std::max(acc[i], a[i]); else acc[i] += a[i]; } }
    }
    for (int i = 0; i < 16; ++i) out[i] = acc[i];
    return 0;
}
