// third_row_sim.cpp -- CPU experiment behind DESIGN.md section 8 "the one route to a third resident row" (round 6): would a level
// sketch in 16-bit cells and an exact table with fp32 accumulators still do the sketch kernel's job?
// Per level of a row (semantics of the row walk as oracle/gfpush_oracle.cpp, graph.h:73-110):
//   A. today's sketch: M 32-bit cells, add ceil(share * 2^31), candidate iff cell >= floor(rmax * dq * 2^31 * (1 - 2^-10))
//   B. 16-bit cells with a PER-LEVEL quantum q = (mass pushed into the level) / 60000: add ceil(share / q) -- the cells of a level
//      then sum to <= 60000 + edges, no cell can overflow 16 bits while the level has < 5 500 edges (else q doubles) --, candidate
//      iff cell >= floor(rmax * dq / q * (1 - 2^-10))
//   C. for B's candidate nodes: fp32 accumulation of the shares (in edge order); SUPERSET of the pushers = nodes with
//      r32 >= rmax * deg * (1 - 2^-16); counted: superset size, true pushers, true pushers MISSED by the superset (must be 0)
// Counts per level, summed over rows; per row the largest level's candidate nodes under A and B (what the exact table must hold).
// Build: g++ -O3 -fopenmp -shared -fPIC.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <omp.h>

static inline uint32_t cell_of(uint32_t k, uint32_t lg) { return ((k & 0xFFFFFFu) * 0x9E3779u) >> (32 - lg); }     // the kernel's sk_cell

extern "C" int third_row_sim(const int32_t* indptr, const int32_t* indices, int64_t n, const int32_t* seeds, int64_t n_seeds,
                             const double* coef, int n_coef, double rmax, uint32_t deg_sat, uint32_t lg_m,
                             // out[level][0..9]: edges, targets, pushers, A cand edges, A cand nodes, B cand edges, B cand nodes, superset, missed, levels whose quantum doubled
                             double* out, uint32_t* row_max_a, uint32_t* row_max_b, uint32_t* row_max_super)
{
    const int L = n_coef - 1;
    const int nth = omp_get_max_threads();
    const uint32_t M = 1u << lg_m;
    std::vector<std::vector<double>> acc(nth, std::vector<double>((size_t)(L + 1) * 10, 0.0));
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        std::vector<double> nxt(n, 0.0);
        std::vector<float> nxt32(n, 0.0f);
        std::vector<int32_t> touched; touched.reserve(1 << 16);
        std::vector<std::pair<int32_t, double>> fr, fr2, edges;
        std::vector<uint32_t> ska(M, 0), skb(M, 0);
#pragma omp for schedule(dynamic, 4)
        for (int64_t it = 0; it < n_seeds; ++it) {
            const int32_t src = seeds[it];
            fr.clear(); fr.push_back({src, 1.0});
            uint32_t mx_a = 0, mx_b = 0, mx_s = 0;
            for (int lvl = 0; lvl < L; ++lvl) {
                edges.clear(); touched.clear();
                double dangling = 0.0; bool any_d = false; double mass = 0.0;
                for (auto& ur : fr) {
                    const int32_t u = ur.first; const double r = ur.second;
                    const uint32_t deg = (uint32_t)(indptr[u + 1] - indptr[u]);
                    if (deg == 0) { dangling += r; any_d = true; }
                    else if (r >= rmax * deg) {
                        const double sh = r / deg;
                        for (int32_t j = indptr[u]; j < indptr[u + 1]; ++j) {
                            const int32_t v = indices[j];
                            if (nxt[v] == 0.0) touched.push_back(v);
                            nxt[v] += sh; nxt32[v] += (float)sh;
                            edges.push_back({v, sh}); mass += sh;
                        }
                    }
                }
                if (any_d) { if (nxt[src] == 0.0) touched.push_back(src); nxt[src] += dangling; nxt32[src] += (float)dangling; edges.push_back({src, dangling}); mass += dangling; }
                double* o = &acc[t][(size_t)(lvl + 1) * 10];
                o[0] += (double)edges.size(); o[1] += (double)touched.size();
                double q = mass / 60000.0;
                while (q > 0.0 && 60000.0 * (mass / 60000.0) / q + (double)edges.size() > 65535.0) { q *= 2.0; o[9] += 1.0; }
                for (auto& e : edges) {
                    const uint32_t c = cell_of((uint32_t)e.first, lg_m);
                    ska[c] += (uint32_t)std::ceil(e.second * 2147483648.0);
                    if (q > 0.0) skb[c] += (uint32_t)std::ceil(e.second / q);
                }
                fr2.clear();
                uint32_t can = 0, cbn = 0, sup = 0;
                for (int32_t v : touched) {
                    const double r = nxt[v];
                    const uint32_t deg = (uint32_t)(indptr[v + 1] - indptr[v]);
                    const uint32_t dq = std::min(deg, deg_sat);
                    const bool pusher = deg == 0 || r >= rmax * deg;
                    if (pusher) o[2] += 1.0;
                    const uint32_t c = cell_of((uint32_t)v, lg_m);
                    const bool ca = (double)ska[c] >= std::floor(rmax * dq * 2147483648.0 * (1.0 - 1.0 / 1024.0));
                    const bool cb = q > 0.0 && (double)skb[c] >= std::floor(rmax * dq / q * (1.0 - 1.0 / 1024.0));
                    if (ca) ++can;
                    if (cb) {
                        ++cbn;
                        const bool in_super = deg == 0 || (double)nxt32[v] >= rmax * deg * (1.0 - 1.0 / 65536.0);
                        if (in_super) ++sup;
                        if (pusher && !in_super) o[8] += 1.0;
                    } else if (pusher) o[8] += 1.0;             // a pusher the 16-bit sketch did not let through: must never happen
                    fr2.push_back({v, r});
                }
                uint64_t cae = 0, cbe = 0;
                for (auto& e : edges) {
                    const int32_t v = e.first;
                    const uint32_t deg = (uint32_t)(indptr[v + 1] - indptr[v]);
                    const uint32_t dq = std::min(deg, deg_sat);
                    const uint32_t c = cell_of((uint32_t)v, lg_m);
                    if ((double)ska[c] >= std::floor(rmax * dq * 2147483648.0 * (1.0 - 1.0 / 1024.0))) ++cae;
                    if (q > 0.0 && (double)skb[c] >= std::floor(rmax * dq / q * (1.0 - 1.0 / 1024.0))) ++cbe;
                }
                o[3] += (double)cae; o[4] += can; o[5] += (double)cbe; o[6] += cbn; o[7] += sup;
                mx_a = std::max(mx_a, can); mx_b = std::max(mx_b, cbn); mx_s = std::max(mx_s, sup);
                for (auto& e : edges) { const uint32_t c = cell_of((uint32_t)e.first, lg_m); ska[c] = 0; skb[c] = 0; }
                for (int32_t v : touched) { nxt[v] = 0.0; nxt32[v] = 0.0f; }
                fr.swap(fr2);
            }
            row_max_a[it] = mx_a; row_max_b[it] = mx_b; row_max_super[it] = mx_s;
        }
    }
    std::memset(out, 0, sizeof(double) * (size_t)(L + 1) * 10);
    for (int t = 0; t < nth; ++t) for (size_t i = 0; i < (size_t)(L + 1) * 10; ++i) out[i] += acc[t][i];
    return 0;
}
