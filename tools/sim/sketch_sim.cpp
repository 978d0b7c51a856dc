// sketch_sim.cpp -- CPU experiment behind the round-4 kernel structure (DESIGN.md "Round 4"): how selective is a KEYLESS
// fixed-point upper-bound table ("sketch") as a filter in front of the exact residue table?
// Per level of a row: every pushed edge (v, share) adds ceil(share * 2^31) to sketch[h(v) % M]; a target can only push
// (graph.h:94: r >= rmax*deg) if its sketch cell reaches floor(rmax*min(deg,deg_sat)*2^31*(1-2^-20)).  Counts, per level:
// edges, distinct targets, true pushers, candidate targets / candidate edges for several M and 1 or 2 hash functions.
// Semantics of the row walk follow oracle/gfpush_oracle.cpp (graph.h:73-110).  Build: g++ -O3 -fopenmp -shared -fPIC.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <omp.h>

static inline uint32_t h1(uint32_t k) { k *= 0x9E3779B1u; k ^= k >> 15; k *= 0x85EBCA77u; return k; }
static inline uint32_t h2(uint32_t k) { k *= 0x7FEB352Du; k ^= k >> 16; k *= 0x846CA68Bu; return k; }
static inline uint32_t slot_of(uint32_t h, uint32_t cap) { return (uint32_t)(((uint64_t)h * cap) >> 32); }

extern "C" int sketch_sim(const int32_t* indptr, const int32_t* indices, int64_t n, const int32_t* seeds, int64_t n_seeds,
                          const double* coef, int n_coef, double rmax, uint32_t deg_sat,
                          const uint32_t* sizes, int n_sizes,
                          // out[level][0..3 + 4*n_sizes): edges, targets, pushers, dropped-mass*1e9 ; per size: cand_nodes_1h, cand_edges_1h, cand_nodes_2h, cand_edges_2h
                          double* out, int out_stride,
                          // per row: max over levels of candidate nodes (1 hash, size index 0..n_sizes-1), max level edges
                          uint32_t* row_max_cand, uint32_t* row_max_edges)
{
    const int L = n_coef - 1;
    const int nth = omp_get_max_threads();
    std::vector<std::vector<double>> acc(nth, std::vector<double>((size_t)(L + 1) * out_stride, 0.0));
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        std::vector<double> nxt(n, 0.0);
        std::vector<int32_t> touched; touched.reserve(1 << 16);
        std::vector<std::pair<int32_t, double>> fr, fr2;
        std::vector<std::pair<int32_t, double>> edges;
        std::vector<std::vector<uint32_t>> sk1(n_sizes), sk2(n_sizes);
        for (int s = 0; s < n_sizes; ++s) { sk1[s].assign(sizes[s], 0); sk2[s].assign(sizes[s], 0); }
        std::vector<uint8_t> cflag;
#pragma omp for schedule(dynamic, 4)
        for (int64_t it = 0; it < n_seeds; ++it) {
            const int32_t src = seeds[it];
            fr.clear(); fr.push_back({src, 1.0});
            uint32_t mx_edges = 0; std::vector<uint32_t> mx_c(n_sizes, 0);
            for (int lvl = 0; lvl < L; ++lvl) {
                edges.clear(); touched.clear();
                double dangling = 0.0; bool any_d = false;
                for (auto& ur : fr) {
                    const int32_t u = ur.first; const double r = ur.second;
                    const uint32_t deg = (uint32_t)(indptr[u + 1] - indptr[u]);
                    if (deg == 0) { dangling += r; any_d = true; }
                    else if (r >= rmax * deg) {
                        const double sh = r / deg;
                        for (int32_t j = indptr[u]; j < indptr[u + 1]; ++j) {
                            const int32_t v = indices[j];
                            if (nxt[v] == 0.0) touched.push_back(v);
                            nxt[v] += sh;
                            edges.push_back({v, sh});
                        }
                    }
                }
                if (any_d) { if (nxt[src] == 0.0) touched.push_back(src); nxt[src] += dangling; }
                double* o = &acc[t][(size_t)(lvl + 1) * out_stride];
                o[0] += (double)edges.size(); o[1] += (double)touched.size();
                mx_edges = std::max<uint32_t>(mx_edges, (uint32_t)edges.size());
                // sketches
                for (int s = 0; s < n_sizes; ++s) {
                    const uint32_t M = sizes[s];
                    for (auto& e : edges) {
                        const uint32_t fx = (uint32_t)std::ceil(e.second * 2147483648.0);
                        sk1[s][slot_of(h1((uint32_t)e.first), M)] += fx;
                        sk2[s][slot_of(h2((uint32_t)e.first), M)] += fx;
                    }
                }
                fr2.clear();
                uint32_t pushers = 0;
                std::vector<uint32_t> cn1(n_sizes, 0), cn2(n_sizes, 0);
                for (int32_t v : touched) {
                    const double r = nxt[v];
                    const uint32_t deg = (uint32_t)(indptr[v + 1] - indptr[v]);
                    if (deg == 0 || r >= rmax * deg) ++pushers;
                    fr2.push_back({v, r});
                }
                o[2] += pushers;
                for (int s = 0; s < n_sizes; ++s) {
                    const uint32_t M = sizes[s];
                    uint64_t ce1 = 0, ce2 = 0;
                    for (auto& e : edges) {
                        const int32_t v = e.first;
                        const uint32_t deg = (uint32_t)(indptr[v + 1] - indptr[v]);
                        const uint32_t dq = std::min(deg, deg_sat);
                        const double thr = std::floor(rmax * dq * 2147483648.0 * (1.0 - 1.0 / 1048576.0)) - 1.0;
                        const uint32_t a = sk1[s][slot_of(h1((uint32_t)v), M)], b = sk2[s][slot_of(h2((uint32_t)v), M)];
                        if ((double)a >= thr) ++ce1;
                        if ((double)a >= thr && (double)b >= thr) ++ce2;
                    }
                    for (int32_t v : touched) {
                        const uint32_t deg = (uint32_t)(indptr[v + 1] - indptr[v]);
                        const uint32_t dq = std::min(deg, deg_sat);
                        const double thr = std::floor(rmax * dq * 2147483648.0 * (1.0 - 1.0 / 1048576.0)) - 1.0;
                        const uint32_t a = sk1[s][slot_of(h1((uint32_t)v), M)], b = sk2[s][slot_of(h2((uint32_t)v), M)];
                        if ((double)a >= thr) ++cn1[s];
                        if ((double)a >= thr && (double)b >= thr) ++cn2[s];
                    }
                    o[4 + 4 * s + 0] += cn1[s]; o[4 + 4 * s + 1] += (double)ce1; o[4 + 4 * s + 2] += cn2[s]; o[4 + 4 * s + 3] += (double)ce2;
                    mx_c[s] = std::max(mx_c[s], cn1[s]);
                    for (auto& e : edges) { sk1[s][slot_of(h1((uint32_t)e.first), M)] = 0; sk2[s][slot_of(h2((uint32_t)e.first), M)] = 0; }
                }
                for (int32_t v : touched) nxt[v] = 0.0;
                fr.swap(fr2);
            }
            for (int s = 0; s < n_sizes; ++s) row_max_cand[it * n_sizes + s] = mx_c[s];
            row_max_edges[it] = mx_edges;
        }
    }
    std::memset(out, 0, sizeof(double) * (size_t)(L + 1) * out_stride);
    for (int t = 0; t < nth; ++t) for (size_t i = 0; i < (size_t)(L + 1) * out_stride; ++i) out[i] += acc[t][i];
    return 0;
}
