// layout_sim.cpp -- CPU experiment behind the round-5 device layout: what do a row's pushes cost in cache lines under the packed
// CSR (indptr lookup + an unaligned run of column words) and under the SELF-ADDRESSED CSR (rows aligned to 64-B units, the
// column word IS the target's unit number, so a pusher needs no indptr lookup), and how large do the per-row structures of the
// 6-byte log get (pushers per level / per row, edges per level, 256-edge steps per wave).
// Semantics of the row walk follow oracle/gfpush_oracle.cpp (graph.h:73-110).  Build: g++ -O3 -fopenmp -shared -fPIC.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <omp.h>

extern "C" int layout_sim(const int32_t* indptr, const int32_t* indices, int64_t n, const int32_t* seeds, int64_t n_seeds,
                          int L, double rmax, uint32_t unit_words, const int64_t* node_pos /* [n+1], in units */, uint32_t deg_sat,
                          // per row: [0] pushes [1] edges [2] max level pushers [3] max level edges
                          //          [4] 128-B lines packed CSR runs [5] 64-B sectors packed runs [6] indptr 128-B lines (distinct per level)
                          //          [7] 128-B lines aligned [8] 64-B sectors aligned [9] pushers whose packed degree is saturated (exact-degree lookups)
                          //          [10] candidates at saturation (cheap test passes with dq == sat) [11] levels
                          int64_t* row_out, int row_stride,
                          // per level sums: [lvl][0] edges [1] pushers [2] targets
                          double* lvl_out, int lvl_stride)
{
    const int nth = omp_get_max_threads();
    std::vector<std::vector<double>> acc(nth, std::vector<double>((size_t)(L + 1) * lvl_stride, 0.0));
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        std::vector<double> nxt(n, 0.0);
        std::vector<int32_t> touched; touched.reserve(1 << 16);
        std::vector<std::pair<int32_t, double>> fr, fr2;
        std::vector<int64_t> lines;
#pragma omp for schedule(dynamic, 4)
        for (int64_t it = 0; it < n_seeds; ++it) {
            int64_t* ro = row_out + it * row_stride;
            std::memset(ro, 0, sizeof(int64_t) * row_stride);
            fr.clear(); fr.push_back({seeds[it], 1.0});
            for (int lvl = 0; lvl < L; ++lvl) {
                touched.clear(); lines.clear();
                double dangling = 0.0; bool any_d = false;
                int64_t lv_push = 0, lv_edges = 0;
                for (auto& ur : fr) {
                    const int32_t u = ur.first; const double r = ur.second;
                    const uint32_t deg = (uint32_t)(indptr[u + 1] - indptr[u]);
                    const uint32_t dq = std::min(deg, deg_sat);
                    if (dq == deg_sat && r >= rmax * dq) ++ro[10];
                    if (deg == 0) { dangling += r; any_d = true; }
                    else if (r >= rmax * deg) {
                        const double sh = r / deg;
                        ++lv_push; lv_edges += deg;
                        if (deg >= deg_sat) ++ro[9];
                        const int64_t b0 = 4ll * indptr[u], b1 = 4ll * indptr[u + 1] - 1;
                        ro[4] += b1 / 128 - b0 / 128 + 1; ro[5] += b1 / 64 - b0 / 64 + 1;
                        lines.push_back((4ll * u) / 128); if ((4ll * (u + 1)) / 128 != (4ll * u) / 128) lines.push_back((4ll * (u + 1)) / 128);
                        const int64_t a0 = node_pos[u] * unit_words * 4ll, a1 = a0 + 4ll * deg - 1;
                        ro[7] += a1 / 128 - a0 / 128 + 1; ro[8] += a1 / 64 - a0 / 64 + 1;
                        for (int32_t j = indptr[u]; j < indptr[u + 1]; ++j) {
                            const int32_t v = indices[j];
                            if (nxt[v] == 0.0) touched.push_back(v);
                            nxt[v] += sh;
                        }
                    }
                }
                if (any_d) { if (nxt[seeds[it]] == 0.0) touched.push_back(seeds[it]); nxt[seeds[it]] += dangling; }
                std::sort(lines.begin(), lines.end());
                ro[6] += std::unique(lines.begin(), lines.end()) - lines.begin();
                ro[0] += lv_push; ro[1] += lv_edges;
                ro[2] = std::max(ro[2], lv_push); ro[3] = std::max(ro[3], lv_edges);
                if (lv_edges) ro[11] = lvl + 1;
                double* o = &acc[t][(size_t)(lvl + 1) * lvl_stride];
                o[0] += (double)lv_edges; o[1] += (double)lv_push; o[2] += (double)touched.size();
                fr2.clear();
                for (int32_t v : touched) { fr2.push_back({v, nxt[v]}); nxt[v] = 0.0; }
                fr.swap(fr2);
            }
        }
    }
    std::memset(lvl_out, 0, sizeof(double) * (size_t)(L + 1) * lvl_stride);
    for (int t = 0; t < nth; ++t) for (size_t i = 0; i < (size_t)(L + 1) * lvl_stride; ++i) lvl_out[i] += acc[t][i];
    return 0;
}
