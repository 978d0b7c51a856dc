// oracle/gfpush_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of GRAND+'s generalized forward push (GFPush), written from the
// algorithm description in SURVEY.md Appendix A.1 and checked against the compiled
// reference (oracle/_ref, built by oracle/Makefile from /root/reference/precompute/
// propagation.cpp) by tests/golden/make_golden.py and tests/test_oracle.py.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  The product path (grand_plus_amd/csrc/*.hip behind include/grandplus.h)
// never links, loads or calls anything in oracle/.
//
// Reference lines restated by each step are cited as graph.h:NN
// (= /root/reference/precompute/graph.h).
//
// Data-structure class is kept the same as the reference on purpose (one pair of
// std::unordered_map<int,double> per seed, OpenMP schedule(dynamic) over seeds,
// graph.h:73-77) so that timing this file on host cores is a fair "port" CPU baseline.
// Differences from the reference, all deliberate:
//   * residue is read BEFORE the entry is removed (graph.h:86-89 reads after erase: UB);
//   * the level swap is a move, not a deep copy (graph.h:102) -- same values;
//   * the K largest are selected with a total order (value desc, then column asc)
//     and written in that order; the reference uses nth_element with value-only
//     compare (graph.h:49-51,115), so its choice among exact ties is arbitrary;
//   * thread count is an argument (reference hard-codes 40, graph.h:41).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>
#include <omp.h>

extern "C" {

// Per-call totals the bench uses for algorithmic bytes:
//   stats[0] = P  : number of (node, level) pushes           (graph.h:94 taken)
//   stats[1] = E  : sum of deg over those pushes             (graph.h:96 iterations)
//   stats[2] = filled output slots                           (graph.h:121 taken)
//   stats[3] = sum over rows of reserve support size         (graph.h:111 res.size())
//   stats[4] = max over rows of reserve support size
//   stats[5] = max over rows,levels of frontier size         (residue map size)
//   stats[6] = number of dangling returns                    (graph.h:91 taken)
//   stats[7] = sum over rows,levels of frontier size
// next_value (optional, n_seeds doubles): the (K+1)-th largest reserve value of every row (0 when the reserve map holds
// at most K nodes).  The parity comparator uses it to PROVE that a row whose index set differs from the checked
// implementation's holds a tie at the K-th position in the full reserve map (graph.h:115 picks arbitrarily there).
int gfpush_oracle_ex(const int32_t* indptr, int64_t n_nodes, const int32_t* indices,
                     const int32_t* seeds, int64_t n_seeds,
                     const double* coef, int n_coef, double rmax, int K,
                     int32_t* row_idx, int32_t* col_idx, double* value,
                     int n_threads, int64_t* stats, double* next_value)
{
    if (!indptr || !indices || !seeds || !coef || !row_idx || !col_idx || !value) return -1;
    if (n_coef < 1 || K < 1 || n_nodes < 0 || n_seeds < 0) return -2;
    if (n_threads < 1) n_threads = 1;
    for (int64_t it = 0; it < n_seeds; ++it)
        if (seeds[it] < 0 || seeds[it] >= n_nodes) return -3;

    int64_t tP = 0, tE = 0, tF = 0, tS = 0, mS = 0, mFr = 0, tD = 0, tFr = 0;

#pragma omp parallel for schedule(dynamic) num_threads(n_threads) \
        reduction(+:tP,tE,tF,tS,tD,tFr) reduction(max:mS,mFr)
    for (int64_t it = 0; it < n_seeds; ++it) {                       // graph.h:73-74
        const int src = seeds[it];                                   // graph.h:79
        std::unordered_map<int, double> res, rsv;                    // graph.h:76-77
        res[src] = 1.0;                                              // graph.h:81
        rsv[src] = 0.0;                                              // graph.h:82

        for (int lvl = 0; lvl + 1 < n_coef; ++lvl) {                 // graph.h:83
            std::unordered_map<int, double> nxt;                     // graph.h:84
            const double c = coef[lvl];
            if ((int64_t)res.size() > mFr) mFr = (int64_t)res.size();
            tFr += (int64_t)res.size();
            for (const auto& kv : res) {                             // graph.h:85-89 (drain)
                const int u = kv.first;
                const double r = kv.second;
                rsv[u] += c * r;                                     // graph.h:90
                const uint32_t deg = (uint32_t)(indptr[u + 1] - indptr[u]);   // graph.h:43-45
                if (deg == 0) {                                      // graph.h:91-93
                    nxt[src] += r;
                    ++tD;
                } else if (r >= rmax * deg) {                        // graph.h:94
                    const double share = r / deg;                    // graph.h:95
                    for (int32_t j = indptr[u]; j < indptr[u + 1]; ++j)   // graph.h:96-99
                        nxt[indices[j]] += share;
                    ++tP;
                    tE += deg;
                }                                                    // else: r dropped
            }
            res = std::move(nxt);                                    // graph.h:102
        }
        {
            const double c = coef[n_coef - 1];
            if ((int64_t)res.size() > mFr) mFr = (int64_t)res.size();
            tFr += (int64_t)res.size();
            for (const auto& kv : res) rsv[kv.first] += c * kv.second;   // graph.h:104-110
        }

        std::vector<std::pair<int, double>> cand(rsv.begin(), rsv.end());  // graph.h:111
        tS += (int64_t)cand.size();
        if ((int64_t)cand.size() > mS) mS = (int64_t)cand.size();
        const size_t k = cand.size() > (size_t)K ? (size_t)K : cand.size();   // graph.h:113
        auto better = [](const std::pair<int, double>& a, const std::pair<int, double>& b) {
            return a.second > b.second || (a.second == b.second && a.first < b.first);
        };
        std::partial_sort(cand.begin(), cand.begin() + k, cand.end(), better);  // graph.h:115
        if (next_value) {
            double nv = 0.0;
            for (size_t i = k; i < cand.size(); ++i) nv = std::max(nv, cand[i].second);
            next_value[it] = nv;
        }
        for (size_t i = 0; i < k; ++i) {                             // graph.h:117-126
            if (cand[i].second > 0.0) {                              // graph.h:121
                const int64_t slot = it * (int64_t)K + (int64_t)i;
                row_idx[slot] = src;
                col_idx[slot] = cand[i].first;
                value[slot]   = cand[i].second;
                ++tF;
            }
        }
    }
    if (stats) {
        stats[0] = tP; stats[1] = tE; stats[2] = tF; stats[3] = tS;
        stats[4] = mS; stats[5] = mFr; stats[6] = tD; stats[7] = tFr;
    }
    return 0;
}

int gfpush_oracle(const int32_t* indptr, int64_t n_nodes, const int32_t* indices,
                  const int32_t* seeds, int64_t n_seeds,
                  const double* coef, int n_coef, double rmax, int K,
                  int32_t* row_idx, int32_t* col_idx, double* value,
                  int n_threads, int64_t* stats)
{
    return gfpush_oracle_ex(indptr, n_nodes, indices, seeds, n_seeds, coef, n_coef, rmax, K, row_idx, col_idx, value,
                            n_threads, stats, nullptr);
}

int gfpush_oracle_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
