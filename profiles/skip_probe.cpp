// CPU feasibility probe (not product, not test): how many (query, iteration) pairs of a
// registration can keep the previous iteration's nearest neighbour WITHOUT a search, exactly.
//
// Rule probed (the one k_icp implements, kernels.hip "margin budget"): after a full search a query
// holds m = (sqrt(second) - sqrt(best)) / (2 sqrt(max(1, th))) where best / second are the smallest
// and second smallest semantically scaled squared distances in its 27-voxel neighbourhood.  Moving
// the query by e changes sqrt(scaled distance) of any candidate by at most sqrt(max(1, th)) e, so
// while the query stays in its home voxel (same candidate set) and the movement it accumulated
// since the search stays below m, the winner is the same point.  Every iteration's movement is
// bounded per query by a |s|_1 + b with (a, b) = (rotation angle, |translation|) of that
// iteration's estimate.  The probe runs the oracle's loop, applies the rule, and CHECKS every
// skipped query against a full search.
//
// Build: g++ -O3 -std=c++17 -fPIC -fopenmp -ffp-contract=off -shared -o /tmp/libskip_probe.so profiles/skip_probe.cpp
#include "../oracle/sage_oracle.cpp"

namespace {
struct Best2 {
    bool found;
    Vec4 nn;
    double best, second;
};
inline Best2 search2(const Map &m, const Vec4 &point, double th) {
    Best2 r{false, {}, std::numeric_limits<double>::max(), std::numeric_limits<double>::max()};
    const int kx = static_cast<int>(point[0] / m.voxel_size);
    const int ky = static_cast<int>(point[1] / m.voxel_size);
    const int kz = static_cast<int>(point[2] / m.voxel_size);
    for (int i = kx - 1; i <= kx + 1; ++i)
        for (int j = ky - 1; j <= ky + 1; ++j)
            for (int k = kz - 1; k <= kz + 1; ++k) {
                auto it = m.map.find(Voxel{i, j, k});
                if (it == m.map.end()) continue;
                for (const auto &nb : it->second.points) {
                    const double dx = nb[0] - point[0], dy = nb[1] - point[1], dz = nb[2] - point[2];
                    double d = SAGE_SQNORM3_NN(dx * dx, dy * dy, dz * dz);
                    if (static_cast<int>(nb[3]) == static_cast<int>(point[3]) ||
                        static_cast<int>(nb[3] * point[3]) == 0)
                        d = d * th;
                    if (d < r.best) {
                        r.second = r.best;
                        r.best = d;
                        r.nn = nb;
                        r.found = true;
                    } else if (d < r.second) {
                        r.second = d;
                    }
                }
            }
    return r;
}
}  // namespace

extern "C" int probe_skip(const void *h, const double *frame, uint64_t n, const double init[7], double max_dist,
                          double kernel, double sem_th, int nthreads, int check, uint64_t *searched /*[500]*/,
                          double *step /*[500]*/, uint64_t *mismatches, double T_out[7]) {
    const Map &m = *static_cast<const Map *>(h);
    nthreads = resolve_threads(nthreads);
    std::vector<double> source(frame, frame + 4 * n);
    sgo_transform_points(init, source.data(), n);
    double T_icp[7] = {0, 0, 0, 1, 0, 0, 0};
    std::vector<double> src(4 * n), tgt(4 * n);
    std::vector<double> budget(n, -1.0);
    std::vector<Vec4> win(n);
    std::vector<char> has(n, 0);
    std::vector<int> home(3 * n, INT32_MIN);
    std::vector<double> spos(3 * n, 0.0), marg(n, -1.0);
    const bool exact_disp = std::getenv("SKIP_EXACT") && std::atoi(std::getenv("SKIP_EXACT"));
    const double cap = std::getenv("SKIP_CAP") ? std::atof(std::getenv("SKIP_CAP")) : 1e300;    // margin cap M (metres)
    const double cm = std::sqrt(std::max(1.0, sem_th));
    double a_k = 0, b_k = 0;           // movement bound of the last estimate: a |s|_1 + b
    *mismatches = 0;
    int it = 0;
    for (; it < 500; ++it) {
        uint64_t nsearch = 0, bad = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : nsearch, bad) schedule(static)
        for (int64_t i = 0; i < static_cast<int64_t>(n); ++i) {
            const Vec4 p = {source[4 * i], source[4 * i + 1], source[4 * i + 2], source[4 * i + 3]};
            const int kx = static_cast<int>(p[0] / m.voxel_size), ky = static_cast<int>(p[1] / m.voxel_size),
                      kz = static_cast<int>(p[2] / m.voxel_size);
            const bool same_home = kx == home[3 * i] && ky == home[3 * i + 1] && kz == home[3 * i + 2];
            const double l1 = std::fabs(p[0]) + std::fabs(p[1]) + std::fabs(p[2]);
            double bud = budget[i] - (a_k * l1 + b_k) * (1.0 + 1e-6) - 1e-12 * (1.0 + l1);
            bool skip = same_home && has[i] && bud > 0.0;
            if (exact_disp) {
                const double ex = p[0] - spos[3 * i], ey = p[1] - spos[3 * i + 1], ez = p[2] - spos[3 * i + 2];
                const double e = std::sqrt(ex * ex + ey * ey + ez * ez) * (1.0 + 1e-9) + 1e-12 * (1.0 + l1);
                skip = same_home && has[i] && e < marg[i];
            }
            if (!skip) {
                ++nsearch;
                const Best2 r = search2(m, p, sem_th);
                has[i] = r.found;
                win[i] = r.nn;
                home[3 * i] = kx; home[3 * i + 1] = ky; home[3 * i + 2] = kz;
                if (r.found) {
                    const double sb = std::sqrt(r.best), ss = r.second == std::numeric_limits<double>::max()
                                                                    ? std::numeric_limits<double>::max()
                                                                    : std::sqrt(r.second);
                    bud = std::min(cap, (ss - sb) / (2.0 * cm)) * (1.0 - 1e-6);
                    marg[i] = bud;
                    spos[3 * i] = p[0]; spos[3 * i + 1] = p[1]; spos[3 * i + 2] = p[2];
                } else {
                    bud = -1.0;       // nothing to keep: search again
                    marg[i] = -1.0;
                }
            } else if (check) {
                const Best2 r = search2(m, p, sem_th);
                if (!r.found || std::memcmp(r.nn.data(), win[i].data(), 32) != 0) ++bad;
            }
            budget[i] = bud;
        }
        searched[it] = nsearch;
        *mismatches += bad;
        // correspondences in query order (serial join)
        uint64_t nc = 0;
        for (uint64_t i = 0; i < n; ++i) {
            if (!has[i]) continue;
            const double dx = win[i][0] - source[4 * i], dy = win[i][1] - source[4 * i + 1],
                         dz = win[i][2] - source[4 * i + 2];
            if (std::sqrt(SAGE_SQNORM3_ACCEPT(dx * dx, dy * dy, dz * dz)) < max_dist) {
                std::memcpy(&src[4 * nc], &source[4 * i], 32);
                std::memcpy(&tgt[4 * nc], win[i].data(), 32);
                ++nc;
            }
        }
        double est[7];
        sgo_align_clouds(src.data(), tgt.data(), nc, kernel, est, nullptr, nullptr, nthreads);
        sgo_transform_points(est, source.data(), n);
        double tmp[7];
        se3_mul(est, T_icp, tmp);
        std::memcpy(T_icp, tmp, 56);
        double lg[6];
        se3_log(est, lg);
        double nrm = 0;
        for (int i = 0; i < 6; ++i) nrm += lg[i] * lg[i];
        nrm = std::sqrt(nrm);
        step[it] = nrm;
        a_k = std::sqrt(lg[3] * lg[3] + lg[4] * lg[4] + lg[5] * lg[5]) * (1.0 + 1e-9);
        b_k = std::sqrt(est[4] * est[4] + est[5] * est[5] + est[6] * est[6]) * (1.0 + 1e-9);
        if (nrm < 1e-4) { ++it; break; }
    }
    se3_mul(T_icp, init, T_out);
    return it;
}
