// ============================================================================
// oracle/flo_oracle.cpp  --  TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// CPU restatement of the reference's five RegistrationInterface plug-ins
// (include/registration/registration_interface.h:11-20):
//   P2PlaneIvox   <- LoamPointToPlaneIVOX<double>   loam_point_to_plane_ivox.h:30-355
//   IcpOptimized  <- IcpOptimized<double>           icp_optimized.h:15-250
//   IncNdt        <- IncrementalNDT                 incremental_ndt.h:16-383
//   LoamFull      <- LoamFull<double>               loam_full_kdtree.h:24-420
//   P2PlaneKd     <- LoamPointToPlaneKdtree<double> loam_point_to_plane_kdtree.h:25-322
// Every quirk Q1..Q14 of SURVEY.md 8a is reproduced on purpose (plus Q15: the
// iVox query keeps a point's previous neighbour list when no candidate is found,
// ivox_map.cpp:21-23 vs :32).  Per-point stage is OpenMP-parallel (reference:
// std::execution::par_unseq), the 6x6 reduction is the reference's sequential
// index-order loop.  Function-static state of the reference (is_first, last_T;
// Q12) is per-instance here: parity for a job is against a fresh process.
//
// Pinned by the reference's own code: oracle/ref_shim compiles the reference's registration / iVox / LOAM sources
// verbatim against an include-shadow shim (oracle/_ref/libref.so) and tests/test_ref_pin.py compares this file with
// it over multi-scan replays of every kind -- exact on every integer / float-only output, <= 2e-14 on FP64 ones.
// What stays unpinned is third-party arithmetic only (Eigen's association order, PCL / FLANN internals: restated in
// flo_linalg.h / flo_kdtree.h / flo_common.h and SHARED with the shim).  SO3Hat/SO3Exp/RPY are additionally pinned by the
// reference's known-answer tests (test/math_function_ut.cpp:9-133,160-192; tests/test_oracle_so3.py).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use this.
// ============================================================================
#include "flo_api.h"
#include "flo_common.h"
#include "flo_features.h"
#include "flo_loop.h"
#include <deque>
#include <set>
#include <memory>
#include <numeric>
#include <cstdio>
#include <omp.h>

namespace flo {

static int g_threads = 0;  // 0 = OpenMP default

struct IterLog { double T[16]; int n_valid; double sum_res; };

struct MatcherBase {
    flo_params p{};
    flo_stats stats{};
    flo_counters counters{};
    std::vector<IterLog> log;
    double last_H[36]{}, last_g[6]{};
    bool instrument = true;  // traffic / tie counters of the kNN stage (flo_set_instrumentation: off for timing runs)
    // per query: the neighbour list it holds now was written by a search that met an exact distance tie (nearest two, or across the K / K + 1
    // boundary) -- the rows whose slot order / membership libstdc++'s introselect decides.  Sticky like the list itself (Q15).  Kinds
    // without tie instrumentation leave it empty.
    std::vector<uint8_t> tie_flag;
    virtual ~MatcherBase() = default;
    virtual int AddCloud(const Cloud& c0, const Cloud& c1) = 0;
    virtual bool Match(const Cloud& s0, const Cloud& s1, double* T, bool update_map) = 0;
    virtual float Fitness(float max_range) const = 0;
    virtual int GetCorr(int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) const = 0;
    virtual size_t MapSize(int) const { return 0; }
    virtual size_t MapDump(int, float*, size_t) const { return 0; }
    void Log(const double* T, int nv, double sr) {
        IterLog l;
        std::memcpy(l.T, T, sizeof(l.T));
        l.n_valid = nv;
        l.sum_res = sr;
        log.push_back(l);
    }
};

// ---- shared: point-to-plane residual (C.1) on 5 neighbours ----------------------------------
// loam_point_to_plane_ivox.h:275-321 == loam_full_kdtree.h:296-342 == loam_point_to_plane_kdtree.h:221-271
// returns false on any early "return" of the reference lambda.
static inline bool plane_residual(const P4 nn[5], const P4& src, const P4& src_t, const double* T, double thres,
                                  double J[6], double& d_abs) {
    double A[15];  // 5x3 col-major
    for (int j = 0; j < 5; ++j) { A[j + 0 * 5] = nn[j].x; A[j + 1 * 5] = nn[j].y; A[j + 2 * 5] = nn[j].z; }
    const double b[5] = {-1.0, -1.0, -1.0, -1.0, -1.0};
    double x[3];
    colpiv_qr_solve<5, 3>(A, b, x);
    const double nrm = norm3(x);
    for (int j = 0; j < 5; ++j) {
        const double r = ((A[j] * x[0] + A[j + 5] * x[1]) + A[j + 10] * x[2]) + 1.0;
        if (std::abs(r) / nrm > thres) return false;
    }
    const double n[3] = {x[0] / nrm, x[1] / nrm, x[2] / nrm};
    const double ps[3] = {double(src.x), double(src.y), double(src.z)};
    const double pt[3] = {double(src_t.x), double(src_t.y), double(src_t.z)};
    const double d = ((pt[0] - A[0]) * n[0] + (pt[1] - A[5]) * n[1]) + (pt[2] - A[10]) * n[2];
    if (norm3(ps) < 81 * d * d) return false;
    const double s = d > 0 ? 1.0 : -1.0;
    // v = R * ps
    const double v[3] = {(T[0] * ps[0] + T[4] * ps[1]) + T[8] * ps[2], (T[1] * ps[0] + T[5] * ps[1]) + T[9] * ps[2],
                         (T[2] * ps[0] + T[6] * ps[1]) + T[10] * ps[2]};
    // -SO3Hat(v).transpose() * n * s  ==  hat(v) * n * s
    J[0] = ((0.0 * n[0] + (-v[2]) * n[1]) + v[1] * n[2]) * s;
    J[1] = ((v[2] * n[0] + 0.0 * n[1]) + (-v[0]) * n[2]) * s;
    J[2] = (((-v[1]) * n[0] + v[0] * n[1]) + 0.0 * n[2]) * s;
    J[3] = n[0] * s;
    J[4] = n[1] * s;
    J[5] = n[2] * s;
    d_abs = std::fabs(d);
    return true;
}

struct PerPoint {  // H_i, g_i, res_i of the reference's per-point vectors
    double H[36];
    double g[6];
    double res;
};
static inline void fill_rank1(PerPoint& pp, const double J[6], double r) {
    for (int b = 0; b < 6; ++b)
        for (int a = 0; a < 6; ++a) pp.H[a + b * 6] = J[a] * J[b];
    for (int a = 0; a < 6; ++a) pp.g[a] = (-J[a]) * r;
    pp.res = r;
}

// LOAM-family Gauss-Newton tail: dx = fullPivQR(H).solve(g); left update; stop rule.
// loam_point_to_plane_ivox.h:167-195 == loam_full_kdtree.h:141-175 == loam_point_to_plane_kdtree.h:108-135
struct LoamLoopState { double last_rot = 0.0, last_pos = 0.0; };
static inline bool loam_update(double* T, const double* H, const double* g, double rot_thr, double pos_thr,
                               LoamLoopState& st, double* dx_out) {
    double dx[6];
    fullpiv_qr_solve<6>(H, g, dx);
    double Rd[9], R[9], Rn[9];
    so3_exp(dx, Rd);
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) R[i + j * 3] = T[i + j * 4];
    mat3_mul(Rd, R, Rn);
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) T[i + j * 4] = Rn[i + j * 3];
    T[12] += dx[3]; T[13] += dx[4]; T[14] += dx[5];
    const double rn = norm3(dx), pn = norm3(dx + 3);
    const double drot = std::fabs(rn - st.last_rot), dpos = std::fabs(pn - st.last_pos);
    st.last_rot = rn; st.last_pos = pn;
    std::memcpy(dx_out, dx, sizeof(dx));
    return (rn < rot_thr && pn < pos_thr) || (drot < 1.0e-4 && dpos < 1.0e-4);
}

// keyframe gate: IsNeedAddCloud  icp_optimized.h:218-234, loam_full_kdtree.h:374-389,
// loam_point_to_plane_kdtree.h:185-202   (function-static last_T -> per-instance, Q12)
struct KeyframeGate {
    bool init = false;
    double last_T[16];
    bool Need(const double* T, double dist_thr, double rot_thr) {
        if (!init) { std::memcpy(last_T, T, sizeof(last_T)); init = true; }
        double Rl[9], Rc[9], Rli[9], Rd[9], rpy[3];
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) { Rl[i + j * 3] = last_T[i + j * 4]; Rc[i + j * 3] = T[i + j * 4]; }
        inverse3(Rl, Rli);
        mat3_mul(Rli, Rc, Rd);
        rotation_to_rpy(Rd, rpy);
        const double dt[3] = {T[12] - last_T[12], T[13] - last_T[13], T[14] - last_T[14]};
        if (norm3(dt) > dist_thr || std::fabs(rpy[0]) > rot_thr || std::fabs(rpy[1]) > rot_thr ||
            std::fabs(rpy[2]) > rot_thr) {
            std::memcpy(last_T, T, sizeof(last_T));
            return true;
        }
        return false;
    }
};

// GetFitnessScore loops (icp_optimized.h:191-215 etc.): float transform, 1-NN, mean of d2 <= max_range
static float fitness_score(const Cloud& src, const double* T, const KdTree& tree, float max_range) {
    float score = 0.0f;
    int nr = 0;
    const Cloud tc = transform_cloud_f(src, T);
    for (size_t i = 0; i < tc.size(); ++i) {
        KdTree::Hit h;
        const float q[3] = {tc[i].x, tc[i].y, tc[i].z};
        if (tree.Knn(q, 1, &h) < 1) continue;
        if (h.d2 <= max_range) { score += h.d2; nr++; }
    }
    return nr > 0 ? score / float(nr) : std::numeric_limits<float>::max();
}

static std::vector<float> flat_xyz(const Cloud& c) {
    std::vector<float> v(c.size() * 3);
    for (size_t i = 0; i < c.size(); ++i) { v[3 * i] = c[i].x; v[3 * i + 1] = c[i].y; v[3 * i + 2] = c[i].z; }
    return v;
}

// =============================================================================================
// LoamPointToPlaneIVOX<double>
// =============================================================================================
struct P2PlaneIvox final : MatcherBase {
    std::unique_ptr<IVoxMap> ivox;
    bool is_first = true;  // loam_point_to_plane_ivox.h:62 (static there)
    double filter_size_map_min = 0.5;  // :351
    std::vector<PerPoint> coeffs;
    std::vector<std::vector<Near>> nearest_points;  // persists across iterations AND Match calls (:257)
    std::vector<char> flags;
    size_t number_planar_point = 0;
    double T_[16];
    double final_T[16];
    Cloud source_copy;
    KdTree fitness_tree;

    P2PlaneIvox() { InitIVox(); }
    void InitIVox() { ivox.reset(new IVoxMap(0.5f, 18, 1000000)); }  // :53-58

    int AddCloud(const Cloud& planar_cloud, const Cloud&) override {  // :60-139
        if (p.is_localization_mode) { is_first = true; InitIVox(); }
        if (is_first) {
            ivox->AddPoints(planar_cloud);
            is_first = false;
        } else {
            Cloud points_to_add, no_downsample;
            const double half = 0.5 * filter_size_map_min;
            for (size_t i = 0; i < number_planar_point && i < planar_cloud.size(); ++i) {
                const P4 pw = transform_point_d(planar_cloud[i], T_);
                if (!nearest_points[i].empty()) {
                    const auto& near = nearest_points[i];
                    const double c[3] = {(std::floor(double(pw.x) / filter_size_map_min) + 0.5) * filter_size_map_min,
                                         (std::floor(double(pw.y) / filter_size_map_min) + 0.5) * filter_size_map_min,
                                         (std::floor(double(pw.z) / filter_size_map_min) + 0.5) * filter_size_map_min};
                    const double d2c[3] = {double(near[0].pt.x) - c[0], double(near[0].pt.y) - c[1], double(near[0].pt.z) - c[2]};
                    if (std::fabs(d2c[0]) > half && std::fabs(d2c[1]) > half && std::fabs(d2c[2]) > half) {
                        no_downsample.push_back(pw);
                        continue;
                    }
                    bool need_add = true;
                    const double e[3] = {double(pw.x) - c[0], double(pw.y) - c[1], double(pw.z) - c[2]};
                    const double dist = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
                    if (near.size() >= 5u) {
                        for (int k = 0; k < 5; ++k) {
                            const double f[3] = {double(near[k].pt.x) - c[0], double(near[k].pt.y) - c[1], double(near[k].pt.z) - c[2]};
                            if ((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2] < dist + 1.0e-6) { need_add = false; break; }
                        }
                    }
                    if (need_add) points_to_add.push_back(pw);
                } else {
                    points_to_add.push_back(pw);
                }
            }
            ivox->AddPoints(points_to_add);
            ivox->AddPoints(no_downsample);
        }
        if (p.is_localization_mode) {
            const auto f = flat_xyz(planar_cloud);
            fitness_tree.Build(f.data(), planar_cloud.size(), 3);
        }
        return 0;
    }

    void PlanerMatch(const Cloud& source) {  // :256-324
        nearest_points.resize(number_planar_point);
        tie_flag.resize(number_planar_point, 0);
        const long n = long(number_planar_point);
        uint64_t probes = 0, hits = 0, cand = 0, ties = 0;
#pragma omp parallel for schedule(static) reduction(+ : probes, hits, cand, ties)
        for (long i = 0; i < n; ++i) {
            const P4& sp = source[size_t(i)];
            const P4 tp = transform_point_d(sp, T_);
            std::vector<Near>& pv = nearest_points[size_t(i)];
            KnnCounters kc;
            const bool rewritten = ivox->GetClosestPoint(tp, pv, instrument ? &kc : nullptr, 5);
            probes += kc.probes; hits += kc.hits; cand += kc.cand; ties += kc.ties;
            if (instrument && rewritten) tie_flag[size_t(i)] = kc.ties ? 1 : 0;
            if (pv.size() < 5) continue;
            P4 nn[5];
            for (int j = 0; j < 5; ++j) nn[j] = pv[size_t(j)].pt;
            double J[6], r;
            if (!plane_residual(nn, sp, tp, T_, p.point_to_planar_thres, J, r)) continue;
            flags[size_t(i)] = 1;
            fill_rank1(coeffs[size_t(i)], J, r);
        }
        counters.probes += probes; counters.hit_voxels += hits; counters.cand_points += cand;
        counters.tie_queries += ties; counters.point_iters += uint64_t(n);
    }

    bool Match(const Cloud& planar_source, const Cloud&, double* T, bool update_map) override {  // :141-216
        source_copy = planar_source;
        const size_t planar_size = planar_source.size();
        number_planar_point = planar_size;
        coeffs.resize(planar_size);
        flags.assign(planar_size, 0);
        std::memcpy(T_, T, sizeof(T_));
        bool has_converge = true;
        LoamLoopState st;
        size_t n_valid = 0;
        double overall_res = 0.0;
        log.clear();
        counters = flo_counters{};
        stats = flo_stats{};
        for (unsigned it = 0; it < p.max_iterations; ++it) {
            double H[36] = {0}, g[6] = {0};
            PlanerMatch(planar_source);
            // SumCoefficient :326-340
            n_valid = 0; overall_res = 0.0;
            for (size_t i = 0; i < number_planar_point; ++i) {
                if (!flags[i]) continue;
                n_valid++;
                for (int k = 0; k < 36; ++k) H[k] += coeffs[i].H[k];
                for (int k = 0; k < 6; ++k) g[k] += coeffs[i].g[k];
                overall_res += coeffs[i].res;
            }
            std::memcpy(last_H, H, sizeof(H)); std::memcpy(last_g, g, sizeof(g));
            const bool stop = loam_update(T_, H, g, p.rotation_converge_thres, p.position_converge_thres, st, stats.last_dx);
            stats.iterations = int(it) + 1;
            Log(T_, int(n_valid), overall_res);
            if (stop) break;
        }
        std::memcpy(T, T_, sizeof(T_));
        std::memcpy(final_T, T_, sizeof(T_));
        if (n_valid < 50u) has_converge = false;
        stats.n_valid = int(n_valid); stats.sum_res = overall_res; stats.n_source = int(planar_size);
        stats.converged = has_converge ? 1 : 0;
        if (has_converge && !p.is_localization_mode && update_map) { AddCloud(planar_source, Cloud()); stats.map_updated = 1; }
        return has_converge;
    }

    float Fitness(float max_range) const override {  // :225-253
        if (!p.is_localization_mode) return std::numeric_limits<float>::max();
        return fitness_score(source_copy, final_T, fitness_tree, max_range);
    }
    int GetCorr(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) const override {
        const size_t n = std::min(cap, number_planar_point);
        for (size_t i = 0; i < n; ++i) {
            const auto& pv = nearest_points[i];
            cnt[i] = uint8_t(pv.size());
            for (size_t j = 0; j < 5; ++j) ids[i * 5 + j] = j < pv.size() ? pv[j].gid : -1;
            valid[i] = uint8_t(flags[i]);
        }
        return int(n);
    }
    size_t MapSize(int) const override { return ivox->NumPoints(); }
    size_t MapDump(int, float* xyz, size_t cap) const override {
        // (gid, point) pairs in gid order; evicted gids are simply absent -> dump compacts,
        // so only meaningful while nothing was evicted (tests check next_gid_ == NumPoints()).
        std::vector<std::pair<int, P4>> all;
        for (auto& kv : ivox->grids_cache_)
            for (size_t k = 0; k < kv.second.points_.size(); ++k) all.push_back({kv.second.gids_[k], kv.second.points_[k]});
        std::sort(all.begin(), all.end(), [](auto& a, auto& b) { return a.first < b.first; });
        const size_t n = std::min(cap, all.size());
        for (size_t i = 0; i < n; ++i) { xyz[3 * i] = all[i].second.x; xyz[3 * i + 1] = all[i].second.y; xyz[3 * i + 2] = all[i].second.z; }
        return all.size();
    }
};

// =============================================================================================
// IcpOptimized<double>
// =============================================================================================
struct IcpOptimized final : MatcherBase {
    Cloud local_map, source;
    std::deque<Cloud> cloud_deque;
    KdTree tree;
    KeyframeGate gate;
    double final_T[16];
    std::vector<int> nn_idx;
    std::vector<char> eff;

    int AddCloud(const Cloud& new_cloud, const Cloud&) override {  // icp_optimized.h:165-189
        if (p.is_localization_mode) {
            local_map = new_cloud;
        } else {
            cloud_deque.push_back(new_cloud);
            if (cloud_deque.size() > p.local_map_size) cloud_deque.pop_front();
            local_map.clear();
            for (const auto& c : cloud_deque) local_map.insert(local_map.end(), c.begin(), c.end());  // (:181-184: down-sampled copy unused)
        }
        local_map = voxel_grid(local_map, p.map_cloud_filter_size);
        const auto f = flat_xyz(local_map);
        tree.Build(f.data(), local_map.size(), 3);
        return 0;
    }

    bool Match(const Cloud& ordered, const Cloud&, double* T, bool update_map) override {  // :54-163
        bool has_converge = false;
        source = voxel_grid(ordered, p.source_cloud_filter_size);
        double Tt[16];
        std::memcpy(Tt, T, sizeof(Tt));
        log.clear(); counters = flo_counters{}; stats = flo_stats{};
        const size_t n = source.size();
        nn_idx.assign(n, -1);
        for (unsigned it = 0; it < p.max_iterations; ++it) {
            const Cloud tc = transform_cloud_f(source, Tt);
            std::vector<PerPoint> all(n);           // per-iteration allocation + zero fill mirrors :70-73
            std::vector<double> err(n * 3, 0.0);
            eff.assign(n, 0);
            std::fill(nn_idx.begin(), nn_idx.end(), -1);
            for (auto& a : all) std::memset(&a, 0, sizeof(a));
            const long ln = long(n);
#pragma omp parallel for schedule(static)
            for (long i = 0; i < ln; ++i) {
                const P4& op = source[size_t(i)];
                const P4& tp = tc[size_t(i)];
                KdTree::Hit h;
                const float q[3] = {tp.x, tp.y, tp.z};
                if (tree.Knn(q, 1, &h) < 1) continue;
                if (double(h.d2) > p.point_search_thres) continue;
                nn_idx[size_t(i)] = h.idx;  // introspection only: ids are reported for accepted correspondences
                const float* mp = tree.point(h.idx);
                const double e[3] = {double(tp.x) - double(mp[0]), double(tp.y) - double(mp[1]), double(tp.z) - double(mp[2])};
                const double o[3] = {double(op.x), double(op.y), double(op.z)};
                double R[9], hat[9], RH[9];
                for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) R[r + j * 3] = Tt[r + j * 4];
                so3_hat(o, hat);
                mat3_mul(R, hat, RH);
                double J[18];  // 3x6 col-major
                for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) { J[r + j * 3] = (r == j) ? 1.0 : 0.0; J[r + (j + 3) * 3] = -RH[r + j * 3]; }
                PerPoint& pp = all[size_t(i)];
                for (int b = 0; b < 6; ++b)
                    for (int a = 0; a < 6; ++a)
                        pp.H[a + b * 6] = (J[0 + a * 3] * J[0 + b * 3] + J[1 + a * 3] * J[1 + b * 3]) + J[2 + a * 3] * J[2 + b * 3];
                for (int a = 0; a < 6; ++a)
                    pp.g[a] = ((-J[0 + a * 3]) * e[0] + (-J[1 + a * 3]) * e[1]) + (-J[2 + a * 3]) * e[2];
                eff[size_t(i)] = 1;
                err[3 * size_t(i)] = e[0]; err[3 * size_t(i) + 1] = e[1]; err[3 * size_t(i) + 2] = e[2];
            }
            counters.point_iters += n;
            double H[36] = {0}, B[6] = {0}, total_res = 0.0;
            int effective = 0;
            for (size_t i = 0; i < n; ++i) {
                if (!eff[i]) continue;
                for (int k = 0; k < 36; ++k) H[k] += all[i].H[k];
                for (int k = 0; k < 6; ++k) B[k] += all[i].g[k];
                effective++;
                total_res += norm3(&err[3 * i]);
            }
            std::memcpy(last_H, H, sizeof(H)); std::memcpy(last_g, B, sizeof(B));
            stats.iterations = int(it) + 1; stats.n_valid = effective; stats.sum_res = total_res;
            PartialPivLU<6> lu;
            lu.compute(H);
            if (lu.determinant() == 0) { Log(Tt, effective, total_res); continue; }
            double inv[36], dx[6];
            lu.inverse(inv);
            for (int a = 0; a < 6; ++a) { double s = 0.0; for (int k = 0; k < 6; ++k) s += inv[a + k * 6] * B[k]; dx[a] = s; }
            Tt[12] += dx[0]; Tt[13] += dx[1]; Tt[14] += dx[2];
            double Rd[9], R[9], Rn[9];
            so3_exp(dx + 3, Rd);
            for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) R[r + j * 3] = Tt[r + j * 4];
            mat3_mul(R, Rd, Rn);
            for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) Tt[r + j * 4] = Rn[r + j * 3];
            std::memcpy(stats.last_dx, dx, sizeof(dx));
            Log(Tt, effective, total_res);
            if (norm3(dx + 3) < p.rotation_converge_thres && norm3(dx) < p.position_converge_thres) { has_converge = true; break; }
        }
        std::memcpy(final_T, Tt, sizeof(Tt));
        std::memcpy(T, Tt, sizeof(Tt));
        stats.n_source = int(n); stats.converged = has_converge ? 1 : 0;
        if (has_converge && gate.Need(final_T, p.dist_thre_add_cloud, p.rot_thre_add_cloud) && !p.is_localization_mode && update_map) {
            AddCloud(transform_cloud_f(source, final_T), Cloud());
            stats.map_updated = 1;
        }
        return has_converge;
    }
    float Fitness(float max_range) const override { return fitness_score(source, final_T, tree, max_range); }
    int GetCorr(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) const override {
        const size_t n = std::min(cap, nn_idx.size());
        for (size_t i = 0; i < n; ++i) { ids[i] = nn_idx[i]; cnt[i] = nn_idx[i] >= 0 ? 1 : 0; valid[i] = uint8_t(eff[i]); }
        return int(n);
    }
    size_t MapSize(int) const override { return local_map.size(); }
    size_t MapDump(int, float* xyz, size_t cap) const override {
        const size_t n = std::min(cap, local_map.size());
        for (size_t i = 0; i < n; ++i) { xyz[3 * i] = local_map[i].x; xyz[3 * i + 1] = local_map[i].y; xyz[3 * i + 2] = local_map[i].z; }
        return local_map.size();
    }
};

// =============================================================================================
// IncrementalNDT
// =============================================================================================
struct IncNdt final : MatcherBase {
    struct VoxelData {  // incremental_ndt.h:65-87
        std::vector<double> points;  // xyz triples
        double mu[3] = {0, 0, 0};
        double sigma[9] = {0};
        double info[9] = {0};
        bool estimated = false;
        int num_points = 0;
        int vid = 0;  // oracle-only: creation id
    };
    using KD = std::pair<Key3, VoxelData>;
    std::list<KD> data;
    std::unordered_map<Key3, std::list<KD>::iterator, SpatialHash> grids;
    bool flag_first_scan = true;
    Cloud source;
    KdTree fitness_tree;
    double final_T[16];
    int next_vid = 0;
    std::vector<int> hit_vid;  // n x 7 (voxel creation id or -1) for the last iteration
    std::vector<char> eff7;
    const Key3 nearby[7] = {{0, 0, 0}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, -1}, {0, 0, 1}};  // :122-127

    static void MeanCov(const std::vector<double>& pts, double* mean, double* cov) {  // :91-110
        const size_t len = pts.size() / 3;
        double s[3] = {0, 0, 0};
        for (size_t k = 0; k < len; ++k) { s[0] = s[0] + pts[3 * k]; s[1] = s[1] + pts[3 * k + 1]; s[2] = s[2] + pts[3 * k + 2]; }
        for (int a = 0; a < 3; ++a) mean[a] = s[a] / double(len);
        double c[9] = {0};
        for (size_t k = 0; k < len; ++k) {
            const double v[3] = {pts[3 * k] - mean[0], pts[3 * k + 1] - mean[1], pts[3 * k + 2] - mean[2]};
            for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) c[i + j * 3] = c[i + j * 3] + v[i] * v[j];
        }
        for (int k = 0; k < 9; ++k) cov[k] = c[k] / double(len - 1);
    }
    void UpdateVoxel(VoxelData& v) const {  // :130-179
        const double minp = p.ndt_min_points_in_voxel, maxp = p.ndt_max_points_in_voxel;
        if (flag_first_scan) {
            if (v.points.size() / 3 > 1u) {
                MeanCov(v.points, v.mu, v.sigma);
                double m[9];
                for (int k = 0; k < 9; ++k) m[k] = v.sigma[k] + ((k % 4 == 0) ? 1.0 : 0.0) * 1.0e-3;
                inverse3(m, v.info);
            } else {
                v.mu[0] = v.points[0]; v.mu[1] = v.points[1]; v.mu[2] = v.points[2];
                for (int k = 0; k < 9; ++k) v.info[k] = ((k % 4 == 0) ? 1.0 : 0.0) * 1.0e2;
            }
            v.estimated = true;
            v.points.clear();
            return;
        }
        if (v.estimated && v.num_points > int(maxp)) return;
        const int npts = int(v.points.size() / 3);
        if (!v.estimated && npts > int(minp)) {
            MeanCov(v.points, v.mu, v.sigma);
            double m[9];
            for (int k = 0; k < 9; ++k) m[k] = v.sigma[k] + ((k % 4 == 0) ? 1.0 : 0.0) * 1e-3;
            inverse3(m, v.info);
            v.estimated = true;
            v.points.clear();
        } else if (v.estimated && npts > int(minp)) {
            double cur_mu[3], cur_var[9], new_mu[3], new_var[9];
            MeanCov(v.points, cur_mu, cur_var);
            // UpdateMeanAndCov :112-120
            const int hm = v.num_points, cn = npts;
            for (int a = 0; a < 3; ++a) new_mu[a] = (double(hm) * v.mu[a] + double(cn) * cur_mu[a]) / double(hm + cn);
            const double dh[3] = {v.mu[0] - new_mu[0], v.mu[1] - new_mu[1], v.mu[2] - new_mu[2]};
            const double dc[3] = {cur_mu[0] - new_mu[0], cur_mu[1] - new_mu[1], cur_mu[2] - new_mu[2]};
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i)
                    new_var[i + j * 3] = (double(hm) * (v.sigma[i + j * 3] + dh[i] * dh[j]) + double(cn) * (cur_var[i + j * 3] + dc[i] * dc[j])) / double(hm + cn);
            std::memcpy(v.mu, new_mu, sizeof(new_mu));
            std::memcpy(v.sigma, new_var, sizeof(new_var));
            v.num_points += npts;
            v.points.clear();
            double U[9], S[3], V[9];
            jacobi_svd3(v.sigma, U, S, V);
            if (S[1] < S[0] * 1e-3) S[1] = S[0] * 1e-3;
            if (S[2] < S[0] * 1e-3) S[2] = S[0] * 1e-3;
            const double il[3] = {1.0 / S[0], 1.0 / S[1], 1.0 / S[2]};
            double VL[9];
            for (int k = 0; k < 3; ++k) for (int i = 0; i < 3; ++i) VL[i + k * 3] = V[i + k * 3] * il[k];
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i)
                    v.info[i + j * 3] = (VL[i + 0 * 3] * U[j + 0 * 3] + VL[i + 1 * 3] * U[j + 1 * 3]) + VL[i + 2 * 3] * U[j + 2 * 3];
        }
    }

    int AddCloud(const Cloud& cloud_world_full, const Cloud&) override {  // :182-227
        const Cloud cloud_world = voxel_grid(cloud_world_full, p.source_cloud_filter_size);
        if (p.is_localization_mode) { const auto f = flat_xyz(cloud_world); fitness_tree.Build(f.data(), cloud_world.size(), 3); }
        struct Less {  // compare_function.h:17-22
            bool operator()(const Key3& a, const Key3& b) const {
                if (a.x != b.x) return a.x < b.x;
                if (a.y != b.y) return a.y < b.y;
                return a.z < b.z;
            }
        };
        std::set<Key3, Less> active;
        const double inv = 1.0 / p.ndt_voxel_size;
        for (const P4& pt : cloud_world) {
            const double pe[3] = {double(pt.x), double(pt.y), double(pt.z)};
            const Key3 key{int(pe[0] * inv), int(pe[1] * inv), int(pe[2] * inv)};
            auto iter = grids.find(key);
            if (iter == grids.end()) {
                VoxelData vd;
                vd.points = {pe[0], pe[1], pe[2]};
                vd.num_points = 1;
                vd.vid = next_vid++;
                data.emplace_front(key, vd);
                grids.insert({key, data.begin()});
                if (data.size() >= size_t(p.ndt_capacity)) { grids.erase(data.back().first); data.pop_back(); }
            } else {
                VoxelData& v = iter->second->second;
                v.points.push_back(pe[0]); v.points.push_back(pe[1]); v.points.push_back(pe[2]);
                if (!v.estimated) v.num_points++;
                data.splice(data.begin(), data, iter->second);
                iter->second = data.begin();
            }
            active.insert(key);
        }
        for (const Key3& k : active) {
            auto it = grids.find(k);
            if (it == grids.end()) {  // reference does grids_[key] (would insert a null iterator and crash); cannot happen unless evicted
                continue;
            }
            UpdateVoxel(it->second->second);
        }
        flag_first_scan = p.is_localization_mode ? true : false;
        return 0;
    }

    bool Match(const Cloud& ordered, const Cloud&, double* T, bool update_map) override {  // :229-337
        source = voxel_grid(ordered, p.source_cloud_filter_size);
        double pose[16];
        std::memcpy(pose, T, sizeof(pose));
        const size_t n = source.size();
        const size_t total = n * 7;
        log.clear(); counters = flo_counters{}; stats = flo_stats{};
        const double inv = 1.0 / p.ndt_voxel_size;
        hit_vid.assign(total, -1);
        for (int it = 0; it < int(p.max_iterations); ++it) {
            std::vector<double> err(total * 3);           // per-iteration allocations mirror :247-250
            std::vector<double> infov(total * 9);
            std::vector<double> jac(total * 18);
            eff7.assign(total, 0);
            std::fill(hit_vid.begin(), hit_vid.end(), -1);
            uint64_t hits = 0;
            const long ln = long(n);
#pragma omp parallel for schedule(static) reduction(+ : hits)
            for (long li = 0; li < ln; ++li) {
                const size_t idx = size_t(li);
                const double pt[3] = {double(source[idx].x), double(source[idx].y), double(source[idx].z)};
                const double q[3] = {((pose[0] * pt[0] + pose[4] * pt[1]) + pose[8] * pt[2]) + pose[12],
                                     ((pose[1] * pt[0] + pose[5] * pt[1]) + pose[9] * pt[2]) + pose[13],
                                     ((pose[2] * pt[0] + pose[6] * pt[1]) + pose[10] * pt[2]) + pose[14]};
                const Key3 key{int(q[0] * inv), int(q[1] * inv), int(q[2] * inv)};
                for (size_t i = 0; i < 7; ++i) {
                    const Key3 rk{key.x + nearby[i].x, key.y + nearby[i].y, key.z + nearby[i].z};
                    auto g = grids.find(rk);
                    const size_t ri = idx * 7 + i;
                    if (g != grids.end() && g->second->second.estimated) {
                        const VoxelData& v = g->second->second;
                        hits++;
                        const double e[3] = {q[0] - v.mu[0], q[1] - v.mu[1], q[2] - v.mu[2]};
                        // e^T * info * e : (e^T info) first, then * e
                        double ei[3];
                        for (int c = 0; c < 3; ++c) ei[c] = (e[0] * v.info[0 + c * 3] + e[1] * v.info[1 + c * 3]) + e[2] * v.info[2 + c * 3];
                        const double res = (ei[0] * e[0] + ei[1] * e[1]) + ei[2] * e[2];
                        if (std::isnan(res) || res > p.ndt_res_outlier_threshold) continue;
                        double R[9], hat[9], RH[9];
                        for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) R[r + j * 3] = pose[r + j * 4];
                        so3_hat(pt, hat);
                        mat3_mul(R, hat, RH);
                        double* J = &jac[ri * 18];
                        for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) { J[r + j * 3] = -RH[r + j * 3]; J[r + (j + 3) * 3] = (r == j) ? 1.0 : 0.0; }
                        std::memcpy(&err[ri * 3], e, sizeof(e));
                        std::memcpy(&infov[ri * 9], v.info, sizeof(v.info));
                        eff7[ri] = 1;
                        hit_vid[ri] = v.vid;
                    }
                }
            }
            counters.point_iters += n; counters.probes += n * 7; counters.hit_voxels += hits;
            double total_res = 0.0;
            int effective = 0;
            double H[36] = {0}, b[6] = {0};
            for (size_t ri = 0; ri < total; ++ri) {
                if (!eff7[ri]) continue;
                const double* e = &err[ri * 3];
                const double* I = &infov[ri * 9];
                const double* J = &jac[ri * 18];
                double ei[3];
                for (int c = 0; c < 3; ++c) ei[c] = (e[0] * I[0 + c * 3] + e[1] * I[1 + c * 3]) + e[2] * I[2 + c * 3];
                total_res += (ei[0] * e[0] + ei[1] * e[1]) + ei[2] * e[2];
                effective++;
                // (J^T * I) 6x3, then * J  and  (-J^T * I) * e
                double JtI[18];  // 6x3 col-major
                for (int c = 0; c < 3; ++c)
                    for (int a = 0; a < 6; ++a)
                        JtI[a + c * 6] = (J[0 + a * 3] * I[0 + c * 3] + J[1 + a * 3] * I[1 + c * 3]) + J[2 + a * 3] * I[2 + c * 3];
                for (int bb = 0; bb < 6; ++bb)
                    for (int a = 0; a < 6; ++a)
                        H[a + bb * 6] += (JtI[a + 0 * 6] * J[0 + bb * 3] + JtI[a + 1 * 6] * J[1 + bb * 3]) + JtI[a + 2 * 6] * J[2 + bb * 3];
                for (int a = 0; a < 6; ++a)
                    b[a] += ((-JtI[a + 0 * 6]) * e[0] + (-JtI[a + 1 * 6]) * e[1]) + (-JtI[a + 2 * 6]) * e[2];
            }
            std::memcpy(last_H, H, sizeof(H)); std::memcpy(last_g, b, sizeof(b));
            stats.iterations = it + 1; stats.n_valid = effective; stats.sum_res = total_res; stats.n_source = int(n);
            if (effective < p.ndt_min_effective_pts) {
                std::memcpy(T, pose, sizeof(pose));
                Log(pose, effective, total_res);
                stats.converged = 0;
                return false;
            }
            PartialPivLU<6> lu;
            lu.compute(H);
            double invH[36], dx[6];
            lu.inverse(invH);
            for (int a = 0; a < 6; ++a) { double s = 0.0; for (int k = 0; k < 6; ++k) s += invH[a + k * 6] * b[k]; dx[a] = s; }
            double Rd[9], R[9], Rn[9];
            so3_exp(dx, Rd);
            for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) R[r + j * 3] = pose[r + j * 4];
            mat3_mul(R, Rd, Rn);
            for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) pose[r + j * 4] = Rn[r + j * 3];
            pose[12] += dx[3]; pose[13] += dx[4]; pose[14] += dx[5];
            std::memcpy(stats.last_dx, dx, sizeof(dx));
            Log(pose, effective, total_res);
            if (norm3(dx) < p.rotation_converge_thres && norm3(dx + 3) < p.position_converge_thres) break;
        }
        const bool has_converge = true;  // :325
        if (!p.is_localization_mode && update_map) {
            AddCloud(transform_cloud_f(source, T), Cloud());  // Q11: input T, not pose (:327-329)
            stats.map_updated = 1;
        }
        std::memcpy(T, pose, sizeof(pose));
        std::memcpy(final_T, pose, sizeof(pose));
        stats.converged = 1;
        return has_converge;
    }
    float Fitness(float max_range) const override {  // :345-372
        if (!p.is_localization_mode) return std::numeric_limits<float>::max();
        return fitness_score(source, final_T, fitness_tree, max_range);
    }
    int GetCorr(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) const override {
        const size_t n = std::min(cap, source.size());
        for (size_t i = 0; i < n; ++i) {
            int c = 0;
            for (int k = 0; k < 7; ++k) { ids[i * 7 + k] = hit_vid[i * 7 + k]; c += eff7[i * 7 + k] ? 1 : 0; }
            cnt[i] = uint8_t(c); valid[i] = c > 0;
        }
        return int(n);
    }
    size_t MapSize(int) const override { return data.size(); }
};

// =============================================================================================
// LoamFull<double>  and  LoamPointToPlaneKdtree<double>
// =============================================================================================
struct FeatureSet {  // one feature class of LoamFull (corner or planar)
    std::vector<PerPoint> coeffs;
    std::vector<char> flags;
    std::vector<int> nn;  // n x 5 map indices of the last iteration
    std::vector<uint8_t> cnt;
    void Reset(size_t n) { coeffs.resize(n); flags.assign(n, 0); nn.assign(n * 5, -1); cnt.assign(n, 0); }
};

// loam_full_kdtree.h:216-272
static inline bool line_residual(const P4 nn[5], const P4& src, const P4& src_t, const double* T, double ratio,
                                 double J[6], double& dist) {
    double P[15];  // 3x5 col-major
    for (int j = 0; j < 5; ++j) { P[0 + j * 3] = nn[j].x; P[1 + j * 3] = nn[j].y; P[2 + j * 3] = nn[j].z; }
    double c[3];
    for (int a = 0; a < 3; ++a) c[a] = ((((P[a] + P[a + 3]) + P[a + 6]) + P[a + 9]) + P[a + 12]) / 5.0;
    double D[15];
    for (int j = 0; j < 5; ++j) for (int a = 0; a < 3; ++a) D[a + j * 3] = P[a + j * 3] - c[a];
    double C[9];
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
            for (int k = 0; k < 5; ++k) s += D[i + k * 3] * D[j + k * 3];
            C[i + j * 3] = s / 5.0;
        }
    double U[9], S[3], V[9];
    jacobi_svd3(C, U, S, V);
    if (S[0] <= ratio * S[1]) return false;
    const double n[3] = {V[0], V[1], V[2]};
    const double ps[3] = {double(src.x), double(src.y), double(src.z)};
    const double pt[3] = {double(src_t.x), double(src_t.y), double(src_t.z)};
    const double a[3] = {pt[0] - c[0], pt[1] - c[1], pt[2] - c[2]};
    const double w[3] = {a[1] * n[2] - a[2] * n[1], a[2] * n[0] - a[0] * n[2], a[0] * n[1] - a[1] * n[0]};
    dist = norm3(w);
    const double u[3] = {w[0] / dist, w[1] / dist, w[2] / dist};
    const double v[3] = {(T[0] * ps[0] + T[4] * ps[1]) + T[8] * ps[2], (T[1] * ps[0] + T[5] * ps[1]) + T[9] * ps[2],
                         (T[2] * ps[0] + T[6] * ps[1]) + T[10] * ps[2]};
    double hn[9], hv[9], M[9], hmn[9];
    so3_hat(n, hn);
    so3_hat(v, hv);
    mat3_mul(hn, hv, M);  // SO3Hat(corner_n) * SO3Hat(R*ps)
    for (int i = 0; i < 3; ++i) J[i] = (M[0 + i * 3] * u[0] + M[1 + i * 3] * u[1]) + M[2 + i * 3] * u[2];  // M^T u
    const double mn[3] = {-n[0], -n[1], -n[2]};
    so3_hat(mn, hmn);  // SO3Hat(-corner_n) * I
    for (int i = 0; i < 3; ++i) J[3 + i] = (hmn[0 + i * 3] * u[0] + hmn[1 + i * 3] * u[1]) + hmn[2 + i * 3] * u[2];
    return true;
}

static void sum_features(const FeatureSet& f, double* H, double* g, size_t& n_valid, double& res) {
    n_valid = 0; res = 0.0;
    for (size_t i = 0; i < f.flags.size(); ++i) {
        if (!f.flags[i]) continue;
        n_valid++;
        for (int k = 0; k < 36; ++k) H[k] += f.coeffs[i].H[k];
        for (int k = 0; k < 6; ++k) g[k] += f.coeffs[i].g[k];
        res += f.coeffs[i].res;
    }
}

struct LoamFull final : MatcherBase {
    std::deque<Cloud> corner_deque, planar_deque;
    Cloud local_corner, local_planar;
    KdTree corner_tree, planar_tree;
    FeatureSet corner, planar;
    KeyframeGate gate;
    double T_[16];

    int AddCloud(const Cloud& planar_cloud, const Cloud& corner_cloud) override {  // loam_full_kdtree.h:65-104
        corner_deque.push_back(corner_cloud);
        planar_deque.push_back(planar_cloud);
        if (planar_deque.size() > p.local_planar_size) planar_deque.pop_front();
        if (corner_deque.size() > p.local_corner_size) corner_deque.pop_front();
        local_planar.clear(); local_corner.clear();
        for (auto& c : planar_deque) local_planar.insert(local_planar.end(), c.begin(), c.end());
        for (auto& c : corner_deque) local_corner.insert(local_corner.end(), c.begin(), c.end());
        if (planar_deque.size() > 5) local_planar = voxel_grid(local_planar, p.planar_voxel_filter_size);
        if (corner_deque.size() > 5) local_corner = voxel_grid(local_corner, p.corner_voxel_filter_size);
        auto f = flat_xyz(local_planar); planar_tree.Build(f.data(), local_planar.size(), 3);
        f = flat_xyz(local_corner); corner_tree.Build(f.data(), local_corner.size(), 3);
        return 0;
    }

    void CornerMatch(const Cloud& src) {  // :211-273
        const long n = long(src.size());
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) {
            const P4& sp = src[size_t(i)];
            const P4 tp = transform_point_d(sp, T_);
            KdTree::Hit h[5];
            const float q[3] = {tp.x, tp.y, tp.z};
            const int k = corner_tree.Knn(q, 5, h);
            corner.cnt[size_t(i)] = 0;
            for (int j = 0; j < 5; ++j) corner.nn[size_t(i) * 5 + j] = -1;
            if (k < 5) continue;  // reference would read out of bounds; maps always hold >= 5 points
            if (double(h[4].d2) > p.point_search_thres) continue;
            corner.cnt[size_t(i)] = 5;  // introspection only: ids are reported for gate-accepted neighbour sets
            for (int j = 0; j < 5; ++j) corner.nn[size_t(i) * 5 + j] = h[j].idx;
            P4 nn[5];
            for (int j = 0; j < 5; ++j) { const float* m = corner_tree.point(h[j].idx); nn[j] = P4{m[0], m[1], m[2], 0}; }
            double J[6], d;
            if (!line_residual(nn, sp, tp, T_, p.line_ratio_thres, J, d)) continue;
            corner.flags[size_t(i)] = 1;
            fill_rank1(corner.coeffs[size_t(i)], J, d);
        }
        counters.point_iters += uint64_t(n);
    }
    void PlanarMatch(const Cloud& src) {  // :275-345
        const long n = long(src.size());
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) {
            const P4& sp = src[size_t(i)];
            const P4 tp = transform_point_d(sp, T_);
            KdTree::Hit h[5];
            const float q[3] = {tp.x, tp.y, tp.z};
            const int k = planar_tree.Knn(q, 5, h);
            planar.cnt[size_t(i)] = 0;
            for (int j = 0; j < 5; ++j) planar.nn[size_t(i) * 5 + j] = -1;
            if (k < 5) continue;
            if (double(h[4].d2) > p.point_search_thres) continue;
            planar.cnt[size_t(i)] = 5;
            for (int j = 0; j < 5; ++j) planar.nn[size_t(i) * 5 + j] = h[j].idx;
            P4 nn[5];
            for (int j = 0; j < 5; ++j) { const float* m = planar_tree.point(h[j].idx); nn[j] = P4{m[0], m[1], m[2], 0}; }
            double J[6], r;
            if (!plane_residual(nn, sp, tp, T_, p.point_to_planar_thres, J, r)) continue;
            planar.flags[size_t(i)] = 1;
            fill_rank1(planar.coeffs[size_t(i)], J, r);
        }
        counters.point_iters += uint64_t(n);
    }

    bool Match(const Cloud& planar_source, const Cloud& corner_source, double* T, bool update_map) override {  // :106-204
        corner.Reset(corner_source.size());
        planar.Reset(planar_source.size());
        bool has_converge = true;
        std::memcpy(T_, T, sizeof(T_));
        LoamLoopState st;
        size_t nvc = 0, nvp = 0;
        double rc = 0, rp = 0;
        log.clear(); counters = flo_counters{}; stats = flo_stats{};
        for (unsigned it = 0; it < p.max_iterations; ++it) {
            double H[36] = {0}, g[6] = {0};
            CornerMatch(corner_source);
            PlanarMatch(planar_source);
            sum_features(corner, H, g, nvc, rc);  // :347-372: corners first, then planars
            sum_features(planar, H, g, nvp, rp);
            std::memcpy(last_H, H, sizeof(H)); std::memcpy(last_g, g, sizeof(g));
            const bool stop = loam_update(T_, H, g, p.rotation_converge_thres, p.position_converge_thres, st, stats.last_dx);
            stats.iterations = int(it) + 1;
            Log(T_, int(nvp), rp);
            if (stop) break;
        }
        std::memcpy(T, T_, sizeof(T_));
        if (nvp < 50) has_converge = false;
        stats.n_valid = int(nvp); stats.n_valid_corner = int(nvc); stats.sum_res = rp; stats.sum_res_corner = rc;
        stats.n_source = int(planar_source.size()); stats.n_source_corner = int(corner_source.size());
        stats.converged = has_converge ? 1 : 0;
        if (has_converge && gate.Need(T_, p.dist_thre_add_cloud, p.rot_thre_add_cloud) && update_map) {
            Cloud tp(planar_source.size()), tc(corner_source.size());
            for (size_t i = 0; i < tp.size(); ++i) tp[i] = transform_point_d(planar_source[i], T_);  // pcl::transformPointCloud(.., Matrix4d)
            for (size_t i = 0; i < tc.size(); ++i) tc[i] = transform_point_d(corner_source[i], T_);
            AddCloud(tp, tc);
            stats.map_updated = 1;
        }
        return has_converge;
    }
    float Fitness(float) const override { return std::numeric_limits<float>::max(); }  // :206-208 FloatNaN
    int GetCorr(int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) const override {
        const FeatureSet& f = slot == 1 ? corner : planar;
        const size_t n = std::min(cap, f.flags.size());
        for (size_t i = 0; i < n; ++i) { for (int j = 0; j < 5; ++j) ids[i * 5 + j] = f.nn[i * 5 + j]; cnt[i] = f.cnt[i]; valid[i] = uint8_t(f.flags[i]); }
        return int(n);
    }
    size_t MapSize(int slot) const override { return slot == 1 ? local_corner.size() : local_planar.size(); }
    size_t MapDump(int slot, float* xyz, size_t cap) const override {
        const Cloud& m = slot == 1 ? local_corner : local_planar;
        const size_t n = std::min(cap, m.size());
        for (size_t i = 0; i < n; ++i) { xyz[3 * i] = m[i].x; xyz[3 * i + 1] = m[i].y; xyz[3 * i + 2] = m[i].z; }
        return m.size();
    }
};

struct P2PlaneKd final : MatcherBase {  // loam_point_to_plane_kdtree.h
    std::deque<Cloud> cloud_deque;
    Cloud local_map, source_copy;
    KdTree tree;
    FeatureSet planar;
    KeyframeGate gate;
    double T_[16], final_T[16];

    int AddCloud(const Cloud& planar_cloud, const Cloud&) override {  // :56-79
        if (p.is_localization_mode) {
            local_map = planar_cloud;
        } else {
            cloud_deque.push_back(planar_cloud);
            if (cloud_deque.size() > p.local_map_size) cloud_deque.pop_front();
            local_map.clear();
            for (auto& c : cloud_deque) local_map.insert(local_map.end(), c.begin(), c.end());
        }
        local_map = voxel_grid(local_map, p.map_cloud_filter_size);
        const auto f = flat_xyz(local_map);
        tree.Build(f.data(), local_map.size(), 3);
        return 0;
    }
    void PlanerMatch(const Cloud& src) {  // :204-272
        const long n = long(src.size());
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) {
            const P4& sp = src[size_t(i)];
            const P4 tp = transform_point_d(sp, T_);
            KdTree::Hit h[5];
            const float q[3] = {tp.x, tp.y, tp.z};
            const int k = tree.Knn(q, 5, h);
            planar.cnt[size_t(i)] = uint8_t(k);
            for (int j = 0; j < 5; ++j) planar.nn[size_t(i) * 5 + j] = j < k ? h[j].idx : -1;
            if (k < 5) continue;
            P4 nn[5];
            for (int j = 0; j < 5; ++j) { const float* m = tree.point(h[j].idx); nn[j] = P4{m[0], m[1], m[2], 0}; }
            double J[6], r;
            if (!plane_residual(nn, sp, tp, T_, p.point_to_planar_thres, J, r)) continue;
            planar.flags[size_t(i)] = 1;
            fill_rank1(planar.coeffs[size_t(i)], J, r);
        }
        counters.point_iters += uint64_t(n);
    }
    bool Match(const Cloud& planar_source, const Cloud&, double* T, bool update_map) override {  // :81-158
        source_copy = planar_source;
        planar.Reset(planar_source.size());
        std::memcpy(T_, T, sizeof(T_));
        bool has_converge = true;
        LoamLoopState st;
        size_t nv = 0;
        double res = 0;
        log.clear(); counters = flo_counters{}; stats = flo_stats{};
        for (unsigned it = 0; it < p.max_iterations; ++it) {
            double H[36] = {0}, g[6] = {0};
            PlanerMatch(planar_source);
            sum_features(planar, H, g, nv, res);
            std::memcpy(last_H, H, sizeof(H)); std::memcpy(last_g, g, sizeof(g));
            const bool stop = loam_update(T_, H, g, p.rotation_converge_thres, p.position_converge_thres, st, stats.last_dx);
            stats.iterations = int(it) + 1;
            Log(T_, int(nv), res);
            if (stop) break;
        }
        std::memcpy(T, T_, sizeof(T_));
        std::memcpy(final_T, T_, sizeof(T_));
        if (nv < 50u) has_converge = false;
        stats.n_valid = int(nv); stats.sum_res = res; stats.n_source = int(planar_source.size());
        stats.converged = has_converge ? 1 : 0;
        if (has_converge && gate.Need(final_T, p.dist_thre_add_cloud, p.rot_thre_add_cloud) && !p.is_localization_mode && update_map) {
            AddCloud(transform_cloud_f(source_copy, final_T), Cloud());
            stats.map_updated = 1;
        }
        return has_converge;
    }
    float Fitness(float max_range) const override { return fitness_score(source_copy, final_T, tree, max_range); }
    int GetCorr(int, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) const override {
        const size_t n = std::min(cap, planar.flags.size());
        for (size_t i = 0; i < n; ++i) { for (int j = 0; j < 5; ++j) ids[i * 5 + j] = planar.nn[i * 5 + j]; cnt[i] = planar.cnt[i]; valid[i] = uint8_t(planar.flags[i]); }
        return int(n);
    }
    size_t MapSize(int) const override { return local_map.size(); }
    size_t MapDump(int, float* xyz, size_t cap) const override {
        const size_t n = std::min(cap, local_map.size());
        for (size_t i = 0; i < n; ++i) { xyz[3 * i] = local_map[i].x; xyz[3 * i + 1] = local_map[i].y; xyz[3 * i + 2] = local_map[i].z; }
        return local_map.size();
    }
};

}  // namespace flo

// =============================================================================================
// C ABI
// =============================================================================================
using namespace flo;

extern "C" {

void* flo_create(int kind, const flo_params* p) {
    if (!p || p->struct_size != sizeof(flo_params)) return nullptr;
    MatcherBase* m = nullptr;
    switch (kind) {
        case FLO_ICP_OPTIMIZED: m = new IcpOptimized(); break;
        case FLO_P2PLANE_IVOX: m = new P2PlaneIvox(); break;
        case FLO_INCREMENTAL_NDT: m = new IncNdt(); break;
        case FLO_LOAM_FULL: m = new LoamFull(); break;
        case FLO_P2PLANE_KDTREE: m = new P2PlaneKd(); break;
        default: return nullptr;
    }
    m->p = *p;
    return m;
}
void flo_destroy(void* h) { delete static_cast<MatcherBase*>(h); }
void flo_set_threads(int n) { g_threads = n; if (n > 0) omp_set_num_threads(n); }
int flo_get_threads(void) { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

int flo_add_cloud(void* h, const float* c0, size_t n0, const float* c1, size_t n1, int stride) {
    auto* m = static_cast<MatcherBase*>(h);
    return m->AddCloud(make_cloud(c0, n0, stride), c1 ? make_cloud(c1, n1, stride) : Cloud());
}
int flo_match(void* h, const float* s0, size_t n0, const float* s1, size_t n1, int stride, double T[16],
              int update_map, flo_stats* stats) {
    auto* m = static_cast<MatcherBase*>(h);
    const bool ok = m->Match(make_cloud(s0, n0, stride), s1 ? make_cloud(s1, n1, stride) : Cloud(), T, update_map != 0);
    if (stats) *stats = m->stats;
    return ok ? 0 : 1;
}
float flo_fitness(void* h, float max_range) { return static_cast<MatcherBase*>(h)->Fitness(max_range); }

int flo_get_iteration_log(void* h, double* T_iters, int32_t* n_valid, double* sum_res, int cap) {
    auto* m = static_cast<MatcherBase*>(h);
    const int n = std::min<int>(cap, int(m->log.size()));
    for (int i = 0; i < n; ++i) {
        if (T_iters) std::memcpy(T_iters + 16 * i, m->log[i].T, sizeof(double) * 16);
        if (n_valid) n_valid[i] = m->log[i].n_valid;
        if (sum_res) sum_res[i] = m->log[i].sum_res;
    }
    return int(m->log.size());
}
int flo_get_correspondences(void* h, int slot, int32_t* ids, uint8_t* cnt, uint8_t* valid, size_t cap) {
    return static_cast<MatcherBase*>(h)->GetCorr(slot, ids, cnt, valid, cap);
}
int flo_get_counters(void* h, flo_counters* out) { *out = static_cast<MatcherBase*>(h)->counters; return 0; }
size_t flo_get_tie_flags(void* h, uint8_t* out, size_t cap) {
    const std::vector<uint8_t>& f = static_cast<MatcherBase*>(h)->tie_flag;
    const size_t n = std::min(cap, f.size());
    if (out && n) std::memcpy(out, f.data(), n);
    return f.size();
}
int flo_get_last_system(void* h, double* H36, double* g6) {
    auto* m = static_cast<MatcherBase*>(h);
    std::memcpy(H36, m->last_H, sizeof(m->last_H));
    std::memcpy(g6, m->last_g, sizeof(m->last_g));
    return 0;
}
void flo_set_instrumentation(void* h, int on) { static_cast<MatcherBase*>(h)->instrument = on != 0; }
void flo_set_tie_break_by_id(int on) { flo::tie_break_by_id() = on != 0; }
size_t flo_map_size(void* h, int slot) { return static_cast<MatcherBase*>(h)->MapSize(slot); }
void flo_set_ivox_capacity(void* h, size_t cap) {  /* test hook: the reference hard-codes 1,000,000 (ivox_map.h:35) */
    auto* m = dynamic_cast<P2PlaneIvox*>(static_cast<MatcherBase*>(h));
    if (m) m->ivox->capacity_ = cap;
}
size_t flo_map_voxels(void* h) {
    auto* m = dynamic_cast<P2PlaneIvox*>(static_cast<MatcherBase*>(h));
    return m ? m->ivox->grids_map_.size() : 0;
}
size_t flo_map_dump(void* h, int slot, float* xyz, size_t cap) { return static_cast<MatcherBase*>(h)->MapDump(slot, xyz, cap); }
size_t flo_ndt_dump(void* h, int32_t* keys, double* mu, double* info, uint8_t* est, int32_t* npts, size_t cap) {
    auto* m = dynamic_cast<IncNdt*>(static_cast<MatcherBase*>(h));
    if (!m) return 0;
    std::vector<const IncNdt::KD*> v;
    for (auto& kd : m->data) v.push_back(&kd);
    std::sort(v.begin(), v.end(), [](auto* a, auto* b) { return a->second.vid < b->second.vid; });
    const size_t n = std::min(cap, v.size());
    for (size_t i = 0; i < n; ++i) {
        keys[3 * i] = v[i]->first.x; keys[3 * i + 1] = v[i]->first.y; keys[3 * i + 2] = v[i]->first.z;
        std::memcpy(mu + 3 * i, v[i]->second.mu, 24);
        std::memcpy(info + 9 * i, v[i]->second.info, 72);
        est[i] = v[i]->second.estimated;
        npts[i] = v[i]->second.num_points;
    }
    return v.size();
}

void* flo_feat_create(const flo_feat_params* p) {
    if (!p || p->struct_size != sizeof(flo_feat_params) || p->vertical_scan <= 0 || p->horizontal_scan <= 0) return nullptr;
    auto* s = new flo::FeatState();
    s->p.rows = p->vertical_scan; s->p.cols = p->horizontal_scan; s->p.h_res = p->horizontal_resolution;
    s->p.min_dist = p->min_distance; s->p.max_dist = p->max_distance; s->p.corner_thr = p->corner_thres; s->p.planar_thr = p->planar_thres;
    return s;
}
void flo_feat_destroy(void* h) { delete static_cast<flo::FeatState*>(h); }
int64_t flo_feat_project(void* h, const void* raw, size_t n, size_t stride, size_t off_xyz, size_t off_i, size_t off_ring) {
    auto* s = static_cast<flo::FeatState*>(h);
    flo::feat_project(*s, static_cast<const uint8_t*>(raw), n, stride, off_xyz, off_i, off_ring);
    return int64_t(s->ordered.size());
}
int flo_feat_extract(void* h) { return flo::feat_extract(*static_cast<flo::FeatState*>(h)) ? 1 : 0; }
uint64_t flo_feat_tie_pairs(void* h) { return static_cast<flo::FeatState*>(h)->tie_pairs; }
void flo_feat_set_sort_mode(void* h, int mode) { static_cast<flo::FeatState*>(h)->sort_mode = mode; }
int flo_col_index(float x, float y, float h_res, int cols) { return flo::col_index(x, y, h_res, cols); }
float flo_fast_atan2f(float y, float x) { return flo::fast_atan2f(y, x); }
size_t flo_feat_get(void* h, int what, void* out, size_t cap) {
    auto* s = static_cast<flo::FeatState*>(h);
    auto copy = [&](const void* src, size_t count, size_t elem) -> size_t {
        if (out) std::memcpy(out, src, std::min(count, cap) * elem);
        return count;
    };
    auto cloud = [&](const std::vector<int>& idx) -> size_t {
        if (out) {
            auto* o = static_cast<flo::FeatPoint*>(out);
            for (size_t k = 0; k < std::min(idx.size(), cap); ++k) o[k] = s->ordered[size_t(idx[k])];
        }
        return idx.size();
    };
    switch (what) {
        case FLO_FEAT_ORDERED: return copy(s->ordered.data(), s->ordered.size(), sizeof(flo::FeatPoint));
        case FLO_FEAT_DEPTH: return copy(s->depth.data(), s->depth.size(), 4);
        case FLO_FEAT_COL: return copy(s->col.data(), s->col.size(), 4);
        case FLO_FEAT_ROW_START: return copy(s->row_start.data(), s->row_start.size(), 4);
        case FLO_FEAT_ROW_END: return copy(s->row_end.data(), s->row_end.size(), 4);
        case FLO_FEAT_CORNER: return cloud(s->corner_idx);
        case FLO_FEAT_PLANAR: return cloud(s->planar_idx);
        case FLO_FEAT_IS_CORNER: return copy(s->is_corner.data(), s->is_corner.size(), 1);
        case FLO_FEAT_ROUGHNESS: return copy(s->roughness.data(), s->roughness.size(), 4);
        case FLO_FEAT_VALID_PRE: return copy(s->valid_pre.data(), s->valid_pre.size(), 1);
        case FLO_FEAT_VALID_POST: return copy(s->valid_post.data(), s->valid_post.size(), 1);
        case FLO_FEAT_CORNER_IDX: return copy(s->corner_idx.data(), s->corner_idx.size(), 4);
        case FLO_FEAT_PLANAR_IDX: return copy(s->planar_idx.data(), s->planar_idx.size(), 4);
        case FLO_FEAT_RAW_INDEX: return copy(s->raw_index.data(), s->raw_index.size(), 4);
        default: return 0;
    }
}

float flo_loop_match(const float* src, size_t ns, const float* tgt, size_t nt, int stride, double T[16], flo_loop_stats* st) {
    static_assert(sizeof(flo_loop_stats) == sizeof(loop::LoopStats), "flo_loop_stats mirrors loop::LoopStats");
    loop::LoopStats ls{};
    const float f = loop::loop_match(make_cloud(src, ns, stride), make_cloud(tgt, nt, stride), T, &ls);
    if (st) std::memcpy(st, &ls, sizeof(ls));
    return f;
}
int flo_ndt_derivatives(const float* src, size_t ns, const float* tgt, size_t nt, int stride, float resolution, const double p[6], double* score,
                        double grad[6], double hess[36]) {
    loop::Ndt ndt;
    ndt.resolution = resolution;
    const Cloud s = make_cloud(src, ns, stride), t = make_cloud(tgt, nt, stride);
    ndt.set_target(t);
    if (!ndt.cells.ok) return -1;
    ndt.source = &s;
    const double c1 = 10.0 * (1.0 - ndt.outlier_ratio), c2 = ndt.outlier_ratio / std::pow(double(resolution), 3), d3 = -std::log(c2);
    ndt.gauss_d1 = -std::log(c1 + c2) - d3;
    ndt.gauss_d2 = -2.0 * std::log((-std::log(c1 * std::exp(-0.5) + c2) - d3) / ndt.gauss_d1);
    *score = ndt.derivatives(grad, hess, loop::pose_from_p(p), p, true);
    return 0;
}
size_t flo_ndt_leaves(const float* tgt, size_t nt, int stride, float resolution, int32_t* idx, int32_t* nr, double* mean3, double* icov9, float* centroid3,
                      size_t cap) {
    loop::TargetCells tc;
    tc.build(make_cloud(tgt, nt, stride), resolution);
    size_t k = 0;
    for (int li : tc.searchable) {
        if (k < cap) {
            const loop::Leaf& l = tc.leaves[li];
            idx[k] = li; nr[k] = l.nr;
            for (int a = 0; a < 3; ++a) { mean3[3 * k + a] = l.mean[a]; centroid3[3 * k + a] = l.centroid[a]; }
            for (int a = 0; a < 9; ++a) icov9[9 * k + a] = l.icov[a];
        }
        ++k;
    }
    return k;
}
void flo_gicp_covariances(const float* c, size_t n, int stride, int k, double eps, double* out9) {
    std::vector<double> out;
    loop::Gicp::covariances(make_cloud(c, n, stride), k, eps, out);
    std::memcpy(out9, out.data(), out.size() * sizeof(double));
}
int flo_gicp_fdf(const float* src, size_t ns, const float* tgt, size_t nt, int stride, const double guess[16], double corr_dist, const double x[6], double* f,
                 double g[6], int32_t* n_corr) {
    loop::Gicp gi;
    gi.corr_dist_threshold = corr_dist;
    gi.max_iterations = 0;  // build the correspondences of the first outer iteration only: align() stops after estimate()
    const Cloud s = make_cloud(src, ns, stride), t = make_cloud(tgt, nt, stride);
    if (s.size() < size_t(gi.k_correspondences) || t.size() < size_t(gi.k_correspondences)) return -1;
    gi.max_inner_iterations = 0;
    gi.align(s, t, loop::m4f_from_d(guess));
    if (gi.idx_src.size() < 4) return -1;
    *n_corr = int32_t(gi.idx_src.size());
    gi.fdf(x, f, g);
    return 0;
}
void flo_jacobi_svd_solve6(const double A[36], const double b[6], double x[6]) { loop::jacobi_svd_solve<6>(A, b, x); }

size_t flo_voxel_grid(const float* in, size_t n, int stride, float leaf, float* out) {
    const Cloud o = voxel_grid(make_cloud(in, n, stride), leaf);
    for (size_t i = 0; i < o.size(); ++i) { out[4 * i] = o[i].x; out[4 * i + 1] = o[i].y; out[4 * i + 2] = o[i].z; out[4 * i + 3] = o[i].i; }
    return o.size();
}
void flo_so3_exp(const double v[3], double R[9]) { so3_exp(v, R); }
void flo_so3_hat(const double v[3], double M[9]) { so3_hat(v, M); }
void flo_rpy(const double R[9], double rpy[3]) { rotation_to_rpy(R, rpy); }
void flo_colpiv_qr_solve_5x3(const double A[15], const double b[5], double x[3]) { colpiv_qr_solve<5, 3>(A, b, x); }
void flo_fullpiv_qr_solve_6(const double A[36], const double b[6], double x[6]) { fullpiv_qr_solve<6>(A, b, x); }
void flo_lu_inverse_6(const double A[36], double inv[36], double* det) {
    PartialPivLU<6> lu;
    lu.compute(A);
    lu.inverse(inv);
    *det = lu.determinant();
}
void flo_inverse3(const double A[9], double inv[9]) { inverse3(A, inv); }
void flo_svd3(const double A[9], double U[9], double S[3], double V[9]) { jacobi_svd3(A, U, S, V); }
int flo_knn_bruteforce(const float* map, size_t m, const float* q, int k, int32_t* idx, float* d2) {
    std::vector<KdTree::Hit> all(m);
    for (size_t i = 0; i < m; ++i) all[i] = KdTree::Hit{l2_simple(q, map + 3 * i), int(i)};
    const int kk = int(std::min<size_t>(size_t(k), m));
    std::partial_sort(all.begin(), all.begin() + kk, all.end(),
                      [](auto& a, auto& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); });
    for (int i = 0; i < kk; ++i) { idx[i] = all[i].idx; d2[i] = all[i].d2; }
    return kk;
}
int flo_kdtree_knn(const float* map, size_t m, const float* queries, size_t nq, int k, int32_t* idx, float* d2) {
    KdTree t;
    t.Build(map, m, 3);
    const long n = long(nq);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        std::vector<KdTree::Hit> h((size_t)k);
        const int c = t.Knn(queries + 3 * i, k, h.data());
        for (int j = 0; j < k; ++j) { idx[i * k + j] = j < c ? h[j].idx : -1; d2[i * k + j] = j < c ? h[j].d2 : INFINITY; }
    }
    return 0;
}

}  // extern "C"
