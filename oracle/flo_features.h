// ============================================================================
// oracle/flo_features.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// CPU restatement of the step immediately before Match in LoamFull_KdTree mode:
//   loam::PointcloudProjector::Project      src/loam/pointcloud_projector.cpp:32-133
//   loam::FeatureExtractor::ExtractFeatures src/loam/feature_extractor.cpp:36-222
//   FastAtan2                               include/common/math_function.h:159-186
// De-skew (LidarDistortionCorrector::ProcessPoint, IMU driven) is outside the scope table: points are
// taken as already corrected (the corrector is the identity here), exactly as preprocessing.cpp:229-232
// hands them over when no IMU motion is present.
//
// Pinned by the reference's own test vectors for the column rule: test/lidar_model_ut.cpp:9-36
// (LidarModel::ColIndex has the same formula, pointcloud_projector.cpp:69-74 / lidar_model.h:67-80)
// -> tests/test_oracle_features.py.  Everything else here: parity unpinned (no reference test exists).
//
// Unspecified in the reference and fixed here: the order std::sort(std::execution::par, ...) leaves
// elements of EQUAL roughness in (feature_extractor.cpp:143-148).  The oracle keeps them in ascending
// position (stable) and counts such adjacent pairs (tie_pairs) so that tests can tell when it matters.
// ============================================================================
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace flo {

struct FeatParams {
    int rows = 0, cols = 0;  // lidar_vertical_scan_, lidar_horizontal_scan_
    float h_res = 0.f;       // lidar_horizontal_resolution_ [rad]
    float min_dist = 0.f, max_dist = 0.f;
    float corner_thr = 0.f, planar_thr = 0.f;
};

struct FeatPoint { float x, y, z, i; };

// include/common/math_function.h:159-186 with Type = float
static inline float fast_atan2f(const float y, const float x) {
    const float p1 = float(0.9997878412794807), p3 = float(-0.3258083974640975), p5 = float(0.1555786518463281),
                p7 = float(-0.04432655554792128);
    const float ax = std::fabs(x), ay = std::fabs(y);
    const float eps = std::numeric_limits<float>::epsilon();
    float a;
    if (ax >= ay) {
        const float c = ay / (ax + eps), c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        const float c = ax / (ay + eps), c2 = c * c;
        a = float(M_PI_2) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = float(M_PI) - a;
    if (y < 0) a = float(2 * M_PI) - a;
    if (a > float(M_PI)) a -= float(2 * M_PI);
    return a;
}

// pointcloud_projector.cpp:69-74 (== LidarModel::ColIndex with FastAtan2, lidar_model.h:67-80)
static inline int col_index(const float x, const float y, const float h_res, const int cols) {
    int col = int(std::round(fast_atan2f(y, x) / h_res)) + cols / 2;
    if (col >= cols) col -= cols;
    return col;
}

struct FeatState {
    FeatParams p;
    // Project outputs (PointcloudCluster members)
    std::vector<FeatPoint> ordered;      // ordered_cloud_
    std::vector<float> depth;            // point_depth_vec_[0 .. N)
    std::vector<int> col;                // point_col_index_vec_[0 .. N)
    std::vector<int> row_start, row_end; // row_start_index_vec_, row_end_index_vec_
    std::vector<int> raw_index;          // oracle-only: which raw point won each ordered slot
    // ExtractFeatures outputs / intermediates
    std::vector<float> roughness;        // per ordered index (0 where the reference leaves it untouched)
    std::vector<uint8_t> valid_pre;      // is_valid_points_ after SelectValidPoints
    std::vector<uint8_t> valid_post;     // ... after SelectFeatures
    std::vector<uint8_t> is_corner;
    std::vector<int> corner_idx, planar_idx;  // ordered-cloud indices in emission order
    uint64_t tie_pairs = 0;
    int sort_mode = 0;  // 0: ties keep index order (the oracle's documented choice, what the HIP kernel reproduces); 1: plain std::sort on the
                        // same 8-byte records = the permutation libstdc++ gives the reference (pinned against oracle/_ref, tests/test_ref_pin.py)
};

// Project: pointcloud_projector.cpp:32-133.  pts: byte-strided raw points, ring per point.
static inline void feat_project(FeatState& s, const uint8_t* raw, size_t n, size_t stride, size_t off_xyz, size_t off_i, size_t off_ring) {
    const FeatParams& p = s.p;
    const size_t cells = size_t(p.rows) * size_t(p.cols);
    std::vector<float> range(cells, std::numeric_limits<float>::max());  // range_mat_
    std::vector<FeatPoint> temp(cells, FeatPoint{0.f, 0.f, 0.f, 0.f});
    std::vector<int> owner(cells, -1);
    for (size_t k = 0; k < n; ++k) {
        const uint8_t* q = raw + k * stride;
        float x, y, z, it;
        uint16_t ring;
        std::memcpy(&x, q + off_xyz, 4); std::memcpy(&y, q + off_xyz + 4, 4); std::memcpy(&z, q + off_xyz + 8, 4);
        std::memcpy(&it, q + off_i, 4); std::memcpy(&ring, q + off_ring, 2);
        const float d = std::sqrt(x * x + y * y + z * z);                  // :64
        if (d < p.min_dist || d > p.max_dist) continue;                   // :66-68
        const int row = int(ring);                                         // :70
        const int c = col_index(x, y, p.h_res, p.cols);                    // :71-76
        if (row >= p.rows || row < 0 || c < 0 || c >= p.cols) continue;   // :87-88
        const size_t idx = size_t(row) * size_t(p.cols) + size_t(c);
        if (range[idx] != std::numeric_limits<float>::max()) continue;    // first point wins (:92-93)
        range[idx] = d;                                                    // :107-111 (identity de-skew)
        temp[idx] = FeatPoint{x, y, z, it};
        owner[idx] = int(k);
    }
    s.ordered.clear(); s.depth.clear(); s.col.clear(); s.raw_index.clear();
    s.row_start.assign(size_t(p.rows), 0); s.row_end.assign(size_t(p.rows), 0);
    int count = 0;
    for (int r = 0; r < p.rows; ++r) {                                     // :115-132
        s.row_start[size_t(r)] = count + 5;
        for (int c = 0; c < p.cols; ++c) {
            const size_t idx = size_t(r) * size_t(p.cols) + size_t(c);
            if (range[idx] == std::numeric_limits<float>::max()) continue;
            s.depth.push_back(range[idx]);
            s.ordered.push_back(temp[idx]);
            s.col.push_back(c);
            s.raw_index.push_back(owner[idx]);
            ++count;
        }
        s.row_end[size_t(r)] = count - 6;
    }
}

// the +-5 neighbour suppression both selection loops share (feature_extractor.cpp:167-186, 194-212)
static inline void feat_suppress(std::vector<uint8_t>& valid, const std::vector<int>& col, const int index) {
    valid[size_t(index)] = 0;
    for (int k = 1; k <= 5; ++k) {
        if (std::abs(col[size_t(index + k)] - col[size_t(index + k - 1)]) > 10) break;
        valid[size_t(index + k)] = 0;
    }
    for (int k = -1; k >= -5; --k) {
        if (std::abs(col[size_t(index + k)] - col[size_t(index + k + 1)]) > 10) break;
        valid[size_t(index + k)] = 0;
    }
}

// ExtractFeatures: feature_extractor.cpp:36-222.  Needs N >= 12 (the reference indexes N-6 .. N-1 and i-5 .. i+6).
static inline bool feat_extract(FeatState& s) {
    const FeatParams& p = s.p;
    const int N = int(s.ordered.size());
    s.roughness.assign(size_t(std::max(N, 0)), 0.f);
    s.valid_pre.assign(size_t(std::max(N, 0)), 1);
    s.is_corner.assign(size_t(std::max(N, 0)), 0);
    s.corner_idx.clear(); s.planar_idx.clear();
    s.tie_pairs = 0;
    if (N < 12) { s.valid_post = s.valid_pre; return false; }
    std::vector<uint8_t>& valid = s.valid_pre;
    const std::vector<float>& d = s.depth;
    // SelectValidPoints :65-117
    for (int i = 0; i < 5; ++i) valid[size_t(i)] = 0;
    for (int i = 1; i <= 6; ++i) valid[size_t(N - i)] = 0;
    for (int i = 5; i < N - 6; ++i) {
        const float d1 = d[size_t(i)], d2 = d[size_t(i + 1)];
        const int cd = std::abs(s.col[size_t(i + 1)] - s.col[size_t(i)]);
        if (cd < 10) {
            if (double(d1 - d2) > 0.3) {
                for (int k = 0; k <= 5; ++k) valid[size_t(i - k)] = 0;
            } else if (double(d2 - d1) > 0.3) {
                for (int k = 1; k <= 6; ++k) valid[size_t(i + k)] = 0;
            }
        }
        const float f1 = std::abs(d[size_t(i - 1)] - d[size_t(i)]), f2 = std::abs(d[size_t(i + 1)] - d[size_t(i)]);
        if (double(f1) > 0.02 * double(d[size_t(i)]) && double(f2) > 0.02 * double(d[size_t(i)])) valid[size_t(i)] = 0;
    }
    // ComputeRoughness :48-62
    for (int i = 5; i < N - 5; ++i) {
        const float r = d[size_t(i - 5)] + d[size_t(i - 4)] + d[size_t(i - 3)] + d[size_t(i - 2)] + d[size_t(i - 1)] + d[size_t(i + 1)] +
                        d[size_t(i + 2)] + d[size_t(i + 3)] + d[size_t(i + 4)] + d[size_t(i + 5)] - 10.0f * d[size_t(i)];
        s.roughness[size_t(i)] = r * r;
    }
    // SelectFeatures :119-222
    struct PF { float rough; int index; };
    std::vector<PF> pf(static_cast<size_t>(N));
    for (int i = 0; i < N; ++i) pf[size_t(i)] = PF{s.roughness[size_t(i)], i};
    s.valid_post = valid;
    std::vector<uint8_t>& v = s.valid_post;
    for (int scan = 0; scan < p.rows; ++scan) {
        for (int i = 0; i < 6; ++i) {
            const int t = (s.row_end[size_t(scan)] - s.row_start[size_t(scan)]) / 6;
            const int b0 = s.row_start[size_t(scan)] + i * t, b1 = s.row_start[size_t(scan)] + (i + 1) * t;
            if (b0 >= b1) continue;
            if (s.sort_mode == 1) std::sort(pf.begin() + b0, pf.begin() + b1, [](const PF& l, const PF& r) -> bool { return l.rough < r.rough; });
            else std::stable_sort(pf.begin() + b0, pf.begin() + b1, [](const PF& l, const PF& r) { return l.rough < r.rough; });
            for (int j = b0 + 1; j < b1; ++j) if (pf[size_t(j)].rough == pf[size_t(j - 1)].rough) ++s.tie_pairs;
            int large = 0;
            for (int j = b1; j >= b0; --j) {  // inclusive upper bound: element b1 belongs to the next block (:151)
                const int index = pf[size_t(j)].index;
                if (pf[size_t(j)].rough > p.corner_thr && v[size_t(index)]) {
                    ++large;
                    if (large <= 20) {
                        s.is_corner[size_t(index)] = 1;
                        s.corner_idx.push_back(index);
                    } else {
                        break;
                    }
                    feat_suppress(v, s.col, index);
                }
            }
            for (int j = b0; j <= b1; ++j) {
                const int index = pf[size_t(j)].index;
                if (v[size_t(index)] && pf[size_t(j)].rough < p.planar_thr) feat_suppress(v, s.col, index);
                if (!s.is_corner[size_t(index)]) s.planar_idx.push_back(index);
            }
        }
    }
    return true;
}

}  // namespace flo
