// ============================================================================
// oracle/flo_kdtree.h  --  TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Stand-in for pcl::KdTreeFLANN<PointXYZI>::nearestKSearch (exact k-NN over
// xyz, float squared-L2, ascending), which the reference calls at
//   icp_optimized.h:85,203   loam_full_kdtree.h:225,289
//   loam_point_to_plane_kdtree.h:215 (5-NN)  and the GetFitnessScore loops.
// PCL/FLANN are not vendored in /root/reference (third-party, version
// unpinned; Noetic => PCL 1.10 / FLANN 1.9.1).  Published behaviour restated:
//   * flann::L2_Simple<float>: d = ((dx*dx) + dy*dy) + dz*dz in float.
//   * KDTreeSingleIndex, checks=-1, eps=0, sorted=true => EXACT k-NN.
//   * Order among exactly equal distances is traversal-defined in FLANN; this
//     oracle fixes (d2, index) ascending and documents it as its choice.
// A plain median-split kd-tree, deliberately a different structure from the
// product's GPU hash grid so the two are independent.
// ============================================================================
#pragma once
#include <vector>
#include <algorithm>
#include <cstdint>
#include <cmath>
#include <limits>

namespace flo {

static inline float l2_simple(const float* a, const float* b) {
    float r = 0.0f, d;
    d = a[0] - b[0]; r += d * d;
    d = a[1] - b[1]; r += d * d;
    d = a[2] - b[2]; r += d * d;
    return r;
}

class KdTree {
public:
    struct Hit { float d2; int idx; };
    void Build(const float* xyz, size_t n, int stride) {
        pts_.resize(n * 3);
        for (size_t i = 0; i < n; ++i) {
            pts_[3 * i + 0] = xyz[i * stride + 0];
            pts_[3 * i + 1] = xyz[i * stride + 1];
            pts_[3 * i + 2] = xyz[i * stride + 2];
        }
        order_.resize(n);
        for (size_t i = 0; i < n; ++i) order_[i] = int(i);
        nodes_.clear();
        nodes_.reserve(n / 4 + 16);
        if (n > 0) BuildRec(0, int(n));
    }
    size_t size() const { return order_.size(); }
    const float* point(int i) const { return &pts_[3 * size_t(i)]; }

    // exact k-NN, result sorted by (d2, idx) ascending; returns count (<=k)
    int Knn(const float* q, int k, Hit* out) const {
        int cnt = 0;
        if (nodes_.empty()) return 0;
        Search(0, q, k, out, cnt);
        return cnt;
    }

private:
    struct Node { int lo, hi, axis, left, right; float split; };
    static constexpr int kLeaf = 12;
    std::vector<float> pts_;
    std::vector<int> order_;
    std::vector<Node> nodes_;

    int BuildRec(int lo, int hi) {
        const int id = int(nodes_.size());
        nodes_.push_back(Node{lo, hi, -1, -1, -1, 0.0f});
        if (hi - lo <= kLeaf) return id;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                const float v = pts_[3 * size_t(order_[i]) + a];
                mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v);
            }
        int axis = 0;
        if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
        if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
        if (!(mx[axis] > mn[axis])) return id;  // all identical: keep as leaf
        const int mid = (lo + hi) / 2;
        std::nth_element(order_.begin() + lo, order_.begin() + mid, order_.begin() + hi,
                         [&](int a, int b) { return pts_[3 * size_t(a) + axis] < pts_[3 * size_t(b) + axis]; });
        const float split = pts_[3 * size_t(order_[mid]) + axis];
        nodes_[id].axis = axis;
        nodes_[id].split = split;
        const int l = BuildRec(lo, mid);
        const int r = BuildRec(mid, hi);
        nodes_[id].left = l;
        nodes_[id].right = r;
        return id;
    }
    static bool Less(const Hit& a, const Hit& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.idx < b.idx); }
    static void Insert(Hit h, int k, Hit* out, int& cnt) {
        if (cnt == k && !Less(h, out[k - 1])) return;
        int pos = (cnt < k) ? cnt++ : k - 1;
        while (pos > 0 && Less(h, out[pos - 1])) { out[pos] = out[pos - 1]; --pos; }
        out[pos] = h;
    }
    void Search(int id, const float* q, int k, Hit* out, int& cnt) const {
        const Node& nd = nodes_[id];
        if (nd.axis < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) {
                const int idx = order_[i];
                Insert(Hit{l2_simple(q, &pts_[3 * size_t(idx)]), idx}, k, out, cnt);
            }
            return;
        }
        const float diff = q[nd.axis] - nd.split;
        const int first = diff < 0.0f ? nd.left : nd.right;
        const int second = diff < 0.0f ? nd.right : nd.left;
        Search(first, q, k, out, cnt);
        // conservative prune: plane distance in double, visit on <= so equal-distance
        // candidates with a smaller index are never missed.
        const double pd = double(diff) * double(diff);
        if (cnt < k || pd <= double(out[cnt - 1].d2) * (1.0 + 1e-6) + 1e-30) Search(second, q, k, out, cnt);
    }
};

}  // namespace flo
