// oracle/ref_shim: pcl::KdTreeFLANN<PointT>::nearestKSearch (TEST INFRASTRUCTURE ONLY): exact k-NN over xyz, float
// squared-L2 (flann::L2_Simple order), ascending -- forwarded to oracle/flo_kdtree.h, the oracle's own stand-in for
// PCL 1.10 / FLANN 1.9.1 (third-party, not in /root/reference).  Ties: (d2, index) ascending, as documented there.
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "flo_kdtree.h"

namespace pcl {
template <class PointT>
class KdTreeFLANN {
    flo::KdTree tree_;
    typename PointCloud<PointT>::ConstPtr cloud_;
public:
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& cloud) {
        cloud_ = cloud;
        std::vector<float> xyz(cloud->size() * 3);
        for (std::size_t i = 0; i < cloud->size(); ++i) { xyz[3 * i] = cloud->points[i].x; xyz[3 * i + 1] = cloud->points[i].y; xyz[3 * i + 2] = cloud->points[i].z; }
        tree_.Build(xyz.data(), cloud->size(), 3);
    }
    int nearestKSearch(const PointT& p, int k, std::vector<int>& idx, std::vector<float>& d2) const {
        std::vector<flo::KdTree::Hit> hits(static_cast<std::size_t>(k > 0 ? k : 0));
        const float q[3] = {p.x, p.y, p.z};
        const int n = k > 0 ? tree_.Knn(q, k, hits.data()) : 0;
        idx.resize(std::size_t(n)); d2.resize(std::size_t(n));
        for (int i = 0; i < n; ++i) { idx[std::size_t(i)] = hits[std::size_t(i)].idx; d2[std::size_t(i)] = hits[std::size_t(i)].d2; }
        return n;
    }
};
}  // namespace pcl
