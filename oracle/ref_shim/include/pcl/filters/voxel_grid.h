// oracle/ref_shim: pcl::VoxelGrid<PointXYZI>::filter (TEST INFRASTRUCTURE ONLY) -- forwarded to
// oracle/flo_common.h::voxel_grid, the oracle's stand-in for PCL 1.10 voxel_grid.hpp applyFilter (third-party, not in
// /root/reference): leaf index from floor(p * inv_leaf) - min_b, std::sort by index, one centroid (xyz + intensity, float
// accumulation) per leaf in ascending index order; input returned unchanged when the index would overflow int32.
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "flo_common.h"

namespace pcl {
template <class PointT>
class VoxelGrid {
    float leaf_ = 0.f;
    typename PointCloud<PointT>::ConstPtr input_;
public:
    void setLeafSize(float lx, float, float) { leaf_ = lx; }
    void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
    void filter(PointCloud<PointT>& out) {
        flo::Cloud in(input_->size());
        for (std::size_t i = 0; i < in.size(); ++i) in[i] = flo::P4{input_->points[i].x, input_->points[i].y, input_->points[i].z, input_->points[i].intensity};
        const flo::Cloud f = flo::voxel_grid(in, leaf_);
        PointCloud<PointT> res;  // the output may be the input object (loam_full_kdtree.h:93-99): build aside, then move in
        res.header = input_->header;
        res.points.resize(f.size());
        for (std::size_t i = 0; i < f.size(); ++i) { PointT p; p.x = f[i].x; p.y = f[i].y; p.z = f[i].z; p.intensity = f[i].i; res.points[i] = p; }
        res.width = std::uint32_t(f.size()); res.height = 1; res.is_dense = true;
        out = res;
    }
};
}  // namespace pcl
