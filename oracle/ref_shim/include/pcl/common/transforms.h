// oracle/ref_shim: pcl::transformPoint / pcl::transformPointCloud (TEST INFRASTRUCTURE ONLY).
// PCL >= 1.9 pcl/common/impl/transforms.hpp, Transformer<Scalar>::se3 without the AVX specialisation (the reference is
// built without -mavx): every output coordinate is ((m(r,0) x + m(r,1) y) + m(r,2) z) + m(r,3) evaluated in `Scalar`
// and cast to float; all other fields are copied.  Same arithmetic as oracle/flo_common.h::transform_point_d.
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

namespace pcl {
template <class PointT, class Scalar>
inline PointT transformPoint(const PointT& p, const Eigen::Transform<Scalar, 3, Eigen::Affine>& tf) {
    const auto& m = tf.matrix();
    PointT r = p;
    const Scalar x = Scalar(p.x), y = Scalar(p.y), z = Scalar(p.z);
    r.x = float(((m(0, 0) * x + m(0, 1) * y) + m(0, 2) * z) + m(0, 3));
    r.y = float(((m(1, 0) * x + m(1, 1) * y) + m(1, 2) * z) + m(1, 3));
    r.z = float(((m(2, 0) * x + m(2, 1) * y) + m(2, 2) * z) + m(2, 3));
    return r;
}
template <class PointT, class Scalar>
inline void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix<Scalar, 4, 4>& T) {
    const Eigen::Transform<Scalar, 3, Eigen::Affine> tf(T);
    if (&in != &out) { out.header = in.header; out.is_dense = in.is_dense; out.points.resize(in.points.size()); out.width = in.width; out.height = in.height; }
    for (std::size_t i = 0; i < in.points.size(); ++i) out.points[i] = transformPoint(in.points[i], tf);
}
}  // namespace pcl
