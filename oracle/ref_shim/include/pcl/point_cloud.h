// oracle/ref_shim: minimal pcl::PointCloud (TEST INFRASTRUCTURE ONLY), PCL 1.10 surface used by the reference's
// registration path: points / size / resize / clear / push_back / emplace_back / at / [] / += / makeShared / Ptr.
#pragma once
#include <memory>
#include <vector>
#include <cstdint>
#include <string>
#include <Eigen/Core>
#include <Eigen/Geometry>

namespace pcl {
struct PCLHeader { std::uint32_t seq = 0; std::uint64_t stamp = 0; std::string frame_id; };

template <class PointT>
class PointCloud {
public:
    using VectorType = std::vector<PointT, Eigen::aligned_allocator<PointT>>;
    using Ptr = std::shared_ptr<PointCloud<PointT>>;       // boost::shared_ptr in PCL 1.10: same surface
    using ConstPtr = std::shared_ptr<const PointCloud<PointT>>;
    using iterator = typename VectorType::iterator;
    using const_iterator = typename VectorType::const_iterator;
    using value_type = PointT;

    PCLHeader header;
    VectorType points;
    std::uint32_t width = 0, height = 0;
    bool is_dense = true;
    Eigen::Vector4f sensor_origin_ = Eigen::Vector4f::Zero();
    Eigen::Quaternionf sensor_orientation_ = Eigen::Quaternionf::Identity();

    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void reserve(std::size_t n) { points.reserve(n); }
    void resize(std::size_t n) { points.resize(n); if (width * height != n) { width = std::uint32_t(n); height = 1; } }
    void clear() { points.clear(); width = 0; height = 0; }
    void push_back(const PointT& p) { points.push_back(p); width = std::uint32_t(points.size()); height = 1; }
    template <class... A> PointT& emplace_back(A&&... a) { points.emplace_back(std::forward<A>(a)...); width = std::uint32_t(points.size()); height = 1; return points.back(); }
    const PointT& at(std::size_t i) const { return points.at(i); }
    PointT& at(std::size_t i) { return points.at(i); }
    const PointT& operator[](std::size_t i) const { return points[i]; }
    PointT& operator[](std::size_t i) { return points[i]; }
    iterator begin() { return points.begin(); }
    iterator end() { return points.end(); }
    const_iterator begin() const { return points.begin(); }
    const_iterator end() const { return points.end(); }
    PointCloud& operator+=(const PointCloud& o) {  // PCL 1.10: appends the points, keeps the newer stamp
        if (o.header.stamp > header.stamp) header.stamp = o.header.stamp;
        const std::size_t n = points.size();
        points.resize(n + o.points.size());
        for (std::size_t i = 0; i < o.points.size(); ++i) points[n + i] = o.points[i];
        width = std::uint32_t(points.size()); height = 1;
        is_dense = is_dense && o.is_dense;
        return *this;
    }
    Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl
