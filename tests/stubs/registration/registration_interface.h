// Minimal stand-ins for the reference headers the adapter includes, so that include/fls_hip_registration.h
// can be syntax- and type-checked in a container without Eigen / PCL / ROS.  TEST SCAFFOLDING ONLY: the
// shapes mirror include/common/data_type.h:26-27,55, include/lidar/pointcloud_cluster.h:13-87 and
// include/registration/registration_interface.h:11-20 of the reference.
#pragma once
#include <cstddef>
#include <initializer_list>
#include <memory>
#include <vector>

struct alignas(16) PCLPointXYZI {  // pcl::PointXYZI: 32 bytes
    float x, y, z, pad0;
    float intensity, pad1, pad2, pad3;
};
static_assert(sizeof(PCLPointXYZI) == 32, "pcl::PointXYZI layout");
struct PCLPointCloudXYZI {
    std::vector<PCLPointXYZI> points;
    std::size_t size() const { return points.size(); }
};
struct Mat4d {  // Eigen::Matrix4d: 16 doubles, column-major
    double m[16];
    double* data() { return m; }
    const double* data() const { return m; }
};
struct alignas(16) PointXYZIRT {  // include/lidar/lidar_point_type.h VelodynePointXYZIRT: 32 bytes
    float x, y, z, pad0;
    float intensity;
    unsigned short ring;
    float time;
};
static_assert(sizeof(PointXYZIRT) == 32, "VelodynePointXYZIRT layout");
struct PCLPointCloudXYZIRT {
    std::vector<PointXYZIRT> points;
    std::size_t size() const { return points.size(); }
};
struct PointcloudCluster {  // include/lidar/pointcloud_cluster.h:13-85 (the members the registration path and the LOAM front-end touch)
    PCLPointCloudXYZIRT raw_cloud_;
    PCLPointCloudXYZI ordered_cloud_, corner_cloud_, planar_cloud_;
    std::vector<float> point_depth_vec_;
    std::vector<int> point_col_index_vec_, row_start_index_vec_, row_end_index_vec_;
};
typedef std::shared_ptr<PointcloudCluster> PointcloudClusterPtr;

class RegistrationInterface {
public:
    virtual bool Match(const PointcloudClusterPtr& source_cloud_cluster, Mat4d& T) = 0;
    virtual ~RegistrationInterface() = default;
    virtual void AddCloudToLocalMap(const std::initializer_list<PCLPointCloudXYZI>& cloud_list) = 0;
    [[nodiscard]] virtual float GetFitnessScore(float max_range) const = 0;
};
