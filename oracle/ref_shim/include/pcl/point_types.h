// oracle/ref_shim: minimal pcl point types (TEST INFRASTRUCTURE ONLY).  PCL is not in this container; the layouts
// below are the published PCL 1.10 ones (pcl/impl/point_types.hpp): 16-byte xyz + padding, intensity in float 4.
#pragma once
#include <cstdint>
#include <Eigen/Core>

#define PCL_ADD_POINT4D                                                                    \
    union EIGEN_ALIGN16 { float data[4]; struct { float x; float y; float z; }; };         \
    inline Eigen::Vector3f getVector3fMap() const { return Eigen::Vector3f(x, y, z); }
#define PCL_ADD_INTENSITY union { struct { float intensity; }; float data_c[4]; };
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, ...)

namespace pcl {
struct EIGEN_ALIGN16 PointXYZ {
    PCL_ADD_POINT4D
    PointXYZ() : PointXYZ(0.f, 0.f, 0.f) {}
    PointXYZ(float _x, float _y, float _z) { x = _x; y = _y; z = _z; data[3] = 1.0f; }
};
struct EIGEN_ALIGN16 PointXYZI {
    PCL_ADD_POINT4D
    PCL_ADD_INTENSITY
    PointXYZI() { x = y = z = 0.0f; data[3] = 1.0f; intensity = 0.0f; data_c[1] = data_c[2] = data_c[3] = 0.0f; }
};
struct PointXY { float x = 0.f, y = 0.f; };
}  // namespace pcl
