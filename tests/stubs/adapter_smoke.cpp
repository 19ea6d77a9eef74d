// Builds the adapter -- against the reference's REAL headers (registration/registration_interface.h, common/data_type.h,
// lidar/pointcloud_cluster.h through the include-shadow shim of oracle/ref_shim) wherever /root/reference exists, against the
// stand-in headers of tests/stubs elsewhere -- and (on a GPU box) drives one registration through it:
// plane z = -1.8 + a wall, scan = the same surface shifted by 5 cm.  Only surface common to both header sets is used.
#ifdef FLS_REAL_REFERENCE_HEADERS
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <execution>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <numeric>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>
#include <Eigen/Dense>
#include <glog/logging.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "common/data_type.h"
#include "lidar/pointcloud_cluster.h"
#endif
#include "fls_hip_registration.h"
#include <cmath>
#include <cstdio>
#include <random>

static PCLPointCloudXYZI make_cloud(int n, double dz, unsigned seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-20.f, 20.f), h(-1.8f, 4.f);
    PCLPointCloudXYZI c;
    for (int i = 0; i < n; ++i) {
        PCLPointXYZI p;
        p.intensity = 0.f;
        if (i % 3 == 0) { p.x = 12.f; p.y = u(rng); p.z = h(rng); }
        else if (i % 3 == 1) { p.x = u(rng); p.y = -9.f; p.z = h(rng); }
        else { p.x = u(rng); p.y = u(rng); p.z = -1.8f; }
        p.z += float(dz);
        c.points.push_back(p);
    }
    return c;
}

int main(int argc, char** argv) {
    if (argc < 2 || fls_device_count() < 1) { std::printf("adapter compiled; no gfx950 device -> not run\n"); return 0; }
    std::shared_ptr<RegistrationInterface> matcher = HipRegistration::PointToPlaneIVOX(0.1, 0.005, 0.001, 10);
    matcher->AddCloudToLocalMap({make_cloud(200000, 0.0, 1)});
    auto cluster = std::make_shared<PointcloudCluster>();
    cluster->planar_cloud_ = make_cloud(20000, -0.05, 2);
    Mat4d T;
    for (int i = 0; i < 16; ++i) T.data()[i] = (i % 5 == 0) ? 1.0 : 0.0;
    const bool ok = matcher->Match(cluster, T);
    std::printf("ok=%d tz=%.4f (expect ~ +0.05) fitness=%g\n", int(ok), T.data()[14], double(matcher->GetFitnessScore(2.0f)));
    return (ok && std::fabs(T.data()[14] - 0.05) < 0.01) ? 0 : 1;
}
