// Builds include/fls_hip_features.h against the stub headers and (on a GPU box, given a raw-cloud file) runs
// Project + ExtractFeatures through it, writing the three clouds for the Python test to compare.
#include "registration/registration_interface.h"
#include "fls_hip_features.h"
#include <cstdio>
#include <cstring>

static bool dump(const char* path, const PCLPointCloudXYZI& c) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    for (const auto& p : c.points) { const float row[4] = {p.x, p.y, p.z, p.intensity}; std::fwrite(row, sizeof(row), 1, f); }
    std::fclose(f);
    return true;
}

int main(int argc, char** argv) {
    if (argc < 3 || fls_device_count() < 1) { std::printf("features adapter compiled; no gfx950 device or no input -> not run\n"); return 0; }
    PointcloudCluster c;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    PointXYZIRT p;
    while (std::fread(&p, sizeof(p), 1, f) == 1) c.raw_cloud_.points.push_back(p);
    std::fclose(f);
    loam::HipFeatureFrontEnd fe(1800, 64, 0.2f / 180.0f * 3.14159265358979323846f, 4.0f, 100.0f, 1.0f, 0.1f);
    if (!fe.Project(c) || !fe.ExtractFeatures(c)) return 3;
    char path[1024];
    std::snprintf(path, sizeof(path), "%s.ordered", argv[2]); dump(path, c.ordered_cloud_);
    std::snprintf(path, sizeof(path), "%s.corner", argv[2]); dump(path, c.corner_cloud_);
    std::snprintf(path, sizeof(path), "%s.planar", argv[2]); dump(path, c.planar_cloud_);
    std::printf("raw=%zu ordered=%zu corner=%zu planar=%zu rows=%zu depth0=%g\n", c.raw_cloud_.size(), c.ordered_cloud_.size(), c.corner_cloud_.size(),
                c.planar_cloud_.size(), c.row_start_index_vec_.size(), c.point_depth_vec_.empty() ? 0.0 : double(c.point_depth_vec_[0]));
    return 0;
}
