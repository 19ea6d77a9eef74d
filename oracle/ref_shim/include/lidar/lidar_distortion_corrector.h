// oracle/ref_shim: stand-in for the reference's include/lidar/lidar_distortion_corrector.h (TEST INFRASTRUCTURE ONLY).
// De-skewing (IMU interpolation per point, SURVEY.md section 2 "de-skew": out of scope) is the one reference header on
// the compiled path that is SHADOWED rather than compiled: PointcloudProjector::Project calls ProcessPoint() for every
// return; here it is the identity (a static sensor), which is also what the oracle and the HIP front-end assume.
#pragma once
#include "common/data_type.h"
#include <memory>

class LidarDistortionCorrector {
public:
    LidarDistortionCorrector() = default;
    bool ProcessPoint(float x, float y, float z, float& xc, float& yc, float& zc, float /*relative_time*/) {
        xc = x; yc = y; zc = z;
        return true;
    }
};
