// CPU-side check of device arithmetic that is written as __host__ __device__ code (compiled with hipcc, run on the host):
//   * plane_fit_5x3_bf<double> and plane_fit_5x3_bf<D2> (branch-free, two points per lane: csrc/plane_fit_x2.hpp)
//       == the branchy device routine plane_fit_5x3 (csrc/linalg_dev.hpp) == the oracle's Eigen restatement
//          flo_colpiv_qr_solve_5x3, BIT FOR BIT, on random, near-degenerate and degenerate neighbour sets;
//   * plane_residual_bf<D2> == a straight transcription of kernels_p2plane.hpp::plane_residual_dev (valid flag, J, |d|).
// No HIP runtime call is made.
#include "../../funny_lidar_slam_amd/csrc/plane_fit_x2.hpp"
#include "../../oracle/flo_api.h"
#include <cstdio>
#include <cstring>
#include <random>

using namespace fls;

static bool same_bits(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a != a && b != b); }
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d  %s (case %d)\n", __FILE__, __LINE__, #c, t); return 1; } } while (0)

// transcription of plane_residual_dev (the branchy original) on top of plane_fit_5x3
static bool residual_ref(const double (&A)[3][5], const double (&ps)[3], const double (&pt)[3], const double* T, double thres, double (&J)[6], double& d_abs) {
    double x[3];
    plane_fit_5x3(A, x);
    const double nrm = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    bool ok = true;
    for (int j = 0; j < 5; ++j) {
        const double r = ((A[0][j] * x[0] + A[1][j] * x[1]) + A[2][j] * x[2]) + 1.0;
        if (fabs(r) / nrm > thres) ok = false;
    }
    if (!ok) return false;
    const double n0 = x[0] / nrm, n1 = x[1] / nrm, n2 = x[2] / nrm;
    const double d = ((pt[0] - A[0][0]) * n0 + (pt[1] - A[1][0]) * n1) + (pt[2] - A[2][0]) * n2;
    const double range = sqrt((ps[0] * ps[0] + ps[1] * ps[1]) + ps[2] * ps[2]);
    if (range < 81 * d * d) return false;
    const double s = d > 0 ? 1.0 : -1.0;
    const double v0 = (T[0] * ps[0] + T[4] * ps[1]) + T[8] * ps[2], v1 = (T[1] * ps[0] + T[5] * ps[1]) + T[9] * ps[2], v2 = (T[2] * ps[0] + T[6] * ps[1]) + T[10] * ps[2];
    J[0] = ((0.0 * n0 + (-v2) * n1) + v1 * n2) * s;
    J[1] = ((v2 * n0 + 0.0 * n1) + (-v0) * n2) * s;
    J[2] = (((-v1) * n0 + v0 * n1) + 0.0 * n2) * s;
    J[3] = n0 * s; J[4] = n1 * s; J[5] = n2 * s;
    d_abs = fabs(d);
    return true;
}

int main() {
    std::mt19937_64 rng(20241022);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> un(-1.0, 1.0);
    int n_invalid = 0, n_valid = 0, n_rank_def = 0;
    double prevA[3][5] = {};
    for (int t = 0; t < 60000; ++t) {
        // neighbour sets: noisy planes (the common case), collinear / duplicated / zero points, wild magnitudes
        double A[3][5];
        const int kind = t % 12;
        double nrm[3] = {nd(rng), nd(rng), nd(rng)}, c[3] = {30 * un(rng), 30 * un(rng), 5 * un(rng)};
        const double nl = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        for (int r = 0; r < 5; ++r) {
            double p[3] = {c[0] + 0.4 * un(rng), c[1] + 0.4 * un(rng), c[2] + 0.4 * un(rng)};
            double off = 0.0;
            for (int a = 0; a < 3; ++a) off += (p[a] - c[a]) * nrm[a] / nl;
            const double noise = (kind < 6 ? 0.01 : kind < 8 ? 0.3 : 0.0) * nd(rng);
            for (int a = 0; a < 3; ++a) A[a][r] = double(float(p[a] - (off - noise) * nrm[a] / nl));  // float-valued, like map points
        }
        if (kind == 8) for (int r = 1; r < 5; ++r) for (int a = 0; a < 3; ++a) A[a][r] = A[a][0] * double(r + 1);   // collinear through the origin
        if (kind == 9) for (int r = 1; r < 5; ++r) for (int a = 0; a < 3; ++a) A[a][r] = A[a][0];                   // five copies of one point
        if (kind == 10) for (int r = 0; r < 5; ++r) for (int a = 0; a < 3; ++a) A[a][r] = (r < 4) ? 0.0 : A[a][r];  // zeros
        if (kind == 11) for (int r = 0; r < 5; ++r) A[2][r] = 0.0;                                                  // a zero column
        double xo[3], xb[3], Acm[15], b[5] = {-1, -1, -1, -1, -1}, xr[3];
        for (int a = 0; a < 3; ++a) for (int r = 0; r < 5; ++r) Acm[r + 5 * a] = A[a][r];
        flo_colpiv_qr_solve_5x3(Acm, b, xo);
        plane_fit_5x3(A, xr);
        plane_fit_5x3_bf<double>(A, xb);
        D2 A2[3][5], x2[3];
        for (int a = 0; a < 3; ++a) for (int r = 0; r < 5; ++r) A2[a][r] = D2{A[a][r], prevA[a][r]};
        double xp[3];
        plane_fit_5x3(prevA, xp);
        plane_fit_5x3_bf<D2>(A2, x2);
        for (int a = 0; a < 3; ++a) {
            CHECK(same_bits(xr[a], xo[a]));
            CHECK(same_bits(xb[a], xo[a]));
            CHECK(same_bits(x2[a].a, xo[a]));
            CHECK(same_bits(x2[a].b, xp[a]));
        }
        if (kind >= 8) ++n_rank_def;
        // residual: pose near identity, source point a few metres out
        double T[16] = {0};
        const double yaw = 0.03 * un(rng);
        T[0] = cos(yaw); T[1] = sin(yaw); T[4] = -sin(yaw); T[5] = cos(yaw); T[10] = 1; T[15] = 1; T[12] = 0.2 * un(rng); T[13] = 0.2 * un(rng); T[14] = 0.05 * un(rng);
        double ps[3] = {double(float(A[0][0] + 0.05 * nd(rng))), double(float(A[1][0] + 0.05 * nd(rng))), double(float(A[2][0] + 0.05 * nd(rng)))};
        double pt[3];
        for (int a = 0; a < 3; ++a) pt[a] = double(float(((T[a] * ps[0] + T[4 + a] * ps[1]) + T[8 + a] * ps[2]) + T[12 + a]));
        double Jr[6] = {0}, dr = 0.0;
        const bool okr = residual_ref(A, ps, pt, T, 0.1, Jr, dr);
        D2 ps2[3], pt2[3], J2[6], d2;
        for (int a = 0; a < 3; ++a) { ps2[a] = D2{ps[a], ps[a]}; pt2[a] = D2{pt[a], pt[a]}; }
        for (int a = 0; a < 3; ++a) for (int r = 0; r < 5; ++r) A2[a][r] = D2{A[a][r], A[a][r]};
        const M2 ok2 = plane_residual_bf<D2>(A2, ps2, pt2, T, 0.1, J2, d2);
        CHECK(ok2.a == okr && ok2.b == okr);
        if (okr) {
            ++n_valid;
            for (int a = 0; a < 6; ++a) { CHECK(same_bits(J2[a].a, Jr[a])); CHECK(same_bits(J2[a].b, Jr[a])); }
            CHECK(same_bits(d2.a, dr));
        } else ++n_invalid;
        std::memcpy(prevA, A, sizeof(A));
    }
    std::printf("device math ok: %d valid, %d invalid residuals, %d rank-deficient fits\n", n_valid, n_invalid, n_rank_def);
    return 0;
}
