// features_host.hpp -- host side of the LOAM feature front-end (include/fls_features.h): owns the device buffers,
// launches kernels_features.hpp, keeps the PointcloudCluster fields the reference's classes fill.
#pragma once
#include "host_maps.hpp"
#include "kernels_features.hpp"
#include "../../include/fls_features.h"
#include <limits>

struct fls_features {
    fls_feature_params p{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    fls::FeatParamsDev pd{};
    fls::RawLayoutDev layout{};
    size_t n_raw = 0;
    int N = 0;
    bool projected = false, extracted = false;
    double project_ms = 0.0, extract_ms = 0.0;

    fls::DevBuf<unsigned char> d_raw, d_valid, d_valid_pre, d_is_corner;
    // introspection arrays (raw_index, roughness, the three flag arrays) come down only when somebody asks for them
    bool have_raw_index = false, have_intro = false;
    fls::DevBuf<unsigned> d_owner;
    fls::DevBuf<int> d_row_count, d_row_start, d_row_end, d_n, d_col, d_raw_index, d_corner_idx, d_corner_cnt, d_planar_idx, d_planar_cnt;
    fls::DevBuf<float4> d_ordered;
    fls::DevBuf<float> d_depth, d_rough;

    std::vector<fls::Pt4> ordered;  // 16-byte xyzi rows (Pt4 reused as a container: the int member holds the intensity bits)
    std::vector<float> depth, rough;
    std::vector<int> col, raw_index, row_start, row_end, corner_idx, planar_idx;
    std::vector<unsigned char> valid_pre, valid_post, is_corner;
    std::vector<fls::PtI> corner, planar, corner_f, planar_f;
    std::vector<int> h_cidx, h_ccnt, h_pidx, h_pcnt;

    ~fls_features() {
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    }

    fls_status init() {
        const float fmax = std::numeric_limits<float>::max();
        const int imax = std::numeric_limits<int>::max();
        if (p.struct_size != sizeof(fls_feature_params)) return FLS_ERR_INVALID;
        if (p.lidar_vertical_scan == imax || p.lidar_horizontal_scan == imax || p.lidar_horizontal_resolution == fmax || p.min_distance == fmax ||
            p.max_distance == fmax || p.corner_thres == fmax || p.planar_thres == fmax)
            return FLS_ERR_INVALID;  // CHECK_NE(..., NaN) of both constructors
        if (p.lidar_vertical_scan <= 0 || p.lidar_horizontal_scan <= 0 || p.lidar_horizontal_scan > fls::kFeatMaxCols ||
            (p.lidar_horizontal_scan - 11) / 6 > fls::kFeatMaxSector || !(p.lidar_horizontal_resolution > 0.f))
            return FLS_ERR_INVALID;
        pd = fls::FeatParamsDev{p.lidar_vertical_scan, p.lidar_horizontal_scan, p.lidar_horizontal_resolution, p.min_distance, p.max_distance,
                                p.corner_thres, p.planar_thres};
        FLS_HIP(hipSetDevice(device));
        FLS_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (auto& e : ev) FLS_HIP(hipEventCreate(&e));
        const size_t cells = size_t(pd.rows) * size_t(pd.cols);
        d_owner.reserve(cells);
        d_row_count.reserve(size_t(pd.rows)); d_row_start.reserve(size_t(pd.rows)); d_row_end.reserve(size_t(pd.rows));
        d_n.reserve(1);
        d_ordered.reserve(cells); d_depth.reserve(cells); d_rough.reserve(cells); d_col.reserve(cells); d_raw_index.reserve(cells);
        d_valid.reserve(cells); d_valid_pre.reserve(cells); d_is_corner.reserve(cells);
        d_corner_idx.reserve(size_t(pd.rows) * 120); d_corner_cnt.reserve(size_t(pd.rows));
        d_planar_idx.reserve(size_t(pd.rows) * size_t(pd.cols + 6)); d_planar_cnt.reserve(size_t(pd.rows));
        return FLS_OK;
    }

    fls_status project(const void* raw, size_t n, const fls_point_layout& L, size_t* n_ordered) {
        if (L.stride_bytes < 16 || (L.stride_bytes & 3u) || (L.xyz_offset & 3u) || (L.intensity_offset & 3u) || (L.ring_offset & 1u) ||
            L.xyz_offset + 12 > L.stride_bytes || L.intensity_offset + 4 > L.stride_bytes || L.ring_offset + 2 > L.stride_bytes ||
            n > 0xFFFFFFF0ull)
            return FLS_ERR_INVALID;
        layout = fls::RawLayoutDev{L.stride_bytes, L.xyz_offset, L.intensity_offset, L.ring_offset};
        n_raw = n;
        projected = extracted = false;
        const size_t cells = size_t(pd.rows) * size_t(pd.cols);
        d_raw.reserve(std::max<size_t>(n * L.stride_bytes, 16));
        FLS_HIP(hipEventRecord(ev[0], stream));
        if (n) FLS_HIP(hipMemcpyAsync(d_raw.p, raw, n * L.stride_bytes, hipMemcpyHostToDevice, stream));
        FLS_HIP(hipMemsetAsync(d_owner.p, 0xFF, cells * sizeof(unsigned), stream));
        if (n)
            hipLaunchKernelGGL(fls::feat_project_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, stream, d_raw.p, unsigned(n), layout, pd,
                               d_owner.p);
        hipLaunchKernelGGL(fls::feat_count_kernel, dim3(unsigned(pd.rows)), dim3(256), 0, stream, d_owner.p, pd, d_row_count.p);
        hipLaunchKernelGGL(fls::feat_compact_kernel, dim3(unsigned(pd.rows)), dim3(256), 0, stream, d_raw.p, layout, d_owner.p, pd, d_row_count.p,
                           d_ordered.p, d_depth.p, d_col.p, d_raw_index.p, d_valid.p, d_row_start.p, d_row_end.p, d_n.p);
        FLS_HIP(hipGetLastError());
        FLS_HIP(hipEventRecord(ev[1], stream));
        FLS_HIP(hipMemcpyAsync(&N, d_n.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        row_start.resize(size_t(pd.rows)); row_end.resize(size_t(pd.rows));
        FLS_HIP(hipMemcpyAsync(row_start.data(), d_row_start.p, size_t(pd.rows) * sizeof(int), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipMemcpyAsync(row_end.data(), d_row_end.p, size_t(pd.rows) * sizeof(int), hipMemcpyDeviceToHost, stream));
        FLS_HIP(hipStreamSynchronize(stream));
        const size_t m = size_t(N);
        ordered.resize(m); depth.resize(m); col.resize(m);
        have_raw_index = have_intro = false;
        if (m) {
            FLS_HIP(hipMemcpyAsync(ordered.data(), d_ordered.p, m * sizeof(float4), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(depth.data(), d_depth.p, m * sizeof(float), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(col.data(), d_col.p, m * sizeof(int), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
        }
        float ms = 0.f;
        FLS_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        project_ms = ms;
        projected = true;
        if (n_ordered) *n_ordered = m;
        return FLS_OK;
    }

    fls_status extract(size_t* n_corner, size_t* n_planar) {
        if (!projected) return FLS_ERR_STATE;
        const size_t m = size_t(N);
        have_intro = false;
        corner_idx.clear(); planar_idx.clear(); corner.clear(); planar.clear(); corner_f.clear(); planar_f.clear();
        if (N >= 12) {
            const size_t rows = size_t(pd.rows);
            h_cidx.resize(rows * 120); h_ccnt.resize(rows); h_pidx.resize(rows * size_t(pd.cols + 6)); h_pcnt.resize(rows);
            FLS_HIP(hipEventRecord(ev[2], stream));
            hipLaunchKernelGGL(fls::feat_valid_rough_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, stream, d_n.p, d_depth.p, d_col.p, d_rough.p,
                               d_valid.p);
            FLS_HIP(hipMemcpyAsync(d_valid_pre.p, d_valid.p, m, hipMemcpyDeviceToDevice, stream));  // snapshot before SelectFeatures edits the flags
            hipLaunchKernelGGL(fls::feat_select_kernel, dim3(unsigned(pd.rows)), dim3(fls::kFeatSelectThreads), 0, stream, d_n.p, pd, d_row_start.p, d_row_end.p, d_rough.p,
                               d_col.p, d_valid.p, d_is_corner.p, d_corner_idx.p, d_corner_cnt.p, d_planar_idx.p, d_planar_cnt.p);
            FLS_HIP(hipGetLastError());
            FLS_HIP(hipEventRecord(ev[3], stream));
            FLS_HIP(hipMemcpyAsync(h_ccnt.data(), d_corner_cnt.p, rows * sizeof(int), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(h_pcnt.data(), d_planar_cnt.p, rows * sizeof(int), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(h_cidx.data(), d_corner_idx.p, h_cidx.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(h_pidx.data(), d_planar_idx.p, h_pidx.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
            float ms = 0.f;
            FLS_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
            extract_ms = ms;
            // corner_cloud_ grows ring by ring, sector by sector (:158-160); planar_cloud_ += per-ring cloud (:220)
            for (size_t r = 0; r < rows; ++r) {
                corner_idx.insert(corner_idx.end(), h_cidx.begin() + r * 120, h_cidx.begin() + r * 120 + h_ccnt[r]);
                planar_idx.insert(planar_idx.end(), h_pidx.begin() + r * size_t(pd.cols + 6), h_pidx.begin() + r * size_t(pd.cols + 6) + h_pcnt[r]);
            }
            auto gather = [&](const std::vector<int>& idx, std::vector<fls::PtI>& out) {
                out.resize(idx.size());
                for (size_t k = 0; k < idx.size(); ++k) std::memcpy(&out[k], &ordered[size_t(idx[k])], sizeof(fls::PtI));
            };
            gather(corner_idx, corner);
            gather(planar_idx, planar);
            if (p.corner_voxel_filter_size > 0.f) corner_f = fls::voxel_grid(corner, p.corner_voxel_filter_size);  // preprocessing.cpp:234-235
            if (p.planar_voxel_filter_size > 0.f) planar_f = fls::voxel_grid(planar, p.planar_voxel_filter_size);  // :236-237
        }
        extracted = true;
        if (n_corner) *n_corner = corner.size();
        if (n_planar) *n_planar = planar.size();
        return FLS_OK;
    }

    void fetch_raw_index() {
        if (have_raw_index) return;
        const size_t m = size_t(N);
        raw_index.resize(m);
        if (m && projected) {
            FLS_HIP(hipSetDevice(device));
            FLS_HIP(hipMemcpyAsync(raw_index.data(), d_raw_index.p, m * sizeof(int), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
        }
        have_raw_index = true;
    }
    void fetch_intro() {
        if (have_intro) return;
        const size_t m = size_t(N);
        rough.assign(m, 0.f); valid_pre.assign(m, 1); valid_post.assign(m, 1); is_corner.assign(m, 0);
        if (m && extracted && N >= 12) {
            FLS_HIP(hipSetDevice(device));
            FLS_HIP(hipMemcpyAsync(rough.data(), d_rough.p, m * sizeof(float), hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(valid_pre.data(), d_valid_pre.p, m, hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(valid_post.data(), d_valid.p, m, hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipMemcpyAsync(is_corner.data(), d_is_corner.p, m, hipMemcpyDeviceToHost, stream));
            FLS_HIP(hipStreamSynchronize(stream));
        }
        have_intro = true;
    }
    template <class T>
    static size_t copy_out(const std::vector<T>& v, void* out, size_t cap) {
        if (out && !v.empty()) std::memcpy(out, v.data(), std::min(cap, v.size()) * sizeof(T));
        return v.size();
    }
    size_t get(int what, void* out, size_t cap) {
        if (what == FLS_FEAT_RAW_INDEX) fetch_raw_index();
        if (what == FLS_FEAT_IS_CORNER || what == FLS_FEAT_ROUGHNESS || what == FLS_FEAT_VALID_PRE || what == FLS_FEAT_VALID_POST) fetch_intro();
        switch (what) {
            case FLS_FEAT_ORDERED: return copy_out(ordered, out, cap);
            case FLS_FEAT_DEPTH: return copy_out(depth, out, cap);
            case FLS_FEAT_COL: return copy_out(col, out, cap);
            case FLS_FEAT_ROW_START: return copy_out(row_start, out, cap);
            case FLS_FEAT_ROW_END: return copy_out(row_end, out, cap);
            case FLS_FEAT_CORNER: return copy_out(corner, out, cap);
            case FLS_FEAT_PLANAR: return copy_out(planar, out, cap);
            case FLS_FEAT_IS_CORNER: return copy_out(is_corner, out, cap);
            case FLS_FEAT_ROUGHNESS: return copy_out(rough, out, cap);
            case FLS_FEAT_VALID_PRE: return copy_out(valid_pre, out, cap);
            case FLS_FEAT_VALID_POST: return copy_out(valid_post, out, cap);
            case FLS_FEAT_CORNER_IDX: return copy_out(corner_idx, out, cap);
            case FLS_FEAT_PLANAR_IDX: return copy_out(planar_idx, out, cap);
            case FLS_FEAT_RAW_INDEX: return copy_out(raw_index, out, cap);
            case FLS_FEAT_CORNER_FILTERED: return copy_out(corner_f, out, cap);
            case FLS_FEAT_PLANAR_FILTERED: return copy_out(planar_f, out, cap);
            default: return 0;
        }
    }
};
