// device_voxelgrid.hpp -- host driver of the device VoxelGrid (kernels_voxelgrid.hpp; contract there).
//
// Input: x | y | z | intensity of n points, resident on the device.  Output: the filtered cloud as x | y | z | intensity
// on the device (ascending leaf index), its size on the host.  Two short host waits per call: the bounds (the leaf box
// decides the number of radix passes and the "leaf size too small" refusal, voxel_grid.hpp:69-74) and the output size.
#pragma once
#include "kernels_voxelgrid.hpp"
#include "matcher_base.hpp"
#include <cstdlib>
#include <climits>

namespace fls {

// stable LSD radix sort of {key, value} pairs (8 bits per pass; values keep their input order among equal keys)
struct DevicePairSort {
    DevBuf<unsigned> keys, vals;  // ping | pong
    DevBuf<unsigned> hist;        // tile counts | scanned
    DevBuf<unsigned> dig_tot;     // [pass][256] keys per digit
    size_t n = 0;
    unsigned* k0 = nullptr;       // input before run(), sorted result after
    unsigned* v0 = nullptr;
    void prepare(size_t count) {
        n = count;
        keys.reserve(2 * n);
        vals.reserve(2 * n);
        k0 = keys.p;
        v0 = vals.p;
    }
    static int passes_for(unsigned long long max_key) {
        int bits = 1;
        while ((1ull << bits) <= max_key) ++bits;
        return (bits + 7) / 8;
    }
    void run(int passes, hipStream_t s) {
        const int ni = int(n), nb = (ni + kVgTile - 1) / kVgTile;
        hist.reserve(size_t(512) * nb);
        dig_tot.reserve(4 * 256);
        FLS_HIP(hipMemsetAsync(dig_tot.p, 0, 4 * 256 * sizeof(unsigned), s));
        unsigned* k1 = k0 == keys.p ? keys.p + n : keys.p;
        unsigned* v1 = v0 == vals.p ? vals.p + n : vals.p;
        for (int pass = 0; pass < passes; ++pass) {
            hipLaunchKernelGGL(vg_hist, dim3(unsigned(nb)), dim3(kVgBlock), 0, s, (const unsigned*)k0, ni, pass * 8, hist.p, nb, dig_tot.p + 256 * pass);
            hipLaunchKernelGGL(vg_scan_rows, dim3(256), dim3(kVgScanBlock), 0, s, (const unsigned*)hist.p, hist.p + 256 * nb, nb,
                               (const unsigned*)(dig_tot.p + 256 * pass));
            hipLaunchKernelGGL(vg_scatter, dim3(unsigned(nb)), dim3(kVgBlock), 0, s, (const unsigned*)k0, (const unsigned*)v0, k1, v1, ni,
                               pass * 8, (const unsigned*)(hist.p + 256 * nb), nb);
            std::swap(k0, k1);
            std::swap(v0, v1);
        }
    }
};

struct DeviceVoxelGrid {
    DevicePairSort sort;
    DevBuf<unsigned> lx, bt;        // bt: block totals | scanned
    DevBuf<float4> sorted;          // the points in sorted order
    DevBuf<float> out;            // x | y | z | i, capacity n each
    DevBuf<VgHeader> d_hdr;
    PinnedBuf<VgHeader> h_hdr;    // [0] = init template, [1] = read-back
    size_t n_in = 0, n_out = 0;
    int last_passes = 0;
    const float* ox() const { return out.p; }
    const float* oy() const { return out.p + n_in; }
    const float* oz() const { return out.p + 2 * n_in; }
    const float* oi() const { return out.p + 3 * n_in; }

    // false: not applicable (empty, too large, or PCL's "leaf size too small" case) -- the caller takes the host path
    bool run(const float* x, const float* y, const float* z, const float* in, size_t n, float leaf, hipStream_t s) {
        n_in = n;
        n_out = 0;
        if (n == 0 || n > size_t(kVgMaxBlocks) * kVgTile) return false;
        h_hdr.reserve(2);
        d_hdr.reserve(1);
        for (int a = 0; a < 3; ++a) { h_hdr.p[0].mn[a] = 0xffffffffu; h_hdr.p[0].mx[a] = 0u; }
        h_hdr.p[0].n_out = 0u; h_hdr.p[0].pad = 0u;
        FLS_HIP(hipMemcpyAsync(d_hdr.p, &h_hdr.p[0], sizeof(VgHeader), hipMemcpyHostToDevice, s));
        const int ni = int(n);
        const int nb1 = (ni + kVgBlock - 1) / kVgBlock, nb2 = (ni + kVgScanBlock - 1) / kVgScanBlock;
        hipLaunchKernelGGL(vg_minmax, dim3(unsigned(std::min(nb1, 48))), dim3(kVgBlock), 0, s, x, y, z, ni, d_hdr.p);  // few blocks: six header atomics each
        FLS_HIP(hipMemcpyAsync(&h_hdr.p[1], d_hdr.p, sizeof(VgHeader), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        const VgHeader& hh = h_hdr.p[1];
        if (hh.mn[0] == 0xffffffffu) return false;  // no finite point: the host path returns the empty cloud
        const float inv = 1.0f / leaf;
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) { mn[a] = vg_unord(hh.mn[a]); mx[a] = vg_unord(hh.mx[a]); }
        const long long dx = (long long)((mx[0] - mn[0]) * inv) + 1, dy = (long long)((mx[1] - mn[1]) * inv) + 1,
                        dz = (long long)((mx[2] - mn[2]) * inv) + 1;
        if (dx * dy * dz > (long long)INT_MAX) return false;
        VgGrid g;
        g.inv = inv;
        long long div_b[3];
        for (int a = 0; a < 3; ++a) {
            g.min_b[a] = int(std::floor(mn[a] * inv));
            div_b[a] = (long long)int(std::floor(mx[a] * inv)) - g.min_b[a] + 1;
        }
        const long long total = div_b[0] * div_b[1] * div_b[2];
        if (total <= 0 || total >= (long long)INT_MAX) return false;
        g.m1 = int(div_b[0]);
        g.m2 = int(div_b[0] * div_b[1]);
        g.total = unsigned(total);
        const int passes = DevicePairSort::passes_for((unsigned long long)total);  // the sentinel `total` itself must sort
        last_passes = passes;

        sort.prepare(n);
        lx.reserve(n);
        bt.reserve(size_t(2 * nb2));
        sorted.reserve(n);
        out.reserve(4 * n);
        hipLaunchKernelGGL(vg_index, dim3(unsigned(nb1)), dim3(kVgBlock), 0, s, x, y, z, ni, g, sort.k0, sort.v0);
        sort.run(passes, s);
        unsigned* const k0 = sort.k0;
        unsigned* const v0 = sort.v0;
        hipLaunchKernelGGL(vg_heads, dim3(unsigned(nb2)), dim3(kVgScanBlock), 0, s, (const unsigned*)k0, (const unsigned*)v0, ni, g.total, x, y, z, in,
                           sorted.p, lx.p, bt.p);
        hipLaunchKernelGGL(vg_scan, dim3(1), dim3(kVgScanBlock), 0, s, (const unsigned*)bt.p, bt.p + nb2, nb2, &d_hdr.p->n_out);
        hipLaunchKernelGGL(vg_centroid, dim3(unsigned(nb1)), dim3(kVgBlock), 0, s, (const unsigned*)k0, (const float4*)sorted.p, ni,
                           (const unsigned*)lx.p, (const unsigned*)(bt.p + nb2), out.p, out.p + n, out.p + 2 * n, out.p + 3 * n);
        FLS_HIP(hipMemcpyAsync(&h_hdr.p[1], d_hdr.p, sizeof(VgHeader), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        FLS_HIP(hipGetLastError());
        n_out = h_hdr.p[1].n_out;
        return true;
    }

    // the filtered cloud on the host (map updates keep their clouds there)
    std::vector<PtI> download(hipStream_t s, std::vector<float>& tmp) const {
        std::vector<PtI> c(n_out);
        if (n_out == 0) return c;
        tmp.resize(4 * n_out);
        for (int a = 0; a < 4; ++a)
            FLS_HIP(hipMemcpyAsync(tmp.data() + size_t(a) * n_out, out.p + size_t(a) * n_in, n_out * sizeof(float), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < n_out; ++i) c[i] = PtI{tmp[i], tmp[n_out + i], tmp[2 * n_out + i], tmp[3 * n_out + i]};
        return c;
    }
};

// the source-scan filter of the kd-tree / NDT matchers (icp_optimized.h:57, incremental_ndt.h:232): host std::sort path by
// default, the device path with FLS_DEVICE_VOXELGRID=1.  With the device path the filtered cloud stays on the device; the
// host copy is fetched only when a map update needs it.
struct SourceFilter {
    bool on_device = false;
    DevScan raw;
    DeviceVoxelGrid vg;
    bool resident = false;  // `source` has not been downloaded yet
    std::vector<float> tmp;
    unsigned long long device_runs = 0, host_runs = 0;
    void init() {
        if (const char* e = std::getenv("FLS_DEVICE_VOXELGRID")) on_device = std::atoi(e) != 0;
    }
    void filter(const float* s0, size_t n0, int stride, float leaf, hipStream_t s, DevScan& scan, std::vector<PtI>& source) {
        resident = false;
        if (on_device && n0 != 0) {
            raw.upload_raw(s0, n0, stride, s, true);
            if (vg.run(raw.x.p, raw.y.p, raw.z.p, raw.xyz.p + 3 * n0, n0, leaf, s)) {
                scan.n = vg.n_out;
                scan.host.clear();
                scan.x.p = const_cast<float*>(vg.ox());
                scan.y.p = const_cast<float*>(vg.oy());
                scan.z.p = const_cast<float*>(vg.oz());
                source.clear();
                resident = true;
                ++device_runs;
                return;
            }
        }
        source = voxel_grid_strided(s0, n0, stride, leaf);
        scan.upload(source, s);
        ++host_runs;
    }
    void materialize(hipStream_t s, std::vector<PtI>& source) {
        if (!resident) return;
        source = vg.download(s, tmp);
        resident = false;
    }
};

}  // namespace fls
