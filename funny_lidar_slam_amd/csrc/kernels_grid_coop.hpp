// kernels_grid_coop.hpp -- cooperative exact k-NN on the uniform cell grid + the per-kind fit kernels of
// the three kinds whose reference uses pcl::KdTreeFLANN (IcpOptimized, LoamFull, LoamPointToPlaneKdtree).
//
//   grid_knn_kernel<K, FLOAT_XFORM>   8 lanes cooperate on one source point: the 27 cells around the query are
//                        dealt round-robin to the lanes (hash probes in flight together), each lane scans the
//                        points of its cells 4 loads at a time into a private top-K of 64-bit keys
//                        {float-bits(d2) : map index} (ties resolve to the lower map index, the rule the CPU
//                        oracle fixes for FLANN's traversal-defined order), then K rounds of DPP group-min.
//                        Exactness: see kernels_knn.hpp (cell >= sqrt(gate) => every point within the gate lies
//                        in the 27 cells).  The un-gated kind falls back to the serial ring search when the
//                        K-th neighbour is not yet certain after the 27 cells.
//                        (The first version ran the whole ring search in one lane per point: 797 us per launch
//                        on the 1e6-point LOAM map.)
//   icp_fit_kernel       IcpOptimized per-point lambda            icp_optimized.h:87-108
//   plane_fit_kernel     LoamFull::PlanarMatch / LoamPointToPlaneKdtree::PlanerMatch (after the k-NN)
//                                                                 loam_full_kdtree.h:291-343, loam_point_to_plane_kdtree.h:219-271
//   corner_fit_kernel    LoamFull::CornerMatch (after the k-NN)   loam_full_kdtree.h:227-271
// Fit kernels: one lane per point, FP64, DPP wave reduction, one partial row per 256-thread workgroup; the
// Gauss-Newton tail runs in gn_solve_loam_kernel / gn_solve_lu_kernel.
#pragma once
#include "kernels_knn.hpp"
#include "kernels_ivox_coop.hpp"

namespace fls {

// the 27 cells of the 3x3x3 block (raster code = (dx+1) + 3 (dy+1) + 9 (dz+1)) ordered nearest-first:
// centre, 6 faces, 12 edges, 8 corners
__device__ const unsigned char kCellOrder[27] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 0, 2, 6, 8, 18, 20, 24, 26};

// the 98 cells of the 5x5x5 block that are not in the inner 3x3x3 (code = (dx+2) + 5 (dy+2) + 25 (dz+2)), nearest-first
__device__ const unsigned char kShellOrder[98] = {
    12, 52, 60, 64, 72, 112, 7, 11, 13, 17, 27, 35, 39, 47, 51, 53, 55, 59, 65, 69, 71, 73, 77, 85, 89, 97, 107, 111, 113, 117, 6, 8, 16, 18,
    26, 28, 30, 34, 40, 44, 46, 48, 76, 78, 80, 84, 90, 94, 96, 98, 106, 108, 116, 118, 2, 10, 14, 22, 50, 54, 70, 74, 102, 110, 114, 122, 1,
    3, 5, 9, 15, 19, 21, 23, 25, 29, 45, 49, 75, 79, 95, 99, 101, 103, 105, 109, 115, 119, 121, 123, 0, 4, 20, 24, 100, 104, 120, 124};

template <int K>
__device__ __forceinline__ void topk_insert(unsigned long long (&t)[K], unsigned (&s)[K], const unsigned long long key, const unsigned slot) {
    if (key < t[K - 1]) {
        t[K - 1] = key;
        s[K - 1] = slot;
#pragma unroll
        for (int j = K - 1; j > 0; --j) {
            if (t[j] < t[j - 1]) {
                const unsigned long long x = t[j]; t[j] = t[j - 1]; t[j - 1] = x;
                const unsigned y = s[j]; s[j] = s[j - 1]; s[j - 1] = y;
            }
        }
    }
}
__device__ __forceinline__ unsigned group8_min_u32(unsigned v) {
    unsigned o = dpp_pair_u32<0>(v); v = o < v ? o : v;
    o = dpp_pair_u32<1>(v); v = o < v ? o : v;
    o = dpp_pair_u32<2>(v); v = o < v ? o : v;
    return v;
}

// (the body is a device function of the block index so that two queries -- the corner and the planar class of the LOAM
// matcher -- can share one launch: grid_knn_dual_kernel below)
// (EMIT = false: nothing is written; lane j of a group returns the j-th neighbour's key / slot, every lane the neighbour
// count, the K-th distance and its query index (-1: none) -- the fused ICP kernel below consumes them in registers)
struct GridKnnLane {
    unsigned long long key;
    unsigned slot;
    int found, q;
    float kth;
    // K == 1 (the fused IcpOptimized kernel): the nearest neighbour's coordinates and the query's own source point, kept from the search -- the fit
    // used to load map point `slot` and the source point again, one more dependent memory round trip in a launch that is a chain of ten (round 6)
    float bx, by, bz, px, py, pz;
};
// UNGATED: the caller searches without a gate (LoamPointToPlaneKdtree): only that instantiation carries the ring walk and the serial
// fallback below (as a run-time test on `gate` they sat in every instantiation's register budget)
template <int K, bool FLOAT_XFORM, bool EMIT = true, bool UNGATED = false>
__device__ __forceinline__ void
grid_knn_body(const int bid, const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
              const GnState* __restrict__ st, const int first, const Pose16& T0, const CellGridDev& cg, const float gate,
              float4* __restrict__ nn_pts /* [n][K] */, unsigned char* __restrict__ nn_cnt, float* __restrict__ kth_d2,
              unsigned char* __restrict__ flag_to_clear /* may be null */, GridKnnLane* __restrict__ lane_out = nullptr) {
    constexpr int G = 8, QPB = 256 / G;  // 27 cells over 8 lanes: 4 rounds
    // XCD-aware order: interleaved chunks of 8 workgroups per XCD (kernels_ivox_coop.hpp; the grid is a multiple of 64).
    // One contiguous eighth per XCD left the XCD that owns the sparse upper rings with all the second-stage searches.
    const int nb = (n + QPB - 1) / QPB;
    const int lb = (((bid >> 3) >> 3) * 8 + (bid & 7)) * 8 + ((bid >> 3) & 7);
    const int sub = threadIdx.x % G;
    const int q = lb * QPB + threadIdx.x / G;
    const bool active = lb < nb && q < n;
    const int done = first ? 0 : st->done;
    double T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = first ? T0.m[k] : st->T[k];
    const float px = active ? sx[q] : 0.f, py = active ? sy[q] : 0.f, pz = active ? sz[q] : 0.f;
    if (done) return;
    if (first && flag_to_clear && active && sub == 0) flag_to_clear[q] = 0;
    float qx, qy, qz;
    if (FLOAT_XFORM) {
        const RtFloat rt = load_rt_float(T);
        xform_f(rt, px, py, pz, qx, qy, qz);
    } else {
        const double x = px, y = py, z = pz;
        qx = (float)(((T[0] * x + T[4] * y) + T[8] * z) + T[12]);
        qy = (float)(((T[1] * x + T[5] * y) + T[9] * z) + T[13]);
        qz = (float)(((T[2] * x + T[6] * y) + T[10] * z) + T[14]);
    }
    const double fx = floor((double)qx * cg.inv_cell), fy = floor((double)qy * cg.inv_cell), fz = floor((double)qz * cg.inv_cell);
    const bool in_range = active && fabs(fx) < (double)(kKeyLimit - 3) && fabs(fy) < (double)(kKeyLimit - 3) && fabs(fz) < (double)(kKeyLimit - 3);
    const int cx = in_range ? (int)fx : 0, cy = in_range ? (int)fy : 0, cz = in_range ? (int)fz : 0;

    // The 27 cells are visited nearest-first (centre, 6 faces, 12 edges, 8 corners): cell order[sub + 8 r] in
    // round r.  Round 0 (centre + faces + one edge) is scanned first; it yields an upper bound B on the K-th
    // neighbour distance (any lane holding K points certifies one), and the remaining 19 cells are probed and
    // scanned only if their box can still contain something closer than B (or than the caller's gate).  The
    // pruning is exact: a skipped cell's minimum distance exceeds B by a 1e-5 relative margin.
    auto cell_of = [&](const int k, int& dx, int& dy, int& dz) {
        const int code = kCellOrder[k < 27 ? k : 0];
        dx = code % 3 - 1; dy = (code / 3) % 3 - 1; dz = code / 9 - 1;
    };
    auto first_load = [&](const bool pv, const unsigned long long key, unsigned& h) -> HashEntry {
        h = hash_key(key) & cg.g.mask;
        return pv ? cg.g.table[h] : HashEntry{kEmptyKey, 0u, 0u};
    };
    auto resolve = [&](const bool pv, const unsigned long long key, unsigned h, HashEntry ek, unsigned& b, unsigned& c) {
        while (pv && ek.key != key && ek.key != kEmptyKey) {
            h = (h + 1) & cg.g.mask;
            ek = cg.g.table[h];
        }
        const bool hit = pv && ek.key == key;
        b = hit ? ek.begin : 0u;
        c = hit ? ek.count : 0u;
    };
    // cell (cx + dx, cy + dy, cz + dz) -> its point range; count = 0 when empty / not wanted
    auto lookup = [&](const bool pv, const int dx, const int dy, const int dz, unsigned& b, unsigned& c) {
        if (cg.win.cells) {  // uniform
            const int wx = cx + dx - cg.win.ox, wy = cy + dy - cg.win.oy, wz = cz + dz - cg.win.oz;
            const bool ok = pv && (unsigned)wx < (unsigned)cg.win.nx && (unsigned)wy < (unsigned)cg.win.ny && (unsigned)wz < (unsigned)cg.win.nz;
            const uint2 e = cg.win.cells[ok ? (unsigned)((wz * cg.win.ny + wy) * cg.win.nx + wx) : 0u];
            b = e.x;
            c = ok ? e.y : 0u;
        } else {
            const unsigned long long key = pack_key(cx + dx, cy + dy, cz + dz);
            unsigned h;
            const HashEntry e0 = first_load(pv, key, h);
            resolve(pv, key, h, e0, b, c);
        }
    };
    unsigned long long t[K];
    unsigned sl[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { t[j] = ~0ull; sl[j] = 0u; }
    int ncand = 0;
    constexpr bool CARRY = (K == 1) && !EMIT;  // this lane's best candidate travels with its key
    float best_x = 0.f, best_y = 0.f, best_z = 0.f;
    auto consider = [&](const float4 p, const unsigned s, const bool ok) {
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // flann::L2_Simple<float>
        if (ok && !(d2 != d2)) {
            ++ncand;
            const unsigned long long kk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(p.w);
            if constexpr (CARRY) { if (kk < t[0]) { best_x = p.x; best_y = p.y; best_z = p.z; } }
            topk_insert<K>(t, sl, kk, s);
        }
    };
    // ---- round 0
    {
        int dx, dy, dz;
        cell_of(sub, dx, dy, dz);
        unsigned b0, c0;
        lookup(in_range, dx, dy, dz, b0, c0);
        const unsigned last = b0 + c0 - 1;  // only dereferenced when c0 > 0
        for (unsigned s = b0; s < b0 + c0; s += 4) {
            const unsigned s1 = s + 1 < last ? s + 1 : last, s2 = s + 2 < last ? s + 2 : last, s3 = s + 3 < last ? s + 3 : last;
            const float4 p0 = cg.g.pts[s], p1 = cg.g.pts[s1], p2 = cg.g.pts[s2], p3 = cg.g.pts[s3];
            consider(p0, s, true);
            consider(p1, s1, s + 1 <= last);
            consider(p2, s2, s + 2 <= last);
            consider(p3, s3, s + 3 <= last);
        }
    }
    // ---- bound after round 0 (non-negative floats order like their bit patterns)
    const unsigned my_kth = (t[K - 1] != ~0ull) ? (unsigned)(t[K - 1] >> 32) : 0x7f800000u;
    float bound = __uint_as_float(group8_min_u32(my_kth));
    bound = gate < bound ? gate : bound;
    // ---- rounds 1..3: probe + scan only the cells that can still matter
    const double qdx = (double)qx, qdy = (double)qy, qdz = (double)qz;
    // squared minimum distance from the query to the box of the cell at offset (dx, dy, dz), shrunk by a 1e-5 margin
    auto box_dmin2 = [&](const int dx, const int dy, const int dz) -> double {
        const double ax = dx > 0 ? (double)(cx + dx) * cg.cell - qdx : (dx < 0 ? qdx - (double)(cx + dx + 1) * cg.cell : 0.0);
        const double ay = dy > 0 ? (double)(cy + dy) * cg.cell - qdy : (dy < 0 ? qdy - (double)(cy + dy + 1) * cg.cell : 0.0);
        const double az = dz > 0 ? (double)(cz + dz) * cg.cell - qdz : (dz < 0 ? qdz - (double)(cz + dz + 1) * cg.cell : 0.0);
        const double bx = ax > 0.0 ? ax : 0.0, by = ay > 0.0 ? ay : 0.0, bz = az > 0.0 ? az : 0.0;
        return ((bx * bx + by * by) + bz * bz) * (1.0 - 1e-5);
    };
    auto plan = [&](const int r, unsigned& b, unsigned& c) {
        const int k = sub + G * r;
        int dx, dy, dz;
        cell_of(k, dx, dy, dz);
        const bool pv = in_range && k < 27 && !(box_dmin2(dx, dy, dz) > (double)bound);
        lookup(pv, dx, dy, dz, b, c);
    };
    unsigned b1, b2, b3, c1, c2, c3;
    plan(1, b1, c1);
    plan(2, b2, c2);
    plan(3, b3, c3);
    const unsigned tot = c1 + c2 + c3;
    // candidate index -> map slot: compare / select chain on values (kernels_ivox_coop.hpp slot_select), no branches
    const unsigned p1 = c1, p2 = c1 + c2, o0 = b1, o1 = b2 - p1, o2 = b3 - p2;
    auto slot_of = [=](const unsigned idx) -> unsigned { return slot_select<3>(idx, p1, p2, 0u, 0u, o0, o1, o2, 0u, 0u); };
    for (unsigned j = 0; j < tot; j += 4) {
        const unsigned last = tot - 1;
        const unsigned i1 = j + 1, i2 = j + 2, i3 = j + 3;
        const unsigned s0 = slot_of(j), s1 = slot_of(i1 < last ? i1 : last), s2 = slot_of(i2 < last ? i2 : last),
                       s3 = slot_of(i3 < last ? i3 : last);
        const float4 p0 = cg.g.pts[s0], p1 = cg.g.pts[s1], p2 = cg.g.pts[s2], p3 = cg.g.pts[s3];
        consider(p0, s0, true);
        consider(p1, s1, i1 <= last);
        consider(p2, s2, i2 <= last);
        consider(p3, s3, i3 <= last);
    }
    // merge the 8 private lists: K rounds of group-min + pop; the j-th neighbour lands in lane sub == j
    unsigned long long mine_key = ~0ull, last_key = ~0ull;
    unsigned mine_slot = 0u;
    int found = 0;
    float kth = INFINITY;
    float mine_x = 0.f, mine_y = 0.f, mine_z = 0.f;
    auto merge = [&]() {
        const int total = group_sum_i32<G>(ncand);
        mine_key = ~0ull; last_key = ~0ull; mine_slot = 0u;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const unsigned long long m = group_min_u64<G>(t[0]);
            const bool owner = (t[0] == m) && (m != ~0ull);
            const unsigned ms = group8_min_u32(owner ? sl[0] : 0xffffffffu);
            if constexpr (CARRY) {  // keys are unique (the map index is the low word): one owner, whose coordinates every lane of the group receives
                mine_x = __uint_as_float(group8_min_u32(owner ? __float_as_uint(best_x) : 0xffffffffu));
                mine_y = __uint_as_float(group8_min_u32(owner ? __float_as_uint(best_y) : 0xffffffffu));
                mine_z = __uint_as_float(group8_min_u32(owner ? __float_as_uint(best_z) : 0xffffffffu));
            }
            if (owner) {
#pragma unroll
                for (int u = 0; u + 1 < K; ++u) { t[u] = t[u + 1]; sl[u] = sl[u + 1]; }
                t[K - 1] = ~0ull;
            }
            if (sub == j) { mine_key = m; mine_slot = ms; }
            last_key = m;
        }
        found = total < K ? total : K;
        kth = (total >= K && last_key != ~0ull) ? __uint_as_float((unsigned)(last_key >> 32)) : INFINITY;
    };
    merge();
    // ---- stage 2 (half-size cells): the inner 27 cells certify the result only if the K-th neighbour lies within one
    // cell size; otherwise the 98 shell cells of the 5x5x5 block (which covers the gate) are examined, nearest-first,
    // each only if its box can hold something closer than the bound (exact K-th so far, or the gate).  The merged
    // top-K is dealt back one entry per lane, so the second merge sees stage-1 results and shell candidates together.
    if (cg.rings == 2) {
        const bool enough = found == K && (double)kth <= cg.cell * cg.cell * (1.0 - 1e-5);
        if (!enough) {  // uniform within the group
            float b2 = found == K ? kth : INFINITY;
            b2 = gate < b2 ? gate : b2;
#pragma unroll
            for (int j = 0; j < K; ++j) { t[j] = ~0ull; sl[j] = 0u; }
            ncand = 0;
            if (sub < K && mine_key != ~0ull) { t[0] = mine_key; sl[0] = mine_slot; ncand = 1; if constexpr (CARRY) { best_x = mine_x; best_y = mine_y; best_z = mine_z; } }
            // 13 shell cells per lane in batches of five: the (pruned) cell lookups of a batch are in flight together, then
            // one flattened candidate loop over the cells that survived -- two memory round trips per batch, not per cell
            auto shell = [&](const int r, unsigned& bb, unsigned& cc) {
                const int k = sub + G * r;
                const int code = kShellOrder[k < 98 ? k : 0];
                const int dx = code % 5 - 2, dy = (code / 5) % 5 - 2, dz = code / 25 - 2;
                const bool pv = in_range && r < 13 && k < 98 && !(box_dmin2(dx, dy, dz) > (double)b2);
                lookup(pv, dx, dy, dz, bb, cc);
            };
#pragma unroll
            for (int batch = 0; batch < 3; ++batch) {
                unsigned sb0, sb1, sb2, sb3, sb4, sc0, sc1, sc2, sc3, sc4;
                shell(5 * batch + 0, sb0, sc0);
                shell(5 * batch + 1, sb1, sc1);
                shell(5 * batch + 2, sb2, sc2);
                shell(5 * batch + 3, sb3, sc3);
                shell(5 * batch + 4, sb4, sc4);
                const unsigned q1 = sc0, q2 = q1 + sc1, q3 = q2 + sc2, q4 = q3 + sc3, stot = q4 + sc4;
                const unsigned u0 = sb0, u1 = sb1 - q1, u2 = sb2 - q2, u3 = sb3 - q3, u4 = sb4 - q4;
                for (unsigned j = 0; j < stot; j += 4) {
                    const unsigned last = stot - 1;
                    const unsigned i1 = j + 1, i2 = j + 2, i3 = j + 3;
                    const unsigned s0 = slot_select<5>(j, q1, q2, q3, q4, u0, u1, u2, u3, u4);
                    const unsigned s1 = slot_select<5>(i1 < last ? i1 : last, q1, q2, q3, q4, u0, u1, u2, u3, u4);
                    const unsigned s2 = slot_select<5>(i2 < last ? i2 : last, q1, q2, q3, q4, u0, u1, u2, u3, u4);
                    const unsigned s3 = slot_select<5>(i3 < last ? i3 : last, q1, q2, q3, q4, u0, u1, u2, u3, u4);
                    const float4 p0 = cg.g.pts[s0], p1 = cg.g.pts[s1], p2 = cg.g.pts[s2], p3 = cg.g.pts[s3];
                    consider(p0, s0, true);
                    consider(p1, s1, i1 <= last);
                    consider(p2, s2, i2 <= last);
                    consider(p3, s3, i3 <= last);
                }
            }
            merge();
        }
    }
    // un-gated search (LoamPointToPlaneKdtree): the 27 cells certify the result only if the K-th neighbour lies within one cell
    // size.  Otherwise the group walks the rings 2 .. kMaxRing TOGETHER (round 3): the shell cells of a ring are dealt round-robin
    // to the eight lanes (eight probe chains in flight instead of one lane probing up to 729 cells one after the other: the serial
    // version took up to 590 us per launch), a cell is probed only if its box can still hold something closer than the K-th
    // distance known so far, and after every ring the merged K-th distance is tested against the ring's radius.  Only a query
    // that is still uncertain after kMaxRing rings (far outside the map) falls back to lane 0's exact brute-force search.
    const double rad2 = cg.cell * cg.cell * (1.0 - 1e-5);
    bool certain = !UNGATED || (found == K && (double)kth <= rad2);
    if constexpr (UNGATED) {
    if (!certain) {  // uniform within the group
        for (int rho = 2; rho <= kMaxRing && !certain; ++rho) {
            const float bnd = found == K ? kth : INFINITY;
#pragma unroll
            for (int j = 0; j < K; ++j) { t[j] = ~0ull; sl[j] = 0u; }
            ncand = 0;
            if (sub < K && mine_key != ~0ull) { t[0] = mine_key; sl[0] = mine_slot; ncand = 1; if constexpr (CARRY) { best_x = mine_x; best_y = mine_y; best_z = mine_z; } }  // the merged top-K, one entry per lane
            const int w = 2 * rho + 1, cube = w * w * w;
            for (int idx = sub; idx < cube; idx += G) {
                const int dx = idx % w - rho, dy = (idx / w) % w - rho, dz = idx / (w * w) - rho;
                const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, az = dz < 0 ? -dz : dz;
                const int m = ax > ay ? (ax > az ? ax : az) : (ay > az ? ay : az);
                if (m != rho || !in_range || box_dmin2(dx, dy, dz) > (double)bnd) continue;
                unsigned bb, cc;
                lookup(true, dx, dy, dz, bb, cc);
                const unsigned last = bb + cc - 1;
                for (unsigned sp = bb; sp < bb + cc; sp += 4) {
                    const unsigned s1 = sp + 1 < last ? sp + 1 : last, s2 = sp + 2 < last ? sp + 2 : last, s3 = sp + 3 < last ? sp + 3 : last;
                    const float4 p0 = cg.g.pts[sp], p1 = cg.g.pts[s1], p2 = cg.g.pts[s2], p3 = cg.g.pts[s3];
                    consider(p0, sp, true);
                    consider(p1, s1, sp + 1 <= last);
                    consider(p2, s2, sp + 2 <= last);
                    consider(p3, s3, sp + 3 <= last);
                }
            }
            merge();
            const double rr = (double)rho * cg.cell;
            certain = found == K && (double)kth <= rr * rr * (1.0 - 1e-5);
        }
    }
    if (!certain) {  // uniform within the group
        if (sub == 0 && active) {
            KnnResult<K> r;
            unsigned long long a = 0, b = 0, c = 0;
            knn_grid<K>(cg, qx, qy, qz, gate, r, a, b, c);
#pragma unroll
            for (int j = 0; j < K; ++j)
                nn_pts[(size_t)q * K + j] = j < r.found ? cg.g.pts[r.slot[j]] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            nn_cnt[q] = (unsigned char)r.found;
            kth_d2[q] = r.found == K ? r.d[K - 1] : INFINITY;
        }
        return;
    }
    }  // UNGATED
    if (!EMIT) {
        lane_out->key = mine_key; lane_out->slot = mine_slot; lane_out->found = found; lane_out->kth = kth; lane_out->q = active ? q : -1;
        lane_out->bx = mine_x; lane_out->by = mine_y; lane_out->bz = mine_z; lane_out->px = px; lane_out->py = py; lane_out->pz = pz;
        return;
    }
    if (sub < K && active)
        nn_pts[(size_t)q * K + sub] = (mine_key != ~0ull) ? cg.g.pts[mine_slot] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    if (sub == 0 && active) { nn_cnt[q] = (unsigned char)found; kth_d2[q] = kth; }
}

template <int K, bool FLOAT_XFORM, bool UNGATED = false>
__global__ void __launch_bounds__(256)
grid_knn_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                const GnState* __restrict__ st, const int first, const Pose16 T0, const CellGridDev cg, const float gate,
                float4* __restrict__ nn_pts /* [n][K] */, unsigned char* __restrict__ nn_cnt, float* __restrict__ kth_d2,
                unsigned char* __restrict__ flag_to_clear /* may be null */) {
    grid_knn_body<K, FLOAT_XFORM, true, UNGATED>((int)blockIdx.x, sx, sy, sz, n, st, first, T0, cg, gate, nn_pts, nn_cnt, kth_d2, flag_to_clear);
}

// two independent queries in one launch: blocks [0, nb_a) serve A, the rest B (both block counts are multiples of 64, so the
// XCD a block lands on is the one its own launch would have given it).  LoamFull: the corner class is 7,680 queries with
// long candidate lists (~1 wave per SIMD, 27 us alone), the planar class 57,600 queries (33 us alone): together ~35 us.
struct GridKnnArgs {
    const float *sx, *sy, *sz;
    int n;
    CellGridDev cg;
    float gate;
    float4* nn_pts;
    unsigned char* nn_cnt;
    float* kth_d2;
    unsigned char* flag_to_clear;
};
// six waves per SIMD (80 VGPRs, no scratch; the natural allocation is 84 = five waves): LoamFull Match 255 -> 249 us.  (Eight waves spill: round 1.)
template <int K, bool FLOAT_XFORM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6)))
grid_knn_dual_kernel(const GnState* __restrict__ st, const int first, const Pose16 T0, const GridKnnArgs a, const GridKnnArgs b, const int nb_a) {
    if ((int)blockIdx.x < nb_a)
        grid_knn_body<K, FLOAT_XFORM>((int)blockIdx.x, a.sx, a.sy, a.sz, a.n, st, first, T0, a.cg, a.gate, a.nn_pts, a.nn_cnt, a.kth_d2, a.flag_to_clear);
    else
        grid_knn_body<K, FLOAT_XFORM>((int)blockIdx.x - nb_a, b.sx, b.sy, b.sz, b.n, st, first, T0, b.cg, b.gate, b.nn_pts, b.nn_cnt, b.kth_d2, b.flag_to_clear);
}

// ---------------------------------------------------------------------------------------------
// grid_knn27_kernel: the gated 5-NN of the LOAM feature maps as ONE stage, built like ivox_knn_kernel: 4 lanes per query, the
// 27 gate-sized cells of the 3x3x3 block looked up together (7 per lane, all loads in flight), the block's candidates cut
// into four equal ranges (per-group LDS table), private sorted top-5 of IEEE-double keys {float-bits(d2) : map index},
// five rounds of DPP group-min.  No pruning, no second stage: every point within the gate lies in the block (cell >=
// sqrt(gate), kernels_knn.hpp), so the result is exact whenever the reference accepts the point (5th neighbour inside the
// gate); the nearest-first / pruned / two-stage kernel above spends its time in dependent memory round trips
// (cell -> points -> bound -> cells -> points -> merge -> shell ...), this one trades them for candidate arithmetic.
// Winners are gathered from the map cloud in its own order (by_id): the key's low word is the map index (tie rule).
// ---------------------------------------------------------------------------------------------
template <bool FLOAT_XFORM>
__global__ void __launch_bounds__(256)
grid_knn27_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                  const GnState* __restrict__ st, const int first, const Pose16 T0, const CellGridDev cg,
                  float4* __restrict__ nn_pts /* [n][5] */, unsigned char* __restrict__ nn_cnt, float* __restrict__ kth_d2,
                  unsigned char* __restrict__ flag_to_clear /* may be null */) {
    constexpr int G = 4, QPB = 256 / G, R = 7, ROW = 36;
    __shared__ __attribute__((aligned(16))) unsigned s_end[QPB][ROW];
    __shared__ __attribute__((aligned(16))) unsigned s_off[QPB][ROW];
    const int nb = (n + QPB - 1) / QPB;
    const int lb = (((blockIdx.x >> 3) >> 3) * 8 + (blockIdx.x & 7)) * 8 + ((blockIdx.x >> 3) & 7);  // interleaved XCD chunks of 8 workgroups
    const int sub = threadIdx.x % G, g = threadIdx.x / G;
    const int q = lb * QPB + g;
    const bool active = lb < nb && q < n;
    const int done = first ? 0 : st->done;
    double T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = first ? T0.m[k] : st->T[k];
    const int qq = active ? q : 0;
    const float px = sx[qq], py = sy[qq], pz = sz[qq];
    if (done) return;
    if (first && flag_to_clear && active && sub == 0) flag_to_clear[q] = 0;
    float qx, qy, qz;
    if (FLOAT_XFORM) {
        const RtFloat rt = load_rt_float(T);
        xform_f(rt, px, py, pz, qx, qy, qz);
    } else {
        const double x = px, y = py, z = pz;
        qx = (float)(((T[0] * x + T[4] * y) + T[8] * z) + T[12]);
        qy = (float)(((T[1] * x + T[5] * y) + T[9] * z) + T[13]);
        qz = (float)(((T[2] * x + T[6] * y) + T[10] * z) + T[14]);
    }
    const double fx = floor((double)qx * cg.inv_cell), fy = floor((double)qy * cg.inv_cell), fz = floor((double)qz * cg.inv_cell);
    const bool in_range = active && fabs(fx) < (double)(kKeyLimit - 3) && fabs(fy) < (double)(kKeyLimit - 3) && fabs(fz) < (double)(kKeyLimit - 3);
    const int cx = in_range ? (int)fx : 0, cy = in_range ? (int)fy : 0, cz = in_range ? (int)fz : 0;
    // this lane's cells: raster codes sub, sub + 4, ... < 27
    auto lookup = [&](const int r, unsigned& b, unsigned& c) {
        const int k = sub + G * r;
        const bool pv = in_range && k < 27;
        const int kk = k < 27 ? k : 0;
        const int dx = kk % 3 - 1, dy = (kk / 3) % 3 - 1, dz = kk / 9 - 1;
        if (cg.win.cells) {  // uniform
            const int wx = cx + dx - cg.win.ox, wy = cy + dy - cg.win.oy, wz = cz + dz - cg.win.oz;
            const bool ok = pv && (unsigned)wx < (unsigned)cg.win.nx && (unsigned)wy < (unsigned)cg.win.ny && (unsigned)wz < (unsigned)cg.win.nz;
            const uint2 e = cg.win.cells[ok ? (unsigned)((wz * cg.win.ny + wy) * cg.win.nx + wx) : 0u];
            b = e.x;
            c = ok ? e.y : 0u;
        } else {
            const unsigned long long key = pack_key(cx + dx, cy + dy, cz + dz);
            unsigned h = hash_key(key) & cg.g.mask;
            HashEntry ek = pv ? cg.g.table[h] : HashEntry{kEmptyKey, 0u, 0u};
            while (pv && ek.key != key && ek.key != kEmptyKey) { h = (h + 1) & cg.g.mask; ek = cg.g.table[h]; }
            const bool hit = pv && ek.key == key;
            b = hit ? ek.begin : 0u;
            c = hit ? ek.count : 0u;
        }
    };
    unsigned b0, b1, b2, b3, b4, b5, b6, c0, c1, c2, c3, c4, c5, c6;
    lookup(0, b0, c0); lookup(1, b1, c1); lookup(2, b2, c2); lookup(3, b3, c3); lookup(4, b4, c4); lookup(5, b5, c5); lookup(6, b6, c6);
    static_assert(R == 7, "seven explicit rounds");
    const double kNone = __hiloint2double((int)kKeyNoneHi, -1);
    double t5[5] = {kNone, kNone, kNone, kNone, kNone};
    int ncand = 0;
    auto consider = [&](const float4 p, const bool ok) {
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;  // flann::L2_Simple<float>
        const bool v = ok && !(d2 != d2);
        ncand += v ? 1 : 0;
        top5_insert_dkey(t5, make_dkey(d2, (unsigned)__float_as_int(p.w), v));
    };
    // balanced split of the block's candidates over the four lanes (ivox_knn_kernel, BAL)
    const unsigned tot = ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + c6);
    const unsigned nz = (c0 ? 1u : 0u) + (c1 ? 1u : 0u) + (c2 ? 1u : 0u) + (c3 ? 1u : 0u) + (c4 ? 1u : 0u) + (c5 ? 1u : 0u) + (c6 ? 1u : 0u);
    const unsigned packed = tot * 32u + nz;  // candidates (< 2^27) and occupied cells (<= 27 per group)
    const unsigned t0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0x00, 0xf, 0xf, true);  // quad broadcasts
    const unsigned t1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0x55, 0xf, 0xf, true);
    const unsigned t2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0xAA, 0xf, 0xf, true);
    const unsigned t3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)packed, 0xFF, 0xf, 0xf, true);
    const unsigned before = (sub > 0 ? t0 : 0u) + (sub > 1 ? t1 : 0u) + (sub > 2 ? t2 : 0u), all = t0 + t1 + t2 + t3;
    const unsigned TOT = all >> 5;
    unsigned pos = before & 31u, run = before >> 5;
#pragma unroll
    for (int k = 0; k < ROW / G; ++k) s_end[g][(ROW / G) * sub + k] = ~0u;  // sentinels behind the last real entry
    if (c0) { s_off[g][pos] = b0 - run; run += c0; s_end[g][pos] = run; ++pos; }
    if (c1) { s_off[g][pos] = b1 - run; run += c1; s_end[g][pos] = run; ++pos; }
    if (c2) { s_off[g][pos] = b2 - run; run += c2; s_end[g][pos] = run; ++pos; }
    if (c3) { s_off[g][pos] = b3 - run; run += c3; s_end[g][pos] = run; ++pos; }
    if (c4) { s_off[g][pos] = b4 - run; run += c4; s_end[g][pos] = run; ++pos; }
    if (c5) { s_off[g][pos] = b5 - run; run += c5; s_end[g][pos] = run; ++pos; }
    if (c6) { s_off[g][pos] = b6 - run; run += c6; s_end[g][pos] = run; ++pos; }
    // a group lives inside one wave and the LDS serves a wave's operations in order: wave-level ordering only
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const unsigned Q = (TOT + 3u) >> 2, a = sub * Q, e = a + Q < TOT ? a + Q : TOT;
        unsigned k = 0;  // first table entry whose range reaches past a
#pragma unroll
        for (int v = 0; v < 7; ++v) {
            const uint4 e4 = *reinterpret_cast<const uint4*>(&s_end[g][4 * v]);
            k += (e4.x <= a ? 1u : 0u) + (e4.y <= a ? 1u : 0u) + (e4.z <= a ? 1u : 0u) + (e4.w <= a ? 1u : 0u);
        }
        for (unsigned idx = a; idx < e; idx += 4) {
            const unsigned E0 = s_end[g][k], E1 = s_end[g][k + 1], E2 = s_end[g][k + 2], E3 = s_end[g][k + 3];
            const unsigned O0 = s_off[g][k], O1 = s_off[g][k + 1], O2 = s_off[g][k + 2], O3 = s_off[g][k + 3], O4 = s_off[g][k + 4];
            const unsigned last = e - 1;
            const unsigned i1 = idx + 1, i2 = idx + 2, i3 = idx + 3;
            const unsigned s0 = slot_select<5>(idx, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s1 = slot_select<5>(i1 < last ? i1 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s2 = slot_select<5>(i2 < last ? i2 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const unsigned s3 = slot_select<5>(i3 < last ? i3 : last, E0, E1, E2, E3, O0, O1, O2, O3, O4);
            const float4 q0 = cg.g.pts[s0], q1 = cg.g.pts[s1], q2 = cg.g.pts[s2], q3 = cg.g.pts[s3];
            consider(q0, true);
            consider(q1, i1 <= last);
            consider(q2, i2 <= last);
            consider(q3, i3 <= last);
            const unsigned nx = idx + 4;
            k += (E0 <= nx ? 1u : 0u) + (E1 <= nx ? 1u : 0u) + (E2 <= nx ? 1u : 0u) + (E3 <= nx ? 1u : 0u);
        }
    }
    // merge: five rounds of group-min + pop (keys are unique: the map index is the low word)
    const int total = group_sum_i32<G>(ncand);
    double last = kNone;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const double m = group_min_dkey<G>(t5[0]);
        const bool mv = dkey_valid(m);
        if (mv && __double_as_longlong(t5[0]) == __double_as_longlong(m)) {
            t5[0] = t5[1]; t5[1] = t5[2]; t5[2] = t5[3]; t5[3] = t5[4]; t5[4] = kNone;
        }
        if (sub == 0 && active) nn_pts[(size_t)q * 5 + j] = mv ? cg.by_id[dkey_slot(m)] : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
        last = m;
    }
    if (sub == 0 && active) {
        nn_cnt[q] = (unsigned char)(total < 5 ? total : 5);
        kth_d2[q] = (total >= 5 && dkey_valid(last)) ? __uint_as_float((unsigned)__double2hiint(last) - kKeyBias) : INFINITY;
    }
}

// one partial row per 256-thread workgroup from the per-wave sums in LDS (fixed order)
__device__ __forceinline__ void block_row_from_wave_sums(double (*wsum)[32], double* __restrict__ partials, const int bid) {
    __syncthreads();
    if (threadIdx.x < 29) {
        const double v = ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
        partials[(size_t)bid * kPartialStride + threadIdx.x] = v;
    }
}
__device__ __forceinline__ void block_row_from_wave_sums(double (*wsum)[32], double* __restrict__ partials) {
    block_row_from_wave_sums(wsum, partials, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// IcpOptimized: e = pt - q, J = [I | -R hat(p)], H = J^T J, B = -J^T e   (icp_optimized.h:87-108)
// ---------------------------------------------------------------------------------------------
// per-point contribution (upper triangle of J^T J, -J^T e, |e|) of one accepted correspondence
__device__ __forceinline__ void icp_point_terms(const double (&T)[16], const float px, const float py, const float pz, const float4 m, double (&Hc)[21],
                                                double (&Bc)[6], double& res) {
    const RtFloat rt = load_rt_float(T);
    float qx, qy, qz;
    xform_f(rt, px, py, pz, qx, qy, qz);
    const double e0 = (double)qx - (double)m.x, e1 = (double)qy - (double)m.y, e2 = (double)qz - (double)m.z;
    const double o0 = px, o1 = py, o2 = pz;
    const double hat[9] = {0.0, o2, -o1, -o2, 0.0, o0, o1, -o0, 0.0};
    double J[18];  // 3x6 column-major: [I | M],  M = -(R * SO3Hat(o)), zero terms kept in place
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            J[r + j * 3] = (r == j) ? 1.0 : 0.0;
            J[r + (j + 3) * 3] = -((T[r] * hat[0 + j * 3] + T[r + 4] * hat[1 + j * 3]) + T[r + 8] * hat[2 + j * 3]);
        }
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) {
            Hc[k] = (J[0 + a * 3] * J[0 + b * 3] + J[1 + a * 3] * J[1 + b * 3]) + J[2 + a * 3] * J[2 + b * 3];
            ++k;
        }
#pragma unroll
    for (int a = 0; a < 6; ++a) Bc[a] = ((-J[0 + a * 3]) * e0 + (-J[1 + a * 3]) * e1) + (-J[2 + a * 3]) * e2;
    res = sqrt((e0 * e0 + e1 * e1) + e2 * e2);
}

// the same correspondence as the three rows of its Jacobian + the residual vector (what the matrix-core reduction consumes)
__device__ __forceinline__ void icp_point_rows(const double (&T)[16], const float px, const float py, const float pz, const float4 m, double (&J)[18],
                                               double (&e)[3], double& res) {
    const RtFloat rt = load_rt_float(T);
    float qx, qy, qz;
    xform_f(rt, px, py, pz, qx, qy, qz);
    e[0] = (double)qx - (double)m.x; e[1] = (double)qy - (double)m.y; e[2] = (double)qz - (double)m.z;
    const double o0 = px, o1 = py, o2 = pz;
    const double hat[9] = {0.0, o2, -o1, -o2, 0.0, o0, o1, -o0, 0.0};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            J[r + j * 3] = (r == j) ? 1.0 : 0.0;
            J[r + (j + 3) * 3] = -((T[r] * hat[0 + j * 3] + T[r + 4] * hat[1 + j * 3]) + T[r + 8] * hat[2 + j * 3]);
        }
    res = sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
}

// IcpOptimized correspondence search AND fit in one launch: the 1-NN of a query ends up in lane 0 of its 8-lane group, which
// forms the point's terms at once (no neighbour arrays through memory, one launch per iteration fewer); every workgroup of the
// search grid writes one partial row (rows of idle workgroups are zero), gn_solve_lu_kernel sums them in row order.
__global__ void __launch_bounds__(256)
icp_knn_fit_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                   GnState* __restrict__ st, const int first, const Pose16 T0, const CellGridDev cg, const float gate, const double max_corr /* squared */,
                   int* __restrict__ nn_id, unsigned char* __restrict__ eff, double* __restrict__ partials,
                   unsigned* __restrict__ ticket /* nullptr: the tail runs as its own launch */, const int shards, const LuTailArgs tail) {
#ifdef FLS_TIMING
    const long long t_begin = (long long)__builtin_readcyclecounter();
#endif
    const int done = first ? 0 : st->done;
    const int it = first ? 0 : st->iter;
    if (done) return;  // uniform over the launch
    __shared__ double wsum[4][32];
    GridKnnLane r;
    r.key = ~0ull; r.slot = 0u; r.found = 0; r.q = -1; r.kth = INFINITY;
    r.bx = r.by = r.bz = r.px = r.py = r.pz = 0.f;
    grid_knn_body<1, true, false>((int)blockIdx.x, sx, sy, sz, n, st, first, T0, cg, gate, nullptr, nullptr, nullptr, nullptr, &r);
#ifdef FLS_TIMING
    const long long t_knn = (long long)__builtin_readcyclecounter();  // (the search of this wave is done)
#endif
    double T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = first ? T0.m[k] : st->T[k];
    const int lane = threadIdx.x & 63;
    double* row = &wsum[threadIdx.x >> 6][0];
    bool contrib = false;
    int id_l = -1;
    const bool owner = (threadIdx.x & 7) == 0 && r.q >= 0;
    // ids are reported for accepted correspondences only; with the fused tail they leave after the hand-off of the partial row (kernels_ivox_coop.hpp)
    auto store_ids = [&]() { if (owner) { nn_id[r.q] = id_l; eff[r.q] = contrib ? 1 : 0; } };
#if FLS_FIT_MFMA
    // the point's three Jacobian rows go to the matrix cores (kernels_p2plane.hpp::reduce_rank1x3_mfma_and_store): no per-point 21 + 6 products,
    // no 522-instruction DPP tree per wave
    double Jr[18], er[3] = {0.0, 0.0, 0.0}, res = 0.0;
#pragma unroll
    for (int k = 0; k < 18; ++k) Jr[k] = 0.0;
    if (owner) {
        if (r.found >= 1 && r.key != ~0ull && !((double)r.kth > max_corr)) {
            const float4 m = make_float4(r.bx, r.by, r.bz, __uint_as_float((unsigned)r.key));  // (= cg.g.pts[r.slot]: the key's low word is the point's id word)
            id_l = __float_as_int(m.w);
            icp_point_rows(T, r.px, r.py, r.pz, m, Jr, er, res);
            contrib = true;
        }
    }
    __shared__ __attribute__((aligned(64))) double mfma_tile[4][512];
    reduce_rank1x3_mfma_and_store(contrib, Jr, er, row, &mfma_tile[threadIdx.x >> 6][0]);
    const double sr = wave_sum_dpp(res);
    if (lane == 63) row[27] = sr;
#else
    double Hc[21], Bc[6], res = 0.0;
#pragma unroll
    for (int k = 0; k < 21; ++k) Hc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) Bc[k] = 0.0;
    if (owner) {
        if (r.found >= 1 && r.key != ~0ull && !((double)r.kth > max_corr)) {
            const float4 m = make_float4(r.bx, r.by, r.bz, __uint_as_float((unsigned)r.key));  // (= cg.g.pts[r.slot])
            id_l = __float_as_int(m.w);
            icp_point_terms(T, r.px, r.py, r.pz, m, Hc, Bc, res);
            contrib = true;
        }
    }
#pragma unroll
    for (int k = 0; k < 21; ++k) { const double v = wave_sum_dpp(Hc[k]); if (lane == 63) row[k] = v; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double v = wave_sum_dpp(Bc[k]); if (lane == 63) row[21 + k] = v; }
    const double sr = wave_sum_dpp(res), sc = wave_sum_dpp(contrib ? 1.0 : 0.0);
    if (lane == 63) { row[27] = sr; row[28] = sc; }
#endif
#ifdef FLS_TIMING
    const long long t_red = (long long)__builtin_readcyclecounter();  // (fit + matrix-core reduction + wave sums of this wave)
#endif
    if (!ticket) { store_ids(); block_row_from_wave_sums(wsum, partials); return; }
    // fused Gauss-Newton tail (round 3): the row goes out write-through, the last workgroup to arrive solves and publishes
    __syncthreads();
#ifdef FLS_TIMING
    const long long t_sync = (long long)__builtin_readcyclecounter();  // (the workgroup's slowest wave has arrived)
#endif
    const double v = threadIdx.x < 29 ? ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x] : 0.0;
    __shared__ unsigned s_ticket;
    __shared__ LuTailSmem sm;
    if (!publish_row_and_arrive(v, threadIdx.x < 29, partials, ticket, shards, s_ticket)) { store_ids(); return; }
#ifdef FLS_TIMING
    if (threadIdx.x == 0) { st->dbg[0] = t_begin; st->dbg[12] = t_knn; st->dbg[13] = t_red; st->dbg[14] = t_sync; }
#endif
    FLS_STAMP(1);
    lu_tail<256, true>(st, sm, partials, (int)gridDim.x, tail, T, it);
    store_ids();
}

__global__ void __launch_bounds__(256)
icp_fit_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
               const GnState* __restrict__ st, const int first, const Pose16 T0, const float4* __restrict__ nn_pts /* [n][1] */,
               const unsigned char* __restrict__ nn_cnt, const float* __restrict__ kth_d2, const double max_corr /* squared */,
               int* __restrict__ nn_id, unsigned char* __restrict__ eff, double* __restrict__ partials) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int done = first ? 0 : st->done;
    double T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = first ? T0.m[k] : st->T[k];
    if (done) return;
    __shared__ double wsum[4][32];
    double Hc[21], Bc[6], res = 0.0;
#pragma unroll
    for (int k = 0; k < 21; ++k) Hc[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) Bc[k] = 0.0;
    bool contrib = false;
    if (i < n) {
        int id = -1;
        if (nn_cnt[i] >= 1 && !((double)kth_d2[i] > max_corr)) {
            const float4 m = nn_pts[i];
            id = __float_as_int(m.w);
            icp_point_terms(T, sx[i], sy[i], sz[i], m, Hc, Bc, res);
            contrib = true;
        }
        nn_id[i] = id;  // ids are reported for accepted correspondences only
        eff[i] = contrib ? 1 : 0;
    }
    const int lane = threadIdx.x & 63;
    double* row = &wsum[threadIdx.x >> 6][0];
#pragma unroll
    for (int k = 0; k < 21; ++k) { const double v = wave_sum_dpp(Hc[k]); if (lane == 63) row[k] = v; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double v = wave_sum_dpp(Bc[k]); if (lane == 63) row[21 + k] = v; }
    const double sr = wave_sum_dpp(res), sc = wave_sum_dpp(contrib ? 1.0 : 0.0);
    if (lane == 63) { row[27] = sr; row[28] = sc; }
    block_row_from_wave_sums(wsum, partials);
}

// what the last workgroup of a fused fit + Gauss-Newton launch needs (LoamFull's dual launch; FLS_FUSED_TAIL)
struct LoamFusedTail {
    const double* partials_a;  // corner rows
    const double* partials_b;  // planar rows
    int nrows_a, nrows_b;
    double rot_thr, pos_thr;
    unsigned* ticket;
    int shards;
    Mailbox* mb;
    unsigned match_id;
};

// ---------------------------------------------------------------------------------------------
// point-to-plane / point-to-line on the 5 exact neighbours left by grid_knn_kernel<5>
// LINE = false: plane (Appendix C.1), LINE = true: corner feature (Appendix C.2)
// ---------------------------------------------------------------------------------------------
template <bool LINE>
__device__ __forceinline__ void
feature_fit_body(const int bid, const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                 const GnState* __restrict__ st, const int first, const Pose16& T0, const float4* __restrict__ nn_pts /* [n][5] */,
                 const unsigned char* __restrict__ nn_cnt, const float* __restrict__ kth_d2, const float gate, const double thres,
                 int* __restrict__ nn_id /* [n][5] */, unsigned char* __restrict__ cnt_out, double* __restrict__ Jst /* [7][n] */,
                 unsigned char* __restrict__ flag, double* __restrict__ partials, const LoamFusedTail* __restrict__ ft = nullptr) {
    const int i = bid * 256 + threadIdx.x;
    const int done = first ? 0 : st->done;
    double T44[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T44[k] = first ? T0.m[k] : st->T[k];
    // (fused tail: the state words the last workgroup needs, loaded before any waiting)
    const double last_rot = (ft && !first) ? st->last_rot : 0.0, last_pos = (ft && !first) ? st->last_pos : 0.0;
    const int it = (ft && !first) ? st->iter : 0;
    if (done) return;
    __shared__ double wsum[4][32];
    bool contrib = false, fresh = false, acc_l = false;
    double J[6] = {0, 0, 0, 0, 0, 0}, res = 0.0;
    int ids_l[5] = {-1, -1, -1, -1, -1};
    // the per-point outputs; with the fused tail they leave AFTER the hand-off of the partial row (kernels_ivox_coop.hpp: a barrier waits for
    // every store issued in front of it)
    auto store_outputs = [&]() {
        if (i >= n) return;
#pragma unroll
        for (int j = 0; j < 5; ++j) nn_id[(size_t)i * 5 + j] = ids_l[j];  // gate-accepted sets only
        cnt_out[i] = acc_l ? 5 : 0;
        if (fresh) {
#pragma unroll
            for (int a = 0; a < 6; ++a) Jst[(size_t)a * n + i] = J[a];
            Jst[(size_t)6 * n + i] = res;
            flag[i] = 1;
        }
    };
    if (i < n) {
        const bool accepted = nn_cnt[i] == 5 && !(kth_d2[i] > gate);
        float4 nn[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) nn[j] = nn_pts[(size_t)i * 5 + j];
#pragma unroll
        for (int j = 0; j < 5; ++j) ids_l[j] = accepted ? __float_as_int(nn[j].w) : -1;
        acc_l = accepted;
        bool valid_now = false;
        if (accepted) {
            const float px = sx[i], py = sy[i], pz = sz[i];
            const double x = px, y = py, z = pz;
            const float ptx = (float)(((T44[0] * x + T44[4] * y) + T44[8] * z) + T44[12]);
            const float pty = (float)(((T44[1] * x + T44[5] * y) + T44[9] * z) + T44[13]);
            const float ptz = (float)(((T44[2] * x + T44[6] * y) + T44[10] * z) + T44[14]);
            if (LINE) valid_now = line_residual_dev(nn, px, py, pz, ptx, pty, ptz, T44, thres, J, res);
            else valid_now = plane_residual_dev(nn, px, py, pz, ptx, pty, ptz, T44, thres, J, res);
        }
        if (valid_now) {
            fresh = true;
            contrib = true;
        } else if (flag[i]) {  // Q1 stale slot
#pragma unroll
            for (int a = 0; a < 6; ++a) J[a] = Jst[(size_t)a * n + i];
            res = Jst[(size_t)6 * n + i];
            contrib = true;
        }
    }
#if FLS_FIT_MFMA
    __shared__ __attribute__((aligned(64))) double mfma_tile[4][512];  // (kernels_p2plane.hpp::reduce_rank1_mfma_and_store)
    reduce_rank1_mfma_and_store(contrib, J, res, &wsum[threadIdx.x >> 6][0], &mfma_tile[threadIdx.x >> 6][0]);
#else
    reduce_rank1_and_store(contrib, J, res, &wsum[threadIdx.x >> 6][0]);
#endif
    if (!ft) { store_outputs(); block_row_from_wave_sums(wsum, partials, bid); return; }
    // fused Gauss-Newton tail (round 3, LOAM dual launch): the row goes out write-through, the last workgroup of the WHOLE launch (both
    // feature classes) sums the corner rows, then the planar rows (SumCoefficient's order) and runs the LOAM-family tail
    __syncthreads();
    const double v = threadIdx.x < 29 ? ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x] : 0.0;
    __shared__ unsigned s_ticket;
    __shared__ LoamTailSmem sm;
    if (threadIdx.x < 29)
        __hip_atomic_store((unsigned long long*)partials + (size_t)bid * kPartialStride + threadIdx.x, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = fanin_last_arriver(ft->ticket, ft->shards);
    __syncthreads();
    if (!s_ticket) { store_outputs(); return; }
    GnState* const stw = const_cast<GnState*>(st);
    loam_tail<256, true>(stw, sm, ft->partials_a, ft->nrows_a, ft->partials_b, ft->nrows_b, ft->rot_thr, ft->pos_thr, T44, last_rot, last_pos, it, ft->mb, ft->match_id);
    store_outputs();
}

template <bool LINE>
__global__ void __launch_bounds__(256)
feature_fit_kernel(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, const int n,
                   const GnState* __restrict__ st, const int first, const Pose16 T0, const float4* __restrict__ nn_pts /* [n][5] */,
                   const unsigned char* __restrict__ nn_cnt, const float* __restrict__ kth_d2, const float gate, const double thres,
                   int* __restrict__ nn_id /* [n][5] */, unsigned char* __restrict__ cnt_out, double* __restrict__ Jst /* [7][n] */,
                   unsigned char* __restrict__ flag, double* __restrict__ partials) {
    feature_fit_body<LINE>((int)blockIdx.x, sx, sy, sz, n, st, first, T0, nn_pts, nn_cnt, kth_d2, gate, thres, nn_id, cnt_out, Jst, flag, partials);
}

// corner (line) and planar fits of one LOAM iteration in one launch: blocks [0, nb_line) fit lines, the rest planes
struct FeatureFitArgs {
    const float *sx, *sy, *sz;
    int n;
    const float4* nn_pts;
    const unsigned char* nn_cnt;
    const float* kth_d2;
    float gate;
    double thres;
    int* nn_id;
    unsigned char* cnt_out;
    double* Jst;
    unsigned char* flag;
    double* partials;
};
template <bool FUSED>
__global__ void __launch_bounds__(256)
feature_fit_dual_kernel(const GnState* __restrict__ st, const int first, const Pose16 T0, const FeatureFitArgs line, const FeatureFitArgs plane, const int nb_line,
                        const LoamFusedTail tail /* FUSED = false: gn_solve_loam_kernel follows */) {
    const LoamFusedTail* const ft = FUSED ? &tail : nullptr;
    if ((int)blockIdx.x < nb_line)
        feature_fit_body<true>((int)blockIdx.x, line.sx, line.sy, line.sz, line.n, st, first, T0, line.nn_pts, line.nn_cnt, line.kth_d2, line.gate, line.thres,
                               line.nn_id, line.cnt_out, line.Jst, line.flag, line.partials, ft);
    else
        feature_fit_body<false>((int)blockIdx.x - nb_line, plane.sx, plane.sy, plane.sz, plane.n, st, first, T0, plane.nn_pts, plane.nn_cnt, plane.kth_d2,
                                plane.gate, plane.thres, plane.nn_id, plane.cnt_out, plane.Jst, plane.flag, plane.partials, ft);
}

}  // namespace fls
