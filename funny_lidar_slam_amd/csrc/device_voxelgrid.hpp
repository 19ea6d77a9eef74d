// device_voxelgrid.hpp -- host driver of the device VoxelGrid (kernels_voxelgrid.hpp; contract there).
//
// Input: x | y | z | intensity of n points, resident on the device.  Output: the filtered cloud as x | y | z | intensity
// on the device (ascending leaf index), its size on the host.  Two short host waits per call: the bounds (the leaf box
// decides the number of radix passes and the "leaf size too small" refusal, voxel_grid.hpp:69-74) and the output size.
#pragma once
#include "kernels_voxelgrid.hpp"
#include "kernels_exactsort.hpp"
#include "kernels_voxelgrid_plan.hpp"
#include "kernels_gridbuild.hpp"
#include "matcher_base.hpp"
#include <atomic>
#include <cstdlib>
#include <climits>

namespace fls {

// blocks of vg_minmax (six header atomics each, all on one line): 48 up to 200 k points as in rounds 2-5 ("few blocks"), one per 4,096 points beyond, at
// most 256.  Round 6: 48 blocks left a 1.55 M-point keyframe deque at 127 points per thread, 58 us in the trace (17 us with 256 blocks); 256 blocks for
// EVERY size cost the 115,200-point scans 30 us of serialised atomics (the index-order A/B legs, +12 %: tools/bench_diff.py caught it).
inline int minmax_blocks(const size_t n) { return int(std::min<size_t>(256, std::max<size_t>(48, n / 4096))); }

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

// std::sort's permutation of {key, value} records on the device (kernels_exactsort.hpp), in place.  Clouds up to kEsTaskMax records
// (every source scan) are ONE begin launch + ONE persistent task launch; beyond that the host steers the level-synchronous top of the
// recursion: it enqueues the expected number of levels, polls a host-mapped word (no stream synchronisation) and tops up two levels
// at a time while ranges longer than kEsTaskMax remain.
// the host-mapped words of the last sort started with FLS_ES_DEBUG set (diagnostics: fls_debug_exact_sort_marks reads them from another host thread
// while a sort is in flight, so the pointer is atomic and published only in debug runs; the owner withdraws it before freeing the memory)
inline std::atomic<EsMailbox*>& es_debug_mailbox() { static std::atomic<EsMailbox*> p{nullptr}; return p; }
struct DeviceExactSort {
    DevBuf<EsSeg> seg_a, seg_b;
    DevBuf<EsWork> work;
    DevBuf<uint2> tile_cnt;
    DevBuf<unsigned long long> tile_pub;  // es_count_scatter_kernel: a tile's published counts + the launch's epoch
    unsigned pub_epoch = 0;
    DevBuf<unsigned> tile_seg, Lp, Rl;
    DevBuf<EsState> st;
    DevBuf<EsQueue> queue;
    DevBuf<unsigned> ready;
    PinnedBuf<EsState> h_st;
    PinnedBuf<EsQueue> h_queue;
    EsMailbox* mb_host = nullptr;
    EsMailbox* mb_dev = nullptr;
    unsigned seq = 0;
    unsigned long long runs = 0, failures = 0, levels = 0;
    int last_levels = 0;   // host-steered path: levels queued by the previous sort, and its size
    size_t last_n = 0;
    ~DeviceExactSort() {
        EsMailbox* mine = mb_host;
        es_debug_mailbox().compare_exchange_strong(mine, nullptr);
        if (mb_host) (void)hipHostFree(mb_host);
    }
    unsigned wait(const unsigned want, hipStream_t s) {
        for (unsigned long long spin = 1;; ++spin) {
            if (__atomic_load_n(&mb_host->seq, __ATOMIC_ACQUIRE) == want) return want;
            if ((spin & 0x3fffu) == 0) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) return __atomic_load_n(&mb_host->seq, __ATOMIC_ACQUIRE);
                if (q != hipErrorNotReady) FLS_HIP(q);
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
    }
    unsigned work_cap = 0, tile_cap = 0;
    // records a workgroup of the task kernel sorts in LDS, by cloud size (kernels_exactsort.hpp, FLS_ES_LDS): FLS_ES_LDS_SMALL / FLS_ES_LDS_BIG override (A/B)
    static unsigned lds_cap_for(const size_t n) {
        static const unsigned small = [] { const char* e = std::getenv("FLS_ES_LDS_SMALL"); const int v = e ? std::atoi(e) : kEsLdsSmall; return unsigned(std::min(std::max(v, 64), kEsLds)); }();
        static const unsigned big = [] { const char* e = std::getenv("FLS_ES_LDS_BIG"); const int v = e ? std::atoi(e) : kEsLds; return unsigned(std::min(std::max(v, 64), kEsLds)); }();
        return n <= size_t(kEsTaskMax) ? small : big;
    }
    // one workgroup per CU at most; as many as there are 2,048-record pieces -- NOT n / lds_cap: with 8,192-record LDS ranges that left the 475,200-record
    // deque of IcpOptimized to 62 workgroups, five ranges of ~76 us each in a row on the busiest one (profiles/r06_vg_large_cloud_filters.txt)
    static unsigned task_grid(const size_t n) {
        static const size_t cap = [] { const char* e = std::getenv("FLS_ES_GRID"); const int v = e ? std::atoi(e) : 256; return size_t(std::min(std::max(v, 8), 256)); }();  // (A/B)
        return unsigned(std::min<size_t>(cap, std::max<size_t>(8, n / kEsLdsSmall + 4)));
    }
    // clouds beyond kEsTaskMax records: ranges longer than this stay with the level-synchronous launches (the host steers them), shorter ones are tasks.
    // kEsTaskMax itself until round 6; the deepest chain of one-workgroup partitions below a 131,072-record range (~290 us at 3 us + 0.38 us per thousand
    // records and 13/16 splits) was the task kernel's critical path.  Measured with the ticket queue, 32,768 / 65,536 / 131,072: planar deque 0.80 / 0.75 /
    // 0.93 ms, IcpOptimized deque 0.48 / 0.53 / 0.65 ms, corner deque 0.37 / 0.42 / 0.44 ms (profiles/r06_h_*) -- FLS_ES_HANDOVER for A/B
    static unsigned handover_threshold() {
        static const unsigned v = [] { const char* e = std::getenv("FLS_ES_HANDOVER"); const int x = e ? std::atoi(e) : 32768; return unsigned(std::min(std::max(x, 4096), kEsTaskMax)); }();
        return v;
    }
    void allocate(const size_t n) {
        if (!mb_host) {
            FLS_HIP(hipHostMalloc((void**)&mb_host, sizeof(EsMailbox), hipHostMallocMapped));
            std::memset(mb_host, 0, sizeof(EsMailbox));
            FLS_HIP(hipHostGetDevicePointer((void**)&mb_dev, mb_host, 0));
        }
        static const bool publish_dbg = std::getenv("FLS_ES_DEBUG") != nullptr;
        if (publish_dbg) es_debug_mailbox().store(mb_host, std::memory_order_release);
        work_cap = unsigned(64 * (n / kEsLdsSmall + 1) + 1024);
        tile_cap = unsigned(n / kEsTile + kEsMaxSeg + 2);
        seg_a.reserve(kEsMaxSeg); seg_b.reserve(kEsMaxSeg);
        work.reserve(work_cap);
        ready.reserve(work_cap);
        tile_cnt.reserve(tile_cap); tile_seg.reserve(tile_cap);
        if (tile_pub.cap < tile_cap) {  // a fresh buffer holds epoch 0, which no launch uses (growth is rare: filled and waited for here)
            tile_pub.reserve(tile_cap);
            FLS_HIP(hipMemset(tile_pub.p, 0, tile_pub.cap * sizeof(unsigned long long)));
            FLS_HIP(hipDeviceSynchronize());
        }
        Lp.reserve(n); Rl.reserve(n);
        st.reserve(1);
        queue.reserve(1);
        h_st.reserve(1);
        h_queue.reserve(1);
    }
    // The one-launch form for clouds up to kEsTaskMax records whose queue a preceding kernel initialises (vg_minmax_plan, EsInitArgs):
    // fused_prepare() before that kernel is queued, fused_launch() behind the kernel that writes the records.  `skip`: device word, non-zero
    // = nothing to sort.  The verdict (EsState::fail) stays on the device: the caller's last kernel forwards it.
    // (kEsTaskMax until round 6: beyond it the host steered the level-synchronous top through a mailbox, two stream synchronisations and a round trip per
    // top-up.  The pre-enqueued guess serves any size: levels that find nothing left are three ~2 us launches, ranges the guess leaves too long become
    // tasks.  FLS_VG_FUSED_MAX for A/B)
    bool fused_ok(const size_t n) const {
        static const size_t lim = [] { const char* e = std::getenv("FLS_VG_FUSED_MAX"); return e ? size_t(std::atoll(e)) : (size_t(1) << 22); }();
        return n >= 2 && n <= lim;
    }
    EsInitArgs fused_prepare(const size_t n) {
        allocate(n);
        return EsInitArgs{st.p, queue.p, ready.p, work_cap};
    }
    // Ranges longer than `big` records are partitioned LEVEL-SYNCHRONOUSLY by the whole device -- three launches per level (es_level_begin / count_scatter /
    // swap, the regime-1 kernels), pre-enqueued without a host round trip: one workgroup needs 3 us + 0.38 us per thousand records
    // for a partition (profiles/r05_d_ndt_global_levels.log: 47 us at 115,200), a level of launches ~16 us whatever the size.  The number of
    // levels is a guess (the larger child keeps ~13/16 of a LiDAR range); levels that find nothing left are empty launches, ranges still longer
    // than `big` after the last one are the task kernel's (one workgroup each, as in round 4).  FLS_ES_BIG = 0 switches the top levels off.
    static unsigned big_threshold() {
        static const unsigned v = [] { const char* e = std::getenv("FLS_ES_BIG"); return e ? unsigned(std::atoi(e)) : 32768u; }();
        return v;
    }
    void fused_launch(unsigned* key, unsigned* val, const size_t n, const unsigned* skip, hipStream_t s) {
        ++runs;
        const unsigned grid = task_grid(n);
        static const bool dbg_marks = std::getenv("FLS_ES_DEBUG") != nullptr;
        if (dbg_marks) std::memset(mb_host->mark, 0, sizeof(mb_host->mark));
        const unsigned big = big_threshold();
        int top = 0;
        if (big != 0u && n > size_t(big)) {
            // (a single scan keeps ~13/16 of a range in the larger child; the keyframe deques beyond kEsTaskMax records -- the same surfaces many times
            // over -- split more evenly: ~0.7 measured, 15 levels for 1.55 M records where 13/16 all the way guesses 19, each empty level three launches)
            size_t m = n;
            for (; m > size_t(kEsTaskMax) && m > size_t(big); m = m * 7 / 10) ++top;
            for (; m > size_t(big); m = m * 13 / 16) ++top;
            static const int extra = [] { const char* e = std::getenv("FLS_ES_TOP_EXTRA"); return e ? std::atoi(e) : 0; }();
            top = std::max(1, top + extra);
        }
        if (top == 0) {
            hipLaunchKernelGGL(es_task_kernel, dim3(grid), dim3(kEsTaskThreads), 0, s, key, val, work.p, ready.p, work_cap, queue.p, Lp.p, Rl.p, st.p,
                               dbg_marks ? mb_dev : (EsMailbox*)nullptr, unsigned(n), skip, lds_cap_for(n));
            return;
        }
        auto next_seq = [&]() { seq = (seq + 1u) & 0x7fffffffu; if (!seq) seq = 1u; return seq; };
        EsSeg* prev = seg_a.p;
        EsSeg* cur = seg_b.p;
        hipLaunchKernelGGL(es_level_begin, dim3(1), dim3(256), 0, s, key, val, unsigned(n), (const EsSeg*)prev, cur, work.p, work_cap, (const unsigned*)Lp.p,
                           (const unsigned*)Rl.p, st.p, (EsMailbox*)nullptr, next_seq(), 1, tile_seg.p, tile_cap, big, 0, queue.p, skip);
        for (int l = 0; l < top; ++l) {
            launch_lists(key, cur, s);
            hipLaunchKernelGGL(es_swap_kernel, dim3(tile_cap), dim3(kEsBlock), 0, s, key, val, cur, (const EsState*)st.p, (const unsigned*)tile_seg.p,
                               (const unsigned*)Lp.p, (const unsigned*)Rl.p);
            std::swap(prev, cur);
            hipLaunchKernelGGL(es_level_begin, dim3(1), dim3(256), 0, s, key, val, unsigned(n), (const EsSeg*)prev, cur, work.p, work_cap, (const unsigned*)Lp.p,
                               (const unsigned*)Rl.p, st.p, (EsMailbox*)nullptr, next_seq(), 0, tile_seg.p, tile_cap, big, l == top - 1 ? 1 : 0, queue.p, skip);
            ++levels;
        }
        hipLaunchKernelGGL(es_task_kernel, dim3(grid), dim3(kEsTaskThreads), 0, s, key, val, work.p, ready.p, work_cap, queue.p, Lp.p, Rl.p, st.p,
                           dbg_marks ? mb_dev : (EsMailbox*)nullptr, 0u, skip, lds_cap_for(n));
    }
    // the stop lists of a level: one launch (es_count_scatter_kernel: tiles look back at their predecessors' published counts), or the two
    // launches it replaces (FLS_ES_LOOKBACK=0: counts, launch boundary, lists)
    void launch_lists(const unsigned* key, EsSeg* cur, hipStream_t s) {
        static const bool lookback = [] { const char* e = std::getenv("FLS_ES_LOOKBACK"); return e ? std::atoi(e) != 0 : true; }();
        if (lookback) {
            if (++pub_epoch == 0u) ++pub_epoch;  // (0 is what a fresh buffer holds)
            hipLaunchKernelGGL(es_count_scatter_kernel, dim3(tile_cap), dim3(kEsBlock), 0, s, key, cur, st.p, (const unsigned*)tile_seg.p, tile_pub.p, pub_epoch,
                               Lp.p, Rl.p);
            return;
        }
        hipLaunchKernelGGL(es_count_kernel, dim3(tile_cap), dim3(kEsBlock), 0, s, key, (const EsSeg*)cur, (const EsState*)st.p, (const unsigned*)tile_seg.p,
                           tile_cnt.p);
        hipLaunchKernelGGL(es_scatter_kernel, dim3(tile_cap), dim3(kEsBlock), 0, s, key, cur, (const EsState*)st.p, (const unsigned*)tile_seg.p,
                           (const uint2*)tile_cnt.p, Lp.p, Rl.p);
    }
    // queues the whole sort on `s`; false: refused before anything ran (sizes).  The verdict of the sort itself (introsort's heap-sort
    // case) is only known once the stream has drained: failed_after_sync().
    bool run(unsigned* key, unsigned* val, const size_t n, hipStream_t s) {
        if (n > (size_t(1) << 22)) return false;
        if (n < 2) return true;
        allocate(n);
        ++runs;
        FLS_HIP(hipMemsetAsync(ready.p, 0, work_cap * sizeof(unsigned), s));
        if (n <= size_t(kEsTaskMax)) {
            // every source scan: no begin launch, no host wait -- the array itself is workgroup 0's first task (open = 1 stands for it)
            FLS_HIP(hipMemsetAsync(st.p, 0, sizeof(EsState), s));
            *h_queue.p = EsQueue{0u, 0u, 1u, 0u};
            FLS_HIP(hipMemcpyAsync(queue.p, h_queue.p, sizeof(EsQueue), hipMemcpyHostToDevice, s));
            const unsigned grid = task_grid(n);
            static const bool dbg_marks = std::getenv("FLS_ES_DEBUG") != nullptr;
            if (dbg_marks) std::memset(mb_host->mark, 0, sizeof(mb_host->mark));
            hipLaunchKernelGGL(es_task_kernel, dim3(grid), dim3(kEsTaskThreads), 0, s, key, val, work.p, ready.p, work_cap, queue.p, Lp.p, Rl.p, st.p,
                               dbg_marks ? mb_dev : (EsMailbox*)nullptr, unsigned(n), (const unsigned*)nullptr, lds_cap_for(n));
            FLS_HIP(hipMemcpyAsync(h_st.p, st.p, sizeof(EsState), hipMemcpyDeviceToHost, s));
            FLS_HIP(hipGetLastError());
            return true;
        }
        auto next_seq = [&]() { seq = (seq + 1u) & 0x7fffffffu; if (!seq) seq = 1u; return seq; };
        EsSeg* prev = seg_a.p;
        EsSeg* cur = seg_b.p;
        hipLaunchKernelGGL(es_level_begin, dim3(1), dim3(256), 0, s, key, val, unsigned(n), (const EsSeg*)prev, cur, work.p, work_cap, (const unsigned*)Lp.p,
                           (const unsigned*)Rl.p, st.p, mb_dev, next_seq(), 1, tile_seg.p, tile_cap, handover_threshold(), 0, (EsQueue*)nullptr, (const unsigned*)nullptr);
        // regime 1 (ranges longer than the hand-over threshold: only clouds beyond 131 k points get here)
        // how many levels to queue before the first look at the mailbox: what the previous sort of this object needed (a keyframe deque changes by one
        // frame in twenty-five between two calls), else a guess from halving splits -- LiDAR leaf indices split ~13/16, so the guess is short and the
        // loop below tops up two levels at a time, one host round trip (~10 us of idle device) each
        int expected = 0;
        for (size_t m = n; m > size_t(handover_threshold()); m = (m + 1) / 2) ++expected;
        int chunk = expected ? expected + 1 : 0;
        if (expected && last_levels > 0 && last_n != 0 && n >= last_n / 2 && n <= last_n * 2) chunk = std::max(1, last_levels);
        int queued_levels = 0;
        for (;;) {
            queued_levels += chunk;
            for (int c = 0; c < chunk; ++c) {
                launch_lists(key, cur, s);
                hipLaunchKernelGGL(es_swap_kernel, dim3(tile_cap), dim3(kEsBlock), 0, s, key, val, cur, (const EsState*)st.p, (const unsigned*)tile_seg.p,
                                   (const unsigned*)Lp.p, (const unsigned*)Rl.p);
                std::swap(prev, cur);
                hipLaunchKernelGGL(es_level_begin, dim3(1), dim3(256), 0, s, key, val, unsigned(n), (const EsSeg*)prev, cur, work.p, work_cap, (const unsigned*)Lp.p,
                                   (const unsigned*)Rl.p, st.p, mb_dev, next_seq(), 0, tile_seg.p, tile_cap, handover_threshold(), 0, (EsQueue*)nullptr, (const unsigned*)nullptr);
                ++levels;
            }
            FLS_HIP(hipGetLastError());
            wait(seq, s);
            if (mb_host->fail) { ++failures; return false; }
            if (mb_host->n_cur == 0u) break;
            chunk = 2;
        }
        last_levels = queued_levels; last_n = n;  // (an over-estimate by at most one top-up: empty levels cost three ~2 us launches)
        // regimes 2 + 3: one persistent launch over the task queue (the ranges regime 1 handed over are its first tasks)
        const unsigned n_work = mb_host->n_work;
        if (n_work) {
            *h_queue.p = EsQueue{0u, n_work, n_work, n_work};
            FLS_HIP(hipMemcpyAsync(queue.p, h_queue.p, sizeof(EsQueue), hipMemcpyHostToDevice, s));
            const unsigned grid = task_grid(n);
            static const bool skip = std::getenv("FLS_ES_SKIP_TASKS") != nullptr;  // (bisecting aid)
            if (!skip) hipLaunchKernelGGL(es_task_kernel, dim3(grid), dim3(kEsTaskThreads), 0, s, key, val, work.p, ready.p, work_cap, queue.p, Lp.p, Rl.p, st.p,
                                          std::getenv("FLS_ES_DEBUG") ? mb_dev : (EsMailbox*)nullptr, 0u, (const unsigned*)nullptr, lds_cap_for(n));
            FLS_HIP(hipMemcpyAsync(h_queue.p, queue.p, sizeof(EsQueue), hipMemcpyDeviceToHost, s));
        }
        FLS_HIP(hipMemcpyAsync(h_st.p, st.p, sizeof(EsState), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipGetLastError());
        return true;
    }
    void print_wg_profile() {
        if (!mb_host) return;
        {  // where the workgroups of the task kernel spent their time (100 MHz ticks -> us)
            double w = 0, g = 0, l = 0, wmax = 0, bmax = 0; unsigned nt = 0, used = 0;
            for (int b = 0; b < 256; ++b) {
                const unsigned* q = mb_host->wg[b];
                if (!q[0] && !q[1] && !q[2]) continue;
                ++used; w += q[0]; g += q[1]; l += q[2]; nt += q[3];
                wmax = std::max(wmax, double(q[0])); bmax = std::max(bmax, double(q[1]) + double(q[2]));
            }
            if (used)
                std::fprintf(stderr, "[fls exact sort] task kernel, %u workgroups, mean per workgroup [us]: waiting for tasks %.1f, partitions out of global memory %.1f, LDS ranges %.1f; "
                             "busiest workgroup %.1f us busy, longest wait %.1f us; %u tasks popped\n", used, 0.01 * w / used, 0.01 * g / used, 0.01 * l / used, 0.01 * bmax, 0.01 * wmax, nt);
            // the eight busiest workgroups, one by one
            std::vector<int> order;
            for (int b = 0; b < 256; ++b) if (mb_host->wg[b][1] || mb_host->wg[b][2]) order.push_back(b);
            std::sort(order.begin(), order.end(), [&](int a, int b) { return mb_host->wg[a][1] + mb_host->wg[a][2] > mb_host->wg[b][1] + mb_host->wg[b][2]; });
            for (size_t i = 0; i < order.size() && i < 8; ++i) {
                const unsigned* q = mb_host->wg[order[i]];
                std::fprintf(stderr, "[fls exact sort]   workgroup %3d: %u tasks, wait %.1f, global %.1f, LDS %.1f us | longest LDS range %.1f us (%u records) | longest global chain %.1f us (from %u records)\n",
                             order[i], q[3], 0.01 * q[0], 0.01 * q[1], 0.01 * q[2], 0.01 * q[4], q[5], 0.01 * q[6], q[7]);
            }
            std::memset(mb_host->wg, 0, sizeof(mb_host->wg));
        }
    }
    // FLS_ES_DEBUG: stage stamps of the last sort (the fused form never copies EsState back)
    bool failed_after_sync_debug_only() {
        static const bool dbg = std::getenv("FLS_ES_DEBUG") && std::atoi(std::getenv("FLS_ES_DEBUG")) != 0;
        if (dbg && mb_host) {
            const unsigned* mk = mb_host->mark;
            auto us = [&](int a, int b) { return mk[a] && mk[b] ? 0.01 * double(int(mk[b] - mk[a])) : -1.0; };
            std::fprintf(stderr, "[fls exact sort] workgroup 0, first task [us]: pop->start %.1f, global partitions %.1f, LDS load %.1f, phase A (workgroup partitions) %.1f, phase B (wave tasks) %.1f, "
                         "ranks + write-back %.1f\n", us(0, 2), us(2, 9), us(9, 10), us(10, 3), us(3, 4), us(5, 6));
            for (int i = 0; i < 32 && mb_host->lvl[i][0]; ++i) {
                const unsigned* q = mb_host->lvl[i];
                auto d = [&](int a, int b) { return 0.01 * double(int(q[b] - q[a])); };
                std::fprintf(stderr, "[fls exact sort]   global partition %2d: %6u records | median %.1f | count %.1f | lists %.1f | swaps %.1f | cut + push %.1f | total %.1f us\n", i, q[0], d(1, 2), d(2, 3),
                             d(3, 4), d(4, 5), d(5, 6), d(1, 6));
            }
            std::memset(mb_host->lvl, 0, sizeof(mb_host->lvl));
            print_wg_profile();
        }
        return false;
    }
    // after the caller's stream synchronisation: did a range hit introsort's depth limit inside es_lds_kernel?
    bool failed_after_sync() {
        static const bool dbg = std::getenv("FLS_ES_DEBUG") && std::atoi(std::getenv("FLS_ES_DEBUG")) != 0;
        if (dbg && mb_host) {
            const unsigned* mk = mb_host->mark;
            auto us = [&](int a, int b) { return mk[a] && mk[b] ? 0.01 * double(int(mk[b] - mk[a])) : -1.0; };
            std::fprintf(stderr, "[fls exact sort] workgroup 0, first task [us]: pop->start %.1f, global partitions %.1f, LDS load %.1f, phase A (workgroup partitions) %.1f, phase B (wave tasks) %.1f, "
                         "ranks + write-back %.1f\n", us(0, 2), us(2, 9), us(9, 10), us(10, 3), us(3, 4), us(5, 6));
        }
        if (dbg) print_wg_profile();
        if (dbg && h_st.p)
            std::fprintf(stderr, "[fls exact sort] regime-1 levels %u, hand-over ranges %u, fail %u; task kernel: %u partitions from global memory, %u ranges (%u records) sorted in LDS\n",
                         h_st.p->level, h_st.p->n_work, h_st.p->fail, h_st.p->pad[0], h_st.p->pad[1], h_st.p->pad[2]);
        if (h_st.p && h_st.p->fail) { ++failures; return true; }
        return false;
    }
};

// FLS_DEVICE_VOXELGRID: 1 (default) the device filters, leaf sums in std::sort's order = bit-identical to pcl::VoxelGrid (round 4);
// 0 the exact host filter (worker pool); 2 the device filters with the round-2/3 contract (stable radix sort: leaf sums in ascending
// point index, last-bit differences on leaves of three or more points) -- A/B only.
inline int device_voxelgrid_mode() {
    if (const char* e = std::getenv("FLS_DEVICE_VOXELGRID")) { const int v = std::atoi(e); return v < 0 ? 0 : v > 2 ? 1 : v; }
    return 1;
}

struct DeviceVoxelGrid {
    DevicePairSort sort;
    DeviceExactSort exact;
    bool exact_order = device_voxelgrid_mode() != 2;  // centroids summed in std::sort's order; false: ascending point index
    unsigned long long exact_runs = 0, exact_declined = 0;
    DevBuf<unsigned> lx, bt;        // bt: block totals | scanned
    DevBuf<float4> sorted;          // the points in sorted order
    DevBuf<float> out;            // x | y | z | i, capacity n each
    DevBuf<VgHeader> d_hdr;
    PinnedBuf<VgHeader> h_hdr;    // [0] = init template, [1] = read-back
    // the uninterrupted form (kernels_voxelgrid_plan.hpp): plan + accumulators on the device, the verdict in a host-mapped mailbox
    DevBuf<VgPlan> d_plan;
    DevBuf<VgAccum> d_acc;
    VgMailbox* vmb_host = nullptr;
    VgMailbox* vmb_dev = nullptr;
    unsigned vseq = 0;
    bool fused = true;            // FLS_VG_FUSED=0: the round-4 sequence (two stream synchronisations), A/B
    unsigned last_status = 0;
    unsigned long long fused_runs = 0;
    size_t n_in = 0, n_out = 0;
    int last_passes = 0;
    ~DeviceVoxelGrid() { if (vmb_host) (void)hipHostFree(vmb_host); }
    DeviceVoxelGrid() { if (const char* e = std::getenv("FLS_VG_FUSED")) fused = std::atoi(e) != 0; }
    DeviceVoxelGrid(const DeviceVoxelGrid&) = delete;
    DeviceVoxelGrid& operator=(const DeviceVoxelGrid&) = delete;
    const float* ox() const { return out.p; }
    const float* oy() const { return out.p + n_in; }
    const float* oz() const { return out.p + 2 * n_in; }
    const float* oi() const { return out.p + 3 * n_in; }

    // false: not applicable (empty, too large, or PCL's "leaf size too small" case) -- the caller takes the host path
    bool run(const float* x, const float* y, const float* z, const float* in, size_t n, float leaf, hipStream_t s) {
        n_in = n;
        n_out = 0;
        if (n == 0 || n > size_t(kVgMaxBlocks) * kVgTile) return false;
        if (fused && exact_order && exact.fused_ok(n)) return run_fused(x, y, z, in, n, leaf, s);
        h_hdr.reserve(2);
        d_hdr.reserve(1);
        for (int a = 0; a < 3; ++a) { h_hdr.p[0].mn[a] = 0xffffffffu; h_hdr.p[0].mx[a] = 0u; }
        h_hdr.p[0].n_out = 0u; h_hdr.p[0].n_bad = 0u;
        FLS_HIP(hipMemcpyAsync(d_hdr.p, &h_hdr.p[0], sizeof(VgHeader), hipMemcpyHostToDevice, s));
        const int ni = int(n);
        const int nb1 = (ni + kVgBlock - 1) / kVgBlock, nb2 = (ni + kVgScanBlock - 1) / kVgScanBlock;
        hipLaunchKernelGGL(vg_minmax, dim3(unsigned(std::min(nb1, minmax_blocks(n)))), dim3(kVgBlock), 0, s, x, y, z, ni, d_hdr.p);
        FLS_HIP(hipMemcpyAsync(&h_hdr.p[1], d_hdr.p, sizeof(VgHeader), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        const VgHeader& hh = h_hdr.p[1];
        if (hh.mn[0] == 0xffffffffu) return false;  // no finite point: the host path returns the empty cloud
        const float inv = 1.0f / leaf;
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) { mn[a] = vg_unord(hh.mn[a]); mx[a] = vg_unord(hh.mx[a]); }
        const float ex = (mx[0] - mn[0]) * inv, ey = (mx[1] - mn[1]) * inv, ez = (mx[2] - mn[2]) * inv;
        if (!(ex < 2147483648.0f && ey < 2147483648.0f && ez < 2147483648.0f)) return false;  // (one extent of 2^31 leaves is already beyond INT_MAX; keeps the products inside int64)
        const long long dx = (long long)ex + 1, dy = (long long)ey + 1, dz = (long long)ez + 1;
        if (dx * dy > (long long)INT_MAX || dx * dy * dz > (long long)INT_MAX) return false;
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
        if (exact_order) {
            // the reference sorts the FINITE points only (non-finite ones never enter its index vector): a cloud with any is the host's
            if (hh.n_bad != 0u || !exact.run(sort.k0, sort.v0, n, s)) { ++exact_declined; return false; }
            ++exact_runs;
        } else {
            sort.run(passes, s);
        }
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
        if (exact_order && exact.failed_after_sync()) { ++exact_declined; return false; }  // (introsort's heap-sort case: the host path sorts)
        n_out = h_hdr.p[1].n_out;
        return true;
    }

    // One uninterrupted stream of launches (kernels_voxelgrid_plan.hpp): bounds + plan + sort-queue initialisation, leaf indices, the exact
    // sort, run heads, offsets + verdict to the mailbox, centroids.  The host waits ONCE, on a host-mapped word the offsets kernel writes --
    // before the centroid kernel has finished; whatever the caller queues next on `s` is ordered behind it.
    bool run_fused(const float* x, const float* y, const float* z, const float* in, size_t n, float leaf, hipStream_t s) {
        if (!vmb_host) {
            FLS_HIP(hipHostMalloc((void**)&vmb_host, sizeof(VgMailbox), hipHostMallocMapped));
            std::memset(vmb_host, 0, sizeof(VgMailbox));
            FLS_HIP(hipHostGetDevicePointer((void**)&vmb_dev, vmb_host, 0));
        }
        if (!d_acc.p) {  // armed once; the last block of every vg_minmax_plan re-arms it
            d_acc.reserve(1);
            d_plan.reserve(1);
            FLS_HIP(hipMemsetAsync(d_acc.p, 0, sizeof(VgAccum), s));  // (the ticket; the rows are written before they are read)
            FLS_HIP(hipStreamSynchronize(s));
        }
        const int ni = int(n);
        const int nb1 = (ni + kVgBlock - 1) / kVgBlock, nb2 = (ni + kVgScanBlock - 1) / kVgScanBlock;
        sort.prepare(n);
        lx.reserve(n);
        bt.reserve(size_t(2 * nb2));
        sorted.reserve(n);
        out.reserve(4 * n);
        const EsInitArgs es = exact.fused_prepare(n);
        vseq = (vseq + 1u) & 0x7fffffffu;
        if (!vseq) vseq = 1u;
        const float inv = 1.0f / leaf;
        // (the reference sorts the FINITE points only: a cloud with a non-finite point is the host's, refuse_bad = 1)
        // (blocks: 48 in rounds 2-4 -- "few blocks: six header atomics each" -- left a 115,200-point scan to 12 k threads, 11-12 us in the trace
        // of the call; 160 blocks = three points per thread and < 1,000 atomics: FLS_VG_MINMAX_BLOCKS for A/B)
        static const int mm_env = [] { const char* e = std::getenv("FLS_VG_MINMAX_BLOCKS"); return e ? std::min(kVgMinmaxMaxBlocks, std::max(1, std::atoi(e))) : 0; }();
        const int mm_blocks = mm_env ? mm_env : int(std::min<size_t>(kVgMinmaxMaxBlocks, std::max<size_t>(160, n / 4096)));  // (one row per block, no atomics but the ticket: more blocks for the keyframe deques)
        hipLaunchKernelGGL(vg_minmax_plan, dim3(unsigned(std::min(nb1, mm_blocks))), dim3(kVgBlock), 0, s, x, y, z, ni, inv, 1, d_acc.p, d_plan.p, es);
        hipLaunchKernelGGL(vg_index_plan, dim3(unsigned(nb1)), dim3(kVgBlock), 0, s, x, y, z, ni, (const VgPlan*)d_plan.p, sort.k0, sort.v0);
        exact.fused_launch(sort.k0, sort.v0, n, &d_plan.p->status, s);
        hipLaunchKernelGGL(vg_heads_plan, dim3(unsigned(nb2)), dim3(kVgScanBlock), 0, s, (const unsigned*)sort.k0, (const unsigned*)sort.v0, ni, (const VgPlan*)d_plan.p,
                           (const EsState*)exact.st.p, x, y, z, in, sorted.p, lx.p, bt.p);
        hipLaunchKernelGGL(vg_scan_publish, dim3(1), dim3(kVgScanBlock), 0, s, (const unsigned*)bt.p, bt.p + nb2, nb2, (const VgPlan*)d_plan.p, (const EsState*)exact.st.p,
                           vmb_dev, vseq);
        hipLaunchKernelGGL(vg_centroid_plan, dim3(unsigned(nb1)), dim3(kVgBlock), 0, s, (const unsigned*)sort.k0, (const float4*)sorted.p, ni, (const VgPlan*)d_plan.p,
                           (const EsState*)exact.st.p, (const unsigned*)lx.p, (const unsigned*)(bt.p + nb2), out.p, out.p + n, out.p + 2 * n, out.p + 3 * n);
        FLS_HIP(hipGetLastError());
        for (unsigned long long spin = 1;; ++spin) {
            if (__atomic_load_n(&vmb_host->seq, __ATOMIC_ACQUIRE) == vseq) break;
            if ((spin & 0x3fffu) == 0) {
                const hipError_t q = hipStreamQuery(s);
                if (q == hipSuccess) { if (__atomic_load_n(&vmb_host->seq, __ATOMIC_ACQUIRE) == vseq) break; FLS_HIP(hipErrorUnknown); }
                if (q != hipErrorNotReady) FLS_HIP(q);
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        ++fused_runs;
        last_status = vmb_host->status;
        if (exact.failed_after_sync_debug_only()) {}
        if (vmb_host->status != kVgOk || vmb_host->sort_fail != 0u) {
            if (vmb_host->sort_fail != 0u) ++exact.failures;
            ++exact_declined;
            FLS_HIP(hipStreamSynchronize(s));  // (the staging the caller reuses; the refused call is rare)
            return false;
        }
        ++exact_runs;
        n_out = vmb_host->n_out;
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

// ---------------------------------------------------------------------------------------------
// The kd-tree kinds' cell grid built on the device (kernels_gridbuild.hpp): exact, so it is the default
// (FLS_DEVICE_GRID_BUILD=0 restores the host build).  Input: the map cloud as x | y | z planes resident on the device.
// One short host wait (the bounds decide the window).  false: declined (empty, a non-finite point, cells outside the key
// range, a window larger than kMaxWindowCells) -- the caller takes the host build, which also produces the error codes.
// ---------------------------------------------------------------------------------------------
struct DeviceGridBuilder {
    DevicePairSort sort;
    DevBuf<VgHeader> d_hdr;
    PinnedBuf<VgHeader> h_hdr;
    unsigned long long builds = 0, declined = 0;
    bool run(CellGridImage& g, const float* x, const float* y, const float* z, size_t n, float cell_size, int n_rings, bool with_by_id, hipStream_t s) {
        if (n == 0 || n > size_t(kVgMaxBlocks) * kVgTile) { ++declined; return false; }
        h_hdr.reserve(2);
        d_hdr.reserve(1);
        for (int a = 0; a < 3; ++a) { h_hdr.p[0].mn[a] = 0xffffffffu; h_hdr.p[0].mx[a] = 0u; }
        h_hdr.p[0].n_out = 0u; h_hdr.p[0].n_bad = 0u;
        FLS_HIP(hipMemcpyAsync(d_hdr.p, &h_hdr.p[0], sizeof(VgHeader), hipMemcpyHostToDevice, s));
        const int ni = int(n), nb = (ni + kVgBlock - 1) / kVgBlock;
        hipLaunchKernelGGL(vg_minmax, dim3(unsigned(std::min(nb, minmax_blocks(n)))), dim3(kVgBlock), 0, s, x, y, z, ni, d_hdr.p);
        FLS_HIP(hipMemcpyAsync(&h_hdr.p[1], d_hdr.p, sizeof(VgHeader), hipMemcpyDeviceToHost, s));
        FLS_HIP(hipStreamSynchronize(s));
        const VgHeader& hh = h_hdr.p[1];
        if (hh.n_bad != 0u || hh.mn[0] == 0xffffffffu) { ++declined; return false; }
        const float inv = 1.0f / cell_size;
        CgWindow w;
        w.inv_cell = inv;
        int mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {  // floorf(p * inv) is monotone in p: the extreme cells come from the extreme coordinates
            const float lo = std::floor(vg_unord(hh.mn[a]) * inv), hi = std::floor(vg_unord(hh.mx[a]) * inv);
            if (!(std::fabs(lo) < float(kKeyLimit) && std::fabs(hi) < float(kKeyLimit))) { ++declined; return false; }
            mn[a] = int(lo); mx[a] = int(hi);
        }
        size_t nc = 1;
        for (int a = 0; a < 3; ++a) nc *= size_t(mx[a] - mn[a] + 1);
        if (nc > GridImage::kMaxWindowCells) { ++declined; return false; }
        w.ox = mn[0]; w.oy = mn[1]; w.oz = mn[2];
        w.nx = mx[0] - mn[0] + 1; w.ny = mx[1] - mn[1] + 1; w.nz = mx[2] - mn[2] + 1;
        sort.prepare(n);
        g.d_pts.reserve(n);
        g.d_cells.reserve(nc);
        g.d_table.reserve(1);  // (never probed: every search kernel takes the window path when the window exists)
        FLS_HIP(hipMemsetAsync(g.d_cells.p, 0, nc * sizeof(uint2), s));
        hipLaunchKernelGGL(cg_keys, dim3(unsigned(nb)), dim3(kVgBlock), 0, s, x, y, z, ni, w, sort.k0, sort.v0);
        sort.run(DevicePairSort::passes_for((unsigned long long)(nc - 1)), s);
        hipLaunchKernelGGL(cg_fill, dim3(unsigned(nb)), dim3(kVgBlock), 0, s, (const unsigned*)sort.k0, (const unsigned*)sort.v0, ni, x, y, z, g.d_pts.p, g.d_cells.p);
        hipLaunchKernelGGL(cg_count, dim3(unsigned(nb)), dim3(kVgBlock), 0, s, (const unsigned*)sort.k0, ni, g.d_cells.p);
        if (with_by_id) {
            g.d_by_id.reserve(n);
            hipLaunchKernelGGL(cg_by_id, dim3(unsigned(nb)), dim3(kVgBlock), 0, s, x, y, z, ni, g.d_by_id.p);
        }
        FLS_HIP(hipGetLastError());
        g.cell = cell_size;
        g.inv_cell = inv;
        g.rings = n_rings;
        g.table.clear(); g.pts.clear(); g.cells.clear();
        g.mask = 0;
        g.used = n;
        g.have_window = true;
        g.win_o[0] = w.ox; g.win_o[1] = w.oy; g.win_o[2] = w.oz;
        g.win_n[0] = w.nx; g.win_n[1] = w.ny; g.win_n[2] = w.nz;
        g.n_cells = 0;  // (not counted on this path)
        ++builds;
        return true;
    }
};

// The local-map deque of the kd-tree kinds on the device (icp_optimized.h:173-184, loam_full_kdtree.h:70-91,
// loam_point_to_plane_kdtree.h:60-71): the clouds of the deque live back to back in four SoA planes, so "concatenate the deque"
// is a pointer + a length.  push_back appends (one host-to-device copy of the NEW cloud + a de-interleave kernel), pop_front
// advances the head; the planes are re-packed when the tail reaches the end.  Only used by the opt-in device map filter.
struct DeviceCloudRing {
    DevBuf<float> planes;  // x | y | z | i, `cap` floats each
    DevBuf<float4> rows;   // staging of the cloud being appended
    PinnedBuf<float4> h_rows;
    size_t cap = 0, head = 0, tail = 0;
    std::deque<size_t> sizes;
    const float* x() const { return planes.p + head; }
    const float* y() const { return planes.p + cap + head; }
    const float* z() const { return planes.p + 2 * cap + head; }
    const float* in() const { return planes.p + 3 * cap + head; }
    size_t size() const { return tail - head; }
    void clear() { head = tail = 0; sizes.clear(); }
    void pop_front() {
        if (sizes.empty()) return;
        head += sizes.front();
        sizes.pop_front();
        if (sizes.empty()) head = tail = 0;
    }
    void make_room(size_t extra, hipStream_t s) {
        if (tail + extra <= cap) return;
        const size_t live = tail - head;
        size_t ncap = cap;
        while (live + extra > ncap || ncap < 65536) ncap = ncap ? ncap * 2 : 65536;
        if (ncap == cap && head >= live) {  // re-pack in place: source and destination do not overlap
            for (int a = 0; a < 4 && live; ++a)
                FLS_HIP(hipMemcpyAsync(planes.p + size_t(a) * cap, planes.p + size_t(a) * cap + head, live * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else {
            if (ncap == cap) ncap *= 2;
            DevBuf<float> np;
            np.reserve(4 * ncap);
            for (int a = 0; a < 4 && live; ++a)
                FLS_HIP(hipMemcpyAsync(np.p + size_t(a) * ncap, planes.p + size_t(a) * cap + head, live * sizeof(float), hipMemcpyDeviceToDevice, s));
            FLS_HIP(hipStreamSynchronize(s));  // the old planes are freed below
            std::swap(planes.p, np.p);  // (np now owns the old planes and frees them)
            std::swap(planes.cap, np.cap);
            cap = ncap;
        }
        head = 0;
        tail = live;
    }
    void push_back(const std::vector<PtI>& c, hipStream_t s) {
        const size_t n = c.size();
        make_room(n, s);
        if (n) {
            static_assert(sizeof(PtI) == sizeof(float4), "PtI rows are float4 rows");
            h_rows.reserve(n);
            rows.reserve(n);
            std::memcpy(h_rows.p, c.data(), n * sizeof(PtI));
            FLS_HIP(hipMemcpyAsync(rows.p, h_rows.p, n * sizeof(float4), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(soa_append, dim3(unsigned((n + kVgBlock - 1) / kVgBlock)), dim3(kVgBlock), 0, s, (const float4*)rows.p, int(n),
                               planes.p + tail, planes.p + cap + tail, planes.p + 2 * cap + tail, planes.p + 3 * cap + tail);
            FLS_HIP(hipGetLastError());
            FLS_HIP(hipStreamSynchronize(s));  // the pinned staging rows are reused by the next push
        }
        tail += n;
        sizes.push_back(n);
    }
};

// Map maintenance of one kd-tree kind on the device: the cell-grid build and the deque + map-side pcl::VoxelGrid (bit-identical to
// the reference's since round 4: kernels_exactsort.hpp; FLS_DEVICE_VOXELGRID=0 = the host filter).  The host deque stays
// authoritative: whatever the device declines is redone by the host path.
struct KdMapDevice {
    bool grid_on_device = true;  // FLS_DEVICE_GRID_BUILD=0: host std::sort + bucket loop + upload (A/B)
    bool vg_on_device = true;    // FLS_DEVICE_VOXELGRID=0: the host filter
    DeviceGridBuilder builder;
    DeviceVoxelGrid vg;
    PinnedBuf<float> stage;
    DevBuf<float> xyz;
    unsigned long long device_filters = 0, host_filters = 0;
    void init() {
        if (const char* e = std::getenv("FLS_DEVICE_GRID_BUILD")) grid_on_device = std::atoi(e) != 0;
        vg_on_device = device_voxelgrid_mode() != 0;
    }
    // grid over a HOST cloud (the exact host filter's output, or an unfiltered cloud)
    fls_status build_from_host(CellGridImage& grid, const std::vector<PtI>& cloud, float cell, hipStream_t s, int rings = 1, bool by_id = false) {
        const size_t n = cloud.size();
        if (grid_on_device && n != 0) {
            stage.reserve(3 * n);
            xyz.reserve(3 * n);
            for (size_t i = 0; i < n; ++i) { stage.p[i] = cloud[i].x; stage.p[n + i] = cloud[i].y; stage.p[2 * n + i] = cloud[i].z; }
            FLS_HIP(hipMemcpyAsync(xyz.p, stage.p, 3 * n * sizeof(float), hipMemcpyHostToDevice, s));
            if (builder.run(grid, xyz.p, xyz.p + n, xyz.p + 2 * n, n, cell, rings, by_id, s)) return FLS_OK;  // (run() synchronises: the staging is free)
            FLS_HIP(hipStreamSynchronize(s));
        }
        return grid.build(cloud, cell, s, rings, by_id);
    }
    // local map = [VoxelGrid of] the concatenated deque, then the grid: all on the device.  false: declined, nothing changed.
    bool filter_and_build(CellGridImage& grid, const DeviceCloudRing& ring, bool filter, float leaf, float cell, int rings, bool by_id, size_t& n_map,
                          hipStream_t s) {
        if (!vg_on_device || !grid_on_device) return false;
        size_t n = ring.size();
        if (n == 0) return false;
        const float *x = ring.x(), *y = ring.y(), *z = ring.z();
        if (filter) {
            if (!vg.run(x, y, z, ring.in(), n, leaf, s)) return false;
            x = vg.ox(); y = vg.oy(); z = vg.oz();
            n = vg.n_out;
            if (n == 0) return false;
        }
        if (!builder.run(grid, x, y, z, n, cell, rings, by_id, s)) return false;
        n_map = n;
        ++device_filters;
        return true;
    }
};

// the source-scan filter of the kd-tree / NDT matchers (icp_optimized.h:57, incremental_ndt.h:232): on the device (default), the
// host std::sort path with FLS_DEVICE_VOXELGRID=0 and for whatever the device declines.  With the device path the filtered cloud
// stays on the device; the host copy is fetched only when a map update needs it.
struct SourceFilter {
    bool on_device = true;
    DevScan raw;
    DeviceVoxelGrid vg;
    bool resident = false;  // `source` has not been downloaded yet
    std::vector<float> tmp;
    unsigned long long device_runs = 0, host_runs = 0;
    void init() {
        on_device = device_voxelgrid_mode() != 0;
    }
    void filter(const float* s0, size_t n0, int stride, float leaf, hipStream_t s, DevScan& scan, std::vector<PtI>& source) {
        resident = false;
        raw_pending = false;
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
    // fls_scan_upload_raw: the RAW scan stays resident (x | y | z | intensity of raw_n points) and every fls_match_resident runs the source
    // filter itself, like the reference's Match does (icp_optimized.h:57, incremental_ndt.h:231-232) -- what bench.py times for configs[0] / [2]
    bool raw_pending = false;
    bool withdrawn = false;  // between fls_scan_upload_raw and the next Match: no filtered scan is resident
    size_t raw_n = 0;
    float raw_leaf = 0.f;
    std::vector<float> raw_host;  // packed xyzi rows of the raw scan: only what the device declines goes back to the host filter
    void upload_raw_only(const float* s0, size_t n0, int stride, float leaf, hipStream_t s, DevScan& scan, std::vector<PtI>& source) {
        // the filtered scan of the PREVIOUS upload is gone from here on (fitness / correspondences / a map update between this call and
        // the next fls_match_resident must not see the old scan): nothing is resident until refilter() has run
        resident = false;
        scan.n = 0;
        scan.host.clear();
        source.clear();
        withdrawn = true;
        raw_pending = true;
        raw_n = n0;
        raw_leaf = leaf;
        raw_host.resize(4 * n0);
        for (size_t i = 0; i < n0; ++i) {
            const float* q = s0 + i * size_t(stride);
            raw_host[4 * i] = q[0]; raw_host[4 * i + 1] = q[1]; raw_host[4 * i + 2] = q[2]; raw_host[4 * i + 3] = intensity_of(q, stride);
        }
        if (n0) { raw.upload_raw(s0, n0, stride, s, true); FLS_HIP(hipStreamSynchronize(s)); }
    }
    void refilter(hipStream_t s, DevScan& scan, std::vector<PtI>& source) {
        resident = false;
        withdrawn = false;
        if (on_device && raw_n != 0 && vg.run(raw.x.p, raw.y.p, raw.z.p, raw.xyz.p + 3 * raw_n, raw_n, raw_leaf, s)) {
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
        source = voxel_grid_strided(raw_host.data(), raw_n, 4, raw_leaf);
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
