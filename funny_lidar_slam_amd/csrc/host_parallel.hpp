// host_parallel.hpp -- a small worker pool for the host side of the product, and an EXACT parallel restatement of
// libstdc++'s std::sort.
//
// Why: pcl::VoxelGrid (include/common/pointcloud_utility.h:216-271 -> PCL voxel_grid.hpp) sums the points of a leaf in the
// order std::sort leaves records with equal leaf index in.  std::sort is unstable, so that order is a property of the
// ALGORITHM (introsort: median-of-three pivot, unguarded Hoare partition, 16-element insertion-sort blocks), not of the data;
// bit-exact parity of the down-sampled clouds needs exactly that permutation.  A different (e.g. radix) sort cannot give it
// -- which is why the device VoxelGrid is opt-in -- but introsort itself parallelises without changing its result: after a
// partition the two sides never interact again, and the final insertion sort never moves a record across a partition
// boundary (everything left of a boundary is <= everything right of it, and insertion only passes STRICTLY greater
// predecessors).  exact_parallel_sort() therefore runs the same partitions on the same ranges as std::sort, the sides as
// independent tasks, and insertion-sorts every final block on its own.  tests/host/host_logic_test.cpp checks the resulting
// permutation against std::sort itself (random keys with heavy ties, all sizes); the heap-sort fallback of introsort
// (recursion deeper than 2 log2 n: adversarial inputs only) is not restated -- the caller gets `false` and runs std::sort.
#pragma once
#if defined(__linux__)
#include <sched.h>
#endif
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace fls {

// Worker pool for short data-parallel phases on the host.  A region belongs to the calling thread, which runs a sequence
// of PHASES; a phase is a set of chunks handed out through one atomic ticket.  Workers are only HELPERS: a worker that wakes
// up late (idle cores take 0.1-0.2 ms, sometimes milliseconds, to come back) joins whatever phase is running or finds the
// region closed -- the caller never waits for a thread that has not claimed a chunk, so the worst case is the sequential
// time.  (A first version with static slices and all-thread barriers stalled on exactly those late wake-ups.)
class HostPool {
public:
    static HostPool& get() {
        static HostPool pool;
        return pool;
    }
    int threads() const { return n_; }

    class Region {
    public:
        // body(chunk) for chunk in [0, n_chunks), on the caller and on every helper that is awake; returns when all are done
        template <class F>
        void phase(const size_t n_chunks, F&& body) {
            phases_.emplace_back();
            Phase& ph = phases_.back();  // stays alive (and at this address) until the region ends: a helper may still hold it
            ph.body = [&body](size_t c) { body(c); };
            ph.n = n_chunks;
            ph.failed = &failed_;
            cur_.store(&ph, std::memory_order_release);
            help(ph);
            while (ph.done.load(std::memory_order_acquire) != n_chunks) spin_pause();
        }

    private:
        friend class HostPool;
        struct Phase {
            std::function<void(size_t)> body;
            size_t n = 0;
            std::atomic<size_t> next{0}, done{0};
            std::atomic<bool>* failed = nullptr;
        };
        static void help(Phase& ph) {
            for (;;) {
                const size_t c = ph.next.fetch_add(1, std::memory_order_acq_rel);
                if (c >= ph.n) return;  // (a helper that arrives after the phase is over only bumps its private counter)
                try { ph.body(c); } catch (...) { if (ph.failed) ph.failed->store(true, std::memory_order_release); }  // never std::terminate a worker
                ph.done.fetch_add(1, std::memory_order_acq_rel);
            }
        }
        void helper_loop() {
            Phase* last = nullptr;
            for (;;) {
                Phase* ph = cur_.load(std::memory_order_acquire);
                if (ph == end_marker()) return;
                if (ph != last && ph != nullptr) { last = ph; help(*ph); }
                else spin_pause();
            }
        }
        static Phase* end_marker() { return reinterpret_cast<Phase*>(uintptr_t(1)); }
        std::deque<Phase> phases_;
        std::atomic<Phase*> cur_{nullptr};
        std::atomic<bool> failed_{false};
    };

    // fn(region) on the calling thread; false when another region is running (the caller then takes its sequential path --
    // e.g. batch lanes filtering at the same time) or the pool has a single thread
    template <class F>
    bool run(F&& fn) {
        if (n_ <= 1) return false;
        std::unique_lock<std::mutex> own(region_mx_, std::try_to_lock);
        if (!own.owns_lock()) return false;
        Region reg;
        {
            std::lock_guard<std::mutex> lk(mx_);
            region_ = &reg;
            ++gen_;
            gen_hint_.store(gen_, std::memory_order_release);
        }
        cv_.notify_all();
        // the region is closed by a guard: if fn throws on the caller (bad_alloc from a resize inside a filter) the workers must not be
        // left looking at the destroyed stack Region
        struct Closer {
            HostPool* pool;
            Region* reg;
            ~Closer() {
                reg->cur_.store(Region::end_marker(), std::memory_order_release);
                {
                    std::lock_guard<std::mutex> lk(pool->mx_);
                    pool->region_ = nullptr;  // a worker that wakes from now on finds no region
                }
                while (pool->inside_.load(std::memory_order_acquire) != 0) spin_pause();
                pool->last_end_ns_.store(now_ns(), std::memory_order_relaxed);
            }
        } closer{this, &reg};
        fn(reg);
        return !reg.failed_.load(std::memory_order_acquire);  // a phase body threw on a helper: the caller redoes the work sequentially
    }
    // the workers are still spinning after a recent region (no wake-up cost): smaller jobs pay off then
    bool hot() const { return n_ > 1 && now_ns() - last_end_ns_.load(std::memory_order_relaxed) < 1000000ll; }
    static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    static void spin_pause() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mx_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }

private:
    HostPool() {
        int n = 8;
        if (const char* e = std::getenv("FLS_HOST_THREADS")) n = std::atoi(e);
        int hw = int(std::thread::hardware_concurrency());
#if defined(__linux__)
        {   // the CPUs this process may actually run on (cgroup cpusets / taskset), not the machine's
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) hw = hw > 0 ? std::min(hw, a) : a; }
        }
#endif
        if (hw > 0) n = std::min(n, std::max(1, hw / 2));
        n_ = std::max(1, std::min(n, 32));
        for (int t = 1; t < n_; ++t) th_.emplace_back([this] { worker(); });
    }
    void worker() {
        unsigned seen = 0;
        for (;;) {
            Region* reg = nullptr;
            {
                std::unique_lock<std::mutex> lk(mx_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                reg = region_;
                if (reg) inside_.fetch_add(1, std::memory_order_acq_rel);
            }
            if (reg) {
                reg->helper_loop();
                inside_.fetch_sub(1, std::memory_order_acq_rel);
                // stay hot for a moment: the next region often follows within a millisecond (the two VoxelGrids of one
                // Match), and a sleeping core takes 0.1 ms or more to come back
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(1500);
                while (gen_hint_.load(std::memory_order_acquire) == seen && std::chrono::steady_clock::now() < until) spin_pause();
            }
        }
    }
    int n_ = 1;
    std::vector<std::thread> th_;
    std::mutex mx_, region_mx_;
    std::condition_variable cv_;
    Region* region_ = nullptr;
    unsigned gen_ = 0;
    bool stop_ = false;
    std::atomic<int> inside_{0};
    std::atomic<unsigned> gen_hint_{0};
    std::atomic<long long> last_end_ns_{0};
};

// ---- std::sort (libstdc++ introsort), restated so that its partitions can run as independent tasks ----
namespace exact_sort_detail {

constexpr long kThreshold = 16;  // _S_threshold

template <class T>
inline void insertion_sort(T* first, T* last) {  // stable, strict '<' -- the arrangement __final_insertion_sort leaves
    if (first == last) return;
    for (T* i = first + 1; i != last; ++i) {
        T v = *i;
        T* j = i;
        while (j != first && v < *(j - 1)) { *j = *(j - 1); --j; }
        *j = v;
    }
}
template <class T>
inline void median_to_first(T* result, T* a, T* b, T* c) {
    if (*a < *b) {
        if (*b < *c) std::iter_swap(result, b);
        else if (*a < *c) std::iter_swap(result, c);
        else std::iter_swap(result, a);
    } else if (*a < *c) std::iter_swap(result, a);
    else if (*b < *c) std::iter_swap(result, c);
    else std::iter_swap(result, b);
}
template <class T>
inline T* unguarded_partition(T* first, T* last, T* pivot) {
    for (;;) {
        while (*first < *pivot) ++first;
        --last;
        while (*pivot < *last) --last;
        if (!(first < last)) return first;
        std::iter_swap(first, last);
        ++first;
    }
}
template <class T>
inline T* partition_pivot(T* first, T* last) {
    T* mid = first + (last - first) / 2;
    median_to_first(first, first + 1, mid, last - 1);
    return unguarded_partition(first + 1, last, first);
}
inline int lg2(unsigned long n) { return 63 - __builtin_clzl(n); }

template <class T>
struct Task { T* first; T* last; int depth; };

}  // namespace exact_sort_detail

// the sequential form (same code path as the parallel one, no pool): used for small inputs and by the unit test
template <class T>
inline bool exact_sort_sequential(T* first, T* last) {
    using namespace exact_sort_detail;
    if (last - first < 2) return true;
    std::vector<Task<T>> stack;
    stack.push_back({first, last, lg2((unsigned long)(last - first)) * 2});
    while (!stack.empty()) {
        Task<T> t = stack.back();
        stack.pop_back();
        while (t.last - t.first > kThreshold) {
            if (t.depth == 0) return false;  // introsort would switch to heap sort here
            --t.depth;
            T* cut = partition_pivot(t.first, t.last);
            stack.push_back({cut, t.last, t.depth});
            t.last = cut;
        }
        insertion_sort(t.first, t.last);
    }
    return true;
}

// pool form: every chunk of a HostPool phase runs exact_sort_worker(shared) -- a worker loop that returns when no task is open,
// so chunks that start after the work is finished return at once.  shared.failed is set if the depth limit was hit: the
// array is then partially permuted -- the caller restores it and runs std::sort.
template <class T>
struct ExactSortShared {
    std::mutex mx;
    std::vector<exact_sort_detail::Task<T>> tasks;
    std::atomic<long> open{0};  // tasks queued or running
    std::atomic<bool> failed{false};
    void reset(T* first, T* last) {
        tasks.clear();
        failed.store(false);
        if (last - first >= 2) {
            tasks.push_back({first, last, exact_sort_detail::lg2((unsigned long)(last - first)) * 2});
            open.store(1);
        } else {
            open.store(0);
        }
    }
};
template <class T>
inline void exact_sort_worker(ExactSortShared<T>& sh) {
    using namespace exact_sort_detail;
    constexpr long kGrain = 2048;  // ranges below this are finished by the thread that owns them
    std::vector<Task<T>> local;
    // An exception in here (bad_alloc from a push_back) must not leave `open` counting a task nobody will finish -- every other
    // participant, the caller included, would spin on it forever: the thrower marks the sort failed, which is an exit
    // condition of its own for every worker; the caller then redoes the work sequentially on the untouched input.
    try {
    for (;;) {
        if (sh.failed.load(std::memory_order_acquire)) return;
        Task<T> t{nullptr, nullptr, 0};
        {
            std::lock_guard<std::mutex> lk(sh.mx);
            if (!sh.tasks.empty()) { t = sh.tasks.back(); sh.tasks.pop_back(); }
        }
        if (t.first == nullptr) {
            if (sh.open.load(std::memory_order_acquire) <= 0) return;
            HostPool::spin_pause();
            continue;
        }
        local.clear();
        local.push_back(t);
        while (!local.empty()) {
            Task<T> u = local.back();
            local.pop_back();
            while (u.last - u.first > kThreshold) {
                if (u.depth == 0 || sh.failed.load(std::memory_order_relaxed)) { sh.failed.store(true); break; }
                --u.depth;
                T* cut = partition_pivot(u.first, u.last);
                const Task<T> right{cut, u.last, u.depth};
                if (right.last - right.first > kGrain) {
                    sh.open.fetch_add(1, std::memory_order_acq_rel);
                    std::lock_guard<std::mutex> lk(sh.mx);
                    sh.tasks.push_back(right);
                } else {
                    local.push_back(right);
                }
                u.last = cut;
            }
            if (!sh.failed.load(std::memory_order_relaxed)) insertion_sort(u.first, u.last);
        }
        sh.open.fetch_sub(1, std::memory_order_acq_rel);
    }
    } catch (...) {
        sh.failed.store(true, std::memory_order_release);
        sh.open.store(0, std::memory_order_release);
    }
}

// std::sort of records whose order is TOTAL (no two compare equal): every correct sort gives the same array, so the pooled
// introsort may simply be finished by std::sort when it gives up.  (Records with ties need the exact permutation: see
// voxel_grid_parallel in host_maps.hpp, which restarts from the untouched input instead.)
template <class T>
inline void pooled_sort_total_order(T* first, T* last) {
    if (last - first >= 65536) {
        HostPool& pool = HostPool::get();
        ExactSortShared<T> sh;
        sh.reset(first, last);
        const bool ran = pool.run([&](HostPool::Region& reg) { reg.phase(size_t(pool.threads()), [&](size_t) { exact_sort_worker(sh); }); });
        if (ran && !sh.failed.load()) return;
    }
    std::sort(first, last);
}

}  // namespace fls
