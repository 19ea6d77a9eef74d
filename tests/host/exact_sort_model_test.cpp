// The closed form behind csrc/kernels_exactsort.hpp, as a CPU model: a level-synchronous restatement of libstdc++'s introsort in which
// every partition is computed from the two "stop lists" (L-stops: keys >= pivot ascending; R-stops: keys <= pivot descending, then the
// pivot slot), K* = #{k : L_k < R_k} disjoint swaps and cut = min(L_{K*+1}, R_{K*}); ranges that exhaust the depth limit (2 log2 n:
// ring-major LiDAR leaf indices do, routinely) go through a restatement of libstdc++'s heap sort (bits/stl_heap.h) -- checked permutation
// for permutation against std::sort on 400 arrays (heavy ties, sorted / reversed / nearly sorted / piecewise-monotone inputs, up to
// 200,000 records; the run reports how many ranges were heap-sorted).  No GPU needed (g++).
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>
struct Rec { uint32_t idx, pt; bool operator<(const Rec& o) const { return idx < o.idx; } };
struct Seg { long first, last; int depth; };
// libstdc++'s heap algorithms restated (bits/stl_heap.h): __adjust_heap / __push_heap / __make_heap / __sort_heap as std::__partial_sort(first, last, last)
// = introsort's fallback at depth 0 uses them
static void push_heap_(Rec* first, long hole, long top, Rec value) {
    long parent = (hole - 1) / 2;
    while (hole > top && first[parent] < value) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
    first[hole] = value;
}
static void adjust_heap_(Rec* first, long hole, long len, Rec value) {
    const long top = hole;
    long second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (first[second] < first[second - 1]) second--;
        first[hole] = first[second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        first[hole] = first[second - 1];
        hole = second - 1;
    }
    push_heap_(first, hole, top, value);
}
static long g_heap = 0;
static void heap_sort_(Rec* first, Rec* last) {
    ++g_heap;
    const long len = last - first;
    if (len >= 2) {
        for (long parent = (len - 2) / 2;; --parent) { Rec v = first[parent]; adjust_heap_(first, parent, len, v); if (parent == 0) break; }
    }
    while (last - first > 1) { --last; Rec v = *last; *last = *first; adjust_heap_(first, 0, last - first, v); }
}

static int lg2(unsigned long n) { return 63 - __builtin_clzl(n); }
// returns false when introsort would heap-sort
static bool model_sort(std::vector<Rec>& a) {
    const long n = (long)a.size();
    if (n < 2) return true;
    std::vector<Seg> cur{{0, n, 2 * lg2((unsigned long)n)}}, nxt;
    std::vector<long> Lp(n), Rl(n);
    std::vector<Seg> fin;
    while (!cur.empty()) {
        nxt.clear();
        for (const Seg& s : cur) {  // (each segment independent: parallel over segments / elements)
            const long first = s.first, last = s.last, m = last - first;
            if (m <= 16) { continue; }
            if (s.depth == 0) { heap_sort_(a.data() + first, a.data() + last); continue; }
            // phase A: median of three to first
            const long A = first + 1, B = first + m / 2, C = last - 1;
            long med;
            if (a[A] < a[B]) { if (a[B] < a[C]) med = B; else if (a[A] < a[C]) med = C; else med = A; }
            else if (a[A] < a[C]) med = A; else if (a[B] < a[C]) med = C; else med = B;
            std::swap(a[first], a[med]);
            const uint32_t p = a[first].idx;
            // phase B/C: L-stops ascending, R-stops (left ranks) -- element-parallel with a prefix scan
            long nL = 0, nR = 0;
            for (long i = first + 1; i < last; ++i) {
                if (a[i].idx >= p) Lp[first + nL++] = i;
                if (a[i].idx <= p) Rl[first + nR++] = i;
            }
            auto Rk = [&](long k) { return k < nR ? Rl[first + nR - 1 - k] : first; };  // k-th R-stop from the right; the pivot itself stops the scan last
            // phase D: K* = #{k : L_k < R_k}; swaps; cut
            long K = 0;
            while (K < nL && Lp[first + K] < Rk(K)) ++K;
            for (long k = 0; k < K; ++k) std::swap(a[Lp[first + k]], a[Rk(k)]);  // all disjoint: parallel
            const long INF = 1L << 60;
            const long cut = std::min(K < nL ? Lp[first + K] : INF, K >= 1 ? Rk(K - 1) : INF);
            nxt.push_back({cut, last, s.depth - 1});
            nxt.push_back({first, cut, s.depth - 1});
        }
        cur.swap(nxt);
    }
    // final insertion sort == stable sort of the whole array by key restricted to blocks that never cross partition boundaries;
    // emulate exactly: plain insertion sort over the whole array
    for (long i = 1; i < n; ++i) { Rec v = a[i]; long j = i; while (j > 0 && v < a[j - 1]) { a[j] = a[j - 1]; --j; } a[j] = v; }
    return true;
}
int main() {
    std::mt19937 rng(7);
    int bad = 0, tested = 0;
    for (int t = 0; t < 400; ++t) {
        const long n = 1 + rng() % (t < 300 ? 3000 : 200000);
        const uint32_t range = 1 + rng() % (t % 3 == 0 ? 8 : t % 3 == 1 ? n / 3 + 1 : 1000000);
        std::vector<Rec> a(n);
        for (long i = 0; i < n; ++i) a[i] = {uint32_t(rng() % range), uint32_t(i)};
        if (t % 7 == 0) std::sort(a.begin(), a.end(), [](const Rec& x, const Rec& y) { return x.idx < y.idx || (x.idx == y.idx && x.pt < y.pt); });  // presorted input
        if (t % 11 == 0) std::reverse(a.begin(), a.end());
        if (t % 5 == 2) { std::sort(a.begin(), a.end(), [](const Rec& x, const Rec& y) { return x.idx < y.idx; }); for (long i = 0; i < n; i += 97) a[i].idx = rng() % range; }  // drives introsort to its depth limit
        if (t % 5 == 3) { for (long i = 0; i < n; ++i) a[i].idx = uint32_t(((i % 1800) * 37 / 100 + (i / 1800) * 5 + (rng() % 3)) % range); }  // ring-major, piecewise monotone
        for (long i = 0; i < n; ++i) a[i].pt = uint32_t(i);
        std::vector<Rec> ref = a, b = a;
        std::sort(ref.begin(), ref.end());
        model_sort(b);
        ++tested;
        for (long i = 0; i < n; ++i) if (ref[i].idx != b[i].idx || ref[i].pt != b[i].pt) { ++bad; std::printf("MISMATCH t=%d n=%ld range=%u at %ld\n", t, n, range, i); break; }
    }
    std::printf("tested %d, mismatches %d, heap-sorted ranges %ld\n", tested, bad, g_heap);
    return bad != 0;
}
