// evict_conflict_model_test.cpp -- CPU model of the eviction selection of the device-side AddPoints (funny_lidar_slam_amd/csrc/
// kernels_ivox_update.hpp: ivox_evict_select) against the reference's sequential insert loop (src/ivox_map/ivox_map.cpp:122-143, restated
// in oracle/flo_common.h IVoxMap::AddPoints): a creation that brings the voxel count to the capacity evicts the LRU list's back INSIDE
// the loop, so a voxel may be evicted and re-created by a later point of the same batch.  The batch form decides, from per-voxel first
// ranks and the sorted creation ranks alone, which old voxels are evicted, which are skipped (touched before their turn) and which are
// evicted AND re-created (touched after their turn: their first touch joins the creation sequence, one more eviction follows).  This
// test runs both on random small maps / batches and compares the evicted voxels (in order), the final voxel set, every voxel's points
// and the final LRU order.  The walk below mirrors the kernel's structure (chunks of candidates, restart after every re-created voxel).
// Test infrastructure: plain C++, no GPU.
#include <algorithm>
#include <cstdio>
#include <list>
#include <map>
#include <random>
#include <vector>

using Key = int;

struct SeqResult {
    std::vector<Key> evicted;               // in eviction order
    std::map<Key, std::vector<int>> voxels;  // key -> point ids (old points: negative ids; batch points: their rank)
    std::vector<Key> lru;                    // front (most recent) first
};

static SeqResult sequential(const std::vector<Key>& old_lru_front_first, const std::map<Key, std::vector<int>>& old_pts, const std::vector<Key>& batch,
                            size_t capacity) {
    std::list<Key> cache(old_lru_front_first.begin(), old_lru_front_first.end());
    std::map<Key, std::list<Key>::iterator> index;
    for (auto it = cache.begin(); it != cache.end(); ++it) index[*it] = it;
    SeqResult r;
    r.voxels = old_pts;
    for (size_t rank = 0; rank < batch.size(); ++rank) {
        const Key k = batch[rank];
        auto f = index.find(k);
        if (f == index.end()) {
            cache.push_front(k);
            index[k] = cache.begin();
            r.voxels[k] = {int(rank)};
            if (index.size() >= capacity) {
                const Key b = cache.back();
                r.evicted.push_back(b);
                index.erase(b);
                r.voxels.erase(b);
                cache.pop_back();
            }
        } else {
            r.voxels[k].push_back(int(rank));
            cache.splice(cache.begin(), cache, f->second);
            index[k] = cache.begin();
        }
    }
    r.lru.assign(cache.begin(), cache.end());
    return r;
}

struct BatchResult {
    bool refused = false;
    std::vector<Key> evicted;
    std::map<Key, std::vector<int>> voxels;
    std::vector<Key> lru;
    size_t recreated = 0;
};

// k-th (0-based) smallest of crank[0..C) U S[0..nS), both ascending, all values distinct; k < C + nS
static unsigned merged_rank(const std::vector<unsigned>& crank, const std::vector<unsigned>& S, unsigned k) {
    const unsigned C = unsigned(crank.size()), nS = unsigned(S.size());
    auto cr = [&](long i) -> unsigned { return i < long(C) ? crank[size_t(i)] : 0xFFFFFFFFu; };
    unsigned j = 0;
    while (j < nS && long(k) - long(j) >= 0 && S[j] < cr(long(k) - long(j))) ++j;
    unsigned v = 0;
    if (long(k) - long(j) >= 0 && long(k) - long(j) < long(C)) v = crank[k - j];
    if (j > 0) v = std::max(v, S[j - 1]);
    return v;
}

static BatchResult batch_form(const std::vector<Key>& old_lru_front_first, const std::map<Key, std::vector<int>>& old_pts, const std::vector<Key>& batch,
                              size_t capacity, unsigned chunk) {
    BatchResult out;
    const unsigned n0 = unsigned(old_lru_front_first.size());
    std::map<Key, unsigned> first, last;
    std::map<Key, std::vector<int>> pts;
    for (size_t r = 0; r < batch.size(); ++r) {
        if (!first.count(batch[r])) first[batch[r]] = unsigned(r);
        last[batch[r]] = unsigned(r);
        pts[batch[r]].push_back(int(r));
    }
    std::vector<unsigned> crank;
    for (auto& kv : first)
        if (!old_pts.count(kv.first)) crank.push_back(kv.second);
    std::sort(crank.begin(), crank.end());
    const unsigned C = unsigned(crank.size());
    unsigned E = 0;
    if (size_t(n0) + C >= capacity) E = unsigned(size_t(n0) + C - capacity + 1);
    const unsigned base_c = capacity - 1 > n0 ? unsigned(capacity - 1 - n0) : 0u;
    std::vector<Key> cand(old_lru_front_first.rbegin(), old_lru_front_first.rend());  // oldest first
    std::vector<unsigned> S;         // first ranks of the re-created voxels, ascending
    std::vector<Key> S_keys;
    std::vector<Key> evict_list(size_t(E) + cand.size() + 1, -1);
    unsigned found = 0;
    for (unsigned j0 = 0; j0 < cand.size() && E; j0 += chunk) {
        if (found >= E + S.size()) break;
        const unsigned end = std::min<unsigned>(j0 + chunk, unsigned(cand.size()));
        unsigned start = j0;
        for (;;) {
            const unsigned Etot = E + unsigned(S.size());
            // "parallel" evaluation of the positions [start, end) with the state as it is
            unsigned first_conflict = end, conflict_idx = 0;
            std::vector<std::pair<unsigned, Key>> commits;
            unsigned un_before = 0;
            for (unsigned p = start; p < end; ++p) {
                const Key k = cand[p];
                const bool un = !first.count(k);
                const unsigned idx = found + un_before;
                if (idx < Etot) {
                    if (un) commits.push_back({idx, k});
                    else if (!(first[k] < merged_rank(crank, S, base_c + idx))) {
                        if (p < first_conflict) { first_conflict = p; conflict_idx = idx; }
                    }
                }
                if (un) ++un_before;
            }
            // keep what precedes the first conflict
            unsigned un_kept = 0;
            for (unsigned p = start; p < first_conflict; ++p)
                if (!first.count(cand[p])) ++un_kept;
            for (auto& c : commits)
                if (c.first < found + un_kept) evict_list[c.first] = c.second;
            if (first_conflict == end) { found += un_kept; break; }
            const Key v = cand[first_conflict];
            evict_list[conflict_idx] = v;
            S_keys.push_back(v);
            S.insert(std::upper_bound(S.begin(), S.end(), first[v]), first[v]);
            found = conflict_idx + 1;
            start = first_conflict + 1;
        }
    }
    const unsigned Etot = E + unsigned(S.size());
    if (found < Etot) { out.refused = true; return out; }  // the selection would reach voxels this batch touched / created
    out.recreated = S.size();
    out.evicted.assign(evict_list.begin(), evict_list.begin() + Etot);
    // the final state
    out.voxels = old_pts;
    for (Key k : out.evicted) out.voxels.erase(k);
    for (auto& kv : pts) {
        const bool re = std::find(S_keys.begin(), S_keys.end(), kv.first) != S_keys.end();
        auto& dst = out.voxels[kv.first];
        if (re) dst.clear();
        dst.insert(dst.end(), kv.second.begin(), kv.second.end());
    }
    // LRU: touched / created voxels by their last rank (most recent first), then the untouched survivors in their old order
    std::vector<std::pair<unsigned, Key>> t;
    for (auto& kv : last) t.push_back({kv.second, kv.first});
    std::sort(t.rbegin(), t.rend());
    for (auto& e : t) out.lru.push_back(e.second);
    for (Key k : old_lru_front_first)
        if (!first.count(k) && out.voxels.count(k)) out.lru.push_back(k);
    return out;
}

int main() {
    std::mt19937 rng(20240924u);
    size_t cases = 0, compared = 0, refused = 0, with_recreate = 0, total_recreated = 0, mismatches = 0, seq_hits_front = 0;
    for (int it = 0; it < 200000; ++it) {
        const size_t capacity = 3 + rng() % 40;
        const unsigned n0 = unsigned(rng() % capacity);  // alive voxels: 0 .. capacity - 1
        const unsigned universe = unsigned(capacity) + 1 + rng() % 30;
        std::vector<Key> keys(universe);
        for (unsigned i = 0; i < universe; ++i) keys[i] = Key(i);
        std::shuffle(keys.begin(), keys.end(), rng);
        std::vector<Key> old_lru(keys.begin(), keys.begin() + n0);
        std::map<Key, std::vector<int>> old_pts;
        int pid = -1;
        for (Key k : old_lru) { const int c = 1 + int(rng() % 3); for (int q = 0; q < c; ++q) old_pts[k].push_back(pid--); }
        const unsigned A = unsigned(rng() % 60);
        std::vector<Key> batch(A);
        const unsigned mode = rng() % 3;
        for (unsigned r = 0; r < A; ++r) {
            if (mode == 0) batch[r] = keys[rng() % universe];
            else if (mode == 1) batch[r] = (rng() % 3) ? keys[n0 + rng() % (universe - n0)] : keys[rng() % universe];  // mostly new voxels
            else batch[r] = (rng() % 4 == 0 && n0) ? old_lru[n0 - 1 - rng() % std::min<unsigned>(n0, 6)] : keys[rng() % universe];  // touches of the LRU tail
        }
        ++cases;
        const SeqResult s = sequential(old_lru, old_pts, batch, capacity);
        const unsigned chunk = 1 + rng() % 9;
        const BatchResult b = batch_form(old_lru, old_pts, batch, capacity, chunk);
        if (b.refused) { ++refused; continue; }
        ++compared;
        if (b.recreated) { ++with_recreate; total_recreated += b.recreated; }
        const bool same = s.evicted == b.evicted && s.voxels == b.voxels && s.lru == b.lru;
        if (!same) {
            if (++mismatches <= 5) {
                std::printf("MISMATCH case %d: capacity %zu n0 %u A %u chunk %u recreated %zu\n", it, capacity, n0, A, chunk, b.recreated);
                std::printf("  seq evicted:");   for (Key k : s.evicted) std::printf(" %d", k);
                std::printf("\n  batch evicted:"); for (Key k : b.evicted) std::printf(" %d", k);
                std::printf("\n");
            }
        }
        (void)seq_hits_front;
    }
    std::printf("cases %zu compared %zu refused %zu with re-created voxels %zu (re-created voxels %zu) mismatches %zu\n", cases, compared, refused, with_recreate,
                total_recreated, mismatches);
    return mismatches == 0 && with_recreate > 1000 ? 0 : 1;
}
