// oracle/ref_shim/include/pstl_omp.hpp -- TEST INFRASTRUCTURE ONLY (forced into the libref_par.so build with -include).
// The reference links TBB for its parallel-STL loops (CMakeLists.txt:96, `std::for_each(std::execution::par / par_unseq, ...)` in
// icp_optimized.h:78, incremental_ndt.h:216,252, loam_point_to_plane_ivox.h:262, loam_full_kdtree.h:216,280, loam_point_to_plane_kdtree.h:209,
// pointcloud_utility.h:153-209).  TBB's headers are absent from this image, and libstdc++ then runs those loops on ONE thread.  Here a
// for_each under `par` / `par_unseq` over random-access iterators becomes an OpenMP loop: the same element function over disjoint index
// ranges on worker threads, which is what TBB's parallel_for does with it.  Every one of those loops writes per-index outputs only (the
// reductions that follow them are sequential in the reference), so the results are the serial build's, bit for bit
// (tests/test_ref_pin.py::test_compiled_reference_parallel_pstl_equals_serial).  `unseq` / `seq` loops and std::sort(par) are left alone.
// (Overloads in namespace std: tolerated for a test shim -- they are more specialised than libstdc++'s policy templates, so the reference's
// calls pick them without any edit to the reference's sources.)
#pragma once
#include <algorithm>
#include <execution>
#include <iterator>
#include <type_traits>
#include <omp.h>

namespace std {
namespace ref_shim_detail {
template <class It, class F>
inline void omp_for_each(It first, It last, F& f) {
    if constexpr (std::is_base_of<std::random_access_iterator_tag, typename std::iterator_traits<It>::iterator_category>::value) {
        const long n = static_cast<long>(last - first);
#pragma omp parallel for schedule(dynamic, 256)
        for (long i = 0; i < n; ++i) f(first[i]);
    } else {
        // (a std::set walk, incremental_ndt.h:216: libstdc++'s PSTL also runs non-random-access ranges serially under `par`)
        for (; first != last; ++first) f(*first);
    }
}
}  // namespace ref_shim_detail
template <class It, class F>
inline void for_each(const __pstl::execution::parallel_policy&, It first, It last, F f) { ref_shim_detail::omp_for_each(first, last, f); }
template <class It, class F>
inline void for_each(const __pstl::execution::parallel_unsequenced_policy&, It first, It last, F f) { ref_shim_detail::omp_for_each(first, last, f); }
}  // namespace std
