"""Shared helpers: build the (HIP matcher, oracle matcher) pair for a reference mode string + YAML block."""
from __future__ import annotations

import numpy as np

from funny_lidar_slam_amd import registration as reg
from oracle import oracle as O

POSE_TOL_M = 1e-4    # BASELINE.json north_star: SE(3) pose within 1e-4 m / 1e-4 rad of the CPU path
POSE_TOL_RAD = 1e-4


def oracle_for(mode: str, y: dict, loc: bool = False) -> O.OracleMatcher:
    if mode == "PointToPlane_IVOX":
        return O.OracleMatcher(O.P2PLANE_IVOX, O.Params(max_iterations=y["optimization_iter_num"], point_to_planar_thres=y["point_to_planar_thres"],
                                                        position_converge_thres=y["position_converge_thres"],
                                                        rotation_converge_thres=y["rotation_converge_thres"], is_localization_mode=int(loc)))
    if mode == "IcpOptimized":
        return O.OracleMatcher(O.ICP_OPTIMIZED, O.Params(max_iterations=y["optimization_iter_num"], local_map_size=y["local_map_size"],
                                                         map_cloud_filter_size=y["local_map_cloud_filter_size"],
                                                         source_cloud_filter_size=y["source_cloud_filter_size"], point_search_thres=y["point_search_thres"],
                                                         position_converge_thres=y["position_converge_thres"], rotation_converge_thres=y["rotation_converge_thres"],
                                                         rot_thre_add_cloud=y["keyframe_delta_rotation"], dist_thre_add_cloud=y["keyframe_delta_distance"],
                                                         is_localization_mode=int(loc)))
    if mode == "IncrementalNDT":
        return O.OracleMatcher(O.INCREMENTAL_NDT, O.Params(ndt_voxel_size=y["ndt_voxel_size"], ndt_res_outlier_threshold=y["ndt_outlier_threshold"],
                                                           source_cloud_filter_size=y["source_cloud_filter_size"],
                                                           rotation_converge_thres=y["rotation_converge_thres"], position_converge_thres=y["position_converge_thres"],
                                                           ndt_min_points_in_voxel=y["ndt_min_points_in_voxel"], ndt_max_points_in_voxel=y["ndt_max_points_in_voxel"],
                                                           ndt_min_effective_pts=y["ndt_min_effective_pts"], ndt_capacity=y["ndt_capacity"],
                                                           max_iterations=y["optimization_iter_num"], is_localization_mode=int(loc)))
    if mode == "LoamFull_KdTree":
        return O.OracleMatcher(O.LOAM_FULL, O.Params(point_to_planar_thres=y["point_to_planar_thres"], point_search_thres=y["point_search_thres"],
                                                     line_ratio_thres=y["line_ratio_thres"], position_converge_thres=y["position_converge_thres"],
                                                     rotation_converge_thres=y["rotation_converge_thres"], dist_thre_add_cloud=y["keyframe_delta_distance"],
                                                     rot_thre_add_cloud=y["keyframe_delta_rotation"], local_corner_size=y["local_corner_map_size"],
                                                     local_planar_size=y["local_planar_map_size"], corner_voxel_filter_size=y["local_corner_voxel_filter_size"],
                                                     planar_voxel_filter_size=y["local_planar_voxel_filter_size"], max_iterations=y["optimization_iter_num"]))
    if mode == "PointToPlane_KdTree":
        return O.OracleMatcher(O.P2PLANE_KDTREE, O.Params(point_to_planar_thres=y["point_to_planar_thres"], position_converge_thres=y["position_converge_thres"],
                                                          rotation_converge_thres=y["rotation_converge_thres"], rot_thre_add_cloud=y["keyframe_delta_rotation"],
                                                          dist_thre_add_cloud=y["keyframe_delta_distance"], local_map_size=y["local_map_size"],
                                                          map_cloud_filter_size=y["local_map_cloud_filter_size"], max_iterations=y["optimization_iter_num"],
                                                          is_localization_mode=int(loc)))
    raise ValueError(mode)


def cluster_for(mode: str, scan, corner=None) -> reg.PointcloudCluster:
    if mode in ("IcpOptimized", "IncrementalNDT"):
        return reg.PointcloudCluster(ordered_cloud_=scan)
    if mode == "LoamFull_KdTree":
        return reg.PointcloudCluster(planar_cloud_=scan, corner_cloud_=corner)
    return reg.PointcloudCluster(planar_cloud_=scan)


def pose_close(Ta, Tb):
    from funny_lidar_slam_amd.synth import pose_error
    dt, dr = pose_error(Ta, Tb)
    return dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, dt, dr


def assert_same_registration(m: reg.RegistrationInterface, o: O.OracleMatcher, ok, T, ok_ref, T_ref, slots=(0,), sets_only_tail=True,
                             max_tie_rows=0):
    """HIP result vs oracle result of one Match on identical inputs:
    same return value, same iteration count, n_valid per iteration identical, pose within 1e-4 m / 1e-4 rad
    at EVERY iteration, correspondence ids bit-exact (slot 0 exact, slots 1..K-1 as a set for the iVox path
    whose reference order is libstdc++-introselect-defined; fully ordered for the kd-tree kinds)."""
    assert ok == ok_ref, (ok, ok_ref)
    assert m.stats.iterations == o.stats.iterations, (m.stats.iterations, o.stats.iterations)
    Tg, nvg, srg = m.iteration_log()
    To, nvo, sro = o.iteration_log()
    assert len(nvg) == len(nvo)
    assert np.array_equal(nvg, nvo), (nvg, nvo)
    for i in range(len(nvo)):
        good, dt, dr = pose_close(Tg[i], To[i])
        assert good, (i, dt, dr)
        assert abs(srg[i] - sro[i]) <= 1e-9 * max(1.0, abs(sro[i])), (i, srg[i], sro[i])
    good, dt, dr = pose_close(T, T_ref)
    assert good, (dt, dr)
    assert m.stats.n_valid == o.stats.n_valid
    assert m.stats.n_valid_corner == o.stats.n_valid_corner
    for slot in slots:
        ids, cnt, valid = m.correspondences(slot)
        ids_r, cnt_r, valid_r = o.correspondences(slot)
        assert ids.shape == ids_r.shape
        assert np.array_equal(cnt, cnt_r)
        assert np.array_equal(valid, valid_r)
        if sets_only_tail:
            bad = (ids[:, 0] != ids_r[:, 0]) | (np.sort(ids, 1) != np.sort(ids_r, 1)).any(1)
        else:
            bad = (ids != ids_r).any(1)
        assert int(bad.sum()) <= max_tie_rows, int(bad.sum())
        if bad.any():
            #  the budget is a COUNT; a genuine mismatch could hide under it.  Every differing row must be one the oracle
            # itself marks as decided by an exact distance tie (libstdc++'s introselect permutation: implementation-defined in the reference)
            tie = o.tie_rows() if hasattr(o, "tie_rows") else None
            assert tie is not None, "rows differ and the oracle keeps no tie flags for this kind"
            assert tie.shape[0] == bad.shape[0] and not (bad & ~tie).any(), (np.flatnonzero(bad & ~tie)[:8], int(bad.sum()), int(tie.sum()))
    return dt, dr
