"""GPU parity of the LOAM feature front-end (include/fls_features.h) against the CPU oracle (oracle/flo_features.h):
every array bit-exact (cell owners, ordered cloud, depth, columns, ring bounds, roughness, flags, selections, clouds)."""
import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, features, registration as reg, synth
from oracle import oracle as O
from tests.test_oracle_features import VELO64

pytestmark = pytest.mark.gpu

GPU_ARGS = dict(lidar_horizontal_scan=1800, lidar_vertical_scan=64, lidar_horizontal_resolution=VELO64["horizontal_resolution"],
                min_distance=4.0, max_distance=100.0)
EXACT = ["ordered", "depth", "col", "row_start", "row_end", "raw_index", "roughness", "valid_pre", "valid_post", "is_corner", "corner_idx",
         "planar_idx", "corner", "planar"]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built):
    assert _lib.device_count() >= 1, "gpu tests need an MI355X (gfx950): the HIP path has no CPU fallback"


def both(raw, corner=1.0, planar=0.1, rows=64, cols=1800, h_res=None, **kw):
    h_res = VELO64["horizontal_resolution"] if h_res is None else h_res
    o = O.OracleFeatures(rows, cols, h_res, 4.0, 100.0, corner, planar)
    n_ref = o.Project(raw)
    ran = o.ExtractFeatures()
    g = features.FeatureFrontEnd(cols, rows, h_res, 4.0, 100.0, corner, planar, **kw)
    n = g.project(raw)
    nc, npl = g.extract()
    assert n == n_ref
    for name in EXACT:
        a, b = g.get(name), o.get(name)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), name
    assert nc == len(o.get("corner_idx")) and npl == len(o.get("planar_idx"))
    return g, o, ran


@pytest.mark.parametrize("seed", [9, 10, 11])
def test_velodyne64_frame_bit_exact(seed):
    scene = synth.make_scene()
    T = synth.random_pose(synth.rng_for(3, seed), 20.0, 5.0)
    raw = synth.cast_raw_scan(scene, T, rng=synth.rng_for(3, 0, seed), **synth.VELODYNE_64)
    g, o, ran = both(raw)
    assert ran and len(g.get("corner_idx")) > 100


def test_velodyne16_and_thresholds():
    """16 rings x 1800 columns (lidar_model.cpp:24-30), other thresholds, ragged rings (half of the returns dropped)."""
    scene = synth.make_scene()
    raw = synth.cast_raw_scan(scene, np.eye(4), rng=synth.rng_for(0, 0, 3), drop_frac=0.5, **synth.VELODYNE_16)
    both(raw, corner=0.5, planar=0.05, rows=16)


def test_shuffled_stream_first_return_wins():
    """The projector keeps the FIRST point of the stream per range-image cell: shuffle the stream, the winners change
    with it -- on both sides alike."""
    scene = synth.make_scene()
    raw = synth.cast_raw_scan(scene, np.eye(4), rng=synth.rng_for(3, 0, 21), dup_frac=0.2, **synth.VELODYNE_64)
    rng = np.random.default_rng(3)
    both(raw[rng.permutation(raw.shape[0])])


def test_degenerate_frames():
    raw = np.zeros(5, dtype=synth.RAW_POINT_DTYPE)
    raw["x"] = 10.0
    g, o, ran = both(raw)
    assert not ran and len(g.get("corner")) == 0 and len(g.get("planar")) == 0
    both(raw[:0])
    # one sparse ring only: most rings empty, row_end < row_start there (the reference's signed block arithmetic)
    scene = synth.make_scene()
    full = synth.cast_raw_scan(scene, np.eye(4), rng=synth.rng_for(3, 0, 5), **synth.VELODYNE_64)
    both(full[full["ring"] == 20])
    both(full[(full["ring"] == 20) | (full["ring"] == 63)][::7])


def test_reference_class_mirror_and_voxel_filters():
    """PointcloudProjector::Project + FeatureExtractor::ExtractFeatures through the reference-shaped classes, then the two
    VoxelGrid filters of preprocessing.cpp:234-237, and the result feeds LoamFull::Match."""
    scene = synth.make_scene()
    raw = synth.cast_raw_scan(scene, np.eye(4), rng=synth.rng_for(3, 0, 9), **synth.VELODYNE_64)
    y = reg.YAML_NCLT_LOAM_FULL
    proj = features.PointcloudProjector(1800, 64, VELO64["horizontal_resolution"], 4.0, 100.0, corner_thres=1.0, planar_thres=0.1,
                                        corner_voxel_filter_size=0.2, planar_voxel_filter_size=0.4)
    ext = features.FeatureExtractor(1.0, 0.1, 1800, 64)
    cl = reg.PointcloudCluster(raw_cloud_=raw)
    proj.Project(cl)
    ext.ExtractFeatures(cl)
    o = O.OracleFeatures(**VELO64)
    o.Project(raw)
    o.ExtractFeatures()
    assert np.array_equal(cl.corner_cloud_, o.get("corner")) and np.array_equal(cl.planar_cloud_, o.get("planar"))
    assert np.array_equal(cl.point_depth_vec_, o.get("depth")) and np.array_equal(cl.row_end_index_vec_, o.get("row_end"))
    for name, leaf, src in (("corner_filtered", 0.2, "corner"), ("planar_filtered", 0.4, "planar")):
        ref = np.zeros((len(o.get(src)), 4), np.float32)
        m = O.lib().flo_voxel_grid(np.ascontiguousarray(o.get(src)).ctypes.data_as(O.C.POINTER(O.C.c_float)), len(ref), 4, leaf,
                                   ref.ctypes.data_as(O.C.POINTER(O.C.c_float)))
        assert np.array_equal(proj.front.get(name), ref[:m]), name
    with pytest.raises(_lib.FlsError):
        features.FeatureExtractor(1.0, 0.1, 1800, 64).ExtractFeatures(reg.PointcloudCluster(raw_cloud_=raw))
    with pytest.raises(_lib.FlsError):
        features.FeatureFrontEnd(1800, 64, VELO64["horizontal_resolution"], 4.0, 100.0, float(np.finfo(np.float32).max), 0.1)  # CHECK_NE(corner_threshold_, FloatNaN)


def test_front_end_feeds_loam_full_match():
    """Raw driver cloud -> GPU feature front-end -> GPU LoamFull::Match, against oracle front-end -> oracle LoamFull:
    the two pipelines see bit-identical feature clouds, hence the usual registration parity."""
    from tests import util
    scene = synth.make_scene()
    cfg = synth.make_config(3, scale=0.25)
    T_gt = synth.random_pose(synth.rng_for(3, 31), 1.5, 0.25)
    raw = synth.cast_raw_scan(scene, T_gt, rng=synth.rng_for(3, 0, 31), max_range=cfg["radius"], **synth.VELODYNE_64)
    g = features.FeatureFrontEnd(1800, 64, VELO64["horizontal_resolution"], 4.0, 100.0, 1.0, 0.1)
    g.project(raw)
    g.extract()
    o = O.OracleFeatures(**VELO64)
    o.Project(raw)
    o.ExtractFeatures()
    corner, planar = g.get("corner")[:, :3].copy(), g.get("planar")[::2, :3].copy()
    assert np.array_equal(corner, o.get("corner")[:, :3]) and np.array_equal(planar, o.get("planar")[::2, :3])
    y = reg.YAML_NCLT_LOAM_FULL
    m = reg.make_matcher("LoamFull_KdTree", y)
    r = util.oracle_for("LoamFull_KdTree", y, False)
    m.AddCloudToLocalMap([cfg["map"], cfg["corner_map"]])
    r.AddCloudToLocalMap(cfg["map"], cfg["corner_map"])
    T = np.eye(4)
    ok = m.Match(reg.PointcloudCluster(planar_cloud_=planar, corner_cloud_=corner), T, update_map=False)
    ok_ref, T_ref = r.Match(planar, np.eye(4), src1=corner, update_map=False)
    util.assert_same_registration(m, r, ok, T, ok_ref, T_ref, slots=(0, 1), max_tie_rows=int(r.counters().tie_queries))
    dt, dr = synth.pose_error(T, T_gt)
    assert ok and dt < 0.1 and dr < 0.01  # the extracted features really register the frame
