"""fls_replicas_*: the native one-process / N-device form of BASELINE configs[4] (SURVEY.md 8e "one process + N host threads";
include/fls_reg.h).  One GPU is visible here, so the device list names it several times: {0, 0, 0} = the owner + two replica
handles on the same device, each fed by its own host thread inside the library -- the partition, the per-device import and the
threading are the ones an 8-GPU node runs, only the devices coincide.  The table must equal, bit for bit, the serial
fls_match_batch of all jobs on the owner (every job = a fresh reference matcher holding the owner's map)."""
import numpy as np
import pytest

from funny_lidar_slam_amd import _lib, registration as reg, synth

pytestmark = pytest.mark.gpu


def rows(oks, Ts, stats):
    return [(bool(oks[j]), Ts[j].tobytes(), stats[j].iterations, stats[j].n_valid, stats[j].sum_res) for j in range(len(oks))]


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_replica_set_table_equals_the_serial_batch(devices):
    n_jobs, scale = 11, 0.1
    cfg0 = synth.make_config(1, job=0, scale=scale)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([cfg0["map"]])
    scans = [synth.make_config(1, job=j, scale=scale, with_map=False)["scan"] for j in range(n_jobs)]
    clusters = [reg.PointcloudCluster(planar_cloud_=s) for s in scans]
    serial = rows(*m.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=4))
    rs = m.Replicas(devices)
    ms = rs.import_ms()
    assert len(ms) == len(devices) and ms[0] == 0.0 and all(t > 0.0 for t in ms[1:])  # the owner serves the first entry itself
    got = rows(*rs.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=4))
    assert got == serial
    # fewer jobs than devices, and none
    few = rows(*rs.MatchBatch(clusters[:2], [np.eye(4)] * 2, lanes=2))
    assert few == serial[:2]
    assert rs.MatchBatch([], np.zeros((0, 4, 4)), lanes=2)[0] == []
    print(f"devices {devices}: import ms per entry {['%.1f' % t for t in ms]}")
    rs.close(); m.close()


@pytest.mark.parametrize("env", [{"FLS_REPLICAS_VIA_BLOB": "1"}, {"FLS_IVOX_DENSE": "0"}, {"FLS_IVOX_DENSE": "0", "FLS_REPLICAS_VIA_BLOB": "1"}])
def test_replica_set_other_forms(env, monkeypatch):
    """the round-3 export blob + import form of the replication, and the per-voxel hash-table image, device-to-device and via the blob"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n_jobs, scale = 7, 0.1
    cfg0 = synth.make_config(1, job=0, scale=scale)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([cfg0["map"]])
    scans = [synth.make_config(1, job=j, scale=scale, with_map=False)["scan"] for j in range(n_jobs)]
    clusters = [reg.PointcloudCluster(planar_cloud_=s) for s in scans]
    serial = rows(*m.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=2))
    rs = m.Replicas([0, 0, 0])
    assert rows(*rs.MatchBatch(clusters, [np.eye(4)] * n_jobs, lanes=2)) == serial
    rs.close(); m.close()


def test_replica_set_of_a_device_maintained_map():
    """the owner's map is in device mode (several Match + update calls, bricks created by the device): the copy takes the device's
    own counts, and the owner carries on afterwards as if nothing had happened"""
    scale = 0.1
    cfg = synth.make_config(1, job=0, scale=scale)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    ref = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    for h in (m, ref):
        h.AddCloudToLocalMap([cfg["map"]])
    frames = [synth.make_config(1, job=j, scale=scale, with_map=False)["scan"] for j in range(1, 7)]
    probe = [reg.PointcloudCluster(planar_cloud_=s) for s in frames[3:]]
    for k, s in enumerate(frames[:3]):
        for h in (m, ref):
            T = np.eye(4)
            h.Match(reg.PointcloudCluster(planar_cloud_=s), T, update_map=True)
        if k == 1:
            rs = m.Replicas([0, 0])  # replicated in the middle of the run ...
    assert m.map_size(103) >= 2  # (device-side AddPoints batches: the owner is in device mode)
    rs.Refresh()                 # ... and again at its end
    serial = rows(*ref.MatchBatch(probe, [np.eye(4)] * len(probe), lanes=2))
    assert rows(*rs.MatchBatch(probe, [np.eye(4)] * len(probe), lanes=2)) == serial
    # the owner itself was not disturbed: it continues exactly like the handle that was never replicated
    for s in frames[3:5]:
        Ta, Tb = np.eye(4), np.eye(4)
        m.Match(reg.PointcloudCluster(planar_cloud_=s), Ta, update_map=True)
        ref.Match(reg.PointcloudCluster(planar_cloud_=s), Tb, update_map=True)
        assert Ta.tobytes() == Tb.tobytes()
    assert m.map_size(0) == ref.map_size(0) and m.map_size(102) == ref.map_size(102)
    rs.close(); m.close(); ref.close()


def test_replica_set_follows_the_owner_after_refresh():
    """the owner's map grows (mapping mode) -> Refresh() re-replicates -> the replicas answer like the owner again"""
    scale = 0.1
    cfg = synth.make_config(1, job=0, scale=scale)
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    m.AddCloudToLocalMap([cfg["map"]])
    rs = m.Replicas([0, 0])
    T = np.eye(4)
    m.Match(reg.PointcloudCluster(planar_cloud_=cfg["scan"]), T, update_map=True)  # grows the owner's map
    scans = [synth.make_config(1, job=j, scale=scale, with_map=False)["scan"] for j in range(1, 5)]
    clusters = [reg.PointcloudCluster(planar_cloud_=s) for s in scans]
    serial = rows(*m.MatchBatch(clusters, [np.eye(4)] * 4, lanes=2))
    rs.Refresh()
    assert rows(*rs.MatchBatch(clusters, [np.eye(4)] * 4, lanes=2)) == serial
    rs.close(); m.close()


def test_replica_set_rejects_what_it_cannot_serve():
    m = reg.make_matcher("IcpOptimized", reg.YAML_NCLT_ICP)  # no exportable image
    with pytest.raises(_lib.FlsError):
        m.Replicas([0])
    m.close()
    m = reg.make_matcher("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
    with pytest.raises(_lib.FlsError):
        m.Replicas([0, 99])  # no such device
    m.close()
