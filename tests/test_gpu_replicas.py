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
