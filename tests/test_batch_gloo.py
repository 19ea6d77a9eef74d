"""Multi-process (world_size 2, gloo, CPU) test of the job sharding used for BASELINE configs[4]:
partitioning is a disjoint cover, each rank runs only its block, the gathered table equals the serial run.
The per-job worker here is the CPU oracle (test infrastructure) -- on a GPU box bench.py plugs the HIP matcher
into the same partition/gather code."""
import os
import socket
import subprocess
import sys

import numpy as np

from funny_lidar_slam_amd import batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FLS_ROOT"])
import numpy as np
import torch.distributed as dist
from funny_lidar_slam_amd import batch, registration as reg, synth
from tests import util

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n_jobs = 5
cfgs = {}
def worker(job):
    cfg = synth.make_config(1, job=job, scale=0.02)   # same map (salted seed), different scan / T_gt per job
    o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)  # fresh handle per job (Q12)
    o.AddCloudToLocalMap(cfg["map"])
    ok, T = o.Match(cfg["scan"], cfg["T_init"], update_map=False)
    return batch.pack_result(T, ok, o.stats.iterations, o.stats.n_valid, o.stats.sum_res)
# map image broadcast (rank 0 "exports", everybody receives the same bytes): the payload here is a stand-in blob
blob = np.frombuffer(bytes(range(256)) * 4099, dtype=np.uint8).copy() if rank == 0 else None
got = batch.broadcast_blob(blob, src=0)
assert got.dtype == np.uint8 and got.size == 256 * 4099 and int(got[1000:1256].astype(np.int64).sum()) == sum(range(256)), got.size
b, e = batch.partition(n_jobs, world, rank)
# every rank casts its own block of scans, rank 0 collects all of them in job order (bench.py: the native one-process leg of configs[4])
mk = lambda j: (np.arange(3 * (100 + 7 * j), dtype=np.float32).reshape(-1, 3) + np.float32(j))   # ragged on purpose
allsc = batch.gather_scans([mk(j) for j in range(b, e)], n_jobs, dst=0)
if rank == 0:
    assert len(allsc) == n_jobs and all(np.array_equal(allsc[j], mk(j)) for j in range(n_jobs))
else:
    assert allsc == []
local = batch.run_block(worker, b, e)
table = batch.gather_results(local, n_jobs, batch.RESULT_WIDTH)
if rank == 0:
    np.save(os.environ["FLS_OUT"], table)
dist.barrier()
dist.destroy_process_group()
'''


def test_partition_is_a_disjoint_cover():
    for n in (0, 1, 5, 512, 513):
        for w in (1, 2, 3, 8):
            blocks = [batch.partition(n, w, r) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert batch.partition(512, 8, 3) == (192, 256)  # BASELINE configs[4]: 64 jobs per GPU


def test_two_rank_gloo_matches_serial(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "table.npy"
    env = dict(os.environ, FLS_ROOT=ROOT, FLS_OUT=str(out), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    table = np.load(out)
    assert table.shape == (5, batch.RESULT_WIDTH)
    # serial reference in this process
    from funny_lidar_slam_amd import registration as reg, synth
    from tests import util
    for job in range(5):
        cfg = synth.make_config(1, job=job, scale=0.02)
        o = util.oracle_for("PointToPlane_IVOX", reg.YAML_NCLT_IVOX)
        o.AddCloudToLocalMap(cfg["map"])
        ok, T = o.Match(cfg["scan"], cfg["T_init"], update_map=False)
        assert np.array_equal(table[job, :16].reshape(4, 4), T)
        assert table[job, 16] == float(ok) and table[job, 17] == o.stats.iterations and table[job, 18] == o.stats.n_valid
    assert len({tuple(row[:16]) for row in table}) == 5  # five different jobs
