"""Sharding of independent scan-to-map registration jobs over the GPUs of one node.

BASELINE.json configs[4]: a batch of independent 64x1800 scan-to-map jobs sharded over 8 MI355X.
One registration is single-GPU; independent jobs partition with NO data-path
collective: every rank owns a contiguous block of job ids (deterministic, static), runs them on its own
GPU against a replica of the map, and the results are gathered once at the end (RCCL all_gather when the
process group is nccl, gloo in the CPU tests).  Per-job state is per handle (reference quirk Q12).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def partition(n_jobs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the block of job ids owned by `rank` (block partition, sizes differ by <= 1)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    base, extra = divmod(n_jobs, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def run_block(worker: Callable[[int], np.ndarray], begin: int, end: int) -> np.ndarray:
    """Run jobs [begin, end) with `worker(job_id) -> flat float64 result vector`; rows = jobs."""
    rows = [np.asarray(worker(j), dtype=np.float64).reshape(-1) for j in range(begin, end)]
    return np.stack(rows) if rows else np.zeros((0, 0))


def gather_results(local: np.ndarray, n_jobs: int, width: int, device=None) -> np.ndarray:
    """All ranks end up with the (n_jobs, width) table, rows ordered by job id.

    Uses torch.distributed (nccl == RCCL on ROCm, gloo on CPU).  Blocks are padded to the largest block so
    that one fixed-size all_gather suffices (the payload is tiny: 20 doubles per job)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return np.asarray(local, dtype=np.float64).reshape(n_jobs, width)
    sizes = [partition(n_jobs, world, r) for r in range(world)]
    cap = max(e - b for b, e in sizes)
    buf = torch.zeros((cap, width), dtype=torch.float64, device=device)
    b, e = sizes[rank]
    if e > b:
        buf[: e - b] = torch.from_numpy(np.asarray(local, dtype=np.float64).reshape(e - b, width)).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    table = np.zeros((n_jobs, width))
    for r, (rb, re_) in enumerate(sizes):
        if re_ > rb:
            table[rb:re_] = out[r][: re_ - rb].cpu().numpy()
    return table


def gather_scans(local_scans: Sequence[np.ndarray], n_jobs: int, dst: int = 0, group=None) -> List[np.ndarray]:
    """The scans of ALL jobs on rank `dst`, in job order; [] on the other ranks.  Every rank casts only its own block of the configs[4] scans
    (`partition`); the native one-process form that rank `dst` also measures needs every scan, and receiving 7/8 of them over a host-side
    (gloo) group costs a second or two where casting them again cost 20 s.  Point-to-point sends of one flat float32
    buffer per rank; `group` must be a CPU (gloo) group -- the scans are host arrays that fls_match_batch uploads itself."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return list(local_scans)
    b, e = partition(n_jobs, world, rank)
    if len(local_scans) != e - b:
        raise ValueError("gather_scans: rank %d holds %d scans for the block [%d, %d)" % (rank, len(local_scans), b, e))
    counts = torch.zeros(n_jobs, dtype=torch.int64)
    for k, sc in enumerate(local_scans):
        counts[b + k] = int(np.asarray(sc).shape[0])
    dist.all_reduce(counts, group=group)  # every job's point count, on every rank
    if rank != dst:
        if int(counts[b:e].sum()) > 0:
            flat = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(sc, np.float32).reshape(-1, 3) for sc in local_scans], axis=0)))
            dist.send(flat, dst=dst, group=group)
        return []
    out: List[np.ndarray] = [None] * n_jobs  # type: ignore[list-item]
    out[b:e] = [np.asarray(sc, np.float32).reshape(-1, 3) for sc in local_scans]
    for r in range(world):
        if r == dst:
            continue
        rb, re_ = partition(n_jobs, world, r)
        total = int(counts[rb:re_].sum())
        buf = torch.empty((total, 3), dtype=torch.float32)
        if total > 0:
            dist.recv(buf, src=r, group=group)
        arr, off = buf.numpy(), 0
        for j in range(rb, re_):
            c = int(counts[j])
            out[j] = arr[off:off + c].copy()
            off += c
    return out


def broadcast_blob(blob, src: int = 0, device=None) -> np.ndarray:
    """The map image exported on rank `src` (RegistrationInterface.ExportMap) to every rank: one broadcast of the size, one of
    the bytes (~25 MB for the 1e6-point iVox map, one RCCL broadcast over xGMI; gloo in the CPU tests).
    `blob` is ignored on the other ranks.  Returns the uint8 array on every rank."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(blob, dtype=np.uint8)
    rank = dist.get_rank()
    size = torch.tensor([int(blob.size) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(size, src=src)
    n = int(size.item())
    if rank == src:
        buf = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.uint8)).to(size.device)
    else:
        buf = torch.empty(n, dtype=torch.uint8, device=size.device)
    dist.broadcast(buf, src=src)
    return buf.cpu().numpy()


def broadcast_map_image(m, src: int = 0, device=None) -> dict:
    """The DEVICE image of rank `src`'s map to every rank ("rank 0 builds the device map image, one broadcast over xGMI"):
    fls_map_image_export writes the flat image straight into the tensor the collective works on -- a CUDA tensor with RCCL (the bytes
    never touch the host), pinned host memory with gloo -- and every other rank's handle takes it with fls_map_image_import and becomes
    a read-only replica.  No host mirror, no re-flatten (round 4: ExportMap 90 ms + ImportMap 78 ms per rank for the 1e6-point map).
    `m`: a RegistrationInterface mirror of the iVox kind on every rank.  Returns the timings of this rank."""
    import time

    import torch
    import torch.distributed as dist

    inited = dist.is_initialized()
    rank = dist.get_rank() if inited else 0
    on_dev = device is not None and str(device).startswith("cuda")
    n = torch.tensor([m.MapImageBytes() if rank == src else 0], dtype=torch.int64, device=device)
    if inited:
        dist.broadcast(n, src=src)
    nbytes = int(n.item())
    if nbytes == 0:
        raise RuntimeError("broadcast_map_image: rank %d has no device image to export" % src)
    if on_dev:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8)
        if torch.cuda.is_available():
            buf = buf.pin_memory()  # device <-> host copies at PCIe speed on both sides
    t0 = time.perf_counter()
    if rank == src:
        m.ExportMapImage(buf.data_ptr(), nbytes, on_dev)  # (synchronous: the image is in `buf` when it returns)
    t1 = time.perf_counter()
    if inited:
        dist.broadcast(buf, src=src)
        if on_dev:
            torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rank != src:
        m.ImportMapImage(buf.data_ptr(), nbytes, on_dev)
    t3 = time.perf_counter()
    return {"image_MB": nbytes / 1e6, "export_ms": 1e3 * (t1 - t0), "broadcast_ms": 1e3 * (t2 - t1), "import_ms": 1e3 * (t3 - t2),
            "buffer": "device" if on_dev else "pinned host"}


def pack_result(T: np.ndarray, ok: bool, iterations: int, n_valid: int, sum_res: float) -> np.ndarray:
    """Job result row: 16 pose doubles (row-major 4x4) + [ok, iterations, n_valid, sum_res]."""
    return np.concatenate([np.asarray(T, dtype=np.float64).reshape(16), [float(ok), float(iterations), float(n_valid), float(sum_res)]])


RESULT_WIDTH = 20
