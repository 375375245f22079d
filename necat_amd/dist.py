"""Launcher glue of a single-volume multi-GPU job: the host-side all-gather the C library asks for
(necat_host_allgather_fn), backed by torch.distributed - plumbing only, the data path (index slices, records) moves
inside libnecat_hip.so over RCCL / HIP IPC."""
from __future__ import annotations

from typing import Callable, List


def torch_allgather(dist, group=None, device=None) -> Callable[[bytes], List[bytes]]:
    """bytes -> list of every rank's bytes, over an initialised torch.distributed process group.  `device` = None
    gathers CPU tensors (gloo); a CUDA device gathers through RCCL (process groups created with backend="nccl")."""
    import torch
    world = dist.get_world_size(group)

    def allgather(b: bytes) -> List[bytes]:
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8)
        if device is not None:
            t = t.to(device)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        return [bytes(x.cpu().numpy().tobytes()) for x in out]
    return allgather


def file_allgather(directory: str, rank: int, nranks: int, timeout_s: float = 120.0) -> Callable[[bytes], List[bytes]]:
    """The same exchange through files of a shared directory (one sequence number per call): for ranks started by hand
    or by the tests without a process group.  Every rank must make the same sequence of calls."""
    import os
    import time
    seq = [0]

    def allgather(b: bytes) -> List[bytes]:
        k = seq[0]
        seq[0] += 1
        mine = os.path.join(directory, "ag_%d_%d" % (k, rank))
        with open(mine + ".tmp", "wb") as f:
            f.write(b)
        os.rename(mine + ".tmp", mine)
        out = []
        t0 = time.time()
        for r in range(nranks):
            p = os.path.join(directory, "ag_%d_%d" % (k, r))
            while not os.path.exists(p):
                if time.time() - t0 > timeout_s:
                    raise TimeoutError("rank %d never wrote %s" % (r, p))
                time.sleep(0.0005)
            with open(p, "rb") as f:
                out.append(f.read())
        return out
    return allgather
