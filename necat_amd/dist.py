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


def file_allgather(directory: str, rank: int, nranks: int, timeout_s: float = 120.0, comm_id: str = "") -> Callable[[bytes], List[bytes]]:
    """The same exchange through files of a shared directory (one sequence number per call): for ranks started by hand
    or by the tests without a process group.  Every rank must make the same sequence of calls.  Two communicators alive at the same
    time in ONE directory (the teams of a pairs schedule) must be given different `comm_id`s: the handshake files are keyed by the
    rank inside the communicator."""
    import os
    import time
    seq = [0]
    token = [None]

    def wait_for(p: str, who: int, t0: float) -> None:
        while not os.path.exists(p):
            if time.time() - t0 > timeout_s:
                raise TimeoutError("rank %d never wrote %s" % (who, p))
            time.sleep(0.0005)

    def put(path: str, text: str) -> None:
        with open(path + ".tmp%d" % rank, "w") as f:
            f.write(text)
        os.rename(path + ".tmp%d" % rank, path)

    def get(path: str):
        try:
            return open(path).read()
        except OSError:
            return None

    def run_token() -> str:
        """One name per RUN: a second communicator or a re-run in the same directory must never read the previous run's files
        (they have the right lengths and stale IPC handles / ncclUniqueIds).  Handshake through atomically replaced files: every
        rank r > 0 publishes a nonce (hello_r); rank 0 announces its token together with the nonces it has seen and re-announces
        whenever a hello file changes; a rank accepts only an announcement that carries ITS nonce and acknowledges it; rank 0
        returns once every rank has acknowledged the current token.  Stale files of an earlier run carry other nonces."""
        import json
        nonce = "%d_%d" % (os.getpid(), time.time_ns())
        tag = ("c%s_" % comm_id) if comm_id else ""
        ann = os.path.join(directory, tag + "run_token")
        t0 = time.time()
        if rank != 0:
            put(os.path.join(directory, tag + "hello_%d" % rank), nonce)
            while True:
                try:
                    a = json.loads(get(ann) or "{}")
                    if a.get("nonces", {}).get(str(rank)) == nonce:
                        put(os.path.join(directory, tag + "ack_%s_%d" % (a["token"], rank)), nonce)
                        try:
                            os.remove(os.path.join(directory, tag + "hello_%d" % rank))        # agreed: the next run publishes its own
                        except OSError:
                            pass
                        return a["token"]
                except ValueError:
                    pass
                if time.time() - t0 > timeout_s:
                    raise TimeoutError("rank 0 never announced a run token for rank %d in %s" % (rank, directory))
                time.sleep(0.0005)
        seen = None
        while True:
            nonces = {str(r): get(os.path.join(directory, tag + "hello_%d" % r)) or (seen or {}).get(str(r)) for r in range(1, nranks)}
            if all(nonces.values()):
                if nonces != seen:
                    put(ann, json.dumps({"token": nonce, "nonces": nonces}))
                    seen = nonces
                if all(get(os.path.join(directory, tag + "ack_%s_%d" % (nonce, r))) == nonces[str(r)] for r in range(1, nranks)):
                    for r in range(1, nranks):
                        try:
                            os.remove(os.path.join(directory, tag + "ack_%s_%d" % (nonce, r)))
                        except OSError:
                            pass
                    return nonce
            if time.time() - t0 > timeout_s:
                raise TimeoutError("the other ranks never showed up in %s" % directory)
            time.sleep(0.0005)

    def allgather(b: bytes) -> List[bytes]:
        if token[0] is None:
            token[0] = run_token()
        k = seq[0]
        seq[0] += 1
        mine = os.path.join(directory, "ag_%s_%d_%d" % (token[0], k, rank))
        with open(mine + ".tmp", "wb") as f:
            f.write(b)
        os.rename(mine + ".tmp", mine)
        out = []
        t0 = time.time()
        for r in range(nranks):
            p = os.path.join(directory, "ag_%s_%d_%d" % (token[0], k, r))
            wait_for(p, r, t0)
            with open(p, "rb") as f:
                out.append(f.read())
        if k >= 2:      # everybody has finished exchange k - 1 (it wrote file k): nobody reads my file k - 2 any more
            try:
                os.remove(os.path.join(directory, "ag_%s_%d_%d" % (token[0], k - 2, rank)))
            except OSError:
                pass
        return out
    return allgather
