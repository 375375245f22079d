"""One rank of a single-volume multi-GPU job, started by tests/test_gpu_shard.py (several ranks on ONE device: the HIP IPC
transport) - builds the index sharded, maps its chunks of the reads, rank 0 keeps the gathered records.

    python tests/tools/shard_worker.py <rank> <nranks> <device> <volume dir> <exchange dir> <out prefix> <k> <scan window> [transport]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from necat_amd import capi, dist as ndist   # noqa: E402
from tests import util                      # noqa: E402


def main():
    rank, nranks, device = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    vdir, xdir, prefix, k, z = sys.argv[4], sys.argv[5], sys.argv[6], int(sys.argv[7]), int(sys.argv[8])
    transport = sys.argv[9] if len(sys.argv) > 9 else "auto"
    ctx = capi.Context(device)
    comm = ctx.comm(rank, nranks, ndist.file_allgather(xdir, rank, nranks), transport)
    vol = ctx.load_volume(os.path.join(vdir, "vol0"))
    kw = dict(util.FAST, kmer_size=k, scan_window=z)
    ix = ctx.build_index_sharded(comm, vol, k, kw["kmer_cnt_cutoff"])
    st = ctx.shard_timings()
    info = {"transport": comm.transport(), "index_exchange_bytes": int(st.index_exchange_bytes), "index_exchange_ms": st.index_exchange_ms,
            "index_local_ms": st.index_local_ms, "index_sharded": int(st.index_sharded)}
    if k <= 13:                              # every rank must hold the COMPLETE index
        stats, offs = ix.download()
        np.save(prefix + "_stats_%d.npy" % rank, stats)
        np.save(prefix + "_offs_%d.npy" % rank, offs)
    chunk = int(os.environ.get("SHARD_CHUNK", "16"))
    c, c_local = ctx.find_candidates_sharded(comm, ix, vol, vol, 0, 0, capi.default_options(**dict(kw, job=0)), True, chunk, 0)
    m4, m_local, ncand = ctx.map_pair_sharded(comm, ix, vol, vol, 0, 0, capi.default_options(**dict(kw, job=1)), True, 1, chunk, 0)
    st = ctx.shard_timings()
    info.update(cands_local=c_local, m4_local=m_local, cands_examined=ncand, reads_local=int(st.reads_local), gather_bytes=int(st.gather_bytes))
    np.save(prefix + "_cands_%d.npy" % rank, np.array(c))
    np.save(prefix + "_m4_%d.npy" % rank, np.array(m4))
    json.dump(info, open(prefix + "_info_%d.json" % rank, "w"))
    ix.free(); vol.free(); comm.close(); ctx.close()


if __name__ == "__main__":
    main()
