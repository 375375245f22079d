"""oc2cns program on the bench partition at several -t: where does the host consensus stop scaling? (profiling tool)"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from necat_amd import build, capi, synth
build.build_cli()
rs = synth.simulate_reads(4_600_000, 40.0, seed=7)
tmp = tempfile.mkdtemp(prefix="cns_scal_")
d = os.path.join(tmp, "vols"); synth.write_volume_dir(d, rs)
ctx = capi.Context(0)
vol = ctx.upload_volume(synth.pack_2bit(rs.codes), rs.nbases, rs.offsets, rs.sizes)
opt0 = capi.default_options(kmer_size=15, scan_window=20, kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3, num_candidates=500, align_size_cutoff=1000, error=0.5, job=0, num_threads=1, use_hdr_as_id=0)
ix = ctx.build_index(vol, 15, 500)
cands = ctx.find_candidates(ix, vol, vol, 0, 0, opt0, True)
ix.free()
part = capi.pcan_single_partition(capi.pack_candidates(cands).tobytes())
vol.free(); ctx.close()
can = os.path.join(d, "c")
open(can + ".p0", "wb").write(part); open(can + ".partitions", "w").write("1\n")
for t in [int(x) for x in sys.argv[1:]] or [16, 64, 128]:
    t0 = time.time()
    r = subprocess.run([build.OC2CNS, "-t", str(t), d, can, os.path.join(tmp, "o1"), os.path.join(tmp, "o2")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, **({"MALLOC_ARENA_MAX": os.environ["ARENAS"]} if "ARENAS" in os.environ else {})))
    m = re.search(r"extension loop ([0-9.]+) s, consensus ([0-9.]+) s", r.stdout)
    print("-t %d: wall %.2f s, %s" % (t, time.time() - t0, m.group(0) if m else r.stderr[-300:]), flush=True)
