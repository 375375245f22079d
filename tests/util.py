"""Shared helpers of the test-suite (datasets, record normalisation)."""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from necat_amd import synth  # noqa: E402

FAST = dict(kmer_size=13, scan_window=20, kmer_cnt_cutoff=500, block_size=2000, block_score_cutoff=3,
            num_candidates=500, align_size_cutoff=1000, ddfs_cutoff=0.25, error=0.5, num_output=500,
            num_threads=4, use_hdr_as_id=0)
SENSITIVE = dict(FAST, scan_window=10)


def make_dataset(tmpdir, genome=200_000, coverage=20.0, seed=3, err=0.12, vol_size=synth.DEFAULT_VOL_SIZE,
                 repeat_frac=0.0, **kw):
    rs = synth.simulate_reads(genome, coverage, seed=seed, err=err, repeat_frac=repeat_frac, **kw)
    d = os.path.join(str(tmpdir), "vols")
    nv = synth.write_volume_dir(d, rs, vol_size)
    return d, rs, nv


def install_golden_volumes(name, tmpdir):
    """Copy tests/golden/<name> (vol files) to tmpdir and write directory files with absolute paths."""
    src = os.path.join(GOLDEN, name)
    dst = os.path.join(str(tmpdir), name)
    os.makedirs(dst, exist_ok=True)
    lines = open(os.path.join(src, "volume_names.txt")).read().splitlines()
    out = []
    for ln in lines:
        p, a, b = ln.split()
        shutil.copy(os.path.join(src, os.path.basename(p)), os.path.join(dst, os.path.basename(p)))
        out.append("%s\t%s\t%s\n" % (os.path.join(dst, os.path.basename(p)), a, b))
    open(os.path.join(dst, "volume_names.txt"), "w").writelines(out)
    shutil.copy(os.path.join(src, "reads_info.txt"), os.path.join(dst, "reads_info.txt"))
    return dst


def m4_key_rows(m):
    """M4 records as sortable tuples; ident_perc compared exactly (it is a deterministic quotient)."""
    return sorted(zip(m["qid"].tolist(), m["sid"].tolist(), m["qdir"].tolist(), m["qoff"].tolist(), m["qend"].tolist(),
                      m["qext"].tolist(), m["qsize"].tolist(), m["sdir"].tolist(), m["soff"].tolist(), m["send"].tolist(),
                      m["sext"].tolist(), m["ssize"].tolist(), m["vscore"].tolist(), m["ident_perc"].tolist()))


def parse_m4_text(path):
    rows = []
    for ln in open(path):
        f = ln.split()
        rows.append((int(f[0]), int(f[1]), f[2], int(f[3]), int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[8]),
                     int(f[9]), int(f[10]), int(f[11])))
    return sorted(rows)


def m4_text_rows(m):
    """The 12 printed columns of DUMP_ASM_M4 as tuples (ident as the %.2f string)."""
    return sorted((int(r["qid"]), int(r["sid"]), "%.2f" % r["ident_perc"], int(r["vscore"]), int(r["qdir"]), int(r["qoff"]),
                   int(r["qend"]), int(r["qsize"]), int(r["sdir"]), int(r["soff"]), int(r["send"]), int(r["ssize"])) for r in m)
