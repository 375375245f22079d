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


def make_long_indel_partition(tmpdir, genome=24_000, coverage=28.0, seed=5, frac=0.5):
    """A read set in which half of the reads carry long indels (synth.add_long_indels) and its candidate partition (oracle
    oc2pmov -j 0 + single-partition pcan): work dir, candidate prefix, partition bytes.  What oc2cns -r 1 has to rescue."""
    from oracle import oracle_api as ora
    rs = synth.add_long_indels(synth.simulate_reads(genome, coverage, seed=seed, err=0.12, repeat_frac=0.1), frac, seed=seed + 1)
    d = os.path.join(str(tmpdir), "vols_indel")
    nv = synth.write_volume_dir(d, rs, 500_000)
    o = ora.options(**dict(FAST, job=0, binary_output=1, num_threads=4))
    rec = b""
    for v in range(nv):
        out = os.path.join(str(tmpdir), "pm_indel_%d" % v)
        ora.pm_main(o, v, d, out)
        rec += open(out, "rb").read()
    part = pcan_single_partition(rec)
    prefix = os.path.join(str(tmpdir), "cands_indel")
    write_partition(prefix, part)
    return d, prefix, part


def make_rm_dataset(tmpdir, seed=13, repeat_frac=0.6, genome_len=150_000, coverage=6.0):
    """Reads (half of them with long indels) in a volume directory and a reference volume file holding the genome they come from
    as three contigs plus an unrelated sequence: what oc2rm_worker maps.  Returns (wrk_dir, reference_path, number of volumes)."""
    import numpy as np
    G = synth.make_genome(genome_len, seed, repeat_frac)
    rs = synth.add_long_indels(synth.simulate_reads(coverage=coverage, seed=seed, err=0.12, genome=G), 0.5, seed=seed + 1, lo=300, hi=1500)
    wrk = os.path.join(str(tmpdir), "vols_rm")
    nv = synth.write_volume_dir(wrk, rs, 300_000)
    cuts = [0, genome_len * 4 // 15, genome_len * 11 // 15, genome_len]
    seqs = [G[cuts[i]:cuts[i + 1]] for i in range(3)] + [np.random.default_rng(seed + 2).integers(0, 4, 20_000, dtype=np.uint8)]
    ref = os.path.join(str(tmpdir), "ref.vol")
    synth.write_volume(ref, np.concatenate(seqs), [len(x) for x in seqs], ["ctg%d" % i for i in range(4)])
    return wrk, ref, nv


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


# ---- consensus stage -----------------------------------------------------------------------

from necat_amd.capi import pcan_single_partition  # noqa: E402,F401  (the role swap of oc2pcan, one partition)


def reference_candidate_partitions(wrk_dir: str, nv: int, tmp: str, kmer_size: int = 13) -> str:
    """candidates of every volume by the REFERENCE's oc2pmov (-j 0 -u 1), partitioned by the reference's oc2pcan; returns the
    candidates path (needs oracle/_ref)"""
    from oracle import oracle_api as ora
    o = ora.options(**dict(FAST, kmer_size=kmer_size, job=0, binary_output=1, num_threads=2))
    can = os.path.join(tmp, "cands")
    with open(can, "wb") as f:
        for v in range(nv):
            ora.run_ref(o, v, wrk_dir, can + ".v%d" % v)
            f.write(open(can + ".v%d" % v, "rb").read())
    ora.run_ref_pcan(wrk_dir, can)
    return can


def write_partition(prefix: str, records: bytes) -> None:
    with open(prefix + ".p0", "wb") as f:
        f.write(records)
    with open(prefix + ".partitions", "w") as f:
        f.write("1\n")


def cns_log_text(res, cands, tmpl_off, reads_codes, reads_off, fnv, full=False) -> str:
    """the log of oracle/cns_ref_harness.c rebuilt from a necat_cns_result (capi.CnsResult)"""
    from necat_amd import capi
    out = []
    for t in range(res.templates.shape[0]):
        T = res.templates[t]
        if not T["examined"]:
            continue
        for k in range(int(T["ovlp_begin"]), int(T["ovlp_end"])):
            ov = res.overlaps[k]
            c = cands[int(ov["cand"])]
            q = reads_codes[reads_off[c["qid"]]:reads_off[c["qid"] + 1]]
            if c["qdir"]:
                q = (3 - q[::-1]).astype(np.uint8)
            tg = reads_codes[reads_off[c["sid"]]:reads_off[c["sid"] + 1]]
            qa, ta = capi.gapped_strings(res.ops(ov), int(ov["align_size"]), q, int(ov["qoff"]), tg, int(ov["toff"]))
            ln = "A\t%d\t%d\t%.17g\t%d\t%s\t%s" % (ov["toff"], ov["tend"], ov["weight"], ov["align_size"], fnv(qa), fnv(ta))
            if full:
                ln += "\t%s\t%s" % (qa.decode(), ta.decode())
            out.append(ln)
        c0 = cands[int(tmpl_off[t])]
        rg = res.ranges[int(T["range_begin"]):int(T["range_end"])]
        ln = "T\t%d\t%d\t%.17g\t%d\t%d\t%d" % (c0["sid"], c0["ssize"], T["ident_cutoff"], T["num_can"], T["num_ovlps"], rg.shape[0])
        for a, b in rg:
            ln += "\t%d\t%d" % (a, b)
        out.append(ln)
    return "\n".join(out) + ("\n" if out else "")
