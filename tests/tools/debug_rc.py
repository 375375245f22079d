"""debug: records of the default path vs knob settings, determinism, first differences"""
import os, sys, tempfile, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import util
from necat_amd import capi

tmp = pathlib.Path(tempfile.mkdtemp())
d, rs, nv = util.make_dataset(tmp, genome=150_000, coverage=18.0, seed=5)
o1 = capi.default_options(**dict(util.FAST, job=1))
def run(env):
    for k, v in env.items(): os.environ[k] = v
    c = capi.Context(0)
    _, m = capi.pm_main(c, o1, 0, d)
    c.close()
    for k in env: del os.environ[k]
    return set(util.m4_key_rows(m))
base = run({"NECAT_RCWALK": "0"})
K = {"NECAT_RCWALK": "1", "NECAT_TAIL_FUSED": "0", "NECAT_WALK_WAVE": "0"}
for name, env in [("ragged0", dict(K, NECAT_RC_RAGGED="0")), ("ragged0 again", dict(K, NECAT_RC_RAGGED="0")), ("ragged1", dict(K, NECAT_RC_RAGGED="1")),
                  ("ragged1 again", dict(K, NECAT_RC_RAGGED="1")), ("ragged1 pool1", dict(K, NECAT_RC_RAGGED="1", NECAT_RC_POOL_MB="1")),
                  ("ckg_all", dict(K, NECAT_RC_RAGGED="1", NECAT_RC_CKG_ALL="1"))]:
    got = run(env)
    print(name, len(base), len(got), "differences:", len(base - got))
