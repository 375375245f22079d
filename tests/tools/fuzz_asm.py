"""Fuzz of asm_core.h (the restated oc2asmpm) against the REFERENCE's own oc2asmpm (oracle/_ref) on random data sets: genome size, coverage, error rate,
repeat content, read lengths, long indels, k / window / candidate count.  CPU only (the oracle's table and aligner stand in for the device).

    python tests/tools/fuzz_asm.py <seed> <n data sets>
"""
import sys,os,shutil,subprocess,time
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT)
import numpy as np
from necat_amd import synth
from oracle import oracle_api as ora
REF=os.path.join(os.path.dirname(ora.REF_PMOV),"oc2asmpm")
import tempfile
BLD=tempfile.mkdtemp(prefix="fuzz_asm_")
subprocess.run(["gcc","-O2","-std=gnu99","-c",os.path.join(ROOT,"oracle","necat_oracle.c"),"-o",BLD+"/o.o"],check=True)
subprocess.run(["g++","-O2","-std=c++17","-ffp-contract=off","-o",BLD+"/check_asmpm",os.path.join(ROOT,"tests","host_core","check_asmpm.cpp"),BLD+"/o.o","-lm","-lpthread"],check=True)
rng=np.random.default_rng(int(sys.argv[1]))
bad=0
for it in range(int(sys.argv[2])):
    seed=int(rng.integers(0,1<<30))
    G=int(rng.choice([3000,8000,30000,60000])); cov=float(rng.choice([6,15,40])); err=float(rng.choice([0.005,0.03,0.08,0.15]))
    rep=float(rng.choice([0.0,0.3,0.8])); ml=float(rng.choice([1500,4000,9000]))
    tmp=BLD+'/w'; shutil.rmtree(tmp,ignore_errors=True); os.makedirs(tmp)
    rs=synth.simulate_reads(G, cov, seed=seed, err=err, repeat_frac=rep, mean_len=ml, sd_len=ml/3, min_len=int(rng.choice([500,1500])))
    if rng.random()<0.4: rs=synth.add_long_indels(rs,0.4,seed=seed+1,lo=100,hi=700)
    wrk=os.path.join(tmp,"vols"); nv=synth.write_volume_dir(wrk, rs, int(rng.choice([100_000,400_000])))
    k=int(rng.choice([11,12,13])); z=int(rng.choice([5,10,20])); n=int(rng.choice([3,20,100]))
    args=("-k %d -z %d -n %d -u 0"%(k,z,n)).split()
    for v in range(min(nv,3)):
        r1=subprocess.run([REF]+args+["-t","1",wrk,str(v),tmp+"/ref.m4"],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
        r2=subprocess.run([BLD+"/check_asmpm"]+args+[wrk,str(v),tmp+"/mine.m4"],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True,env=dict(os.environ,CHECK_ASM_BATCH=str(it&1)))
        if r1.returncode!=0: print("REF FAILED",seed,G,cov,err,rep,args,v,r1.stdout[-200:]); continue
        a=open(tmp+"/ref.m4","rb").read(); b=open(tmp+"/mine.m4","rb").read() if r2.returncode==0 else b"<crash>"
        if a!=b: bad+=1; print("MISMATCH",seed,G,cov,err,rep,ml,args,v,len(a),len(b),r2.returncode)
    print(it,"ok" if not bad else "bad",G,cov,err,rep,ml,args,nv,flush=True)
print("bad",bad)
