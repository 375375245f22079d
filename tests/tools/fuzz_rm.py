"""Fuzz of the host side of necat_map_reference (rm_host.h: containment, drop, rescue on the read's stretch of the reference) against the REFERENCE's own
oc2rm_worker (oracle/_ref) on random data sets and options.  CPU only (the oracle's seeding and aligner stand in for the device).

    python tests/tools/fuzz_rm.py <seed> <n data sets>
"""
import sys,os,shutil,subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT)
import numpy as np
from necat_amd import synth
from oracle import oracle_api as ora
import tempfile
BLD=tempfile.mkdtemp(prefix="fuzz_rm_")
subprocess.run(["gcc","-O2","-std=gnu99","-c",os.path.join(ROOT,"oracle","necat_oracle.c"),"-o",BLD+"/o.o"],check=True)
subprocess.run(["g++","-O2","-std=c++17","-ffp-contract=off","-o",BLD+"/check_rm",os.path.join(ROOT,"tests","host_core","check_rm.cpp"),BLD+"/o.o","-lm","-lpthread"],check=True)
rng=np.random.default_rng(int(sys.argv[1])); bad=0
for it in range(int(sys.argv[2])):
    seed=int(rng.integers(0,1<<30)); glen=int(rng.choice([20000,80000,200000])); rep=float(rng.choice([0.0,0.4,0.8])); err=float(rng.choice([0.05,0.12,0.2]))
    tmp=BLD+'/w'; shutil.rmtree(tmp,ignore_errors=True); os.makedirs(tmp)
    G=synth.make_genome(glen,seed,rep)
    rs=synth.simulate_reads(coverage=float(rng.choice([3,8])),seed=seed,err=err,genome=G,mean_len=float(rng.choice([2000,8000])),sd_len=1500,min_len=800)
    rs=synth.add_long_indels(rs,float(rng.choice([0.0,0.5,0.9])),seed=seed+1,lo=200,hi=2500)
    wrk=os.path.join(tmp,"vols"); nv=synth.write_volume_dir(wrk,rs,300_000)
    ncut=int(rng.integers(1,6)); cuts=sorted(set([0,glen]+[int(x) for x in rng.integers(1000,glen-1000,ncut)]))
    seqs=[G[cuts[i]:cuts[i+1]] for i in range(len(cuts)-1)]
    if rng.random()<0.5: seqs.append(rng.integers(0,4,5000,dtype=np.uint8))
    ref=os.path.join(tmp,"ref.vol"); synth.write_volume(ref,np.concatenate(seqs),[len(x) for x in seqs],["c%d"%i for i in range(len(seqs))])
    args=("-k %d -z %d -n %d -a %d -b %d -e %s -i 0"%(int(rng.choice([11,12,13])),int(rng.choice([5,10])),int(rng.choice([2,20])),int(rng.choice([400,1500])),int(rng.choice([1000,2000])),rng.choice(["0.5","0.25"]))).split()
    r1=subprocess.run([ora.REF_RM]+args+["-t","1",wrk,ref,tmp+"/ref.m4"],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
    r2=subprocess.run([BLD+"/check_rm"]+args+[wrk,ref,tmp+"/mine.m4"],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
    a=open(tmp+"/ref.m4").read() if r1.returncode==0 else "<ref crash>"; b=open(tmp+"/mine.m4").read() if r2.returncode==0 else "<crash>"
    if a!=b: bad+=1; print("MISMATCH",seed,glen,rep,err,args,len(a.splitlines()),len(b.splitlines()))
    print(it,"ok" if a==b else "BAD",glen,rep,err,args,len(a.splitlines()),r2.stdout.strip(),flush=True)
print("bad",bad)
