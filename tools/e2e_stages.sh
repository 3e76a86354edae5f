# measurement only (GPU box): the two drop-in scripts on the configs[2] files with SVJG_VERBOSE=1, stage timers on stderr
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python3 - <<'PY'
import os, sys, time, subprocess, tempfile
ROOT=os.getcwd()
for p in (ROOT, os.path.join(ROOT,"svjedi-graph_amd"), os.path.join(ROOT,"tools")): sys.path.insert(0,p)
import synth
base="/dev/shm"
work=tempfile.mkdtemp(prefix="svjg_e2e_", dir=base); pre=os.path.join(work,"c3")
n_aln,n_sv,n_chrom,mix,seed = synth.CONFIGS["c3"]
t=time.time(); synth.generate(pre, n_aln, n_sv, n_chrom, mix, seed, write_gaf=True, threads=16); print("gen", round(time.time()-t,1), flush=True)
env=dict(os.environ, SVJG_VERBOSE="1")
for i in range(2):
    t=time.time()
    p=subprocess.run([sys.executable, f"{ROOT}/svjedi-graph_amd/filter-alignments.py","-a",pre+".gaf","-g",pre+".gfa","-p",pre], env=env, capture_output=True, text=True)
    print("filter", round(time.time()-t,2), p.returncode); print(p.stderr[-1500:])
    t=time.time()
    p=subprocess.run([sys.executable, f"{ROOT}/svjedi-graph_amd/predict-genotype.py","-d",pre+"_informative_aln.json","-v",pre+".vcf","-o",pre+"_g.vcf","--minsupport","3"], env=env, capture_output=True, text=True)
    print("genotype", round(time.time()-t,2), p.returncode); print(p.stderr[-800:])
import shutil; shutil.rmtree(work)
PY
