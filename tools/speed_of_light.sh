#!/bin/bash
# measurement only (GPU box): the speed-of-light trio of k_classify_main, three times per workload -> gpurun_out/<out>/speed_of_light.txt
#   (a) memory only: tools/ubench/skeleton (the kernel's global memory operations in its geometry, no byte work)
#   (b) instructions only: the shipped instruction stream with the memory side made cheap — every read starts in the first 1/4096 of
#       the genome (tools/synth.py SVJG_SYNTH_HOT: the displacements and records touched are a few KB, served by the L1 / L2) and the
#       count updates are left out (-DSVJG_ABLATE build, SVJG_DIAG=8) — plus the arithmetic floor SQ_INSTS_VALU x 4 cycles / 1024 SIMDs
#   (c) the shipped kernel, and the same build with only the count updates left out
# build first (CPU): tools/mkvariant.sh ablate -DSVJG_ABLATE ; hipcc --offload-arch=gfx950 -O3 -o tools/ubench/skeleton tools/ubench/skeleton.hip
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-sol}; mkdir -p $O; shift || true
WL="${*:-c3 c4shard}"
cd $R
T=$O/speed_of_light.txt
ms() { python3 -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.4f' % r['kernel_ms']['classify_main'])"; }
B="python3 bench.py --no-cpu-baseline --no-e2e --no-north-star --steps 12 --warmup 3"
echo "speed-of-light trio of k_classify_main (ms per launch, HIP events; one pass at a time: SVJG_BENCH_SYNC=1), $(date -u +%F)" > $T
for W in $WL; do
  echo "== $W" >> $T
  python3 tools/sol_hits.py $W $O/hits_$W.u32 >> $T 2>> $O/err.txt        # the real stream of count updates (C oracle), for the skeleton
  for i in 1 2 3; do echo "(a) memory only, run $i:" >> $T; timeout -k 10 300 tools/ubench/skeleton $W 10 $O/hits_$W.u32 >> $T 2>&1; done
  echo "(a') memory only, uniform random counters instead of the real stream of count updates:" >> $T; timeout -k 10 300 tools/ubench/skeleton $W 10 >> $T 2>&1
  rm -f $O/hits_$W.u32
  [ -n "$ONLY_A" ] && continue
  for i in 1 2 3; do echo "(b) instructions only (hot 1/4096 of the genome, no count updates), run $i: $(SVJG_HIP_LIB=$R/build/lib_ablate.so SVJG_DIAG=8 SVJG_SYNTH_HOT=4096 SVJG_BENCH_SYNC=1 timeout -k 10 300 $B --workload $W 2>> $O/err.txt | ms) ms" >> $T; done
  echo "(b') hot 1/4096 of the genome WITH count updates into the few counters of that stretch (serialised: not a floor): $(SVJG_HIP_LIB=$R/build/lib_ablate.so SVJG_DIAG=0 SVJG_SYNTH_HOT=4096 SVJG_BENCH_SYNC=1 timeout -k 10 300 $B --workload $W 2>> $O/err.txt | ms) ms" >> $T
  for i in 1 2 3; do echo "(c) shipped, run $i: $(SVJG_BENCH_SYNC=1 timeout -k 10 300 $B --workload $W 2>> $O/err.txt | ms) ms" >> $T; done
  echo "(c') shipped stream without the count updates (ablation build, SVJG_DIAG=8): $(SVJG_HIP_LIB=$R/build/lib_ablate.so SVJG_DIAG=8 SVJG_BENCH_SYNC=1 timeout -k 10 300 $B --workload $W 2>> $O/err.txt | ms) ms" >> $T
  echo "(c'') ablation build as shipped (SVJG_DIAG=0): $(SVJG_HIP_LIB=$R/build/lib_ablate.so SVJG_DIAG=0 SVJG_BENCH_SYNC=1 timeout -k 10 300 $B --workload $W 2>> $O/err.txt | ms) ms" >> $T
done
cat $T
