#!/bin/bash
# round 6: (1) the epilogue in two phases (bias + lane exchange of every piece first, while the rows it reads are in flight): stamps of proj / fc2 / dfc2, one-phase against two-phase debug builds;
# bench pairs of the release builds; (2) the CU-hog lines with the hold sized to the backward
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6p; mkdir -p $O
cd $R
for rep in 1 2; do for v in dbg1p dbg; do for s in proj fc2 dfc2; do echo "== $v $s" >> $O/pstamps.txt; DEVIAS_LIB_PATH=$R/tools/exp/libdevias_amd_$v.so timeout 120 python3 tools/gemm_pstamps.py $s 2>&1 | grep -E "epilogue interval|back-to-back" >> $O/pstamps.txt; done; done; done; cat $O/pstamps.txt | cut -c1-200
for rep in 1 2 3; do for v in rel1p ""; do L=$R/devias_amd/libdevias_amd.so; [ -n "$v" ] && L=$R/tools/exp/libdevias_amd_$v.so
  DEVIAS_LIB_PATH=$L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-full-step --no-probes 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib ${v:-two-phase}', round(d['ms_per_step'],3), 'ms')" >> $O/bench_pairs.txt; done; done; cat $O/bench_pairs.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-probes > $O/bench_plain.json 2> /dev/null
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-probes --cu-hog 16 > $O/bench_hog16.json 2> $O/bench_hog16.err
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-full-step --no-probes --cu-hog 16 --reserve-cus 16 > $O/bench_hog16_res16.json 2> $O/bench_hog16_res16.err
for f in bench_plain bench_hog16 bench_hog16_res16; do python3 -c "
import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), 'clips/s', round(d['ms_per_step'],2), 'ms', d['config']['workload'][-120:])"; done
timeout 400 python3 tools/ab_inproc.py hog=16 gemm_reserve_cus=0,16 gemm_dynamic=0,1 > $O/cu_hog_ab.txt 2>&1; grep -v amdgpu.ids $O/cu_hog_ab.txt
