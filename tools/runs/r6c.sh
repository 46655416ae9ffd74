#!/bin/bash
# round 6, call 3: dgrad GEMMs on transposed weight copies (option gemm_wt), non-temporal stores of the saved pre-activation (gemm_aux_nt): tests + in-process A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_regions_gpu.py tests/test_measured_path_gpu.py -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "persistent" > $O/tests2.txt 2>&1; tail -2 $O/tests2.txt
timeout 900 python3 tools/ab_inproc.py gemm_wt=0,1 gemm_aux_nt=0,1 gemm_dynamic=0,1 gemm_wt=0,1 > $O/ab.txt 2>&1; grep -v amdgpu.ids $O/ab.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
