R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
for a in "" "--cu-hog 16" "--cu-hog 16 --reserve-cus 16" "--cu-hog 16 --reserve-cus 24" "--reserve-cus 16" "--cu-hog 32 --reserve-cus 32" "--cu-hog 8 --reserve-cus 8"; do
  n=$(echo "base $a" | tr -d ' -' )
  timeout 600 python bench.py --no-cpu-baseline --no-full-step $a > $O/bench_$n.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$n.json").read().strip().splitlines()[-1])
print(f"{'$a':36s} {d['value']:7.1f} clips/s {d['ms_per_step']:7.2f} ms")
PY
done
