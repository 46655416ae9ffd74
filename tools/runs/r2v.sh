R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DEVIAS_GEMM_RING4=1
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/tools/pmc_gemm.py > /dev/null 2> $O/p$i.err
done
python3 - <<'PY'
import csv, glob, collections, os, statistics as st
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r2v'
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm256' not in k: continue
        res[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in res.items():
    print(k[:80])
    for c,vals in sorted(v.items()):
        print(f"   {c:32s} {st.median(vals):16.0f}")
PY
