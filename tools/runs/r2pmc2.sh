R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/tools/pmc_gemm.py > /dev/null 2> $O/p$i.err
done
python3 - <<'PY'
import csv, glob, collections, os, statistics as st
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r2ad'
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'gemm256' not in k: continue
        res[k][r['Counter_Name']].append(float(r['Counter_Value']))
        res[k]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
with open(O+'/summary.txt','w') as fo:
    for k,v in res.items():
        fo.write(k[:80]+'\n')
        for c,vals in sorted(v.items()):
            fo.write(f"   {c:32s} {st.median(vals):16.0f}\n")
print(open(O+'/summary.txt').read())
PY
