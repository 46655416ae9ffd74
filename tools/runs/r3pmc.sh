set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 0 42; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p1_$c -- python3 $R/tools/attn_one.py $c > /dev/null 2> $O/e1_$c.txt
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2_$c -- python3 $R/tools/attn_one.py $c > /dev/null 2> $O/e2_$c.txt
done
ls -R $O | head -30
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3pmc'
for d in sorted(glob.glob(O+'/p*')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        print(os.path.basename(d))
        for k,v in acc.items():
            if 'mhsa' in k: print('  ',k, {a:round(b/3/1e6,2) for a,b in v.items()})
PY
