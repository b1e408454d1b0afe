#!/bin/bash
# PMC counters for one single-kernel harness script.  Usage: bash scripts/pmc_one.sh <kernel-name-substring> <script.py> [args]
R=$PWD; export TMPDIR=/tmp; cd /tmp
KN=$1; shift
python $R/scripts/"$@"
rm -rf $R/gpurun_out/pmc_one
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_one/$tag -o pmc -- python $R/scripts/"$@" > /dev/null 2>&1
done
cd $R
KN=$KN python - <<'PY'
import csv, glob, collections, os
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("gpurun_out/pmc_one/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        if os.environ["KN"] not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    print(k)
    for c,val in sorted(v.items()): print("   %-28s %.4g   (per launch)"%(c, val/n[(k,c)]))
PY
