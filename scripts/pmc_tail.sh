#!/bin/bash
# PMC counters for one fused-tail configuration (separate passes; counters only).  Usage: bash scripts/pmc_tail.sh B OH OW C C4 CN [C2 stride]
R=$PWD; export TMPDIR=/tmp; cd /tmp
ARGS="${@:-64 60 80 128 512 128}"
python $R/scripts/tail_one.py $ARGS
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE32_INSTS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_tail/$tag -o pmc -- python $R/scripts/tail_one.py $ARGS > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_tail/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        if "pw_chain" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    print(k)
    for c,val in sorted(v.items()): print("   %-28s %.4g"%(c, val/25))
PY
