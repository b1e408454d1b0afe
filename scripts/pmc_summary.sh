#!/bin/bash
# PMC summary of ONE single-kernel harness as a JSON file (separate counter passes; counters only - no trace domains).
# Usage: bash scripts/pmc_summary.sh <out-json> <kernel-name-substring> <script.py> [args]
#   -> MFMA-busy, SQ_WAIT_ANY / issue-stall / active shares of the wave cycles, VALU and SALU instructions per MFMA, LDS bank
#      conflicts, L2 hit rate and request bytes, HBM FETCH / WRITE bytes (FETCH_SIZE doubled: gfx950 correction, MI355X_MICROARCH.md)
R=$PWD; export TMPDIR=/tmp; cd /tmp
OUT=$1; KN=$2; shift; shift
rm -rf $R/gpurun_out/pmc_sum
python $R/scripts/"$@" > $R/gpurun_out/pmc_sum_unprofiled.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sum/$tag -o pmc -- python $R/scripts/"$@" > /dev/null 2>&1
done
cd $R
KN="$KN" OUT="$OUT" CMD="$*" python - <<'PY'
import csv, glob, collections, os, json
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("gpurun_out/pmc_sum/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ", "")
        if os.environ["KN"] not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
out={"command": "scripts/" + os.environ["CMD"], "unprofiled_run": open("gpurun_out/pmc_sum_unprofiled.log").read().strip().splitlines()[-1:],
     "note": "per launch; rocprofv3 --pmc in 5 separate passes (counters only); SQ_* wave-cycle counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES "
             "and GRBM_GUI_ACTIVE (summed over the 8 XCDs) in cycles; HBM read = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB", "kernels": {}}
for k,v in agg.items():
    c={name: val/n[(k,name)] for name,val in v.items()}
    cyc=c.get("GRBM_GUI_ACTIVE",0)/8
    d={"counters": {a: round(b,1) for a,b in sorted(c.items())}}
    wc=c.get("SQ_WAVE_CYCLES",0)
    if cyc and c.get("SQ_VALU_MFMA_BUSY_CYCLES"): d["mfma_busy_frac_of_simd_cycles"]=round(c["SQ_VALU_MFMA_BUSY_CYCLES"]/(cyc*1024),4)
    d["cycles_per_launch"]=round(cyc)
    if wc:
        d["wave_cycle_shares"]={"wait_any": round(c.get("SQ_WAIT_ANY",0)/wc,3), "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY",0)/wc,3), "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY",0)/wc,3)}
    if c.get("SQ_INSTS_MFMA"):
        d["valu_per_mfma"]=round((c.get("SQ_INSTS_VALU",0)-c["SQ_INSTS_MFMA"])/c["SQ_INSTS_MFMA"],2)
        d["salu_per_mfma"]=round(c.get("SQ_INSTS_SALU",0)/c["SQ_INSTS_MFMA"],2)
        d["lds_insts_per_mfma"]=round(c.get("SQ_INSTS_LDS",0)/c["SQ_INSTS_MFMA"],2)
    if c.get("SQ_LDS_IDX_ACTIVE"): d["lds_bank_conflict_frac"]=round(c.get("SQ_LDS_BANK_CONFLICT",0)/c["SQ_LDS_IDX_ACTIVE"],4)
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c.get("TCC_HIT_sum",0)+c.get("TCC_MISS_sum",0)>0:
        d["l2_hit_rate"]=round(c["TCC_HIT_sum"]/(c["TCC_HIT_sum"]+c["TCC_MISS_sum"]),4)
    if c.get("TCC_REQ_sum"): d["l2_request_bytes"]=round(c["TCC_REQ_sum"]*128)
    d["hbm_read_bytes"]=round(2*c.get("FETCH_SIZE",0)*1024); d["hbm_write_bytes"]=round(c.get("WRITE_SIZE",0)*1024)
    out["kernels"][k]=d
json.dump(out, open(os.environ["OUT"],"w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} for k, v in out["kernels"].items()}, indent=1))
PY
