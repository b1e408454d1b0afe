#!/bin/bash
# GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) per launch for the p8 kernel's ablation builds: cycle counts are
# comparable across variants, wall times are not (each variant settles at its own clock).
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/p8_cycles
for v in p80 p832 p817 p818 p819 p823 bfrag3; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/p8_cycles/$v -o pmc -- python $R/scripts/conv_one.py ${SHAPE:-64 60 80 256 256 3 1} $v > $R/gpurun_out/p8_cycles_$v.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/p8_cycles/*/")):
    agg=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob(d+"*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "conv_igemm" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    tag=os.path.basename(d.rstrip("/"))
    us=open("gpurun_out/p8_cycles_%s.log"%tag).read().strip().split()[-1]
    g=agg["GRBM_GUI_ACTIVE"]/max(n["GRBM_GUI_ACTIVE"],1)/8
    wc=agg["SQ_WAVE_CYCLES"]/max(n["SQ_WAVE_CYCLES"],1)
    print("%-8s us/launch(profiled) %7s  cycles/launch %9.0f  => clock %.2f GHz | wave-cycles(quad) %.3g wait_any %.2f wait_inst %.2f active %.2f" % (
        tag, us, g, g/float(us)/1e3, wc, agg["SQ_WAIT_ANY"]/n["SQ_WAIT_ANY"]/wc, agg["SQ_WAIT_INST_ANY"]/n["SQ_WAIT_INST_ANY"]/wc, agg["SQ_ACTIVE_INST_ANY"]/n["SQ_ACTIVE_INST_ANY"]/wc))
PY
