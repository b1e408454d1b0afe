#!/bin/bash
# Round-6 measurement legs that need no new kernels: baseline line, K = 64 / 128 kernel stats (which kernels carry the K cost), and the
# four-in-flight overlap trace with power / clock samples.  usage: bash scripts/r6_measure.sh <tag>
T=${1:-a}; O=$PWD/gpurun_out; R=$PWD; P=r6_${T}
mkdir -p $O
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape"
python bench.py --steps 20 --warmup 5 $F > $O/${P}_bench.json 2> $O/${P}_bench.err; cut -c1-300 $O/${P}_bench.json
export TMPDIR=/tmp; cd /tmp
# ---- K legs: kernel stats in the loop + isolated costs (one batch in flight, one stream)
for leg in "k32 mp3d 32 routing_r5.json" "k64 scannet 64 routing_r5_scannet_k64.json" "k128 mp3d 128 routing_r5_fp8_k128.json"; do
  set -- $leg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${P}_$1 -o bench -- python $R/bench.py --steps 24 --warmup 5 $F --config $2 --k $3 --routing $R/profiles/$4 > $O/${P}_$1_stats.log 2>&1
  cp $(find $O/prof_${P}_$1 -name "*kernel_stats.csv" | head -1) $O/${P}_$1_kernel_stats.csv
  tail -1 $O/${P}_$1_stats.log | cut -c1-200
  (cd $R; bash scripts/prof_isolated.sh ${P}_$1 --config $2 --k $3 --routing $R/profiles/$4 > $O/${P}_$1_isolated.log 2>&1; cp $O/iso_${P}_$1.tsv $O/${P}_$1_isolated_kernel_costs.txt)
  rm -rf $O/prof_${P}_$1 $O/prof_${P}_$1.log
done
# ---- overlap: kernel trace of the default loop (four in flight) with rocm-smi sampled next to it
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | tr '\n' ' '; echo; sleep 0.15; done ) > $O/${P}_smi.txt &
SMI=$!
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${P}_overlap -o bench -- python $R/bench.py --steps 60 --warmup 5 $F > $O/${P}_overlap.log 2>&1
kill $SMI
tail -1 $O/${P}_overlap.log | cut -c1-200
cd $R
python scripts/overlap_report.py $(find $O/prof_${P}_overlap -name "*kernel_trace.csv" | head -1) $O/${P}_smi.txt $O/${P}_overlap.json 40 | cut -c1-250
rm -rf $O/prof_${P}_overlap
