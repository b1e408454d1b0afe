#!/bin/bash
# GPU call b of round 3: new tail kernel (tests, old-vs-new timing, PMC), K-control kernel test, bench line, host profile, backbone attribution
cd $GRAFT_REPO_ROOT 2>/dev/null || true
O=gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "bottleneck_tail or backbone_fused or force_k" 2>&1 | tail -5 > $O/r3_b_pytest.log
python -m pytest tests/test_stages_gpu.py -x -q -k "backbone" 2>&1 | tail -5 >> $O/r3_b_pytest.log
{
for cfg in "64 60 80 128 512 128" "64 60 80 128 512 0" "64 120 160 64 256 64" "64 120 160 64 256 128"; do
  echo "== $cfg"; NOPESAC_TAIL_NO_RT4=1 python scripts/tail_one.py $cfg; python scripts/tail_one.py $cfg
done
} > $O/r3_b_tail_timing.log 2>&1
bash scripts/pmc_tail.sh 64 60 80 128 512 128 > $O/r3_b_pmc_res3_tail.log 2>&1
bash scripts/pmc_tail.sh 64 120 160 64 256 64 > $O/r3_b_pmc_res2_tail.log 2>&1
python bench.py --no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs > $O/r3_b_bench.json 2> $O/r3_b_bench.err
python scripts/host_profile.py > $O/r3_b_host_profile.log 2>&1
python scripts/bf16_attribution.py --only "backbone fp32 through" --out $O/bf16_attribution_backbone.json > $O/r3_b_attr.log 2>&1
tail -3 $O/r3_b_pytest.log; cat $O/r3_b_tail_timing.log; tail -c 600 $O/r3_b_bench.json; tail -6 $O/r3_b_attr.log
