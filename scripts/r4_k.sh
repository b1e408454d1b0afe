#!/bin/bash
# what each stage costs once batches overlap: truncated pipelines, 4 in flight
O=gpurun_out
F="--no-cpu-baseline --no-boundary --no-fp32-path --no-accuracy --no-other-configs --no-tape --routing $O/routing_r4.json --steps 40"
for a in backbone head nocam; do echo "ablate $a: $(python bench.py $F --ablate $a 2>/dev/null | tail -1)"; done | tee $O/r4_k_ablate.txt
python bench.py $F --stages 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full', d['value'], d['ms_per_step']); print(d.get('stage_ms_main_stream'))" | tee -a $O/r4_k_ablate.txt
