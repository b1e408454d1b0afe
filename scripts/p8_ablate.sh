#!/bin/bash
# timing of the p8 kernel's ablation builds (no A DMA / no B DMA / no DMA / no DMA + no fragment reads) and zero-data run
for v in 0 17 18 19 23; do echo -n "variant $v: "; python scripts/conv_one.py 64 60 80 256 256 3 1 p8$v | tail -1; done
echo -n "variant 0 zero data: "; python scripts/conv_one.py 64 60 80 256 256 3 1 p80 x zero | tail -1
echo -n "bfrag3: "; python scripts/conv_one.py 64 60 80 256 256 3 1 bfrag3 | tail -1
