# res3's edge tails: the 96-pixel form (default) vs the 128-pixel form (NOPESAC_TAIL_NO_RT4H=1), interleaved
for rep in 1 2 3; do
for cfg in "64 60 80 128 512 128 256 2" "64 60 80 128 512 256"; do
python scripts/tail_one.py $cfg | tail -1
NOPESAC_TAIL_NO_RT4H=1 python scripts/tail_one.py $cfg | tail -1 | sed 's/^/   128-pixel form: /'
done; done
