set -u
O=gpurun_out/r03d
mkdir -p $O
timeout 900 python tools/wino_variants.py --reps 10 > $O/wino_variants.txt 2>&1
cat $O/wino_variants.txt | cut -c1-330
