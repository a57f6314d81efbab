set -u
O=gpurun_out/r03ba
mkdir -p $O
( timeout 600 python tools/partition_probe.py --reps 5 --t 32,64,96 > $O/partition_probe.txt 2> $O/partition_probe.err ); tail -12 $O/partition_probe.txt; tail -3 $O/partition_probe.err
( timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "dual" > $O/pytest_dual.txt 2>&1 ); tail -5 $O/pytest_dual.txt
run() {  # tag, env...
  tag=$1; shift
  ( env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu --no-f32mfma $PAR > $O/bench_c2_$tag.json 2> $O/bench_c2_$tag.err )
  python -c "
import json; d=json.load(open('$O/bench_c2_$tag.json')); print('$tag', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'frac_step', round(d['roofline']['frac_step'],3), {k[5:24]: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 4.0}, (d.get('parity') or {}).get('rel_err_x0_recon'))" || tail -3 $O/bench_c2_$tag.err
}
PAR=--no-parity
run single BBDM_DUAL_CHAIN=0
PAR=
run t64 BBDM_DUAL_T=64
PAR=--no-parity
run t32 BBDM_DUAL_T=32
run t96 BBDM_DUAL_T=96
run t64_96_32 BBDM_DUAL_T=64,65536:96,4096:32
run single2 BBDM_DUAL_CHAIN=0
