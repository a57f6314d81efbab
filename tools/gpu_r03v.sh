set -u
O=gpurun_out/r03v
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "phase or winograd or bf3p" > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -3 $O/tests1.log
( timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --dump-ops $O/c2_per_launch.md > $O/bench_c2.json 2> $O/bench_c2.err ); tail -2 $O/bench_c2.err
python -c "
import json; d=json.load(open('$O/bench_c2.json')); print('c2', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['frac_step'], {k:v for k,v in d['parity'].items() if k.startswith('rel')}); print({k: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 0.3})"
( BBDM_UPSAMPLE_PHASES=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --no-parity --no-f32mfma > $O/bench_c2_nophase.json 2> $O/bench_c2_nophase.err )
python -c "
import json; d=json.load(open('$O/bench_c2_nophase.json')); print('c2 no phases', round(d['ms_per_step'],3)); print({k: round(v,2) for k,v in sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1]) if v > 0.3})"
