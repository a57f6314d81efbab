set -u
O=gpurun_out/r03u
mkdir -p $O
( timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err ); tail -2 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4 graph', round(d['ms_per_step'],3), d['parity'] and {k:v for k,v in d['parity'].items() if k.startswith('rel')})"
( BBDM_TRAIN_GRAPH=0 timeout 300 python bench.py --workload c4 --no-cpu --no-parity > $O/bench_c4_eager.json 2> $O/bench_c4_eager.err ); python -c "
import json; d=json.load(open('$O/bench_c4_eager.json')); print('c4 eager', round(d['ms_per_step'],3))"
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_backward_kernels_gpu.py -q -x > $O/tests1.log 2>&1; echo "rc=$?" >> $O/tests1.log; tail -3 $O/tests1.log
