#!/bin/bash
set -u
O=gpurun_out/r02n
mkdir -p $O
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "statistics or wino" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -h "passed\|failed\|rc=\|FAILED" $O/tests.log | tail -3
for w in c2 c3; do ( timeout 400 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu > $O/bench_$w.json 2> $O/bench_$w.err ); python - <<PY
import json
d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms_per_step'].items()})
PY
done
