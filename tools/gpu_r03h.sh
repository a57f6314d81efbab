set -u
O=gpurun_out/r03h
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_parity_gpu.py -q -x -k "attention" > $O/tests_attn.log 2>&1; echo "tests rc=$?" >> $O/tests_attn.log; tail -2 $O/tests_attn.log
BBDM_ATTN_WAVES=8 timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_parity_gpu.py -q -x -k "attention" > $O/tests_attn8.log 2>&1; echo "tests rc=$?" >> $O/tests_attn8.log; tail -2 $O/tests_attn8.log
for v in "BBDM_ATTN_BF3=2" "BBDM_ATTN_BF3=1" "BBDM_ATTN_WAVES=8"; do ( env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-parity --no-f32mfma > $O/bench_c2_$v.json 2> $O/err.txt ); python - "$v" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r03h/bench_c2_%s.json"%sys.argv[1]))
print(sys.argv[1], round(d["ms_per_step"],2), "ms attention", round(d["kernel_ms_per_step"]["bbdm_attention_f32"],2))
PY
done
