mkdir -p gpurun_out/r01f
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 400 python bench.py --steps 10 --warmup 3 --dump-ops gpurun_out/r01f/c2_ops.md > gpurun_out/r01f/bench_c2.json 2> gpurun_out/r01f/bench_c2.err
for w in c1 c3 c5 c4; do timeout 200 python bench.py --workload $w --no-cpu > gpurun_out/r01f/bench_$w.json 2> gpurun_out/r01f/bench_$w.err; done
rocprofv3 --kernel-trace --stats -d gpurun_out/r01f/prof_c2 -o c2 -- python bench.py --workload c2 --steps 3 --warmup 1 --no-cpu > gpurun_out/r01f/prof_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r01f/pmc_fetch -o pmc -- python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu > gpurun_out/r01f/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r01f/pmc_write -o pmc -- python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu > gpurun_out/r01f/pmc_write.log 2>&1
find gpurun_out/r01f -name "*.db" | head
tail -c 600 gpurun_out/r01f/bench_c2.json
