# GPU call A of round 2: full GPU test suite, the contract bench (with its extras), HBM-traffic PMC passes, kernel stats.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02a_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a_tests.log ) 
tail -3 gpurun_out/r02a_tests.log
( timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/r02a_bench_cfg2.json 2> gpurun_out/r02a_bench_cfg2.err; echo "bench rc=$?" )
tail -c 600 gpurun_out/r02a_bench_cfg2.json
bash tools/pmc_traffic.sh > gpurun_out/r02a_traffic.log 2>&1
tail -8 gpurun_out/r02a_traffic.log
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r02a_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_a -name "*results.db" | head -1) gpurun_out/r02a_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 (cfg2; 3 edits: warm-up, timed, HIP-event pass)" > /dev/null 2>&1
head -20 gpurun_out/r02a_bench_kernel_stats.md
