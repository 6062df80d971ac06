cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -q -m gpu -x --durations=8 > gpurun_out/r03c_tests.log 2>&1; echo "pytest rc=$?" ); tail -14 gpurun_out/r03c_tests.log
( FK_ATTN_TAIL=0 timeout 200 python tools/ab_attention.py tail0 > gpurun_out/r03c_ab_attn.log 2>&1; FK_ATTN_TAIL=1 timeout 200 python tools/ab_attention.py tail1 >> gpurun_out/r03c_ab_attn.log 2>&1; echo "ab_attn rc=$?" ); grep attention gpurun_out/r03c_ab_attn.log
( AB_ARMS="tail=0,side=0;tail=1,side=0;tail=0,side=1;tail=1,side=1" timeout 500 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r03c_ab_edit_1024.log 2>&1; echo "ab_edit1024 rc=$?" ); tail -9 gpurun_out/r03c_ab_edit_1024.log
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r03c_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_c -name "*results.db" | head -1) gpurun_out/r03c_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass)" > /dev/null 2>&1
head -40 gpurun_out/r03c_bench_kernel_stats.md
