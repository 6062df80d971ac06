cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python bench.py --workload cfg3_batch32_1024x1024_28step --steps 1 --warmup 1 --cpu-baseline none --no-extra > gpurun_out/r02o_bench_cfg3.json 2> gpurun_out/r02o_bench_cfg3.err; echo "cfg3 rc=$?" )
python -c "
import json; d=json.load(open('gpurun_out/r02o_bench_cfg3.json')); r=d['roofline']; print('cfg3', d['value'], d['ms_per_step'], 'gemm', r['achieved'], 'attn', r['other_kernels']['attention']['tflops'])"
( timeout 600 python bench.py --workload cfg2cli_512x512_cond1mp_28step --steps 3 --warmup 1 --cpu-baseline none --no-extra > gpurun_out/r02o_bench_cfg2cli.json 2> gpurun_out/r02o_bench_cfg2cli.err; echo "cfg2cli rc=$?" )
python -c "
import json; d=json.load(open('gpurun_out/r02o_bench_cfg2cli.json')); r=d['roofline']; print('cfg2cli', d['value'], d['ms_per_step'], 'gemm', r['achieved'], 'attn', r['other_kernels']['attention']['tflops'])"
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r02o_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_a -name "*results.db" | head -1) gpurun_out/r02o_bench_kernel_stats.md "python bench.py --steps 1 --warmup 1 --no-extra --cpu-baseline none (cfg2; 3 edits: warm-up, timed, HIP-event pass)" > /dev/null 2>&1
head -16 gpurun_out/r02o_bench_kernel_stats.md
