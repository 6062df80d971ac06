# closing: train-step tests after the gradient hand-out change, cfg 3 line on the final kernels, kernel stats of the 1024^2 edit
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_hip_train_step.py tests/test_hip_backward.py -x -q -m gpu > gpurun_out/r02ag_tests.log 2>&1; echo "pytest rc=$?" ); tail -2 gpurun_out/r02ag_tests.log
cd /tmp && export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o bench -- python $GRAFT_REPO_ROOT/bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none > $GRAFT_REPO_ROOT/gpurun_out/r02ag_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_g -name "*results.db" | head -1) gpurun_out/r02ag_1024_kernel_stats.md "python bench.py --workload single_1024x1024_28step --steps 1 --warmup 1 --no-extra --cpu-baseline none (3 edits: warm-up, timed with the MLP-up GEMM on the second stream, HIP-event pass on one stream)" > /dev/null 2>&1
head -14 gpurun_out/r02ag_1024_kernel_stats.md
( timeout 600 python bench.py --workload cfg3_batch32_1024x1024_28step --steps 1 --warmup 1 --cpu-baseline none > gpurun_out/r02ag_cfg3.json 2> gpurun_out/r02ag_cfg3.err; echo "cfg3 rc=$?" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02ag_cfg3.json')); r=d['roofline']
print('cfg3', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], 'attn', r['other_kernels']['attention']['tflops'])
PY
