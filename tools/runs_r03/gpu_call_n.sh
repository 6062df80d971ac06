# call N: training packs (aliased QKV operands, per-block modulation GEMMs), gradients cast straight into the optimiser's chunk
# on one rank: parity of everything that reads the packs, then the cfg 5 step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_hip_mmdit.py tests/test_hip_pipeline.py tests/test_hip_train_step.py tests/test_hip_train_seam.py tests/test_hip_cfg5.py tests/test_hip_training.py tests/test_hip_backward.py -q -m gpu -x > gpurun_out/r03n_tests.log 2>&1; echo "pytest rc=$?" ); tail -8 gpurun_out/r03n_tests.log
( timeout 600 python tools/train_prof.py > gpurun_out/r03n_cfg5.json 2> gpurun_out/r03n_cfg5.err; echo "cfg5 rc=$?" ); tail -2 gpurun_out/r03n_cfg5.err
python -c "
import json; d=json.loads(open('gpurun_out/r03n_cfg5.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','peak_memory_gb','frac_of_mfma_peak_3x_forward','zero2_buckets')})"
( timeout 600 python tools/train_ops_prof.py 45 > gpurun_out/r03n_train_ops.txt 2> gpurun_out/r03n_train_ops.err; echo "ops rc=$?" ); grep -v "anonymous namespace" gpurun_out/r03n_train_ops.txt | cut -c1-200 | head -30
