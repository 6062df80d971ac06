# call Q: K-major operands -- GEMM parity (no split-K in the reference), gradients bit for bit across FK_BWD_K_MAJOR levels,
# the cfg 5 step at levels 0 / 1 / 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_train_step.py tests/test_hip_train_seam.py -q -m gpu -k "gemm or train or seam or k_major or backward" -x > gpurun_out/r03q_tests.log 2>&1; echo "pytest rc=$?" ); tail -12 gpurun_out/r03q_tests.log | cut -c1-200
for lv in 0 1 2 0 1 2; do
  ( FK_BWD_K_MAJOR=$lv TRAIN_STEPS=4 timeout 400 python tools/train_prof.py > gpurun_out/r03q_cfg5_$lv.json 2> gpurun_out/r03q_cfg5_$lv.err; echo "cfg5 level $lv rc=$?" )
  python -c "
import json; d=json.loads(open('gpurun_out/r03q_cfg5_$lv.json').read().strip().splitlines()[-1]); print('level $lv', {k:d[k] for k in ('ms_per_step','host_enqueue_ms_per_step','peak_memory_gb')})"
done
