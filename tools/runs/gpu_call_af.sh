# A/B: weight-gradient GEMMs of the backward on a second stream (FK_OVERLAP_WGRAD=1) vs one stream (0)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_hip_train_step.py -x -q -m gpu > gpurun_out/r02af_tests.log 2>&1; echo "pytest rc=$?" ); tail -2 gpurun_out/r02af_tests.log
for ov in 0 1 0 1; do
  FK_OVERLAP_WGRAD=$ov timeout 300 python -c "
import bench, torch, json
r = bench.train_step_bench(torch.device('cuda', 0), steps=3, warmup=1)
print('wgrad overlap $ov:', round(r['ms_per_step'], 1), 'ms/step', round(r['value'], 4), 'samples/s, loss', r['loss'])
" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02af_cfg5.txt
done
