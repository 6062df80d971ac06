# Round 5, call N: after making the stream-K seam's merge symmetric (both products rounded): outputs and log-sum-exps of the two
# forwards at B1 H24 S5632, the attention / stream-K / training tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for k in 4 8; do echo "kernel $k S5632 default grid"; FK_ATTN_KERNEL=$k timeout 100 python tools/attn_dump.py 1 24 5632 /tmp/k$k.pt; done
python tools/attn_diff.py /tmp/k4.pt /tmp/k8.pt 24
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05n_attn_diff.txt
( timeout 900 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream or batch32" tests/test_hip_training.py -k attention tests/test_hip_train_step.py > gpurun_out/r05n_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05n_tests.log ); tail -4 gpurun_out/r05n_tests.log
