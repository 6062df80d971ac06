# Round 5, call K: the 4-wave attention forward as the default, inside the edits: cfg 2 and the 1024^2 edit with FK_ATTN_KERNEL=8
# and 4 (same box, same build), then the attention parity tests incl. the two-kernel bit-equality test and the training tests
# (the log-sum-exp output now comes from the 4-wave kernel).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r05k_edit_ab.txt
: > $O
for k in 8 4 8 4; do
  FK_ATTN_KERNEL=$k timeout 400 python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline none 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['roofline']['workloads']['cfg2_single_512x512_28step']
print('cfg2 k$k', round(d['value'],4), 'img/s', round(d['ms_per_step'],1), 'ms  attention', w.get('attention_tflops'), 'TF/s  gemm', w.get('gemm_tflops'))" >> $O
done
for k in 8 4; do
  FK_ATTN_KERNEL=$k timeout 400 python bench.py --workload single_1024x1024_28step --steps 2 --warmup 1 --no-extra --cpu-baseline none 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); w=d['roofline']['workloads'].get('single_1024x1024_28step', {})
print('1024sq k$k', round(d['value'],5), 'img/s', round(d['ms_per_step'],1), 'ms  attention', w.get('attention_tflops'), 'TF/s  gemm', w.get('gemm_tflops'))" >> $O
done
cat $O
( timeout 900 python -m pytest -x -q tests/test_hip_kernels.py -k attention tests/test_hip_cfg3.py -k "attention or stream" tests/test_hip_training.py -k attention tests/test_hip_train_step.py > gpurun_out/r05k_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05k_tests.log ); tail -4 gpurun_out/r05k_tests.log
