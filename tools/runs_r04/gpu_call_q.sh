# Round 4, call Q: attention forward with the ring stage as a compile-time constant of the tile body (DS immediates instead of
# address additions): targeted tests, A/B against the previous build of the library (FK_LIB_PATH), then the full GPU suite.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
BASE=$PWD/gpt_image_edit_amd/libfk_ab_base.so
( timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "attention" > gpurun_out/r04q_tests_attn.log 2>&1; echo "attn tests rc=$?" | tee -a gpurun_out/r04q_tests_attn.log ); tail -3 gpurun_out/r04q_tests_attn.log
for i in 1; do
  ( AB_SHAPES=1x2560,1x8704,4x8704 FK_LIB_PATH=$BASE timeout 120 python tools/ab_attention_split.py 2>/dev/null | sed "s/^/base $i: /" ) | tee -a gpurun_out/r04q_ab_isolated.txt
  ( AB_SHAPES=1x2560,1x8704,4x8704 timeout 120 python tools/ab_attention_split.py 2>/dev/null | sed "s/^/new  $i: /" ) | tee -a gpurun_out/r04q_ab_isolated.txt
done
for arm in base new; do
  if [ "$arm" = "base" ]; then export FK_LIB_PATH=$BASE; else unset FK_LIB_PATH; fi
  ( timeout 200 python bench.py --workload single_1024x1024_28step --steps 2 --warmup 1 --no-extra --cpu-baseline none --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm 1024^2', d['value'], d['ms_per_step'])" ) | tee -a gpurun_out/r04q_ab_edit.txt
done
unset FK_LIB_PATH
( timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04q_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04q_tests.log ); tail -3 gpurun_out/r04q_tests.log
