# Round 4, call B: stream-K attention forward -- parity tests, isolated A/B against the plain grid, A/B inside the 1024^2 edit.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_cfg3.py tests/test_hip_cfg5.py tests/test_hip_pipeline.py "tests/test_hip_kernels.py" -m gpu -x -q -s -k "attention or cfg3 or cfg5 or smoke or full_size or graph" > gpurun_out/r04b_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04b_tests.log ); tail -4 gpurun_out/r04b_tests.log
grep -h "stream-K\|outlier" gpurun_out/r04b_tests.log | cut -c1-260 | head -40
( timeout 300 python tools/ab_attention_split.py > gpurun_out/r04b_ab_attention_split.txt 2>&1; echo "ab rc=$?" ); cat gpurun_out/r04b_ab_attention_split.txt | tail -8
( AB_ARMS="split=0;split=1" timeout 400 python tools/ab_edit_plans.py single_1024x1024_28step 2 1 > gpurun_out/r04b_ab_edit_1024.txt 2>&1; echo "ab edit rc=$?" ); tail -6 gpurun_out/r04b_ab_edit_1024.txt
( AB_ARMS="split=0;split=1" timeout 400 python tools/ab_edit_plans.py cfg2cli_512x512_cond1mp_28step 2 1 > gpurun_out/r04b_ab_edit_cli.txt 2>&1; echo "ab edit cli rc=$?" ); tail -6 gpurun_out/r04b_ab_edit_cli.txt
