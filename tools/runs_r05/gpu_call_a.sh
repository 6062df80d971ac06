# Round 5, call A: (1) power probes of the GEMM main loop's ingredients (tools/power_probe.hip, built in-tree);
# (2) the tightened parity tests (direct HIP-vs-bf16-oracle bounds, full depth at S = 2560) and smoke().
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 120 tools/build/power_probe 1.5 > gpurun_out/r05a_power_probe.txt 2>&1; echo "probe rc=$?" ) ; cat gpurun_out/r05a_power_probe.txt
( timeout 1500 python -m pytest -x -q -s tests/test_hip_pipeline.py::test_smoke_entry tests/test_hip_pipeline.py::test_edit_matches_oracle_pipeline tests/test_hip_pipeline.py::test_28_step_edit_matches_oracle_pipeline tests/test_hip_mmdit.py::test_full_depth_mmdit_matches_block_streamed_oracle tests/test_hip_mmdit.py::test_full_depth_mmdit_at_cfg2_size_matches_bf16_oracle > gpurun_out/r05a_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r05a_tests.log )
grep -E "direct bound|smoke:|passed|failed|Error|oracle on the host|oracle at S" gpurun_out/r05a_tests.log | tail -30
