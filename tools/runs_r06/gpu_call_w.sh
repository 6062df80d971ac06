# Round 6, call W: the GPU suite with every round-6 switch turned OFF (the paths of round 5 stay green behind them)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( FK_GEMM10=0 FK_BWD_BLOCK_API=0 FK_VAE_FUSED_ATTN=0 FK_KMAJOR_MFMA=32 timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_kernels.py::test_gemm10_one_workgroup_per_cu_form_and_the_plan_that_picks_it --deselect tests/test_hip_train_step.py::test_block_level_backward_entry_points_give_the_same_bits --deselect tests/test_hip_vae.py::test_vae_decode_512sq_allocates_nothing_of_size_S_squared 2>&1 | tail -6 ) > gpurun_out/r06w_tests_switches_off.log 2>&1
cat gpurun_out/r06w_tests_switches_off.log
