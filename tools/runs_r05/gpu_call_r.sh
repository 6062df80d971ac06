# Round 5, call R: (1) beyond-L2 traffic of the SHIPPED attention forward (4 waves x 64 rows) -- calibration + the three attention
# items of tools/traffic_target.py, one counter group per pass; (2) kernel trace of the cfg 5 optimisation step on the final tree.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
TRAFFIC_SKIP="ln_modulate,gemm,vae" PMC_PASSES="time fetch write hit" bash tools/pmc_traffic.sh
python tools/pmc_traffic_summary.py gpurun_out/traffic gpurun_out/r05r_traffic_attention4 2>&1 | tail -3
cat gpurun_out/r05r_traffic_attention4.md | tail -8
cd /tmp && export TMPDIR=/tmp
( TRAIN_STEPS=3 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o train -- python $GRAFT_REPO_ROOT/tools/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r05r_proftrain_stdout.log 2>&1; echo "proftrain rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_r -name "*results.db" | head -1) gpurun_out/r05r_train_step_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): python tools/train_prof.py = 2 warm-up + 3 timed core steps, then the end-to-end steps (VAE encodes + VLM forward + core step), incl. model / optimiser-state construction" > /dev/null 2>&1
head -30 gpurun_out/r05r_train_step_kernel_stats.md
tail -2 gpurun_out/r05r_proftrain_stdout.log | cut -c1-600
