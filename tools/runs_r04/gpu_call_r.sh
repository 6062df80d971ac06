# Round 4, call R: kernel trace of the cfg 5 step with the fp32-class VAE encodes in its end-to-end part (refreshes the train-step stats).
cd /tmp && export TMPDIR=/tmp
( TRAIN_STEPS=2 TRAIN_WARMUP=1 timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o train -- python $GRAFT_REPO_ROOT/tools/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r04r_proftrain_stdout.log 2>&1; echo "proftrain rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_r -name "*results.db" | head -1) gpurun_out/r04r_train_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): TRAIN_STEPS=2 TRAIN_WARMUP=1 python tools/train_prof.py = 1 warm-up + 2 timed core steps, then 4 end-to-end steps (two fp32-class VAE encodes + VLM forward + core step), incl. model / optimiser-state construction" > /dev/null 2>&1
head -30 gpurun_out/r04r_train_kernel_stats.md
