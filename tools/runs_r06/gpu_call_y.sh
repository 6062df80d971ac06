# Round 6, call Y: kernel trace of the cfg 5 core step (2 warm-up + 3 timed steps, no end-to-end part) on the last tree.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
( TRAIN_E2E=0 TRAIN_STEPS=3 TRAIN_WARMUP=2 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o train -- python $GRAFT_REPO_ROOT/tools/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r06y_train_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_t -name "*results.db" | head -1) gpurun_out/r06y_train_step_kernel_stats.md "cfg 5 core train step (1024^2, bs 1, full depth): TRAIN_E2E=0 TRAIN_STEPS=3 TRAIN_WARMUP=2 python tools/train_prof.py = 5 core steps incl. model / optimiser-state construction" > /dev/null 2>&1
head -50 gpurun_out/r06y_train_step_kernel_stats.md
grep -a '^{' gpurun_out/r06y_train_stdout.log | tail -1 | cut -c1-600
