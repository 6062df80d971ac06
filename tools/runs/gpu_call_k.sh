cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cat > /tmp/train_prof.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
torch.cuda.set_device(0)
print(bench.train_step_bench(torch.device("cuda", 0), steps=2, warmup=1))
PY
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o train -- python /tmp/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r02k_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_t -name "*results.db" | head -1) gpurun_out/r02k_train_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): 3 steps (1 warm-up + 2 timed) incl. model / optimiser-state construction" > /dev/null 2>&1
head -34 gpurun_out/r02k_train_kernel_stats.md
