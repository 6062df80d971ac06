cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_backward.py tests/test_hip_train_step.py -q -x > gpurun_out/r02l_tests.log 2>&1; echo "pytest rc=$?" )
tail -3 gpurun_out/r02l_tests.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/train_prof.py <<'PY'
import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
torch.cuda.set_device(0)
r = bench.train_step_bench(torch.device("cuda", 0), steps=3, warmup=1)
r["max_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(r))
PY
( timeout 600 python /tmp/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r02l_cfg5.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02l_cfg5.err; echo "cfg5 rc=$?" )
cut -c1-330 $GRAFT_REPO_ROOT/gpurun_out/r02l_cfg5.json
( timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o train -- python /tmp/train_prof.py > $GRAFT_REPO_ROOT/gpurun_out/r02l_prof_stdout.log 2>&1; echo "prof rc=$?" )
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_t -name "*results.db" | head -1) gpurun_out/r02l_train_kernel_stats.md "cfg 5 train step (1024^2, bs 1, full depth): 4 steps (1 warm-up + 3 timed) incl. model / optimiser-state construction" > /dev/null 2>&1
head -24 gpurun_out/r02l_train_kernel_stats.md
