cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_train_step.py -q -x > gpurun_out/r02j_train_tests.log 2>&1; echo "pytest rc=$?" )
tail -3 gpurun_out/r02j_train_tests.log
( timeout 900 python - > gpurun_out/r02j_cfg5.json 2> gpurun_out/r02j_cfg5.err <<'PY'
import json, torch, bench
torch.cuda.set_device(0)
r = bench.train_step_bench(torch.device("cuda", 0), steps=3, warmup=1)
r["max_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(r))
PY
echo "cfg5 rc=$?" )
cat gpurun_out/r02j_cfg5.json | cut -c1-700; tail -3 gpurun_out/r02j_cfg5.err
