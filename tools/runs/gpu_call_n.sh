cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_train_step.py -q -x > gpurun_out/r02n_tests.log 2>&1; echo "pytest rc=$?" )
tail -3 gpurun_out/r02n_tests.log
( timeout 600 python - > gpurun_out/r02n_cfg5.json 2> gpurun_out/r02n_cfg5.err <<'PY'
import json, torch, bench
torch.cuda.set_device(0)
r = bench.train_step_bench(torch.device("cuda", 0), steps=3, warmup=1)
r["max_memory_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(r))
PY
echo "cfg5 rc=$?" )
python -c "
import json; d=json.load(open('gpurun_out/r02n_cfg5.json')); print({k:d[k] for k in ('value','ms_per_step','max_memory_GB','frac_of_mfma_peak_3x_forward')})"
tail -2 gpurun_out/r02n_cfg5.err
