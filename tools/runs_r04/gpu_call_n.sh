# Round 4, call N: the driver's bench command on the final tree (cfg 3 batch now among the extras), then the attention main-loop
# order inside the 512^2 edit (FK_ATTN_ILV, separate processes, alternating).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
t0=$(date +%s)
( timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04n_bench_driver_cmd.json 2> gpurun_out/r04n_bench_driver_cmd.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s" )
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04n_bench_driver_cmd.json'))
r=d['roofline']; e=d.get('extra',{})
print('cfg2', d['value'], d['ms_per_step'], 'gemm', r['achieved'], r['frac'], r['launches_per_edit'], 'attn', r['other_kernels']['attention']['tflops'])
x=e.get('single_1024x1024_28step'); print('1024', x and x['value'])
print('cfg3', e.get('cfg3_batch32_1024x1024_28step'))
t=e.get('cfg5_train_step_1024x1024_bs1',{}); print('cfg5', t.get('ms_per_step'), t.get('error'), (t.get('T_step_e2e') or {}).get('ms_per_step'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
for i in 1 2; do for ilv in -1 1; do
  if [ "$ilv" = "-1" ]; then unset FK_ATTN_ILV; else export FK_ATTN_ILV=$ilv; fi
  ( timeout 200 python bench.py --steps 6 --warmup 2 --no-extra --cpu-baseline none --no-roofline > gpurun_out/r04n_ilv_${ilv}_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04n_ilv_${ilv}_$i.json')); print('ILV=$ilv run $i', d['value'], d['ms_per_step'])" )
done; done
