# Round 4, call O: the fp32-class encoder's 3 x 3 convolutions on the halo kernel (parity + timing); 2-rank smokes of the final bench.py.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 400 python -m pytest tests/test_hip_vae.py -m gpu -x -q -s > gpurun_out/r04o_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04o_tests.log ); grep "fp32\|passed\|failed\|Error" gpurun_out/r04o_tests.log | tail -24
( timeout 300 python tools/time_vae_encode.py > gpurun_out/r04o_vae_encode_time.txt 2>&1; echo "time rc=$?" ); grep -v amdgpu gpurun_out/r04o_vae_encode_time.txt | tail -4
( FK_VAE_HALO=0 timeout 300 python tools/time_vae_encode.py > gpurun_out/r04o_vae_encode_time_nohalo.txt 2>&1; echo "time(no halo) rc=$?" ); grep -v amdgpu gpurun_out/r04o_vae_encode_time_nohalo.txt | tail -2
export FK_BENCH_BACKEND=gloo
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --cpu-baseline none > gpurun_out/r04o_n2_weak.json 2> gpurun_out/r04o_n2_weak.err; echo "n2 weak rc=$?" )
unset FK_BENCH_BACKEND
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r04o_n2_weak.json')); print('n2', d['value'], d['n_gpus'], d['dist'], list(d.get('extra',{}).keys()))
except Exception as e: print('ERR', e)
PY
