# N > 1 code path of bench.py on a 1-GPU box: two ranks sharing the GPU over gloo (FK_BENCH_BACKEND), weak and strong scaling
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export FK_BENCH_BACKEND=gloo
( timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r02z_n2_weak.json 2> gpurun_out/r02z_n2_weak.err; echo "weak rc=$?" )
tail -c 1500 gpurun_out/r02z_n2_weak.json; echo
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --scaling strong --global-batch 4 --no-extra > gpurun_out/r02z_n2_strong.json 2> gpurun_out/r02z_n2_strong.err; echo "strong rc=$?" )
python - <<'PY'
import json
for n in ("weak", "strong"):
    try:
        d = json.loads(open(f"gpurun_out/r02z_n2_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["n_gpus"], d["scaling"], d["config"]["global_batch"], d["dist"], list(d.get("extra", {}).keys()))
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/r02z_n2_{n}.err").read()[-1500:])
PY
