# Round 6, call R: the last tree (fused VAE attention with 1 / 2 / 4 waves per workgroup by grid size, trimmed bench strings): VAE tests,
# the decode / encode A/B again, then the driver's bench command once more -> profiles/r06_bench_driver_cmd.json
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_vae.py tests/test_hip_pipeline.py -m gpu -q 2>&1 | tail -3 )
python - > gpurun_out/r06r_vae_attention_ab.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from gpt_image_edit_amd import vae as hv
BF = torch.bfloat16
vae = hv.HipAutoencoderKL(device="cuda", init="synthetic", seed=1)
g = torch.Generator(device="cuda").manual_seed(0)
for side in (512, 1024):
    z = torch.randn(1, 16, side // 8, side // 8, generator=g, device="cuda").to(BF)
    img = (torch.rand(1, 3, side, side, generator=g, device="cuda") * 2 - 1)
    for fused in (True, False, True, False):
        hv.FUSED_MID_ATTENTION = fused
        for name, fn in (("decode", lambda: vae.decode(z, return_dict=False)), ("encode", lambda: vae.encode(img))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); e1.synchronize()
            print(f"{side}^2 {name} mid attention {'fused      ' if fused else 'three-launch'}: {e0.elapsed_time(e1) / 5:7.3f} ms", flush=True)
PY
grep -v amdgpu.ids gpurun_out/r06r_vae_attention_ab.txt
( timeout 1700 python bench.py --steps 20 --warmup 5 > gpurun_out/r06r_bench_driver_cmd.json 2> gpurun_out/r06r_bench_driver_cmd.err; echo "bench rc=$?" ); tail -3 gpurun_out/r06r_bench_driver_cmd.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r06r_bench_driver_cmd.json'))
print('line bytes', len(json.dumps(d)))
print('cfg2', round(d['value'], 4), round(d['ms_per_step'], 1), d['ms_per_step_hip_events'])
print(json.dumps(d['roofline']['workloads']))
PY
