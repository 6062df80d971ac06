# Round 4, call I: block-level C entry points -- bit-identity tests, host enqueue time of the edit per calling form.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_hip_mmdit.py tests/test_hip_pipeline.py tests/test_hip_kernels.py -m gpu -x -q -s -k "block_entry or mmdit_forward or edit_matches or full_size or graph or smoke or tile_choice or hot_gemm" > gpurun_out/r04i_tests.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r04i_tests.log ); tail -3 gpurun_out/r04i_tests.log
for api in 0 1 2; do ( FK_BLOCK_API=$api timeout 300 python bench.py --steps 3 --warmup 1 --no-extra --cpu-baseline none > gpurun_out/r04i_bench_api$api.json 2> gpurun_out/r04i_bench_api$api.err; echo "bench api=$api rc=$?" ); python -c "
import json; d=json.load(open('gpurun_out/r04i_bench_api$api.json')); print('FK_BLOCK_API=$api', d['value'], d['ms_per_step'], 'host enqueue', d['host']['enqueue_ms_per_step'])"; done
python - <<'PY'
# pure host cost of enqueueing one forward (the GPU kept behind by tiny shapes is not possible at full width; instead: time the
# enqueue of ONE full-depth forward at S = 2560 from an idle queue, where nothing blocks)
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from gpt_image_edit_amd import transformer
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev)
inp = bench.make_inputs("cfg2_single_512x512_28step", dev, seed=42)
bench.run_edit(pipe, inp); torch.cuda.synchronize()
tr = pipe.transformer
B, S_txt, S_img = 1, 512, 2048
g = torch.Generator(device=dev).manual_seed(1)
hs = torch.randn(B, S_img, 64, generator=g, device=dev).to(torch.bfloat16)
enc = torch.randn(B, S_txt, 4096, generator=g, device=dev).to(torch.bfloat16)
pooled = torch.randn(B, 768, generator=g, device=dev).to(torch.bfloat16)
from gpt_image_edit_amd.helpers import _prepare_latent_image_ids as ids
img_ids = torch.cat([ids(1, 32, 32, dev, torch.bfloat16), ids(1, 32, 32, dev, torch.bfloat16)])
kw = dict(hidden_states=hs, encoder_hidden_states=enc, pooled_projections=pooled, timestep=torch.tensor([0.5], device=dev).to(torch.bfloat16),
          guidance=torch.full((1,), 3.5, device=dev), txt_ids=torch.zeros(S_txt, 3, device=dev, dtype=torch.bfloat16), img_ids=img_ids, return_dict=False)
for api in (0, 1, 2):
    transformer.BLOCK_API = api
    tr(**kw); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); c0 = time.thread_time()
        tr(**kw)
        ts.append((time.perf_counter() - t0, time.thread_time() - c0))
        torch.cuda.synchronize()
    ts.sort()
    print(f"FK_BLOCK_API={api}: host time to enqueue one full-depth forward from an idle queue: wall {ts[2][0]*1e3:.2f} ms, thread CPU {ts[2][1]*1e3:.2f} ms (median of 5; the GPU needs ~34 ms for it)")
PY
