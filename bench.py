"""Contract benchmark: edited images / second of the FLUX-Kontext hot path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is ONE full edit of the hot path over one batch of synthetic inputs already resident in HBM:
condition-image VAE encode -> 28 x (MMDiT forward + fused Euler update) -> VAE decode
(+ for N > 1 the one real exchange of the path: an RCCL all-gather of the final packed latents).
Default workload = BASELINE.json configs[1]: single 512x512 edit, 28 steps, bf16, 1 GPU, with the
canonical synthetic shapes of SURVEY.md section 8(d): S_txt = 512, true 512^2 target and condition
(`max_area = 512^2`, `_auto_resize = False`), S = 2560, guidance 3.5, full 19 + 38 block FLUX-Kontext
transformer and the FLUX VAE with seeded random-init weights (no checkpoints offline).
The prompt encoders (Qwen2.5-VL / T5 / CLIP, reused as-is on PyTorch-ROCm) are upstream of the path:
their outputs (prompt_embeds, pooled) are inputs here.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel family (bf16 MFMA GEMM, all epilogues): algorithmic FLOPs of its
                 launches in one edit / their summed duration, measured live with HIP events on the
                 launch stream in an instrumented edit after the timed region;
  cpu_baseline : the CPU oracle (fp32 torch restatement, `oracle/`) timed on this box's host cores on
                 a bounded sample (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, MI355X_MICROARCH.md
WORKLOADS = {
    # name: (batch, height, width, cond_h, cond_w, S_txt)
    "cfg2_single_512x512_28step": (1, 512, 512, 512, 512, 512),
    "cfg2cli_512x512_cond1mp_28step": (1, 512, 512, 1024, 1024, 512),
    "cfg3_batch32_1024x1024_28step": (32, 1024, 1024, 1024, 1024, 512),
    "single_1024x1024_28step": (1, 1024, 1024, 1024, 1024, 512),
}


def build_pipeline(device, n_double=19, n_single=38):
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=n_double, num_single_layers=n_single)
    tr = HipFluxTransformer2DModel(cfg, device=device, init="synthetic", seed=0)
    vae = HipAutoencoderKL(device=device, init="synthetic", seed=1)
    return FluxKontextPipeline(tr, vae)


def make_inputs(workload, device, seed):
    B, H, W, Hc, Wc, S_txt = WORKLOADS[workload]
    g = torch.Generator(device=device).manual_seed(seed)
    cond = (torch.randint(0, 256, (B, 3, Hc, Wc), generator=g, device=device).float() / 255.0 - 0.5) / 0.5
    emb = torch.randn(B, S_txt, 4096, generator=g, device=device).to(BF)
    pooled = torch.randn(B, 768, generator=g, device=device).to(BF)
    noise = torch.randn(B, 16, H // 8, W // 8, generator=g, device=device).to(BF)
    return dict(B=B, H=H, W=W, cond=cond, emb=emb, pooled=pooled, noise=noise, S_txt=S_txt,
                S_tgt=(H // 16) * (W // 16), S_cond=(Hc // 16) * (Wc // 16))


def run_edit(pipe, inp, steps28=28):
    lat = pipe._pack_latents(inp["noise"], inp["B"], 16, inp["H"] // 8, inp["W"] // 8)
    return pipe(image=inp["cond"], prompt_embeds=inp["emb"], pooled_prompt_embeds=inp["pooled"],
                height=inp["H"], width=inp["W"], num_inference_steps=steps28, guidance_scale=3.5, latents=lat,
                output_type="pt_raw", max_area=inp["H"] * inp["W"], _auto_resize=False)


def instrumented_edit(pipe, inp):
    """Per-kernel-family HIP-event timing of one edit (events recorded on the launch stream)."""
    from gpt_image_edit_amd import ops
    rec = {"gemm": [], "attention": [], "conv": []}
    st = torch.cuda.current_stream()

    def wrap(fn, fam, flops_of):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            out = fn(*a, **k)
            e1.record(st)
            rec[fam].append((flops_of(a, k, out), e0, e1))
            return out
        return inner

    def gemm_flops(a, k, out):
        A, Wt = a[0], a[1]
        M = A.numel() // A.shape[-1]
        return 2.0 * M * Wt.shape[0] * Wt.shape[1]

    def attn_flops(a, k, out):
        B, H, S, hd = a[0].shape
        return 4.0 * B * H * S * S * hd

    def conv_flops(a, k, out):
        x, cout = a[0], a[3]
        ks = k.get("ksize", 3)
        return 2.0 * out.numel() // out.shape[-1] * cout * ks * ks * x.shape[-1]

    def grouped_flops(a, k, out):
        return sum(2.0 * (pr["a"].numel() // pr["a"].shape[-1]) * pr["w"].shape[0] * pr["w"].shape[1] for pr in a[0])

    orig = (ops.gemm, ops.gemm_grouped, ops.attention, ops.conv2d_nhwc)
    ops.gemm, ops.gemm_grouped, ops.attention, ops.conv2d_nhwc = (
        wrap(ops.gemm, "gemm", gemm_flops), wrap(ops.gemm_grouped, "gemm", grouped_flops),
        wrap(ops.attention, "attention", attn_flops), wrap(ops.conv2d_nhwc, "conv", conv_flops))
    try:
        run_edit(pipe, inp)
        torch.cuda.synchronize()
    finally:
        ops.gemm, ops.gemm_grouped, ops.attention, ops.conv2d_nhwc = orig
    out = {}
    for fam, lst in rec.items():
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in lst)
        fl = sum(f for f, _, _ in lst)
        out[fam] = dict(launches=len(lst), ms=ms, flops=fl, tflops=(fl / (ms * 1e-3) / 1e12) if ms > 0 else 0.0)
    return out


def cpu_baseline(workload, budget_blocks=(1, 1)):
    """Oracle (fp32 torch on the host) on a bounded sample: `budget_blocks` double + single MMDiT blocks
    at the workload's sequence length, extrapolated to 19 + 38 blocks x 28 steps (the MMDiT is 99.8 % of
    the edit's FLOPs; VAE and embedders are left out of the estimate, which therefore favours the CPU)."""
    from gpt_image_edit_amd import flux_spec
    from oracle import mmdit
    B, H, W, Hc, Wc, S_txt = WORKLOADS[workload]
    B = 1  # per-image cost; the CPU has no batching advantage at these sizes
    S_img = (H // 16) * (W // 16) + (Hc // 16) * (Wc // 16)
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=1, num_single_layers=1)
    shapes = {k: v for k, v in flux_spec.flux_param_shapes(cfg).items()
              if k.startswith("transformer_blocks.0.") or k.startswith("single_transformer_blocks.0.")}
    sd = flux_spec.synthetic_state(shapes, seed=5)
    g = torch.Generator().manual_seed(0)
    h = torch.randn(B, S_img, 3072, generator=g)
    c = torch.randn(B, S_txt, 3072, generator=g)
    temb = torch.randn(B, 3072, generator=g)
    from oracle.helpers import prepare_latent_image_ids
    ids = torch.cat([torch.zeros(S_txt, 3), prepare_latent_image_ids(1, S_img)])
    rope = mmdit.rope_tables(ids)
    with torch.no_grad():
        t0 = time.perf_counter()
        for _ in range(budget_blocks[0]):
            c2, h2 = mmdit.double_block(sd, "transformer_blocks.0.", h, c, temb, rope)
        t1 = time.perf_counter()
        s = torch.cat([c, h], dim=1)
        for _ in range(budget_blocks[1]):
            mmdit.single_block(sd, "single_transformer_blocks.0.", s, temb, rope)
        t2 = time.perf_counter()
    t_d, t_s = (t1 - t0) / budget_blocks[0], (t2 - t1) / budget_blocks[1]
    t_edit = 28 * (19 * t_d + 38 * t_s)
    return dict(value=1.0 / t_edit, unit="images/s", cores=torch.get_num_threads(), kind="port",
                host_cpus=os.cpu_count(),
                sample=f"fp32 oracle, {budget_blocks[0]} double + {budget_blocks[1]} single MMDiT block fwd at "
                       f"S={S_txt + S_img} (B=1): {t_d:.2f}s / {t_s:.2f}s per block, extrapolated x(19,38) blocks "
                       f"x28 steps = {t_edit:.0f}s per image (VAE + embedders excluded)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_single_512x512_28step", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one process per GPU; FK_BENCH_BACKEND=gloo (+ ranks sharing a GPU) exists only to smoke-test the N > 1
    # code path on a 1-GPU box
    backend = os.environ.get("FK_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from gpt_image_edit_amd import dp
    pipe = build_pipeline(device)
    inp = make_inputs(args.workload, device, seed=42 + rank)  # every rank edits its own shard (weak scaling)

    def one_step():
        out = run_edit(pipe, inp)
        if world > 1:
            dp.all_gather_latents(out.latents)
        return out

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out.images.float()).all(), "non-finite output image"

    B = inp["B"]
    images = B * world * args.steps
    S = inp["S_txt"] + inp["S_tgt"] + inp["S_cond"]
    result = {
        "metric": "edited images/sec, 28-step FLUX-Kontext (VAE encode + 28 x MMDiT + VAE decode)",
        "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random-init weights, random inputs)",
        "config": {"workload": args.workload, "batch_per_gpu": B, "global_batch": B * world,
                   "height": inp["H"], "width": inp["W"], "S_txt": inp["S_txt"], "S_tgt": inp["S_tgt"],
                   "S_cond": inp["S_cond"], "seq_len": S, "num_inference_steps": 28, "guidance_scale": 3.5,
                   "blocks": "19 double + 38 single", "parallelism": f"dp{world}"},
    }
    if rank == 0 and not args.no_roofline:
        fam = instrumented_edit(pipe, inp)
        gm = fam["gemm"]
        result["roofline"] = {
            "kernel": "gemm2_kernel<*> + gemm5_kernel<*> + gemm_bf16_kernel<*> (bf16 MFMA GEMM family: all MMDiT / VAE linears)", "bound": "mfma", "achieved": gm["tflops"],
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gm["tflops"] / PEAK_BF16_TFLOPS, "traffic": None,
            "launches_per_edit": gm["launches"], "ms_per_edit": gm["ms"], "algorithmic_tflop_per_edit": gm["flops"] / 1e12,
            "other_kernels": {k: {"ms_per_edit": v["ms"], "tflops": v["tflops"], "launches": v["launches"]}
                              for k, v in fam.items() if k != "gemm"},
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
