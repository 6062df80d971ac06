"""Contract benchmark: edited images / second of the FLUX-Kontext hot path on N MI355X (one process per GPU).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (WORLD_SIZE unset) re-executes itself as `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...` (`self_launch_argv`), i.e. the reference's
own launch form (`univa/eval/gedit/step1_gen_samples.py:82-92`: torchrun + `init_process_group`); under a launcher
(WORLD_SIZE set) it runs as one rank.

A "step" is ONE full edit of the hot path over one batch of synthetic inputs already resident in HBM:
condition-image VAE encode -> 28 x (MMDiT forward + fused Euler update) -> VAE decode
(+ for N > 1 the one real exchange of the path: an RCCL all-gather of the final packed latents).
Default workload = BASELINE.json configs[1]: single 512x512 edit, 28 steps, bf16, 1 GPU, with the
canonical synthetic shapes of SURVEY.md section 8(d): S_txt = 512, true 512^2 target and condition
(`max_area = 512^2`, `_auto_resize = False`), S = 2560, guidance 3.5, full 19 + 38 block FLUX-Kontext
transformer and the FLUX VAE with seeded random-init weights (no checkpoints offline).
The prompt encoders (Qwen2.5-VL / T5 / CLIP, reused as-is on PyTorch-ROCm) are upstream of the path:
their outputs (prompt_embeds, pooled) are inputs here; their cost is reported separately (`extra.prompt_encode`).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : the dominant kernel family (bf16 MFMA GEMM, all epilogues): algorithmic FLOPs of its
                 launches in one edit / their summed duration, measured live with HIP events on the
                 launch stream in an instrumented edit after the timed region; `traffic` = HBM bytes per
                 launch of the family's dominant kernel from the committed PMC passes (`TRAFFIC_FILE`: the newest
                 profiles/rNN_traffic.json);  `roofline.workloads` = the headline numbers of EVERY workload of the run;
  cpu_baseline : the CPU oracle (fp32 torch restatement, `oracle/`) timed on this box's host cores
                 (N = 1, rank 0 only): BASELINE.json configs[0] as defined -- all FOUR denoise steps at full depth
                 + VAE encode + decode executed (`--cpu-baseline cfg1`, the default; ~4.5 min of host time);
                 the 28-step figure is stated as an extrapolation of the executed steps;
  extra        : the 1024 x 1024 half of BASELINE.json's metric (`single_1024x1024_28step`, same run, after the
                 timed region: 1 warm-up + 3 timed edits per GPU, its own roofline), the CLI's ~1 MP condition
                 shape (`cfg2cli_512x512_cond1mp_28step`, S = 5632; N = 1), cfg 3 (B = 32 at 1024^2: 1 warm-up batch
                 + 3 timed batches, HIP-event median; N = 1), for N > 1 a cfg 4 slice (4 edits per GPU at 1024^2) and
                 the prompt-encode time T_prompt / T_e2e of SURVEY.md section 8(d);
  extra.cfg5_* : one stage-2 optimisation step of the denoiser (BASELINE.json configs[4]) on this GPU, samples/s;
  dist         : world size and backend as torch.distributed reports them + `rccl_ranks_seen`, the rank ids an
                 actual `all_gather_into_tensor` returned.

Other workloads (`--workload`): cfg 3 (`cfg3_batch32_1024x1024_28step`), cfg 4 (`cfg4_batch256_dp8`: 32 edits per
GPU, weak; `--scaling strong` keeps `--global-batch` fixed and shards it `items[rank::world]` like the reference's
eval generators), and `cfg4_slice4_1024x1024_28step` (a 4-per-GPU slice of cfg 4 that fits short runs).
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
PEAK_HBM_GBPS = 8000.0     # HBM3E peak (spec), MI355X_MICROARCH.md; a float4 copy reaches 6.29 TB/s
WORKLOADS = {
    # name: (batch per GPU, height, width, cond_h, cond_w, S_txt)
    "cfg2_single_512x512_28step": (1, 512, 512, 512, 512, 512),
    "cfg2cli_512x512_cond1mp_28step": (1, 512, 512, 1024, 1024, 512),
    "cfg3_batch32_1024x1024_28step": (32, 1024, 1024, 1024, 1024, 512),
    "single_1024x1024_28step": (1, 1024, 1024, 1024, 1024, 512),
    "cfg4_batch256_dp8": (32, 1024, 1024, 1024, 1024, 512),
    "cfg4_slice4_1024x1024_28step": (4, 1024, 1024, 1024, 1024, 512),
}
EXTRA_WORKLOAD = "single_1024x1024_28step"
CFG3_WORKLOAD = "cfg3_batch32_1024x1024_28step"
CLI_WORKLOAD = "cfg2cli_512x512_cond1mp_28step"
CFG4_SLICE_WORKLOAD = "cfg4_slice4_1024x1024_28step"
TRAFFIC_FILE = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"))
                     if os.path.exists(f)), os.path.join(ROOT, "profiles", "r04_traffic.json"))


def build_pipeline(device, n_double=19, n_single=38):
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.pipeline import FluxKontextPipeline
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG, num_layers=n_double, num_single_layers=n_single)
    tr = HipFluxTransformer2DModel(cfg, device=device, init="synthetic", seed=0)
    vae = HipAutoencoderKL(device=device, init="synthetic", seed=1)
    return FluxKontextPipeline(tr, vae)      # FK_GRAPH=1: the denoise loop of every edit as one hipGraph launch


def make_inputs(workload, device, seed, batch=None):
    B, H, W, Hc, Wc, S_txt = WORKLOADS[workload]
    if batch is not None:
        B = batch
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


def instrumented_edit(pipe, inp, steps28=28):
    """Per-kernel-family HIP-event timing of one edit of `steps28` denoise steps (events recorded on the launch stream; the
    per-launch rates do not depend on the step count, so the B = 32 pass runs 4 steps instead of 28).  For this pass the single
    blocks' MLP-up GEMM, which the timed edits run on a second stream beside the QKV GEMM and the attention
    (transformer.OVERLAP_MLP), is kept on the launch stream: a launch's duration can only be bracketed -- and priced
    against the roofline -- when nothing else shares the chip with it."""
    from gpt_image_edit_amd import ops, transformer
    overlap, transformer.OVERLAP_MLP = transformer.OVERLAP_MLP, False
    use_graph, pipe.use_graph = pipe.use_graph, False      # per-launch brackets need the eager loop
    block_api, transformer.BLOCK_API = transformer.BLOCK_API, 0   # ... and one host call per launch (same kernels, same bits)
    rec = {"gemm": [], "attention": [], "conv": []}
    st = torch.cuda.current_stream()

    def wrap(fn, fam, flops_of, bytes_of=None):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            out = fn(*a, **k)
            e1.record(st)
            rec[fam].append((flops_of(a, k, out), e0, e1, bytes_of(a, k, out) if bytes_of else 0.0))
            return out
        return inner

    # ALGORITHMIC bytes per launch: every operand read once, the output written once
    def attn_bytes(a, k, out):
        B, H, S, hd = a[0].shape
        return 4.0 * B * H * S * hd * 2              # Q, K, V read + O written, bf16

    def conv_bytes(a, k, out):                       # conv2d_nhwc / conv3x3_halo (x, w_packed, bias, cout, ...): input + weights + output
        res = k.get("res")
        return (a[0].numel() + a[1].numel() + out.numel() + (res.numel() if res is not None else 0)) * 2.0

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

    def halo_flops(a, k, out):   # conv3x3_halo(x, w_packed, bias, cout, ...): the GroupNorm-stats launch in front of it is not in the bracket
        return 2.0 * out.numel() // out.shape[-1] * a[3] * 9 * a[0].shape[-1]

    orig = (ops.gemm, ops.gemm_grouped, ops.attention, ops.conv2d_nhwc, ops.conv3x3_halo)
    ops.gemm, ops.gemm_grouped, ops.attention, ops.conv2d_nhwc, ops.conv3x3_halo = (
        wrap(ops.gemm, "gemm", gemm_flops), wrap(ops.gemm_grouped, "gemm", grouped_flops),
        wrap(ops.attention, "attention", attn_flops, attn_bytes), wrap(ops.conv2d_nhwc, "conv", conv_flops, conv_bytes),
        wrap(ops.conv3x3_halo, "conv", halo_flops, conv_bytes))
    try:
        run_edit(pipe, inp, steps28)
        torch.cuda.synchronize()
    finally:
        ops.gemm, ops.gemm_grouped, ops.attention, ops.conv2d_nhwc, ops.conv3x3_halo = orig
        transformer.OVERLAP_MLP = overlap
        transformer.BLOCK_API = block_api
        pipe.use_graph = use_graph
    if len(rec["attention"]) < 57 * steps28 or len(rec["gemm"]) < 3 * 57 * steps28:      # grouped launches count once
        raise RuntimeError(f"instrumented edit bracketed {len(rec['attention'])} attention / {len(rec['gemm'])} GEMM launches: "
                           "the blocks' kernels were not enqueued one host call per launch")
    out = {}
    for fam, lst in rec.items():
        ms = sum(e0.elapsed_time(e1) for _, e0, e1, _ in lst)
        fl = sum(f for f, _, _, _ in lst)
        by = sum(b for _, _, _, b in lst)
        out[fam] = dict(launches=len(lst), ms=ms, flops=fl, tflops=(fl / (ms * 1e-3) / 1e12) if ms > 0 else 0.0,
                        algorithmic_bytes=by, gbps=(by / (ms * 1e-3) / 1e9) if ms > 0 else 0.0)
    return out


def roofline_of(fam, workload):
    """`roofline` object of the bench line from the instrumented edit's family sums (+ the committed PMC traffic)."""
    gm = fam["gemm"]
    traffic, traffic_note, traffic_classes = None, None, None
    if os.path.exists(TRAFFIC_FILE):
        try:
            tr = json.load(open(TRAFFIC_FILE))
            ent = tr.get("gemm", {}).get(workload) or tr.get("gemm", {}).get("default")
            if ent:
                traffic, traffic_note = ent["hbm_bytes_per_launch"], ent.get("note")
                traffic_classes = ent.get("classes")
        except Exception as e:  # a malformed side file must not cost the bench line
            traffic_note = f"unreadable {TRAFFIC_FILE}: {e}"
    rl = {
        "kernel": "bf16 MFMA GEMM family (gemm8 / gemm9 / gemm10 / gemm_mix / gemm_bf16 kernels: every MMDiT and VAE linear)",
        "bound": "mfma", "achieved": gm["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": gm["tflops"] / PEAK_BF16_TFLOPS, "traffic": traffic,
        "launches_per_edit": gm["launches"], "ms_per_edit": gm["ms"], "algorithmic_tflop_per_edit": gm["flops"] / 1e12,
        # the other MFMA kernels of the path: rate against the MFMA peak AND achieved HBM GB/s on their algorithmic bytes
        "other_kernels": {k: {"ms_per_edit": round(v["ms"], 3), "tflops": round(v["tflops"], 1), "launches": v["launches"],
                              "frac_of_mfma_peak": round(v["tflops"] / PEAK_BF16_TFLOPS, 4),
                              "hbm_gbps_algorithmic": round(v["gbps"], 1), "frac_of_hbm_peak": round(v["gbps"] / PEAK_HBM_GBPS, 4)}
                          for k, v in fam.items() if k != "gemm"},
        "traffic_source": os.path.relpath(TRAFFIC_FILE, ROOT) + " (committed PMC passes; not counted in this run)",
    }
    # ACHIEVED bytes per second from the counters (committed PMC passes: FETCH_SIZE + WRITE_SIZE per launch over the profiled
    # duration; beyond the XCD L2s, Infinity-Cache hits included) beside the algorithmic figure of this run
    try:
        tr = json.load(open(TRAFFIC_FILE))
        counted = {"attention": [v for k, v in tr.get("attention", {}).items()], "conv": [v for k, v in tr.get("vae", {}).items()]}
        for fam_name, ents in counted.items():
            if ents and fam_name in rl["other_kernels"]:
                e = max(ents, key=lambda v: v["hbm_bytes_per_launch"])
                rl["other_kernels"][fam_name].update(hbm_gbps_counted=round(e["gbps_counted"], 1), counted_x_algorithmic=round(e["ratio"], 2),
                                                     frac_of_hbm_peak_counted=round(e["gbps_counted"] / PEAK_HBM_GBPS, 4))
    except Exception:
        pass
    if traffic_note:      # the side file's note names the kernel and the calibration: its first sentence is enough in the line
        rl["traffic_note"] = traffic_note.split(": FETCH_SIZE")[0][:110] + " (beyond the XCD L2s, Infinity-Cache hits included)"
    if traffic_classes:   # counted bytes beyond the XCD L2s per launch of EVERY GEMM launch class of the workload
        rl["traffic_x_algorithmic"] = {k: v["ratio"] for k, v in traffic_classes.items()}
    return rl


def cpu_baseline(workload, mode="full"):
    """The CPU oracle (fp32 torch on the host cores) on a bounded sample of the workload.

    mode "full": BASELINE.json configs[0] (the reference's CPU diffusers path through the cli-equivalent plumbing:
    512^2, 4 steps, fp32) with ONE of its denoise steps executed at full depth -- embedders, 19 double + 38 single
    blocks at the workload's sequence length, output head -- plus the VAE encode of the condition image and the VAE
    decode; the 19 / 38 blocks of a kind run on one shared seeded weight set (same arithmetic, 2 GB instead of 47.6 GB).  The
    4-step (cfg 1) and 28-step (cfg 2) figures are extrapolations `t_enc + n * t_step + t_dec`, labelled as such.
    mode "blocks": 1 double + 1 single block only (a few seconds), extrapolated x(19, 38) x 28.
    """
    from gpt_image_edit_amd import flux_spec
    from oracle import mmdit
    from oracle import vae as ovae
    from oracle.helpers import prepare_latent_image_ids
    _, H, W, Hc, Wc, S_txt = WORKLOADS[workload]
    B = 1  # per-image cost; the CPU has no batching advantage at these sizes
    S_img = (H // 16) * (W // 16) + (Hc // 16) * (Wc // 16)
    full = flux_spec.flux_param_shapes(flux_spec.FLUX_KONTEXT_CONFIG)
    g = torch.Generator().manual_seed(0)
    ids = torch.cat([torch.zeros(S_txt, 3), prepare_latent_image_ids(1, S_img)])
    rope = mmdit.rope_tables(ids)
    common = dict(unit="images/s", cores=torch.get_num_threads(), kind="port", host_cpus=os.cpu_count())

    def block_state(prefix, seed=5):
        return flux_spec.synthetic_state({k: v for k, v in full.items() if k.startswith(prefix)}, seed=seed)

    with torch.no_grad():
        if mode == "blocks":
            h = torch.randn(B, S_img, 3072, generator=g)
            c = torch.randn(B, S_txt, 3072, generator=g)
            temb = torch.randn(B, 3072, generator=g)
            sd = block_state("transformer_blocks.0.")
            sd.update(block_state("single_transformer_blocks.0."))
            t0 = time.perf_counter()
            mmdit.double_block(sd, "transformer_blocks.0.", h, c, temb, rope)
            t1 = time.perf_counter()
            mmdit.single_block(sd, "single_transformer_blocks.0.", torch.cat([c, h], dim=1), temb, rope)
            t2 = time.perf_counter()
            t_d, t_s = t1 - t0, t2 - t1
            t_edit = 28 * (19 * t_d + 38 * t_s)
            return dict(common, value=1.0 / t_edit,
                        sample=f"fp32 oracle, 1 double + 1 single MMDiT block fwd at S={S_txt + S_img} (B=1): {t_d:.2f}s / "
                               f"{t_s:.2f}s per block, EXTRAPOLATED x(19,38) blocks x28 steps = {t_edit:.0f}s per image "
                               f"(VAE + embedders excluded)")
        # ---- one full-depth denoise step ------------------------------------------------------------------------
        # (the blocks of a kind share ONE seeded weight set, generated outside the timed span: the seeded generator
        #  is single-threaded and would cost more than the arithmetic; the data dependence through all 57 blocks is kept)
        tokens = torch.randn(B, S_img, 64, generator=g)
        enc = torch.randn(B, S_txt, 4096, generator=g)
        pooled = torch.randn(B, 768, generator=g)
        sd = block_state("x_embedder.")
        for pre in ("context_embedder.", "time_text_embed.", "norm_out.", "proj_out.", "transformer_blocks.0.",
                    "single_transformer_blocks.0."):
            sd.update(block_state(pre))
        # a SECOND, distinct weight set for one double + one single block (block index 1): run in the middle of the
        # stack and timed by itself, so that the line shows whether sharing one set flatters the CPU (cache-warm weights)
        for pre in ("transformer_blocks.1.", "single_transformer_blocks.1."):
            sd.update(block_state(pre, seed=6))
        n_steps = 4 if mode == "cfg1" else 1
        t_steps, t_shared, t_distinct = [], {}, {}

        def timed(fn, store, key):
            t = time.perf_counter()
            out = fn()
            store.setdefault(key, []).append(time.perf_counter() - t)
            return out
        for step in range(n_steps):
            t0 = time.perf_counter()
            h = mmdit.linear(sd, "x_embedder", tokens)
            c = mmdit.linear(sd, "context_embedder", enc)
            temb = mmdit.time_text_embed(sd, torch.full((B,), 500.0), torch.full((B,), 3500.0), pooled)
            for i in range(19):
                if i == 9:
                    c, h = timed(lambda: mmdit.double_block(sd, "transformer_blocks.1.", h, c, temb, rope), t_distinct, "double")
                else:
                    c, h = timed(lambda: mmdit.double_block(sd, "transformer_blocks.0.", h, c, temb, rope), t_shared, "double")
            s = torch.cat([c, h], dim=1)
            for i in range(38):
                if i == 19:
                    s = timed(lambda: mmdit.single_block(sd, "single_transformer_blocks.1.", s, temb, rope), t_distinct, "single")
                else:
                    s = timed(lambda: mmdit.single_block(sd, "single_transformer_blocks.0.", s, temb, rope), t_shared, "single")
            e = mmdit.linear(sd, "norm_out.linear", torch.nn.functional.silu(temb))
            scale, shift = e.chunk(2, dim=1)
            hh = mmdit.layer_norm(s[:, S_txt:]) * (1 + scale)[:, None, :] + shift[:, None, :]
            v = mmdit.linear(sd, "proj_out", hh)
            t_steps.append(time.perf_counter() - t0)
        t_step = sum(t_steps) / len(t_steps)
        med = lambda xs: sorted(xs)[len(xs) // 2]  # noqa: E731
        weights_note = {k: dict(shared_set_median_s=med(t_shared[k]), distinct_set_s=med(t_distinct[k])) for k in ("double", "single")}
        # ---- VAE either side (weights generated outside the timed spans: "model load excluded") ------------------
        sd_v = flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=1)
        img = torch.rand(B, 3, Hc, Wc, generator=g) * 2 - 1
        t0 = time.perf_counter()
        ovae.encode_for_pipeline(sd_v, img)
        t_enc = time.perf_counter() - t0
        z = torch.randn(B, 16, H // 8, W // 8, generator=g)
        t0 = time.perf_counter()
        ovae.decode_for_pipeline(sd_v, z)
        t_dec = time.perf_counter() - t0
    assert torch.isfinite(v).all()
    t4, t28 = t_enc + 4 * t_step + t_dec, t_enc + 28 * t_step + t_dec
    flops_step = mmdit.flops_forward(S_txt + S_img)
    if n_steps == 4:   # BASELINE.json configs[0] as defined: all four steps executed
        t4 = t_enc + sum(t_steps) + t_dec
        what = (f"cfg 1 as defined: 4 denoise steps executed ({', '.join(f'{t:.1f}' for t in t_steps)} s) + VAE encode {t_enc:.1f}s + decode "
                f"{t_dec:.1f}s = {t4:.0f}s per image; `value` = 28-step EXTRAPOLATION 1 / (enc + 28 x mean step + dec) = 1 / {t28:.0f}s")
    else:
        what = (f"ONE full-depth denoise step {t_step:.1f}s + VAE encode {t_enc:.1f}s + decode {t_dec:.1f}s executed; cfg 1 (1 / {t4:.0f}s) and "
                f"`value` (28 steps, 1 / {t28:.0f}s) are EXTRAPOLATIONS of that step")
    return dict(common, value=1.0 / t28, cfg1_4step_images_per_s=1.0 / t4, steps_executed=n_steps,
                steps_not_executed=[] if n_steps == 4 else [2, 3, 4], torch_num_threads=torch.get_num_threads(),
                t_step_s=t_step, t_steps_s=[round(t, 2) for t in t_steps], t_vae_encode_s=t_enc, t_vae_decode_s=t_dec,
                gflops_step=flops_step / t_step / 1e9, block_time_shared_vs_distinct_weights=weights_note,
                sample=f"fp32 oracle (oracle/, CPU restatement) at S={S_txt + S_img}, B=1: {what}")


def prompt_encode_time(device, batch=1):
    """T_prompt of SURVEY.md section 8(d): the Qwen2.5-VL-7B forward(s) + projector the reference cli runs per edit
    (`univa/serve/cli.py:199-234`: two VLM forwards, the second with the condition image), random-init weights,
    reused as-is from `transformers` on PyTorch-ROCm -- NOT part of `value`."""
    from gpt_image_edit_amd.qwen_adaptor import bench_prompt_encode
    return bench_prompt_encode(device, batch=batch)


def train_step_bench(device, steps=3, warmup=2, world=1, e2e=True):
    """BASELINE.json configs[4] on this rank's GPU: one stage-2 optimisation step of the denoiser at 1024^2, batch 1 per
    GPU (S_txt = 256 projected VLM tokens + 256 T5 prefix tokens, + 4096 target + 4096 condition tokens), the parameters
    the reference un-freezes (`only_tune_image_branch` subset of the MMDiT + the denoise_projector), activations stored
    or one checkpoint per block (`auto`), AdamW on ZeRO-2-sharded fp32 state (with world > 1:
    fp32 gradient reduce-scatter + bf16 parameter all-gather over RCCL).  Synthetic latents / embeddings / weights.
    Two warm-up steps: the first builds workspaces and transposed weights, the second is the first to start from updated
    parameters (what every later step does)."""
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.projector import HipDenoiseProjector
    from gpt_image_edit_amd.train_step import DenoiserTrainStep
    from gpt_image_edit_amd.transformer import HipFluxTransformer2DModel

    def flops_forward(S, D=3072, n_double=19, n_single=38):   # BASELINE.md section 2 (embedders omitted)
        nb = n_double + n_single
        return nb * 24 * D * D * S + nb * 4 * S * S * D + 2 * (n_double * 12 + n_single * 3 + 2) * D * D
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(device)
    model = HipFluxTransformer2DModel(dict(flux_spec.FLUX_KONTEXT_CONFIG), device=device, init="synthetic", seed=0)
    projector = HipDenoiseProjector(device=device, init="synthetic", seed=1)
    ts = DenoiserTrainStep(model, sharded=True, projector=projector, keep_grads=False)   # gradients live in the ZeRO buckets only
    g = torch.Generator(device=device).manual_seed(7)
    B, h, w, L_vlm, L_t5 = 1, 128, 128, 256, 256
    S_txt = L_vlm + L_t5
    batch = dict(model_input=torch.randn(B, 16, h, w, generator=g, device=device),
                 cond_latents=torch.randn(B, 16, h, w, generator=g, device=device),
                 noise=torch.randn(B, 16, h, w, generator=g, device=device),
                 sigmas=torch.rand(B, generator=g, device=device) * 0.8 + 0.1,
                 vlm_hidden=torch.randn(B, L_vlm, 3584, generator=g, device=device).to(BF),
                 prefix_prompt_embeds=torch.randn(B, L_t5, 4096, generator=g, device=device).to(BF),
                 pooled=torch.randn(B, 768, generator=g, device=device).to(BF))
    for _ in range(warmup):
        out = ts.step(**batch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0, c0 = time.perf_counter(), time.thread_time()
    for _ in range(steps):
        out = ts.step(**batch)
    t_host = (time.perf_counter() - t0) / steps      # the host is done enqueueing; the GPU may still be working
    t_cpu = (time.thread_time() - c0) / steps       # CPU seconds of the enqueueing thread per step: an UPPER bound on the pure host work
    torch.cuda.synchronize()                         # (it still counts whatever the runtime spins while the launch queue is full)
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out["loss"]).all()
    S = S_txt + 2 * (h // 2) * (w // 2)
    n_train = sum(ts._param(k).numel() for k in ts.trainable_names())
    fwd = flops_forward(S)
    e2e_res = None
    if e2e and world == 1:
        try:
            e2e_res = train_step_e2e(device, ts, batch, L_vlm)
        except Exception as e:   # the extra must never cost the step's own number
            e2e_res = {"error": f"{type(e).__name__}: {e}"}
    return {"value": B * world / dt, "unit": "samples/s", "ms_per_step": dt * 1e3, "n_gpus": world, "steps": steps, "warmup": warmup,
            "loss": float(out["loss"].item()), "trainable_params": n_train, "seq_len": S,
            "peak_memory_gb": torch.cuda.max_memory_allocated(device) / 1e9, "host_enqueue_ms_per_step": t_host * 1e3,
            "host_work_ms_per_step": t_cpu * 1e3,
            "T_step_e2e": e2e_res,
            "zero2_buckets": len(ts.opt.layout.buckets),
            "forward_tflop": fwd / 1e12,
            "model_tflops_3x_forward": 3 * fwd / dt / 1e12,
            "frac_of_mfma_peak_3x_forward": 3 * fwd / dt / 1e12 / PEAK_BF16_TFLOPS,
            "what": "train_denoiser.py:829-1181 on the ZeRO-2 layout; host_work = thread CPU time = runtime spin on the full queue "
                    "(block backwards: 13 ms of host work, profiles/r06_bwd_block_api.txt)"}


def train_step_e2e(device, ts, batch, L_vlm, steps=3):
    """The rest of the reference's optimisation step around the core step (train_denoiser.py:887-1093): VAE encode of the
    1024^2 target and of the 1024^2 condition image (`.latent_dist.sample()`, shift / scale) and the frozen Qwen2.5-VL
    forward that produces the hidden states the denoise_projector reads -- then the core step on those tensors.
    The VAE encodes run on HipAutoencoderKL's fp32-class encoder from an fp32 checkpoint (the reference's stage-2 config
    sets `vae_fp32: true`; three-term split-bf16 products, fp32 activations; the bf16 encoder is timed beside it).
    Caveats, stated in the result: the VLM is the stock transformers model with random-init 7B weights reused as-is on
    PyTorch-ROCm, T5 prefix embeddings are given."""
    from gpt_image_edit_amd import flux_spec
    from gpt_image_edit_amd.qwen_adaptor import UnivaQwen2p5VL, build_vlm, qwen25vl_config, synthetic_turn
    from gpt_image_edit_amd.vae import HipAutoencoderKL
    vae = HipAutoencoderKL(device=device)
    vae.load_fp32_state_dict(flux_spec.synthetic_state(flux_spec.vae_param_shapes(), seed=2, device=device, dtype=torch.float32))
    cfg = qwen25vl_config("7b")
    front = UnivaQwen2p5VL(build_vlm(cfg, device), lambda hidden: hidden)      # the projector trains inside the step: hand over the hidden states
    turn = synthetic_turn(cfg, device)
    g = torch.Generator(device=device).manual_seed(11)
    B, _, h, w = batch["model_input"].shape
    target = torch.rand(B, 3, 8 * h, 8 * w, generator=g, device=device) * 2 - 1
    cond = torch.rand(B, 3, 8 * h, 8 * w, generator=g, device=device) * 2 - 1
    vc = vae.config
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def one():
        ev[0].record()
        z_t = (vae.encode(target, fp32=True).latent_dist.sample() - vc.shift_factor) * vc.scaling_factor   # :897-903
        z_c = (vae.encode(cond, fp32=True).latent_dist.sample() - vc.shift_factor) * vc.scaling_factor     # :887-890 + kontext scaling
        ev[1].record()
        hidden = front(**turn, output_type="denoise_embeds")[:, :L_vlm]                                    # :1073-1093 (frozen VLM)
        ev[2].record()
        out = ts.step(**dict(batch, model_input=z_t, cond_latents=z_c, vlm_hidden=hidden.to(BF)))
        ev[3].record()
        return out
    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out["loss"]).all()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    vae.encode(target)
    e0.record()
    vae.encode(target), vae.encode(cond)
    e1.record()
    torch.cuda.synchronize()
    return {"ms_per_step": dt * 1e3, "samples_per_s": B / dt, "steps": steps,
            "last_step_ms": {"vae_encode_x2": ev[0].elapsed_time(ev[1]), "vlm_forward": ev[1].elapsed_time(ev[2]),
                             "core_step": ev[2].elapsed_time(ev[3])},
            "vae_encode": "fp32-class (vae_fp32: true)",
            "vae_encode_x2_bf16_ms": e0.elapsed_time(e1),
            "peak_memory_gb": torch.cuda.max_memory_allocated(device) / 1e9,
            "caveats": "random-init Qwen2.5-VL-7B, T5 prefix given"}


def timed_edits(pipe, inp, steps, warmup, world, device, backend):
    """The contract's timed region (barrier + synchronize either side of EXACTLY `steps` edits, wall clock, MAX over ranks)
    and, beside it, every step's duration from HIP events on the launch stream (median of >= 3)."""
    from gpt_image_edit_amd import dp

    def one_step():
        out = run_edit(pipe, inp)
        if world > 1:
            dp.all_gather_latents(out.latents)
        return out

    for _ in range(warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    st = torch.cuda.current_stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i].record(st)
        out = one_step()
    ev[steps].record(st)
    timed_edits.host_enqueue_s = (time.perf_counter() - t0) / steps   # the host is done enqueueing; the GPU may still be working
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_step = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    timed_edits.hip_event_ms = {"median": per_step[len(per_step) // 2], "min": per_step[0], "max": per_step[-1],
                                "mean": sum(per_step) / len(per_step), "n": steps}
    if world > 1:
        tt = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out.images.float()).all(), "non-finite output image"
    return elapsed


def _compact(x):
    """Floats to 6 significant digits (the line must stay well under the 8 KB the driver keeps of stdout's tail)."""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    if isinstance(x, dict):
        return {k: _compact(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_compact(v) for v in x]
    return x


def _slim_roofline(rl):
    """The roofline of an `extra` workload without the strings the main one already carries; the other MFMA kernels as their rates
    only (the line must stay under the 8 KB the driver keeps of stdout's tail)."""
    out = {k: v for k, v in rl.items() if k not in ("kernel", "traffic_note", "traffic_source", "traffic_x_algorithmic", "bound", "peak",
                                                    "unit", "other_kernels")}
    out["other_kernels"] = {k: {"ms_per_edit": v["ms_per_edit"], "tflops": v["tflops"], "hbm_gbps_algorithmic": v["hbm_gbps_algorithmic"]}
                            for k, v in rl.get("other_kernels", {}).items()}
    return out


def workload_summary(value, ms_per_step, rl=None, **more):
    """One entry of `roofline.workloads`: the driver's parsed record keeps `roofline` whole, so every workload's headline
    numbers (throughput, GEMM-family and attention rates against the MFMA peak) ride inside it."""
    d = {"value": round(value, 5), "ms_per_step": round(ms_per_step, 2)}
    if rl:
        att = rl.get("other_kernels", {}).get("attention", {})
        d.update(gemm_tflops=round(rl["achieved"], 1), gemm_frac=round(rl["frac"], 4),
                 attention_tflops=att.get("tflops"), attention_frac=att.get("frac_of_mfma_peak"))
    d.update(more)
    return d


def self_launch_argv(n_gpus, argv, port, script=None):
    """The command `python bench.py --gpus N ...` turns into when no launcher started it: one process per GPU under
    torch.distributed.run on this node (127.0.0.1 rendezvous: the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ranks_seen(world, device, backend):
    """Rank ids as ONE all_gather_into_tensor returns them (nccl = RCCL: on the GPUs; gloo smoke: host tensors)."""
    dev = device if backend == "nccl" else "cpu"
    mine = torch.tensor([dist.get_rank()], device=dev, dtype=torch.int32)
    out = torch.full((world,), -1, device=dev, dtype=torch.int32)
    dist.all_gather_into_tensor(out, mine)
    return [int(v) for v in out.cpu()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_single_512x512_28step", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch per GPU; strong: --global-batch fixed, sharded items[rank::world]")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: total edits per step (default 8 x batch)")
    ap.add_argument("--cpu-baseline", default="cfg1", choices=["full", "cfg1", "blocks", "none"],
                    help="cfg1 (default): BASELINE.json configs[0] as defined, all four steps executed (~4 min of host time); "
                         "full: one full-depth CPU step + VAE; blocks: 1 + 1 blocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extras (1024^2 edit, cfg 3 batch, prompt encode, cfg 5 train step)")
    args = ap.parse_args()
    if args.no_cpu_baseline:
        args.cpu_baseline = "none"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started the way the 1-GPU bench is: become the launcher (step1_gen_samples.py:82-92 is torchrun + init_process_group)
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        raise SystemExit(subprocess.call(self_launch_argv(args.gpus, sys.argv[1:], _free_port()), env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
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
    B_w = WORKLOADS[args.workload][0]
    if args.scaling == "strong":
        G = args.global_batch or 8 * B_w
        mine = dp.shard_indices(G, rank, world)      # the reference's inference_list[rank::world_size]
        if not mine:
            raise SystemExit(f"--global-batch {G} leaves rank {rank} of {world} without work")
        batch, global_batch = len(mine), G
    else:
        batch, global_batch = B_w, B_w * world
    inp = make_inputs(args.workload, device, seed=42 + rank, batch=batch)  # every rank edits its own shard

    elapsed = timed_edits(pipe, inp, args.steps, args.warmup, world, device, backend)
    images = global_batch * args.steps
    S = inp["S_txt"] + inp["S_tgt"] + inp["S_cond"]
    result = {
        "metric": "edited images/sec, 28-step FLUX-Kontext (VAE encode + 28 x MMDiT + VAE decode)",
        "value": images / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic (seeded random-init weights, random inputs)",
        "config": {"workload": args.workload, "batch_per_gpu": batch, "global_batch": global_batch,
                   "height": inp["H"], "width": inp["W"], "S_txt": inp["S_txt"], "S_tgt": inp["S_tgt"],
                   "S_cond": inp["S_cond"], "seq_len": S, "num_inference_steps": 28, "guidance_scale": 3.5,
                   "blocks": "19 double + 38 single", "parallelism": f"dp{world}"},
        "ms_per_step_hip_events": timed_edits.hip_event_ms,
        "host": {"enqueue_ms_per_step": timed_edits.host_enqueue_s * 1e3, "denoise_loop_as_hipgraph": bool(pipe.use_graph),
                 # which of the bit-identical kernel forms this run used (defaults unless the environment says otherwise)
                 "attention_forward": os.environ.get("FK_ATTN_KERNEL", "4") + " waves",
                 "gemm_mfma": "16x16x32" if getattr(__import__("gpt_image_edit_amd.ops", fromlist=["LAUNCH"]).LAUNCH, "gemm_mfma", 0) in (0, 16) else "32x32x16",
                 "train_bwd_k_major": int(os.environ.get("FK_BWD_K_MAJOR", "2"))},
        "dist": {"world_size": dist.get_world_size() if world > 1 else 1,
                 "rccl_ranks_seen": ranks_seen(world, device, backend) if world > 1 else [0],
                 "backend": (dist.get_backend() + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else "none (single process)",
                 "collective": "one all_gather_into_tensor of the packed final latents per step" if world > 1 else None},
    }
    summaries = {}
    if rank == 0 and not args.no_roofline:
        i_steps = 28 if batch <= 4 else 4          # a B = 32 pass brackets 4 denoise steps (same launches, same rates)
        result["roofline"] = roofline_of(instrumented_edit(pipe, inp, i_steps), args.workload)
        result["roofline"]["instrumented_denoise_steps"] = i_steps
        summaries[args.workload] = workload_summary(result["value"], result["ms_per_step"], result["roofline"], unit="images/s")

    # ---- the other workloads of BASELINE.json in the same run ----------------------------------------------------------
    extra = {}

    def extra_workload(name, k, warm, seed, i_steps=28, all_ranks=True):
        """`k` timed steps (after `warm`) of workload `name` on every rank (weak scaling) or on this process only; own
        roofline from an instrumented pass of `i_steps` denoise steps; headline numbers into `roofline.workloads`."""
        w = world if all_ranks else 1
        inp_x = make_inputs(name, device, seed=seed + rank)
        torch.cuda.reset_peak_memory_stats(device)
        el = timed_edits(pipe, inp_x, k, warm, w, device, backend)
        Bx = inp_x["B"]
        ex = {"value": w * Bx * k / el, "unit": "images/s", "n_gpus": w, "steps": k, "warmup": warm,
              "ms_per_step": el / k * 1e3, "scaling": "weak", "ms_per_step_hip_events": timed_edits.hip_event_ms,
              "config": {"batch_per_gpu": Bx, "height": inp_x["H"], "width": inp_x["W"],
                         "seq_len": inp_x["S_txt"] + inp_x["S_tgt"] + inp_x["S_cond"]},
              "peak_memory_gb": torch.cuda.max_memory_allocated(device) / 1e9}
        if rank == 0 and not args.no_roofline:
            rl = roofline_of(instrumented_edit(pipe, inp_x, i_steps), name)
            more = {"instrumented_denoise_steps": i_steps} if i_steps != 28 else {}
            summaries[name] = workload_summary(ex["value"], ex["ms_per_step"], rl, unit="images/s", steps=k, warmup=warm,
                                               ms_median_hip_events=round(ex["ms_per_step_hip_events"]["median"], 2), **more)
            ex["roofline"] = dict(_slim_roofline(rl), **more)
        extra[name] = ex

    single = WORKLOADS[args.workload][0] == 1 and not args.no_extra
    if single and args.workload != EXTRA_WORKLOAD:
        # the 1024^2 half of BASELINE.json's metric (every rank: weak scaling at 1024^2)
        extra_workload(EXTRA_WORKLOAD, 3, 1, seed=142)
    if single and world > 1 and os.environ.get("FK_BENCH_CFG4", "1") != "0":
        # BASELINE.json configs[3] (batch 256 over 8 GPUs) as a 4-per-GPU slice: the N > 1 curve at 1024^2 with B > 1
        try:
            extra_workload(CFG4_SLICE_WORKLOAD, 2, 1, seed=342)
        except Exception as e:
            extra[CFG4_SLICE_WORKLOAD] = {"error": f"{type(e).__name__}: {e}"}
    if single and rank == 0 and world == 1 and args.workload != CLI_WORKLOAD:
        # the shape the reference CLI really runs for a 512^2 request: condition image resized to ~1 MP (flux_pipeline.py:960-972)
        try:
            extra_workload(CLI_WORKLOAD, 3, 1, seed=442)
        except Exception as e:
            extra[CLI_WORKLOAD] = {"error": f"{type(e).__name__}: {e}"}
    if single and rank == 0 and world == 1 and os.environ.get("FK_BENCH_CFG3", "1") != "0":
        # BASELINE.json configs[2] (B = 32 at 1024^2: one GPU's share of the reference's batch runs) by SURVEY.md section
        # 8(d)'s protocol: 1 warm-up batch + 3 timed batches (HIP-event median beside the wall-clock mean; a batch takes
        # ~2 min), then a 4-step instrumented pass of the same batch for its roofline (M = 278 528 rows per GEMM).
        # FK_BENCH_CFG3_STEPS=k,w shortens it for builder runs.
        try:
            k3, w3 = (int(v) for v in os.environ.get("FK_BENCH_CFG3_STEPS", "3,1").split(","))
            extra_workload(CFG3_WORKLOAD, k3, w3, seed=242, i_steps=4)
        except Exception as e:
            extra[CFG3_WORKLOAD] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            del pipe
            torch.cuda.empty_cache()
            pe = prompt_encode_time(device)
            t_edit = elapsed / args.steps
            pe["T_s"] = t_edit
            pe["T_e2e_s"] = pe["T_prompt_s"] + t_edit
            extra["prompt_encode"] = pe
        except Exception as e:  # the extras must never cost the contract line
            extra["prompt_encode"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            torch.cuda.empty_cache()
            t5 = extra["cfg5_train_step_1024x1024_bs1"] = train_step_bench(device)
            summaries["cfg5_train_step_1024x1024_bs1"] = workload_summary(
                t5["value"], t5["ms_per_step"], unit="samples/s", frac_of_mfma_peak_3x_forward=round(t5["frac_of_mfma_peak_3x_forward"], 4),
                host_work_ms_per_step=round(t5["host_work_ms_per_step"], 1),
                T_step_e2e_ms=(t5.get("T_step_e2e") or {}).get("ms_per_step"))
            torch.cuda.empty_cache()
        except Exception as e:
            extra["cfg5_train_step_1024x1024_bs1"] = {"error": f"{type(e).__name__}: {e}"}
    if extra:
        result["extra"] = extra
    if "roofline" in result and summaries:
        result["roofline"]["workloads"] = summaries
    if rank == 0 and world == 1 and args.cpu_baseline != "none":
        result["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_baseline)
    if rank == 0:
        print(json.dumps(_compact(result)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
