"""FluxKontextPipeline -- MI355X-native counterpart of the reference's pipeline driver.

Mirrors the call surface and semantics of ``univa/utils/flux_pipeline.py::FluxKontextPipeline.__call__``
(:732-1138) for the embeds-driven path every reference caller uses (``univa/serve/cli.py:239-248``,
``univa/eval/gedit/step1_gen_samples.py:194-203``, ``train_denoiser.py:1587-1600``):
same argument names and defaults, the ``max_area`` / preferred-resolution quirks, the same
latent packing, ids, sigma schedule, 28-step Euler loop and VAE decode -- but the loop body is

    transformer (HIP MMDiT)  ->  fused slice + Euler update (one HIP kernel, in place)

over ONE persistent token buffer [B, S_tgt + S_cond, 64] (target tokens first, condition tokens after:
the reference's per-step ``torch.cat([latents, image_latents], dim=1)`` disappears), with all per-step
scalars prepared before the loop so the 28 steps enqueue without a host sync.

Pixels either side of the path: ``image_processor.VaeImageProcessor`` restates the
diffusers helper the reference pipeline calls; uint8 pixels (PIL / numpy / uint8 tensors) take two fused HIP kernels
(normalise + nearest resize + cast in front of the VAE encoder, clamp + quantise behind the decoder).
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import helpers, image_processor, ops
from .scheduler import FlowMatchEulerDiscreteScheduler

BF16 = torch.bfloat16


class FluxPipelineOutput(SimpleNamespace):
    pass


class FluxKontextPipeline:
    def __init__(self, transformer, vae, scheduler=None, text_encoder=None, tokenizer=None,
                 text_encoder_2=None, tokenizer_2=None, use_graph=None):
        """use_graph (default: FK_GRAPH=1 in the environment): run the conditioning pass + the whole denoise loop of an
        edit as ONE hipGraph launch (captured once per call shape, replayed afterwards; bit-identical to the eager loop).
        The loop allocates nothing and never synchronises, so the capture is a plain stream capture; what it buys is the
        host: ~5 400 ctypes launches per edit become one graph launch -- insurance for 8 ranks sharing one host."""
        self.use_graph = (os.environ.get("FK_GRAPH", "0") == "1") if use_graph is None else bool(use_graph)
        self._loop_graph = None       # (key, static tensors, torch.cuda.CUDAGraph) of the latest call shape
        self.transformer = transformer
        self.vae = vae
        self.scheduler = scheduler or FlowMatchEulerDiscreteScheduler()
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        self.text_encoder_2, self.tokenizer_2 = text_encoder_2, tokenizer_2
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        self.latent_channels = vae.config.latent_channels
        self.default_sample_size = 128
        self.tokenizer_max_length = tokenizer.model_max_length if tokenizer is not None else 77   # :253-255
        self.image_processor = image_processor.VaeImageProcessor(vae_scale_factor=self.vae_scale_factor * 2)
        self._interrupt = False
        self._guidance_scale = None
        self._num_timesteps = 0

    # static helpers the training script reaches for directly (train_denoiser.py:925,1009,1021,1098)
    _pack_latents = staticmethod(helpers._pack_latents)
    _unpack_latents = staticmethod(helpers._unpack_latents)
    _prepare_latent_image_ids = staticmethod(helpers._prepare_latent_image_ids)

    def to(self, device):
        self.transformer.to(device)
        self.vae.to(device)
        return self

    @property
    def device(self):
        return self.transformer.device

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    def enable_vae_slicing(self): self.vae.enable_slicing()
    def disable_vae_slicing(self): self.vae.disable_slicing()
    def enable_vae_tiling(self): self.vae.enable_tiling()
    def disable_vae_tiling(self): self.vae.disable_tiling()
    def maybe_free_model_hooks(self): pass

    # ---- input validation: the cases that can still occur with embeds-only input ----------------------
    def check_inputs(self, height, width, prompt_embeds, pooled_prompt_embeds, max_sequence_length=512):
        m = self.vae_scale_factor * 2
        if height % m != 0 or width % m != 0:
            print(f"`height` and `width` have to be divisible by {m} but are {height} and {width}. "
                  f"Dimensions will be resized accordingly")
        if prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and "
                             "`prompt_embeds` undefined.")
        if pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed. Make "
                             "sure to generate `pooled_prompt_embeds` from the same text encoder that was used to "
                             "generate `prompt_embeds`.")
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f"`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}")

    # flux_pipeline.py:266-438 (the encoders are transformers' CLIPTextModel / T5EncoderModel, reused as they are)
    def _get_t5_prompt_embeds(self, prompt, num_images_per_prompt=1, max_sequence_length=512, device=None, dtype=None):
        if self.text_encoder_2 is None or self.tokenizer_2 is None:
            raise ValueError("string prompts need `text_encoder_2` / `tokenizer_2` (T5); pass `prompt_embeds` instead")
        device = device or self.device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        ids = self.tokenizer_2(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                               return_length=False, return_overflowing_tokens=False, return_tensors="pt").input_ids
        untruncated = self.tokenizer_2(prompt, padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] >= ids.shape[-1] and not torch.equal(ids, untruncated):
            removed = self.tokenizer_2.batch_decode(untruncated[:, self.tokenizer_max_length - 1:-1])
            print(f"The following part of your input was truncated because `max_sequence_length` is set to "
                  f" {max_sequence_length} tokens: {removed}")
        enc_device = next(self.text_encoder_2.parameters()).device
        embeds = self.text_encoder_2(ids.to(enc_device), output_hidden_states=False)[0]
        embeds = embeds.to(dtype=self.text_encoder_2.dtype, device=device)
        n, seq_len, _ = embeds.shape
        return embeds.repeat(1, num_images_per_prompt, 1).view(n * num_images_per_prompt, seq_len, -1)

    def _get_clip_prompt_embeds(self, prompt, num_images_per_prompt=1, device=None):
        if self.text_encoder is None or self.tokenizer is None:
            raise ValueError("string prompts need `text_encoder` / `tokenizer` (CLIP); pass `pooled_prompt_embeds` "
                             "instead")
        device = device or self.device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        ids = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer_max_length, truncation=True,
                             return_overflowing_tokens=False, return_length=False, return_tensors="pt").input_ids
        untruncated = self.tokenizer(prompt, padding="longest", return_tensors="pt").input_ids
        if untruncated.shape[-1] >= ids.shape[-1] and not torch.equal(ids, untruncated):
            removed = self.tokenizer.batch_decode(untruncated[:, self.tokenizer_max_length - 1:-1])
            print(f"The following part of your input was truncated because CLIP can only handle sequences up to"
                  f" {self.tokenizer_max_length} tokens: {removed}")
        enc_device = next(self.text_encoder.parameters()).device
        pooled = self.text_encoder(ids.to(enc_device), output_hidden_states=False).pooler_output
        pooled = pooled.to(dtype=self.text_encoder.dtype, device=device)
        n = pooled.shape[0]
        return pooled.repeat(1, num_images_per_prompt).view(n * num_images_per_prompt, -1)

    def encode_prompt(self, prompt, prompt_2=None, device=None, num_images_per_prompt=1, prompt_embeds=None,
                      pooled_prompt_embeds=None, max_sequence_length=512, lora_scale=None):
        """(prompt_embeds, pooled_prompt_embeds, text_ids): CLIP pooled output of ``prompt``, T5 last hidden state of
        ``prompt_2 or prompt``; given embeddings pass through untouched (flux_pipeline.py:361-438)."""
        device = device or self.device
        if prompt_embeds is None:
            prompt = [prompt] if isinstance(prompt, str) else prompt
            prompt_2 = prompt_2 or prompt
            prompt_2 = [prompt_2] if isinstance(prompt_2, str) else prompt_2
            pooled_prompt_embeds = self._get_clip_prompt_embeds(prompt, num_images_per_prompt, device)
            prompt_embeds = self._get_t5_prompt_embeds(prompt_2, num_images_per_prompt, max_sequence_length, device)
        dtype = self.text_encoder.dtype if self.text_encoder is not None else self.transformer.dtype
        text_ids = torch.zeros(prompt_embeds.shape[1], 3).to(device=device, dtype=dtype)
        return prompt_embeds, pooled_prompt_embeds, text_ids

    def _encode_vae_image(self, image, nhwc=False):
        """(mode(vae.encode(image)) - shift) * scale, the affine fused into the layout kernel (:600-613)."""
        cfg = self.vae.config
        return self.vae.encode(image, post_add=-cfg.shift_factor, post_mul=cfg.scaling_factor,
                               nhwc=nhwc).latent_dist.mode()

    def prepare_latents(self, image, batch_size, num_channels_latents, height, width, dtype, device,
                        generator=None, latents=None):
        """Same contract as flux_pipeline.py:648-708 -> (latents, image_latents, latent_ids, image_ids)."""
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}. Make sure the batch size matches the length "
                             f"of the generators.")
        height = 2 * (int(height) // (self.vae_scale_factor * 2))
        width = 2 * (int(width) // (self.vae_scale_factor * 2))
        image_latents = image_ids = None
        if image is not None:
            image = image.to(device=device)
            if isinstance(image, image_processor.NhwcPixels):  # fused pixel route (explicit marker, not a shape test)
                image_latents = self._encode_vae_image(image.tensor, nhwc=True)
            elif image.shape[1] != self.latent_channels:
                image_latents = self._encode_vae_image(image)
            else:
                image_latents = image.to(dtype)
            n = image_latents.shape[0]
            if batch_size > n and batch_size % n == 0:
                image_latents = torch.cat([image_latents] * (batch_size // n), dim=0)
            elif batch_size > n:
                raise ValueError(f"Cannot duplicate `image` of batch size {n} to {batch_size} text prompts.")
            ih, iw = image_latents.shape[2:]
            image_latents = self._pack_latents(image_latents, batch_size, num_channels_latents, ih, iw)
            image_ids = self._prepare_latent_image_ids(batch_size, ih // 2, iw // 2, device, dtype)
            image_ids[..., 0] = 1  # condition tokens: first id = 1
        latent_ids = self._prepare_latent_image_ids(batch_size, height // 2, width // 2, device, dtype)
        if latents is None:
            shape = (batch_size, num_channels_latents, height, width)
            if isinstance(generator, list):
                # diffusers' randn_tensor: sample i comes from its own generator (on that generator's device)
                noise = torch.cat([torch.randn((1, *shape[1:]), generator=g, device=g.device, dtype=dtype).to(device)
                                   for g in generator], dim=0)
            else:
                gdev = generator.device if isinstance(generator, torch.Generator) else device
                noise = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
            latents = self._pack_latents(noise, batch_size, num_channels_latents, height, width)
        else:
            latents = latents.to(device=device, dtype=dtype)
        return latents, image_latents, latent_ids, image_ids

    @torch.no_grad()
    def __call__(self, image=None, prompt=None, prompt_2=None, negative_prompt=None, negative_prompt_2=None,
                 true_cfg_scale=1.0, height=None, width=None, num_inference_steps=28, sigmas=None,
                 guidance_scale=3.5, num_images_per_prompt=1, generator=None, latents=None, prompt_embeds=None,
                 pooled_prompt_embeds=None, negative_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 output_type="pil", return_dict=True, joint_attention_kwargs=None, callback_on_step_end=None,
                 callback_on_step_end_tensor_inputs=("latents",), max_sequence_length=512,
                 max_area=1024 ** 2, _auto_resize=True):
        device = self.device
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make "
                             "sure to only forward one of the two.")   # :510-514
        if prompt is not None:
            # string prompts: the pipeline's own CLIP / T5 encoders (:899-906); every caller in the reference passes
            # embeddings instead (VLM tokens and/or T5), which skip this
            prompt_embeds, pooled_prompt_embeds, _ = self.encode_prompt(
                prompt, prompt_2, device=device, max_sequence_length=max_sequence_length)
        if negative_prompt is not None and negative_prompt_embeds is None and true_cfg_scale > 1:
            negative_prompt_embeds, negative_pooled_prompt_embeds, _ = self.encode_prompt(
                negative_prompt, negative_prompt_2, device=device, max_sequence_length=max_sequence_length)
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        multiple_of = self.vae_scale_factor * 2
        oh, ow = height, width
        height, width = helpers.fit_to_max_area(height, width, max_area, multiple_of)
        if (height, width) != (oh, ow):
            print(f"Generation `height` and `width` have been adjusted to {height} and {width} to fit the model "
                  f"requirements.")
        self.check_inputs(height, width, prompt_embeds, pooled_prompt_embeds, max_sequence_length)
        self._guidance_scale = guidance_scale
        self._interrupt = False
        has_neg = negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None
        do_true_cfg = true_cfg_scale > 1 and has_neg  # :928
        batch_size = prompt_embeds.shape[0] * num_images_per_prompt
        prompt_embeds = prompt_embeds.to(device=device, dtype=BF16)
        pooled_prompt_embeds = pooled_prompt_embeds.to(device=device, dtype=BF16)
        if num_images_per_prompt > 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
            pooled_prompt_embeds = pooled_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=device, dtype=BF16)  # :436
        if do_true_cfg:
            # The reference runs a second transformer call per step on the negative embeddings (:1080-1095).  Samples
            # are independent in the MMDiT, so here the positive and the negative pass are ONE call on a batch of 2B
            # (twice the GEMM rows per weight read), followed by the fused combine kernel.
            neg_e = negative_prompt_embeds.to(device=device, dtype=BF16)
            neg_p = negative_pooled_prompt_embeds.to(device=device, dtype=BF16)
            if num_images_per_prompt > 1:
                neg_e = neg_e.repeat_interleave(num_images_per_prompt, dim=0)
                neg_p = neg_p.repeat_interleave(num_images_per_prompt, dim=0)
            if neg_e.shape != prompt_embeds.shape:
                raise NotImplementedError("true CFG needs negative_prompt_embeds of the positive embeddings' shape "
                                          "(pad the shorter prompt; the joint pass shares one text length)")
            model_embeds = torch.cat([prompt_embeds, neg_e], dim=0)
            model_pooled = torch.cat([pooled_prompt_embeds, neg_p], dim=0)
        else:
            model_embeds, model_pooled = prompt_embeds, pooled_prompt_embeds
        model_batch = model_embeds.shape[0]

        # 3. condition image: preferred-resolution snap + nearest resize (VaeImageProcessor tensor path)
        u8 = image_processor.as_uint8_nhwc(image) if image is not None else None
        if u8 is not None:
            # uint8 pixels: cli.prepare_condition_images + resize + preprocess + .to(bf16) as ONE HIP gather
            ih, iw = self.image_processor.get_default_height_width(u8.permute(0, 3, 1, 2))
            ih, iw = helpers.preferred_condition_size(ih, iw, multiple_of, _auto_resize)
            image = image_processor.pixels_to_latent_input(u8, ih, iw, device)
        elif image is not None and not (isinstance(image, torch.Tensor) and image.size(1) == self.latent_channels):
            if not isinstance(image, torch.Tensor) or image.dim() != 4:
                raise NotImplementedError("pass the condition image as a [N,3,H,W] tensor in [-1,1] (cli.py:99-116) "
                                          "or as uint8 pixels (PIL / numpy / uint8 tensor [N,H,W,3])")
            ih, iw = self.image_processor.get_default_height_width(image)
            ih, iw = helpers.preferred_condition_size(ih, iw, multiple_of, _auto_resize)
            image = self.image_processor.resize(image.float(), ih, iw) if (ih, iw) != tuple(image.shape[2:]) else image
            image = self.image_processor.preprocess(image, ih, iw)

        # 4. latents
        num_channels_latents = self.transformer.config.in_channels // 4
        latents, image_latents, latent_ids, image_ids = self.prepare_latents(
            image, batch_size, num_channels_latents, height, width, BF16, device, generator, latents)
        S_tgt = latents.shape[1]
        if image_ids is not None:
            latent_ids = torch.cat([latent_ids, image_ids], dim=0)
            tokens = torch.cat([latents, image_latents], dim=1).contiguous()  # built ONCE, updated in place
        else:
            tokens = latents.contiguous().clone()

        # 5. timesteps (host float32 arithmetic, like diffusers' numpy path)
        sig = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps) if sigmas is None else sigmas
        cfg = self.scheduler.config
        mu = helpers.calculate_shift(S_tgt, cfg.get("base_image_seq_len", 256), cfg.get("max_image_seq_len", 4096),
                                     cfg.get("base_shift", 0.5), cfg.get("max_shift", 1.15))
        self.scheduler.set_timesteps(sigmas=sig, mu=mu, device="cpu")
        timesteps = self.scheduler.timesteps
        self._num_timesteps = len(timesteps)
        # `t.expand(B).to(bf16)` then `/ 1000` (flux_pipeline.py:1065,1069): all steps prepared up front
        t_model = (timesteps.to(BF16) / 1000)[:, None].expand(-1, model_batch).contiguous().to(device)
        guidance = None
        if self.transformer.config.guidance_embeds:
            guidance = torch.full([model_batch], guidance_scale, device=device, dtype=torch.float32)
        self.scheduler.set_begin_index(0)
        model_tokens = torch.empty((model_batch, *tokens.shape[1:]), device=device, dtype=BF16) if do_true_cfg else tokens

        # 6. denoising loop: no allocation, no host sync
        L = SimpleNamespace(tokens=tokens, model_tokens=model_tokens, t_model=t_model, guidance=guidance, pooled=model_pooled,
                            embeds=model_embeds, text_ids=text_ids, latent_ids=latent_ids, S_tgt=S_tgt, batch_size=batch_size,
                            do_true_cfg=do_true_cfg, true_cfg_scale=true_cfg_scale, jak=joint_attention_kwargs or {},
                            geom=(height, width, None if image is None else tuple(image.shape[-2:])),
                            dsigma=[self.scheduler.dsigma(i) for i in range(len(timesteps))])
        if self.use_graph and callback_on_step_end is None and not joint_attention_kwargs and not self._interrupt:
            tokens = self._denoise_graph(L)
        else:
            self._denoise(L, callback_on_step_end, timesteps)

        latents = tokens[:, :S_tgt]
        if output_type == "latent":
            image_out = latents.contiguous()
        else:
            z = self._unpack_latents(latents, height, width, self.vae_scale_factor).contiguous()
            vcfg = self.vae.config
            img = self.vae.decode(z, return_dict=False, pre_div=vcfg.scaling_factor, pre_add=vcfg.shift_factor)[0]
            image_out = self.postprocess(img, output_type)
        if not return_dict:
            return (image_out,)
        return FluxPipelineOutput(images=image_out, latents=latents)  # .latents: packed final latents (DP gather)

    def _denoise(self, L, callback_on_step_end=None, timesteps=None):
        """Conditioning of all steps in one pass, then the loop (flux_pipeline.py:1052-1120) over the persistent buffers."""
        if hasattr(self.transformer, "prepare_conditioning"):  # all steps' modulation vectors in one pass
            self.transformer.prepare_conditioning(L.t_model, L.guidance, L.pooled)
        tokens, B = L.tokens, L.batch_size
        for i in range(len(L.dsigma)):
            if self._interrupt:
                continue
            if L.do_true_cfg:
                L.model_tokens[:B].copy_(tokens)
                L.model_tokens[B:].copy_(tokens)
            noise_pred = self.transformer(
                hidden_states=L.model_tokens, timestep=L.t_model[i], guidance=L.guidance,
                pooled_projections=L.pooled, encoder_hidden_states=L.embeds,
                txt_ids=L.text_ids, img_ids=L.latent_ids, joint_attention_kwargs=L.jak,
                return_dict=False)[0]
            if L.do_true_cfg:
                noise_pred = ops.true_cfg(noise_pred[:B], noise_pred[B:], L.true_cfg_scale)
            ops.euler_step(tokens, noise_pred, L.S_tgt, L.dsigma[i])
            if callback_on_step_end is not None:
                out = callback_on_step_end(self, i, timesteps[i], {"latents": tokens[:, :L.S_tgt]})
                if out and "latents" in out:
                    tokens[:, :L.S_tgt].copy_(out["latents"])

    _GRAPH_INPUTS = ("tokens", "t_model", "guidance", "pooled", "embeds", "text_ids", "latent_ids")

    def _denoise_graph(self, L):
        """The same work as ONE graph launch.  The captured kernels read and write fixed buffers, so the call's tensors
        are copied into the graph's own (a few MB); everything the host decides during an eager pass -- launch plans,
        cached RoPE tables, the rows of the prepared conditioning, the Euler step sizes -- is frozen into the graph and
        therefore part of its key.  Returns the token buffer the graph updates."""
        # weights rewritten in place since the last forward (load_state_dict, a LoRA merge, an optimiser step): packed() checks
        # the source parameters' version stamps, re-packs the fused copies if needed and bumps the serial the key carries --
        # a replay never mixes fresh unfused weights with stale fused ones
        if hasattr(self.transformer, "packed"):
            self.transformer.packed()
        key = (tuple(L.tokens.shape), tuple(L.embeds.shape), tuple(L.latent_ids.shape), L.geom, L.do_true_cfg,
               float(L.true_cfg_scale), tuple(L.dsigma), L.guidance is None, L.S_tgt,
               getattr(self.transformer, "_pack_serial", 0), ops.launch_config_epoch())
        cur = torch.cuda.current_stream()
        if self._loop_graph is None or self._loop_graph[0] != key:
            self._loop_graph = None
            G = SimpleNamespace(**vars(L))
            for n in self._GRAPH_INPUTS:
                t = getattr(L, n)
                setattr(G, n, None if t is None else t.clone())
            G.model_tokens = torch.empty_like(L.model_tokens) if L.do_true_cfg else G.tokens
            side = torch.cuda.Stream(device=L.tokens.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):       # eager warm-up off the capture: kernel attributes, workspaces, RoPE cache
                self._denoise(G)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._denoise(G)
            self._loop_graph = (key, G, graph)
        _, G, graph = self._loop_graph
        for n in self._GRAPH_INPUTS:
            t = getattr(L, n)
            if t is not None:
                getattr(G, n).copy_(t)
        graph.replay()
        # the graph's own buffer is overwritten by the next same-shape call: hand out a copy, as the eager path hands out
        # fresh tensors (results of consecutive calls -- a list of latents, the DP gather -- must stay distinct)
        return G.tokens.clone()

    @staticmethod
    def postprocess(image, output_type="pil"):
        """VaeImageProcessor.postprocess (flux_pipeline.py:1130); 'pil' / 'np_uint8' quantise on the GPU."""
        return image_processor.VaeImageProcessor.postprocess(image, output_type)
