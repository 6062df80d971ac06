"""One optimisation step of the denoiser on the HIP path (BASELINE.json configs[4]).

Counterpart of the body of the reference's training loop, ``train_denoiser.py:935-1181``, for the shipped FLUX-Kontext
configuration (continuous timesteps, unit loss weights, guidance 1.0, AdamW, global-norm clipping at 1.0,
``only_tune_image_branch``): noisy input mixed and packed (``fk_flow_noisy_tokens_bf16``), MMDiT forward with one
checkpoint per block and backward (``backward.FluxBackward``), flow-matching loss fused with its gradient
(``fk_flow_loss_bf16``), squared gradient norm (``fk_sumsq``) and AdamW with the clipping coefficient folded in and the
bf16 parameter copy written in the same pass (``fk_adamw_step``).  The step's scalars (which parameters train, sigma
sampling, shift) are ``training.py``; the data-parallel exchange is ``zero.py``.  No torch arithmetic touches an
activation, gradient or parameter.

With ``projector=`` the ``denoise_projector`` trains along (it is in the reference's trainable set,
``train_denoiser.py:71-119``): ``vlm_hidden`` -- the frozen VLM's last hidden states -- goes through
``HipDenoiseProjector.forward_train``, the optional T5 ``prefix_prompt_embeds`` are appended as the reference's
``UnivaDenoiseTower.forward`` does (``modeling_univa_denoise_tower.py:62-70``), and the gradient of ``prompt_embeds``
that the MMDiT backward returns is carried through both Linears.  Its parameters appear as ``denoise_projector.*``.
The reference builds a ``joint_attention_kwargs['attention_mask']`` for padded multi-resolution batches
(``train_denoiser.py:907-916``) but ``UnivaDenoiseTower.forward`` pops it without passing it on
(``modeling_univa_denoise_tower.py:77``), so no mask ever reaches the attention: there is none here either.
"""
import torch

from . import helpers, ops
from .backward import FluxBackward

BF16 = torch.bfloat16


class DenoiserTrainStep:
    def __init__(self, model, lr=1e-6, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, trainable=None,
                 sharded=False, group=None, store_activations="auto", projector=None, keep_grads=True, bucket_numel=None):
        """sharded=True: the optimiser state lives in ``zero.ShardedAdamW`` (ZeRO-2: one flat bf16 parameter buffer the
        model's trainable tensors become views of, fp32 gradients reduce-scattered over the data-parallel ranks, this
        rank's slice of master + moments updated, parameters all-gathered); works unchanged with one process.
        keep_grads=False (sharded only): ``forward_backward`` hands every block's gradients to the optimiser's buckets and
        does not keep them (saves the 8 GB of bf16 gradients; ``step()['grads']`` is then empty)."""
        self.model = model
        self.projector = projector
        self.bw = FluxBackward(model, trainable, store_activations=store_activations)
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.state = {}     # name -> (fp32 master, exp_avg, exp_avg_sq)
        self.opt = None
        self.keep_grads = keep_grads
        self._sunk = {}     # name -> (data_ptr, version) of the gradients already handed to the sharded optimiser
        if sharded:
            from .zero import DEFAULT_BUCKET, ShardedAdamW, backward_order
            names = sorted(self.trainable_names())
            # buckets in the order the backward pass finishes the gradients: a bucket's reduce-scatter is issued the moment
            # its last block is done and runs under the backward of the earlier blocks (zero2.json: overlap_comm)
            self.opt = ShardedAdamW({k: self._param(k).data for k in names}, lr=lr, betas=betas, eps=eps,
                                    weight_decay=weight_decay, max_grad_norm=max_grad_norm, group=group,
                                    order=backward_order(names), bucket_numel=bucket_numel or DEFAULT_BUCKET)
            for k in names:
                self._param(k).data = self.opt.params[k]     # the forward now reads views of the flat buffer
            model._packed = None

    PROJ = "denoise_projector."

    def trainable_names(self):
        names = set(self.bw.trainable)
        if self.projector is not None:
            names |= {self.PROJ + k for k in self.projector.state_dict()}
        return names

    def _param(self, name):
        if name.startswith(self.PROJ):
            return self.projector.p(name[len(self.PROJ):])
        return self.model.p(name)

    def _state(self, name):
        st = self.state.get(name)
        if st is None:
            p = self._param(name)
            st = (p.detach().float().contiguous(), torch.zeros(p.shape, device=p.device, dtype=torch.float32),
                  torch.zeros(p.shape, device=p.device, dtype=torch.float32))
            self.state[name] = st
        return st

    @torch.no_grad()
    def prepare_inputs(self, model_input, cond_latents, noise, sigmas, prompt_embeds, pooled, guidance_scale=1.0):
        """The denoiser's keyword arguments for one batch, as the reference assembles them (``train_denoiser.py:996-1059,
        1064-1104``): noisy target tokens (mix + 2x2 packing in ONE kernel) followed by the condition tokens, ids with the
        condition's first coordinate set to 1, zero ``txt_ids``, ``timesteps / 1000`` in bf16, the guidance vector.
        Returns (kwargs of ``HipFluxTransformer2DModel.forward``, number of target tokens)."""
        dev = self.model.device
        B, C, h, w = model_input.shape
        S_tgt = (h // 2) * (w // 2)
        S_cond = 0 if cond_latents is None else (cond_latents.shape[2] // 2) * (cond_latents.shape[3] // 2)
        tokens = torch.empty(B, S_tgt + S_cond, 4 * C, device=dev, dtype=BF16)
        ops.flow_noisy_tokens(model_input.contiguous(), noise.contiguous(), sigmas.contiguous(), out=tokens[:, :S_tgt])
        ids = helpers._prepare_latent_image_ids(B, h // 2, w // 2, dev, BF16)
        if cond_latents is not None:
            ch, cw = cond_latents.shape[2], cond_latents.shape[3]
            tokens[:, S_tgt:].copy_(helpers._pack_latents(cond_latents.to(BF16), B, C, ch, cw))
            cids = helpers._prepare_latent_image_ids(B, ch // 2, cw // 2, dev, BF16)
            cids[..., 0] = 1
            ids = torch.cat([ids, cids], dim=0)
        txt_ids = torch.zeros(prompt_embeds.shape[1], 3, device=dev, dtype=BF16)
        guidance = torch.full([B], guidance_scale, device=dev, dtype=torch.float32)
        timestep = (sigmas * 1000.0).to(BF16) / 1000            # `timesteps / 1000` as the model receives it (:1073)
        return dict(hidden_states=tokens, encoder_hidden_states=prompt_embeds, pooled_projections=pooled, timestep=timestep,
                    img_ids=ids, txt_ids=txt_ids, guidance=guidance), S_tgt

    @staticmethod
    def _latent_map(m, B, h, w, dev):
        """A per-pixel weight map as the loss kernel reads it: fp32 [B, 1, h, w] on the device, nearest-resized to the latent
        size when it comes at another resolution (``F.interpolate(..., mode='nearest')``, train_denoiser.py:1131-1148)."""
        if m is None:
            return None
        m = m.to(device=dev, dtype=torch.float32)
        if m.dim() != 4 or m.shape[0] != B or m.shape[1] != 1:
            raise ValueError("area_mask_weights / weight_mask must be [B, 1, H, W]")
        if tuple(m.shape[-2:]) != (h, w):
            m = torch.nn.functional.interpolate(m, size=(h, w), mode="nearest")
        return m.contiguous()

    @torch.no_grad()
    def forward_backward(self, model_input, cond_latents, noise, sigmas, prompt_embeds=None, pooled=None, guidance_scale=1.0,
                         vlm_hidden=None, prefix_prompt_embeds=None, weighting=None, area_mask_weights=None, weight_mask=None):
        """(loss fp64 [1], grads, d_prompt_embeds) for one batch of equally sized samples; model_input / noise fp32
        [B,16,h,w] (VAE latents already shifted and scaled), cond_latents the same or None, sigmas fp32 [B].
        Either ``prompt_embeds`` (projector frozen / absent) or ``vlm_hidden`` [B,L,3584] (+ optional T5 prefix).
        The loss as the stage-2 config sets it up (``mask_weight_type: 'log'``; train_denoiser.py:1106-1166): ``weighting``
        fp32 [B] (``compute_loss_weighting_for_sd3`` / ``sigmas_as_weight``; None = ones), ``area_mask_weights`` [B,1,H,W]
        (the dataset's per-pixel area weights), ``weight_mask`` [B,1,H,W] (1 inside a padded sample's true extent; with it
        the sum is divided by ``weight_mask.sum() * C`` instead of the element count)."""
        n_proj = 0
        if vlm_hidden is not None:
            if self.projector is None or prompt_embeds is not None:
                raise ValueError("vlm_hidden needs projector= at construction and excludes prompt_embeds")
            prompt_embeds = self.projector.forward_train(vlm_hidden)
            n_proj = prompt_embeds.shape[1]
            if prefix_prompt_embeds is not None:
                prompt_embeds = torch.cat([prompt_embeds, prefix_prompt_embeds.to(BF16)], dim=1)
        elif prompt_embeds is None:
            raise ValueError("one of prompt_embeds / vlm_hidden is required")
        inp, S_tgt = self.prepare_inputs(model_input, cond_latents, noise, sigmas, prompt_embeds, pooled, guidance_scale)
        pred = self.bw.forward(inp["hidden_states"], inp["encoder_hidden_states"], inp["pooled_projections"], inp["timestep"],
                               inp["img_ids"], inp["txt_ids"], inp["guidance"])
        B, _, h, w = model_input.shape
        dev = pred.device
        if weighting is not None:
            weighting = weighting.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        loss, grad = ops.flow_loss(pred[:, :S_tgt], model_input.contiguous(), noise.contiguous(), weight=weighting,
                                   area_mask_weights=self._latent_map(area_mask_weights, B, h, w, dev),
                                   weight_mask=self._latent_map(weight_mask, B, h, w, dev))
        dsample = torch.zeros_like(pred)
        dsample[:, :S_tgt].copy_(grad)
        sink = None
        if self.opt is not None:
            # a second forward_backward before optimizer_step is a further micro-batch of the same step (the reference's
            # gradient_accumulation_steps): its gradients are ADDED to the optimiser's chunks, the step uses their mean
            self.opt.begin_micro_batch()

            def sink(block_grads):
                self.opt.accumulate(block_grads)
                for k, g in block_grads.items():
                    self._sunk[k] = (g.data_ptr(), g._version)
        grads, d_enc = self.bw.backward(dsample, sink=sink, keep=self.keep_grads or self.opt is None)
        if n_proj:
            pg = {self.PROJ + k: g for k, g in self.projector.backward(d_enc[:, :n_proj]).items()}
            if sink is not None:
                sink(pg)
            if self.keep_grads or self.opt is None:
                grads.update(pg)
        return loss, grads, d_enc

    @torch.no_grad()
    def optimizer_step(self, grads):
        """Global-norm clipping + AdamW on fp32 masters; the bf16 parameters of the model are rewritten in the same pass."""
        names = sorted(grads)
        if self.opt is not None:
            # whatever forward_backward has not already handed to the buckets (gradients from another source); tensors
            # without a gradient this step -- e.g. the projector on a batch that came with ready prompt_embeds -- count as 0
            rest = {}
            for k, g in grads.items():
                stamp = self._sunk.get(k)
                if stamp is None:
                    rest[k] = g
                elif stamp != (g.data_ptr(), g._version):
                    raise RuntimeError(f"optimizer_step: the gradient passed for {k} is not the one forward_backward already "
                                       "handed to the sharded optimiser (it was replaced or modified in place afterwards) and "
                                       "would be ignored: with sharded=True accumulate by calling forward_backward again, and "
                                       "scale through lr / max_grad_norm, not on the returned tensors")
            if rest:
                self.opt.accumulate(rest)
            self._sunk = {}
            norm = self.opt.step()
            self.step_count = self.opt.step_count
            self.bw.refresh()
            return norm * norm
        sumsq = ops.sumsq([grads[k].contiguous() for k in names])
        self.step_count += 1
        for k in names:
            master, m1, m2 = self._state(k)
            ops.adamw_step(master, grads[k].contiguous(), m1, m2, self.step_count, self.lr, self.betas, self.eps,
                           self.weight_decay, grad_sumsq=sumsq, max_grad_norm=self.max_grad_norm, param_bf16=self._param(k).data)
        self.bw.refresh()
        return sumsq

    def discard(self):
        """Drop the gradients of the backward passes since the last optimiser step (a step that is deliberately skipped, e.g.
        a non-finite loss; the reference's ``optimizer.zero_grad()``, train_denoiser.py:1180).  With ``sharded=True`` it finishes
        the reduce-scatters already in flight and starts none; the decision to skip must be the same on every rank (take it
        on an all-reduced loss or flag).  Without it a skipped backward pass would be summed into the
        next step as a further micro-batch."""
        if self.opt is not None:
            self.opt.zero_grad()
        self._sunk = {}

    def state_dict(self):
        """Optimiser state for ``accelerator.save_state``-style checkpoints (train_denoiser.py:1229): the ZeRO-2 shard of
        this rank (``zero.ShardedAdamW.state_dict``) or, unsharded, the per-tensor fp32 masters and moments."""
        if self.opt is not None:
            return dict(kind="sharded", opt=self.opt.state_dict())
        return dict(kind="per_tensor", step=self.step_count,
                    state={k: tuple(t.detach().cpu().clone() for t in st) for k, st in self.state.items()})

    @torch.no_grad()
    def load_state_dict(self, sd):
        """Resume (train_denoiser.py:349-367, 769): restores the state and rewrites the model's trainable bf16 parameters from
        the fp32 masters, so the next step continues bit for bit."""
        if (sd.get("kind") == "sharded") != (self.opt is not None):
            raise ValueError("optimiser state was saved with another `sharded` setting")
        self._sunk = {}
        if self.opt is not None:
            self.opt.load_state_dict(sd["opt"])
            self.step_count = self.opt.step_count
        else:
            self.step_count = int(sd["step"])
            self.state = {}
            for k, (master, m1, m2) in sd["state"].items():
                p = self._param(k)
                self.state[k] = tuple(t.to(p.device) for t in (master, m1, m2))
                p.data.copy_(self.state[k][0])
        self.bw.refresh()
        self.model._packed = None

    def step(self, **batch):
        loss, grads, d_enc = self.forward_backward(**batch)
        sumsq = self.optimizer_step(grads)
        return dict(loss=loss, grad_sumsq=sumsq, d_prompt_embeds=d_enc, grads=grads)
