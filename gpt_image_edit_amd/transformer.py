"""HipFluxTransformer2DModel -- the MI355X-native FLUX-Kontext MMDiT behind the reference's seam.

Drop-in for the ``transformer=`` object the reference injects into its pipeline
(``FluxKontextPipeline.from_pretrained(flux_path, transformer=denoiser, ...)``, reference
``univa/serve/cli.py:64-68``) and for ``model.denoise_tower.denoiser``
(``univa/models/modeling_univa_denoise_tower.py:21``): same call signature as the pipeline uses at
``univa/utils/flux_pipeline.py:1067-1077``, same ``.config.in_channels / .guidance_embeds``, ``.dtype``,
and the diffusers state-dict key names, so real checkpoints load unchanged.

All arithmetic runs in libfk.so (hand-written gfx950 HIP kernels) through the C ABI; torch only
owns the device buffers.  There is no eager / CPU fallback: without a GPU + libfk.so it raises.

Per denoise step the forward issues, on the current HIP stream and with no host sync:
  embedders (skinny GEMMs) -> ONE modulation GEMM for all 57 blocks -> 19 double blocks
  {joint LN+modulate, grouped fused-QKV GEMM (text+image in one grid; RMSNorm+RoPE+layout in its epilogue), attention, grouped gated
   out-proj GEMM, joint LN+modulate, grouped MLP-up (GELU) and MLP-down (gated residual) GEMMs} -> 38 single blocks {LN+modulate, QKV GEMM, MLP-up GEMM(+GELU), qkv_post, attention,
   gated proj_out GEMM over [attn | mlp]} -> final LN+modulate -> proj_out.
Text and image streams live in ONE [B, S, D] residual buffer (text rows first), so the double->single
transition needs no concatenation and attention always sees one contiguous sequence.
"""
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import flux_spec, ops
from .param_tree import ParamTreeMixin, build_param_tree

BF16 = torch.bfloat16
# FK_FUSE_QKV=0 keeps RMSNorm+RoPE as the separate fk_qkv_post_bf16 pass (A/B measurement, identical results)
FUSE_QKV = os.environ.get("FK_FUSE_QKV", "1") != "0"
# FK_OVERLAP_MLP=0 / 1 / auto: never / always / by the attention grid's last-round waste (HipFluxTransformer2DModel.
# _overlap_pays) run the single blocks' MLP-up GEMM on a second stream (identical results).  Default 0 since round 3: with
# the mixed / split-K GEMM grids the second stream no longer pays (1024^2 edit, one box, interleaved: 4033.5 ms on one
# stream, 4039.6 ms with it; round 2 measured +2.6 % for it)
# single-stream order of a single block's two projections of n (A/B, FK_MLP_FIRST=1: MLP-up before the QKV GEMM; default:
# after the attention, which then reads q / k / v while they are cache-warm -- cfg 2 on one box, three interleaved runs
# each: 1.0149 / 1.0156 / 1.0150 images/s against 1.0145 / 0.9909 / 0.9646)
MLP_FIRST = os.environ.get("FK_MLP_FIRST", "0") == "1"
# FK_BLOCK_API: how the 57 blocks of a forward are enqueued.  0 = one ctypes call per kernel launch (~190 per forward),
# 1 = one per block (fk_double_block_fwd / fk_single_block_fwd), 2 = ONE per forward (fk_mmdit_blocks_fwd; default).  The C
# entry points issue the same launches with the same arguments: identical bits (tests/test_hip_mmdit.py).
BLOCK_API = int(os.environ.get("FK_BLOCK_API", "2"))
OVERLAP_MLP = {"0": False, "1": True, "auto": "auto"}.get(os.environ.get("FK_OVERLAP_MLP", "0"), False)


def rope_tables(ids, axes_dim=(16, 56, 56), theta=10000.0):
    """FluxPosEmbed (step-invariant, computed once per call shape): ids [S,3] -> cos, sin fp32 [S, sum(axes_dim)],
    frequencies in fp64 like diffusers.  Runs on the device the ids live on -- a handful of
    tiny fp64 kernels, no host round trip: the reference's training loop builds ``txt_ids`` afresh every step
    (``modeling_univa_denoise_tower.py:73-75``), and a ``.cpu()`` here would synchronise the device each time."""
    pos = ids.detach().to(torch.float32)
    cos_parts, sin_parts = [], []
    for i, d in enumerate(axes_dim):
        inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=pos.device) / d))
        ang = pos[:, i].to(torch.float64)[:, None] * inv[None, :]
        cos_parts.append(torch.repeat_interleave(torch.cos(ang), 2, dim=1).float())
        sin_parts.append(torch.repeat_interleave(torch.sin(ang), 2, dim=1).float())
    return torch.cat(cos_parts, dim=1).contiguous(), torch.cat(sin_parts, dim=1).contiguous()


class HipFluxTransformer2DModel(ParamTreeMixin, nn.Module):
    """Module tree = the diffusers ``FluxTransformer2DModel``'s: ``transformer_blocks.{i}.attn.to_q`` ... are
    parameter-holding sub-modules (``param_tree.ParamNode``), so the reference's ``named_modules()`` selection
    (``train_denoiser.py:538-543``), ``named_parameters()`` and ``state_dict()`` see the names of Appendix C."""

    def __init__(self, config=None, device="cuda", dtype=BF16, init="empty", seed=0):
        super().__init__()
        if dtype != BF16:
            raise ValueError("the HIP path computes in bf16 (fp32 accumulate); dtype must be torch.bfloat16")
        cfg = dict(flux_spec.FLUX_KONTEXT_CONFIG)
        cfg.update(config or {})
        self.config = SimpleNamespace(**cfg)
        self.num_heads = cfg["num_attention_heads"]
        if cfg["attention_head_dim"] != 128:
            raise ValueError("attention_head_dim must be 128 (kernel tiling)")
        self.inner_dim = self.num_heads * 128
        shapes = flux_spec.flux_param_shapes(cfg)
        if init == "synthetic":
            state = flux_spec.synthetic_state(shapes, seed=seed, device=device, dtype=dtype,
                                              gen_device="cuda" if str(device).startswith("cuda") else None)
        else:
            state = {k: torch.empty(s, device=device, dtype=dtype) for k, s in shapes.items()}
        self._names = list(shapes.keys())
        self.__dict__["_pmap"] = build_param_tree(self, state, requires_grad=False)
        self._packed = None
        self._train_packs = False     # training (backward.FluxBackward sets it): no fused copy of anything an optimiser rewrites
        self._ws = {}
        self._rope_cache = {}
        self._freqs = None
        self._cond = None

    # ---- state dict: the module tree carries the diffusers key names (no mangling) ----------------------------
    def load_state_dict(self, state_dict, strict=True, **kwargs):
        self._packed = None
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

    @property
    def dtype(self):
        return BF16

    @property
    def device(self):
        return self.p("x_embedder.weight").device

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._ws = {}
        self._rope_cache = {}
        self._freqs = None
        self._cond = None
        return super()._apply(fn, *a, **k)

    # ---- one-time weight packing (fused QKV, all-block modulation) -----------------------------------------
    def packed(self):
        """The fused operands, rebuilt when a source parameter of a fused COPY was rewritten since it was made (an optimiser
        step, ``param.data = ...``) or an aliased source moved: the check is one (data_ptr, version) tuple."""
        pk = self._packed
        if pk is None or pk.versions != self.param_versions(pk.sources) or pk.alias_ptrs != self._ptrs(pk.aliased):
            pk = self.pack_weights()
        return pk

    def _ptrs(self, names):
        pm = self._pmap
        return tuple(pm[n].data_ptr() for n in names)

    def _fuse(self, pk, names):
        """q | k | v (weights [3D, D] or biases [3D]) as ONE operand: a view when the three already lie side by side in
        memory (``zero.backward_order`` puts them so in the flat ZeRO buffer: an optimiser step then needs no re-pack), else
        a copy, whose sources join the version check."""
        ts = [self.p(n).data for n in names]
        side_by_side = all(t.is_contiguous() and t.untyped_storage().data_ptr() == ts[0].untyped_storage().data_ptr() for t in ts) and all(
            b.data_ptr() == a.data_ptr() + a.numel() * a.element_size() for a, b in zip(ts, ts[1:]))
        if side_by_side and self._train_packs:
            shape = (sum(t.shape[0] for t in ts),) + tuple(ts[0].shape[1:])
            pk.aliased += names
            return ts[0].new_empty(0).set_(ts[0].untyped_storage(), ts[0].storage_offset(), shape)
        pk.sources += names
        t = torch.cat(ts).contiguous()
        pk.copies.append((t, names))
        return t

    def repack(self, changed):
        """After ``changed`` parameters were rewritten behind torch's version counters (``fk_adamw_step`` writes through raw
        pointers): refresh, in place, the fused copies built from them.  Aliased operands need nothing."""
        pk = self._packed
        if pk is None:
            return
        if pk.mod_w is not None and any(n in changed for n in pk.mod_sources):
            self._packed = None
            return
        for t, names in pk.copies:
            if any(n in changed for n in names):
                torch.cat([self.p(n).data for n in names], out=t)
        pk.versions = self.param_versions(pk.sources)        # `changed` is everything that was rewritten: current again

    def pack_weights(self):
        c, D = self.config, self.inner_dim
        pk = SimpleNamespace(double=[], single=[], sources=[], aliased=[], copies=[])
        mod_w, mod_b, mod_names, off = [], [], [], 0
        qkv = lambda p, trio, wb: [p + f"attn.{n}.{wb}" for n in trio]  # noqa: E731
        for i in range(c.num_layers):
            p = f"transformer_blocks.{i}."
            blk = SimpleNamespace()
            blk.wqkv_img = self._fuse(pk, qkv(p, ("to_q", "to_k", "to_v"), "weight"))
            blk.bqkv_img = self._fuse(pk, qkv(p, ("to_q", "to_k", "to_v"), "bias"))
            blk.wqkv_txt = self._fuse(pk, qkv(p, ("add_q_proj", "add_k_proj", "add_v_proj"), "weight"))
            blk.bqkv_txt = self._fuse(pk, qkv(p, ("add_q_proj", "add_k_proj", "add_v_proj"), "bias"))
            blk.mod_img, blk.mod_txt = off, off + 6 * D
            off += 12 * D
            mod_names += [p + "norm1.linear", p + "norm1_context.linear"]
            pk.double.append(blk)
        for i in range(c.num_single_layers):
            p = f"single_transformer_blocks.{i}."
            blk = SimpleNamespace()
            blk.wqkv = self._fuse(pk, qkv(p, ("to_q", "to_k", "to_v"), "weight"))
            blk.bqkv = self._fuse(pk, qkv(p, ("to_q", "to_k", "to_v"), "bias"))
            blk.mod = off
            off += 3 * D
            mod_names.append(p + "norm.linear")
            pk.single.append(blk)
        pk.mod_out = off
        off += 2 * D
        mod_names.append("norm_out.linear")
        pk.mod_total = off
        pk.mod_sources = [n + s for n in mod_names for s in (".weight", ".bias")]
        if self._train_packs:
            # training: the 6.5 GB of modulation weights are half of the trainable parameters -- no fused copy to rebuild after
            # every optimiser step; the conditioning runs one small-M GEMM per block on the parameters themselves
            pk.mod_w = pk.mod_b = None
            pk.mod_parts, o = [], 0
            for n in mod_names:
                w = self.p(n + ".weight")
                pk.mod_parts.append((n, o, w.shape[0]))
                o += w.shape[0]
            assert o == off
        else:
            pk.mod_w = torch.cat([self.p(n + ".weight") for n in mod_names]).contiguous()
            pk.mod_b = torch.cat([self.p(n + ".bias") for n in mod_names]).contiguous()
            pk.sources += pk.mod_sources
        pk.versions, pk.alias_ptrs = self.param_versions(pk.sources), self._ptrs(pk.aliased)
        self.__dict__["_pack_serial"] = pk.serial = self.__dict__.get("_pack_serial", 0) + 1
        self._packed = pk
        return pk

    # ---- per-shape workspace (allocated once; the hot loop never allocates) -------------------------------
    def _workspace(self, B, S_txt, S_img):
        key = (B, S_txt, S_img)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        dev, D, H = self.device, self.inner_dim, self.num_heads
        S = S_txt + S_img
        S_pad = (S + 63) // 64 * 64
        e = lambda *shape: torch.empty(shape, device=dev, dtype=BF16)  # noqa: E731
        ws = SimpleNamespace(
            S=S, S_pad=S_pad,
            s=e(B, S, D), n=e(B, S, D), qkv=e(B, S, 3 * D), q=e(B, H, S, 128), k=e(B, H, S, 128),
            o=e(B, S, D), ff=e(B, S, 4 * D), cat=e(B, S, 5 * D),
            mod=e(B, self._packed.mod_total), temb=e(B, D), act=e(B, D), tproj=e(B, 256), e1=e(B, D),
            t_emb=e(B, D), g_emb=e(B, D), p_emb=e(B, D),
        )
        self._ws = {key: ws}  # keep only the latest shape
        return ws

    def _overlap_pays(self, B, S, cus=256):
        """Second stream for the single blocks' MLP-up GEMM?  Yes when the attention grid (256 query rows per workgroup,
        one workgroup per CU) needs more than one round and leaves >= 8 % of its rounds' CU-time empty."""
        nwg = B * self.num_heads * ((S + 255) // 256)
        rounds = (nwg + cus - 1) // cus
        return nwg > cus and (rounds * cus - nwg) / (rounds * cus) >= 0.08

    def _side_stream(self):
        st = self.__dict__.get("_side")
        if st is None:
            st = torch.cuda.Stream(device=self.device)
            self.__dict__["_side"] = st
            self.__dict__["_side_events"] = (torch.cuda.Event(), torch.cuda.Event())
        return st

    def _rope(self, txt_ids, img_ids, packed=False):
        # Step-invariant: keyed on the identity of the id tensors (the pipeline passes the same objects for
        # all 28 steps), so the hot loop never reads ids back to the host.  Strong refs keep the ids alive.
        key = (id(txt_ids), id(img_ids), txt_ids._version, img_ids._version)
        hit = self._rope_cache.get(key)
        if hit is None:
            t2 = txt_ids[0] if txt_ids.dim() == 3 else txt_ids
            i2 = img_ids[0] if img_ids.dim() == 3 else img_ids
            ids = torch.cat([t2.float().to(self.device), i2.float().to(self.device)], dim=0)
            cos, sin = rope_tables(ids, self.config.axes_dims_rope)           # on the device: no host sync
            hit = (cos, sin, txt_ids, img_ids, ops.pack_rope(cos, sin, check=False))
            self._rope_cache = {key: hit}
        return hit[4] if packed else (hit[0], hit[1])

    # ---- conditioning: temb and the modulation vectors of every block ------------------------------------
    def _conditioning(self, timestep, guidance, pooled, tproj, e1, t_emb, g_emb, p_emb, temb, act, mod):
        """CombinedTimestepGuidanceTextProjEmbeddings + every block's Linear(SiLU(temb)) (rows independent)."""
        c, P, pk = self.config, self.p, self._packed
        if self._freqs is None:
            self._freqs = torch.exp(-math.log(10000.0) * torch.arange(128, dtype=torch.float32) / 128).to(self.device)
        te = "time_text_embed."
        ops.timestep_proj(timestep if timestep.dtype in (BF16, torch.float32) else timestep.float(), self._freqs, out=tproj)
        ops.gemm(tproj, P(te + "timestep_embedder.linear_1.weight"), P(te + "timestep_embedder.linear_1.bias"), out=e1, epilogue=ops.FK_EPI_SILU)
        ops.gemm(e1, P(te + "timestep_embedder.linear_2.weight"), P(te + "timestep_embedder.linear_2.bias"), out=t_emb)
        if c.guidance_embeds:
            if guidance is None:
                raise ValueError("guidance is required when config.guidance_embeds is True")
            ops.timestep_proj(guidance if guidance.dtype in (BF16, torch.float32) else guidance.float(), self._freqs, out=tproj)
            ops.gemm(tproj, P(te + "guidance_embedder.linear_1.weight"), P(te + "guidance_embedder.linear_1.bias"), out=e1, epilogue=ops.FK_EPI_SILU)
            ops.gemm(e1, P(te + "guidance_embedder.linear_2.weight"), P(te + "guidance_embedder.linear_2.bias"), out=g_emb)
        else:
            g_emb.zero_()
        ops.gemm(pooled, P(te + "text_embedder.linear_1.weight"), P(te + "text_embedder.linear_1.bias"), out=e1, epilogue=ops.FK_EPI_SILU)
        ops.gemm(e1, P(te + "text_embedder.linear_2.weight"), P(te + "text_embedder.linear_2.bias"), out=p_emb)
        ops.add3(t_emb, g_emb, p_emb, out=temb)
        # every block's modulation vectors in one weight-streaming GEMM
        ops.silu(temb, out=act)
        if pk.mod_w is not None:
            ops.gemm(act, pk.mod_w, pk.mod_b, out=mod)
        else:   # training packs: per block, on the parameters themselves (same kernel, same rows: same bits)
            for n, o, rows in pk.mod_parts:
                ops.gemm(act, P(n + ".weight"), P(n + ".bias"), out=mod[:, o:o + rows])

    @torch.no_grad()
    def prepare_conditioning(self, timesteps, guidance, pooled_projections):
        """Optional: hand over ALL steps' timesteps ([N, B] bf16/fp32, t/1000) before a denoise loop.

        temb depends only on (timestep, guidance, pooled) -- never on the latents -- so the modulation vectors
        of all N steps are computed here in ONE pass (the 6.5 GB modulation weight is streamed once instead of
        N times).  A later ``forward(timestep=timesteps[i], guidance=guidance, pooled_projections=...)`` with
        these very tensors reuses row i; any other call computes its conditioning on the fly.  Every row goes
        through the same kernels as the per-step path, so results are bit-identical."""
        pk = self.packed()
        N, B = timesteps.shape
        D, dev = self.inner_dim, self.device
        ts = timesteps.contiguous()
        M = N * B
        e = lambda *shape: torch.empty(shape, device=dev, dtype=BF16)  # noqa: E731
        g_all = guidance.repeat(N).contiguous() if guidance is not None else None
        pooled = pooled_projections.to(BF16)
        p_all = pooled.repeat(N, 1).contiguous()
        mod = e(M, pk.mod_total)
        self._conditioning(ts.view(M), g_all, p_all, e(M, 256), e(M, D), e(M, D), e(M, D), e(M, D), e(M, D), e(M, D), mod)
        sources = [n for n in self._names if n.startswith("time_text_embed.")] + list(pk.mod_sources)
        self._cond = SimpleNamespace(ts=ts, N=N, B=B, step_bytes=B * ts.element_size(), guidance=guidance,
                                     pooled=pooled_projections, mod=mod.view(N, B, pk.mod_total), sources=sources,
                                     versions=self.param_versions(sources))

    def _cached_modulation(self, timestep, guidance, pooled_in):
        cd = self._cond
        if cd is None or timestep is None or timestep.dtype != cd.ts.dtype or timestep.numel() != cd.B:
            return None
        off = timestep.data_ptr() - cd.ts.data_ptr()
        if off < 0 or off % cd.step_bytes or off // cd.step_bytes >= cd.N or guidance is not cd.guidance:
            return None
        if pooled_in is not cd.pooled and pooled_in.data_ptr() != cd.pooled.data_ptr():
            return None
        if cd.versions != self.param_versions(cd.sources):
            self._cond = None          # an embedder / modulation weight was rewritten since (an optimiser step): stale
            return None
        return cd.mod[off // cd.step_bytes]

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None,
                img_ids=None, txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict=True,
                **unused):
        """Same arguments as diffusers' FluxTransformer2DModel.forward as called by the reference
        pipeline (flux_pipeline.py:1067-1077) and training loop (train_denoiser.py:1095-1104, through
        ``UnivaDenoiseTower.forward``).  ``timestep`` is t/1000.  Returns ``(sample,)``.

        Under ``torch.enable_grad()`` with a parameter (or ``encoder_hidden_states``) that requires grad, the call
        records ONE autograd node (``_FluxTrainFunction``): its forward is ``backward.FluxBackward.forward`` (the HIP
        training forward), its backward ``FluxBackward.backward`` -- so the reference's ``accelerator.backward(loss)``
        (``train_denoiser.py:1172``) fills ``.grad`` of exactly the parameters it un-froze and any stock optimiser
        can step them.  Otherwise (inference) nothing is recorded."""
        if not hidden_states.is_cuda:
            raise RuntimeError("HipFluxTransformer2DModel needs GPU tensors: there is no CPU fallback")
        if joint_attention_kwargs and joint_attention_kwargs.get("attention_mask") is not None:
            raise NotImplementedError("attention_mask (padded multi-resolution training batches, train_denoiser.py:1086-1091) "
                                      "is not supported: batch equally sized samples (cfg 5 trains bs 1 per GPU)")
        if torch.is_grad_enabled():
            names = self.grad_parameter_names()
            if names or (encoder_hidden_states is not None and encoder_hidden_states.requires_grad):
                sample = self._forward_train(names, hidden_states, encoder_hidden_states, pooled_projections, timestep,
                                             img_ids, txt_ids, guidance)
                return SimpleNamespace(sample=sample) if return_dict else (sample,)
        return self._forward_infer(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                                   guidance, return_dict)

    def grad_parameter_names(self):
        """Names of the parameters with ``requires_grad`` (what the reference's selection loop un-froze)."""
        return [n for n, prm in self._pmap.items() if prm.requires_grad]

    def _forward_train(self, names, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                       guidance):
        from .backward import FluxBackward, FluxTrainFunction
        key = tuple(names)
        bw = self.__dict__.get("_autograd_bw")
        if bw is None or bw.trainable_key != key:
            bw = FluxBackward(self, trainable=names,
                              store_activations=False if self.gradient_checkpointing else "auto")
            bw.trainable_key = key
            self.__dict__["_autograd_bw"] = bw
        bw.store_activations = False if self.gradient_checkpointing else "auto"
        params = [self._pmap[n] for n in names]
        return FluxTrainFunction.apply(bw, names, hidden_states, encoder_hidden_states, pooled_projections, timestep,
                                       img_ids, txt_ids, guidance, *params)

    @torch.no_grad()
    def _forward_infer(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
                       guidance, return_dict):
        c, D, H = self.config, self.inner_dim, self.num_heads
        pk = self.packed()
        B, S_img, _ = hidden_states.shape
        S_txt = encoder_hidden_states.shape[1]
        ws = self._workspace(B, S_txt, S_img)
        cos, sin = self._rope(txt_ids, img_ids)
        cs = self._rope(txt_ids, img_ids, packed=True)   # (cos, sin) per rotary pair: what the fused QKV epilogue reads
        if cos.shape[0] != ws.S:
            raise ValueError(f"txt_ids + img_ids give {cos.shape[0]} positions for a sequence of {ws.S}")
        hs = hidden_states.to(BF16).contiguous()
        enc = encoder_hidden_states.to(BF16).contiguous()
        pooled = pooled_projections.to(BF16).contiguous()
        P = self.p
        s, n = ws.s, ws.n
        h, cx = s[:, S_txt:], s[:, :S_txt]          # image / text residual streams (views)
        n_img, n_txt = n[:, S_txt:], n[:, :S_txt]

        # -- embedders -------------------------------------------------------------------------------
        ops.gemm(hs, P("x_embedder.weight"), P("x_embedder.bias"), out=h)
        ops.gemm(enc, P("context_embedder.weight"), P("context_embedder.bias"), out=cx)
        mod = self._cached_modulation(timestep, guidance, pooled_projections)
        if mod is None:
            self._conditioning(timestep, guidance, pooled, ws.tproj, ws.e1, ws.t_emb, ws.g_emb, ws.p_emb, ws.temb,
                               ws.act, ws.mod)
            mod = ws.mod

        def chunk(off, j):
            return mod[:, off + j * D: off + (j + 1) * D]

        block_api = BLOCK_API if (FUSE_QKV and not MLP_FIRST and not (OVERLAP_MLP and (OVERLAP_MLP != "auto" or self._overlap_pays(B, ws.S)))
                                  and S_txt > 0) else 0
        if block_api:
            self._blocks_by_c_entry(ws, pk, mod, cs, B, S_txt, S_img, block_api)
        else:
            self._blocks_by_kernel_calls(ws, pk, mod, cos, sin, cs, B, S_txt)

        # -- output head: AdaLayerNormContinuous (scale first, then shift) + proj_out -----------------------
        ops.ln_modulate(h, chunk(pk.mod_out, 1), chunk(pk.mod_out, 0), out=n_img)
        # the result is a FRESH tensor every call (64 channels per token: tiny; the caching allocator serves it without
        # a device sync): callers such as the reference pipeline's true-CFG branch keep one call's output while
        # making the next (flux_pipeline.py:1067-1095), which a persistent workspace buffer would silently alias
        sample = ops.gemm(n_img, P("proj_out.weight"), P("proj_out.bias"))
        if not return_dict:
            return (sample,)
        return SimpleNamespace(sample=sample)

    def _blocks_by_kernel_calls(self, ws, pk, mod, cos, sin, cs, B, S_txt):
        """The 19 + 38 blocks as one ctypes call per kernel launch (FK_BLOCK_API=0; also the route of the A/B switches
        FK_FUSE_QKV=0, FK_MLP_FIRST=1, FK_OVERLAP_MLP)."""
        P, D = self.p, self.inner_dim
        s, n = ws.s, ws.n
        h, cx = s[:, S_txt:], s[:, :S_txt]
        n_img, n_txt = n[:, S_txt:], n[:, :S_txt]

        def chunk(off, j):
            return mod[:, off + j * D: off + (j + 1) * D]

        # -- double-stream blocks ----------------------------------------------------------------------
        for i, blk in enumerate(pk.double):
            p = f"transformer_blocks.{i}."
            mi, mt = blk.mod_img, blk.mod_txt  # chunks: shift, scale, gate, shift_mlp, scale_mlp, gate_mlp
            # text + image streams share every launch: joint LN+modulate, grouped GEMMs (one grid, two weights)
            ops.ln_modulate2(s, chunk(mt, 0), chunk(mt, 1), chunk(mi, 0), chunk(mi, 1), S_txt, out=n)
            if FUSE_QKV:  # RMSNorm + RoPE + head-major q / k come out of the projection GEMM's epilogue
                ops.gemm_grouped([dict(a=n_img, w=blk.wqkv_img, bias=blk.bqkv_img, out=ws.qkv[:, S_txt:],
                                       qkv=dict(q_out=ws.q, k_out=ws.k, wq=P(p + "attn.norm_q.weight"),
                                                wk=P(p + "attn.norm_k.weight"), cs=cs, s_offset=S_txt)),
                                  dict(a=n_txt, w=blk.wqkv_txt, bias=blk.bqkv_txt, out=ws.qkv[:, :S_txt],
                                       qkv=dict(q_out=ws.q, k_out=ws.k, wq=P(p + "attn.norm_added_q.weight"),
                                                wk=P(p + "attn.norm_added_k.weight"), cs=cs, s_offset=0))],
                                 epilogue=ops.FK_EPI_QKV)
            else:
                ops.gemm_grouped([dict(a=n_img, w=blk.wqkv_img, bias=blk.bqkv_img, out=ws.qkv[:, S_txt:]),
                                  dict(a=n_txt, w=blk.wqkv_txt, bias=blk.bqkv_txt, out=ws.qkv[:, :S_txt])])
                ops.qkv_post(ws.qkv, ws.q, ws.k, P(p + "attn.norm_q.weight"), P(p + "attn.norm_k.weight"),
                             P(p + "attn.norm_added_q.weight"), P(p + "attn.norm_added_k.weight"), cos, sin, S_txt)
            ops.attention(ws.q, ws.k, ws.qkv[:, :, 2 * D:], ws.o)
            ops.gemm_grouped([dict(a=ws.o[:, S_txt:], w=P(p + "attn.to_out.0.weight"), bias=P(p + "attn.to_out.0.bias"),
                                   out=h, res=h, gate=chunk(mi, 2)),
                              dict(a=ws.o[:, :S_txt], w=P(p + "attn.to_add_out.weight"), bias=P(p + "attn.to_add_out.bias"),
                                   out=cx, res=cx, gate=chunk(mt, 2))], epilogue=ops.FK_EPI_GATE_RES)
            ops.ln_modulate2(s, chunk(mt, 3), chunk(mt, 4), chunk(mi, 3), chunk(mi, 4), S_txt, out=n)
            ops.gemm_grouped([dict(a=n_img, w=P(p + "ff.net.0.proj.weight"), bias=P(p + "ff.net.0.proj.bias"), out=ws.ff[:, S_txt:]),
                              dict(a=n_txt, w=P(p + "ff_context.net.0.proj.weight"), bias=P(p + "ff_context.net.0.proj.bias"),
                                   out=ws.ff[:, :S_txt])], epilogue=ops.FK_EPI_GELU_TANH)
            ops.gemm_grouped([dict(a=ws.ff[:, S_txt:], w=P(p + "ff.net.2.weight"), bias=P(p + "ff.net.2.bias"),
                                   out=h, res=h, gate=chunk(mi, 5)),
                              dict(a=ws.ff[:, :S_txt], w=P(p + "ff_context.net.2.weight"), bias=P(p + "ff_context.net.2.bias"),
                                   out=cx, res=cx, gate=chunk(mt, 5))], epilogue=ops.FK_EPI_GATE_RES)

        # -- single-stream blocks on the joint sequence ------------------------------------------------------
        # Every launch is a grid of whole-CU workgroups that runs in rounds of 256; at batch 1 the last round of each is
        # partly empty (attention 240 / 816 workgroups, GEMMs 1.6-6.4 rounds).  The block's MLP-up GEMM depends only on
        # the normalised input, so it goes to a second stream and its tiles fill the CUs the QKV GEMM's and the
        # attention's last rounds leave idle; one event pair per block, same kernels, same numbers.  Measured (one box,
        # images/s without / with): S = 2560 0.977 / 0.959, S = 5632 0.355 / 0.370, S = 8704 0.238 / 0.244 -- it pays
        # where the attention grid wastes a good part of its last round, and costs where one round holds everything.
        use_side = self._overlap_pays(B, ws.S) if OVERLAP_MLP == "auto" else OVERLAP_MLP
        side = self._side_stream() if use_side else None
        if side is not None:
            main = torch.cuda.current_stream()
            ev_n, ev_mlp = self._side_events
        for i, blk in enumerate(pk.single):
            p = f"single_transformer_blocks.{i}."
            m0 = blk.mod  # chunks: shift, scale, gate
            ops.ln_modulate(s, chunk(m0, 0), chunk(m0, 1), out=n)
            if side is not None:
                # the MLP branch needs only n: it runs beside the QKV projection and the attention
                ev_n.record(main)
                side.wait_event(ev_n)
                with torch.cuda.stream(side):
                    ops.gemm(n, P(p + "proj_mlp.weight"), P(p + "proj_mlp.bias"), out=ws.cat[:, :, D:],
                             epilogue=ops.FK_EPI_GELU_TANH)
                    ev_mlp.record(side)
            if side is None and MLP_FIRST:
                ops.gemm(n, P(p + "proj_mlp.weight"), P(p + "proj_mlp.bias"), out=ws.cat[:, :, D:],
                         epilogue=ops.FK_EPI_GELU_TANH)
            if FUSE_QKV:
                ops.gemm(n, blk.wqkv, blk.bqkv, out=ws.qkv, epilogue=ops.FK_EPI_QKV,
                         qkv=dict(q_out=ws.q, k_out=ws.k, wq=P(p + "attn.norm_q.weight"),
                                  wk=P(p + "attn.norm_k.weight"), cs=cs, s_offset=0))
            else:
                ops.gemm(n, blk.wqkv, blk.bqkv, out=ws.qkv)
                ops.qkv_post(ws.qkv, ws.q, ws.k, P(p + "attn.norm_q.weight"), P(p + "attn.norm_k.weight"), None, None,
                             cos, sin, 0)
            ops.attention(ws.q, ws.k, ws.qkv[:, :, 2 * D:], ws.cat[:, :, :D])
            if side is not None:
                main.wait_event(ev_mlp)
            elif not MLP_FIRST:
                ops.gemm(n, P(p + "proj_mlp.weight"), P(p + "proj_mlp.bias"), out=ws.cat[:, :, D:],
                         epilogue=ops.FK_EPI_GELU_TANH)
            ops.gemm(ws.cat, P(p + "proj_out.weight"), P(p + "proj_out.bias"), out=s,
                     epilogue=ops.FK_EPI_GATE_RES, res=s, gate=chunk(m0, 2))


    def _block_weight_structs(self, pk):
        """fk_double_block_weights / fk_single_block_weights of every block (arrays, built once per set of weight pointers): shared
        by the forward's and the backward's block-level entry points."""
        from . import libfk
        P = self.p
        names_d = ("attn.norm_q.weight", "attn.norm_k.weight", "attn.norm_added_q.weight", "attn.norm_added_k.weight",
                   "attn.to_out.0.weight", "attn.to_out.0.bias", "attn.to_add_out.weight", "attn.to_add_out.bias",
                   "ff.net.0.proj.weight", "ff.net.0.proj.bias", "ff_context.net.0.proj.weight", "ff_context.net.0.proj.bias",
                   "ff.net.2.weight", "ff.net.2.bias", "ff_context.net.2.weight", "ff_context.net.2.bias")
        names_s = ("attn.norm_q.weight", "attn.norm_k.weight", "proj_mlp.weight", "proj_mlp.bias", "proj_out.weight", "proj_out.bias")
        ptrs = []
        for i, blk in enumerate(pk.double):
            p = f"transformer_blocks.{i}."
            ptrs.append((blk.wqkv_img.data_ptr(), blk.bqkv_img.data_ptr(), blk.wqkv_txt.data_ptr(), blk.bqkv_txt.data_ptr())
                        + tuple(P(p + k).data_ptr() for k in names_d))
        for i, blk in enumerate(pk.single):
            p = f"single_transformer_blocks.{i}."
            ptrs.append((blk.wqkv.data_ptr(), blk.bqkv.data_ptr()) + tuple(P(p + k).data_ptr() for k in names_s))
        ptrs = tuple(ptrs)
        st = self.__dict__.get("_block_structs")
        if st is None or st.ptrs != ptrs:
            nd, ns = len(pk.double), len(pk.single)
            dbl, sgl = (libfk.DoubleBlockWeights * max(nd, 1))(), (libfk.SingleBlockWeights * max(ns, 1))()
            for i, blk in enumerate(pk.double):
                for f, v in zip(libfk.DOUBLE_BLOCK_FIELDS, ptrs[i]):
                    setattr(dbl[i], f, v)
                dbl[i].mod_off_img, dbl[i].mod_off_txt = blk.mod_img, blk.mod_txt
            for i, blk in enumerate(pk.single):
                for f, v in zip(libfk.SINGLE_BLOCK_FIELDS, ptrs[nd + i]):
                    setattr(sgl[i], f, v)
                sgl[i].mod_off = blk.mod
            st = SimpleNamespace(ptrs=ptrs, dbl=dbl, sgl=sgl, nd=nd, ns=ns)
            self.__dict__["_block_structs"] = st
        return st

    def _blocks_by_c_entry(self, ws, pk, mod, cs, B, S_txt, S_img, api):
        """The same launches through the block-level C entry points: argument structs built once per (weights, workspace)
        and re-used; per forward only the modulation pointer changes."""
        from . import libfk
        lib = libfk.load()
        st = self._block_weight_structs(pk)
        sk, slots = ops.splitk_workspace(ws.s.device)
        aw = ops.attention_workspace(ws.s.device)
        key = tuple(getattr(ws, f).data_ptr() for f in ("s", "n", "qkv", "q", "k", "o", "ff", "cat")) + (
            cs.data_ptr(), sk.data_ptr(), aw.data_ptr(), B, S_txt, S_img)
        bw = self.__dict__.get("_block_ws")
        if bw is None or bw[0] != key:
            c = libfk.BlockWs()
            for f in ("s", "n", "qkv", "q", "k", "o", "ff", "cat"):
                setattr(c, f, getattr(ws, f).data_ptr())
            c.rope_cs, c.splitk_ws, c.attn_ws, c.attn_ws_bytes, c.splitk_slots = cs.data_ptr(), sk.data_ptr(), aw.data_ptr(), aw.numel(), slots
            c.B, c.S_txt, c.S_img, c.H, c.eps = B, S_txt, S_img, self.num_heads, 1e-6
            bw = (key, c, (cs, sk, aw))            # strong references keep the buffers the struct points into alive
            self.__dict__["_block_ws"] = bw
        c = bw[1]
        L = ops.LAUNCH       # the host's launch defaults travel with the call (the library keeps no launch state)
        c.gemm_variant, c.gemm_plan, c.gemm_group_m, c.gemm_mfma, c.attn_grid = L.gemm_variant, L.gemm_plan, L.gemm_group_m, L.gemm_mfma, L.attn_grid
        import ctypes
        c.gemm_variant_used = ctypes.pointer(ops._variant_slot())
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        mp, mbs = ctypes.c_void_p(mod.data_ptr()), mod.stride(0)
        if api >= 2:
            libfk.check(lib.fk_mmdit_blocks_fwd(ctypes.byref(c), st.dbl, st.nd, st.sgl, st.ns, mp, mbs, stream), "fk_mmdit_blocks_fwd")
            return
        for i in range(st.nd):
            libfk.check(lib.fk_double_block_fwd(ctypes.byref(c), ctypes.byref(st.dbl[i]), mp, mbs, stream), "fk_double_block_fwd")
        for i in range(st.ns):
            libfk.check(lib.fk_single_block_fwd(ctypes.byref(c), ctypes.byref(st.sgl[i]), mp, mbs, stream), "fk_single_block_fwd")

    # The reference training code calls this on the denoiser (train_denoiser.py:486) because its 80 GB GPUs cannot hold
    # a 1024^2 sample's activations.  Here it selects FluxBackward's recompute policy (one checkpoint per block, the
    # block re-run in the backward pass: the reference's memory behaviour) for the autograd path; without the call the
    # training forward keeps every block's intermediates when they fit the activation budget (37 GB at 1024^2, bs 1:
    # sized for 288 GB of HBM) and nothing is recomputed.  Both policies give bit-identical gradients
    # (tests/test_hip_train_step.py).  ``DenoiserTrainStep(store_activations=...)`` chooses explicitly.
    gradient_checkpointing = False

    def enable_gradient_checkpointing(self):
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    # ---- HF-style checkpoint IO (train_denoiser.py:493 ``save_pretrained``; cli.py:64-68 ``from_pretrained``) -------
    def save_pretrained(self, save_directory, max_shard_size=5 << 30, **unused):
        """Write the diffusers layout of a transformer directory: ``config.json`` (``_class_name``
        ``FluxTransformer2DModel``) + ``diffusion_pytorch_model*.safetensors`` with the keys of Appendix C."""
        import json
        import os as _os
        from . import checkpoint
        _os.makedirs(save_directory, exist_ok=True)
        checkpoint.save_sharded(self.state_dict(), save_directory, max_shard_bytes=int(max_shard_size))
        cfg = {"_class_name": "FluxTransformer2DModel", "_diffusers_version": "0.32.2"}
        cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.config).items()})
        with open(_os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(cfg, f, indent=1)
        return save_directory

    @classmethod
    def from_pretrained(cls, directory, device="cuda", subfolder=None, torch_dtype=BF16, **unused):
        """Load a transformer directory written by diffusers' / this class's ``save_pretrained`` (or a FLUX pipeline
        directory with ``subfolder='transformer'``)."""
        import os as _os
        from . import checkpoint
        d = _os.path.join(directory, subfolder) if subfolder else directory
        cfg = checkpoint._config_of(d, flux_spec.FLUX_KONTEXT_CONFIG)
        model = cls(config=cfg, device=device, dtype=torch_dtype, init="empty")
        state = checkpoint.read_state_dict(d, dtype=BF16)
        checkpoint.check_against(flux_spec.flux_param_shapes(cfg), state, f"FLUX transformer at {d}")
        model.load_state_dict(state, strict=True)
        return model
