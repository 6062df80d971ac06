"""Command-line front end of the HIP path -- counterpart of the reference's ``univa/serve/cli.py``.

Same flags and defaults (``cli.py:271-283``), same helper names and semantics (``update_size`` :82-97,
``prepare_condition_images`` :99-116, ``load_pipe`` :58-76) and the same generation call (:239-248); what changes
is what sits behind ``pipe``: ``HipFluxTransformer2DModel`` + ``HipAutoencoderKL`` filled from the same checkpoints
(``gpt_image_edit_amd.checkpoint``).

Prompt understanding: the Qwen2.5-VL backbone and T5 / CLIP come from the installed
``transformers`` (reused as-is), wrapped by ``gpt_image_edit_amd.qwen_adaptor`` (task head, ``denoise_embeds`` forward,
HIP ``denoise_projector``) -- nothing is imported from the reference package.  ``--prompt_embeds FILE`` feeds the
generation half from saved embeddings instead (a ``torch.save``d dict with ``prompt_embeds`` [1,L,4096] and
``pooled_prompt_embeds`` [1,768]), which is also how the parity tests drive it offline.
"""
import argparse
import os

import numpy as np
import torch

from .. import checkpoint, flux_spec
from ..anyres_util import dynamic_resize
from ..pipeline import FluxKontextPipeline
from ..scheduler import FlowMatchEulerDiscreteScheduler
from ..transformer import HipFluxTransformer2DModel
from ..vae import HipAutoencoderKL

seed = 42  # cli.py:20-26 seeds everything with 42 and draws the edit's noise from Generator("cuda").manual_seed(seed)
generate_image_temp = "./generate_image_{}.png"


def load_pipe(denoiser, flux_path, device):
    """``FluxKontextPipeline.from_pretrained(flux_path, transformer=denoiser, torch_dtype=bf16).to(device)``.

    ``denoiser``: a ``HipFluxTransformer2DModel``, or a checkpoint directory (a UniWorld model directory with
    ``denoise_tower.denoiser.*`` keys, cli.py:127, or a diffusers FLUX directory), or None = ``flux_path``'s own
    transformer.  Returns (pipe, tokenizers, text_encoders) like the reference; the T5/CLIP encoders are loaded
    with ``transformers`` when ``flux_path`` holds them, else the two lists contain None."""
    if not isinstance(denoiser, HipFluxTransformer2DModel):
        src = denoiser or flux_path
        cfg = checkpoint.flux_transformer_config(flux_path) if os.path.isdir(os.path.join(flux_path, "transformer")) else None
        model = HipFluxTransformer2DModel(cfg, device=device)
        checkpoint.load_flux_transformer(model, src)
        denoiser = model
    vae = HipAutoencoderKL(device=device)
    checkpoint.load_vae(vae, flux_path)
    sched = FlowMatchEulerDiscreteScheduler(**checkpoint.scheduler_config(flux_path))
    tokenizers, text_encoders = [None, None], [None, None]
    try:  # reused as-is; absent offline
        from transformers import CLIPTextModel, CLIPTokenizer, T5EncoderModel, T5TokenizerFast
        if os.path.isdir(os.path.join(flux_path, "text_encoder")):
            tokenizers = [CLIPTokenizer.from_pretrained(flux_path, subfolder="tokenizer"),
                          T5TokenizerFast.from_pretrained(flux_path, subfolder="tokenizer_2")]
            text_encoders = [CLIPTextModel.from_pretrained(flux_path, subfolder="text_encoder", torch_dtype=torch.bfloat16).to(device),
                             T5EncoderModel.from_pretrained(flux_path, subfolder="text_encoder_2", torch_dtype=torch.bfloat16).to(device)]
    except Exception as e:  # pragma: no cover
        print(f"text encoders not loaded ({type(e).__name__}: {e}); pass --prompt_embeds")
    pipe = FluxKontextPipeline(denoiser, vae, sched, text_encoder=text_encoders[0], tokenizer=tokenizers[0],
                               text_encoder_2=text_encoders[1], tokenizer_2=tokenizers[1])
    return pipe, tokenizers, text_encoders


def update_size(i1, i2, anyres="any_11ratio", anchor_pixels=1024 * 1024):
    """Edit size for a turn (reference cli.py:82-97): with no input image a square of ``anchor_pixels``; otherwise the
    ``anyres`` bucket of the input's size -- of the MEAN width / height when two images are given."""
    from PIL import Image
    sizes = [Image.open(path).size for path in (i1, i2) if path]        # PIL: (width, height)
    if not sizes:
        side = int(anchor_pixels ** 0.5)
        return side, side
    mean_w = sizes[0][0] if len(sizes) == 1 else sum(w for w, _ in sizes) / len(sizes)
    mean_h = sizes[0][1] if len(sizes) == 1 else sum(h for _, h in sizes) / len(sizes)
    return dynamic_resize(int(mean_h), int(mean_w), anyres, anchor_pixels=anchor_pixels)


def prepare_condition_images(image_paths, device):
    """float32 [-1,1] condition tensor [N,3,H,W] exactly as the reference builds it (cli.py:99-116)."""
    from PIL import Image
    if not image_paths:
        return None
    cond = []
    for p in image_paths:
        img = Image.open(p).convert("RGB")
        t = torch.tensor(np.array(img), dtype=torch.float32) / 255.0
        cond.append((t.permute(2, 0, 1) - 0.5) / 0.5)
    return torch.stack(cond).to(device, dtype=torch.float32)


def prepare_condition_pixels(image_paths):
    """The same images as uint8 [N,H,W,3]: the pipeline then normalises / resizes / casts them in one HIP kernel
    (bit-identical to the float route; tests/test_hip_pixels.py)."""
    from PIL import Image
    if not image_paths:
        return None
    arrs = [np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8) for p in image_paths]
    return torch.from_numpy(np.stack(arrs))


def generate_image(pipe, prompt_embeds, pooled_prompt_embeds, history_image_paths, new_h, new_w, args, fused_pixels=True):
    """The generation call of cli.py:236-248."""
    cond = prepare_condition_pixels(history_image_paths) if fused_pixels else prepare_condition_images(history_image_paths, pipe.device)
    return pipe(
        image=cond,
        prompt_embeds=prompt_embeds,
        pooled_prompt_embeds=pooled_prompt_embeds,
        height=new_h,
        width=new_w,
        num_inference_steps=args.num_inference_steps,
        guidance_scale=args.guidance_scale,
        generator=torch.Generator(device="cuda").manual_seed(seed),
    ).images[0]


def run_t5_only(pipe, text_encoders, tokenizers, text, image1=None, image2=None, args=None):
    """One edit conditioned on T5 (256 tokens) + CLIP only, no VLM -- ``run_model_and_return_samples`` of the
    reference's ``univa/eval/imgedit/step1_gen_samples_T5_only.py:140-183``: size from ``update_size`` with
    ``anchor_pixels = height * width``, the first image resized to that size (PIL bilinear, which is what
    torchvision's ``Resize`` does for a PIL input) and normalised to [-1, 1], ``encode_prompt(..., 256, device, 1)``,
    then the pipeline call with its default generator."""
    from PIL import Image

    from ..prompt_embedding import encode_prompt
    new_h, new_w = update_size(image1, image2, "any_11ratio", anchor_pixels=args.height * args.width)
    pipeline_image = None
    if image1:
        cond = Image.open(image1).convert("RGB").resize((new_w, new_h), Image.BILINEAR)
        pipeline_image = torch.from_numpy(np.asarray(cond).copy()).unsqueeze(0)      # uint8 [1, H, W, 3]: fused HIP route
    with torch.no_grad():
        t5_prompt_embeds, pooled_prompt_embeds = encode_prompt(text_encoders, tokenizers, text, 256, pipe.device, 1)
    return pipe(
        image=pipeline_image,
        prompt_embeds=t5_prompt_embeds,
        pooled_prompt_embeds=pooled_prompt_embeds,
        height=new_h,
        width=new_w,
        num_inference_steps=args.num_inference_steps,
        guidance_scale=args.guidance_scale,
        num_images_per_prompt=getattr(args, "num_images_per_prompt", 1),
    ).images


def build_parser():
    parser = argparse.ArgumentParser(description="Model and component paths")
    parser.add_argument("--model_path", type=str, required=True)
    parser.add_argument("--flux_path", type=str, required=True)
    parser.add_argument("--no_auto_hw", action="store_true")
    parser.add_argument("--height", type=int, default=1024)
    parser.add_argument("--width", type=int, default=1024)
    parser.add_argument("--num_inference_steps", type=int, default=28)
    parser.add_argument("--guidance_scale", type=float, default=3.5)
    parser.add_argument("--ocr_enhancer", action="store_true")
    parser.add_argument("--no_joint_with_t5", action="store_true")
    # additions of this front end
    parser.add_argument("--prompt_embeds", type=str, default=None,
                        help="torch-saved dict(prompt_embeds, pooled_prompt_embeds): skip the VLM / T5 / CLIP stage")
    parser.add_argument("--images", type=str, default="", help="comma-separated condition images (with --prompt_embeds)")
    parser.add_argument("--output", type=str, default=generate_image_temp.format(0))
    parser.add_argument("--t5_only", type=str, default=None, metavar="INSTRUCTION",
                        help="one edit from T5 + CLIP embeddings of INSTRUCTION only (no VLM), with --images")
    return parser


def main(args):
    device = torch.device("cuda")
    pipe, tokenizers, text_encoders = load_pipe(args.model_path, args.flux_path, device)
    if args.prompt_embeds:
        blob = torch.load(args.prompt_embeds, map_location="cpu", weights_only=True)
        urls = [u.strip() for u in args.images.split(",") if u.strip()]
        new_h, new_w = args.height, args.width
        if urls and not args.no_auto_hw:
            new_h, new_w = update_size(urls[0], urls[1] if len(urls) > 1 else None, "any_11ratio",
                                       anchor_pixels=args.height * args.width)
        img = generate_image(pipe, blob["prompt_embeds"], blob["pooled_prompt_embeds"], urls, new_h, new_w, args)
        img.save(args.output)
        print(f"Assistant: generate image at {args.output}")
        return
    if args.t5_only:
        if text_encoders[0] is None or text_encoders[1] is None:
            raise SystemExit(f"--t5_only needs the text_encoder / text_encoder_2 folders under {args.flux_path}")
        urls = [u.strip() for u in args.images.split(",") if u.strip()] + [None, None]
        imgs = run_t5_only(pipe, text_encoders, tokenizers, args.t5_only, urls[0], urls[1], args)
        imgs[0].save(args.output)
        print(f"Assistant: generate image at {args.output}")
        return
    chat(args, pipe, tokenizers, text_encoders, device)


def smart_resize(height, width, factor=28, min_pixels=4 * 28 * 28, max_pixels=16384 * 28 * 28):
    """(height, width) the Qwen2-VL front end resizes an image to: both multiples of ``factor`` (= 2 x 14-pixel patches
    merged 2 x 2), area within [min_pixels, max_pixels], aspect kept as far as the rounding allows.  Restated from the
    published algorithm of ``qwen-vl-utils`` (``vision_process.smart_resize``; un-pinned third-party dependency of the
    reference, ``requirements.txt:34``, not installed here), which the reference applies through
    ``process_vision_info(conversation)`` (``univa/serve/cli.py:189``) with the per-image ``min_pixels = max_pixels =
    448 * 448`` of ``cli.py:172``."""
    import math
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = max(factor, round(height / factor) * factor)
    w_bar = max(factor, round(width / factor) * factor)
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = math.floor(height / beta / factor) * factor
        w_bar = math.floor(width / beta / factor) * factor
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def vision_inputs(conversation):
    """``process_vision_info(conversation)[0]`` of the reference's cli (:189): every image entry of the conversation, in
    order, opened as RGB and resized (PIL's default filter, bicubic) to ``smart_resize`` of its own size under the
    entry's ``min_pixels`` / ``max_pixels`` (or of the entry's ``resized_height`` x ``resized_width`` when it names them:
    ``univa/eval/gedit/step1_gen_samples.py:119-124``) -- so that a 448 x 448-pixel budget yields the 16 x 16 merged-patch grid
    (256 image tokens for a square image) whatever the processor's own defaults are.  None when there is no image."""
    from PIL import Image
    out = []
    for message in conversation:
        for c in message["content"]:
            if c.get("type") != "image":
                continue
            img = c["image"] if isinstance(c["image"], Image.Image) else Image.open(c["image"])
            img = img.convert("RGB")
            w, h = img.size
            if "resized_height" in c and "resized_width" in c:      # a forced size wins over the pixel budget (gedit generator)
                h, w = c["resized_height"], c["resized_width"]
            rh, rw = smart_resize(h, w, 28, c.get("min_pixels", 4 * 28 * 28), c.get("max_pixels", 16384 * 28 * 28))
            out.append(img.resize((rw, rh)))
    return out or None


def load_main_model_and_processor(model_path, device, min_pixels=448 * 448, max_pixels=448 * 448):
    """(model, task_head, processor) like reference cli.py:30-55, built from this package's adaptor: the stock
    Qwen2.5-VL weights of the UniWorld directory, its ``denoise_tower.denoise_projector.*`` on the HIP projector and
    ``task_head_final.pt``."""
    from transformers import AutoProcessor

    from ..projector import HipDenoiseProjector
    from ..qwen_adaptor import TaskHead, UnivaQwen2p5VL, load_vlm
    proj = HipDenoiseProjector(device=device)
    proj.load_state_dict(checkpoint.read_projector(model_path))
    model = UnivaQwen2p5VL(load_vlm(model_path, device), proj)
    task_head = TaskHead().to(device)
    task_head.load_state_dict(checkpoint.read_task_head(model_path))
    processor = AutoProcessor.from_pretrained(model_path, min_pixels=min_pixels, max_pixels=max_pixels)
    return model, task_head, processor


def chat(args, pipe, tokenizers, text_encoders, device):
    """The interactive loop of reference cli.py:118-267 (text and / or image URLs per turn; empty turn exits)."""
    from ..prompt_embedding import encode_prompt
    from ..qwen_adaptor import encode_edit_prompt
    model, task_head, processor = load_main_model_and_processor(args.model_path, device)
    conversation, history_image_paths, n_generated = [], [], 0
    print("Interactive UniWorld-V1 Chat (Exit if input is empty)")
    while True:
        txt = input("Text prompt (or press Enter to skip): ").strip()
        img_input = input("Image URLs (comma-separated, or press Enter to skip): ").strip()
        if not img_input and not txt:
            print("Exit.")
            return
        if args.ocr_enhancer:
            raise SystemExit("--ocr_enhancer needs the reference's OCR service (out of scope, SURVEY.md section 2)")
        content = [{"type": "text", "text": txt}] if txt else []
        urls = [u.strip() for u in img_input.split(",") if u.strip()]
        new_h, new_w = args.height, args.width
        for url in urls:
            content.append({"type": "image", "image": url, "min_pixels": 448 * 448, "max_pixels": 448 * 448})
            history_image_paths.append(url)
        if urls:
            new_h, new_w = update_size(urls[0], urls[1] if len(urls) > 1 else None, "any_11ratio",
                                       anchor_pixels=args.height * args.width)
        conversation.append({"role": "user", "content": content})
        chat_text = processor.apply_chat_template(conversation, tokenize=False, add_generation_prompt=True)
        chat_text = "<|im_end|>\n".join(chat_text.split("<|im_end|>\n")[1:])      # drop the system turn (cli.py:186)
        images = vision_inputs(conversation)                                        # process_vision_info (cli.py:189)
        inputs = processor(text=[chat_text], images=images, padding=True, return_tensors="pt").to(device)
        # The processor applies its own smart_resize to what process_vision_info already resized (a 600 x 800 input becomes
        # 364 x 504, below min_pixels, and is scaled up again to 392 x 532): the reference does exactly this double resize
        # silently (cli.py:189-196) and so does this loop -- the grid the model sees is the processor's.
        t5_embeds, pooled = encode_prompt(text_encoders, tokenizers, txt if not args.no_joint_with_t5 else "", 256, device, 1)
        turn = encode_edit_prompt(model, task_head, inputs, t5_embeds, joint_with_t5=not args.no_joint_with_t5)
        if turn["generate"]:
            image = generate_image(pipe, turn["prompt_embeds"], pooled, history_image_paths, new_h, new_w, args)
            img_url = generate_image_temp.format(n_generated)
            n_generated += 1
            image.save(img_url)
            conversation.append({"role": "assistant", "content": [{"type": "image", "image": img_url}]})
            history_image_paths.append(img_url)
            print(f"Assistant: generate image at {img_url}\n")
        else:
            generated = model.vlm.generate(**inputs, max_new_tokens=128)
            reply = processor.batch_decode([o[len(i):] for i, o in zip(inputs.input_ids, generated)],
                                           skip_special_tokens=True, clean_up_tokenization_spaces=False)[0]
            print(f"Assistant: {reply}\n")
            conversation.append({"role": "assistant", "content": [{"type": "text", "text": reply}]})


if __name__ == "__main__":
    main(build_parser().parse_args())
