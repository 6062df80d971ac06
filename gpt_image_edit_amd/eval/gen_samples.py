"""Data-parallel batch generation of edited images: one process per GPU, strided shard, skip-existing resume, PNG to disk.

Counterpart of the reference's torchrun eval generators -- ``univa/eval/gedit/step1_gen_samples.py`` (``init_gpu_env``
:82-92, ``main`` :208-253; the imgedit / T5-only variants share the loop) -- on the HIP pipeline:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m gpt_image_edit_amd.eval.gen_samples --model_path <UniWorld ckpt> --flux_path <FLUX.1-Kontext-dev> \\
        --gedit_prompt_path gedit_edit.json --gedit_image_dir imgs/ --output_dir out/ [--t5_only]

Every rank holds a full model replica and edits ``inference_list[rank::world]`` (:239) -- no collective inside the 28
denoise steps; an item whose output file already exists is skipped (:247), which is what makes a killed job resumable
(not with ``--gather_latents``: finished items are edited again, their latents are not on disk).
``--gather_latents`` adds the one real exchange of the path (``dp.all_gather_latents``: an RCCL all-gather of the finished
packed latents over xGMI, BASELINE.json configs[3]) and writes them, in item order, from rank 0.

``run(args, edit_fn)`` is the loop itself with the edit injected (``edit_fn(prompt, image_path) -> (PIL image or uint8
HWC array, packed latents [1, S, 64] or None)``); ``main`` builds ``edit_fn`` from the checkpoint (serve/cli.py's loaders).
"""
import argparse
import json
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import dp

__all__ = ["build_inference_list", "run", "build_parser", "main", "gedit_size", "build_turn"]


def build_inference_list(data, output_dir):
    """[(prompt, output_path, key, image_path)] in the JSON's order (``step1_gen_samples.py:229-237``)."""
    return [(value["prompt"], os.path.join(output_dir, value["id"]), key, value["id"]) for key, value in data.items()]


def set_seed(seed, rank):
    """accelerate's ``set_seed(seed, device_specific=True)`` as the reference calls it (:218): seed + rank everywhere."""
    seed = int(seed) + int(rank)
    import random
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


def _save(image, path):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    tmp = path + f".tmp{os.getpid()}"                       # a killed job never leaves a truncated file that resume would skip
    if hasattr(image, "save"):
        image.save(tmp, format=os.path.splitext(path)[1].lstrip(".").upper().replace("JPG", "JPEG") or "PNG")
    else:
        from PIL import Image
        Image.fromarray(np.asarray(image)).save(tmp, format="PNG")
    os.replace(tmp, path)


def run(args, edit_fn, rank=None, world=None):
    """The generation loop.  Returns dict(done=[keys edited now], skipped=[keys whose output existed], latents=gathered
    packed latents in item order or None)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    os.makedirs(args.output_dir, exist_ok=True)
    with open(args.gedit_prompt_path, "r") as f:
        data = json.load(f)
    items = build_inference_list(data, args.output_dir)
    mine = dp.shard(items, rank, world)                     # inference_list[rank::world_size]
    done, skipped, latents = [], [], []
    for prompt, output_path, key, image_path in mine:
        if os.path.exists(output_path) and not getattr(args, "gather_latents", False):
            skipped.append(key)                             # resume: finished items are not redone
            continue
        src = os.path.join(args.gedit_image_dir, image_path) if args.gedit_image_dir else image_path
        image, lat = edit_fn(prompt, src)
        if os.path.exists(output_path):
            skipped.append(key)
        else:
            _save(image, output_path)
            done.append(key)
        if lat is not None:
            latents.append(lat)
    gathered = None
    if getattr(args, "gather_latents", False):
        if len(items) % world != 0:
            raise ValueError("--gather_latents needs the item count divisible by the world size (pad the list)")
        if len(latents) != len(mine):
            raise RuntimeError(f"--gather_latents: rank {rank} holds {len(latents)} latents for its {len(mine)} items "
                               "(the edit function returned no latents, or the shard is empty)")
        local = torch.cat(latents, dim=0)
        full = dp.all_gather_latents(local)                 # ONE collective: [world * n_local, S, 64], rank-major
        gathered = full[torch.tensor(dp.unshard_order(len(items), world), device=full.device)] if world > 1 else full
        if rank == 0 and getattr(args, "latents_out", None):
            torch.save({"keys": [k for _, _, k, _ in items], "latents": gathered.cpu()}, args.latents_out)
    return dict(done=done, skipped=skipped, latents=gathered)


def build_parser():
    p = argparse.ArgumentParser(description="data-parallel edit generation (one process per GPU)")
    p.add_argument("--model_path", type=str, required=True)
    p.add_argument("--flux_path", type=str, required=True)
    p.add_argument("--gedit_prompt_path", type=str, required=True, help='JSON {key: {"prompt": ..., "id": relative image path}}')
    p.add_argument("--gedit_image_dir", type=str, default="")
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--num_inference_steps", type=int, default=28)
    p.add_argument("--guidance_scale", type=float, default=3.5)
    p.add_argument("--num_images_per_prompt", type=int, default=1)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--t5_only", action="store_true", help="condition on T5 + CLIP only (eval/imgedit/step1_gen_samples_T5_only.py)")
    p.add_argument("--joint_with_t5", action=argparse.BooleanOptionalAction, default=True,
                   help="append the T5 embeddings of the prompt to the VLM's (the reference's eval config default)")
    p.add_argument("--only_use_t5", action="store_true", help="prompt_embeds = T5 only, VLM forward still run (:166-167)")
    p.add_argument("--gather_latents", action="store_true",
                   help="all-gather the finished packed latents (one collective); re-edits items whose output already exists, "
                        "since their latents are not on disk")
    p.add_argument("--latents_out", type=str, default=None)
    return p


def gedit_size(image_path, height, width):
    """(vis_h, vis_w, gen_h, gen_w) of one GEdit item exactly as ``run_model_and_return_samples`` derives them (:99-116):
    the VLM sees the image at a fixed 448 x 448; the edit runs at the ``any_17ratio`` bucket of the input's own size,
    scaled to ``height * width`` pixels on a stride of 16 (``compute_size``, NOT the cli's ``dynamic_resize`` on 32)."""
    from PIL import Image

    from ..anyres_util import compute_size, pick_ratio
    ow, oh = Image.open(image_path).size
    rw, rh = pick_ratio(oh, ow, anyres="any_17ratio")
    gen_h, gen_w = compute_size(rw, rh, stride=16, anchor_pixels=height * width)
    return 448, 448, gen_h, gen_w


def build_turn(prompt_text, image1=None, image2=None, vis_hw=(448, 448)):
    """The single user turn of a GEdit item (:117-131): the image entries FIRST -- each forced to ``resized_height`` x
    ``resized_width`` = 448 x 448 whatever its aspect -- then the text.  Returns (conversation, image_paths)."""
    content, image_paths = [], []
    for img in (image1, image2):
        if img:
            content.append({"type": "image", "image": img, "resized_height": vis_hw[0], "resized_width": vis_hw[1]})
            image_paths.append(img)
    if prompt_text:
        content.append({"type": "text", "text": prompt_text})
    return [{"role": "user", "content": content}], image_paths


def main(args):
    """Per-rank entry (torch.distributed.run): build the replica, run the loop."""
    if args.gather_latents and args.t5_only:
        raise SystemExit("--gather_latents needs the packed latents of every edit; the --t5_only route returns images only")
    rank, local_rank, world = dp.init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    set_seed(args.seed, rank)
    from ..serve import cli
    pipe, tokenizers, text_encoders = cli.load_pipe(args.model_path, args.flux_path, device)
    if args.t5_only:
        def edit_fn(prompt, image_path):
            return cli.run_t5_only(pipe, text_encoders, tokenizers, prompt, image1=image_path, args=args)[0], None
    else:
        # run_model_and_return_samples (:95-205): one user turn = the item's image (448 x 448 for the VLM) + the text, ONE VLM
        # forward for the denoise embeddings (no task-head routing: every item is an edit), T5 / CLIP on the prompt
        # (joint_with_t5) or on "" , the condition image at its own resolution, the pipeline at gedit_size's size
        from ..prompt_embedding import encode_prompt
        model, _task_head, processor = cli.load_main_model_and_processor(args.model_path, device)

        @torch.no_grad()
        def edit_fn(prompt, image_path):
            _, _, gen_h, gen_w = gedit_size(image_path, args.height, args.width)
            conversation, image_paths = build_turn(prompt, image_path)
            chat_text = processor.apply_chat_template(conversation, tokenize=False, add_generation_prompt=True)
            chat_text = "<|im_end|>\n".join(chat_text.split("<|im_end|>\n")[1:])
            inputs = processor(text=[chat_text], images=cli.vision_inputs(conversation), padding=True, return_tensors="pt").to(device)
            lvlm = model(input_ids=inputs.input_ids, pixel_values=getattr(inputs, "pixel_values", None),
                         attention_mask=inputs.attention_mask, image_grid_thw=getattr(inputs, "image_grid_thw", None),
                         output_type="denoise_embeds")
            t5_embeds, pooled = encode_prompt(text_encoders, tokenizers, prompt if args.joint_with_t5 else "", 256, device, 1)
            if args.only_use_t5:
                prompt_embeds = t5_embeds
            else:
                prompt_embeds = torch.cat([lvlm, t5_embeds.to(lvlm.device, lvlm.dtype)], dim=1) if args.joint_with_t5 else lvlm
            out = pipe(image=cli.prepare_condition_pixels(image_paths), prompt_embeds=prompt_embeds,
                       pooled_prompt_embeds=pooled, height=gen_h, width=gen_w, num_inference_steps=args.num_inference_steps,
                       guidance_scale=args.guidance_scale, num_images_per_prompt=args.num_images_per_prompt)
            return out.images[0], out.latents[:1]
    res = run(args, edit_fn, rank, world)
    print(f"[rank {rank}/{world}] edited {len(res['done'])}, skipped {len(res['skipped'])} existing", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(build_parser().parse_args())
