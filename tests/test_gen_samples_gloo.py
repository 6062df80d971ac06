"""world_size-2 CPU test (gloo) of the data-parallel batch generator ``gpt_image_edit_amd/eval/gen_samples.py`` on a stub
edit function: strided shard (``inference_list[rank::world]``, reference ``univa/eval/gedit/step1_gen_samples.py:239``),
PNG per item, skip-existing resume (:247), the optional final all-gather of the packed latents in item order."""
import json
import os
import socket
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_ITEMS = 6


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _edit_fn(calls):
    def edit(prompt, image_path):                      # stand-in for the HIP pipeline: the item id is in the prompt
        i = int(prompt.split("#")[1])
        calls.append(i)
        img = np.full((8, 8, 3), 10 * i, dtype=np.uint8)
        return img, torch.full((1, 4, 64), float(i), dtype=torch.bfloat16)
    return edit


def _worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gpt_image_edit_amd import dp
    from gpt_image_edit_amd.eval import gen_samples
    dp.init_from_env(backend="gloo")
    args = SimpleNamespace(gedit_prompt_path=os.path.join(tmp, "prompts.json"), gedit_image_dir=os.path.join(tmp, "in"),
                           output_dir=os.path.join(tmp, "out"), gather_latents=False, latents_out=None)
    calls = []
    first = gen_samples.run(args, _edit_fn(calls))
    mine = list(range(N_ITEMS))[rank::world]
    ok = sorted(calls) == mine and first["done"] == [f"k{i}" for i in mine] and first["skipped"] == []
    dist.barrier()
    # resume: everything exists now -> nothing is edited again
    calls2 = []
    second = gen_samples.run(args, _edit_fn(calls2))
    ok = ok and calls2 == [] and second["done"] == [] and second["skipped"] == [f"k{i}" for i in mine]
    dist.barrier()
    # a killed job: one output missing -> only that item is redone, by the rank that owns it
    if rank == 0:
        os.remove(os.path.join(tmp, "out", "sub", "img3.png"))
    dist.barrier()
    calls3 = []
    gen_samples.run(args, _edit_fn(calls3))
    ok = ok and calls3 == ([3] if 3 in mine else [])
    dist.barrier()
    # the one exchange of the path: packed latents of every item, in item order, on every rank
    args.gather_latents, args.latents_out = True, os.path.join(tmp, "latents.pt")
    res = gen_samples.run(args, _edit_fn([]))
    lat = res["latents"]
    ok = ok and lat.shape == (N_ITEMS, 4, 64) and all(float(lat[i, 0, 0]) == float(i) for i in range(N_ITEMS))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gen_samples_shard_resume_and_gather_gloo_world2(tmp_path):
    from PIL import Image
    os.makedirs(tmp_path / "in" / "sub")
    data = {f"k{i}": {"prompt": f"make it blue #{i}", "id": f"sub/img{i}.png"} for i in range(N_ITEMS)}
    json.dump(data, open(tmp_path / "prompts.json", "w"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
    for i in range(N_ITEMS):                           # every item exactly once, readable, with its own content
        img = np.asarray(Image.open(tmp_path / "out" / "sub" / f"img{i}.png"))
        assert img.shape == (8, 8, 3) and int(img[0, 0, 0]) == 10 * i
    assert not [f for f in os.listdir(tmp_path / "out" / "sub") if ".tmp" in f]
    saved = torch.load(tmp_path / "latents.pt")
    assert saved["keys"] == [f"k{i}" for i in range(N_ITEMS)] and saved["latents"].shape == (N_ITEMS, 4, 64)


def test_gen_samples_single_process_and_parser(tmp_path):
    from gpt_image_edit_amd.eval import gen_samples
    os.makedirs(tmp_path / "in")
    json.dump({"a": {"prompt": "x #1", "id": "a.png"}, "b": {"prompt": "y #2", "id": "b.png"}}, open(tmp_path / "p.json", "w"))
    args = gen_samples.build_parser().parse_args(["--model_path", "m", "--flux_path", "f", "--gedit_prompt_path", str(tmp_path / "p.json"),
                                                  "--gedit_image_dir", str(tmp_path / "in"), "--output_dir", str(tmp_path / "o")])
    assert args.num_inference_steps == 28 and args.guidance_scale == 3.5 and args.height == 1024
    calls = []
    res = gen_samples.run(args, _edit_fn(calls), rank=0, world=1)
    assert calls == [1, 2] and res["done"] == ["a", "b"] and sorted(os.listdir(tmp_path / "o")) == ["a.png", "b.png"]
    assert gen_samples.set_seed(42, 3) == 45


def test_gedit_turn_and_sizes_follow_the_reference_generator(tmp_path):
    """The conversation and the edit size of one GEdit item as ``univa/eval/gedit/step1_gen_samples.py:95-131`` builds them.
    Expected sizes: the reference's own ``pick_ratio(oh, ow, 'any_17ratio')`` + ``compute_size(rw, rh, stride=16,
    anchor_pixels=...)`` executed in the build container (``univa.utils.anyres_util`` imports there)."""
    from PIL import Image

    from gpt_image_edit_amd.eval import gen_samples
    from gpt_image_edit_amd.serve import cli
    expected = {  # (width, height) of the input -> (gen_h, gen_w) at 1024^2 and at 512^2 anchor pixels
        (640, 480): ((880, 1184), (432, 592)), (480, 640): ((1184, 880), (592, 432)), (1024, 1024): ((1024, 1024), (512, 512)),
        (1920, 1080): ((752, 1392), (368, 688)), (500, 1500): ((1552, 656), (768, 320)), (333, 777): ((1552, 656), (768, 320)),
        (1000, 700): ((832, 1248), (416, 624)),
    }
    for (w, h), (big, small) in expected.items():
        path = str(tmp_path / f"in_{w}x{h}.png")
        Image.new("RGB", (w, h), (w % 256, h % 256, 7)).save(path)
        assert gen_samples.gedit_size(path, 1024, 1024) == (448, 448) + big
        assert gen_samples.gedit_size(path, 512, 512) == (448, 448) + small
        # the cli's rule (dynamic_resize on stride 32, 11-ratio family) is a different function: not what this generator uses
        convo, paths = gen_samples.build_turn("make it snow", path)
        assert paths == [path] and len(convo) == 1 and convo[0]["role"] == "user"
        kinds = [c["type"] for c in convo[0]["content"]]
        assert kinds == ["image", "text"]                                   # image entries first, then the text
        entry = convo[0]["content"][0]
        assert (entry["resized_height"], entry["resized_width"]) == (448, 448) and "min_pixels" not in entry
        (seen,) = cli.vision_inputs(convo)
        assert seen.size == (448, 448)                                      # forced square whatever the input's aspect
    convo, paths = gen_samples.build_turn("two", "a.png", "b.png")
    assert [c["type"] for c in convo[0]["content"]] == ["image", "image", "text"] and paths == ["a.png", "b.png"]
    convo, paths = gen_samples.build_turn("", "a.png")
    assert [c["type"] for c in convo[0]["content"]] == ["image"]
    args = gen_samples.build_parser().parse_args(["--model_path", "m", "--flux_path", "f", "--gedit_prompt_path", "p", "--output_dir", "o"])
    assert args.joint_with_t5 and not args.only_use_t5 and args.num_images_per_prompt == 1


def test_gather_latents_without_latents_is_refused_up_front(tmp_path):
    from gpt_image_edit_amd.eval import gen_samples
    import pytest
    with open(tmp_path / "p.json", "w") as f:
        json.dump({"k0": {"prompt": "x#0", "id": "a.png"}}, f)
    args = SimpleNamespace(gedit_prompt_path=str(tmp_path / "p.json"), gedit_image_dir="", output_dir=str(tmp_path / "out"),
                           gather_latents=True, latents_out=None)
    with pytest.raises(RuntimeError, match="holds 0 latents"):
        gen_samples.run(args, lambda prompt, path: (np.zeros((4, 4, 3), np.uint8), None), rank=0, world=1)
    a = gen_samples.build_parser().parse_args(["--model_path", "m", "--flux_path", "f", "--gedit_prompt_path", "p", "--output_dir", "o",
                                               "--gather_latents", "--t5_only"])
    with pytest.raises(SystemExit, match="t5_only"):
        gen_samples.main(a)
