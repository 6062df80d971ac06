"""world_size-2 CPU test (gloo) of the data-parallel harness: strided shard + one final all-gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gpt_image_edit_amd import dp
    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    items = list(range(n_items))
    assert dp.shard(items, rank, world) == items[rank::world]           # the reference's strided shard
    assert dp.shard_indices(n_items, rank, world) == items[rank::world]

    def edit(item):  # stand-in for pipe(...).latents: value encodes the item id
        return torch.full((1, 4, 8), float(item), dtype=torch.bfloat16)

    full = dp.generate_sharded(edit, items)
    ok = full.shape == (n_items, 4, 8) and all(float(full[i, 0, 0]) == float(i) for i in range(n_items))
    local = dp.generate_sharded(edit, items, gather=False)
    ok = ok and local.shape[0] == len(items[rank::world])
    g = dp.all_gather_latents(torch.full((2, 3, 5), float(rank), dtype=torch.bfloat16))
    ok = ok and g.shape == (2 * world, 3, 5) and float(g[0, 0, 0]) == 0.0 and float(g[-1, 0, 0]) == float(world - 1)
    out_q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_shard_and_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_unshard_order_single_process():
    from gpt_image_edit_amd import dp
    for n, w in [(8, 2), (12, 4), (8, 8), (4, 1)]:
        gathered = [it for r in range(w) for it in list(range(n))[r::w]]   # rank-major gathered order
        order = dp.unshard_order(n, w)
        assert [gathered[order[i]] for i in range(n)] == list(range(n))
    x = torch.randn(2, 3, 4)
    assert dp.all_gather_latents(x) is x  # no process group: identity
