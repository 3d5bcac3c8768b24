"""CPU, world_size 2 over gloo: prompt sharding, the flat weight broadcast and the latent gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lavie_amd import prompt_dp


def test_shard_prompts_partition():
    for n, w in ((64, 8), (5, 2), (7, 4), (8, 8)):
        shards = [prompt_dp.shard_prompts(n, r, w) for r in range(w)]
        assert sorted(i for s in shards for i in s) == list(range(n))
        assert max(map(len, shards)) - min(map(len, shards)) <= 1
    with pytest.raises(ValueError):
        prompt_dp.shard_prompts(4, 2, 2)


def test_flat_layout_alignment():
    layout, total = prompt_dp.flat_layout({"b": (3,), "a": (2, 5), "c": (16,)})
    assert [n for n, _, _ in layout] == ["a", "b", "c"]
    assert all(off % 8 == 0 for _, off, _ in layout) and total == 16 + 8 + 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, num_prompts, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = {"w1": (4, 3), "w2": (5,), "w0": (2, 2, 2)}
        sd = {k: torch.arange(float(torch.tensor(v).prod())).reshape(v) + i for i, (k, v) in enumerate(shapes.items())} if rank == 0 else None
        views = prompt_dp.broadcast_weights(shapes, sd, "cpu", dtype=torch.float32)
        want = {k: torch.arange(float(torch.tensor(v).prod())).reshape(v) + i for i, (k, v) in enumerate(shapes.items())}
        assert all(torch.equal(views[k], want[k]) for k in shapes)

        def denoise_one(idx):          # stand-in for a CFG denoising loop: encodes which prompt and which rank
            return torch.full((1, 2, 3), float(idx * 10 + rank))

        order, lats = prompt_dp.run_prompts(denoise_one, num_prompts)
        assert order == list(range(num_prompts))
        for i, t in zip(order, lats):
            assert t.shape == (1, 2, 3) and float(t[0, 0, 0]) == i * 10 + (i % world)
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_run_prompts_rejects_zero_prompts_on_every_rank():
    with pytest.raises(ValueError):
        prompt_dp.run_prompts(lambda i: torch.zeros(1, 2), 0)


@pytest.mark.parametrize("num_prompts", [4, 5, 1])        # 1: fewer prompts than ranks — rank 1 idles, nobody hangs
def test_two_ranks_gloo(num_prompts):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, num_prompts, ret), nprocs=2, join=True)
    assert ret.get(0) and ret.get(1)
