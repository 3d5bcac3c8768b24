"""Prompt-level data parallelism: one process per GPU, independent CFG denoising loops.

The reference samples prompts one after another on one GPU (base/pipelines/sample.py:78-91) and has
no inference-time collective.  Prompts are independent, so they shard with no data-path exchange:
prompt i runs on rank i mod W.  Only two collectives exist, both outside the denoise loop
(torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests):
  * once per model load: broadcast of ONE flat fp16 weight buffer from rank 0 (1.818 GB for the base
    model — a single large message, ring-pipelined over the xGMI links);
  * once per batch: all_gather of each rank's finished latents (327,680 B per video in fp16).
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

Shape = Tuple[int, ...]


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_prompts(num_prompts: int, rank: int, world_size: int) -> List[int]:
    """Indices owned by `rank`: i mod W == rank (round-robin keeps ranks within one prompt of each other)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_prompts, world_size))


def flat_layout(shapes: Dict[str, Shape]) -> Tuple[List[Tuple[str, int, Shape]], int]:
    """Deterministic (name, offset, shape) layout of the flat weight buffer; offsets 8-element aligned."""
    layout, off = [], 0
    for name in sorted(shapes):
        n = 1
        for s in shapes[name]:
            n *= s
        layout.append((name, off, tuple(shapes[name])))
        off += (n + 7) // 8 * 8
    return layout, off


def broadcast_weights(shapes: Dict[str, Shape], state_dict: Optional[Dict[str, torch.Tensor]], device,
                      dtype: torch.dtype = torch.float16, src: int = 0) -> Dict[str, torch.Tensor]:
    """Rank `src` supplies `state_dict`; every rank returns name -> view into one flat device buffer."""
    rank, size = world()
    layout, total = flat_layout(shapes)
    flat = torch.empty(total, dtype=dtype, device=device)
    if rank == src:
        if state_dict is None:
            raise ValueError("the source rank must provide the state dict")
        for name, off, shape in layout:
            t = state_dict[name]
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
            flat[off:off + t.numel()].copy_(t.reshape(-1).to(dtype), non_blocking=True)
    if size > 1:
        dist.broadcast(flat, src=src)
    out = {}
    for name, off, shape in layout:
        n = 1
        for s in shape:
            n *= s
        out[name] = flat[off:off + n].view(shape)
    return out


def gather_latents(local: torch.Tensor, counts: Sequence[int]) -> List[torch.Tensor]:
    """all_gather of per-rank latents [n_r, ...] with possibly different n_r (padded to max)."""
    rank, size = world()
    if size == 1:
        return [local]
    if len(counts) != size:
        raise ValueError("counts must list every rank")
    nmax = max(counts)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(size)]
    dist.all_gather(bufs, pad)
    return [b[:c] for b, c in zip(bufs, counts)]


def run_prompts(denoise_one: Callable[[int], torch.Tensor], num_prompts: int) -> Tuple[List[int], List[torch.Tensor]]:
    """Runs this rank's prompts through `denoise_one(prompt_index) -> latents [1, ...]`, gathers everything and
    returns (prompt indices, latents) in global prompt order on every rank.  With fewer prompts than ranks the idle
    ranks contribute a zero-row buffer (rank 0, which always owns prompt 0, tells them the latent shape), so every rank
    reaches the collective: no rank ever raises while the others wait in all_gather."""
    rank, size = world()
    if num_prompts < 1:                      # known to every rank: all of them raise, none enters a collective
        raise ValueError("run_prompts needs at least one prompt")
    mine = shard_prompts(num_prompts, rank, size)
    outs = [denoise_one(i) for i in mine]
    counts = [len(shard_prompts(num_prompts, r, size)) for r in range(size)]
    local = torch.cat(outs, dim=0) if outs else None
    if size > 1 and min(counts) == 0:
        meta = [(tuple(local.shape[1:]), local.dtype) if rank == 0 else None]
        dist.broadcast_object_list(meta, src=0)
        if local is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            local = torch.zeros((0,) + meta[0][0], dtype=meta[0][1], device=dev)
    parts = gather_latents(local, counts)
    order, tensors = [], []
    for r, part in enumerate(parts):
        for j, idx in enumerate(shard_prompts(num_prompts, r, size)):
            order.append(idx)
            tensors.append(part[j:j + 1])
    pairs = sorted(zip(order, tensors), key=lambda p: p[0])
    return [p[0] for p in pairs], [p[1] for p in pairs]
