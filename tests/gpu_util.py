"""Helpers shared by the -m gpu parity tests (the HIP path vs the fp32 CPU oracle)."""
import torch

# fp16 storage + fp32 accumulation against an fp32 oracle: SURVEY.md §8c starting points
TOL_OP = 2e-3          # rel-L2 per operator
TOL_BLOCK = 4e-3       # resnet / transformer block
TOL_UNET = 1e-2        # one whole UNet forward


def rel_l2(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-12)).item()


def h16(t: torch.Tensor) -> torch.Tensor:
    """fp16 copy on the GPU."""
    return t.to(device="cuda", dtype=torch.float16).contiguous()


def f32(t: torch.Tensor) -> torch.Tensor:
    return t.to(device="cuda", dtype=torch.float32).contiguous()


def q16(t: torch.Tensor) -> torch.Tensor:
    """Round-trips a CPU fp32 tensor through fp16 so oracle and kernel see identical inputs."""
    return t.to(torch.float16).to(torch.float32)


def rows(x: torch.Tensor) -> torch.Tensor:
    """CPU [n, c, h, w] -> channels-last rows [(n h w), c]."""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def unrows(r: torch.Tensor, n: int, h: int, w: int) -> torch.Tensor:
    return r.reshape(n, h, w, -1).permute(0, 3, 1, 2).contiguous()
