"""Host-side mirror of `vsr/models/pipeline_stable_diffusion_upscale_video_3d.py` (reference): the latent video upscaling
loop of the VSR stage, and the 8-frame chunk driver of `vsr/sample.py:104-123`.

`__call__` keeps the reference's keywords (prompt / image / num_inference_steps / guidance_scale / noise_level /
negative_prompt / eta / generator / latents / prompt_embeds / negative_prompt_embeds / output_type / callback).  The loop
(:706-735) runs the engine with the fused guidance + scheduler-step kernel: per step the UNet sees
[latents | noised low-res frames | 0] for the [negative | prompt] halves, then `lavie_cfg_sampler_step` applies guidance, the
scheduler update in its five-coefficient form and writes the next fp16 model input.  Text encoder and VAE are stock
PyTorch-ROCm objects a caller may attach (outside the latents metric): without them pass `prompt_embeds` and use
`output_type="latent"`."""
import inspect
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import torch

from .. import ops
from ..scheduling_ddpm import DDPMScheduler, randn_tensor


@dataclass
class StableDiffusionPipelineOutput:
    images: torch.Tensor
    nsfw_content_detected: Optional[list] = None


class VideoUpscalePipeline:
    def __init__(self, unet, scheduler, low_res_scheduler=None, vae=None, text_encoder=None, tokenizer=None,
                 max_noise_level: int = 350):
        self.unet, self.scheduler, self.vae, self.text_encoder, self.tokenizer = unet, scheduler, vae, text_encoder, tokenizer
        # the x4-upscaler's low_res_scheduler (its scheduler_config.json is not part of the reference tree): DDPM, cosine betas
        self.low_res_scheduler = low_res_scheduler or DDPMScheduler(beta_schedule="squaredcos_cap_v2")
        self.max_noise_level = max_noise_level

    @property
    def device(self):
        return self.unet.device

    def check_inputs(self, image, noise_level, callback_steps, prompt, prompt_embeds, negative_prompt_embeds):
        """pipeline_stable_diffusion_upscale_video_3d.py:376-439 (the checks that do not need a tokenizer)."""
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`")
        if prompt is not None and self.text_encoder is None:
            raise ValueError("no text encoder attached: pass prompt_embeds / negative_prompt_embeds")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape")
        if image is None:
            raise ValueError("`image` input cannot be undefined.")
        if noise_level > self.max_noise_level:
            raise ValueError(f"`noise_level` has to be <= {self.max_noise_level} but is {noise_level}")

    def prepare_latents_3d(self, batch_size, num_channels_latents, seq_len, height, width, dtype, device, generator, latents=None):
        """:441-452."""
        shape = (batch_size, num_channels_latents, seq_len, height, width)
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------ the denoise loop (:706-735)
    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, image: torch.Tensor, ctx: torch.Tensor, noise_level: int,
                num_inference_steps: int, guidance_scale: float, eta: float = 0.0, generator=None,
                callback: Optional[Callable] = None, callback_steps: int = 1) -> torch.Tensor:
        """latents fp32 [P, 4, F, h, w]; image = the NOISED low-res frames [P, 3, F, h, w]; ctx fp16 [2P, n, d] =
        [negative | prompt] -> denoised latents fp32."""
        if guidance_scale <= 1.0:
            raise NotImplementedError("guidance_scale <= 1 (no classifier-free guidance) is outside the fused MI355X loop")
        dev = latents.device
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        fractional = bool(getattr(sch, "fractional_timesteps", False))
        timesteps = [float(t) if fractional else int(t) for t in sch.timesteps]
        takes_eta = "eta" in inspect.signature(sch.coefficients).parameters
        in_scale = getattr(sch, "model_input_scale", None)
        x = latents.to(torch.float32).contiguous().clone()
        p = x.shape[0]
        model_in = torch.empty((2 * p,) + tuple(x.shape[1:]), dtype=torch.float16, device=dev)
        ops.latents_to_model_input(x, model_in, in_scale(timesteps[0]) if in_scale else 1.0)
        low = torch.cat([image, image], dim=0).to(device=dev, dtype=torch.float16).contiguous()       # :641
        labels = torch.full((2 * p,), int(noise_level), dtype=torch.int64)                              # :644
        noise_dev = torch.empty_like(x)
        t_dev = torch.tensor(timesteps, dtype=torch.float32, device=dev)
        self.unet.prepare(2 * p, x.shape[2], x.shape[3], x.shape[4], ctx.shape[1])
        ctx = self.unet.cache_context(ctx)       # text keys / values once per chunk, not once per block and step
        try:                                     # an exception in a callback or kernel must not leave the engine holding ctx
            for i, t in enumerate(timesteps):
                eps = self.unet(model_in, t_dev[i], low, encoder_hidden_states=ctx, class_labels=labels).sample        # :716-718
                coeffs = sch.coefficients(t, eta) if takes_eta else sch.coefficients(t)
                noise = None
                if coeffs[4] != 0.0:
                    noise = noise_dev.copy_(randn_tensor(x.shape, generator=generator, device=dev, dtype=torch.float32))
                next_scale = in_scale(timesteps[i + 1]) if in_scale and i + 1 < len(timesteps) else 1.0
                ops.cfg_ddpm_step(eps, x, noise, model_in, guidance_scale, coeffs, next_scale)                          # :721-726
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, x)
        finally:
            self.unet.cache_context(None)
        return x

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, image: Optional[torch.Tensor] = None,
                 num_inference_steps: int = 75, guidance_scale: float = 9.0, noise_level: int = 20, negative_prompt=None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents: Optional[torch.Tensor] = None, prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, output_type: Optional[str] = "latent",
                 return_dict: bool = True, callback=None, callback_steps: int = 1):
        self.check_inputs(image, noise_level, callback_steps, prompt, prompt_embeds, negative_prompt_embeds)
        if prompt is not None:
            raise NotImplementedError("text encoding is not built: pass prompt_embeds / negative_prompt_embeds")
        if negative_prompt_embeds is None:
            raise ValueError("negative_prompt_embeds is required for classifier-free guidance")
        device = self.device
        batch = prompt_embeds.shape[0] * (num_images_per_prompt or 1)
        ctx = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0).to(device, torch.float16).contiguous()
        image = image.to(device=device, dtype=torch.float32)
        # 5. add noise to the low-res frames at `noise_level` (:629-633)
        noise = randn_tensor(image.shape, generator=generator, device=device, dtype=torch.float32)
        image = self.low_res_scheduler.add_noise(image, noise, torch.tensor([noise_level] * image.shape[0]))
        seq_len, height, width = image.shape[2:]
        latents = self.prepare_latents_3d(batch, 4, seq_len, height, width, torch.float32, device, generator, latents)
        if 4 + image.shape[1] != self.unet.config.in_channels:                                       # :692-701
            raise ValueError(f"Incorrect configuration settings! The config of `pipeline.unet` expects "
                             f"{self.unet.config.in_channels} but received 4 + {image.shape[1]} channels")
        latents = self.denoise(latents, image, ctx, noise_level, num_inference_steps, guidance_scale, eta, generator, callback,
                               callback_steps)
        if output_type != "latent":
            if self.vae is None:
                raise ValueError("no vae attached: use output_type='latent'")
            b, c, f, h, w = latents.shape
            frames = latents.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
            scale = 1.0 / self.vae.config.scaling_factor                                                # :354-358
            dec = [self.vae.decode(frames[i:i + 4].to(next(self.vae.parameters()).dtype) * scale).sample.clamp(-1, 1).cpu()
                   for i in range(0, b * f, 4)]                                                        # :757-770, short_seq 4
            latents = torch.cat(dec, dim=0)
        if not return_dict:
            return (latents,)
        return StableDiffusionPipelineOutput(images=latents)


def upscale_in_chunks(pipeline: VideoUpscalePipeline, vframes: torch.Tensor, short_seq: int = 8, **kw) -> torch.Tensor:
    """vsr/sample.py:104-123: clips longer than `short_seq` frames go through the pipeline `short_seq` frames at a time
    (same generator across chunks); outputs are concatenated along the frame axis."""
    total = vframes.shape[2]
    if total <= short_seq:
        return pipeline(image=vframes, **kw).images
    outs = [pipeline(image=vframes[:, :, s:min(total, s + short_seq)], **kw).images for s in range(0, total, short_seq)]
    return torch.cat(outs, dim=2)
