"""Host-side mirror of `base/pipelines/pipeline_videogen.py` (reference): `VideoGenPipeline`.

Keeps the reference call surface (`__call__` keywords, `prompt_embeds=` / `negative_prompt_embeds=` /
`latents=` / `generator=` / `callback`, pipeline_videogen.py:512-535) and its loop semantics
(662-689): CFG batch [negative | prompt], `u + s (c - u)`, DDPM ancestral step.  The loop body runs
on the MI355X: the UNet through liblavie_hip.so and CFG + scheduler step as one fused kernel.
CLIP text encoding and VAE decoding are NOT part of this package's compute path: pass any stock
PyTorch-ROCm `tokenizer`/`text_encoder`/`vae` objects to use them, or work with embeddings and
`output_type="latent"` (what the benchmark measures: video-latents/s)."""
from dataclasses import dataclass
import inspect
import os
import sys
from typing import Callable, List, Optional, Union

import torch

from . import ops
from .scheduling_ddpm import DDPMScheduler, randn_tensor


@dataclass
class StableDiffusionPipelineOutput:
    video: torch.Tensor


class VideoGenPipeline:
    # With guidance the UNet input is the latents twice (line 666): let the engine compute the layers in front of the first text
    # cross-attention once per step (UNet3DConditionModel.set_cfg_shared_input).  False = both halves computed, as the reference does.
    cfg_shared_prefix = True

    def __init__(self, vae=None, text_encoder=None, tokenizer=None, unet=None, scheduler=None, clip_model=None,
                 clip_processor=None):
        if unet is None:
            raise ValueError("unet is required")
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet = unet
        self.scheduler = scheduler or DDPMScheduler()
        self.vae_scale_factor = 8                        # 2 ** (len(vae.config.block_out_channels) - 1) for SD-1.x
        self._copy_stream = None

    def to(self, device):
        self.unet.to(device)
        for m in (self.vae, self.text_encoder):
            if m is not None:
                m.to(device)
        return self

    # ------------------------------------------------------------------ memory knobs of the reference pipeline object
    # base/pipelines/sample.py:72 calls enable_xformers_memory_efficient_attention() unconditionally; the diffusers base class also
    # offers attention slicing and the pipeline VAE slicing / tiling (pipeline_videogen.py:174-204).  They trade speed for memory in
    # the stock attention (`_sliced_attention`, xformers) and the stock VAE; here attention is always the fused flash-style HIP
    # kernel (scores are never materialised) and 288 GB of HBM make the VAE knobs moot, so they are accepted and change nothing.
    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        """No-op: attn1 / attn2 always run the fused online-softmax kernel (attention.hip); nothing to switch on."""
        return None

    def disable_xformers_memory_efficient_attention(self):
        return None

    def set_attention_slice(self, slice_size="auto"):
        """No-op (unet.py:297-360 slices the materialised score matrix; this engine has none)."""
        return None

    def enable_attention_slicing(self, slice_size="auto"):
        return self.set_attention_slice(slice_size)

    def disable_attention_slicing(self):
        return self.set_attention_slice(None)

    def enable_vae_slicing(self):
        """Forwarded to the attached VAE when it has the switch (a stock diffusers AutoencoderKL), else a no-op."""
        if self.vae is not None and hasattr(self.vae, "enable_slicing"):
            self.vae.enable_slicing()

    def disable_vae_slicing(self):
        if self.vae is not None and hasattr(self.vae, "disable_slicing"):
            self.vae.disable_slicing()

    def enable_vae_tiling(self):
        if self.vae is not None and hasattr(self.vae, "enable_tiling"):
            self.vae.enable_tiling()

    def disable_vae_tiling(self):
        if self.vae is not None and hasattr(self.vae, "disable_tiling"):
            self.vae.disable_tiling()

    # ------------------------------------------------------------------ the reference's YAML (base/configs/sample.yaml:16-40)
    @staticmethod
    def from_sample_yaml(path_or_dict, unet, vae=None, text_encoder=None, tokenizer=None):
        """Builds (pipeline, call_kwargs, cfg) from the keys base/pipelines/sample.py reads off its OmegaConf object:
        `sample_method` + `beta_start` / `beta_end` / `beta_schedule` choose and configure the scheduler (sample.py:44-63),
        `video_length`, `image_size`, `num_sampling_steps`, `guidance_scale` become the `__call__` keywords of sample.py:83-89,
        `seed` is returned in cfg for torch.manual_seed (sample.py:22-23).  `use_fp16` and
        `enable_xformers_memory_efficient_attention` are read by nobody in sample.py and are ignored here as well.
        PyYAML's safe loader replaces omegaconf (absent from the image); interpolation / `${...}` references are not resolved."""
        if isinstance(path_or_dict, dict):
            cfg = dict(path_or_dict)
        else:
            import yaml
            with open(path_or_dict) as f:
                cfg = yaml.safe_load(f) or {}
        method = cfg.get("sample_method", "ddpm")
        betas = dict(beta_start=float(cfg.get("beta_start", 1e-4)), beta_end=float(cfg.get("beta_end", 0.02)),
                     beta_schedule=cfg.get("beta_schedule", "linear"))
        if method == "ddpm":
            scheduler = DDPMScheduler(**betas)
        elif method == "ddim":
            from .scheduling_ddim import DDIMScheduler
            scheduler = DDIMScheduler(**betas)
        elif method == "eulerdiscrete":
            from .scheduling_euler_discrete import EulerDiscreteScheduler
            scheduler = EulerDiscreteScheduler(**betas)
        else:
            raise NotImplementedError(f"sample_method {method!r} (sample.py:44-63 knows ddim / eulerdiscrete / ddpm)")
        size = cfg.get("image_size", [320, 512])
        call_kwargs = dict(video_length=int(cfg.get("video_length", 16)), height=int(size[0]), width=int(size[1]),
                           num_inference_steps=int(cfg.get("num_sampling_steps", 50)),
                           guidance_scale=float(cfg.get("guidance_scale", 7.5)))
        pipe = VideoGenPipeline(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet, scheduler=scheduler)
        return pipe, call_kwargs, cfg

    @property
    def device(self):
        return self.unet.device

    # ------------------------------------------------------------------ prompt handling (273-420)
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds,
                       negative_prompt_embeds):
        if prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("no tokenizer/text_encoder attached: pass prompt_embeds / negative_prompt_embeds")
            ids = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length,
                                 truncation=True, return_tensors="pt").input_ids
            prompt_embeds = self.text_encoder(ids.to(device))[0]
        prompt_embeds = prompt_embeds.to(device=device)
        bs, n, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, n, -1)
        if not do_cfg:
            return prompt_embeds
        if negative_prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("no tokenizer/text_encoder attached: pass negative_prompt_embeds")
            neg = [""] * bs if negative_prompt is None else ([negative_prompt] if isinstance(negative_prompt, str) else negative_prompt)
            ids = self.tokenizer(neg, padding="max_length", max_length=n, truncation=True, return_tensors="pt").input_ids
            negative_prompt_embeds = self.text_encoder(ids.to(device))[0]
        negative_prompt_embeds = negative_prompt_embeds.to(device=device)
        negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, n, -1)
        return torch.cat([negative_prompt_embeds, prompt_embeds])           # line 418: unconditional half first

    def check_inputs(self, prompt, height, width, callback_steps, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly, but"
                             f" got: `prompt_embeds` {prompt_embeds.shape} != `negative_prompt_embeds` {negative_prompt_embeds.shape}.")

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
            latents = latents.to(device=device, dtype=dtype)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents(self, latents):
        """pipeline_videogen.py:422-429 (stock PyTorch-ROCm VAE; outside the latents/s metric)."""
        if self.vae is None:
            raise ValueError("no vae attached: use output_type='latent'")
        b, c, f, h, w = latents.shape
        frames = (latents / 0.18215).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
        video = self.vae.decode(frames.to(next(self.vae.parameters()).dtype)).sample
        video = video.reshape(b, f, *video.shape[1:]).permute(0, 1, 3, 4, 2)
        return ((video / 2 + 0.5) * 255).add_(0.5).clamp_(0, 255).to(dtype=torch.uint8).cpu().contiguous()

    # ------------------------------------------------------------------ the denoise loop (662-689)
    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, ctx: torch.Tensor, num_inference_steps: int, guidance_scale: float,
                generator=None, callback: Optional[Callable] = None, callback_steps: int = 1, eta: float = 0.0) -> torch.Tensor:
        """latents fp32 [P, C, F, h, w] on the device, ctx fp16 [2P, n, d] = [negative | prompt] (guidance_scale > 1) or
        [P, n, d] = prompt only (guidance_scale <= 1: no classifier-free guidance, :626) -> denoised fp32."""
        dev = latents.device
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        # Euler's timesteps are fractional (linspace) and reach the UNet as they are; DDPM / DDIM timesteps are integers
        fractional = bool(getattr(sch, "fractional_timesteps", False))
        timesteps = [float(t) if fractional else int(t) for t in sch.timesteps]
        # scheduler.scale_model_input (pipeline_videogen.py:667) as a scalar the fused kernel applies to the fp16 model input
        in_scale = getattr(sch, "model_input_scale", None)
        # `eta` goes to the scheduler only if its step takes one (DDIM), as prepare_extra_step_kwargs does
        # (pipeline_videogen.py:431-446); DDPM ignores it
        takes_eta = "eta" in inspect.signature(sch.coefficients).parameters
        do_cfg = guidance_scale > 1.0
        x = latents.to(torch.float32).contiguous().clone()
        p = x.shape[0]
        nb = 2 * p if do_cfg else p                        # model batch (:666)
        if ctx.shape[0] != nb:
            raise ValueError(f"ctx has {ctx.shape[0]} rows, expected {nb} for {p} latents at guidance_scale={guidance_scale}")
        model_in = torch.empty((nb,) + tuple(x.shape[1:]), dtype=torch.float16, device=dev)
        first_scale = in_scale(timesteps[0]) if in_scale else 1.0
        if do_cfg:
            ops.latents_to_model_input(x, model_in, first_scale)
        else:
            ops.latents_to_model_input1(x, model_in, first_scale)
        self.unet.prepare(nb, x.shape[2], x.shape[3], x.shape[4], ctx.shape[1])
        # the context is the same tensor for every step: its keys / values are computed once (the reference recomputes
        # them in each block of each step, attention.py:177-178)
        ctx = self.unet.cache_context(ctx) if hasattr(self.unet, "cache_context") else ctx

        # per-step noise: drawn on the host only when the caller's generator lives there, then staged through
        # two pinned slots on a side stream so that neither the device nor the host waits for the other
        gens = generator if isinstance(generator, list) else ([generator] if generator is not None else [])
        if isinstance(generator, list):                    # one generator per latent, as randn_tensor takes them (:504)
            if len(gens) != p:
                raise ValueError(f"got a list of {len(gens)} generators for {p} latents")
            if len({g.device.type for g in gens}) != 1:
                raise ValueError("a list of generators must live on one device type")
        host_noise = bool(gens) and gens[0].device.type == "cpu"
        noise_dev = torch.empty_like(x)
        if host_noise:
            pinned = [torch.empty(x.shape, dtype=torch.float32).pin_memory() for _ in range(2)]
            staged = [torch.empty_like(x) for _ in range(2)]
            copy_done = [None, None]        # slot's H2D copy finished  -> host may refill the pinned slot
            step_done = [None, None]        # slot's consumer finished  -> side stream may overwrite the device slot
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        t_dev = torch.tensor(timesteps, dtype=torch.float32, device=dev)

        # with guidance the two halves of model_in are the same latents (written by this loop's own step kernel): the engine may
        # compute the layers in front of the first text cross-attention once
        shared = do_cfg and self.cfg_shared_prefix and hasattr(self.unet, "set_cfg_shared_input")
        try:                                     # an exception in a callback or kernel must not leave the engine holding ctx
            if shared:
                self.unet.set_cfg_shared_input(True)
                if os.environ.get("LAVIE_DEBUG_CHECK_SHARED") == "1" and not torch.equal(model_in[:p], model_in[p:]):
                    raise RuntimeError("cfg_shared_prefix: the two halves of the model input differ")
            for i, t in enumerate(timesteps):
                eps = self.unet(model_in, t_dev[i], encoder_hidden_states=ctx).sample      # line 670
                coeffs = sch.coefficients(t, eta) if takes_eta else sch.coefficients(t)
                noise = None
                slot = i & 1
                if coeffs[4] != 0.0:            # the step adds noise (DDPM: every step but the last; DDIM: only with eta > 0)
                    if host_noise:
                        if copy_done[slot] is not None:
                            copy_done[slot].synchronize()
                        if isinstance(generator, list):
                            for j, g in enumerate(gens):
                                torch.randn(x.shape[1:], generator=g, dtype=torch.float32, out=pinned[slot][j])
                        else:
                            torch.randn(x.shape, generator=generator, dtype=torch.float32, out=pinned[slot])
                        if step_done[slot] is not None:
                            self._copy_stream.wait_event(step_done[slot])
                        with torch.cuda.stream(self._copy_stream):
                            staged[slot].copy_(pinned[slot], non_blocking=True)
                        copy_done[slot] = torch.cuda.Event()
                        copy_done[slot].record(self._copy_stream)
                        main.wait_event(copy_done[slot])
                        noise = staged[slot]
                    elif isinstance(generator, list):
                        for j, g in enumerate(gens):
                            noise_dev[j].normal_(generator=g)
                        noise = noise_dev
                    else:
                        noise = noise_dev.normal_(generator=generator) if generator is not None else noise_dev.normal_()
                next_scale = in_scale(timesteps[i + 1]) if in_scale and i + 1 < len(timesteps) else 1.0
                if do_cfg:
                    ops.cfg_ddpm_step(eps, x, noise, model_in, guidance_scale, coeffs, next_scale)   # lines 667, 679-683 fused
                else:
                    ops.sampler_step(eps, x, noise, model_in, coeffs, next_scale)                    # lines 667, 683
                if host_noise and coeffs[4] != 0.0:
                    step_done[slot] = torch.cuda.Event()
                    step_done[slot].record(main)
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, x)
        finally:
            # clean-up must not mask an exception raised inside the loop: each call is attempted, a failure of its own is re-raised
            # only when the loop itself finished
            pending = sys.exc_info()[1]
            cleanup_error = None
            for undo in ((lambda: self.unet.set_cfg_shared_input(False)) if shared else None,
                         (lambda: self.unet.cache_context(None)) if hasattr(self.unet, "cache_context") else None):
                if undo is None:
                    continue
                try:
                    undo()
                except Exception as e:      # noqa: BLE001
                    cleanup_error = cleanup_error or e
            if cleanup_error is not None and pending is None:
                raise cleanup_error
        return x

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, image_tensor=None, height: Optional[int] = None,
                 width: Optional[int] = None, video_length: int = 16, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True, callback=None, callback_steps: int = 1,
                 cross_attention_kwargs=None):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None:
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device = self.device
        do_cfg = guidance_scale > 1.0
        ctx = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt, prompt_embeds,
                                  negative_prompt_embeds).to(torch.float16).contiguous()
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, video_length,
                                       height, width, torch.float32, device, generator, latents)
        latents = self.denoise(latents, ctx, num_inference_steps, guidance_scale, generator, callback, callback_steps, eta)
        if output_type == "latent":
            video = latents
        else:
            video = self.decode_latents(latents)
        if not return_dict:
            return (video,)
        return StableDiffusionPipelineOutput(video=video)
