#!/usr/bin/env python3
"""Headline benchmark: video-latents/sec of the LaVie base T2V denoising path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: spawns its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input = ONE fully denoised video
latent [4,16,40,64] per GPU: 50 DDPM steps x (CFG UNet forward at batch 2 + fused CFG/DDPM update)
(BASELINE.json configs[1]; reference loop pipeline_videogen.py:662-689).  Prompt-level data
parallelism: every rank denoises its own prompts (weak scaling), weights are broadcast once from
rank 0, finished latents are all-gathered inside the timed region.  Prints ONE JSON line on rank 0.

Extra objects on that line:
  roofline          the dominant kernel by device time, igemm_patch_kernel (halo-patch 3x3 convolution): algorithmic
                    FLOP per launch / average launch duration, both measured live with HIP events on the launch
                    stream around exactly that kernel inside the timed region, against the 2.5 PFLOP/s dense fp16
                    MFMA peak; `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes
  roofline_conv_class / roofline_linear_class   whole classes (all 3x3-conv / all plain-GEMM launches), from the extra
                    instrumented forward after the timed region (an event pair costs ~11 us of stream time)
  roofline_temporal the temporal-attention core (HBM-bound): algorithmic bytes 4*tokens*C*2 per launch / duration
  roofline_fused_*  the three row-resident level-0 kernels.  These and roofline_temporal carry launch-attached
                    events over five forwards right after the timed region: instrumenting them inside it cost 0.75 % of `value`
  cfg_shared_prefix the guided loop computes the layers in front of the first text cross-attention once per step (the two
                    halves of the CFG batch are the same latents); `both_halves_computed` = the same job without that, timed
                    after the headline region
  batched           k prompts per UNet forward (batch 2k), timed after the headline region; never `value`
  kernel_breakdown  every kernel class, from one extra instrumented UNet forward after the timed region
  cpu_baseline      the fp32 CPU oracle (kind "port") on this host's cores: 1 warm-up + 2 timed full CFG denoise steps
                    (BASELINE.md §4), extrapolated x50
  reference_order   the same job with none of the algebraic savings of the loop structure (both CFG halves computed in every layer,
                    Upsample3D as the 9-tap conv over the upsampled image): the reference's order of work, timed after the headline
  interp_forward    BASELINE.json configs[3] regression guard (round 4): the interpolation UNet at 61 frames, latent 40x64, guidance
                    batch 2 — ms per DDIM step (UNet forward + fused guidance / update), its temporal kernel's HBM fraction
  vsr_forward       BASELINE.json configs[4] regression guard: one 8-frame chunk of 320x512 latents through UNet3DVSRModel, ms per forward
                    (both legs: single-process runs only, after the headline region, never `value`; --headline-only / --no-extra-legs skip them)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAMES, LAT_H, LAT_W, CTX_LEN, CTX_DIM = 16, 40, 64, 77, 768
DDPM_STEPS, GUIDANCE = 50, 7.5
UNET_TFLOP = 16.219           # algorithmic TFLOP of one CFG forward at this config
ATTN2_KV_TFLOP = 0.0945       # of which: the attn2 to_k / to_v projections of the 77 text tokens in the 16 blocks.
                              # cache_context() runs them ONCE per prompt, not once per step: a video executes
                              # 50 x (UNET_TFLOP - ATTN2_KV_TFLOP) + ATTN2_KV_TFLOP, and that is what the whole-path rates count
CFG_SHARED_TFLOP = 0.3197     # of which NOT executed when the guided loop shares the layers in front of the first text cross-attention
                              # between the two halves of the batch (identical latents, pipeline_videogen.py:666): half of conv_in
                              # (1.9 GFLOP), down_blocks.0.resnets.0's two convs (302), and proj_in / qkv / self-attention of
                              # down_blocks.0.attentions.0 (16.8 + 50.3 + 268.4)
UPSAMPLE_PARITY_TFLOP = 0.7549  # of which NOT executed by the parity form of the three Upsample3D convs (conv3x3 of a nearest-x2 image = four
                              # 2x2 convs on the source with pre-summed weights: 4 C instead of 9 C multiply-adds per output element):
                              # 5/9 of 604 + 604 + 151 GFLOP
PEAK_MFMA_TFLOPS = 2500.0     # dense fp16, gfx950 (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0
CLASS_NAMES = ["conv3x3_igemm", "linear_igemm", "attention", "temporal_attention", "group_norm", "layer_norm", "other",
               "conv3x3_patch_kernel(subset of conv3x3_igemm)", "fused_temporal_sub_block", "fused_feed_forward",
               "fused_text_cross_attention_sub_block"]


def synth_inputs(idx, device):
    """SURVEY.md §8d synthetic inputs for prompt `idx` (CPU generators: device independent)."""
    g = torch.Generator().manual_seed(1000 + idx)
    pe = torch.randn(1, CTX_LEN, CTX_DIM, generator=g)
    ne = torch.randn(1, CTX_LEN, CTX_DIM, generator=g)
    lat = torch.randn(1, 4, FRAMES, LAT_H, LAT_W, generator=torch.Generator().manual_seed(2000 + idx))
    return pe.to(device), ne.to(device), lat.to(device)


def profile_begin(lib, mask, max_events):
    from lavie_amd import _lib
    _lib.check(lib.lavie_profile_begin(mask, max_events), "lavie_profile_begin")


def profile_end(lib):
    from lavie_amd import _lib
    n = len(CLASS_NAMES)
    launches = (ctypes.c_longlong * n)()
    ms, fl, by = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.lavie_profile_end(stream, launches, ms, fl, by), "lavie_profile_end")
    return [dict(name=CLASS_NAMES[i], launches=int(launches[i]), ms=ms[i], flops=fl[i], bytes=by[i]) for i in range(n)]


def host_cores():
    """CPU threads this process may really use: cgroup quota / affinity, not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd_fp32, threads, timed_steps=2):
    """the fp32 CPU oracle (kind "port": proven equal to the imported reference in the build container)
    on this host's cores, the SAME workload as the GPU run — full 16 frames at the 40x64 latent, CFG batch 2 — driven
    through the loop of pipeline_videogen.py:662-689: 1 warm-up iteration, then `timed_steps` consecutive iterations of
    (UNet forward at batch 2 + CFG combine + DDPM step), extrapolated x50 (every iteration runs the same graph)."""
    from oracle import unet_fp32 as O
    from oracle.ddpm import DDPMSchedule
    torch.set_num_threads(threads)
    pe, ne, lat = synth_inputs(0, "cpu")
    ctx = torch.cat([ne, pe])
    sch = DDPMSchedule()
    sch.set_timesteps(DDPM_STEPS)
    gen = torch.Generator().manual_seed(3000)
    x = lat.clone()
    times = []
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps[:1 + timed_steps]):
            t0 = time.perf_counter()
            eps = O.unet_forward(sd_fp32, torch.cat([x, x]), t, ctx)
            guided = eps[0:1] + GUIDANCE * (eps[1:2] - eps[0:1])
            x = sch.step(guided, t, x, torch.randn(x.shape, generator=gen))
            times.append(time.perf_counter() - t0)
    per_step = sum(times[1:]) / timed_steps
    per_video = per_step * DDPM_STEPS
    return {"value": 1.0 / per_video, "unit": "video-latents/s", "cores": torch.get_num_threads(), "kind": "port",
            "seconds_per_step": per_step, "seconds_sampled": sum(times), "step_seconds": [round(v, 2) for v in times],
            "sample": f"1 warm-up + {timed_steps} timed consecutive CFG denoise steps (fp32 UNet forward at batch 2, all {FRAMES} "
                      f"frames, 40x64 latent, + CFG + DDPM step) = {sum(times[1:]):.1f} s timed; x{DDPM_STEPS} steps extrapolated "
                      f"to {per_video:.0f} s per video-latent",
            "host_cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(), "cpu_model": cpu_model_string(),
            "torch_version": torch.__version__}


# ------------------------------------------------------------------ self-launch: python bench.py --gpus N without a launcher
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start N fresh rank processes (one per GPU) with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, exactly as torch.distributed.run would, wait for them, relay rank 0's
    JSON line and return non-zero if any rank failed.  The parent has not touched the GPU (nothing before this point
    initialises HIP) and never does: the children are ordinary child processes, not an exec of this one.
    Every child is polled: as soon as one exits non-zero the others are terminated (a rank that dies early would
    otherwise leave the rest in rendezvous or a collective until the backend's timeout), and they are terminated too
    if the parent is interrupted."""
    import subprocess
    import threading
    port = _free_port()
    procs = []
    lines0 = []

    def drain(pipe):                               # rank 0's stdout, read continuously so that the child never blocks on it
        for line in pipe:
            lines0.append(line)

    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        reader = threading.Thread(target=drain, args=(procs[0].stdout,), daemon=True)
        reader.start()
        failed = False
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                failed = True
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.5)
        if failed:
            time.sleep(1.0)                        # ranks failing together (bad flag, no GPU) report their own codes
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        reader.join(timeout=10)
    finally:
        for p in procs:                            # interrupted parent / unexpected error: leave no rank behind
            if p.poll() is None:
                p.kill()
    codes = [p.returncode for p in procs]
    for line in lines0:                            # the contract is ONE JSON line on stdout: library chatter (gloo prints
        if line.lstrip().startswith("{"):          # its rendezvous banner to stdout) goes to stderr
            sys.stdout.write(line if line.endswith("\n") else line + "\n")
        elif line.strip():
            sys.stderr.write(line if line.endswith("\n") else line + "\n")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write(f"bench.py: rank(s) failed (rank, exit code; negative = terminated after another rank failed): {bad}\n")
        return 1
    return 0


def launcher_selftest(rank, world, args):
    """--selftest-launcher (CPU, gloo; tests/test_bench_launcher.py): the distributed plumbing of the real run — init,
    weight broadcast, per-rank work, latent all_gather, barrier, MAX-over-ranks timing, one JSON line from rank 0 — around
    a stand-in for the denoiser.  Its line is labelled as such and carries no metric."""
    from lavie_amd import prompt_dp
    if os.environ.get("LAVIE_BENCH_SELFTEST_FAIL_RANK") == str(rank):     # test hook: this rank dies before the rendezvous
        return 3
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    shapes = {"w": (8, 4), "b": (8,)}
    sd = {k: torch.full(v, 0.5) for k, v in shapes.items()} if rank == 0 else None
    views = prompt_dp.broadcast_weights(shapes, sd, "cpu", dtype=torch.float32)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    outs = [views["w"].sum().reshape(1, 1) * (rank + world * i + 1) for i in range(args.steps)]
    gathered = prompt_dp.gather_latents(torch.cat(outs), [args.steps] * world)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "data": "stub (no denoiser ran; not a measurement)", "n_gpus": world,
                          "ranks_seen": len(gathered), "steps": args.steps,
                          "values": [float(v) for g in gathered for v in g.flatten()], "backend": "gloo"}))
    if world > 1:
        dist.destroy_process_group()
    return 0

def _class_rows(lib, fn):
    """one event-instrumented call of fn(): per-class rows"""
    profile_begin(lib, 0x7FF, 8192)
    fn()
    return profile_end(lib)


def interp_leg(lib, device, steps=5):
    """BASELINE.json configs[3]: interpolation UNet (sparse-causal attn1, FF -> temporal order, 8 input channels) at F = 61,
    latent 40x64, guidance batch 2, respaced DDIM (interpolation/configs/sample.yaml:14-37): `steps` timed DDIM steps after two
    warm-up steps, then one instrumented forward for the class table and the temporal kernel's HBM fraction."""
    from lavie_amd import spec, weights
    from lavie_amd.config import INTERPOLATION_CONFIG
    from lavie_amd.interpolation import UNet3DConditionModel as InterpUNet, create_diffusion
    F = 61
    sd = weights.synth_state_dict(spec.param_shapes(INTERPOLATION_CONFIG), 0)
    net = InterpUNet(init_weights=False, sample_size=64, in_channels=8, cross_attention_dim=CTX_DIM, use_first_frame=True)
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    del sd
    net = net.to(device, torch.float16)
    g = torch.Generator().manual_seed(0)
    z = torch.cat([torch.randn(1, 4, F, LAT_H, LAT_W, generator=g)] * 2).to(device)
    xs = torch.cat([torch.randn(1, 4, F, LAT_H, LAT_W, generator=g)] * 2).to(device)
    ctx = torch.randn(2, CTX_LEN, CTX_DIM, generator=g).to(device)
    d = create_diffusion("50")
    mk = dict(encoder_hidden_states=ctx, class_labels=None)
    d._ddim_loop_hip(net, z.shape, z, mk, 0.0, xs, True, max_steps=2)          # warm-up: packs weights, sizes the workspace
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = d._ddim_loop_hip(net, z.shape, z, mk, 0.0, xs, True, max_steps=steps)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    x8 = torch.cat([z, xs], dim=1).half()
    rows = _class_rows(lib, lambda: net(x8, 500, encoder_hidden_states=ctx.half()))
    temp = rows[3]
    res = {"workload": "BASELINE.json configs[3]: interpolation UNet, 61 frames x 320x512 (latent 8x61x40x64 in, 4 out), guidance batch 2, "
                       "one respaced-DDIM step = UNet forward + fused guidance/update, fp16, random-init 909M-param UNet",
           "interp_forward_ms": ms, "timed_steps": steps, "outputs_finite": bool(torch.isfinite(out).all()),
           "round2_value_ms": 90.3, "round2_source": "profiles/r02_interp_f61_bench_10steps.json",
           "classes": [dict(name=r["name"], launches=r["launches"], ms=round(r["ms"], 3)) for r in rows if r["launches"]]}
    if temp["launches"] and temp["ms"] > 0:
        bw = temp["bytes"] / (temp["ms"] * 1e-3) / 1e9
        res["temporal_kernel"] = {"kernel": "temporal_stream_kernel<NT = 4> (F = 61 padded to 64, plain softmax, no rotary / bias)",
                                  "bound": "hbm", "achieved": bw, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bw / PEAK_HBM_GBS,
                                  "launches": temp["launches"], "avg_launch_us": 1e3 * temp["ms"] / temp["launches"]}
    del net
    torch.cuda.empty_cache()
    return res


def vsr_leg(lib, device, iters=2):
    """BASELINE.json configs[4], the stage that is 82 % of the cascade: one 8-frame chunk (vsr/sample.py:99-129) of 320x512 latents
    through UNet3DVSRModel (vsr/configs/unet_3d_config.json, 691 M parameters), guidance batch 2: ms per forward."""
    from lavie_amd import spec, weights
    from lavie_amd.config import VSR_CONFIG
    from lavie_amd.vsr import UNet3DVSRModel
    F, H, W = 8, 320, 512
    sd = weights.synth_state_dict(spec.param_shapes(VSR_CONFIG), 0)
    net = UNet3DVSRModel(init_weights=False, sample_size=128, down_temporal_idx=(0, 1, 2, 3), mid_temporal=True, up_temporal_idx=(0, 1, 2, 3))
    net.load_state_dict({k: v.half() for k, v in sd.items()})
    del sd
    net = net.to(device, torch.float16)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, F, H, W, generator=g).half().to(device)
    low = torch.randn(2, 3, F, H, W, generator=g).half().to(device)
    ctx = torch.randn(2, CTX_LEN, 1024, generator=g).half().to(device)
    labels = torch.tensor([20, 20])
    out = net(x, 500, low, encoder_hidden_states=ctx, class_labels=labels).sample      # warm-up (packs weights, sizes the workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        out = net(x, 500 - i, low, encoder_hidden_states=ctx, class_labels=labels).sample
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    rows = _class_rows(lib, lambda: net(x, 400, low, encoder_hidden_states=ctx, class_labels=labels))
    res = {"workload": "BASELINE.json configs[4] (VSR stage, 82 % of the cascade): UNet3DVSRModel, one 8-frame chunk of 320x512 latents "
                       "(4 noisy + 3 low-res channels), guidance batch 2, fp16, random-init 691M-param UNet",
           "vsr_forward_ms": ms, "timed_forwards": iters, "outputs_finite": bool(torch.isfinite(out).all()),
           "round2_value_ms": 359.0, "round2_source": "profiles/r02_vsr_unet_bench_final.json",
           "classes": [dict(name=r["name"], launches=r["launches"], ms=round(r["ms"], 3)) for r in rows if r["launches"]]}
    del net
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed video-latents per GPU")
    ap.add_argument("--warmup", type=int, default=1, help="untimed video-latents per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CFG denoise steps of the CPU baseline after 1 warm-up (0 = skip)")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)      # CPU test of the rank plumbing
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = cgroup/affinity share)")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event roofline instrumentation")
    ap.add_argument("--graph", action="store_true",
                    help="replay the UNet forward from a hipGraph (A/B switch; implies --no-profile: events cannot be captured)")
    ap.add_argument("--ddpm-steps", type=int, default=DDPM_STEPS, help=argparse.SUPPRESS)   # debugging only
    ap.add_argument("--no-cfg-shared-prefix", action="store_true",
                    help="compute both halves of the CFG batch in every layer, as the reference does (default: the layers in front of "
                         "the first text cross-attention run once per step; the JSON line reports both rates)")
    ap.add_argument("--no-upsample-parity", action="store_true",
                    help="run the three Upsample3D convs as 9-tap convs over the (virtual) upsampled image, as the reference computes "
                         "them (default: four 2x2 convs on the source image with pre-summed weights, 2.25x fewer FLOP)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the legs timed after the headline region (both CFG halves computed; k prompts per forward): what "
                         "tools/collect_profiles.sh traces, so that per-kernel averages are those of the headline configuration")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the interpolation (configs[3]) and VSR (configs[4]) regression-guard legs that rank 0 runs after the headline")
    ap.add_argument("--prompts-per-forward", type=int, default=4,
                    help="after the headline (single-prompt) measurement, also time k prompts batched into ONE UNet forward "
                         "(batch 2k; SURVEY §8e 'batched B = 2k if memory-profitable'; BASELINE.json configs[2] readiness) and "
                         "report it as the extra object `batched`; 1 = skip")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one (before anything touches the GPU)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if args.selftest_launcher:
        raise SystemExit(launcher_selftest(rank, world, args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path exists); build with __graft_entry__.build() and run on the GPU box")
    # LAVIE_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): every rank uses device 0 and the collectives go
    # through gloo, because RCCL refuses two ranks on one device.  Real runs: one GPU per rank, backend nccl = RCCL.
    share = os.environ.get("LAVIE_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from lavie_amd import _lib, prompt_dp, spec, weights
    from lavie_amd.pipeline_videogen import VideoGenPipeline
    from lavie_amd.scheduling_ddpm import DDPMScheduler
    from lavie_amd.unet import UNet3DConditionModel
    lib = _lib.load()
    if os.environ.get("LAVIE_FORCE_TILE"):          # A/B timing only (tools/ab_bench.py): GEMM kernel selection override
        lib.lavie_debug_force_tile(int(os.environ["LAVIE_FORCE_TILE"], 0))

    # ---- model: random-init weights of the full architecture (909 M parameters), rank 0 -> RCCL broadcast
    shapes = spec.param_shapes()
    t_setup = time.perf_counter()
    sd32 = weights.synth_state_dict(shapes, args.seed) if rank == 0 else None
    views = prompt_dp.broadcast_weights(shapes, sd32, device)
    net = UNet3DConditionModel(sample_size=64, cross_attention_dim=CTX_DIM, init_weights=False)
    for name, p in net.named_parameters():
        p.data = views[name]
    net.prepare(2, FRAMES, LAT_H, LAT_W, CTX_LEN)
    if args.graph:
        net.enable_graph(True)
        args.no_profile = True
    pipe = VideoGenPipeline(unet=net, scheduler=DDPMScheduler(beta_start=1e-4, beta_end=0.02, beta_schedule="linear"))
    pipe.cfg_shared_prefix = not args.no_cfg_shared_prefix
    if args.no_upsample_parity:
        _lib.check(lib.lavie_debug_fused_mask(_lib.FUSED_DEFAULT & ~16), "lavie_debug_fused_mask")
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t_setup

    n_videos = args.warmup + args.steps
    inputs = [synth_inputs(rank + world * i, device) for i in range(n_videos)]
    torch.cuda.synchronize()

    def one_video(i):
        pe, ne, lat = inputs[i]
        gen = torch.Generator().manual_seed(3000 + rank + world * i)
        return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, height=LAT_H * 8, width=LAT_W * 8,
                    video_length=FRAMES, num_inference_steps=args.ddpm_steps, guidance_scale=GUIDANCE, generator=gen,
                    output_type="latent").video

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_video(i)
    barrier()

    use_prof = not args.no_profile
    if use_prof:   # the dominant kernel and temporal attention only, with events attached to the kernel launches themselves
        # (hipExtLaunchKernelGGL: no extra packets).  Scope-style hipEventRecord pairs cost ~11 us of stream time each:
        # instrumenting every class that way (~270 pairs per forward) slowed the timed region by 7 % (measured).
        # Only the dominant kernel is instrumented inside the timed region (19 launches per forward): with the temporal and the
        # three fused kernels as well (45) the events cost 0.75 % of the region (measured, same box: 1101.8 vs 1093.5 ms per video).
        profile_begin(lib, 1 << 7, 2 * 30 * args.ddpm_steps * args.steps + 1024)
    t0 = time.perf_counter()
    outs = [one_video(args.warmup + i) for i in range(args.steps)]
    local_lat = torch.cat(outs, dim=0).to(torch.float16)
    gathered = prompt_dp.gather_latents(local_lat, [args.steps] * world)       # the only data-path collective
    barrier()
    elapsed = time.perf_counter() - t0
    timed = profile_end(lib) if use_prof else None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = all(bool(torch.isfinite(g).all()) for g in gathered)

    # ---- extra: k prompts per UNet forward (batch 2k), what a rank of configs[2] (8 prompts per GPU) would run.  Separate
    # from the headline: the metric's configuration is ONE prompt at a time (configs[1]).
    batched = None
    kpf = 1 if args.headline_only else max(1, min(args.prompts_per_forward, 4))              # the engine takes UNet batches up to 8
    if kpf > 1:
        def one_batch(base):
            sets = [synth_inputs(rank + world * (base + j), device) for j in range(kpf)]
            gens = [torch.Generator().manual_seed(3000 + rank + world * (base + j)) for j in range(kpf)]
            return pipe(prompt_embeds=torch.cat([s_[0] for s_ in sets]), negative_prompt_embeds=torch.cat([s_[1] for s_ in sets]),
                        latents=torch.cat([s_[2] for s_ in sets]), height=LAT_H * 8, width=LAT_W * 8, video_length=FRAMES,
                        num_inference_steps=args.ddpm_steps, guidance_scale=GUIDANCE, generator=gens, output_type="latent").video
        nb = max(1, (args.steps + kpf - 1) // kpf)
        one_batch(0)                                             # untimed: workspace growth + warm-up at the batched shape
        barrier()
        tb = time.perf_counter()
        bouts = [one_batch(1000 + i * kpf) for i in range(nb)]
        barrier()
        bel = time.perf_counter() - tb
        if world > 1:
            tt = torch.tensor([bel], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bel = float(tt.item())
        batched = {"k": kpf, "unet_batch": 2 * kpf, "batches": nb, "value": nb * kpf * world / bel, "unit": "video-latents/s",
                   "ms_per_video": 1e3 * bel / (nb * kpf), "outputs_finite": all(bool(torch.isfinite(o).all()) for o in bouts),
                   "note": "k prompts share one UNet forward (reference loops prompts one at a time, sample.py:78-91); "
                           "not the headline configuration"}
        net.prepare(2, FRAMES, LAT_H, LAT_W, CTX_LEN)

    # ---- the same job with both halves of the CFG batch computed in every layer (the reference's order of work), timed after the
    # headline region on fewer videos: reported beside `value`, never as it
    both_halves = None
    if pipe.cfg_shared_prefix and not args.headline_only:
        nv = min(args.steps, 2)
        pipe.cfg_shared_prefix = False
        one_video(0)                     # untimed: the first video after the switch runs ~1.5 % slow (measured)
        barrier()
        t2 = time.perf_counter()
        for i in range(nv):
            one_video(args.warmup + i)
        barrier()
        el2 = time.perf_counter() - t2
        pipe.cfg_shared_prefix = True
        if world > 1:
            tt = torch.tensor([el2], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        both_halves = {"videos_per_rank": nv, "value": nv * world / el2, "ms_per_step": 1000.0 * el2 / nv}
    # ---- the reference's order of work throughout: both CFG halves in every layer AND Upsample3D as the 9-tap conv over the upsampled
    # image (the text K / V projections stay cached per prompt: no switch exists for that one, 0.09 of 16.2 TFLOP per forward)
    reference_order = None
    if not args.headline_only and (pipe.cfg_shared_prefix or not args.no_upsample_parity):
        nv = min(args.steps, 2)
        was_shared = pipe.cfg_shared_prefix
        pipe.cfg_shared_prefix = False
        _lib.check(lib.lavie_debug_fused_mask(_lib.FUSED_DEFAULT & ~16), "lavie_debug_fused_mask")
        net.prepare(2, FRAMES, LAT_H, LAT_W, CTX_LEN)
        one_video(0)
        barrier()
        t3 = time.perf_counter()
        for i in range(nv):
            one_video(args.warmup + i)
        barrier()
        el3 = time.perf_counter() - t3
        pipe.cfg_shared_prefix = was_shared
        _lib.check(lib.lavie_debug_fused_mask(_lib.FUSED_DEFAULT & ~16 if args.no_upsample_parity else _lib.FUSED_DEFAULT), "lavie_debug_fused_mask")
        net.prepare(2, FRAMES, LAT_H, LAT_W, CTX_LEN)
        if world > 1:
            tt = torch.tensor([el3], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el3 = float(tt.item())
        reference_order = {"videos_per_rank": nv, "value": nv * world / el3, "ms_per_step": 1000.0 * el3 / nv,
                           "what": "both CFG halves computed in every layer and Upsample3D as a 9-tap conv over the upsampled image "
                                   "(--no-cfg-shared-prefix --no-upsample-parity); text K / V still once per prompt"}
    total_videos = args.steps * world
    shared_tflop = CFG_SHARED_TFLOP if pipe.cfg_shared_prefix else 0.0
    parity_tflop = 0.0 if args.no_upsample_parity else UPSAMPLE_PARITY_TFLOP
    video_tflop = (UNET_TFLOP - ATTN2_KV_TFLOP - shared_tflop - parity_tflop) * args.ddpm_steps + ATTN2_KV_TFLOP      # text K/V once per prompt
    result = {
        "metric": "video-latents/sec (16f x 320x512, 50 DDPM steps)",
        "value": total_videos / elapsed,
        "unit": "video-latents/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: base T2V, one prompt per GPU at a time, 16x320x512 "
                               "(latent 4x16x40x64), DDPM 50 steps, CFG 7.5 (UNet batch 2), fp16, random-init 909M-param UNet",
                   "ddpm_steps": args.ddpm_steps, "guidance_scale": GUIDANCE, "latent": [4, FRAMES, LAT_H, LAT_W],
                   "prompts_per_gpu": args.steps, "parallelism": f"prompt-dp{world}" + ("-shared-gpu-rehearsal" if share else ""),
                   "collectives": "1 weight broadcast (setup), 1 latent all_gather (timed)"},
        "outputs_finite": finite,
        "hip_graph": bool(args.graph),
        "setup_seconds": setup_s,
        "achieved_tflops_whole_path": video_tflop * total_videos / elapsed / world,
        "mfma_fraction_whole_path": video_tflop * total_videos / elapsed / world / PEAK_MFMA_TFLOPS,
        "tflop_per_video_executed": video_tflop,
        "cfg_shared_prefix": {"enabled": bool(pipe.cfg_shared_prefix),
                              "what": "classifier-free guidance feeds the UNet the same latents twice (pipeline_videogen.py:666); conv_in, "
                                      "down_blocks.0.resnets.0 and GroupNorm / proj_in / self-attention of down_blocks.0.attentions.0 see no text, "
                                      "so the guided loop computes them for one half and copies (lavie_unet_set_cfg_shared_input); outputs equal "
                                      "the plain forward's to rounding (tests/test_gpu_engine.py)",
                              "tflop_not_executed_per_forward": shared_tflop,
                              "both_halves_computed": both_halves},
        "upsample_parity": {"enabled": not args.no_upsample_parity,
                            "what": "Upsample3D = nearest x2 then a 3x3 conv (resnet.py:44-79): the nine taps of an output pixel fall on 2 x 2 source "
                                    "pixels, so each output parity is a 2x2 conv on the source image with the coinciding taps' weights summed at "
                                    "load time (one extra fp16 rounding of the summed weights; parity tests vs F.conv2d on the upsampled image)",
                            "tflop_not_executed_per_forward": parity_tflop},
        "ranks_seen": dist.get_world_size() if world > 1 else 1,
        "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if world > 1 else "none (single process)",
    }
    if batched is not None:
        batched["vs_single_prompt"] = batched["value"] / result["value"]
        result["batched"] = batched
    if reference_order is not None:
        result["reference_order"] = reference_order

    def instrumented_forwards(mask, n):
        """n UNet forwards as the guided loop runs them (cached context, shared CFG prefix), event-instrumented; class rows."""
        pe_, ne_, lat_ = inputs[0]
        ctx_ = net.cache_context(torch.cat([ne_, pe_]).half().contiguous())      # text K / V once per prompt, as the denoise loop
        x2_ = torch.cat([lat_, lat_]).half().contiguous()
        net.set_cfg_shared_input(pipe.cfg_shared_prefix)
        net(x2_, 500, encoder_hidden_states=ctx_)
        torch.cuda.synchronize()
        profile_begin(lib, mask, 4096)
        t1_ = time.perf_counter()
        for _ in range(n):
            net(x2_, 500, encoder_hidden_states=ctx_)
        rows_ = profile_end(lib)
        ms_ = 1e3 * (time.perf_counter() - t1_) / n
        net.set_cfg_shared_input(False)
        net.cache_context(None)
        return rows_, ms_

    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # filled from rocprofv3 --pmc passes, see DESIGN.md
    if rank == 0 and timed is not None:
        # the other kernels with launch-attached events: five forwards right after the timed region (same process, same clocks)
        aux, _ = instrumented_forwards((1 << 3) | (1 << 8) | (1 << 9) | (1 << 10), 5)
        temp, conv = aux[3], timed[7]
        if conv["launches"]:
            a = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
            result["roofline"] = {"kernel": "igemm_patch_kernel<0> (3x3 conv stride 1, halo-patch implicit GEMM, MFMA 16x16x32 f16; "
                                            "every launch of this kernel in the timed region, all grid sizes)",
                                  "bound": "mfma", "achieved": a, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": a / PEAK_MFMA_TFLOPS, "traffic": None, "launches": conv["launches"],
                                  "avg_launch_us": 1e3 * conv["ms"] / conv["launches"],
                                  "flop_per_launch": conv["flops"] / conv["launches"]}
        if temp["launches"]:
            bw = temp["bytes"] / (temp["ms"] * 1e-3) / 1e9
            result["roofline_temporal"] = {"kernel": "temporal_stream_kernel (persistent, LDS-DMA two tiles ahead; levels 1-3 and mid: level 0 runs the fused kernel)",
                                           "source": "launch-attached events, five forwards right after the timed region", "bound": "hbm", "achieved": bw,
                                           "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": bw / PEAK_HBM_GBS,
                                           "traffic": None, "launches": temp["launches"],
                                           "avg_launch_us": 1e3 * temp["ms"] / temp["launches"],
                                           "bytes_per_launch": temp["bytes"] / temp["launches"]}
        for key, row, what, ref in (
                ("roofline_fused_temporal", aux[8],
                 "temporal_block_kernel (level 0: norm_temp + q|k|v projections + rotary / bias / softmax / PV + to_out + residual in ONE launch; "
                 "rows stay in registers, only weights cross LDS)", "attention.py:548-555, 580-667"),
                ("roofline_fused_feed_forward", aux[9],
                 "geglu_mlp_kernel (level 0: norm3 + ff1 + GEGLU + ff2 + residual in ONE launch; the [T, 4C] intermediate never exists)",
                 "attention.py:558"),
                ("roofline_fused_cross_attention", aux[10],
                 "cross_block_kernel (level 0: attn1.to_out + residual + norm2 + attn2.to_q + softmax(q K^T) V over the cached text keys + "
                 "attn2.to_out + residual in ONE launch; K / V travel in the weight stream)", "attention.py:513-534")):
            if row["launches"]:
                tf = row["flops"] / (row["ms"] * 1e-3) / 1e12
                bw = row["bytes"] / (row["ms"] * 1e-3) / 1e9
                result[key] = {"kernel": what, "replaces": ref, "source": "launch-attached events, five forwards right after the timed region",
                               "bound": "mfma", "achieved": tf, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / PEAK_MFMA_TFLOPS, "launches": row["launches"], "avg_launch_us": 1e3 * row["ms"] / row["launches"],
                               # SURVEY §8d's fused definition of the algorithmic bytes: x in + x' out + the weights once
                               "algorithmic_bytes_per_launch": row["bytes"] / row["launches"], "hbm_achieved_gbs": bw,
                               "hbm_frac": bw / PEAK_HBM_GBS}
        if os.path.isfile(pmc):
            tr = json.load(open(pmc))
            for key in ("roofline", "roofline_temporal", "roofline_fused_temporal", "roofline_fused_feed_forward", "roofline_fused_cross_attention"):
                if key in result and key in tr:
                    result[key]["traffic"] = tr[key]
                    result[key]["traffic_source"] = ("profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                      "of this command (tools/collect_profiles.sh), gfx950 corrections applied; not re-measured in this run")

    # ---- full per-class breakdown: one extra instrumented forward, outside the timed region
    if rank == 0 and use_prof:
        rows, fwd_ms = instrumented_forwards(0x7FF, 1)
        for key, row, kernels in (("roofline_conv_class", rows[0], "igemm_patch_kernel, igemm_pp_kernel<true>, igemm_kernel<..., true, ...>, splitk_reduce_kernel"),
                                  ("roofline_linear_class", rows[1], "igemm_ppx_kernel, igemm_pp_kernel<false>, igemm_kernel<..., false, ...>, splitk_reduce_kernel")):
            if row["launches"] and row["ms"] > 0:     # whole classes, from the instrumented forward AFTER the timed region
                a = row["flops"] / (row["ms"] * 1e-3) / 1e12
                result[key] = {"kernels": kernels, "source": "one instrumented UNet forward after the timed region", "bound": "mfma",
                               "achieved": a, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": a / PEAK_MFMA_TFLOPS,
                               "launches": row["launches"], "avg_launch_us": 1e3 * row["ms"] / row["launches"],
                               # the second roof: every operand element once (A, W, C [+ R]) / time against 8 TB/s.  The
                               # short-K (K = 320 / 640) half of the linear class is bound by this one, not by MFMA
                               "algorithmic_gbytes": row["bytes"] / 1e9,
                               "hbm_achieved_gbs": row["bytes"] / (row["ms"] * 1e-3) / 1e9,
                               "hbm_frac": row["bytes"] / (row["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS}
                pk = {"roofline_linear_class": "linear", "roofline_conv_class": "conv_class"}[key]
                if os.path.isfile(pmc) and pk in json.load(open(pmc)):      # HBM bytes per launch, committed rocprofv3 --pmc passes
                    result[key]["traffic"] = json.load(open(pmc))[pk]
        result["kernel_breakdown"] = {
            "unet_forward_ms_instrumented": fwd_ms,
            "classes": [dict(name=r["name"], launches=r["launches"], ms=round(r["ms"], 4),
                             tflops=(r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 and r["flops"] > 0 else None),
                             gbs=(r["bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else None)) for r in rows]}

    # ---- regression guards for the widened configurations (never `value`): BASELINE.json configs[3] and configs[4]
    if rank == 0 and world == 1 and not args.headline_only and not args.no_extra_legs:
        for key, leg in (("interp_forward", interp_leg), ("vsr_forward", vsr_leg)):
            try:
                result[key] = leg(lib, device)
            except Exception as e:     # a failing guard leg must not cost the headline line
                result[key] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0 and world == 1 and args.cpu_steps > 0:
        result["cpu_baseline"] = cpu_baseline({k: v.float() for k, v in sd32.items()}, args.cpu_threads or host_cores(),
                                              args.cpu_steps)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
