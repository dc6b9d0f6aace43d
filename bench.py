#!/usr/bin/env python
"""Headline benchmark: images/sec of the AmodalDAv2 ViT-L forward pass at 518x518, batch 32 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A step = one forward of the hot path over one synthetic batch that is already
resident in HBM (plus, for N > 1, the RCCL all-gather of the per-image depth maps -- the only exchange
the path has; images are independent so the batch shards with no other collective => weak scaling).
Rank 0 prints ONE JSON line.  Weights are deterministic synthetic ones (no checkpoints exist offline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "amodal-depth-anything_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

# algorithmic work per image, 518x518, MAC = 2 FLOP (counted on the reference with torch's flop counter)
GFLOP_PER_IMAGE = {"vits": 119.36, "vitb": 396.26, "vitl": 1389.65}
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md (never the 2:1-sparse figure)


def cpu_baseline(encoder, size, guide_type, loss):
    """The fp32 CPU oracle (a restatement of the reference's PyTorch path, pinned against it by tests/golden)
    timed on this box's host cores on a bounded sample: ONE forward of 2 images.  Baseline only."""
    from oracle import dav2_oracle as O
    from src.models import get_model
    from src.util.synth_weights import fill_state_dict_, make_inputs
    m = get_model("AmodalDAv2", guide_type=guide_type, loss_stategy=loss, encoder=encoder, pretrained=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    fill_state_dict_(sd, 0)
    nimg = 2
    x, _, mask, obs = make_inputs(nimg, size, size, 0)
    threads = torch.get_num_threads()
    t0 = time.perf_counter()
    O.amodal_forward(sd, encoder, guide_type, loss, x, None, mask, obs)
    dt = time.perf_counter() - t0
    return {"value": nimg / dt, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"1 forward of {nimg} synthetic {size}x{size} images, {encoder} fp32 torch CPU oracle ({dt:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--encoder", default="vitl")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=518)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true", help="skip the per-kernel HIP-event brackets")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or bool(os.environ.get("ADA_BENCH_FORCE_DIST"))  # the env switch exercises the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    import hip_ext
    from src.models import get_model
    from src.util.synth_weights import fill_state_dict_, make_inputs
    hip_ext.load()  # fails loudly if the HIP library is missing

    guide_type, loss = "mask+observation", "entire_target_object"
    model = get_model("AmodalDAv2", guide_type=guide_type, loss_stategy=loss, encoder=args.encoder, pretrained=False).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    fill_state_dict_(sd, 0)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev)
    B = args.batch
    x, _, mask, obs = make_inputs(B, args.size, args.size, seed=100 + rank, device=dev)
    gathered = torch.empty(world * B, 1, args.size, args.size, device=dev) if use_dist else None
    pending = [None]

    def step():
        with torch.no_grad():
            out = model(x, guide_rgb=None, guide_mask=mask, observation=obs)
        if use_dist:
            # per-image depth maps to every rank (north_star: RCCL all-gather over xGMI).  Issued asynchronously: the 34 MB
            # exchange of step i rides under the forward of step i+1; it is completed before the next one is issued and
            # before the timed region ends, so every step's gather is inside the measurement.
            if pending[0] is not None:
                pending[0].wait()
            pending[0] = dist.all_gather_into_tensor(gathered, out, async_op=True)
        return out

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    for _ in range(args.warmup):
        step()
    drain()
    timer = None
    if not args.no_kernel_timer:
        timer = hip_ext.KernelTimer()
        hip_ext.set_timer(timer)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    hip_ext.set_timer(None)
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        images = world * B * args.steps
        value = images / dt
        op_name = "f16" if hip_ext.operand_dtype() == torch.float16 else "bf16"
        line = {
            "metric": "images/sec at 518x518 bs=32 ViT-L (AmodalDAv2 forward)", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": op_name, "data": "synthetic",
            "config": {"workload": f"AmodalDAv2 {args.encoder} guide=mask+observation, {B} x 3x{args.size}x{args.size} RGB+mask+observation per GPU",
                       "global_batch": world * B, "parallelism": f"dp{world} (images sharded, all-gather of depth maps)" if world > 1 else "single GPU",
                       "weights": "deterministic synthetic (src/util/synth_weights.py)", "accumulate": "f32"},
        }
        gflop = GFLOP_PER_IMAGE.get(args.encoder)
        if gflop and args.size == 518:
            line["model_tflops"] = value * gflop / 1e3 / world  # per GPU, whole forward
        if timer is not None:
            summ = timer.summary()
            ig, at = summ.get("igemm"), summ.get("attention")
            if ig:
                ach = ig["work_total"] / (ig["ms_total"] * 1e-3) / 1e12
                line["roofline"] = {"bound": "mfma", "kernel": "igemm_kernel (all dense contractions: linear, conv3x3, convT, 1x1)",
                                    "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                                    "traffic": None, "launches": ig["launches"], "avg_launch_ms": ig["ms_avg"],
                                    "share_of_step": ig["ms_total"] / (1e3 * dt)}
            if at:
                ach = at["work_total"] / (at["ms_total"] * 1e-3) / 1e12
                line["roofline_attention"] = {"bound": "mfma", "kernel": "attention_kernel", "achieved": ach, "peak": MFMA_PEAK_TFLOPS,
                                              "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": None,
                                              "launches": at["launches"], "avg_launch_ms": at["ms_avg"],
                                              "share_of_step": at["ms_total"] / (1e3 * dt)}
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and "roofline" in line:
            try:  # HBM/L2-fabric bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json)
                t = json.load(open(pmc))
                line["roofline"]["traffic"] = t.get("igemm_bytes_per_launch")
                if "roofline_attention" in line:
                    line["roofline_attention"]["traffic"] = t.get("attention_bytes_per_launch")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.encoder, args.size, guide_type, loss)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
