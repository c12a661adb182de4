#!/usr/bin/env python
"""Config 4: ONE huge class trained data-parallel, N1 = 784-450-(10,10)-450-784, adam, global batch 65 536.

    python tools/bench_dp.py --rows-per-gpu 262144 --epochs 2                      # 1 GPU (world 1: no collective)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
           tools/bench_dp.py --rows-per-gpu 1250000 --epochs 2                     # 8 GPUs: 10M x 784 in all

Every rank generates its row stripes of every global mini-batch on its own GPU (the 31 GB matrix never exists on one
device), then runs smrt_amd.autoencode.DataParallelTrainer: smrt_vae_grad_step_async -> all_reduce(SUM) over RCCL ->
smrt_vae_apply_grads_async per step, cost read once per epoch.  Rank 0 prints one JSON line (whole-job samples/s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

D, HIDDEN, Z = 784, [450], 10
FLOP_PER_SAMPLE = 6 * (784 * 450 + 2 * 450 * 10 + 10 * 450 + 450 * 784) - 2 * 784 * 450


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-gpu", type=int, default=262144)
    ap.add_argument("--global-batch", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SMRT_BENCH_BACKEND", "nccl")      # gloo: rehearse several ranks on ONE GPU
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from smrt_amd import _hip
    from smrt_amd.autoencode.data_parallel import DataParallelTrainer, plan_stripes
    from smrt_amd.autoencode.topography import VAETopography

    n_rows = args.rows_per_gpu * world
    plan = plan_stripes(n_rows, args.global_batch, world)
    n_local = sum(int(b[rank + 1] - b[rank]) for _, _, b in plan)
    g = torch.Generator(device=dev).manual_seed(42 + rank)
    X = (torch.rand((n_local, D), generator=g, device=dev) * (torch.rand((n_local, D), generator=g, device=dev) > 0.8)).float()
    sess = _hip.VaeHandle(n_features=D, hidden=HIDDEN, n_latent=Z, activation="sigmoid", optimizer="adam", learning_rate=0.001,
                          l2_penalty=1e-4, clip=True, gclip_min=-5.0, gclip_max=5.0, eps=1e-10, dropout=1.0, seed=42,
                          max_batch=(args.global_batch + world - 1) // world + 1, device=local)
    topo = VAETopography(n_hidden=HIDDEN, input_shape=D, activation="sigmoid", n_latent_factors=Z, random_state=42)
    sess.set_params(VAETopography.join_flat(topo.initial_parameters()))
    tr = DataParallelTrainer(sess, (X, D), n_rows, args.global_batch, rank, world, dist if world > 1 else None)
    tr.run_epochs(1)                                        # warm-up
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    costs, secs = tr.run_epochs(args.epochs)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        sps = n_rows * args.epochs / dt
        print(json.dumps({"metric": "VAE train samples/sec (data-parallel, config 4)", "value": sps, "unit": "samples/s",
                          "n_gpus": world, "rows": n_rows, "global_batch": args.global_batch, "epochs": args.epochs,
                          "steps_per_epoch": len(plan), "ms_per_step": 1e3 * dt / (args.epochs * len(plan)),
                          "tflops_per_gpu": sps * FLOP_PER_SAMPLE / 1e12 / world,
                          "allreduce_floats": int(sess.dp_count), "final_cost": float(costs[-1]), "dtype": "f32",
                          "scaling": "weak"}))
    sess.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
