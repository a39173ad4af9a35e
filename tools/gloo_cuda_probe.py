#!/usr/bin/env python3
"""Debug probe (GPU): is gloo's all-reduce of CUDA fp16 tensors (two ranks sharing cuda:0) exact and repeatable?"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    n = 12_599_920
    a = [(torch.randn(n, generator=torch.Generator().manual_seed(r)) * 3).half().to(dev) for r in range(2)]
    want = (a[0].float() + a[1].float()).half()
    bad, diff_runs = 0, 0
    first = None
    for it in range(6):
        t = a[rank].clone()
        junk = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)  # keep the stream busy in front of the collective
        t2 = t * 1.0  # produced by a kernel right before the collective
        work = dist.all_reduce(t2, async_op=True)
        other = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)
        work.wait()
        out = t2 + 0.0  # consumed by a kernel right after
        torch.cuda.synchronize()
        bad += int((out.view(torch.int16) != want.view(torch.int16)).sum().item())
        if first is None:
            first = out.clone()
        diff_runs += int((out.view(torch.int16) != first.view(torch.int16)).sum().item())
    q.put((rank, bad, diff_runs))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=w, args=(r, q)) for r in range(2)]
    [p.start() for p in ps]
    print("gloo CUDA fp16 all-reduce (rank, elements != exact sum over 6 runs, elements != first run):", [q.get(timeout=300) for _ in ps])
    [p.join() for p in ps]
