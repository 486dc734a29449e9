"""Probe: can two RCCL ranks share one GPU?  (decides how the overlap test is built)"""
import os, socket, sys, time
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group('nccl', rank=rank, world_size=world)
        x = torch.ones(1 << 20, device='cuda') * (rank + 1)
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print(rank, 'nccl ok', x[0].item(), flush=True)
    except Exception as e:
        print(rank, 'nccl failed:', repr(e)[:300], flush=True)


if __name__ == '__main__':
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
