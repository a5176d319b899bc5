"""CPU, world_size=2 over gloo: the data-parallel gradient reducer averages gradients across ranks
and leaves every rank with identical parameters after an optimizer step."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, bucket_mb, comm_dtype, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from holocron_amd.parallel import GradReducer, broadcast_parameters
    torch.manual_seed(100 + rank)                      # different init per rank on purpose
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    broadcast_parameters(model, 0)
    red = GradReducer(model.parameters(), bucket_mb=bucket_mb, comm_dtype=comm_dtype)
    assert len(red.buckets) >= 1
    torch.manual_seed(7)
    data = torch.randn(2 * world, 8)
    target = torch.randn(2 * world, 4)
    x, t = data[rank * 2:(rank + 1) * 2], target[rank * 2:(rank + 1) * 2]
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        loss = ((model(x) - t) ** 2).sum()
        loss.backward()
        red.finalize()
        with torch.no_grad():
            for p in model.parameters():
                p -= 0.01 * p.grad
    # serial reference: the full batch on one process, gradient / world
    if rank == 0:
        torch.manual_seed(100)
        ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        for _ in range(2):
            for p in ref.parameters():
                p.grad = None
            (((ref(data) - target) ** 2).sum() / world).backward()
            with torch.no_grad():
                for p in ref.parameters():
                    p -= 0.01 * p.grad
        tol = 1e-5 if comm_dtype == torch.float32 else 2e-2
        for a, b in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(a, b, atol=tol, rtol=tol), (a - b).abs().max()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    dist.destroy_process_group()


def _run(bucket_mb, comm_dtype):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, bucket_mb, comm_dtype, None), nprocs=2, join=True)


def test_grad_reducer_world2_single_bucket():
    _run(32.0, torch.float32)


def test_grad_reducer_world2_many_small_buckets():
    _run(0.0001, torch.float32)


def test_grad_reducer_world2_bf16_comm():
    _run(0.0002, torch.bfloat16)
