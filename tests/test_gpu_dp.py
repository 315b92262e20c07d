"""GPU, world_size 2 on ONE card: the data-parallel step with the real HIP modules.  RCCL refuses two ranks on one device,
so the collective here is gloo (ProcessGroupGloo stages device tensors through the host); what is under test is the
replica / shard / flat-gradient logic around the HIP kernels, with the same FlatGradAllReduce object bench.py drives over
RCCL.  Two processes use the card at once (the box allows six)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu
LAYERS = [(32, 32, 2, 2 ** i) for i in range(4)]
GLOBAL_BATCH, L = 6, 300


def _net():
    from wavenet_speech_amd.modules.wavenet import WaveNet
    torch.manual_seed(5)
    return WaveNet(32, 2, LAYERS, 32, softmax=False)


def _data():
    g = torch.Generator().manual_seed(21)
    q = torch.randint(0, 32, (GLOBAL_BATCH, L), generator=g)
    return O.one_hot_encoding(q, 32), torch.randn(GLOBAL_BATCH, 32, L, generator=g)


def _worker(rank, world, port, out):
    from wavenet_speech_amd.parallel import FlatGradAllReduce, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    net = _net().to(dev)
    sync = FlatGradAllReduce(net.parameters())
    x, cot = _data()
    b, e = shard_bounds(GLOBAL_BATCH, rank, world)
    xs, cs = x[b:e].to(dev), cot[b:e].to(dev)
    for _ in range(2):   # two steps through the same object: the flat buffer is re-pointed every time
        sync.zero()
        ((net(xs) * cs).sum() / (e - b)).backward()
        sync.reduce()
    flat = sync.flat.cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)      # replicas agree bit for bit
    if rank == 0:
        torch.save({k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}, out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    # single process, whole batch, loss averaged over it: (a) the HIP path itself, (b) the CPU oracle
    net = _net()
    sd = {k: v.clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x, cot = _data()
    net = net.to("cuda:0")
    slopes, remove = O.capture_leaky_slopes(net)
    y = net(x.to("cuda:0"))
    remove()
    ((y * cot.to("cuda:0")).sum() / GLOBAL_BATCH).backward()
    (O.wavenet(x, sd, LAYERS, False, impl="aten", slopes=slopes) * cot).sum().div(GLOBAL_BATCH).backward()
    for k, p in net.named_parameters():
        if sd[k].grad is None:
            assert k not in got and p.grad is None, k      # no gradient in the reference <=> None here, with or without data parallelism
            continue
        assert O.rel_err(got[k], p.grad.cpu()) < 1e-5, ("vs single-process HIP", k)
        assert O.rel_err(got[k], sd[k].grad) < 1e-4, ("vs oracle", k)
