"""
Utterance-level data parallelism (new: the reference is single-process).

One process per GPU; every rank holds a full replica and its own shard of the minibatch.  The only exchange
step of a training iteration is the gradient all-reduce: all parameter gradients live as views into ONE flat
fp32 buffer, so a step issues a single RCCL all-reduce (64 MB at 256 ch x 30 blocks) over xGMI -- ring
collectives on xGMI are per-link bound, so one large message beats many small ones -- followed by a 1/world
scale.  `torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(global_batch, rank, world):
    """[begin, end) of the utterances rank `rank` owns; the remainder goes to the lowest ranks."""
    base, extra = divmod(global_batch, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class FlatGradAllReduce(object):
    """Keeps `p.grad` of every parameter as a view into one flat buffer and all-reduces that buffer.

        sync = FlatGradAllReduce(model.parameters())
        for batch in ...:
            sync.zero()                 # instead of optimizer.zero_grad()
            loss(model(batch)).backward()
            sync.reduce()               # sum over ranks, then / world  (== grad of the global-batch mean loss
                                        #  when each rank's loss is averaged over its own shard)
            optimizer.step()
    """

    def __init__(self, params, group=None, average=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.average = average
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(n, dtype=ref.dtype, device=ref.device)
        off = 0
        self.views = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v  # someone (e.g. zero_grad(set_to_none=True)) detached the view: re-attach

    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def reduce(self, async_op=False):
        for p, v in zip(self.params, self.views):  # autograd replaced a view (first backward after None): fold it back
            if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        if not (dist.is_available() and dist.is_initialized()):
            return None
        w = self.world()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            return work
        if self.average and w > 1:
            self.flat.div_(w)
        return None

    def payload_bytes(self):
        return self.flat.numel() * self.flat.element_size()
