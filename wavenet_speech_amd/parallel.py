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


class _Done(object):
    def wait(self):
        return True


class _PendingAverage(object):
    """handle of an asynchronous flat-gradient all-reduce: wait() finishes the collective, then divides by world"""

    def __init__(self, work, flat, scale):
        self.work, self.flat, self.scale = work, flat, scale

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.scale != 1:
                self.flat.div_(self.scale)
        return True


class FlatGradAllReduce(object):
    """All parameter gradients of a replica in ONE flat buffer, all-reduced with ONE collective.

        sync = FlatGradAllReduce(model.parameters())
        for batch in ...:
            sync.zero()                 # instead of optimizer.zero_grad(): grads -> None (no memset, and autograd then
                                        #   ASSIGNS each gradient instead of launching one add kernel per parameter)
            loss(model(batch)).backward()
            sync.reduce()               # gather into the flat buffer (one multi-tensor copy), all-reduce (sum),
                                        #   / world, and point every p.grad at its slice of the buffer
            optimizer.step()

    With each rank's loss averaged over its own shard, the result equals the gradient of the global-batch mean loss.
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
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self):
        for p in self.params:
            p.grad = None

    def world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def gather(self):
        """copy the freshly computed gradients into the flat buffer and re-point p.grad at the buffer.  A parameter that received NO
        gradient this step (the last block's unused residual path: autograd leaves it None in the reference too) keeps
        p.grad = None -- replicas are identical, so it is None on every rank -- and only its slice of the flat buffer is zeroed:
        a weight-decay optimizer must not start moving parameters the single-process path never touches."""
        src, dst, live = [], [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()                     # parameter did not take part in this step
                live.append(False)
                continue
            live.append(True)
            if p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        for p, v, on in zip(self.params, self.views, live):
            p.grad = v if on else None

    def reduce(self, async_op=False):
        """gather + all-reduce (+ 1/world).  async_op=True returns a handle whose wait() completes the collective AND
        applies the 1/world average (call it before optimizer.step()); otherwise returns None when done."""
        self.gather()
        return self.reduce_gathered(async_op)

    def reduce_gathered(self, async_op=False):
        """the collective half of reduce(): all-reduce (+ 1/world) of a flat buffer that gather() has already filled (the
        gather of a HIP-graph-captured step is part of the graph, the all-reduce runs between the graphs: graphs.GraphedStep)"""
        if not (dist.is_available() and dist.is_initialized()):
            return _Done() if async_op else None
        w = self.world()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        scale = w if (self.average and w > 1) else 1
        if async_op:
            return _PendingAverage(work, self.flat, scale)
        if scale != 1:
            self.flat.div_(scale)
        return None

    def payload_bytes(self):
        return self.flat.numel() * self.flat.element_size()
