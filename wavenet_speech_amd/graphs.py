"""
HIP-graph capture of a whole training step.

At the small configurations (BASELINE configs[1]: 128 channels x 11 blocks x 4096 steps) a training step is ~280 kernel
launches of 5-50 us each and the HOST -- Python, ctypes, autograd bookkeeping, the HIP launch path -- needs ~5 ms to issue
them: longer than the GPU needs to run them.  The step has no data-dependent control flow (fixed shapes, device-side error
flags, a device-side dynamic gradient scale), so it is captured once into a HIP graph and replayed: one host call per step.

    step = GraphedStep(lambda: loss_fn(model(x_static)), params=model.parameters(), optimizer=opt, sync=sync)
    for batch in loader:
        x_static.copy_(batch)        # inputs are static tensors: refill them, then
        loss = step()                # replay forward + backward (+ gradient gather) [+ eager all-reduce] + optimizer

What is captured is exactly what the eager step launches (the same HIP kernels through the same C ABI on torch's capture
stream), so results are bitwise those of the eager step.  torch supplies the capture machinery (torch.cuda.CUDAGraph is
hipGraph on ROCm) and the graph-private memory pool; nothing here is a tracing compiler.
"""
import torch

from . import _flags, series


class GraphedStep(object):
    def __init__(self, forward_loss, params, optimizer=None, sync=None, warmup=3, capture_optimizer=True):
        """forward_loss(): builds the autograd graph on STATIC input tensors and returns the scalar loss.
        sync: optional parallel.FlatGradAllReduce (its gather is captured; the all-reduce itself runs eagerly between the two
        graphs when a process group exists).  optimizer: stepped inside a second graph when it is capturable (torch.optim.*
        with capturable=True), eagerly otherwise."""
        self.forward_loss, self.optimizer, self.sync = forward_loss, optimizer, sync
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = torch.cuda.CUDAGraph()
        self.opt_graph = None
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):
            # warm-up on the capture stream: the series pool, the allocator and every lazily built plan reach their steady
            # state here, so the capture holds launches only (a pooled buffer taken during capture is not re-zeroed)
            for _ in range(max(1, warmup)):
                self._eager_step()
            torch.cuda.current_stream().synchronize()
            self._zero()
            # pooled series buffers are allocated outside the graph's private pool: the graph keeps its own references, so the
            # pool's idle-cap eviction can never free memory a replay still addresses
            self.buffers = []
            series.POOL.hold = self.buffers
            try:
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.loss = self._fwd_bwd()
            finally:
                series.POOL.hold = None
            capturable = optimizer is not None and capture_optimizer and all(g.get("capturable", False) for g in optimizer.param_groups)
            if capturable:
                self.opt_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.opt_graph, stream=self.stream):
                    optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        self.distributed = distributed

    def _zero(self):
        if self.sync is not None:
            self.sync.zero()
        else:
            for p in self.params:
                p.grad = None

    def _fwd_bwd(self):
        loss = self.forward_loss()
        loss.backward()
        if self.sync is not None:
            self.sync.gather()
        return loss.detach()

    def _eager_step(self):
        self._zero()
        self._fwd_bwd()
        if self.sync is not None:
            self.sync.reduce_gathered()
        if self.optimizer is not None:
            self.optimizer.step()

    def replay_forward_backward(self):
        self.graph.replay()
        return self.loss

    def reduce(self):
        if self.sync is not None:
            self.sync.reduce_gathered()

    def step_optimizer(self):
        if self.opt_graph is not None:
            self.opt_graph.replay()
        elif self.optimizer is not None:
            self.optimizer.step()

    def __call__(self):
        self.replay_forward_backward()
        self.reduce()
        self.step_optimizer()
        return self.loss

    def check(self):
        """raise if a device-side error flag (fp16 overflow, bad labels) was set by a replayed step"""
        _flags.check_device_flags()
