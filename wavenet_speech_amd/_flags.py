"""
Device-side error flags without host synchronisation.

Several kernels refuse bad input on the device -- an fp16 store that overflows, a class label or quantisation level outside
its range -- by raising a device counter (and poisoning their result with NaN / contributing nothing; none of them reads
out of bounds).  Turning the counter into a Python exception needs its value on the host.  Reading it with .item() stalls
the stream: at configs[2] in the fp16 modes the two reads per training step (end of forward, end of backward) left the GPU
idle for 3.5 ms of a 71 ms step while the host caught up with its launches.

So in TRAINING calls the counter is copied to pinned host memory asynchronously and looked at by the next call that passes
through here -- or by check_device_flags(), which waits.  The exception arrives one call late; the step that tripped it has
NaN / inf results of its own.  Calls outside autograd (inference) check at once: their results are consumed directly.
WN_FLAG_CHECK=sync checks at once everywhere.

Under HIP-graph capture (graphs.GraphedStep) nothing may read back or record host-visible events: a flag noted while the
current stream is capturing is only REMEMBERED (the tensor lives in the graph's memory and is rewritten by every replay);
check_device_flags() reads the remembered ones too.
"""
import os

import torch


class _Watch(object):
    SLOTS = 128

    def __init__(self):
        self.pinned = None
        self.next = 0
        self.pending = []          # (slot, event, message)
        self.captured = []         # (flag tensor in graph memory, message): flags of captured steps, re-read on demand

    def note(self, flag, message, at_once):
        """flag: int32 device tensor [1]; message: str or callable(count) -> str"""
        if flag is None:
            return
        if flag.is_cuda and torch.cuda.is_current_stream_capturing():
            self.captured.append((flag, message))
            return
        if at_once or os.environ.get("WN_FLAG_CHECK") == "sync":
            n = int(flag.item())
            if n:
                raise RuntimeError(message(n) if callable(message) else message)
            return
        if self.pinned is None:
            self.pinned = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
        if len(self.pending) >= self.SLOTS - 1:
            self.poll(wait=True)
        slot = self.next
        self.next = (self.next + 1) % self.SLOTS
        self.pinned[slot:slot + 1].copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((slot, ev, message))

    def poll(self, wait=False):
        if not wait and self.pending and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return                     # no event queries from inside a capture
        keep, hit = [], None
        for slot, ev, message in self.pending:
            if wait:
                ev.synchronize()
            if wait or ev.query():
                n = int(self.pinned[slot])
                if n and hit is None:
                    hit = message(n) if callable(message) else message
            else:
                keep.append((slot, ev, message))
        self.pending = keep
        if wait and hit is None:
            for flag, message in self.captured:
                n = int(flag.item())
                if n:
                    hit = message(n) if callable(message) else message
                    break
        if hit is not None:
            raise RuntimeError(hit + " [reported after the call that tripped it]")


WATCH = _Watch()


def check_device_flags():
    """wait for every outstanding device flag (fp16 overflow, labels / levels out of range) and raise if one is set"""
    WATCH.poll(wait=True)
