"""Input side of the hot path: a pinned-memory, double-buffered host->device feed for
batches of 40^3 TSDF grids, so that a data-loader bound loop (scripts/train_giga.py:198-211, where every batch is
`.to(device)`-copied from pageable memory inside the step, train_giga.py:141-151) overlaps the next batch's PCIe
transfer with the current step's kernels.  The grids are what `vgn.io.read_voxel_grid` /
`DatasetVoxelOccFile.__getitem__` return (io.py:97-99, dataset_voxel.py:69-93): float32 (1,40,40,40) per scene."""
import numpy as np
import torch

from . import _capi


class TSDFFeed:
    """Iterates over an iterable of host batches (numpy arrays or CPU tensors, or tuples/lists of them) and
    yields the same structure as device tensors.  Two pinned staging slots per leaf; copies run on a side HIP
    stream and the consumer's stream waits on the copy's event, never on the host."""

    def __init__(self, batches, device=None, depth=2):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _capi.GigaHipError("TSDFFeed stages batches for a HIP device; there is no CPU path")
        self._it = iter(batches)
        self._depth = max(2, int(depth))
        self._stream = torch.cuda.Stream(self.device)
        self._slots = [dict() for _ in range(self._depth)]
        self._queue = []
        self._n = 0

    def _stage_leaf(self, slot, path, a):
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a.contiguous()
        pin = slot.get(path)
        if pin is None or pin.shape != t.shape or pin.dtype != t.dtype:
            pin = slot[path] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        pin.copy_(t)                                      # pageable -> pinned (host memcpy)
        return pin.to(self.device, non_blocking=True)     # pinned -> device, asynchronous on the side stream

    def _stage(self, slot, path, item):
        if isinstance(item, (tuple, list)):
            return type(item)(self._stage(slot, path + (i,), x) for i, x in enumerate(item))
        return self._stage_leaf(slot, path, item)

    def _prefetch(self):
        try:
            item = next(self._it)
        except StopIteration:
            return False
        slot = self._slots[self._n % self._depth]
        ready = slot.get("_free")
        if ready is not None:
            ready.synchronize()                           # the copy that last used these pinned buffers is done
        with torch.cuda.stream(self._stream):
            dev_item = self._stage(slot, (), item)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        slot["_free"] = ev
        self._queue.append((dev_item, ev))
        self._n += 1
        return True

    def __iter__(self):
        while len(self._queue) < self._depth - 1 and self._prefetch():
            pass
        while self._queue:
            self._prefetch()                              # keep one transfer in flight behind the consumer
            dev_item, ev = self._queue.pop(0)
            torch.cuda.current_stream(self.device).wait_event(ev)
            _record_stream(dev_item, torch.cuda.current_stream(self.device))
            yield dev_item


def _record_stream(item, stream):
    if isinstance(item, (tuple, list)):
        for x in item:
            _record_stream(x, stream)
    else:
        item.record_stream(stream)
