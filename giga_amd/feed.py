"""Input side of the hot path: a pinned-memory, ring-buffered host->device feed for batches of
40^3 TSDF grids, so that a data-loader bound loop (scripts/train_giga.py:198-211, where every batch is `.to(device)`-copied
from pageable memory inside the step, train_giga.py:141-151) overlaps the next batches' PCIe transfers with the current
step's kernels.  The grids are what `vgn.io.read_voxel_grid` / `DatasetVoxelOccFile.__getitem__` return (io.py:97-99,
dataset_voxel.py:69-93): float32 (1,40,40,40) per scene; `giga_amd.dataset.GraspOccBatches` produces whole batches.

Design (what the measurements in tools/gpu_feed_bench.py forced):
  * a staging THREAD pulls host batches, copies them pageable -> pinned and enqueues the pinned -> device copies on a side HIP
    stream; the consumer thread only launches kernels.  (Pulling a DataLoader batch maps a fresh 8 MB shared-memory segment
    and the pageable -> pinned memcpy faults it in: ~3 ms per batch, more than a 2.4 ms training step, if done in line.)
  * the device tensors live in a RING of `depth + 1` preallocated slots.  The first version allocated each batch on the side
    stream and `record_stream`-ed it for the consumer: the caching allocator then cannot reuse a block until the consumer's
    event completes, so every step paid hipMalloc/hipFree (device-synchronising on ROCm) -- 40 ms per fed step against
    2.4 ms resident.
  * host copies are single-threaded memcpys (see _stage_leaf): torch's OpenMP-parallel copy_ stalled the whole process.
  * contract: a yielded batch is valid until `depth` further batches have been requested (its slot is then overwritten,
    ordered behind the consumer stream's work by an event); keep a batch longer -> clone it."""
import queue
import threading

import numpy as np
import torch

from . import _capi


class _Slot:
    def __init__(self):
        self.pin, self.dev = {}, {}
        self.copied = None         # event on the side stream: pinned -> device copies of this slot are done
        self.released = None       # event on the consumer stream: the consumer has enqueued its last use of this slot


class TSDFFeed:
    """Iterates over an iterable of host batches (numpy arrays or CPU tensors, or nested tuples/lists of them) and yields the
    same structure as device tensors (views of the ring slots)."""

    def __init__(self, batches, device=None, depth=2):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _capi.GigaHipError("TSDFFeed stages batches for a HIP device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._batches = batches
        self._depth = max(2, int(depth))
        self._stream = torch.cuda.Stream(self.device)
        self._slots = [_Slot() for _ in range(self._depth + 1)]

    # -- staging thread ---------------------------------------------------------------------------------
    def _stage_leaf(self, slot, path, a):
        t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a.contiguous()
        pin = slot.pin.get(path)
        if pin is None or pin.shape != t.shape or pin.dtype != t.dtype:
            pin = slot.pin[path] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            slot.dev[path] = torch.empty(t.shape, dtype=t.dtype, device=self.device)
        # pageable / shared memory -> pinned: a plain single-threaded memcpy through numpy views (releases the GIL).  NOT
        # pin.copy_(t): torch parallelises a host-to-host copy over its whole OpenMP pool, and on a CPU-quota'd container
        # (256 visible cores, far fewer granted) the spinning team gets the whole process throttled for a scheduler period:
        # measured as 90 ms stalls of BOTH threads every few batches, 20-40 ms per fed step on average
        np.copyto(pin.numpy(), t.numpy())
        dev = slot.dev[path]
        dev.copy_(pin, non_blocking=True)                     # pinned -> device, asynchronous on the side stream
        return dev

    def _stage(self, slot, path, item):
        if isinstance(item, (tuple, list)):
            return type(item)(self._stage(slot, path + (i,), x) for i, x in enumerate(item))
        return self._stage_leaf(slot, path, item)

    def _stage_ring(self, slot, rb):
        """A RingBatch (giga_amd.dataset.GraspOccRing): its leaves already sit in page-locked shared memory, so the DMA reads
        them directly; the ring slot goes back to the reader processes once these copies have completed."""
        devs = []
        for i, h in enumerate(rb.leaves):
            full = slot.dev.get(("ring", i))
            if full is None or full.shape[1:] != h.shape[1:] or full.dtype != h.dtype or full.shape[0] < h.shape[0]:
                full = slot.dev[("ring", i)] = torch.empty((max(h.shape[0], rb.ring.bs),) + tuple(h.shape[1:]), dtype=h.dtype,
                                                            device=self.device)
            d = full[:h.shape[0]]
            d.copy_(h, non_blocking=True)
            devs.append(d)
        return rb.tree(devs)

    def _producer(self, out, stop):
        pending = []                                          # (copy event, RingBatch) whose host slot is still being read
        try:
            with torch.cuda.device(self.device):
                n = 0
                ring = getattr(self._batches, "pin", None)
                if ring is not None:                          # page-lock the shared-memory ring once
                    L = _capi.lib()
                    ring(lambda ptr, nbytes: L.giga_host_register(ptr, nbytes))
                for item in self._batches:
                    if stop.is_set():                         # the consumer left: hand every host slot we still hold back
                        if hasattr(item, "release"):
                            item.release()
                        return                                # (the `finally` below releases the pending slots)
                    while pending and (pending[0][0].query() or len(pending) >= 2):   # hand host slots back early: the ring must not run dry
                        ev, rb = pending.pop(0)
                        ev.synchronize()
                        rb.release()
                    slot = self._slots[n % len(self._slots)]
                    if slot.copied is not None:
                        slot.copied.synchronize()             # the copies that last READ this slot's pinned buffers are done
                    with torch.cuda.stream(self._stream):
                        if slot.released is not None:
                            self._stream.wait_event(slot.released)   # the consumer's kernels on this slot's old batch come first
                        is_ring = hasattr(item, "leaves") and hasattr(item, "release")
                        dev_item = self._stage_ring(slot, item) if is_ring else self._stage(slot, (), item)
                        slot.copied = torch.cuda.Event()
                        slot.copied.record(self._stream)
                    if is_ring:
                        pending.append((slot.copied, item))
                    out.put(("ok", (dev_item, slot)))         # blocks while `depth` batches are waiting: the ring never laps
                    n += 1
            out.put(("end", None))
        except BaseException as e:  # noqa: BLE001  (re-raised in the consumer)
            out.put(("err", e))
        finally:                                              # on EVERY path -- end of data, stop, an error in a DMA or stage step --
            for ev, rb in pending:                            # the ring slots whose copies are still in flight go back to the readers
                try:
                    ev.synchronize()
                except Exception:  # noqa: BLE001  (a failed copy must not strand the slot)
                    pass
                rb.release()
            pending.clear()

    # -- consumer -----------------------------------------------------------------------------------------
    def __iter__(self):
        # queue of depth - 1 + the batch in the consumer's hands + the one being staged = depth + 1 slots
        out = queue.Queue(maxsize=self._depth - 1)
        stop = threading.Event()
        t = threading.Thread(target=self._producer, args=(out, stop), daemon=True)
        t.start()
        try:
            while True:
                kind, payload = out.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise payload
                dev_item, slot = payload
                torch.cuda.current_stream(self.device).wait_event(slot.copied)
                yield dev_item
                # resumed = the consumer has enqueued all its work on this batch and asks for the next one.  Recorded BEFORE
                # the next get(): the producer can reach this slot again only after that get() has made room in the queue
                slot.released = torch.cuda.Event()
                slot.released.record(torch.cuda.current_stream(self.device))
        finally:
            stop.set()
            while t.is_alive():
                try:
                    out.get_nowait()
                except queue.Empty:
                    t.join(0.01)
