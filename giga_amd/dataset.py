"""Input side of the hot path: a batched reader for the reference's training data.

Counterpart of `vgn.dataset_voxel.DatasetVoxelOccFile` (reference src/vgn/dataset_voxel.py:55-106; `augment=True` = its
`apply_transform`, :114-135, restated below) and
`vgn.io.read_voxel_grid` (src/vgn/io.py:97-99) for the data layout `scripts/train_giga.py:118-138` trains on:

    <root>/scenes/<scene_id>.npz            key "grid":  (1, 40, 40, 40) float32 TSDF          (io.py:88-99)
    <raw_root>/grasps.csv                   scene_id, qx, qy, qz, qw, x, y, z, width, label    (io.py:56-81)
    <raw_root>/setup.json                   "size" (workspace edge length)                      (io.py:11-26)
    <raw_root>/occ/<scene_id>/*.npz         keys "points" (n, 3), "occ" (n,)                    (dataset_voxel.py:93-101)

The reference reads one grasp per `__getitem__` through pandas `.loc` lookups and lets DataLoader workers collate; here the
table is turned into arrays once, whole batches are assembled (`batch(indices)`), decompression of the `.npz` members runs
on a thread pool (zlib releases the GIL), decoded grids are kept in an LRU cache (many grasps share a scene), and
`GraspOccBatches` produces batches on a background thread so that `giga_amd.feed.TSDFFeed` can stage batch k+1 while the
GPU runs step k.  `item(i)` returns exactly the reference's per-item tuple and, with `rng="global"`, draws from the same
global RNGs in the same order (torch.randint for the occupancy file, np.random.choice for the point sample), which is how
the parity tests compare it with the reference class."""
import json
import os
import queue
import threading
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch


def read_voxel_grid(root, scene_id):
    """io.py:97-99."""
    return np.load(Path(root) / "scenes" / (scene_id + ".npz"))["grid"]


def apply_transform(voxel_grid, orientation, position, rng="global"):
    """dataset_voxel.py:114-135 (`scripts/train_giga.py:260 --augment`): a random quarter turn about z and a random shift along z,
    about the centre (20, 20, 20) of the voxel grid, applied to the TSDF (scipy's nearest-neighbour `affine_transform`, the
    reference's own dependency) and to the grasp pose.  voxel_grid (1, 40, 40, 40) is modified in place like the reference's;
    orientation is a scipy Rotation, position the (3,) array __getitem__ hands over -- the METRIC position of the grasp table
    (:72), which the reference transforms with these voxel-unit offsets before it normalises it (:80); restated as it is.
    rng == "global": the reference's draws from numpy's global generator, in its order (choice(4), then uniform(6, 34))."""
    from scipy import ndimage
    from scipy.spatial.transform import Rotation
    if rng == "global":
        angle = np.pi / 2.0 * np.random.choice(4)
        z_offset = np.random.uniform(6, 34) - position[2]
    else:
        angle = np.pi / 2.0 * int(rng.integers(4))
        z_offset = rng.uniform(6, 34) - position[2]
    r_aug = Rotation.from_rotvec(np.r_[0.0, 0.0, angle])
    t_aug = np.asarray(np.r_[0.0, 0.0, z_offset], np.double)
    centre = np.asarray(np.r_[20.0, 20.0, 20.0], np.double)
    ident = Rotation.from_quat([0.0, 0.0, 0.0, 1.0])
    # utils/transform.py:44-60: (R1, t1) * (R2, t2) = (R1 R2, R1 t2 + t1); inverse = (R^-1, -R^-1 t)
    def mul(a, b):
        return a[0] * b[0], a[0].apply(b[1]) + a[1]

    def inv(a):
        r = a[0].inv()
        return r, -r.apply(a[1])

    T = mul(mul((ident, centre), (r_aug, t_aug)), inv((ident, centre)))      # T_center * T_augment * T_center^-1
    Ti = inv(T)
    voxel_grid[0] = ndimage.affine_transform(voxel_grid[0], Ti[0].as_matrix(), Ti[1], order=0)
    position = T[0].apply(position) + T[1]
    orientation = T[0] * orientation
    return voxel_grid, orientation, position


class GraspOccDataset:
    def __init__(self, root, raw_root, num_point_occ=2048, cache_scenes=4096, workers=8, augment=False):
        import pandas as pd
        from scipy.spatial.transform import Rotation
        self.root, self.raw_root = Path(root), Path(raw_root)
        self.num_point_occ = num_point_occ
        self.augment = augment
        df = pd.read_csv(self.raw_root / "grasps.csv")                       # io.py:80-81
        with (self.raw_root / "setup.json").open("r") as f:                  # io.py:19-26
            self.size = json.load(f)["size"]
        self.scene_ids = df["scene_id"].astype(str).tolist()
        quat = self._quat = df.loc[:, "qx":"qw"].to_numpy(np.single)         # dataset_voxel.py:71
        self._pos = df.loc[:, "x":"z"].to_numpy(np.single)                   # :72
        self._width = df["width"].to_numpy().astype(np.single)               # :73
        self.label = df["label"].to_numpy().astype(np.int64)                 # :74 (np.long)
        # :83-87  rotations[0] = ori, rotations[1] = ori * R_z(pi): the gripper's 180-degree symmetry
        ori = Rotation.from_quat(quat)
        flip = Rotation.from_rotvec(np.pi * np.r_[0.0, 0.0, 1.0])
        self.rotations = np.empty((len(df), 2, 4), dtype=np.single)
        if len(df):
            self.rotations[:, 0] = ori.as_quat()
            self.rotations[:, 1] = (ori * flip).as_quat()
        self.pos = self._pos / self.size - 0.5                               # :80
        self.width = self._width / self.size                                 # :81
        self._grids = OrderedDict()
        self._cache_scenes = cache_scenes
        self._lock = threading.Lock()
        self._workers = max(1, workers)
        self._pool, self._pool_pid = None, None            # created lazily, per process (the object may be forked into
        self._occ_paths = {}                                # DataLoader workers, where inherited threads do not exist)

    def _executor(self):
        if self._pool is None or self._pool_pid != os.getpid():
            self._pool, self._pool_pid = ThreadPoolExecutor(max_workers=self._workers), os.getpid()
        return self._pool

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_pool"], d["_pool_pid"], d["_lock"], d["_grids"] = None, None, None, OrderedDict()
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._lock = threading.Lock()

    def __len__(self):
        return len(self.scene_ids)

    # -- pieces ----------------------------------------------------------------------------------------
    def grid(self, scene_id):
        with self._lock:
            g = self._grids.get(scene_id)
            if g is not None:
                self._grids.move_to_end(scene_id)
                return g
        g = read_voxel_grid(self.root, scene_id)
        with self._lock:
            self._grids[scene_id] = g
            while len(self._grids) > self._cache_scenes:
                self._grids.popitem(last=False)
        return g

    def occ_paths(self, scene_id):
        p = self._occ_paths.get(scene_id)
        if p is None:
            p = self._occ_paths[scene_id] = list((self.raw_root / "occ" / scene_id).glob("*.npz"))   # dataset_voxel.py:94
        return p

    def _read_occ(self, scene_id, rng):
        """dataset_voxel.py:93-102.  rng == "global": the reference's own draws (torch.randint, np.random.choice)."""
        paths = self.occ_paths(scene_id)
        n_pt = self.num_point_occ
        if rng == "global":
            k = torch.randint(high=len(paths), size=(1,), dtype=int).item()
        else:
            k = int(rng.integers(len(paths)))
        data = np.load(paths[k])
        points, occ = data["points"], data["occ"]
        n_all = points.shape[0]
        if rng == "global":
            idxs = np.random.choice(np.arange(n_all), size=(n_pt,), replace=n_pt > n_all)           # :137-139
        else:
            idxs = rng.choice(n_all, size=(n_pt,), replace=n_pt > n_all)
        return points[idxs] / self.size - 0.5, occ[idxs]                                            # :89-90

    # -- the reference's per-item view -------------------------------------------------------------------
    def item(self, i, rng="global"):
        """DatasetVoxelOccFile.__getitem__ (dataset_voxel.py:69-91): x (40,40,40), (label, rotations (2,4), width), pos (3,),
        occ_points (M,3), occ (M,)."""
        sid = self.scene_ids[i]
        if self.augment:
            x, rot, pos = self._augmented(i, rng)
            occ_points, occ = self._read_occ(sid, rng)
            return x, (self.label[i], rot, self.width[i]), pos, occ_points, occ
        x = self.grid(sid)[0]
        occ_points, occ = self._read_occ(sid, rng)
        return x, (self.label[i], self.rotations[i], self.width[i]), self.pos[i], occ_points, occ

    def _augmented(self, i, rng):
        """dataset_voxel.py:77-87 with augment: the transformed grid (a copy: the cache keeps the file's), the two gripper
        rotations of the transformed orientation and the normalised transformed position (float64, as the reference's)."""
        from scipy.spatial.transform import Rotation
        grid = np.array(self.grid(self.scene_ids[i]), copy=True)
        grid, ori, pos = apply_transform(grid, Rotation.from_quat(self._quat[i]), self._pos[i], rng)
        pos = pos / self.size - 0.5
        rot = np.empty((2, 4), dtype=np.single)
        rot[0] = ori.as_quat()
        rot[1] = (ori * Rotation.from_rotvec(np.pi * np.r_[0.0, 0.0, 1.0])).as_quat()
        return grid[0], rot, pos

    # -- whole batches -------------------------------------------------------------------------------------
    def batch(self, indices, rng="global"):
        """The collated batch torch's DataLoader would hand to `prepare_batch` (train_giga.py:141-151), as numpy arrays:
        x (B,40,40,40) f32, (label (B,) i64, rotations (B,2,4) f32, width (B,) f32), pos (B,3) f32, occ_points (B,M,3), occ (B,M)."""
        indices = np.asarray(indices, dtype=np.int64)
        sids = [self.scene_ids[i] for i in indices]
        if self.augment:                                     # per item: augmentation draws, then the occupancy draws (reference order)
            subs = [rng] * len(sids) if rng == "global" else (
                rng.spawn(len(sids)) if hasattr(rng, "spawn") else [np.random.default_rng(rng.integers(1 << 62)) for _ in sids])
            items = [self.item(int(i), r) for i, r in zip(indices, subs)]
            x = np.stack([it[0] for it in items]).astype(np.float32, copy=False)
            y = (self.label[indices], np.stack([it[1][1] for it in items]), self.width[indices])
            return x, y, np.stack([it[2] for it in items]), np.stack([it[3] for it in items]), np.stack([it[4] for it in items])
        if rng == "global":                                  # reference order: item by item, sequential draws
            occ = [self._read_occ(s, rng) for s in sids]
            grids = [self.grid(s) for s in sids]
        else:
            subs = rng.spawn(len(sids)) if hasattr(rng, "spawn") else [np.random.default_rng(rng.integers(1 << 62)) for _ in sids]
            ex = self._executor()
            fo = [ex.submit(self._read_occ, s, r) for s, r in zip(sids, subs)]
            fg = {s: ex.submit(self.grid, s) for s in dict.fromkeys(sids)}
            occ = [f.result() for f in fo]
            grids = [fg[s].result() for s in sids]
        x = np.stack([g[0] for g in grids]).astype(np.float32, copy=False)
        y = (self.label[indices], self.rotations[indices], self.width[indices])
        return x, y, self.pos[indices], np.stack([o[0] for o in occ]), np.stack([o[1] for o in occ])


class _BatchSet(torch.utils.data.Dataset):
    """Map-style view for torch's DataLoader with batch_size=None: the sampler hands out keys (seed, epoch, k, indices) and an
    element is the WHOLE batch, assembled in a worker process with a generator seeded from the key."""

    def __init__(self, ds):
        self.ds = ds

    def __getitem__(self, key):
        seed, epoch, k, idx = key
        x, (lab, rot, wid), pos, op, occ = self.ds.batch(idx, rng=np.random.default_rng([seed, epoch, k]))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        return t(x), (t(lab), t(rot), t(wid)), t(pos), t(op), t(occ)


class _EpochKeys(torch.utils.data.Sampler):
    def __init__(self):
        self.keys = []

    def __iter__(self):
        return iter(self.keys)

    def __len__(self):
        return len(self.keys)


class GraspOccBatches:
    """Iterable of host batches, produced ahead of the consumer.  One pass = one epoch (`shuffle` reorders per epoch with the
    seeded generator; the last short batch is kept unless `drop_last`).
      workers = 0 : a background THREAD assembles the batches (numpy arrays).  Fine for small sets and tests, but the reader
                    is Python/zip-bound and shares the GIL with the training loop: measured 60-80 ms per fed step against
                    2.4 ms resident (tools/gpu_feed_bench.py).
      workers > 0 : torch DataLoader worker PROCESSES, each assembling whole batches (tensors arrive through shared memory),
                    as the reference's own loader does per item (train_giga.py:132: num_workers=8)."""

    def __init__(self, dataset, batch_size, shuffle=True, seed=0, drop_last=False, prefetch=3, workers=0):
        self.ds, self.bs, self.shuffle, self.drop_last, self.prefetch = dataset, batch_size, shuffle, drop_last, prefetch
        self.workers, self.seed, self.epoch = int(workers), int(seed), 0
        self._rng = np.random.default_rng(seed)
        self._loader, self._keys = None, None                # worker processes persist across epochs (starting 8-16 of them
                                                             # costs seconds: more than a whole small epoch of 3 ms steps)

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def _chunks(self):
        order = self._rng.permutation(len(self.ds)) if self.shuffle else np.arange(len(self.ds))
        chunks = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        if self.drop_last and chunks and len(chunks[-1]) < self.bs:
            chunks.pop()
        return chunks

    def __iter__(self):
        chunks = self._chunks()
        self.epoch += 1
        if self.workers > 0:
            if self._loader is None:
                self._keys = _EpochKeys()
                self._loader = torch.utils.data.DataLoader(_BatchSet(self.ds), batch_size=None, sampler=self._keys,
                                                           num_workers=self.workers, prefetch_factor=max(2, self.prefetch),
                                                           persistent_workers=True)
            self._keys.keys = [(self.seed, self.epoch, k, c) for k, c in enumerate(chunks)]
            yield from self._loader
            return
        q = queue.Queue(maxsize=max(1, self.prefetch))
        stop = threading.Event()

        def produce():
            try:
                for c in chunks:
                    if stop.is_set():
                        return
                    q.put(("ok", self.ds.batch(c, rng=self._rng)))
                q.put(("end", None))
            except BaseException as e:  # noqa: BLE001  (surfaced in the consumer)
                q.put(("err", e))

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise item
                yield item
        finally:
            stop.set()
            while t.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    t.join(0.01)


class RingBatch:
    """One batch living in slot `slot` of a GraspOccRing: `leaves` are views of the slot's shared-memory tensors (valid until
    `release()`), `tree(leaves)` rebuilds the (x, (label, rotations, width), pos, occ_points, occ) structure from any
    same-length list (host views, device copies, ...)."""

    def __init__(self, ring, slot, n):
        self.ring, self.slot, self.n = ring, slot, n
        self.leaves = [t[:n] for t in ring.buffers[slot]]

    @staticmethod
    def tree(v):
        return v[0], (v[1], v[2], v[3]), v[4], v[5], v[6]

    def host(self):
        return self.tree(self.leaves)

    def release(self):
        if self.ring is not None:
            self.ring._free(self.slot)
            self.ring = None


def _ring_worker(ds, buffers, tasks, done):
    """Reader process: assemble whole batches straight into the shared-memory slot named by the task."""
    torch.set_num_threads(1)
    while True:
        task = tasks.get()
        if task is None:
            return
        slot, seed, epoch, k, idx = task
        try:
            x, (lab, rot, wid), pos, op, occ = ds.batch(idx, rng=np.random.default_rng([seed, epoch, k]))
            for dst, src in zip(buffers[slot], (x, lab, rot, wid, pos, op, occ)):
                np.copyto(dst.numpy()[:len(idx)], src, casting="same_kind")
            done.put((epoch, k, slot, len(idx), None))
        except BaseException as e:  # noqa: BLE001
            import traceback
            done.put((epoch, k, slot, 0, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


class GraspOccRing:
    """Reader PROCESSES that write whole batches into a ring of preallocated shared-memory slots, delivered in order as
    RingBatch objects.  No per-batch pickling, file descriptors or mmaps (torch's DataLoader ships every batch in a fresh
    shared-memory segment: ~3 ms per 9 MB batch on the consumer side, more than a training step), and the slots can be
    page-locked once (`pin`), so the H->D DMA reads them directly -- TSDFFeed does that when it is handed a ring.
    One pass = one epoch; same batches as GraspOccBatches(workers > 0) for the same seed.  The consumer must `release()`
    every RingBatch (TSDFFeed does, once the copy to the device has completed)."""

    def __init__(self, dataset, batch_size, workers=8, shuffle=True, seed=0, drop_last=False, slots=None):
        import multiprocessing as mp
        self.ds, self.bs, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.seed, self.epoch = int(seed), 0
        self._rng = np.random.default_rng(seed)
        self.workers = max(1, int(workers))
        n_slots = int(slots) if slots else self.workers + 4
        probe = dataset.batch([0], rng=np.random.default_rng(0))             # leaf dtypes / trailing shapes
        flat = (probe[0], probe[1][0], probe[1][1], probe[1][2], probe[2], probe[3], probe[4])
        self.buffers = [[torch.empty((self.bs,) + a.shape[1:], dtype=torch.from_numpy(np.asarray(a)).dtype).share_memory_()
                         for a in flat] for _ in range(n_slots)]
        ctx = mp.get_context("fork")                          # children inherit the shared mappings and the dataset tables
        self._tasks, self._done = ctx.Queue(), ctx.Queue()
        self._procs = [ctx.Process(target=_ring_worker, args=(dataset, self.buffers, self._tasks, self._done), daemon=True)
                       for _ in range(self.workers)]
        for p in self._procs:
            p.start()
        self._free_slots = list(range(n_slots))
        self._lock = threading.Lock()
        self._room = threading.Condition(self._lock)
        self._pinned = None

    def __len__(self):
        n = len(self.ds)
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def pin(self, register):
        """Page-lock every slot once: register(ptr, nbytes) -> 0 on success (giga_host_register).  Returns True if all
        slots are pinned (a failure leaves the ring usable as ordinary pageable memory)."""
        if self._pinned is None:
            self._pinned = all(register(t.data_ptr(), t.numel() * t.element_size()) == 0 for slot in self.buffers for t in slot)
        return self._pinned

    def _free(self, slot):
        with self._room:
            self._free_slots.append(slot)
            self._room.notify()

    def close(self):
        for _ in self._procs:
            self._tasks.put(None)
        for p in self._procs:
            p.join(timeout=2)
            if p.is_alive():
                p.terminate()
        self._procs = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __iter__(self):
        order = self._rng.permutation(len(self.ds)) if self.shuffle else np.arange(len(self.ds))
        chunks = [order[i:i + self.bs] for i in range(0, len(order), self.bs)]
        if self.drop_last and chunks and len(chunks[-1]) < self.bs:
            chunks.pop()
        self.epoch += 1
        epoch = self.epoch
        submitted, received, delivered, ready = 0, 0, 0, {}
        failed = False
        try:
            while delivered < len(chunks):
                # hand out as many tasks as there are free slots (the consumer's release() refills the list)
                with self._room:
                    while submitted < len(chunks) and self._free_slots:
                        slot = self._free_slots.pop()
                        self._tasks.put((slot, self.seed, epoch, submitted, chunks[submitted]))
                        submitted += 1
                    if delivered not in ready and received == submitted:
                        self._room.wait(timeout=0.05)         # every slot is out with the consumer: wait for a release
                        continue
                while delivered not in ready:
                    ep, k, slot, n, err = self._done.get()
                    if ep != epoch:                           # a straggler of an abandoned pass: only its slot matters
                        self._free(slot)
                        continue
                    received += 1
                    if err is not None:
                        failed = True
                        self._free(slot)
                        raise RuntimeError("reader process failed: " + err)
                    ready[k] = (slot, n)
                slot, n = ready.pop(delivered)
                delivered += 1
                yield RingBatch(self, slot, n)
        finally:
            # The consumer may stop mid-epoch (break, an exception, TSDFFeed's stop path): every task already handed to a
            # reader still completes into its slot.  Collect those results here and give their slots -- and the slots of
            # assembled but undelivered batches -- back, so that the next pass neither sees an old batch under a new index
            # nor starves for slots.  Results are tagged with the epoch; anything from another pass is discarded above.
            for slot, _n in ready.values():
                self._free(slot)
            ready.clear()
            import queue as _queue
            waited = 0.0
            while received < submitted and self._procs:
                try:
                    ep, _k, slot, _n, _err = self._done.get(timeout=0.25)
                except _queue.Empty:
                    waited += 0.25                            # a dead reader never answers: its slot is lost, the ring keeps the
                    if waited >= (5.0 if failed else 30.0) or not all(p.is_alive() for p in self._procs):   # others; do not sit out the
                        break                                 # whole time-out for it
                    continue
                self._free(slot)
                if ep == epoch:
                    received += 1


def network_inputs(batch):
    """Device batch (as staged by TSDFFeed from a GraspOccBatches item) -> (x, pos (B,1,3), pos_occ, y) in the dtypes and
    shapes the training step consumes: the tensor plumbing of `prepare_batch` (train_giga.py:141-151) minus the copies."""
    x, (label, rotations, width), pos, pos_occ, occ = batch
    return (x.float(), pos.float().unsqueeze(1), pos_occ.float(),
            (label.float(), rotations.float(), width.float(), occ.float()))
