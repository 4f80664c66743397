"""SURVEY 8f-3, reader half: the batched reader of the reference's training data (giga_amd/dataset.py) against golden G12
(the reference's own DatasetVoxelOccFile.__getitem__ on the synthetic on-disk dataset of giga_amd/synth.py (write_training_set)) and, when
/root/reference is present, against the live reference class including the random occupancy-file choice."""
import os

import numpy as np
import pytest
import torch

from giga_amd import dataset
from giga_amd import synth as make_dataset
from oracle import ref_bootstrap


def _same_item(a, b):
    xa, (la, ra, wa), pa, opa, oa = a
    xb, (lb, rb, wb), pb, opb, ob = b
    assert np.array_equal(np.asarray(xa), np.asarray(xb)) and np.asarray(xa).dtype == np.asarray(xb).dtype
    for u, v in ((la, lb), (ra, rb), (wa, wb), (pa, pb), (opa, opb), (oa, ob)):
        u, v = np.asarray(u), np.asarray(v)
        assert u.shape == v.shape and u.dtype == v.dtype and np.array_equal(u, v)


def test_items_match_golden_g12(tmp_path, golden):
    g = golden("g12_dataset_items.npz")
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    n = make_dataset.write_training_set(root, raw, seed=int(g["dataset_seed"]), occ_files=(1, 1))
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=int(g["num_point_occ"]))
    assert len(ds) == n == int(g["n"])
    for k in g["items"]:
        k = int(k)
        torch.manual_seed(100 + k); np.random.seed(200 + k)
        x, (label, rot, width), pos, op, occ = ds.item(k)
        assert x.shape == (40, 40, 40) and x.dtype == np.float32
        assert np.array_equal(x[::5, ::5, ::5], g[f"x_sub_{k}"]) and abs(x.astype(np.float64).sum() - float(g[f"x_sum_{k}"])) < 1e-9
        for got, key in ((label, "label"), (rot, "rot"), (width, "width"), (pos, "pos"), (op, "occ_points"), (occ, "occ")):
            ref = g[f"{key}_{k}"]
            got = np.asarray(got)
            assert got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref), (k, key)


def test_batches_and_epoch_iteration(tmp_path):
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    n = make_dataset.write_training_set(root, raw, n_scenes=4, grasps_per_scene=3, seed=2)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=50, workers=3)
    torch.manual_seed(5); np.random.seed(6)
    b = ds.batch([4, 1, 7])                                   # "global" rng: item by item, like a 0-worker DataLoader
    torch.manual_seed(5); np.random.seed(6)
    items = [ds.item(i) for i in (4, 1, 7)]
    assert b[0].shape == (3, 40, 40, 40) and b[3].shape == (3, 50, 3) and b[4].shape == (3, 50)
    for r, it in enumerate(items):                            # occ draws come first per item in batch(): same sequence
        assert np.array_equal(b[0][r], it[0]) and np.array_equal(b[1][1][r], it[1][1]) and np.array_equal(b[2][r], it[2])
    seen, sizes = [], []
    for xb, (lab, rot, wid), pos, op, occ in dataset.GraspOccBatches(ds, 5, shuffle=True, seed=3):
        sizes.append(len(lab))
        assert xb.dtype == np.float32 and rot.shape[1:] == (2, 4) and op.shape[1:] == (50, 3)
        seen += [tuple(np.round(p, 6)) for p in pos]
    assert sum(sizes) == n and sizes[:-1] == [5] * (len(sizes) - 1)
    assert sorted(seen) == sorted(tuple(np.round(p, 6)) for p in ds.pos)     # every grasp exactly once per epoch
    assert len(list(dataset.GraspOccBatches(ds, 5, drop_last=True))) == n // 5
    # worker processes: same epoch coverage, tensors instead of arrays, reproducible for a given seed
    src = dataset.GraspOccBatches(ds, 5, shuffle=True, seed=3, workers=2)
    ep1, ep1b = list(src), list(src)                          # two epochs from the same (persistent) workers
    ep2 = list(dataset.GraspOccBatches(ds, 5, shuffle=True, seed=3, workers=2))
    assert sum(len(b[1][0]) for b in ep1) == n and torch.is_tensor(ep1[0][0]) and ep1[0][0].dtype == torch.float32
    assert sum(len(b[1][0]) for b in ep1b) == n and not all(torch.equal(a[2], c[2]) for a, c in zip(ep1, ep1b))   # reshuffled
    for a, c in zip(ep1, ep2):
        assert torch.equal(a[0], c[0]) and torch.equal(a[3], c[3]) and torch.equal(a[4], c[4])
    # the tensor plumbing of prepare_batch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    x, pos1, pocc, y = dataset.network_inputs((t(b[0]), tuple(t(v) for v in b[1]), t(b[2]), t(b[3]), t(b[4])))
    assert pos1.shape == (3, 1, 3) and all(v.dtype == torch.float32 for v in (x, pos1, pocc) + y)


def test_shared_memory_ring_delivers_the_same_batches(tmp_path):
    """GraspOccRing (reader processes writing into shared-memory slots) == GraspOccBatches(workers > 0) for the same seed,
    in order, including the short last batch and a second (reshuffled) epoch; slots are recycled through release()."""
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    n = make_dataset.write_training_set(root, raw, n_scenes=5, grasps_per_scene=5, seed=6)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=40, workers=2)
    ring = dataset.GraspOccRing(ds, 4, workers=3, shuffle=True, seed=11, slots=5)
    ref = dataset.GraspOccBatches(ds, 4, shuffle=True, seed=11, workers=2)
    try:
        for epoch in range(2):
            got = []
            for rb in ring:
                x, (lab, rot, wid), pos, op, occ = rb.host()
                got.append(tuple(t.clone() for t in (x, lab, rot, wid, pos, op, occ)))
                rb.release()
            want = [(b[0], *b[1], b[2], b[3], b[4]) for b in ref]
            assert len(got) == len(want) == (n + 3) // 4 and got[-1][0].shape[0] == n % 4
            for g, w in zip(got, want):
                for a, b in zip(g, w):
                    assert a.dtype == b.dtype and torch.equal(a, b)
        seen = []
        assert ring.pin(lambda ptr, nbytes: seen.append((ptr, nbytes)) or 0) and len(seen) == 5 * 7
    finally:
        ring.close()


def test_shared_memory_ring_survives_an_abandoned_epoch(tmp_path):
    """A consumer that stops mid-epoch (break / exception / TSDFFeed's stop path) leaves submitted tasks behind: their results
    must not surface as batches of the NEXT pass, and their slots (and those of assembled but undelivered batches) must not
    starve the ring.  Three abandoned passes, then two complete ones that equal
    GraspOccBatches for the same seed and epoch count."""
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    n = make_dataset.write_training_set(root, raw, n_scenes=6, grasps_per_scene=6, seed=8)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=40, workers=2)
    ring = dataset.GraspOccRing(ds, 4, workers=3, shuffle=True, seed=21, slots=4)
    ref = dataset.GraspOccBatches(ds, 4, shuffle=True, seed=21, workers=2)
    try:
        for stop_after in (1, 2, 0):                          # abandoned passes (the reference iterator advances its epoch too)
            want = [(b[0], *b[1], b[2], b[3], b[4]) for b in ref]
            it = iter(ring)
            for k in range(stop_after):
                rb = next(it)
                assert torch.equal(rb.host()[0], want[k][0])
                rb.release()
            if stop_after == 0:
                next(it).release()
            it.close()                                        # GeneratorExit: the ring collects what is still in flight
            assert sorted(ring._free_slots) == [0, 1, 2, 3]
        for _ in range(2):
            want = [(b[0], *b[1], b[2], b[3], b[4]) for b in ref]
            got = []
            for rb in ring:
                got.append(tuple(t.clone() for t in (rb.host()[0], *rb.host()[1], *rb.host()[2:])))
                rb.release()
            assert len(got) == len(want) == (n + 3) // 4
            for g, w in zip(got, want):
                for a, b in zip(g, w):
                    assert a.dtype == b.dtype and torch.equal(a, b)
            assert sorted(ring._free_slots) == [0, 1, 2, 3]
    finally:
        ring.close()


@pytest.mark.reference
@pytest.mark.skipif(not ref_bootstrap.reference_available(), reason="/root/reference not present")
def test_items_match_live_reference_class(tmp_path):
    from oracle.make_feed_goldens import reference_items
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    make_dataset.write_training_set(root, raw, seed=4, occ_files=(2, 4))          # several occupancy files: the random pick matters
    idx = (0, 5, 9, 17, 23)
    _, ref = reference_items(root, raw, idx, 128)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=128)
    for i, r in zip(idx, ref):
        torch.manual_seed(100 + i); np.random.seed(200 + i)
        _same_item(ds.item(i), r)


def test_augmented_items_match_golden_g13(tmp_path, golden):
    """augment=True (scripts/train_giga.py:260; dataset_voxel.py:77-78,114-135): golden G13 is the reference's own class with
    augment=True on the same synthetic dataset and seeds as G12 -- transformed grid (nearest-neighbour quarter turn about z plus a
    z shift), transformed position (float64, normalised AFTER the transform as the reference does) and the two gripper rotations;
    values AND dtypes must match, the numpy draws happen in the reference's order (choice(4), uniform, then the point sample)."""
    g = golden("g13_dataset_items_augmented.npz")
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    n = make_dataset.write_training_set(root, raw, seed=int(g["dataset_seed"]), occ_files=(1, 1))
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=int(g["num_point_occ"]), augment=True)
    plain = dataset.GraspOccDataset(root, raw, num_point_occ=int(g["num_point_occ"]))
    assert len(ds) == n == int(g["n"])
    moved = 0
    for k in g["items"]:
        k = int(k)
        torch.manual_seed(100 + k); np.random.seed(200 + k)
        x, (label, rot, width), pos, op, occ = ds.item(k)
        assert x.shape == (40, 40, 40) and x.dtype == np.float32
        assert np.array_equal(x[::3, ::3, ::3], g[f"x_sub_{k}"]) and abs(x.astype(np.float64).sum() - float(g[f"x_sum_{k}"])) < 1e-9
        for got, key in ((label, "label"), (rot, "rot"), (width, "width"), (pos, "pos"), (op, "occ_points"), (occ, "occ")):
            ref = g[f"{key}_{k}"]
            got = np.asarray(got)
            assert got.shape == ref.shape and got.dtype == ref.dtype and np.array_equal(got, ref), (k, key)
        moved += int(not np.array_equal(x, plain.grid(plain.scene_ids[k])[0]))
        assert np.array_equal(plain.grid(plain.scene_ids[k]), np.load(os.path.join(root, "scenes", plain.scene_ids[k] + ".npz"))["grid"])
    assert moved >= 3                                          # (the transform really changes the grids; the cache keeps the originals)
    # whole batches: item by item with the global generators, i.e. what a 0-worker DataLoader would collate
    torch.manual_seed(9); np.random.seed(10)
    b = ds.batch([2, 5, 2])
    torch.manual_seed(9); np.random.seed(10)
    items = [ds.item(i) for i in (2, 5, 2)]
    for r, it in enumerate(items):
        assert np.array_equal(b[0][r], it[0]) and np.array_equal(b[1][1][r], it[1][1]) and np.array_equal(b[2][r], it[2])
        assert np.array_equal(b[3][r], it[3]) and np.array_equal(b[4][r], it[4])
    assert not np.array_equal(b[0][0], b[0][2]) or np.array_equal(b[2][0], b[2][2])     # the same grasp twice: independent draws
    rb = ds.batch([1, 4], rng=np.random.default_rng(3))        # the generator form used by the batch iterators
    assert rb[0].shape == (2, 40, 40, 40) and rb[2].shape == (2, 3) and rb[1][1].shape == (2, 2, 4)


@pytest.mark.reference
@pytest.mark.skipif(not ref_bootstrap.reference_available(), reason="/root/reference not present")
def test_augmented_items_match_live_reference_class(tmp_path):
    from oracle.make_feed_goldens import reference_items
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    make_dataset.write_training_set(root, raw, seed=5, occ_files=(2, 3))
    idx = (1, 6, 12, 20)
    _, ref = reference_items(root, raw, idx, 96, augment=True)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=96, augment=True)
    for i, r in zip(idx, ref):
        torch.manual_seed(100 + i); np.random.seed(200 + i)
        _same_item(ds.item(i), r)
