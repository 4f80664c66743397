"""CPU: the host half of the grasp selection (`giga_amd.detection._grasp_collect`).  The device compacts the NMS
survivors in arbitrary (atomic) order; the host must return them in the reference's order,
`reversed(np.argsort(scores))` over the ascending-index `np.argwhere` list (detection_implicit.py:160-166), ties
included.  The survivors are taken from the oracle's `select_indices` on volumes with plateaus, scrambled the way the
device would hand them over, and pushed through the product's host code with CPU tensors as buffers."""
import numpy as np
import torch

from giga_amd import synth
from giga_amd.detection import _GraspBuffers, _grasp_collect
from oracle import post_oracle as PO


def _survivor_volume(seed, R, plateau):
    tsdf, qual, rot, width = synth.post_volumes(seed, R)
    q = PO.bound(PO.process(tsdf, qual.copy(), width), 0.3 / R)
    if plateau:                                        # equal scores far apart: ties in the sort
        q = np.round(q * 8) / 8
    return q.astype(np.float32), rot, width


def _fill(buf, b, flat, score, rot, width, n_ge_threshold, rng):
    perm = rng.permutation(len(flat))                  # the device's compaction order is arbitrary
    k = len(flat)
    buf.counters[b, 0], buf.counters[b, 1] = int(n_ge_threshold), k
    buf.cand_index[b, :k] = torch.from_numpy(flat[perm].astype(np.int32))
    buf.cand_score[b, :k] = torch.from_numpy(score[perm])
    buf.cand_rot[b, :k] = torch.from_numpy(rot.reshape(-1, 4)[flat[perm]])
    buf.cand_width[b, :k] = torch.from_numpy(width.reshape(-1)[flat[perm]])


def test_host_ordering_matches_reference_order_with_ties():
    R, rng = 24, np.random.default_rng(5)
    cases = [(1, False, 0.5, False), (2, True, 0.5, False), (3, True, 0.6, False), (3, True, 0.99, True), (4, False, 2.0, True)]
    buf = _GraspBuffers(len(cases), R, 1024, "cpu")
    buf.pack.zero_()
    want = []
    plateau_expected = [c[1] and not c[3] for c in cases]      # tie cases really contain equal scores
    for b, (seed, plateau, th, force) in enumerate(cases):
        q, rot, width = _survivor_volume(seed, R, plateau)
        idx_all, sc_all = PO.select_indices(q, threshold=th, force_detection=False if not force else True)
        # what the device hands over: every NMS survivor of the thresholded (or, in best-only mode, LOW_TH) volume
        qq = q.copy(); qq[qq < PO.LOW_TH] = 0
        n_ge = int((qq >= th).sum())
        best_only = force and n_ge == 0
        if not best_only:
            qq[qq < th] = 0
        from scipy import ndimage
        mx = ndimage.maximum_filter(qq, size=4)
        surv = np.argwhere((qq == mx) & (qq > 0))
        flat = (surv[:, 0] * R + surv[:, 1]) * R + surv[:, 2]
        _fill(buf, b, flat, qq.reshape(-1)[flat], rot, width, n_ge, rng)
        want.append((idx_all, sc_all, best_only))
    out = _grasp_collect(buf, force_detection=True)
    for b, (idx, sc, best_only) in enumerate(want):
        if not cases[b][3]:                            # force_detection off for this scene: emulate by its own call
            single = _GraspBuffers(1, R, 1024, "cpu"); single.pack.copy_(torch.zeros_like(single.pack))
            k = int(buf.counters[b, 1])
            single.counters[0] = buf.counters[b]
            for name in ("cand_index", "cand_score", "cand_rot", "cand_width"):
                getattr(single, name)[0, :k] = getattr(buf, name)[b, :k]
            got = _grasp_collect(single, force_detection=False)[0]
        else:
            got = out[b]
        assert got["best_only"] == best_only
        assert np.array_equal(got["index"], idx), (b, got["index"][:5], idx[:5])
        assert np.array_equal(got["score"], sc)
        assert len(idx) > 0 and (not plateau_expected[b] or len(np.unique(sc)) < len(sc))


def test_capacity_overflow_is_reported():
    buf = _GraspBuffers(2, 8, 4, "cpu")
    buf.pack.zero_()
    buf.counters[1, 1] = 5                             # more survivors than `cap`: the caller must retry with R^3
    assert _grasp_collect(buf, force_detection=False) is None
