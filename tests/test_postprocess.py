"""Grasp post-processing (detection_implicit.py:87-174): oracle vs goldens captured from the reference's own
process/bound/select (CPU), and the device kernels vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

from giga_amd import synth, weights
from oracle import post_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g6_postprocess.npz")
R = 40


def _cases():
    g = np.load(GOLD)
    for c in range(3):
        seed, out_th, qual_th, force = g[f"c{c}_params"]
        yield c, int(seed), float(out_th), float(qual_th), bool(force), g


def _oracle(seed, out_th, qual_th, force):
    tsdf, qual, rot, width = synth.post_volumes(seed, R)
    q = post_oracle.process(tsdf[None], qual.copy(), width, out_th=out_th)
    q = post_oracle.bound(q, 0.3 / R)
    idx, scores = post_oracle.select_indices(q, threshold=qual_th, force_detection=force)
    return (tsdf, qual, rot, width), q, idx, scores


def test_oracle_matches_reference_goldens():
    lin = synth.inference_lattice(R).reshape(R, R, R, 3)[:, 0, 0, 0]
    for c, seed, out_th, qual_th, force, g in _cases():
        (tsdf, qual, rot, width), q, idx, scores = _oracle(seed, out_th, qual_th, force)
        assert np.array_equal(q[::2, ::2, ::2], g[f"c{c}_qual_s2"])
        assert int((q > 0).sum()) == int(g[f"c{c}_nonzero"])
        assert np.array_equal(scores, g[f"c{c}_scores"])
        assert np.array_equal(lin[idx], g[f"c{c}_centers"])
        assert np.array_equal(width[tuple(idx.T)], g[f"c{c}_widths"])


def test_bound_limits():
    assert post_oracle.bound_limits(0.3 / 40) == (2, 2, 7)
    from giga_amd.detection import bound_limits
    assert bound_limits(0.3 / 40) == (2, 2, 7)


@pytest.mark.gpu
def test_device_postprocess_matches_reference_goldens():
    from giga_amd.detection import grasp_select
    dev = torch.device("cuda:0")
    lin = synth.inference_lattice(R).reshape(R, R, R, 3)[:, 0, 0, 0]
    for c, seed, out_th, qual_th, force, g in _cases():
        (tsdf, qual, rot, width), q_ref, idx_ref, sc_ref = _oracle(seed, out_th, qual_th, force)
        t = lambda a: torch.from_numpy(a).to(dev)[None]
        sel, vol = grasp_select(t(tsdf), t(qual).reshape(1, -1), t(rot).reshape(1, -1, 4), t(width).reshape(1, -1),
                                out_th=out_th, threshold=qual_th, force_detection=force, return_volume=True)
        vol = vol[0].cpu().numpy()
        assert np.array_equal(vol == 0, q_ref == 0)                      # mask / gates / bound: exact
        np.testing.assert_allclose(vol, q_ref, rtol=0, atol=1e-7)        # gaussian in scipy's arithmetic
        s = sel[0]
        assert np.array_equal(s["index"], idx_ref)
        np.testing.assert_allclose(s["score"], g[f"c{c}_scores"], rtol=0, atol=1e-7)
        assert np.array_equal(lin[s["index"]], g[f"c{c}_centers"])
        assert np.array_equal(s["width"], g[f"c{c}_widths"])
        q = g[f"c{c}_quats"]
        dots = np.abs((s["rot"] * q).sum(-1))                            # scipy Rotation may flip the sign
        np.testing.assert_allclose(dots, 1.0, atol=1e-5)


@pytest.mark.gpu
def test_device_postprocess_batched_and_edges():
    """Batch of scenes in one call == per-scene calls; all-empty and constant volumes (plateau NMS)."""
    from giga_amd.detection import grasp_select
    dev = torch.device("cuda:0")
    vols = [synth.post_volumes(s, R) for s in (3, 4, 5)]
    st = lambda i: torch.from_numpy(np.stack([v[i] for v in vols])).to(dev)
    tsdf, qual, rot, width = st(0), st(1).reshape(3, -1), st(2).reshape(3, -1, 4), st(3).reshape(3, -1)
    batch = grasp_select(tsdf, qual, rot, width, out_th=0.1, threshold=0.8)
    for b in range(3):
        one = grasp_select(tsdf[b:b + 1], qual[b:b + 1], rot[b:b + 1], width[b:b + 1], out_th=0.1, threshold=0.8)[0]
        tq = post_oracle.bound(post_oracle.process(vols[b][0][None], vols[b][1].copy(), vols[b][3], out_th=0.1), 0.3 / R)
        idx, sc = post_oracle.select_indices(tq, threshold=0.8)
        for k in ("index", "score", "rot", "width"):
            assert np.array_equal(batch[b][k], one[k])
        assert np.array_equal(batch[b]["index"], idx)
    # nothing observed -> nothing valid -> no grasps, also with force_detection
    z = torch.zeros(1, R, R, R, device=dev)
    sel = grasp_select(z, qual[:1], rot[:1], width[:1], force_detection=True)
    assert len(sel[0]["score"]) == 0 and sel[0]["best_only"]
    # constant quality on a fully valid volume: every interior voxel ties with its window maximum
    ones = torch.ones(1, R, R, R, device=dev)
    cq = torch.full((1, R ** 3), 0.95, device=dev)
    cw = torch.full((1, R ** 3), 0.1, device=dev)
    sel, vol = grasp_select(ones, cq, rot[:1], cw, return_volume=True)
    tq = post_oracle.bound(post_oracle.process(np.ones((1, R, R, R), np.float32), np.full((R, R, R), 0.95, np.float32),
                                               np.full((R, R, R), 0.1, np.float32)), 0.3 / R)
    idx, sc = post_oracle.select_indices(tq)
    assert len(sel[0]["score"]) == len(sc) == 36 * 36 * 33
    assert np.array_equal(np.sort(sel[0]["index"].view("i8,i8,i8"), axis=0), np.sort(idx.astype(np.int64).view("i8,i8,i8"), axis=0))


@pytest.mark.gpu
def test_planner_hipgraph_replay_matches_eager(sd7):
    """VGNImplicit(use_graph=True) replays the network as one hipGraph; results must equal the eager launches,
    also when the input changes between replays."""
    from giga_amd import networks
    from giga_amd.detection import VGNImplicit
    dev = torch.device("cuda:0")
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).eval()
    kw = dict(net=net, force_detection=True, qual_th=0.6, out_th=0.1, best=True)
    eager, graphed = VGNImplicit(None, "giga", **kw), VGNImplicit(None, "giga", use_graph=True, **kw)

    class State:
        pass
    for scene in (0, 1, 2, 0):
        st = State()
        st.tsdf = synth.tsdf_batch(scene, 1, realistic=True)
        ga, sa, _ = eager(st)
        gb, sb, _ = graphed(st)
        assert len(ga) == len(gb) and len(ga) > 0
        assert np.array_equal(sa, sb)
        for a, b in zip(ga, gb):
            assert np.array_equal(a["rotation"], b["rotation"]) and np.array_equal(a["translation"], b["translation"])
            assert a["width"] == b["width"]
    # another batch size through the same network evicts the one-entry workspace caches the graph was captured with;
    # the graph holds its own references, so replaying it afterwards must neither read nor clobber recycled memory
    st = State()
    st.tsdf = synth.tsdf_batch(2, 1, realistic=True)
    want = eager(st)[1]
    batch = torch.from_numpy(synth.tsdf_batch(10, 3, realistic=True)).to(dev)
    sel3 = eager.plan_batch(batch)
    junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(16)]     # reuse whatever was freed
    assert np.array_equal(graphed(st)[1], want)
    again = eager.plan_batch(batch)
    assert all(np.array_equal(a["score"], b["score"]) for a, b in zip(sel3, again))
    assert all(bool(torch.isnan(j).all()) for j in junk)
    del junk
    # new weights invalidate the captured graph (it bakes in the packed-weight buffer)
    net.load_state_dict(weights.make_state_dict(8))
    st = State()
    st.tsdf = synth.tsdf_batch(1, 1, realistic=True)
    ga, sa, _ = eager(st)
    gb, sb, _ = graphed(st)
    assert len(ga) == len(gb) and np.array_equal(sa, sb)


@pytest.mark.gpu
def test_planner_matches_reference_planner_golden_g10(golden, sd7):
    """The whole planner call against golden G10 = the reference's VGNImplicit.__call__ (network on the 40^3 lattice,
    process, bound, select, metric conversion) with the same weights: same grasps in the same order."""
    from giga_amd import networks
    from giga_amd.detection import VGNImplicit
    g = golden("g10_planner.npz")
    dev = torch.device("cuda:0")
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).eval()
    planner = VGNImplicit(None, "giga", net=net, best=True, force_detection=True, qual_th=float(g["qual_th"]),
                          out_th=float(g["out_th"]))

    class State:
        pass
    for k in range(2):
        st = State()
        st.tsdf = synth.tsdf_batch(int(g[f"s{k}_scene"]), 1, realistic=True)
        grasps, scores, _ = planner(st)
        ref_scores = g[f"s{k}_scores"]
        assert len(grasps) == len(ref_scores) > 0
        np.testing.assert_allclose(scores, ref_scores, rtol=0, atol=2e-5)
        np.testing.assert_allclose(np.array([x["translation"] for x in grasps]), g[f"s{k}_translation"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.array([x["width"] for x in grasps]), g[f"s{k}_width"], rtol=0, atol=1e-5)
        dots = np.abs((np.array([x["rotation"] for x in grasps]) * g[f"s{k}_quat"]).sum(-1))
        np.testing.assert_allclose(dots, 1.0, atol=1e-5)
