"""Pin the CPU oracle (oracle/giga_oracle.py) to fixtures captured from the reference itself
(oracle/make_goldens.py).  fp32 throughout; tolerance 2e-5 absolute on O(1) values (the oracle
uses the closed-form axis mean instead of scatter_mean and torch.where instead of masked writes,
so summation order differs slightly from the reference)."""
import numpy as np
import torch

from giga_amd import synth, weights
from oracle import giga_oracle as O

TOL = 2e-5


def close(a, b, tol=TOL):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max()
    assert err <= tol, f"max abs err {err:.3e} > {tol}"


def test_param_inventory():
    # 581 863 parameters, encoder 476 800
    assert weights.num_params(True) == 581863
    shapes = weights.giga_param_shapes()
    assert sum(int(np.prod(s)) for k, s in shapes.items() if k.startswith("encoder.")) == 476800


def test_g1_encoder(golden, sd7):
    g = golden("g1_encoder.npz")
    x = torch.from_numpy(synth.tsdf_batch(0, 2))
    planes = O.encoder_forward(sd7, x)
    for k in O.PLANES:
        assert planes[k].shape == (2, 32, 40, 40)
        close(planes[k][:, :, ::2, ::2], g[f"plane_{k}_s2"])
        close(planes[k][0, 5], g[f"plane_{k}_b0c5"])
        s = g[f"plane_{k}_sums"]
        assert abs(planes[k].double().sum().item() - s[0]) <= 1e-5 * max(1.0, s[1])


def test_g2_model_and_raw_heads(golden, sd7):
    g = golden("g2_decoder.npz")
    x = torch.from_numpy(synth.tsdf_batch(0, 2))
    p = torch.from_numpy(synth.query_points(0, 2, 2048, stream=1, half_width=0.6))
    qual, rot, width, tsdf = O.model_forward(sd7, x, p, p_tsdf=p)
    close(qual, g["qual"]); close(rot, g["rot"]); close(width, g["width"]); close(tsdf, g["tsdf"])
    planes = O.encoder_forward(sd7, x)
    for h in weights.HEADS:
        close(O.decoder_forward(sd7, h, p, planes), g["raw_" + h])
    assert p.abs().max() > 0.5          # both clamps of normalize_coordinate are exercised


def test_g2b_decoder_on_random_planes(golden, sd7):
    g = golden("g2b_decoder_random_planes.npz")
    rng = np.random.default_rng(int(g["plane_seed"]))
    rp = {k: torch.from_numpy(rng.standard_normal((2, 32, 40, 40)).astype(np.float32))
          for k in ("xz", "xy", "yz")}
    p = torch.from_numpy(synth.query_points(0, 2, 2048, stream=1, half_width=0.6))
    for h in weights.HEADS:
        close(O.decoder_forward(sd7, h, p, rp), g["raw_" + h], tol=5e-5)


def test_g3_inference_lattice(golden, sd7):
    g = golden("g3_lattice.npz")
    lat = torch.from_numpy(synth.inference_lattice())
    assert lat.shape == (1, 64000, 3)
    close(lat[0, [0, 1, 40, 1600, 63999]], g["lattice_first_last"], tol=0)
    close(O.inference_lattice(), lat, tol=0)
    x = torch.from_numpy(synth.tsdf_batch(int(g["scene"]), 1))
    q, r, w = O.model_forward(sd7, x, lat)
    sub = g["subset"]
    close(q[0, sub], g["qual"]); close(r[0, sub], g["rot"]); close(w[0, sub], g["width"])
    for t, name in ((q, "qual"), (r, "rot"), (w, "width")):
        s = g[name + "_sums"]
        assert abs(t.double().sum().item() - s[0]) <= 2e-5 * max(1.0, s[1])


def test_g4_train_losses_and_grads(golden, sd7):
    g = golden("g4_train_step.npz")
    B, M, s0 = int(g["B"]), int(g["M"]), int(g["first_scene"])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd7.items()}
    x = torch.from_numpy(synth.tsdf_batch(s0, B))
    pos = torch.from_numpy(synth.query_points(s0, B, 1, stream=2))
    pos_occ = torch.from_numpy(synth.query_points(s0, B, M, stream=3))
    y = tuple(torch.from_numpy(a) for a in synth.train_labels(s0, B, M))
    out = O.model_forward(sd, x, pos, p_tsdf=pos_occ)
    loss, d = O.train_loss(O.train_select(out), y)
    for k in ("loss_qual", "loss_rot", "loss_width", "loss_occ", "loss_all"):
        assert abs(d[k].item() - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    loss.backward()
    names = [str(n) for n in g["grad_names"]]
    for n, ref in zip(names, g["grad_norms"]):
        got = sd[n].grad.double().norm().item()
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-6) + 1e-9, (n, got, ref)
    close(sd["decoder_qual.fc_out.weight"].grad, g["grad_fc_out_qual"], tol=1e-6)
    close(sd["encoder.conv_in.weight"].grad, g["grad_conv_in_w"], tol=1e-5)


def test_g5_edges(golden, sd7):
    g = golden("g5_edges.npz")
    for name, val in (("zeros", 0.0), ("ones", 1.0)):
        pl = O.encoder_forward(sd7, torch.full((1, 40, 40, 40), val))
        for k in O.PLANES:
            close(pl[k][:, :, ::4, ::4], g[f"{name}_plane_{k}_s4"])
    pe = torch.from_numpy(g["edge_points"])
    x = torch.from_numpy(synth.tsdf_batch(int(g["edge_scene"]), 1))
    q, r, w, t = O.model_forward(sd7, x, pe, p_tsdf=pe)
    close(q, g["edge_qual"]); close(r, g["edge_rot"]); close(w, g["edge_width"]); close(t, g["edge_tsdf"])


def test_g7_generation_eval_points(golden, sd7):
    """Occupancy logits of the reference's Generator3D.eval_points (generation.py:326-358) == oracle decoder_tsdf."""
    g = golden("g7_generation.npz")
    x = torch.from_numpy(synth.tsdf_batch(int(g["first_scene"]), 1))
    p = torch.from_numpy(synth.query_points(int(g["first_scene"]), 1, int(g["n"]), stream=int(g["stream"]),
                                            half_width=float(g["half_width"])))
    got = O.infer_geo(sd7, x, p)[0].numpy()
    assert np.abs(got - g["logits"]).max() < 2e-5


def test_g8_detach_gradients(golden, sd7):
    """giga_detach (networks.py:143-169): the oracle's detach_tsdf gradients == the reference network's."""
    g = golden("g8_detach.npz")
    B, M, s0 = int(g["B"]), int(g["M"]), int(g["first_scene"])
    x = torch.from_numpy(synth.tsdf_batch(s0, B))
    pos = torch.from_numpy(synth.query_points(s0, B, 1, stream=2))
    pos_occ = torch.from_numpy(synth.query_points(s0, B, M, stream=3))
    y = tuple(torch.from_numpy(a) for a in synth.train_labels(s0, B, M))
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd7.items()}
    with torch.enable_grad():
        loss, _ = O.train_loss(O.train_select(O.model_forward(sdg, x, pos, p_tsdf=pos_occ, detach_tsdf=True)), y)
        loss.backward()
    assert abs(loss.item() - float(g["loss_all"])) < 1e-5
    for n, ref in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        got = sdg[n].grad.double().norm().item()
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-6) + 1e-9, (n, got, ref)
    ref_w = g["grad_conv_in_w"]
    assert np.abs(sdg["encoder.conv_in.weight"].grad.numpy() - ref_w).max() < 1e-5 * np.abs(ref_w).max()


def test_g9_ablation_variants(golden):
    """giga_aff / giga_geo of the reference (networks.py:65-141) == the oracle restricted to their heads."""
    g = golden("g9_variants.npz")
    x = torch.from_numpy(synth.tsdf_batch(int(g["first_scene"]), 2))
    p = torch.from_numpy(synth.query_points(int(g["first_scene"]), 2, 100))
    q, r, w = O.model_forward(weights.make_state_dict(int(g["aff_seed"]), with_tsdf=False), x, p)
    for got, key in ((q, "aff_qual"), (r, "aff_rot"), (w, "aff_width")):
        assert np.abs(got.numpy() - g[key]).max() < 2e-5
    t = O.infer_geo(weights.make_state_dict(int(g["geo_seed"]), heads=("decoder_tsdf",)), x, p)
    assert np.abs(t.numpy() - g["geo_tsdf"]).max() < 2e-5 and np.abs(t.numpy() - g["geo_occ_logits"]).max() < 2e-5


def test_g11_training_helpers(golden):
    """select / loss_fn / prepare_batch of scripts/train_giga.py:141-195 (golden from the reference's own functions)
    == the oracle's restatement (the fused HIP loss is held to the same golden in tests/test_gpu_training.py)."""
    g = golden("g11_train_helpers.npz")
    B, M, s0 = int(g["B"]), int(g["M"]), int(g["first_scene"])
    t = torch.from_numpy
    y = tuple(t(a) for a in synth.train_labels(s0, B, M))
    heads = (t(g["qual"]), t(g["rot"]), t(g["width"]), t(g["logit"]))
    for sel, lossf in ((O.train_select, O.train_loss),):
        yp = sel(heads)
        assert np.array_equal(yp[3].numpy(), g["sel_occ"])
        loss, d = lossf(yp, y)
        assert abs(float(loss) - float(g["loss"])) < 1e-6
        for k in ("loss_qual", "loss_rot", "loss_width", "loss_occ", "loss_all"):
            assert abs(float(d[k]) - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), k
    pc = t(synth.tsdf_batch(s0, B)[:, None])
    pos = t(synth.query_points(s0, B, 1, stream=2)[:, 0])
    pos_occ = t(synth.query_points(s0, B, M, stream=3))
    # the synthetic batch has the shapes the reference's prepare_batch hands to the network (train_giga.py:141-151)
    pb = (pc.float(), None, pos.unsqueeze(1).float(), pos_occ.float())
    sh = g["prepare_shapes"]
    assert list(pb[0].shape) == [int(v) for v in sh[0][:pb[0].dim()]]
    assert list(pb[2].shape) == [int(v) for v in sh[1][:pb[2].dim()]] and list(pb[3].shape) == [int(v) for v in sh[2][:pb[3].dim()]]
