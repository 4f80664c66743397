"""GPU parity: the HIP path (through the module API -> C ABI) against the CPU oracle and against the
reference-generated golden fixtures.  Tolerances: fp32 path 1e-4 absolute on O(1) outputs (fp32
MFMA = fp32 fma chains, summation order differs from ATen's); f16 path per the north star 1e-3 on
the head outputs that matter (sigmoid qual / unit rot), looser on raw logits (stated per test)."""
import numpy as np
import pytest
import torch

from giga_amd import networks, synth, weights
from oracle import giga_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def net(dev, sd7):
    n = networks.get_network("giga")
    n.load_state_dict(sd7)
    return n.to(dev).eval()


def maxerr(a, b):
    return (a.detach().float().cpu() - torch.as_tensor(b).float()).abs().max().item()


def test_native_library_is_loaded():
    from giga_amd import _capi
    assert _capi.lib().giga_abi_version() == 3
    maps = open("/proc/self/maps").read()
    assert "libgiga_hip.so" in maps


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("fp16x3", 1e-4), ("fp16", 2e-2)])
def test_encoder_matches_oracle_and_g1(net, dev, sd7, golden, prec, tol):
    net.set_precision(prec)
    x = torch.from_numpy(synth.tsdf_batch(0, 2))
    with torch.no_grad():
        planes = net.encode_inputs(x.to(dev))
        ref = O.encoder_forward(sd7, x)
    g = golden("g1_encoder.npz")
    for k in O.PLANES:
        assert planes[k].shape == (2, 32, 40, 40)
        assert maxerr(planes[k], ref[k]) < tol, k
        assert maxerr(planes[k][:, :, ::2, ::2], g[f"plane_{k}_s2"]) < tol, k
    # NHWC image used by the decoder is the same data
    nhwc = planes.nhwc.permute(0, 1, 4, 2, 3).float().cpu()
    assert (nhwc - torch.stack([ref[k] for k in O.PLANES])).abs().max().item() < tol


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("fp16x3", 1e-4), ("fp16", 1e-2)])
def test_decoder_heads_on_foreign_planes_g2b(net, dev, sd7, golden, prec, tol):
    """LocalDecoder.forward(p, c_plane) with reference-layout planes (decoder.py:133)."""
    g = golden("g2b_decoder_random_planes.npz")
    rng = np.random.default_rng(int(g["plane_seed"]))
    rp = {k: torch.from_numpy(rng.standard_normal((2, 32, 40, 40)).astype(np.float32)).to(dev)
          for k in ("xz", "xy", "yz")}
    p = torch.from_numpy(synth.query_points(0, 2, 2048, stream=1, half_width=0.6)).to(dev)
    with torch.no_grad():
        for h in weights.HEADS:
            dec = getattr(net, h)
            dec.precision = prec
            out = dec(p, rp)
            assert maxerr(out, g["raw_" + h]) < tol * max(1.0, float(np.abs(g["raw_" + h]).max())), h


# Plain 'fp16' is the mode OUTSIDE the 1e-3 contract.  Its stated envelope (the one tests/test_gpu_c4_shapes.py holds it to at the
# bench shapes): 1e-2 on sigmoid(qual), 2e-2 on the unit quaternion -- a raw error of ~3e-3 divided by a small norm -- and on the raw
# width / occupancy logits.  The quaternion of this golden sits at 1.0-1.1e-2 with EITHER U-Net kernel (conv16 measured 1.02e-2,
# conv32 1.0-1.1e-2: different summation orders, same envelope), so the kernel is a parameter of the test and both carry the same bound.
@pytest.mark.parametrize("prec,tol,kernel", [("fp32", 1e-4, None), ("fp16x3", 1e-4, None), ("fp16", 1e-2, "conv16"), ("fp16", 1e-2, "conv32")])
def test_model_forward_g2(net, dev, sd7, golden, prec, tol, kernel):
    net.set_precision(prec)
    if kernel is not None:
        net.set_unet_kernel(kernel)
    g = golden("g2_decoder.npz")
    x = torch.from_numpy(synth.tsdf_batch(0, 2)).to(dev)
    p = torch.from_numpy(synth.query_points(0, 2, 2048, stream=1, half_width=0.6)).to(dev)
    try:
        with torch.no_grad():
            qual, rot, width, tsdf = net(x, p, p_tsdf=p)
    finally:
        if kernel is not None:
            net.set_unet_kernel("auto")
    assert qual.shape == (2, 2048) and rot.shape == (2, 2048, 4) and width.shape == (2, 2048) and tsdf.shape == (2, 2048)
    assert maxerr(qual, g["qual"]) < tol
    assert maxerr(rot, g["rot"]) < tol * (1.5 if prec == "fp16" else 1)     # fp16: 1.5e-2 (measured 1.0-1.1e-2 with either kernel)
    assert maxerr(width, g["width"]) < tol * 2
    assert maxerr(tsdf, g["tsdf"]) < tol * 2


@pytest.mark.parametrize("prec,tol", [("fp16x3", 2e-5), ("fp16", 1.5e-2)])
def test_auto_kernel_choice_across_the_16_scene_threshold(net, dev, sd7, prec, tol):
    """`auto` runs the f16-class U-Net on conv32 up to 16 scenes and on conv16 beyond (include/giga_hip.h): the same scene evaluated
    in a batch of 16 and in a batch of 17 goes through different kernels.  Both stay inside the mode's envelope of each other
    (f16x3: 2e-5 -- fp32-grade either way; plain f16: the f16 envelope), a forced kernel makes the two batches agree bit for bit,
    and giga_encoder_last_path reports which kernels ran."""
    from giga_amd import _capi
    net.set_precision(prec)
    x17 = torch.from_numpy(synth.tsdf_batch(40, 17)).to(dev)
    p17 = torch.from_numpy(synth.query_points(40, 17, 512, stream=1)).to(dev)
    L = _capi.lib()
    with torch.no_grad():
        a = net(x17[:16], p17[:16]); path16 = L.giga_encoder_last_path()
        b = net(x17, p17); path17 = L.giga_encoder_last_path()
        assert (path16 & _capi.PATH_CONV32) and not (path17 & _capi.PATH_CONV32), (path16, path17)
        for u, v in zip(a, b):
            scale = max(1.0, float(v.abs().max()))
            assert maxerr(u[0], v[0].cpu().numpy()) <= tol * scale
        net.set_unet_kernel("conv16")
        try:
            c = net(x17[:16], p17[:16]); d = net(x17, p17)
        finally:
            net.set_unet_kernel("auto")
        for u, v in zip(c, d):
            assert torch.equal(u[0], v[0])


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("fp16x3", 1e-4), ("fp16", 1e-2)])
def test_inference_lattice_g3_and_predict(net, dev, sd7, golden, prec, tol):
    from giga_amd.detection import predict, query_lattice
    net.set_precision(prec)
    g = golden("g3_lattice.npz")
    pos = query_lattice(40, dev)
    assert pos.shape == (1, 64000, 3)
    q, r, w = predict(synth.tsdf_batch(int(g["scene"]), 1), pos, net, dev)
    assert q.shape == (64000,) and r.shape == (64000, 4) and w.shape == (64000,)
    sub = g["subset"]
    assert np.abs(q[sub] - g["qual"]).max() < tol
    assert np.abs(r[sub] - g["rot"]).max() < tol
    assert np.abs(w[sub] - g["width"]).max() < tol * 2
    if prec != "fp16":
        for arr, name in ((q, "qual"), (r, "rot"), (w, "width")):
            s = g[name + "_sums"]
            assert abs(arr.astype(np.float64).sum() - s[0]) < 2e-5 * max(1.0, s[1])


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-5), ("fp16x3", 2e-5), ("fp16", 1.5e-2)])
def test_lattice_fast_path_equals_generic_path(net, dev, sd7, prec, tol):
    """The registered inference lattice (shared by a batch of scenes) takes the resampled-plane path;
    a plain copy of the same points takes the generic gather path.  Same arithmetic, so fp32 agrees
    to rounding; both agree with the oracle."""
    from giga_amd.detection import predict_batch, query_lattice
    net.set_precision(prec)
    lat = query_lattice(40, dev)                          # registered -> lattice path
    x = torch.from_numpy(synth.tsdf_batch(60, 3)).to(dev)
    plain = lat.clone().expand(3, -1, -1).contiguous()    # unregistered -> generic path
    with torch.no_grad():
        fast = predict_batch(x, lat, net)
        slow = net(x, plain)
        ref = O.model_forward(sd7, x[1:2].cpu(), lat.cpu())
    for a, b, r in zip(fast, slow, ref):
        assert a.shape == b.shape and a.shape[0] == 3
        assert maxerr(a, b.cpu()) < tol
        assert maxerr(a[1:2], r) < max(tol, 1e-4)
    net.set_precision("fp32")


def test_unregistered_reference_lattice_is_detected_from_data(net, dev, sd7):
    """The reference's own VGNImplicit builds its query lattice itself (detection_implicit.py:28-31) and hands it to the
    network; that tensor was never registered with giga_amd, yet it must take the lattice fast path (recognised from its
    data, once per tensor), while a tensor of the same shape that is NOT a lattice takes the generic gather."""
    from giga_amd import convonet
    net.set_precision("fp32")
    R = 40
    lin = torch.linspace(start=-0.5, end=0.5 - 1.0 / R, steps=R)              # detection_implicit.py:28-31, verbatim recipe
    x_, y_, z_ = torch.meshgrid(lin, lin, lin, indexing="ij")
    pos = torch.stack((x_, y_, z_), dim=-1).float().unsqueeze(0).to(dev).view(1, R * R * R, 3)
    x = torch.from_numpy(synth.tsdf_batch(11, 1)).to(dev)
    before = dict(convonet.LATTICE_STATS)
    with torch.no_grad():
        fast = net(x, pos)
        again = net(x, pos)                                  # cached: no second detection
        bent = pos.clone(); bent[0, 12345, 1] += 1e-3        # same shape, not a lattice
        slow = net(x, bent)
        ref = O.model_forward(sd7, x.cpu(), pos.cpu())
    after = convonet.LATTICE_STATS
    assert after["fast"] - before["fast"] == 2 and after["detected"] - before["detected"] == 1
    assert after["generic"] - before["generic"] == 1
    for a, b, c, r in zip(fast, again, slow, ref):
        assert torch.equal(a, b)
        assert maxerr(a, r) < 1e-4
        keep = torch.ones(R ** 3, dtype=torch.bool); keep[12345] = False
        assert maxerr(a[0][keep.to(dev)], c[0][keep.to(dev)].cpu()) < 2e-5
    pos[0, 7, 0] += 0.25                                     # an in-place edit bumps the version: re-checked, now generic
    with torch.no_grad():
        net(x, pos)
    assert convonet.LATTICE_STATS["generic"] - before["generic"] == 2


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("fp16x3", 1e-4), ("fp16", 1e-2)])
def test_edge_cases_g5(net, dev, sd7, golden, prec, tol):
    """All-zero / all-one volumes, query points exactly on +-0.5 and on cell centres (bilinear weights 0 / 1).  Plain fp16 is held
    to its stated envelope (1e-2: a throughput mode outside the 1e-3 contract), the fp32-grade modes to 1e-4."""
    net.set_precision(prec)
    g = golden("g5_edges.npz")
    try:
        with torch.no_grad():
            for name, val in (("zeros", 0.0), ("ones", 1.0)):
                pl = net.encode_inputs(torch.full((1, 40, 40, 40), val, device=dev))
                for k in O.PLANES:
                    want = g[f"{name}_plane_{k}_s4"]
                    assert maxerr(pl[k][:, :, ::4, ::4], want) < tol * max(1.0, float(np.abs(want).max()))
            pe = torch.from_numpy(g["edge_points"]).to(dev)
            x = torch.from_numpy(synth.tsdf_batch(int(g["edge_scene"]), 1)).to(dev)
            q, r, w, t = net(x, pe, p_tsdf=pe)
        assert maxerr(q, g["edge_qual"]) < tol and maxerr(r, g["edge_rot"]) < 2 * tol
        assert maxerr(w, g["edge_width"]) < 2 * tol and maxerr(t, g["edge_tsdf"]) < 2 * tol
    finally:
        net.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "fp16x3"])
def test_ragged_and_tiny_batches(net, dev, sd7, prec):
    """N not a multiple of the 32-point tile, N = 1 (train_giga's single grasp query), B = 1 and 5."""
    net.set_precision(prec)
    for B, N, M in ((1, 1, 7), (5, 1, 2048), (3, 33, 95), (2, 257, 1)):
        x = torch.from_numpy(synth.tsdf_batch(40, B))
        p = torch.from_numpy(synth.query_points(40, B, N, stream=4))
        pt = torch.from_numpy(synth.query_points(40, B, M, stream=5))
        with torch.no_grad():
            out = net(x.to(dev), p.to(dev), p_tsdf=pt.to(dev))
            ref = O.model_forward(sd7, x, p, p_tsdf=pt)
        for a, b in zip(out, ref):
            assert a.shape == b.shape
            assert maxerr(a, b) < 2e-4, (B, N, M)


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("fp16x3", 1e-4), ("fp16", 2e-2)])
@pytest.mark.parametrize("B", [33, 44])
def test_batches_that_do_not_fill_groups_of_eight(net, dev, sd7, prec, tol, B):
    """From 32 scenes up conv_in runs one workgroup group per scene and maps whole groups of 8 scenes onto the 8 XCDs; the
    scenes beyond the last full group take the plain order.  Every scene of such a batch equals the same scene run alone
    (the small-batch kernels), and the first / last scene equal the oracle."""
    net.set_precision(prec)
    x = torch.from_numpy(synth.tsdf_batch(700, B)).to(dev)
    with torch.no_grad():
        pl = net.encode_inputs(x)
        for k in (0, 7, 8, 31, 32, B - 1):
            one = net.encode_inputs(x[k:k + 1].contiguous())
            for key in ("xz", "xy", "yz"):
                assert maxerr(pl[key][k:k + 1], one[key].cpu()) < (1e-5 if prec != "fp16" else 1e-2), (k, key)
    for k in (0, B - 1):
        ref = O.encoder_forward(sd7, x[k:k + 1].cpu())
        for key in ("xz", "xy", "yz"):
            scale = max(1.0, float(ref[key].abs().max()))
            assert maxerr(pl[key][k:k + 1], ref[key]) < tol * scale, (k, key)
    net.set_precision("fp32")


@pytest.mark.parametrize("prec", ["fp32", "fp16x3", "fp16"])
def test_empty_and_degenerate_inputs(net, dev, prec):
    """Empty batch, zero query points, zero occupancy points, non-contiguous / float64 inputs, mismatched batch sizes."""
    net.set_precision(prec)
    x = torch.from_numpy(synth.tsdf_batch(3, 2)).to(dev)
    p = torch.from_numpy(synth.query_points(3, 2, 40, stream=1)).to(dev)
    with torch.no_grad():
        ref = net(x, p, p_tsdf=p)
        q, r, w = net(x[:0], p[:0])                          # no scenes
        assert q.shape == (0, 40) and r.shape == (0, 40, 4) and w.shape == (0, 40)
        q, r, w, t = net(x, p[:, :0], p_tsdf=p)              # no grasp queries
        assert q.shape == (2, 0) and r.shape == (2, 0, 4) and torch.equal(t, ref[3])
        q, r, w, t = net(x, p, p_tsdf=p[:, :0])              # no occupancy queries
        assert t.shape == (2, 0) and torch.equal(q, ref[0]) and torch.equal(r, ref[1])
        # non-contiguous views and float64 inputs are accepted and converted (the reference calls .float() itself)
        xt = x.permute(0, 3, 2, 1).contiguous().permute(0, 3, 2, 1)
        pw = torch.cat((p, p), 2)[:, :, 3:]
        out = net(xt.double(), pw.double(), p_tsdf=pw)
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        planes = net.encode_inputs(x)
        with pytest.raises(ValueError):
            net.decode(p[:1], planes)                        # one scene of points, two scenes of planes
        with pytest.raises(ValueError):
            net.encode_inputs(x[:, :39])                     # not a 40^3 grid
    from giga_amd import _capi
    with pytest.raises(_capi.GigaHipError):
        net(x.cpu(), p.cpu())                                # no CPU fallback
    net.set_precision("fp32")


def test_scene_independence_and_determinism(net, dev):
    """Batch dim is carried untouched: scene i of a batch == the same scene alone; reruns are bit-identical."""
    net.set_precision("fp32")
    x = torch.from_numpy(synth.tsdf_batch(7, 4)).to(dev)
    p = torch.from_numpy(synth.query_points(7, 4, 500, stream=6)).to(dev)
    with torch.no_grad():
        full = net(x, p, p_tsdf=p)
        again = net(x, p, p_tsdf=p)
        one = net(x[2:3].contiguous(), p[2:3].contiguous(), p_tsdf=p[2:3].contiguous())
    for a, b in zip(full, again):
        assert torch.equal(a, b)
    # (not bit-identical: the number of ix-slabs of the projection kernel, hence the fp32 summation
    #  order of the yz plane, depends on the batch size)
    for a, b in zip(full, one):
        assert maxerr(a[2:3], b.cpu()) < 1e-5


def test_full_size_properties_c2_c4(net, dev):
    """BASELINE sizes (B=32 x 2048, and 64 000 lattice queries): size-independent properties --
    unit quaternions, qual in (0,1), finite outputs, fp16 within 1e-2 of fp32."""
    x = torch.from_numpy(synth.tsdf_batch(100, 32)).to(dev)
    p = torch.from_numpy(synth.query_points(100, 32, 2048, stream=7)).to(dev)
    with torch.no_grad():
        net.set_precision("fp32")
        q, r, w, t = net(x, p, p_tsdf=p)
        net.set_precision("fp16")
        q16, r16, w16, t16 = net(x, p, p_tsdf=p)
        net.set_precision("fp16x3")
        qs, rs, ws_, ts = net(x, p, p_tsdf=p)
        lat = torch.from_numpy(synth.inference_lattice()).to(dev)
        ql, rl, wl = net(x[:1].contiguous(), lat)
    for v in (q, r, w, t, ql, rl, wl):
        assert torch.isfinite(v).all()
    assert (r.norm(dim=-1) - 1).abs().max().item() < 1e-5 and (rl.norm(dim=-1) - 1).abs().max().item() < 1e-3
    assert q.min().item() > 0 and q.max().item() < 1
    assert maxerr(q16, q.cpu()) < 1e-2 and maxerr(r16, r.cpu()) < 2e-2 and maxerr(w16, w.cpu()) < 2e-2
    # the split mode at full size: within 2e-5 of the fp32 kernels on every output of all 65 536 points per head
    for a, b in ((qs, q), (rs, r), (ws_, w), (ts, t)):
        assert maxerr(a, b.cpu()) < 2e-5
    net.set_precision("fp32")
    # two scenes of the 32-scene batch against the oracle (the batch takes the one-x-part conv_in kernels, the
    # small-batch tests the five-x-part ones), and one of them against the same scene run alone
    sd = weights.make_state_dict(7)
    for k in (5, 31):
        ref = O.model_forward(sd, x[k:k + 1].cpu(), p[k:k + 1].cpu(), p_tsdf=p[k:k + 1].cpu())
        for got, want in zip((q, r, w, t), ref):
            assert maxerr(got[k:k + 1], want) < 1e-4
    with torch.no_grad():
        alone = net(x[5:6].contiguous(), p[5:6].contiguous(), p_tsdf=p[5:6].contiguous())
    for got, one in zip((q, r, w, t), alone):
        assert maxerr(got[5:6], one.cpu()) < 1e-5


def test_variants_aff_geo(dev, golden):
    """giga_aff (no occupancy head) and giga_geo (occupancy only) run on the same kernels; held to the
    oracle and to golden G9 (the reference's own ablation networks)."""
    g9 = golden("g9_variants.npz")
    sd_aff = weights.make_state_dict(3, with_tsdf=False)
    aff = networks.get_network("giga_aff"); aff.load_state_dict(sd_aff); aff = aff.to(dev).eval()
    x = torch.from_numpy(synth.tsdf_batch(9, 2)); p = torch.from_numpy(synth.query_points(9, 2, 100))
    with torch.no_grad():
        out = aff(x.to(dev), p.to(dev))
        ref = O.model_forward(sd_aff, x, p)
    for a, b in zip(out, ref):
        assert maxerr(a, b) < 1e-4
    for a, key in zip(out, ("aff_qual", "aff_rot", "aff_width")):
        assert maxerr(a, g9[key]) < 1e-4
    sd_geo = weights.make_state_dict(4, heads=("decoder_tsdf",))
    geo = networks.get_network("giga_geo"); geo.load_state_dict(sd_geo); geo = geo.to(dev).eval()
    with torch.no_grad():
        t = geo.infer_geo(x.to(dev), p.to(dev))
        ref_t = O.infer_geo(sd_geo, x, p)
        occ = geo.decode_occ(p.to(dev), geo.encode_inputs(x.to(dev)))
    assert maxerr(t, ref_t) < 1e-4
    assert maxerr(occ.logits, ref_t) < 1e-4
    assert maxerr(t, g9["geo_tsdf"]) < 1e-4 and maxerr(occ.logits, g9["geo_occ_logits"]) < 1e-4


@pytest.mark.gpu
def test_generation_eval_points_and_grid(sd7, golden):
    """encode once, query occupancy many times (generation.py:326-358) == oracle decoder_tsdf."""
    from giga_amd.generation import Generator3D
    dev = torch.device("cuda:0")
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).eval()
    gen = Generator3D(net, device=dev)
    x = torch.from_numpy(synth.tsdf_batch(70, 2))
    c = gen.encode(x)
    planes = O.encoder_forward(sd7, x)
    for rnd, n in enumerate((1, 777, 4096)):               # MISE-like rounds on the cached planes
        p = torch.from_numpy(synth.query_points(70, 2, n, stream=10 + rnd, half_width=0.55))
        ref = O.decoder_forward(sd7, "decoder_tsdf", p, planes)
        got = gen.eval_points(p, c).cpu()
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() < 1e-4
    chunked = Generator3D(net, points_batch_size=1000, device=dev, threshold=0.3, upsampling_steps=2)   # host queries in 5 chunks
    assert torch.equal(chunked.eval_points(p, c), got.to(dev)) and chunked.mesh_options["upsampling_steps"] == 2
    one = gen.eval_points(p[0], {k: v[:1] for k, v in c.items()}).cpu()      # (N,3) form, reference-style dict
    assert (one - ref[0]).abs().max().item() < 1e-4
    g7 = golden("g7_generation.npz")                        # the reference's own Generator3D.eval_points on scene 70
    p7 = torch.from_numpy(synth.query_points(70, 1, int(g7["n"]), stream=int(g7["stream"]), half_width=float(g7["half_width"])))[0]
    got7 = gen.eval_points(p7, {k: v[:1] for k, v in c.items()}).cpu().numpy()
    assert np.abs(got7 - g7["logits"]).max() < 1e-4
    grid = gen.occupancy_grid(c, resolution=16).cpu()
    ref_grid = O.decoder_forward(sd7, "decoder_tsdf", gen.grid_points(16).cpu().expand(2, -1, -1), planes)
    assert (grid.reshape(2, -1) - ref_grid).abs().max().item() < 1e-4


@pytest.mark.gpu
def test_tsdf_feed_pinned_double_buffer(sd7):
    """the pinned, ring-buffered H->D feed delivers every batch intact and in order, also when the ring laps
    several times, when the consumer keeps the GPU busy between batches, and with a short last batch."""
    from giga_amd.feed import TSDFFeed
    dev = torch.device("cuda:0")
    host = [(synth.tsdf_batch(10 * i, 3 if i < 10 else 2)[:, None], torch.from_numpy(synth.query_points(10 * i, 3 if i < 10 else 2, 5)))
            for i in range(11)]
    seen, sums = 0, []
    busy = torch.randn(2048, 2048, device=dev)
    for i, (xb, pb) in enumerate(TSDFFeed(host, dev)):
        assert xb.is_cuda and pb.is_cuda and xb.shape == host[i][0].shape
        for _ in range(3):
            busy = busy @ busy * 1e-3                         # asynchronous consumer work that reads the batch afterwards
        sums.append((xb.double().sum() + busy[0, 0].double() * 0).item() if i % 2 else xb.double().sum())
        assert torch.equal(xb.cpu(), torch.from_numpy(host[i][0])) and torch.equal(pb.cpu(), host[i][1])
        seen += 1
    assert seen == 11
    for i, v in enumerate(sums):
        assert abs(float(v) - float(host[i][0].astype(np.float64).sum())) < 1e-6


@pytest.mark.gpu
def test_shared_ring_feed_end_to_end(tmp_path):
    """Reader processes -> page-locked shared-memory ring -> DMA into the device ring (TSDFFeed over GraspOccRing): every
    batch arrives intact and in order over two epochs, and matches the plain GraspOccBatches path."""
    from giga_amd import dataset
    from giga_amd.feed import TSDFFeed
    dev = torch.device("cuda:0")
    root, raw = str(tmp_path / "data"), str(tmp_path / "raw")
    synth.write_training_set(root, raw, n_scenes=6, grasps_per_scene=7, seed=3)
    ds = dataset.GraspOccDataset(root, raw, num_point_occ=64, workers=2)
    ring = dataset.GraspOccRing(ds, 8, workers=3, shuffle=True, seed=5)
    ref = dataset.GraspOccBatches(ds, 8, shuffle=True, seed=5, workers=2)
    try:
        for _ in range(2):
            want = [(b[0], *b[1], b[2], b[3], b[4]) for b in ref]
            k = 0
            for x, (lab, rot, wid), pos, op, occ in TSDFFeed(ring, dev):
                for a, b in zip((x, lab, rot, wid, pos, op, occ), want[k]):
                    assert a.is_cuda and a.dtype == b.dtype and torch.equal(a.cpu(), b)
                k += 1
            assert k == len(want)
        assert ring._pinned is True
    finally:
        ring.close()


@pytest.mark.gpu
def test_folded_final_conv_matches_piecewise_path(sd7):
    """net(x, p, p_tsdf) folds conv_final into the decoders' fc_c (GIGA_FOLD_FINAL); the piecewise reference-style calls
    (encode_inputs -> decode / decode_occ) exchange the FINAL planes.  Both must agree with each other and the oracle."""
    dev = torch.device("cuda:0")
    x = torch.from_numpy(synth.tsdf_batch(90, 3))
    p = torch.from_numpy(synth.query_points(90, 3, 333, stream=4, half_width=0.55))
    ref = O.model_forward(sd7, x, p, p_tsdf=p)
    for prec, tol_pair, tol_ref in (("fp32", 2e-5, 1e-4), ("fp16x3", 2e-5, 1e-4), ("fp16", 1e-2, 1e-2)):
        net = networks.get_network("giga")
        net.load_state_dict(sd7)
        net = net.to(dev).eval().set_precision(prec)
        with torch.no_grad():
            fused = net(x.to(dev), p.to(dev), p_tsdf=p.to(dev))
            c = net.encode_inputs(x.to(dev))
            piece = net.decode(p.to(dev), c) + (net.decode_occ(p.to(dev), c).logits,)
        for a, b, r in zip(fused, piece, ref):
            assert (a - b).abs().max().item() < tol_pair
            assert (a.cpu() - r).abs().max().item() < tol_ref


@pytest.mark.gpu
def test_batch_256_c3_scene_consistency(net, dev):
    """BASELINE c3 runs 256 scenes (32 per GPU on 8 GPUs; here all on one): scene k of the big batch equals the same
    scene alone, for both precisions (size-independent property; exercises the large-batch index arithmetic)."""
    x = torch.from_numpy(synth.tsdf_batch(500, 256)).to(dev)
    p = torch.from_numpy(synth.query_points(500, 256, 64, stream=8)).to(dev)
    for prec, tol in (("fp32", 1e-5), ("fp16x3", 1e-5), ("fp16", 1e-2)):
        net.set_precision(prec)
        with torch.no_grad():
            full = net(x, p, p_tsdf=p)
            for k in (0, 131, 255):
                one = net(x[k:k + 1].contiguous(), p[k:k + 1].contiguous(), p_tsdf=p[k:k + 1].contiguous())
                for a, b in zip(full, one):
                    assert maxerr(a[k:k + 1], b.cpu()) < tol
    net.set_precision("fp32")


def test_persistent_unet_launches_in_flight_on_several_streams(net, dev):
    """The persistent U-Net launch places its workgroups by ticket among the workgroups already resident on their XCD, so that
    several such launches -- each a full device's worth of spinning workgroups -- can be in flight at once without holding each
    other's CUs in a cycle (csrc/giga_encoder.hip::unet_mega_kernel; the round-2 form needed a one-stream contract).  Four
    streams enqueue 32-scene and 2-scene encoder calls back to back without any synchronisation in between; every result must
    equal the same call made alone, and nothing may trap (a group that cannot fill traps after about a second)."""
    streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
    xs = [torch.from_numpy(synth.tsdf_batch(900 + 40 * k, 32 if k % 2 == 0 else 2)).to(dev) for k in range(4)]
    try:
        for prec in ("fp16", "fp32"):
            net.set_precision(prec)
            net.set_persistent_unet(True)                       # (fp32 at 2 scenes takes the persistent form only when asked)
            with torch.no_grad():
                want = [net.encode_inputs(x)["xz"].clone() for x in xs]
                torch.cuda.synchronize()
                got = [[] for _ in xs]
                for rep in range(25):
                    for k, (st, x) in enumerate(zip(streams, xs)):
                        with torch.cuda.stream(st):
                            got[k].append(net.encode_inputs(x)["xz"])
                torch.cuda.synchronize()
            for k in range(4):
                for g in got[k]:
                    assert torch.equal(g, want[k]), (prec, k)
    finally:
        net.set_persistent_unet(False)
        net.set_precision("fp32")


def test_more_encoder_calls_in_flight_than_persistent_launches_allowed(net, dev):
    """Eight streams, no synchronisation in between: more than the four persistent U-Net launches the library allows in flight per
    device (a launch can park one unfilled group of <= 7 workgroups per XCD; 5 x 7 > 32 slots).  The library tracks the completion
    events of its last four persistent launches and gives the fifth concurrent call one launch per layer instead
    (csrc/giga_encoder.hip::persistent_slot) -- same results bit for bit in the f16-class modes -- so nothing may trap and every
    result must equal the same call made alone."""
    streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
    xs = [torch.from_numpy(synth.tsdf_batch(1300 + 40 * k, (32, 2, 11, 1)[k % 4])).to(dev) for k in range(8)]
    try:
        for prec in ("fp16", "fp16x3"):
            net.set_precision(prec)
            with torch.no_grad():
                want = [net.encode_inputs(x)["yz"].clone() for x in xs]
                torch.cuda.synchronize()
                got = [[] for _ in xs]
                for rep in range(12):
                    for k, (st, x) in enumerate(zip(streams, xs)):
                        with torch.cuda.stream(st):
                            got[k].append(net.encode_inputs(x)["yz"])
                torch.cuda.synchronize()
            for k in range(8):
                for g in got[k]:
                    assert torch.equal(g, want[k]), (prec, k)
    finally:
        net.set_precision("fp32")


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8, 10, 11, 32])
def test_persistent_unet_kernel_is_bit_identical_to_per_layer_launches(net, dev, B):
    """The U-Net as ONE persistent launch (unet_mega_kernel: groups of 8 workgroups inside one XCD walk their images through all
    layers, barriers through that XCD's own L2) computes every image with the same instruction sequence as the per-layer
    launches: planes and head outputs are bit-identical in the f16-class modes.  In fp32 the units of a layer's ragged last
    round are summed in parts (conv16_run's tail split: ((p0 + p1) + p2) + p3 instead of one chain), and which units those are
    depends on how many images a weight group walks, so fp32 agrees to rounding only.
    Three launch forms: "layers" (GIGA_LAYERWISE_UNET), the default (persistent for every batch size in the f16-class modes,
    from 8 scenes up in fp32) and the forced persistent launch (GIGA_PERSIST_UNET).  Batches of 1-10 scenes put one image or
    none on a group; 11 and 32 scenes several, unevenly."""
    x = torch.from_numpy(synth.tsdf_batch(500, B)).to(dev)
    p = torch.from_numpy(synth.query_points(500, B, 64, stream=9)).to(dev)
    try:
        for prec in ("fp32", "fp16x3", "fp16"):
            net.set_precision(prec)
            got = {}
            for flag in ("layers", False, True):
                net.set_persistent_unet(flag)
                with torch.no_grad():
                    for _ in range(3):
                        planes = net.encode_inputs(x)
                        out = net(x, p, p_tsdf=p)
                got[flag] = [planes[k].clone() for k in ("xz", "xy", "yz")] + [o.clone() for o in out]
            for flag in (False, True):
                for a, b in zip(got["layers"], got[flag]):
                    if prec == "fp32":
                        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max())), (prec, flag)
                    else:
                        assert torch.equal(a, b), (prec, flag)
    finally:
        net.set_persistent_unet(False)
        net.set_precision("fp32")


def test_forget_device_state_between_calls(net, dev):
    """giga_forget_device_state() drops the library's per-device bookkeeping (raised dynamic-LDS limits, persistent launches in
    flight: what a host calls after hipDeviceReset); the next calls must set everything up again and give the same bits."""
    from giga_amd import _capi
    x = torch.from_numpy(synth.tsdf_batch(40, 3)).to(dev)
    p = torch.from_numpy(synth.query_points(40, 3, 256, stream=2)).to(dev)
    try:
        for prec in ("fp32", "fp16x3", "fp16"):
            net.set_precision(prec)
            with torch.no_grad():
                a = [t.clone() for t in net(x, p, p_tsdf=p)]
                torch.cuda.synchronize()
                _capi.lib().giga_forget_device_state()
                b = net(x, p, p_tsdf=p)
            for u, v in zip(a, b):
                assert torch.equal(u, v), prec
    finally:
        net.set_precision("fp32")
