"""GPU parity of the training path (scripts/train_giga.py:198-211): loss and the gradient of EVERY parameter
tensor from the HIP backward against torch autograd through the CPU oracle, and against the reference-
generated golden G4 (losses + per-tensor gradient norms).  fp32; weight gradients are reduced with atomics,
so the tolerance is relative (2e-3 of the tensor's max |grad|, plus 1e-6 absolute)."""
import numpy as np
import pytest
import torch

from giga_amd import networks, synth, weights
from giga_amd.training import giga_loss
from oracle import giga_oracle as O

pytestmark = pytest.mark.gpu


def ref_style_loss(out, y):
    """The caller-side helpers of scripts/train_giga.py:154-195 (the oracle's restatement, plain torch on the device):
    what the reference's own training script runs on top of the drop-in network."""
    return O.train_loss(O.train_select(out), y)


LOSSES = {"torch-helpers": ref_style_loss, "fused": giga_loss}


def _batch(first, B, M):
    x = torch.from_numpy(synth.tsdf_batch(first, B))
    pos = torch.from_numpy(synth.query_points(first, B, 1, stream=2))
    pos_occ = torch.from_numpy(synth.query_points(first, B, M, stream=3))
    y = tuple(torch.from_numpy(a) for a in synth.train_labels(first, B, M))
    return x, pos, pos_occ, y


def _oracle_grads(sd, x, pos, pos_occ, y, detach_tsdf=False):
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.model_forward(sdg, x, pos, p_tsdf=pos_occ, detach_tsdf=detach_tsdf)
    loss, d = O.train_loss(O.train_select(out), y)
    loss.backward()
    return loss.item(), {k: v.grad for k, v in sdg.items()}, d


@pytest.mark.parametrize("loss_kind", ["torch-helpers", "fused"])
def test_train_step_gradients_match_oracle_and_g4(golden, sd7, loss_kind):
    dev = torch.device("cuda:0")
    g4 = golden("g4_train_step.npz")
    B, M, s0 = int(g4["B"]), int(g4["M"]), int(g4["first_scene"])
    x, pos, pos_occ, y = _batch(s0, B, M)
    ref_loss, ref_grads, ref_d = _oracle_grads(sd7, x, pos, pos_occ, y)
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    loss, d = LOSSES[loss_kind](out, tuple(t.to(dev) for t in y))
    for k in ("loss_qual", "loss_rot", "loss_width", "loss_occ", "loss_all"):
        assert abs(d[k].item() - float(g4[k])) <= 1e-4 * max(1.0, abs(float(g4[k]))), k
    assert abs(loss.item() - ref_loss) < 1e-5
    loss.backward()
    worst = []
    for name, prm in net.named_parameters():
        ref = ref_grads[name]
        got = prm.grad.detach().cpu()
        assert got.shape == ref.shape, name
        scale = ref.abs().max().item()
        err = (got - ref).abs().max().item()
        worst.append((err / (scale + 1e-12), name, err, scale))
        assert err <= 2e-3 * scale + 1e-6, (name, err, scale)
    names = [str(n) for n in g4["grad_names"]]
    got_norms = {n: p.grad.double().norm().item() for n, p in net.named_parameters()}
    for n, ref in zip(names, g4["grad_norms"]):
        assert abs(got_norms[n] - ref) <= 2e-3 * max(ref, 1e-6) + 1e-8, (n, got_norms[n], ref)


@pytest.mark.parametrize("M,loss_kind", [(96, "torch-helpers"), (2048, "fused")])
def test_gradients_at_batch_32(sd7, M, loss_kind):
    """From 32 scenes up the conv_in kernels (forward and backward) switch to the one-x-part decomposition; hold that
    variant's gradients to the oracle too, including the full BASELINE c5 shape (B = 32, M = 2048)."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = _batch(300, 32, M)
    ref_loss, ref_grads, _ = _oracle_grads(sd7, x, pos, pos_occ, y)
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    loss, _ = LOSSES[loss_kind](net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
    assert abs(loss.item() - ref_loss) < 1e-5
    loss.backward()
    for name, prm in net.named_parameters():
        ref, got = ref_grads[name], prm.grad.detach().cpu()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-3 * scale + 1e-6, name


def test_plane_gradient_gather_with_all_queries_in_one_cell(sd7):
    """The worst case of the binned gather: every occupancy query of a scene in ONE pixel cell (scene 0: 4096 points inside a
    1e-3 cube; scene 1: all points clamped onto one corner of the volume) -- cell lists of 4096 points, which the workgroup ranks
    cooperatively instead of one lane sorting them by insertion (8 M serial LDS steps: the step took milliseconds).  Gradients
    against the oracle; the step must also stay in the time of a normal one."""
    import time
    dev = torch.device("cuda:0")
    B, M = 2, 4096
    x, pos, pos_occ, y = _batch(710, B, M)
    g = torch.Generator().manual_seed(5)
    pos_occ = pos_occ.clone()
    pos_occ[0] = torch.tensor([0.1013, -0.2031, 0.3047]) + 1e-3 * torch.rand(M, 3, generator=g)
    pos_occ[1] = torch.tensor([0.7, 0.9, -0.8]) + 0.05 * torch.rand(M, 3, generator=g)         # outside: clamped to the corner cell
    ref_loss, ref_grads, _ = _oracle_grads(sd7, x, pos, pos_occ, y)
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    times = []
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _ = giga_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
        loss.backward()
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    assert abs(loss.item() - ref_loss) < 1e-5
    for name, ref in ref_grads.items():
        got = net.get_parameter(name).grad.cpu()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-3 * scale + 1e-6, name
    assert min(times) < 0.02, times                          # (a serial sort of 4096 points per cell costs far more than 20 ms)


@pytest.mark.parametrize("B,M", [(3, 301), (2, 4096), (5, 256)])
def test_plane_gradient_gather_path(sd7, B, M):
    """From 256 occupancy queries per scene up (to 4096) the plane gradient of the occupancy head is built by a binned gather
    (csrc/giga_decoder_bwd.hip::plane_gather_kernel: sample_plane_feature backward, decoder.py:117-122 through autograd) instead
    of fp32 atomics.  Ragged sizes (M not a multiple of 4 / 64, the LDS capacity), queries ON the cube's faces and corners
    (footprints in the last pixel row / column, the clamped coordinates of common.py:238-261) and many queries in ONE pixel
    cell (long cell lists); gradients against the oracle, and the encoder gradients -- which only see the decoders through
    the plane gradient -- again after a second identical step (the gather sums in a fixed order; what remains of run-to-run
    differences are the atomics of the weight-gradient reductions)."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = _batch(700, B, M)
    pos_occ = pos_occ.clone()
    pos_occ[:, 0:8] = torch.tensor([[-0.5, -0.5, -0.5], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [-0.5, 0.5, -0.5],
                                    [0.5, 0.0, 0.0], [0.0, 0.5, 0.0], [0.0, 0.0, 0.5], [0.6, -0.7, 0.55]])
    pos_occ[:, 8:72] = torch.tensor([0.1013, -0.2031, 0.3047]) + 1e-3 * torch.rand(64, 3, generator=torch.Generator().manual_seed(3))
    ref_loss, ref_grads, _ = _oracle_grads(sd7, x, pos, pos_occ, y)
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
        assert abs(loss.item() - ref_loss) < 1e-5
        loss.backward()
        runs.append({n: q.grad.detach().clone() for n, q in net.named_parameters()})
    for name, ref in ref_grads.items():
        got = runs[0][name].cpu()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-3 * scale + 1e-6, name
        if name.startswith("encoder."):
            assert (runs[0][name] - runs[1][name]).abs().max().item() <= 1e-4 * scale + 1e-7, name


def test_bf16_training_step(sd7, monkeypatch):
    """BASELINE c5 arithmetic: bf16 operands (fp32 accumulate) in the U-Net's forward and data-gradient convolutions.
    bf16 keeps 8 significant bits, and this U-Net has no normalisation layers: on the synthetic weights the planes move by ~1 %
    of their range, which flips ReLU masks downstream, so per-tensor gradients of the WHOLE bf16 step differ from fp32 autograd
    by 10-30 % (tests/diag/gpu_bf16_diag.py) -- that is the number format, not the kernels.  The kernels are therefore held to
    (a) forward: EVERY U-Net layer equals the oracle's layer evaluated on the same input with bf16-ROUNDED operands and fp32
        accumulation (2e-5 of the layer's range: only the summation order differs);
    (b) backward: with an fp32 forward, bf16 data-gradient convolutions change no parameter gradient by more than 2 % (rel. L2,
        measured 0.7 %) under a smooth objective;
    (c) the whole bf16 step: joint loss within 1 % of fp32, gradient direction preserved (cosine > 0.97, measured 0.99);
    (d) the bf16 fragment images derived on the device equal the host packer's."""
    from giga_amd import _capi
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = _batch(300, 32, 256)
    bf = lambda t: t.bfloat16().float()  # noqa: E731
    # (a) layer by layer: the oracle's layer on the GPU's OWN stage input, operands rounded to bf16, against the GPU's stage output
    #     (whole-network comparisons decorrelate: a 1e-7 accumulation-order difference flips a rounding decision somewhere, and
    #     after ten layers two valid bf16 evaluations differ by the rounding noise itself, ~1 % -- tests/diag/gpu_bf16_stages.py)
    import ctypes
    import torch.nn.functional as F
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).eval().set_precision("bf16")
    Bs = 2
    with torch.no_grad():
        got = net.encode_inputs(x[:Bs].to(dev))
    ws = net.encoder._ws.snapshot()[-1]
    off = (ctypes.c_size_t * 17)()
    assert _capi.lib().giga_encoder_workspace_layout(Bs, _capi.PRECISION["bf16"], off) == 0
    names = ["P0", "A0", "S0", "Q0", "A1", "S1", "Q1", "A2", "S2", "U0", "A3", "A4", "U1", "A5", "A6"]
    ch = dict(zip(names, (32, 32, 32, 32, 64, 64, 64, 128, 128, 64, 64, 64, 32, 32, 32)))
    hw = dict(zip(names, (40, 40, 40, 20, 20, 20, 10, 10, 10, 20, 20, 20, 40, 40, 40)))

    def stage(nm):
        n = 3 * Bs * hw[nm] * hw[nm] * ch[nm]
        o = off[names.index(nm)]
        return ws[o:o + 4 * n].view(torch.float32).view(3 * Bs, hw[nm], hw[nm], ch[nm]).permute(0, 3, 1, 2).cpu()

    W = lambda k: bf(sd7["encoder.unet." + k + ".weight"])  # noqa: E731
    Bi = lambda k: sd7["encoder.unet." + k + ".bias"]  # noqa: E731
    c3 = lambda k, t: F.relu(F.conv2d(bf(t), W(k), Bi(k), padding=1))  # noqa: E731
    up = lambda k, t: F.conv_transpose2d(bf(t), W(k), Bi(k), stride=2)  # noqa: E731
    layers = [("A0", lambda: c3("down_convs.0.conv1", stage("P0"))), ("S0", lambda: c3("down_convs.0.conv2", stage("A0"))),
              ("Q0", lambda: F.max_pool2d(stage("S0"), 2, 2)), ("A1", lambda: c3("down_convs.1.conv1", stage("Q0"))),
              ("S1", lambda: c3("down_convs.1.conv2", stage("A1"))), ("Q1", lambda: F.max_pool2d(stage("S1"), 2, 2)),
              ("A2", lambda: c3("down_convs.2.conv1", stage("Q1"))), ("S2", lambda: c3("down_convs.2.conv2", stage("A2"))),
              ("U0", lambda: up("up_convs.0.upconv", stage("S2"))),
              ("A3", lambda: c3("up_convs.0.conv1", torch.cat((stage("U0"), stage("S1")), 1))),
              ("A4", lambda: c3("up_convs.0.conv2", stage("A3"))), ("U1", lambda: up("up_convs.1.upconv", stage("A4"))),
              ("A5", lambda: c3("up_convs.1.conv1", torch.cat((stage("U1"), stage("S0")), 1))),
              ("A6", lambda: c3("up_convs.1.conv2", stage("A5")))]
    for nm, fn in layers:
        want = fn()
        err = (stage(nm) - want).abs().max().item()
        assert err <= 2e-5 * max(1.0, want.abs().max().item()), (nm, err)
    final = F.conv2d(bf(stage("A6")), W("conv_final"), Bi("conv_final"))
    planes = torch.stack([got[k] for k in O.PLANES]).reshape(3 * Bs, 32, 40, 40).cpu()
    assert (planes - final).abs().max().item() <= 2e-5 * final.abs().max().item()
    exact = O.encoder_forward(sd7, x[:Bs])
    e_exact = max(maxerr_t(got[k], exact[k]) for k in O.PLANES)
    assert 1e-3 < e_exact < 8e-2, e_exact                    # bf16-level deviation from the fp32 planes (measured ~2.5e-2)
    # (b) fp32 forward + bf16 dgrad, smooth objective
    g5 = torch.Generator().manual_seed(5)
    R = [torch.randn(32, 1, generator=g5), torch.randn(32, 1, 4, generator=g5), torch.randn(32, 1, generator=g5),
         torch.randn(32, 256, generator=g5) / 16]
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd7.items()}
    sum((o * r).sum() for o, r in zip(O.model_forward(sdg, x, pos, p_tsdf=pos_occ), R)).backward()
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train().set_train_precision("bf16_convs")
    monkeypatch.setattr(_capi, "ENC_BF16", 0)                # forward stays fp32 for this part (and the decoders: "bf16_convs")
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    sum((o * r.to(dev)).sum() for o, r in zip(out, R)).backward()
    worst = max((((p.grad.cpu() - sdg[n].grad).norm() / (sdg[n].grad.norm() + 1e-12)).item(), n) for n, p in net.named_parameters())
    print("bf16 dgrad convolutions, fp32 forward: worst relative L2 gradient error", worst)
    assert worst[0] <= 2e-2, worst
    monkeypatch.undo()
    # (c) the whole bf16 step on the joint loss: bf16 convolutions AND bf16 decoder heads (tests/test_gpu_train16.py holds the
    #     decoder kernels to their operand-rounded reference)
    net.set_train_precision("bf16")
    net.zero_grad(set_to_none=True)
    ref_loss, ref_grads, _ = _oracle_grads(sd7, x, pos, pos_occ, y)
    loss, _ = giga_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
    assert abs(loss.item() - ref_loss) < 1e-2 * abs(ref_loss)
    loss.backward()
    fg = torch.cat([p.grad.reshape(-1).cpu() for _, p in net.named_parameters()])
    fr = torch.cat([ref_grads[n].reshape(-1) for n, _ in net.named_parameters()])
    cos = torch.nn.functional.cosine_similarity(fg, fr, dim=0).item()
    print("bf16 step, joint loss:", loss.item(), "vs fp32", ref_loss, "gradient cosine", cos)
    assert cos > 0.97
    # (d) device-derived bf16 images == host packer's
    st = net._train_state
    flat = torch.cat([p.detach().reshape(-1) for p in net._ordered_params()]).cpu()
    host_fwd, host_bwd = _capi.pack_weights(flat, 15), _capi.pack_bwd_weights(flat, 15)
    # forward blob: the bf16 conv fragments are its last region (giga_layout.h); the training blob leaves the f16 images empty
    conv = [(0, 32, 32), (0, 32, 32), (0, 32, 64), (0, 64, 64), (0, 64, 128), (0, 128, 128), (1, 128, 64), (0, 128, 64), (0, 64, 64),
            (1, 64, 32), (0, 64, 32), (0, 32, 32), (2, 32, 32)]
    tail = sum((co // 16 * (4 if k == 1 else 1)) * (9 if k == 0 else 1) * (ci // 32) * 1024 for k, ci, co in conv)
    # (the conv32 images of round 4 -- f16, f16x3 pairs, bf16: 4 fragments per (32-channel slice, tap, 16-channel chunk) -- follow them)
    c32 = sum(4 * (co // 32 * (4 if k == 1 else 1)) * (9 if k == 0 else 1) * (ci // 16) * 1024 for k, ci, co in conv)
    wino = sum(16 * ci * co * 4 for k, ci, co in conv if k == 0)    # round 6: the Winograd images of the 3x3 layers (giga_wino.h)
    n = st.blob.numel() - 4 * 59 * 1024 - wino - 256         # (behind: the bf16 decoder images of round 5 -- test_gpu_train16.py --, the Winograd images and the stamp)
    w0 = st.blob.numel() - 256 - wino
    _capi.check(_capi.lib().giga_derive_winograd(_capi.ptr(st.blob), _capi.ptr(st.bwd_blob), _capi.stream_ptr(dev)), "giga_derive_winograd")   # (the fp32 step's derive)
    torch.cuda.synchronize()
    assert torch.equal(st.blob.cpu()[w0:w0 + wino], host_fwd[w0:w0 + wino]), "device-derived Winograd images != host pack (bit for bit)"
    wb = st.bwd_blob.numel() - 256 - wino                      # (the backward blob ends the same way: Winograd images of the data-gradient convolutions, stamp)
    assert torch.equal(st.bwd_blob.cpu()[wb:wb + wino], host_bwd[wb:wb + wino]), "device-derived data-gradient Winograd images != host pack"
    assert host_bwd[wb:wb + wino].any()
    assert torch.equal(st.blob.cpu()[n - c32 - tail:n - c32], host_fwd[n - c32 - tail:n - c32]), \
        "device repack + derive != host pack (forward bf16 fragments)"
    assert torch.equal(st.bwd_blob.cpu()[:-256], host_bwd[:-256]), "device repack + derive != host pack (backward blob)"   # (last 256 B: the host blob's stamp)
    assert _capi.lib().giga_derive_bf16_fragments(None, None, None) == -1
    # and a few optimizer steps in bf16 reduce the loss like the fp32 run does
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)
    first = None
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
        loss.backward(); opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < 0.7 * first


def test_bf16_training_step_at_the_c5_shape(sd7):
    """BASELINE config c5 at its full shape (B = 32 scenes, 1 grasp query + M = 2048 occupancy queries): the whole bf16 step
    against fp32 autograd through the oracle -- joint loss within 1 %, every loss term within 2 %, gradient direction preserved
    (cosine > 0.97 over all 581 863 parameters, and per head / encoder) -- and bit-for-bit repeatable forward outputs.  The
    per-kernel bf16 parity (operand-rounded oracle, layer by layer) is test_bf16_training_step (a); per-tensor gradients of a
    whole bf16 step differ from fp32 by the format's 10-30 % (see there), so they are not compared tensor by tensor."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = _batch(300, 32, 2048)
    ref_loss, ref_grads, ref_d = _oracle_grads(sd7, x, pos, pos_occ, y)
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train().set_train_precision("bf16")
    yd = tuple(t.to(dev) for t in y)
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    loss, d = giga_loss(out, yd)
    assert abs(loss.item() - ref_loss) < 1e-2 * abs(ref_loss), (loss.item(), ref_loss)
    for k in ("loss_qual", "loss_rot", "loss_width", "loss_occ"):
        assert abs(d[k].item() - ref_d[k].item()) <= 2e-2 * max(abs(ref_d[k].item()), 1e-3), (k, d[k].item(), ref_d[k].item())
    loss.backward()
    names = [n for n, _ in net.named_parameters()]
    got = {n: p.grad.detach().reshape(-1).cpu() for n, p in net.named_parameters()}
    cos = lambda sel: torch.nn.functional.cosine_similarity(torch.cat([got[n] for n in sel]),  # noqa: E731
                                                            torch.cat([ref_grads[n].reshape(-1) for n in sel]), dim=0).item()
    total = cos(names)
    print("bf16 step at B=32, M=2048: loss", loss.item(), "vs fp32", ref_loss, "gradient cosine", total)
    assert total > 0.97
    for part in ("decoder_qual", "decoder_rot", "decoder_width", "decoder_tsdf", "encoder"):
        c = cos([n for n in names if n.startswith(part)])
        print("   gradient cosine,", part, c)
        assert c > (0.90 if part == "encoder" else 0.95), (part, c)
    for n in names:
        assert torch.isfinite(got[n]).all(), n
    out2 = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    for a, b in zip(out, out2):
        assert torch.equal(a.detach(), b.detach())


def maxerr_t(a, b):
    return (a.detach().float().cpu() - b.detach().float()).abs().max().item()


def test_fused_loss_matches_reference_helpers_g11(golden):
    """giga_loss (csrc/giga_loss.hip) against golden G11 -- the reference's OWN select + loss_fn on fixed head outputs --
    and its gradients against autograd through the torch restatement of those helpers."""
    dev = torch.device("cuda:0")
    g = golden("g11_train_helpers.npz")
    B, M, s0 = int(g["B"]), int(g["M"]), int(g["first_scene"])
    t = torch.from_numpy
    y = tuple(t(a) for a in synth.train_labels(s0, B, M))
    heads = [t(g["qual"]), t(g["rot"]), t(g["width"]), t(g["logit"])]
    hd = [h.clone().to(dev).requires_grad_(True) for h in heads]
    loss, d = giga_loss(tuple(hd), tuple(a.to(dev) for a in y))
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    for k in ("loss_qual", "loss_rot", "loss_width", "loss_occ", "loss_all"):
        assert abs(float(d[k]) - float(g[k])) <= 2e-6 * max(1.0, abs(float(g[k]))), k
    (3.0 * loss).backward()                                  # a non-trivial upstream gradient
    hc = [h.clone().requires_grad_(True) for h in heads]
    ref_loss, _ = ref_style_loss(tuple(hc), y)
    (3.0 * ref_loss).backward()
    for a, b, name in zip(hd, hc, ("qual", "rot", "width", "occ")):
        scale = b.grad.abs().max().item()
        assert (a.grad.cpu() - b.grad).abs().max().item() <= 1e-5 * scale + 1e-9, name
    # label = 0 scenes get no rotation / width gradient; saturated logits keep ATen's clamped BCE arithmetic
    z = torch.tensor([[-120.0, -30.0, 0.0, 30.0, 120.0]] * 2)
    hz = [torch.tensor([[1e-9], [1.0 - 1e-7]]), torch.nn.functional.normalize(torch.randn(2, 1, 4), dim=2), torch.rand(2, 1), z]
    yz = (torch.tensor([0.0, 1.0]), torch.nn.functional.normalize(torch.randn(2, 2, 4), dim=2), torch.rand(2) * 0.3,
          torch.tensor([[1.0, 0.0, 1.0, 0.0, 0.0]] * 2))
    a = [h.clone().to(dev).requires_grad_(True) for h in hz]
    b = [h.clone().requires_grad_(True) for h in hz]
    la, _ = giga_loss(tuple(a), tuple(v.to(dev) for v in yz))
    lb, _ = ref_style_loss(tuple(b), yz)
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    la.backward(); lb.backward()
    for u, v in zip(a, b):
        assert torch.allclose(u.grad.cpu(), v.grad, rtol=1e-4, atol=1e-6)


def test_interleaved_forwards_and_recycled_buffers(sd7):
    """The large step buffers are recycled.  Two graphs alive at once (same weights) get separate sets and both backward
    correctly; a second backward through a finished graph, and a backward after the weights were re-packed for a later
    forward (an optimizer step in between), raise instead of silently using stale data."""
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train()
    x, pos, pos_occ, y = (t.to(dev) if torch.is_tensor(t) else tuple(a.to(dev) for a in t) for t in _batch(70, 2, 64))
    la, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    la.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    with pytest.raises(RuntimeError):
        la.backward()                                        # autograd / the recycled-buffer check: the graph is spent
    net.zero_grad(set_to_none=True)
    l1, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    l2, _ = giga_loss(net(x.flip(0), pos.flip(0), p_tsdf=pos_occ.flip(0)), tuple(t.flip(0) for t in y))
    l1.backward()                                            # the second forward must not have clobbered these activations
    for p, g in zip(net.parameters(), g1):
        assert (p.grad - g).abs().max().item() <= 2e-3 * g.abs().max().item() + 1e-6
    net.zero_grad(set_to_none=True)
    l2.backward()                                            # same batch with the scenes permuted: same gradients
    for p, g in zip(net.parameters(), g1):
        assert (p.grad - g).abs().max().item() <= 2e-3 * g.abs().max().item() + 1e-6
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    l3, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
    opt.step()                                               # weights change ...
    l4, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)        # ... and this forward rebuilds the weight images
    with pytest.raises(RuntimeError, match="re-packed"):
        l3.backward()
    l4.backward()


def test_flattened_parameters_train_identically(sd7):
    """net.flatten_parameters(): one flat leaf, the named parameters become views.  Same losses as the 164-tensor run over five
    fused-Adam steps, reference state-dict keys and shapes unchanged, inference sees the trained weights, .to() un-flattens."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = (t.to(dev) if torch.is_tensor(t) else tuple(a.to(dev) for a in t) for t in _batch(55, 4, 256))
    runs = []
    for flat in (False, True):
        net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train()
        prm = net.flatten_parameters() if flat else list(net.parameters())
        assert (len(prm) == 1 and prm[0].numel() == 581863) if flat else len(prm) == 164
        opt = torch.optim.Adam(prm, lr=1e-4, fused=True)
        losses = []
        for _ in range(5):
            opt.zero_grad(set_to_none=True)
            loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
            loss.backward(); opt.step(); losses.append(loss.item())
        runs.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, net))
    (l0, s0, _), (l1, s1, netf) = runs
    assert np.allclose(l0, l1, rtol=1e-4, atol=0), (l0, l1)      # (weight gradients are reduced with atomics: not bit-identical)
    assert list(s0.keys()) == list(s1.keys()) == list(sd7.keys())
    # Adam normalises every element's step to ~lr: an element whose gradient is at rounding-noise level (the plane gradients are
    # summed with atomics: two runs of the SAME configuration differ there in the last bit) takes +lr in one run and -lr in the
    # other, so single elements are 2-3 lr apart after five steps in any two runs (measured: discrete outcomes 5e-7, 2.5e-5, 6.7e-5,
    # 2.3e-4 for lr = 1e-4, flat or not).  Bound: half of what five steps can move an element apart, and all but 1e-3 of the elements
    # within a fifth of one step.
    for k in s0:
        d = (s0[k] - s1[k]).abs()
        assert s1[k].shape == sd7[k].shape and d.max().item() < 5e-4, k
    far = sum((s0[k] - s1[k]).abs().gt(2e-5).sum().item() for k in s0) / sum(v.numel() for v in s0.values())
    assert far < 1e-3, far
    with torch.no_grad():
        out = netf(x, pos, p_tsdf=pos_occ)
        ref = O.model_forward({k: v.cpu() for k, v in s1.items()}, x.cpu(), pos.cpu(), p_tsdf=pos_occ.cpu())
    for a, r in zip(out, ref):
        assert (a.cpu() - r).abs().max().item() < 1e-4
    netf = netf.to(dev)                                       # (a no-op move still goes through _apply)
    assert netf.__dict__.get("_flat_param") is None and all(q.requires_grad for q in netf.parameters())


def test_inference_after_fused_adam_steps_uses_the_new_weights(sd7):
    """torch.optim.Adam(fused=True) updates parameters without bumping their version counters, so the packed-weight cache
    of the inference path cannot see those steps; the training forward marks it stale instead.  No eval()/train() toggle
    here on purpose."""
    dev = torch.device("cuda:0")
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev)
    x, pos, pos_occ, y = (t.to(dev) if torch.is_tensor(t) else tuple(a.to(dev) for a in t) for t in _batch(20, 2, 128))
    with torch.no_grad():
        before = net(x, pos, p_tsdf=pos_occ)                 # packs + caches the blob
    opt = torch.optim.Adam(net.parameters(), lr=1e-2, fused=True)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
        loss.backward(); opt.step()
    with torch.no_grad():
        after = net(x, pos, p_tsdf=pos_occ)
        ref = O.model_forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, x.cpu(), pos.cpu(), p_tsdf=pos_occ.cpu())
    assert (after[3] - before[3]).abs().max().item() > 1e-3      # the weights really moved
    for a, r in zip(after, ref):
        assert (a.cpu() - r).abs().max().item() < 1e-4


def test_giga_detach_matches_reference_golden_g8(golden, sd7):
    """Per-tensor gradient norms of the reference's own giga_detach network on the G4 batch (golden G8)."""
    dev = torch.device("cuda:0")
    g = golden("g8_detach.npz")
    B, M, s0 = int(g["B"]), int(g["M"]), int(g["first_scene"])
    x, pos, pos_occ, y = _batch(s0, B, M)
    net = networks.get_network("giga_detach")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    loss, _ = ref_style_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
    assert abs(loss.item() - float(g["loss_all"])) < 1e-4
    loss.backward()
    got = {n: p.grad.double().norm().item() for n, p in net.named_parameters()}
    for n, ref in zip([str(n) for n in g["grad_names"]], g["grad_norms"]):
        assert abs(got[n] - ref) <= 2e-3 * max(ref, 1e-6) + 1e-8, (n, got[n], ref)


def test_giga_detach_gradients(sd7):
    """giga_detach (networks.py:143-169): the occupancy loss must not reach the encoder; the heads are unchanged."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = _batch(50, 3, 257)
    ref_loss, ref_grads, _ = _oracle_grads(sd7, x, pos, pos_occ, y, detach_tsdf=True)
    _, attached, _ = _oracle_grads(sd7, x, pos, pos_occ, y)
    net = networks.get_network("giga_detach")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    loss, _ = ref_style_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
    assert abs(loss.item() - ref_loss) < 1e-5
    loss.backward()
    differs = 0
    for name, prm in net.named_parameters():
        ref, got = ref_grads[name], prm.grad.detach().cpu()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-3 * scale + 1e-6, name
        if name.startswith("encoder.") and (attached[name] - ref).abs().max().item() > 1e-2 * scale:
            differs += 1
    assert differs > 0          # the detached and attached encoder gradients really are different


def test_sgd_steps_track_the_oracle(sd7):
    """Three optimizer steps (device-side repack each step) follow the oracle's trajectory."""
    dev = torch.device("cuda:0")
    B, M = 2, 512
    x, pos, pos_occ, y = _batch(30, B, M)
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train()
    opt = torch.optim.SGD(net.parameters(), lr=1e-2)
    sd = {k: v.clone() for k, v in sd7.items()}
    losses, ref_losses = [], []
    for _ in range(3):
        opt.zero_grad()
        loss, _ = ref_style_loss(net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev)), tuple(t.to(dev) for t in y))
        loss.backward(); opt.step(); losses.append(loss.item())
        rl, rg, _ = _oracle_grads(sd, x, pos, pos_occ, y)
        sd = {k: v - 1e-2 * rg[k] for k, v in sd.items()}
        ref_losses.append(rl)
    assert np.allclose(losses, ref_losses, rtol=0, atol=2e-4), (losses, ref_losses)
    assert losses[-1] < losses[0]


def test_training_steps_do_not_accumulate_device_memory(sd7):
    """The autograd node must not keep its activation workspace (207 MB at B=32) alive past the backward: with the
    cyclic GC off, the allocated bytes after step 6 equal those after step 2 (a node holding its own outputs in a
    plain attribute is a reference cycle that only the GC frees)."""
    import gc
    dev = torch.device("cuda:0")
    net = networks.get_network("giga")
    net.load_state_dict(sd7)
    net = net.to(dev).train()
    x, pos, pos_occ, y = (t.to(dev) if torch.is_tensor(t) else tuple(a.to(dev) for a in t) for t in _batch(40, 8, 512))
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    gc.collect()
    gc.disable()
    try:
        seen = []
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
            loss.backward()
            opt.step()
            del loss
            torch.cuda.synchronize()
            seen.append(torch.cuda.memory_allocated(dev))
    finally:
        gc.enable()
    assert seen[5] - seen[1] < (1 << 20), seen


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adam_matches_torch_adam(wd):
    """giga_amd.optim.FlatAdam (one HIP launch, giga_adam_step) = torch.optim.Adam (scripts/train_giga.py:49) step for step on
    a buffer of the model's size (581 863 elements: exercises the n % 4 tail); same state-dict layout."""
    from giga_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    n = 581863
    p0 = torch.randn(n, generator=g) * 0.1
    a = torch.nn.Parameter(p0.clone().to(dev))
    b = torch.nn.Parameter(p0.clone().to(dev))
    oa = torch.optim.Adam([a], lr=2e-4, weight_decay=wd)
    ob = FlatAdam([b], lr=2e-4, weight_decay=wd)
    for it in range(6):
        grad = (torch.randn(n, generator=g) * (10.0 ** (it % 3 - 1))).to(dev)
        a.grad = grad.clone(); b.grad = grad.clone()
        oa.step(); ob.step()
        assert (a.detach() - b.detach()).abs().max().item() < 2e-7, it        # steps are <= lr = 2e-4: 1e-3 relative of a step
    sa, sb = oa.state_dict()["state"][0], ob.state_dict()["state"][0]
    assert set(sa) == set(sb) and float(sa["step"]) == float(sb["step"]) == 6.0
    # moments: fp32 rounding of two formulations of the same update (lerp vs b1 m + (1 - b1) g)
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("B", [2, 32])
def test_bf16_weight_gradient_kernels_with_rounded_operands(sd7, B):
    """Every bf16 3x3 weight-gradient kernel (csrc/giga_encoder_bwd.hip: conv3_wgrad_bf16_kernel and the tap-per-wave variant
    conv3_wgrad_bf16_taps_kernel, selected per layer) held to what it is supposed to compute, layer by layer, on the operands the
    GPU itself used:  dW[co][ci][ky][kx] = sum over images and pixels of bf16(dY[y, x, co]) * bf16(X[y + ky - 1, x + kx - 1, ci])
    with fp32 accumulation, X zero outside the image (encoder/unet.py:14-23: padding 1) -- so a dropped tap at an image border, a
    wrong strip boundary or a stale operand shows as a large relative error of ONE tap, which the 2 % whole-network bound of
    test_bf16_training_step cannot see.  dY and X are read back from the workspaces the C ABI exposes
    (giga_encoder_workspace_layout, giga_backward_workspace_layout); bias gradients are the column sums of the UNROUNDED dY.
    Tolerance: 2e-5 of the tensor's range + the fp32 summation noise of ~1e5 products (1e-5 relative of the sum of |products|).
    B = 2 runs the small-batch strip schedules, B = 32 the c5 shape."""
    import ctypes

    import torch.nn.functional as F
    from giga_amd import _capi
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = _batch(700, B, 256)
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train().set_train_precision("bf16")
    g5 = torch.Generator().manual_seed(11)
    R = [torch.randn(B, 1, generator=g5), torch.randn(B, 1, 4, generator=g5), torch.randn(B, 1, generator=g5),
         torch.randn(B, 256, generator=g5) / 16]
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    sum((o * r.to(dev)).sum() for o, r in zip(out, R)).backward()
    torch.cuda.synchronize()
    st = net._train_state
    sb = [b for b in st._pool if b.key == (B, 1, 256)][-1]
    fo, bo = (ctypes.c_size_t * 17)(), (ctypes.c_size_t * 15)()
    assert _capi.lib().giga_encoder_workspace_layout(B, _capi.PRECISION["bf16"], fo) == 0
    assert _capi.lib().giga_backward_workspace_layout(B, bo) == 0
    fn = ["P0", "A0", "S0", "Q0", "A1", "S1", "Q1", "A2", "S2", "U0", "A3", "A4", "U1", "A5", "A6"]
    bn = ["gA6", "gA5", "gC1", "gA4", "gA3", "gC0", "gS2", "gA2", "gQ1", "gS1", "gA1", "gQ0", "gS0", "gA0", "gP0"]
    ch = dict(zip(fn, (32, 32, 32, 32, 64, 64, 64, 128, 128, 64, 64, 64, 32, 32, 32)))
    hw = dict(zip(fn, (40, 40, 40, 20, 20, 20, 10, 10, 10, 20, 20, 20, 40, 40, 40)))

    def act(nm):                                             # forward activation, NCHW fp64 with bf16-rounded values
        n = 3 * B * hw[nm] * hw[nm] * ch[nm]
        o = fo[fn.index(nm)]
        t = sb.ws[o:o + 4 * n].view(torch.float32).view(3 * B, hw[nm], hw[nm], ch[nm]).permute(0, 3, 1, 2)
        return t.bfloat16().double().cpu()

    def grad_of(nm):                                         # dLoss / d(activation nm), NCHW fp32 as the GPU left it
        n = 3 * B * hw[nm] * hw[nm] * ch[nm]
        o = bo[bn.index("g" + nm)]
        return sb.wsb[o:o + 4 * n].view(torch.float32).view(3 * B, hw[nm], hw[nm], ch[nm]).permute(0, 3, 1, 2).cpu()

    grads = dict(net.named_parameters())
    # (layer key, dY = gradient of its output activation, inputs in concatenation order)
    layers = [("down_convs.0.conv1", "A0", ("P0",)), ("down_convs.0.conv2", "S0", ("A0",)), ("down_convs.1.conv1", "A1", ("Q0",)),
              ("down_convs.1.conv2", "S1", ("A1",)), ("down_convs.2.conv1", "A2", ("Q1",)), ("down_convs.2.conv2", "S2", ("A2",)),
              ("up_convs.0.conv1", "A3", ("U0", "S1")), ("up_convs.0.conv2", "A4", ("A3",)),
              ("up_convs.1.conv1", "A5", ("U1", "S0")), ("up_convs.1.conv2", "A6", ("A5",))]
    worst = 0.0
    for key, ynm, xs in layers:
        dy32 = grad_of(ynm)
        dy = dy32.bfloat16().double()
        xin = torch.cat([act(nm) for nm in xs], 1)
        want = torch.nn.grad.conv2d_weight(xin, (dy.shape[1], xin.shape[1], 3, 3), dy, padding=1)
        # the sum of |products|: what fp32 accumulation noise scales with
        mag = torch.nn.grad.conv2d_weight(xin.abs(), (dy.shape[1], xin.shape[1], 3, 3), dy.abs(), padding=1)
        got = grads[f"encoder.unet.{key}.weight"].grad.double().cpu()
        err = (got - want).abs()
        tol = 2e-5 * float(want.abs().max()) + 1e-5 * mag
        bad = err > tol
        assert not bool(bad.any()), (key, B, int(bad.sum()), float(err.max()), float(want.abs().max()),
                                     [tuple(int(v) for v in i) for i in bad.nonzero()[:4]])
        # per-tap check: no tap of any (co, ci) pair may be off as a whole (a dropped border tap shifts a tap's mean)
        tap_err = (got - want).abs().mean(dim=(0, 1)) / want.abs().mean(dim=(0, 1))
        assert float(tap_err.max()) < 2e-3, (key, B, tap_err)
        worst = max(worst, float((err / (mag * 1e-5 + 2e-5 * want.abs().max())).max()))
        gb = grads[f"encoder.unet.{key}.bias"].grad.double().cpu()
        wb = dy32.double().sum(dim=(0, 2, 3))
        assert float((gb - wb).abs().max()) <= 2e-5 * max(1.0, float(wb.abs().max())) + 1e-6 * float(dy32.double().abs().sum(dim=(0, 2, 3)).max()), key
    print("bf16 weight gradients with rounded operands: worst error / tolerance", worst)


def test_conv_in_relu_mask_of_the_training_forward(sd7):
    """GIGA_CONVIN_MASK: the training forward leaves the sign bits of conv_in's pre-activations in the encoder workspace (the
    backward takes its ReLU mask from them instead of recomputing the convolution).  Layout (csrc/giga_encoder.hip): per
    (scene, workgroup row wy = 2 * iy-group + channel half, ix) and lane (j = lane & 15, g = lane >> 4) one 16-byte word; value
    k = (zg * 5 + ip) * 4 + r -- channel 16 (wy & 1) + j, iy = 10 (wy >> 1) + 2 ip + (g >> 1), iz = 8 zg + 4 (g & 1) + r -- in
    word k >> 5 at bit (n - 1 - (k & 31)), n = 32 (4 in the last word); a set bit = negative.  Against the oracle's Conv3d
    (voxels.py:36,106-107) for both batch regimes of the kernel (5 x-parts below 32 scenes, 1 from 32 up)."""
    import ctypes
    import torch.nn.functional as F
    from giga_amd import _capi
    dev = torch.device("cuda:0")
    for B in (3, 32):
        x, pos, pos_occ, y = _batch(700, B, 64)
        net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train()
        net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
        sb = net._train_state._pool[0]
        n = B * 8 * 40 * 64 * 4
        total = _capi.lib().giga_encoder_workspace_bytes(B, 0)
        words = sb.ws[total - ((n * 4 + 255) // 256 * 256):][:n * 4].view(torch.int32).cpu().numpy().view(np.uint32).reshape(B, 8, 40, 64, 4)
        pre = F.conv3d(x[:, None], sd7["encoder.conv_in.weight"], sd7["encoder.conv_in.bias"], padding=1).numpy()   # (B, 32, ix, iy, iz)
        bad = tot = 0
        lane = np.arange(64); j = lane & 15; g = lane >> 4
        for k in range(100):
            zg, ip, r = k // 20, (k // 4) % 5, k % 4
            wd, bit = k >> 5, (3 if k >= 96 else 31) - (k & 31)
            got = (words[..., wd] >> np.uint32(bit)) & 1                      # (B, 8, 40, 64)
            for wy in range(8):
                ch = 16 * (wy & 1) + j; iy = 10 * (wy >> 1) + 2 * ip + (g >> 1); iz = 8 * zg + 4 * (g & 1) + r
                ref = pre[:, ch, :, iy, iz].transpose(1, 2, 0)                # (lanes, B, ix) -> (B, ix, lanes)
                want = ref < 0
                sure = np.abs(ref) > 1e-5                                     # (fp32 summation order near zero)
                bad += int(((got[:, wy] != want) & sure).sum()); tot += int(sure.sum())
        assert bad == 0, (B, bad, tot)


def test_weight_gradients_on_the_side_stream(sd7, monkeypatch):
    """The encoder backward enqueues its weight gradients on a library-owned second stream (csrc/giga_encoder_bwd.hip: fork events
    behind the links of the data-gradient chain, one join) and reduces the 3x3 layers' partials in ONE launch.  Same gradients as the
    single-stream form with one reduce per layer (GIGA_WGRAD_STREAM=0, GIGA_WGRAD_ONE_REDUCE=0) up to the atomics of the plane
    gradients and of the decoders' weight gradients (1e-5 of the tensor's range -- measured up to 1.3e-6 at conv_in, the end of the
    chain; the reduces themselves run in the same order; a missing partial or a race is O(1)); also from a caller on its own torch stream,
    and again after giga_forget_device_state() dropped the library's stream handles."""
    from giga_amd import _capi
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = (t.to(dev) if torch.is_tensor(t) else tuple(a.to(dev) for a in t) for t in _batch(90, 5, 300))
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train()

    def grads():
        net.zero_grad(set_to_none=True)
        loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters()}

    monkeypatch.setenv("GIGA_WGRAD_STREAM", "0"); monkeypatch.setenv("GIGA_WGRAD_ONE_REDUCE", "0")
    ref = grads()
    monkeypatch.setenv("GIGA_WGRAD_STREAM", "1"); monkeypatch.setenv("GIGA_WGRAD_ONE_REDUCE", "1")

    def check(got):
        for n, r in ref.items():
            assert (got[n] - r).abs().max().item() <= 1e-5 * r.abs().max().item() + 1e-9, n

    for _ in range(3):
        check(grads())
    own = torch.cuda.Stream(device=dev)
    own.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(own):
        g = grads()
    own.synchronize()
    check(g)
    torch.cuda.synchronize()
    _capi.lib().giga_forget_device_state()                    # (no device reset here: the old handles are simply abandoned)
    check(grads())


def test_training_forward_backward_captured_in_a_hip_graph(sd7):
    """A caller may capture forward + loss + backward into a hipGraph (torch.cuda.graph): the backward's second stream is forked from
    and joined back into the capturing stream with events, so the capture takes both branches and ends with nothing left outside it.
    Replays reproduce the eager gradients (to the atomics' rounding)."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ, y = (t.to(dev) if torch.is_tensor(t) else tuple(a.to(dev) for a in t) for t in _batch(95, 4, 256))
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train()
    flat = net.flatten_parameters()[0]

    def step():
        loss, _ = giga_loss(net(x, pos, p_tsdf=pos_occ), y)
        loss.backward()
        return loss

    flat.grad = None
    step()
    ref = flat.grad.detach().clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):                                   # warm-up on the capture stream (allocator, lazy handles)
            flat.grad = None
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    flat.grad = torch.zeros_like(flat)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()
    for _ in range(2):
        flat.grad.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert (flat.grad - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert abs(loss.item() - step().item()) < 1e-6


@pytest.mark.parametrize("model,prec,B,N,M", [("giga", "fp32", 1, 1, 64), ("giga", "bf16", 7, 3, 300), ("giga", "bf16", 33, 1, 256),
                                              ("giga_aff", "fp32", 5, 2, 0), ("giga_aff", "bf16", 17, 1, 0), ("giga_detach", "bf16", 4, 1, 256)])
def test_backward_at_odd_shapes_with_and_without_the_side_stream(model, prec, B, N, M, monkeypatch):
    """Ragged batch sizes on both sides of the kernels' size switches (1, 7, 33 scenes; several grasp queries per scene; no occupancy
    head at all: giga_aff), fp32 and bf16: the backward with its weight gradients on the second stream and ONE reduce launch against the
    single-stream form with per-layer reduces -- same sums in the same order (fp32: 1e-5 of the tensor's range, what the decoders'
    atomics leave; bf16: 2e-3, a rounding flip in the heads' atomically summed plane gradients moves a bf16 operand) -- and finite."""
    dev = torch.device("cuda:0")
    sd = weights.make_state_dict(7, with_tsdf=model != "giga_aff")
    net = networks.get_network(model); net.load_state_dict(sd); net = net.to(dev).train().set_train_precision(prec)
    x = torch.from_numpy(synth.tsdf_batch(640, B)).to(dev)
    pos = torch.from_numpy(synth.query_points(640, B, N, stream=2)).to(dev)
    pos_occ = torch.from_numpy(synth.query_points(640, B, M, stream=3)).to(dev) if M else None
    g = torch.Generator().manual_seed(3)
    R = None

    def grads():
        nonlocal R
        net.zero_grad(set_to_none=True)
        out = net(x, pos, p_tsdf=pos_occ) if M else net(x, pos)
        if R is None:
            R = [torch.randn(o.shape, generator=g).to(dev) / max(1, o[0].numel()) for o in out]
        sum((o * r).sum() for o, r in zip(out, R)).backward()
        return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}

    monkeypatch.setenv("GIGA_WGRAD_STREAM", "0"); monkeypatch.setenv("GIGA_WGRAD_ONE_REDUCE", "0")
    ref = grads()
    monkeypatch.setenv("GIGA_WGRAD_STREAM", "1"); monkeypatch.setenv("GIGA_WGRAD_ONE_REDUCE", "1")
    got = grads()
    assert set(ref) == set(got) and len(ref) >= 100
    tol = 1e-5 if prec == "fp32" else 2e-3
    for n, r in ref.items():
        assert torch.isfinite(got[n]).all(), n
        assert (got[n] - r).abs().max().item() <= tol * r.abs().max().item() + 1e-9, n
