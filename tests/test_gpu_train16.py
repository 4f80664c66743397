"""GPU parity of the bf16 training decoder (csrc/giga_decoder_train16.hip; BASELINE c5 arithmetic in the decoder heads): forward
outputs, the gradient of EVERY decoder parameter and the plane gradients of the fused kernels against the operand-rounded
reference tests/dect_ref.py (plain torch; every product operand rounded to bf16 where the kernel rounds it, fp64 sums), on the
planes the GPU's own fp32 encoder produced.  What remains between the two is the accumulation order (fp32 MFMA trees against fp64
sums) and the bf16 roundings that a 1e-6 difference tips: one part in 2^9 of ONE operand, after which that point's later activations
round independently in the two evaluations (a few percent of the points; two valid bf16 evaluations of such a point differ by the
format's own noise, up to ~1 %).  Forward outputs: 95 % of the points within 3e-5 of the range, 99.5 % within 2e-3, all within 2e-2;
gradient tensors: 1e-2 of their range element-wise, 3e-3 in relative L2 at the c5 shape (65 568 points; measured <= 1.4e-3) and
8e-3 for the small cases, where one decorrelated point is 1 % of the points of a grasp head (measured 1.6e-4 ... 4.5e-3).  A wrong
index, a missing term or a transposed tile is O(1).
Reference code path: conv_onet/models/decoder.py:117-176, layers.py:39-47, models/__init__.py:111-124 under autograd
(scripts/train_giga.py:198-211)."""
import pytest
import torch

from giga_amd import _capi, networks, synth, weights
from oracle import giga_oracle as O
from tests import dect_ref as R

pytestmark = pytest.mark.gpu

HEADS = weights.HEADS


def _batch(first, B, N, M):
    x = torch.from_numpy(synth.tsdf_batch(first, B))
    pos = torch.from_numpy(synth.query_points(first, B, N, stream=2))
    pos_occ = torch.from_numpy(synth.query_points(first, B, M, stream=3))
    return x, pos, pos_occ


def _reference(sd, planes_nchw, pts, douts):
    """planes_nchw: {'xz','xy','yz'} fp32 (B,32,40,40) leaves; pts / douts per head.  Returns outputs, parameter gradients, plane gradients."""
    planes = {k: v.clone().requires_grad_(True) for k, v in planes_nchw.items()}
    outs, grads = {}, {}
    for head in HEADS:
        if head not in pts:
            continue
        p = pts[head]
        B, N = p.shape[0], p.shape[1]
        c = O.sample_features(p, planes)                                  # (B, N, 96), fp32, differentiable w.r.t. the planes
        raw, saved = R.head_forward(sd, head + ".", c.detach().reshape(B * N, 96), p.reshape(B * N, 3))
        raw = raw.reshape(B, N, -1)
        if raw.shape[-1] == 1:
            raw = raw[..., 0]
        y, dO = R.epilogue_backward(head, raw, douts[head])
        outs[head] = y
        g, dc = R.head_backward(sd, head + ".", saved, dO.reshape(B * N, -1))
        grads.update({head + "." + k: v for k, v in g.items()})
        c.backward(dc.reshape(B, N, 96))
    return outs, grads, {k: v.grad for k, v in planes.items()}


@pytest.mark.parametrize("B,N,M", [(2, 1, 256), (3, 37, 300), (32, 1, 2048)])
def test_bf16_decoder_against_operand_rounded_reference(sd7, monkeypatch, B, N, M):
    dev = torch.device("cuda:0")
    x, pos, pos_occ = _batch(500, B, N, M)
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train().set_train_precision("bf16")
    monkeypatch.setattr(_capi, "ENC_BF16", 0)                # fp32 encoder forward: the planes are the fp32 oracle's
    out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    g5 = torch.Generator().manual_seed(5)
    Rw = [torch.randn(B, N, generator=g5), torch.randn(B, N, 4, generator=g5), torch.randn(B, N, generator=g5),
          torch.randn(B, M, generator=g5) / 16]
    sum((o * r.to(dev)).sum() for o, r in zip(out, Rw)).backward()
    st = net._train_state
    sb = st._pool[0]
    nhwc = sb.nhwc.detach().cpu()                             # (3, B, 40, 40, 32): the planes the decoders sampled
    planes = {k: nhwc[i].permute(0, 3, 1, 2).contiguous() for i, k in enumerate(O.PLANES)}
    pts = {"decoder_qual": pos, "decoder_rot": pos, "decoder_width": pos, "decoder_tsdf": pos_occ}
    douts = dict(zip(HEADS, Rw))
    outs, grads, gplanes = _reference(sd7, planes, pts, douts)
    # forward outputs (post sigmoid / normalize; occupancy logits raw)
    for o, head in zip(out, HEADS):
        ref = outs[head]
        err = (o.detach().cpu() - ref).abs()
        scale = max(1.0, ref.abs().max().item())
        assert err.max().item() <= 2e-2 * scale, (head, err.max().item())
        assert (err > 2e-3 * scale).float().mean().item() <= 0.005, (head, (err > 2e-3 * scale).float().mean().item())
        assert (err > 3e-5 * scale).float().mean().item() <= 0.05, (head, (err > 3e-5 * scale).float().mean().item())
    # every decoder parameter gradient
    worst = (-1.0, "")
    for name, prm in net.named_parameters():
        if not name.startswith("decoder_"):
            continue
        ref = grads[name]
        got = prm.grad.detach().cpu().reshape(ref.shape)
        scale = max(ref.abs().max().item(), 1e-6)
        err = (got - ref).abs().max().item()
        l2 = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
        worst = max(worst, (l2, name))
        assert err <= 1e-2 * scale + 1e-6, (name, err, scale)
        assert l2 <= (3e-3 if B * M >= 65536 else 8e-3), (name, l2)
    print(f"bf16 decoder B={B} N={N} M={M}: worst relative L2 gradient error {worst}")
    # plane gradients: the first region of the backward workspace, NHWC [3][B][40][40][32]
    gp = sb.wsb[:3 * B * 40 * 40 * 32 * 4].view(torch.float32).view(3, B, 40, 40, 32).cpu()
    for i, k in enumerate(O.PLANES):
        ref = gplanes[k].permute(0, 2, 3, 1)
        scale = ref.abs().max().item()
        assert (gp[i] - ref).abs().max().item() <= 1e-2 * scale, (k, (gp[i] - ref).abs().max().item(), scale)
        assert ((gp[i] - ref).norm() / ref.norm()).item() <= (3e-3 if B * M >= 65536 else 8e-3), k


def test_bf16_decoder_images_derived_on_the_device_equal_the_host_packer(sd7):
    """giga_derive_bf16_fragments rebuilds the bf16 head images from the fp32 images after every repack with the function the host
    packer uses (csrc/giga_dect.h): byte-identical, and the images change when the weights do."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ = _batch(510, 2, 1, 64)
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train().set_train_precision("bf16")
    net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
    st = net._train_state
    flat = torch.cat([p.detach().reshape(-1) for p in net._ordered_params()]).cpu()
    host_fwd, host_bwd = _capi.pack_weights(flat, 15), _capi.pack_bwd_weights(flat, 15)
    nf, nb = 4 * 59 * 1024, 4 * 51 * 1024                    # the last regions of the two blobs (giga_layout.h)
    wino = 16 * 48128 * 4                                    # behind the forward blob's decoder images: the Winograd images of the ten 3x3
    fe = -256 - wino                                         # layers (round 6, sum of cin * cout = 48 128), then the 256-byte stamp of a host blob
    assert torch.equal(st.blob.cpu()[fe - nf:fe], host_fwd[fe - nf:fe])
    assert torch.equal(st.bwd_blob.cpu()[fe - nb:fe], host_bwd[fe - nb:fe])          # (the backward blob ends likewise: data-gradient Winograd images, stamp)
    assert host_fwd[fe - nf:fe].any() and host_bwd[fe - nb:fe].any()


def test_bf16_decoder_is_deterministic(sd7):
    """Weight gradients leave the fused kernel as per-workgroup partial tiles summed in a fixed order; the occupancy head's plane
    gradient is gathered in a fixed order: two runs of the decoder parameters' gradients are bit-identical for that head."""
    dev = torch.device("cuda:0")
    x, pos, pos_occ = _batch(520, 8, 1, 512)
    net = networks.get_network("giga"); net.load_state_dict(sd7); net = net.to(dev).train().set_train_precision("bf16")
    runs = []
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        out = net(x.to(dev), pos.to(dev), p_tsdf=pos_occ.to(dev))
        sum(o.sum() for o in out).backward()
        runs.append({n: p.grad.detach().clone() for n, p in net.named_parameters() if n.startswith("decoder_tsdf")})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]), n
