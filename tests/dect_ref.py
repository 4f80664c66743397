"""Operand-rounded reference of the bf16 training decoder (csrc/giga_decoder_train16.hip), plain torch on the CPU (test helper).

The kernel evaluates LocalDecoder.forward (reference conv_onet/models/decoder.py:133-176, layers.py:39-47) and its backward with
bf16 MFMA operands and fp32 accumulation.  This file restates that arithmetic with the rounding made explicit -- every operand of
every product is rounded to bf16 exactly where the kernel rounds it, sums are fp32/fp64 -- so that the GPU result can be held
to accumulation-order tolerances instead of a bf16-sized envelope around the fp32 oracle:
    operands rounded once: sampled features c, relu(net), relu(h), every weight matrix, the gradients DN[b], DH[b], dO;
    exact (hi + lo bf16 pairs): the query coordinates p, fc_p.weight, the folded biases; fp32: fc_0 / fc_out biases, the residual
    stream, the ReLU masks (taken from the rounded activations: bf16(relu(v)) != 0).
"""
import torch

NBLK = 5


def bf(t):
    return t.to(torch.bfloat16).to(t.dtype)


def hl(t):
    """hi + lo bf16 pair of an fp32 tensor (what the aux fragment carries)."""
    h = bf(t.float())
    return (h.double() + bf(t.float() - h).double()).to(t.dtype)


def head_forward(sd, prefix, c, p, acc=torch.float64):
    """c (P, 96) fp32 sampled features, p (P, 3).  Returns (out (P, out_dim) raw, saved activations)."""
    W = lambda k: sd[prefix + k + ".weight"].float()  # noqa: E731
    B = lambda k: sd[prefix + k + ".bias"].float()  # noqa: E731
    cb = bf(c.float()).to(acc)
    p32 = p.float()
    p_hi = bf(p32)
    p_lo = bf(p32 - p_hi)
    wp = W("fc_p")
    wh = bf(wp)
    wl = bf(wp - wh)
    net = cb @ bf(W("fc_c.0")).to(acc).T + p_hi.to(acc) @ wh.to(acc).T + p_lo.to(acc) @ wh.to(acc).T + p_hi.to(acc) @ wl.to(acc).T \
        + hl(B("fc_p") + B("fc_c.0")).to(acc)
    xn, xh = [], []
    for b in range(NBLK):
        net32 = net.float()
        xn.append(bf(torch.relu(net32)))
        hh = xn[b].to(acc) @ bf(W(f"blocks.{b}.fc_0")).to(acc).T + B(f"blocks.{b}.fc_0").to(acc)
        if b + 1 < NBLK:
            net = net + cb @ bf(W(f"fc_c.{b + 1}")).to(acc).T + hl(B(f"fc_c.{b + 1}") + B(f"blocks.{b}.fc_1")).to(acc)
        else:
            net = net + hl(B(f"blocks.{b}.fc_1")).to(acc)
        xh.append(bf(torch.relu(hh.float())))
        net = net + xh[b].to(acc) @ bf(W(f"blocks.{b}.fc_1")).to(acc).T
    xo = bf(torch.relu(net.float()))
    out = xo.to(acc) @ bf(W("fc_out")).to(acc).T + B("fc_out").to(acc)
    saved = dict(cb=cb, p_hi=p_hi, p_lo=p_lo, xn=xn, xh=xh, xo=xo)
    return out.float(), saved


def head_backward(sd, prefix, saved, dO, acc=torch.float64):
    """dO (P, out_dim): gradient w.r.t. the RAW head output.  Returns ({param name: grad}, dc (P, 96))."""
    W = lambda k: sd[prefix + k + ".weight"].float()  # noqa: E731
    cb, xn, xh, xo = saved["cb"], saved["xn"], saved["xh"], saved["xo"]
    g = {}
    dOb = bf(dO.float()).to(acc)
    g["fc_out.weight"] = dOb.T @ xo.to(acc)
    g["fc_out.bias"] = dOb.sum(0)
    G = (dO.float().to(acc) @ bf(W("fc_out")).to(acc)) * (xo != 0)
    G = G.float()                                           # the kernel holds DN in fp32 registers
    Gb = bf(G)
    g[f"blocks.{NBLK - 1}.fc_1.bias"] = Gb.to(acc).sum(0)
    dc = torch.zeros_like(cb)
    for b in range(NBLK - 1, -1, -1):
        g[f"blocks.{b}.fc_1.weight"] = Gb.to(acc).T @ xh[b].to(acc)
        dh = (Gb.to(acc) @ bf(W(f"blocks.{b}.fc_1")).to(acc)).float() * (xh[b] != 0)
        Hb = bf(dh)
        g[f"blocks.{b}.fc_0.weight"] = Hb.to(acc).T @ xn[b].to(acc)
        g[f"blocks.{b}.fc_0.bias"] = Hb.to(acc).sum(0)
        dn = (Hb.to(acc) @ bf(W(f"blocks.{b}.fc_0")).to(acc)).float() * (xn[b] != 0)
        G = G + dn
        Gb = bf(G)
        g[f"fc_c.{b}.weight"] = Gb.to(acc).T @ cb
        g[f"fc_c.{b}.bias"] = Gb.to(acc).sum(0)
        if b > 0:
            g[f"blocks.{b - 1}.fc_1.bias"] = Gb.to(acc).sum(0)
        dc = dc + Gb.to(acc) @ bf(W(f"fc_c.{b}")).to(acc)
    g["fc_p.weight"] = Gb.to(acc).T @ saved["p_hi"].to(acc) + Gb.to(acc).T @ saved["p_lo"].to(acc)
    g["fc_p.bias"] = Gb.to(acc).sum(0)
    return {k: v.float() for k, v in g.items()}, dc.float()


def epilogue_backward(head, raw, dout):
    """Gradient w.r.t. the raw head output from the gradient w.r.t. the post-epilogue output (models/__init__.py:111-124):
    decoder_qual -> sigmoid, decoder_rot -> F.normalize(dim=-1), others identity (the occupancy logits are returned raw)."""
    raw = raw.detach().clone().requires_grad_(True)
    if head == "decoder_qual":
        y = torch.sigmoid(raw)
    elif head == "decoder_rot":
        y = torch.nn.functional.normalize(raw, dim=-1)
    else:
        y = raw
    y.backward(dout)
    return y.detach(), raw.grad
