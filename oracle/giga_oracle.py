"""CPU oracle: a from-the-spec restatement of GIGA's dense forward path in plain torch fp32.

TEST INFRASTRUCTURE ONLY.  Nothing under `giga_amd/` imports this file; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may (as the checker / the timed
CPU baseline, never as the thing shipped).  The product path raises if its HIP library is absent.

Pinning: this oracle is checked against outputs of the *reference itself*, imported in the build
container through `oracle/ref_bootstrap.py` (tests/test_oracle_vs_reference.py, live) and against
the committed fixtures `tests/golden/*.npz` that `oracle/make_goldens.py` captured from the
reference (these travel to the GPU box; the reference does not).  The reference has no tests or
golden vectors of its own for this path.

Third-party arithmetic on the path: `torch_scatter.scatter_mean` (torch-scatter==2.0.6,
reference environment.yaml:145; call site encoder/voxels.py:65) is absent from /root/reference.
Its published semantics (sum of sources per index / max(count,1)) combined with the fixed
voxel-centre coordinates of voxels.py:95-103 make every plane cell the arithmetic mean of exactly
the 40 voxels along the projected axis; that closed form is what is restated here.

Every function cites the reference lines (relative to /root/reference/src/vgn) it follows.
All functions are pure: weights come in as a state-dict with the reference's key names.
"""
import torch
import torch.nn.functional as F

PLANES = ("xz", "xy", "yz")            # order of encoder kwargs 'plane_type' (networks.py:95)
_PLANE_AXES = {"xz": (0, 2), "xy": (0, 1), "yz": (1, 2)}   # ConvONets/common.py:246-251
GRASP_HEADS = ("decoder_qual", "decoder_rot", "decoder_width")


# --------------------------------------------------------------------------------------------
# coordinates  (ConvONets/common.py:238-261)
# --------------------------------------------------------------------------------------------
def normalize_coordinate(p, plane, padding=0.0):
    """common.py:238-261.  p (B,N,3) -> (B,N,2) in [0, 1-10e-6].

    `10e-6` is the reference's literal (= 1e-5).  All arithmetic in fp32 like the reference:
    the divisor is the python double 1+padding+10e-6 applied to an fp32 tensor.
    """
    a0, a1 = _PLANE_AXES[plane]
    xy = torch.stack((p[..., a0], p[..., a1]), dim=-1)
    xy = xy / (1 + padding + 10e-6) + 0.5
    xy = torch.where(xy >= 1, torch.full_like(xy, 1 - 10e-6), xy)
    xy = torch.where(xy < 0, torch.zeros_like(xy), xy)
    return xy


# --------------------------------------------------------------------------------------------
# encoder  (ConvONets/encoder/voxels.py:89-121, encoder/unet.py)
# --------------------------------------------------------------------------------------------
def conv_in_relu(sd, x):
    """voxels.py:36,106-107: relu(Conv3d(1,32,3,padding=1)(x[:,None])) -> (B,32,D,D,D)."""
    return F.relu(F.conv3d(x[:, None], sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"],
                           padding=1))


def project_planes(feat):
    """voxels.py:57-67 + common.py:238-261,303-318 + scatter_mean, in closed form.

    feat (B,C,X,Y,Z).  Plane tensors are (B,C,H,W) with W = first listed axis, H = second
    (coordinate2index: index = x0 + reso*x1, then reshape(reso, reso)).
      xz: plane[b,c,iz,ix] = mean_y feat[b,c,ix,iy,iz]
      xy: plane[b,c,iy,ix] = mean_z feat
      yz: plane[b,c,iz,iy] = mean_x feat
    """
    return {
        "xz": feat.mean(dim=3).transpose(2, 3).contiguous(),
        "xy": feat.mean(dim=4).transpose(2, 3).contiguous(),
        "yz": feat.mean(dim=2).transpose(2, 3).contiguous(),
    }


def unet_forward(sd, x, prefix="encoder.unet.", rnd=None):
    """unet.py:225-239 for UNet(32, in_channels=32, depth=3, start_filts=32, merge='concat').
    rnd (test aid, not in the reference): a rounding applied to the INPUT and WEIGHT of every convolution (bias and
    accumulation stay fp32), e.g. lambda t: t.bfloat16().float() to emulate bf16 MFMA operands."""
    if rnd is not None:
        st = unet_stages(sd, x, prefix, rnd)
        return st["OUT"]

    def c3(name, t):
        return F.relu(F.conv2d(t, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"],
                               padding=1))

    skips = []
    for i in range(3):                                   # DownConv.forward unet.py:66-72
        x = c3(f"down_convs.{i}.conv1", x)
        x = c3(f"down_convs.{i}.conv2", x)
        skips.append(x)
        if i < 2:
            x = F.max_pool2d(x, 2, 2)
    for i in range(2):                                   # UpConv.forward unet.py:101-114
        up = F.conv_transpose2d(x, sd[prefix + f"up_convs.{i}.upconv.weight"],
                                sd[prefix + f"up_convs.{i}.upconv.bias"], stride=2)
        x = torch.cat((up, skips[-(i + 2)]), dim=1)      # (from_up, from_down) unet.py:109
        x = c3(f"up_convs.{i}.conv1", x)
        x = c3(f"up_convs.{i}.conv2", x)
    return F.conv2d(x, sd[prefix + "conv_final.weight"], sd[prefix + "conv_final.bias"])


def unet_stages(sd, x, prefix="encoder.unet.", rnd=None):
    """Same arithmetic as unet_forward, returning every intermediate activation (NCHW) under the
    workspace names of giga_encoder.hip (A0 S0 Q0 A1 S1 Q1 A2 S2 U0 A3 A4 U1 A5 A6 OUT).  rnd: see unet_forward."""
    q = rnd if rnd is not None else (lambda t: t)

    def c3(name, t):
        return F.relu(F.conv2d(q(t), q(sd[prefix + name + ".weight"]), sd[prefix + name + ".bias"], padding=1))

    def up(name, t):
        return F.conv_transpose2d(q(t), q(sd[prefix + name + ".weight"]), sd[prefix + name + ".bias"], stride=2)

    st = {}
    st["A0"] = c3("down_convs.0.conv1", x); st["S0"] = c3("down_convs.0.conv2", st["A0"])
    st["Q0"] = F.max_pool2d(st["S0"], 2, 2)
    st["A1"] = c3("down_convs.1.conv1", st["Q0"]); st["S1"] = c3("down_convs.1.conv2", st["A1"])
    st["Q1"] = F.max_pool2d(st["S1"], 2, 2)
    st["A2"] = c3("down_convs.2.conv1", st["Q1"]); st["S2"] = c3("down_convs.2.conv2", st["A2"])
    st["U0"] = up("up_convs.0.upconv", st["S2"])
    st["A3"] = c3("up_convs.0.conv1", torch.cat((st["U0"], st["S1"]), 1)); st["A4"] = c3("up_convs.0.conv2", st["A3"])
    st["U1"] = up("up_convs.1.upconv", st["A4"])
    st["A5"] = c3("up_convs.1.conv1", torch.cat((st["U1"], st["S0"]), 1)); st["A6"] = c3("up_convs.1.conv2", st["A5"])
    st["OUT"] = F.conv2d(q(st["A6"]), q(sd[prefix + "conv_final.weight"]), sd[prefix + "conv_final.bias"])
    return st


def encoder_forward(sd, x, rnd=None):
    """LocalVoxelEncoder.forward voxels.py:89-121.  x (B,40,40,40) -> {'xz','xy','yz'}: (B,32,40,40).  rnd: see unet_forward."""
    planes = project_planes(conv_in_relu(sd, x))
    return {k: unet_forward(sd, planes[k], rnd=rnd) for k in PLANES}


# --------------------------------------------------------------------------------------------
# decoder  (ConvONets/conv_onet/models/decoder.py:117-176, layers.py:39-47)
# --------------------------------------------------------------------------------------------
def sample_plane_feature(p, c, plane, padding=0.0):
    """decoder.py:117-122: bilinear grid_sample, border padding, align_corners=True."""
    xy = normalize_coordinate(p, plane, padding)
    vgrid = 2.0 * xy[:, :, None] - 1.0
    return F.grid_sample(c, vgrid, padding_mode="border", align_corners=True,
                         mode="bilinear").squeeze(-1)


def sample_features(p, planes, padding=0.0):
    """decoder.py:136-147 (concat_feat=True): (B,N,96) in order xz,xy,yz."""
    c = torch.cat([sample_plane_feature(p, planes[k], k, padding) for k in PLANES], dim=1)
    return c.transpose(1, 2)


def decoder_mlp(sd, head, p, c):
    """decoder.py:160-176 + ResnetBlockFC.forward layers.py:39-47.  Raw head output."""
    def lin(name, t):
        return F.linear(t, sd[f"{head}.{name}.weight"], sd[f"{head}.{name}.bias"])

    net = lin("fc_p", p.float())
    for i in range(5):
        net = net + lin(f"fc_c.{i}", c)
        h = lin(f"blocks.{i}.fc_0", F.relu(net))
        net = net + lin(f"blocks.{i}.fc_1", F.relu(h))
    return lin("fc_out", F.relu(net)).squeeze(-1)


def decoder_forward(sd, head, p, planes):
    """LocalDecoder.forward decoder.py:133-176: (B,N) if out_dim==1 else (B,N,out_dim)."""
    return decoder_mlp(sd, head, p, sample_features(p, planes))


# --------------------------------------------------------------------------------------------
# model  (ConvONets/conv_onet/models/__init__.py:42-124)
# --------------------------------------------------------------------------------------------
def decode(sd, p, planes):
    """models/__init__.py:111-124: sigmoid(qual), L2-normalised rot (eps 1e-12), raw width."""
    qual = torch.sigmoid(decoder_forward(sd, "decoder_qual", p, planes))
    rot = F.normalize(decoder_forward(sd, "decoder_rot", p, planes), dim=2)
    width = decoder_forward(sd, "decoder_width", p, planes)
    return qual, rot, width


def model_forward(sd, x, p, p_tsdf=None, detach_tsdf=False):
    """ConvolutionalOccupancyNetwork.forward models/__init__.py:42-67 (tsdf = raw logits; detach_tsdf :61-63)."""
    planes = encoder_forward(sd, x)
    out = decode(sd, p, planes)
    if p_tsdf is not None:
        if detach_tsdf:
            planes = {k: v.detach() for k, v in planes.items()}
        out = out + (decoder_forward(sd, "decoder_tsdf", p_tsdf, planes),)
    return out


def infer_geo(sd, x, p_tsdf):
    """models/__init__.py:69-72."""
    return decoder_forward(sd, "decoder_tsdf", p_tsdf, encoder_forward(sd, x))


# --------------------------------------------------------------------------------------------
# callers  (detection_implicit.py:28-31, scripts/train_giga.py:154-195)
# --------------------------------------------------------------------------------------------
def inference_lattice(resolution=40):
    """detection_implicit.py:28-31: (1, R^3, 3) lattice linspace(-0.5, 0.5-1/R, R)^3, 'ij'."""
    lin = torch.linspace(-0.5, 0.5 - 1.0 / resolution, resolution)
    x, y, z = torch.meshgrid(lin, lin, lin, indexing="ij")
    return torch.stack((x, y, z), dim=-1).float().reshape(1, resolution ** 3, 3)


def train_select(out):
    """scripts/train_giga.py:154-158."""
    qual, rot, width, occ = out
    return qual.squeeze(-1), rot.squeeze(1), width.squeeze(-1), torch.sigmoid(occ)


def train_loss(y_pred, y):
    """scripts/train_giga.py:161-195.  y = (label, rotations (B,2,4), width, occ (B,M))."""
    label_pred, rot_pred, width_pred, occ_pred = y_pred
    label, rotations, width, occ = y
    loss_qual = F.binary_cross_entropy(label_pred, label, reduction="none")

    def quat(pred, target):
        return 1.0 - torch.abs(torch.sum(pred * target, dim=1))

    loss_rot = torch.min(quat(rot_pred, rotations[:, 0]), quat(rot_pred, rotations[:, 1]))
    loss_width = F.mse_loss(40 * width_pred, 40 * width, reduction="none")
    loss_occ = F.binary_cross_entropy(occ_pred, occ, reduction="none").mean(-1)
    loss = loss_qual + label * (loss_rot + 0.01 * loss_width) + loss_occ
    return loss.mean(), {"loss_qual": loss_qual.mean(), "loss_rot": loss_rot.mean(),
                         "loss_width": loss_width.mean(), "loss_occ": loss_occ.mean(),
                         "loss_all": loss.mean()}
