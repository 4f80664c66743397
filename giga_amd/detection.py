"""Inference-side caller of the hot path: counterpart of the network part of the reference's
`vgn.detection_implicit` (/root/reference/src/vgn/detection_implicit.py:17-31, 99-113).

The 40^3 query lattice and `predict()` keep the reference's shapes and dtypes.  The post-processing the
reference runs on the host with scipy (`process` / `bound` / `select`, detection_implicit.py:87-174) is
available on the device as `grasp_select` (C ABI `giga_grasp_select`), and `VGNImplicit` chains network and
post-processing without leaving HBM: only the surviving grasps (a few dozen floats) cross PCIe."""
import ctypes
import time

import numpy as np
import torch

from . import _capi, synth

LOW_TH = 0.5       # detection_implicit.py:15


def query_lattice(resolution=40, device=None):
    """detection_implicit.py:28-31 -> (1, R^3, 3) float32 (bit-identical to the reference's self.pos).

    The returned tensor is registered as "the inference lattice": passing it (the same object) to the
    network selects the lattice fast path of the decoder, which samples each plane once per lattice
    coordinate pair instead of once per point, and it may be shared by a whole batch of scenes."""
    from .convonet import register_lattice
    pos = torch.from_numpy(synth.inference_lattice(resolution))
    if device is not None:
        pos = pos.to(device)
    lin = pos[0, :: resolution * resolution, 0].clone()      # the R coordinates (x varies slowest)
    return register_lattice(pos, lin)


def predict(tsdf_vol, pos, net, device):
    """detection_implicit.py:99-113: tsdf_vol (1,40,40,40) numpy -> qual (64000,), rot (64000,4), width (64000,)."""
    assert tsdf_vol.shape == (1, 40, 40, 40)
    tsdf_vol = torch.from_numpy(np.ascontiguousarray(tsdf_vol, dtype=np.float32)).to(device)
    with torch.no_grad():
        qual_vol, rot_vol, width_vol = net(tsdf_vol, pos)
    return (qual_vol.cpu().squeeze().numpy(), rot_vol.cpu().squeeze().numpy(),
            width_vol.cpu().squeeze().numpy())


def predict_batch(tsdf_batch, pos, net):
    """Batched device-resident variant: tsdf_batch (B,40,40,40) cuda tensor, pos (1|B,N,3) -> device tensors."""
    from .convonet import _lattice_of
    B = tsdf_batch.shape[0]
    if pos.shape[0] == 1 and B > 1 and _lattice_of(pos) is None:
        pos = pos.expand(B, -1, -1).contiguous()          # a registered lattice is shared as-is
    with torch.no_grad():
        return net(tsdf_batch, pos)


def bound_limits(voxel_size, limit=(0.02, 0.02, 0.055)):
    """detection_implicit.py:87-91: the number of boundary voxels zeroed along x, y (both ends) and z (bottom)."""
    return tuple(int(l / voxel_size) for l in limit)


class _GraspBuffers:
    """Device buffers of one `giga_grasp_select` call.  Counters and candidate arrays are views of ONE int32 block, so
    that with a small `cap` everything the host needs comes back in a single D->H copy."""

    def __init__(self, B, R, cap, dev):
        V = R * R * R
        self.B, self.R, self.cap = B, R, cap
        off = self.offsets(B, cap)
        self.pack = torch.empty(off[-1], dtype=torch.int32, device=dev)
        self.counters = self.pack[off[0]:off[0] + 2 * B].view(B, 2)
        self.cand_index = self.pack[off[1]:off[1] + B * cap].view(B, cap)
        self.cand_score = self.pack[off[2]:off[2] + B * cap].view(torch.float32).view(B, cap)
        self.cand_rot = self.pack[off[3]:off[3] + 4 * B * cap].view(torch.float32).view(B, cap, 4)
        self.cand_width = self.pack[off[4]:off[4] + B * cap].view(torch.float32).view(B, cap)
        self.qual_out = torch.empty(B, V, device=dev)
        self.ws = torch.empty(_capi.lib().giga_grasp_workspace_bytes(B, R), dtype=torch.uint8, device=dev)

    @staticmethod
    def offsets(B, cap):
        """int32 offsets of the five sections (+ the total), each rounded up to 4 words: the kernels store quaternions as
        float4, so every section starts 16-byte aligned whatever B is (an odd B used to leave cand_rot 8-byte aligned)."""
        sizes = (2 * B, B * cap, B * cap, 4 * B * cap, B * cap)
        off, at = [], 0
        for n in sizes:
            off.append(at)
            at += (n + 3) // 4 * 4
        return off + [at]

    def host_views(self, pack_h):
        B, cap = self.B, self.cap
        off = self.offsets(B, cap)
        cnt = pack_h[off[0]:off[0] + 2 * B].reshape(B, 2)
        idx = pack_h[off[1]:off[1] + B * cap].reshape(B, cap)
        score = pack_h[off[2]:off[2] + B * cap].view(np.float32).reshape(B, cap)
        rot = pack_h[off[3]:off[3] + 4 * B * cap].view(np.float32).reshape(B, cap, 4)
        width = pack_h[off[4]:off[4] + B * cap].view(np.float32).reshape(B, cap)
        return cnt, idx, score, rot, width


def _grasp_params(R, voxel_size, out_th, threshold, force_detection, max_filter_size, gaussian_filter_sigma, min_width,
                  max_width, limit):
    if voxel_size is None:
        voxel_size = 0.3 / R                                  # detection_implicit.py:41
    lx, ly, lz = bound_limits(voxel_size, limit)
    return _capi.GraspParams(float(gaussian_filter_sigma), min_width, max_width, out_th, LOW_TH, threshold,
                             lx, ly, lz, int(max_filter_size), int(bool(force_detection)))


def _grasp_launch(tsdf, qual, rot, width, prm, buf):
    """Enqueue the four post-processing kernels on the current stream (no synchronisation; hipGraph-capturable)."""
    B, R = buf.B, buf.R
    V = R * R * R
    tsdf = tsdf.reshape(B, V).float().contiguous()
    qual = qual.reshape(B, V).float().contiguous()
    rot = rot.reshape(B, V, 4).float().contiguous()
    width = width.reshape(B, V).float().contiguous()
    with torch.cuda.device(qual.device):
        _capi.check(_capi.lib().giga_grasp_select(
            _capi.ptr(tsdf), _capi.ptr(qual), _capi.ptr(rot), _capi.ptr(width), B, R, ctypes.byref(prm),
            _capi.ptr(buf.qual_out), _capi.ptr(buf.counters), buf.cap, _capi.ptr(buf.cand_index), _capi.ptr(buf.cand_score),
            _capi.ptr(buf.cand_rot), _capi.ptr(buf.cand_width), _capi.ptr(buf.ws), buf.ws.numel(),
            _capi.stream_ptr(qual.device)), "giga_grasp_select")


def _grasp_collect(buf, force_detection):
    """D->H of the survivors and the host-side ordering.  Returns None if a scene produced more candidates than
    `cap` (the caller then repeats with the full capacity)."""
    B, R, cap = buf.B, buf.R, buf.cap
    if buf.pack.numel() * 4 <= (1 << 20):                     # small block: one copy brings everything
        cnt, idx_h, score_h, rot_h, width_h = buf.host_views(buf.pack.cpu().numpy())
        if B and cnt[:, 1].max() > cap:
            return None
    else:
        cnt = buf.counters.cpu().numpy()                      # synchronises; 8 bytes per scene
        if B and cnt[:, 1].max() > cap:
            return None
        kmax = int(cnt[:, 1].max()) if B else 0
        idx_h = buf.cand_index[:, :kmax].cpu().numpy()
        score_h = buf.cand_score[:, :kmax].cpu().numpy()
        rot_h = buf.cand_rot[:, :kmax].cpu().numpy()
        width_h = buf.cand_width[:, :kmax].cpu().numpy()
    out = []
    for b in range(B):
        k = int(cnt[b, 1])
        best_only = bool(force_detection) and cnt[b, 0] == 0
        flat = idx_h[b, :k]
        # the reference sorts with reversed(np.argsort(scores)) over the argwhere (ascending index) order
        first = np.argsort(flat, kind="stable")
        order = first[np.asarray(list(reversed(np.argsort(score_h[b, :k][first]))), dtype=np.int64)]
        if best_only:
            order = order[:1]
        flat = flat[order].astype(np.int64)
        out.append({"index": np.stack((flat // (R * R), (flat // R) % R, flat % R), -1).reshape(-1, 3),
                    "score": score_h[b, :k][order], "rot": rot_h[b, :k][order], "width": width_h[b, :k][order],
                    "best_only": best_only})
    return out


def grasp_select(tsdf, qual, rot, width, voxel_size=None, out_th=0.5, threshold=0.9, force_detection=False,
                 max_filter_size=4, gaussian_filter_sigma=1.0, min_width=0.033, max_width=0.233,
                 limit=(0.02, 0.02, 0.055), return_volume=False):
    """process + bound + select (detection_implicit.py:115-143, 87-97, 146-174) for B scenes on the device.

    tsdf (B,R,R,R) | (B,1,R,R,R), qual (B,R^3), rot (B,R^3,4), width (B,R^3): float32 device tensors (the
    network outputs of `predict_batch`).  Returns a list of B dicts with numpy arrays sorted by descending
    score: index (K,3) voxel indices, score (K,), rot (K,4) quaternions, width (K,), best_only flag; plus the
    processed quality volume (B,R,R,R) (device tensor) when return_volume is set."""
    _capi.require_device(tsdf, qual, rot, width)
    B, R = qual.shape[0], tsdf.shape[-1]
    prm = _grasp_params(R, voxel_size, out_th, threshold, force_detection, max_filter_size, gaussian_filter_sigma,
                        min_width, max_width, limit)
    out = None
    for cap in (1024, R * R * R):                             # nearly always a few dozen survivors; plateaus need all
        buf = _GraspBuffers(B, R, cap, qual.device)
        _grasp_launch(tsdf, qual, rot, width, prm, buf)
        out = _grasp_collect(buf, force_detection)
        if out is not None:
            break
    if return_volume:
        return out, buf.qual_out.view(B, R, R, R)
    return out


class VGNImplicit:
    """Device-resident counterpart of the reference planner (detection_implicit.py:17-85): network on the 40^3
    lattice + grasp post-processing, one D->H copy of the selected grasps.  `__call__(state)` keeps the
    reference's contract (state.tsdf a (1,R,R,R) numpy grid or an object with get_grid()/voxel_size/size) and
    returns (grasps, scores, toc) where each grasp is a dict {rotation (xyzw quat), translation (m), width (m)}
    -- the fields of vgn.grasp.Grasp/Transform, which live outside this package.  `plan_batch` is the batched,
    tensor-in form."""

    def __init__(self, model_path, model_type, best=False, force_detection=False, qual_th=0.9, out_th=0.5,
                 visualize=False, resolution=40, net=None, seed=None, use_graph=False, **kwargs):
        from .networks import load_network
        self.device = torch.device("cuda")
        self.net = net if net is not None else load_network(model_path, self.device, model_type=model_type)
        self.qual_th, self.best, self.force_detection, self.out_th = qual_th, best, force_detection, out_th
        self.visualize = visualize
        self.resolution = resolution
        self.pos = query_lattice(resolution, self.device)
        self._lin_host = self.pos[0, :: resolution * resolution, 0].cpu().numpy()
        self._rng = np.random.default_rng(seed)
        self.use_graph = bool(use_graph)
        self._graphs = {}
        self._graph_weights = None

    def plan_batch(self, tsdf, tsdf_process=None, voxel_size=None):
        """tsdf (B,R,R,R) device tensor -> per-scene candidate dicts (see grasp_select), lattice positions added."""
        sel = None
        if self.use_graph and tsdf.shape[0] <= 4:              # launch-bound regime only: at larger batches the eager
            sel = self._graphed_plan(tsdf, tsdf_process, voxel_size)   # stream is GPU-bound and graph nodes do not pipeline
        if sel is None:
            qual, rot, width = predict_batch(tsdf, self.pos, self.net)
            sel = grasp_select(tsdf if tsdf_process is None else tsdf_process, qual, rot, width, voxel_size=voxel_size,
                               out_th=self.out_th, threshold=self.qual_th, force_detection=self.force_detection,
                               max_filter_size=8 if self.visualize else 4)
        lin = self._lin_host
        for s in sel:
            s["position"] = lin[s["index"]]                    # center_vol[i, j, k], detection_implicit.py:181
        return sel

    def _graphed_plan(self, tsdf, tsdf_process, voxel_size):
        """Network (~17 launches) AND post-processing (memset + 4 kernels) replayed as ONE hipGraph launch per batch
        size, then one small D->H copy: a single-scene plan is launch-bound, not compute-bound.  Inputs live in
        static buffers owned by the graph.  Returns None (eager path) if a scene has more survivors than the graph's
        candidate capacity."""
        B, R = tsdf.shape[0], self.resolution
        # a graph bakes in the packed-weight buffer and the precision: new weights / precision -> new capture
        blob = self.net.packed_blob(self.device)
        wkey = (self.net.precision, self.net._packed.key)      # parameter storage + versions, not the blob address
        if self._graph_weights is None or self._graph_weights[0] != wkey:
            self._graphs.clear()
            self._graph_weights = (wkey, blob)                 # keeps the captured buffer alive with its graphs
        key = (B, None if voxel_size is None else float(voxel_size))
        ent = self._graphs.get(key)
        if ent is None:
            static_in = torch.empty_like(tsdf, dtype=torch.float32).contiguous()
            static_proc = torch.empty_like(static_in)
            static_in.copy_(tsdf)
            static_proc.copy_(tsdf if tsdf_process is None else tsdf_process)
            prm = _grasp_params(R, voxel_size, self.out_th, self.qual_th, self.force_detection,
                                8 if self.visualize else 4, 1.0, 0.033, 0.233, (0.02, 0.02, 0.055))
            buf = _GraspBuffers(B, R, 1024, self.device)

            def run():
                qual, rot, width = predict_batch(static_in, self.pos, self.net)
                _grasp_launch(static_proc, qual, rot, width, prm, buf)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):                      # warm-up outside capture (workspaces, attributes)
                for _ in range(2):
                    run()
            torch.cuda.current_stream(self.device).wait_stream(side)
            from . import convonet
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                run()
            # the graph bakes in the encoder / lattice workspaces it used (the scratch caches are keyed by stream, so the
            # capture got its own entries); the caches evict in LRU order, so the graph keeps its own references
            keep = (self.net.encoder._ws.snapshot(), convonet._LATTICE_WS.snapshot())
            ent = self._graphs[key] = (graph, static_in, static_proc, buf, (prm, keep))
        graph, static_in, static_proc, buf, _ = ent
        static_in.copy_(tsdf)
        static_proc.copy_(tsdf if tsdf_process is None else tsdf_process)
        graph.replay()
        return _grasp_collect(buf, self.force_detection)

    def __call__(self, state, scene_mesh=None, aff_kwargs={}):
        tsdf_process = state.tsdf_process if hasattr(state, "tsdf_process") else state.tsdf
        if isinstance(state.tsdf, np.ndarray):
            tsdf_vol, voxel_size, size = state.tsdf, 0.3 / self.resolution, 0.3
        else:
            tsdf_vol, voxel_size, size = state.tsdf.get_grid(), tsdf_process.voxel_size, state.tsdf.size
            tsdf_process = tsdf_process.get_grid()
        R = self.resolution
        tic = time.time()
        t = torch.from_numpy(np.ascontiguousarray(tsdf_vol, np.float32).reshape(1, R, R, R)).to(self.device)
        tp = t if tsdf_process is tsdf_vol else torch.from_numpy(
            np.ascontiguousarray(tsdf_process, np.float32).reshape(1, R, R, R)).to(self.device)
        sel = self.plan_batch(t, tp, voxel_size)[0]
        toc = time.time() - tic
        k = len(sel["score"])
        p = np.arange(k) if self.best else self._rng.permutation(k)       # detection_implicit.py:65-68
        grasps = [{"rotation": sel["rot"][i], "translation": (sel["position"][i] + 0.5) * size,
                   "width": float(sel["width"][i]) * size} for i in p]
        return grasps, sel["score"][p], toc
