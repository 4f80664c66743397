#!/usr/bin/env python
"""Benchmark of the GIGA dense inference path on MI355X (driver contract in the task brief).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one forward pass of BASELINE.json configs[1] on every rank:
    32 synthetic 40^3 TSDF scenes per GPU, the literal `train_giga` call shape
    (1 grasp query point for the three grasp heads + 2048 occupancy queries per scene),
    encoder + decoders in exact fp32 (fp32 MFMA), inputs resident in HBM, outputs left in HBM.
Scenes are independent, so ranks are pure replicas on different scenes (weak scaling); the only
collectives are the MAX-reduce of the elapsed times and one all_gather of per-rank counters at the
end (RCCL over xGMI; the group is created after the timed region, whose two barriers run over gloo).

One JSON line is printed by rank 0; besides the contract keys it carries
  roofline      - the dominant kernel of the timed workload, timed live with HIP events on the
                  launch stream inside the timed steps
  cpu_baseline  - the CPU oracle (a port of the reference's PyTorch path) on the host cores
  extra         - config c4 (64 000 grasp queries/scene, f16-MFMA fused decoder) numbers
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MATRIX_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32-input MFMA == fp32 vector peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # dense f16/bf16 MFMA
# Sustained peaks measured on the round-1 box by tools/mfma_peak.hip (profiles/r01j_hw_peaks.txt): back-to-back
# register-resident MFMAs on all 1024 SIMDs.  Reported beside the spec fraction; `frac` stays spec-based.
MEASURED_F32_MATRIX_TFLOPS = 156.3    # v_mfma_f32_32x32x2_f32 (99.4 % of spec)
MEASURED_F16_MFMA_TFLOPS = 2350.6     # v_mfma_f32_32x32x16_f16 (94 % of spec: the matrix-core clock sags to 2.24 GHz)
FLOP_ENCODER = 1_132_953_600          # SURVEY.md 8d / BASELINE.md section 4
FLOP_HEAD = {"qual": 51_456, "rot": 51_648, "width": 51_456, "tsdf": 51_456}
FLOP_GRASP3 = 154_560

# U-Net layer table (kind, cin, cout, H, W): algorithmic FLOPs per image = 2*H*W*taps*cin*cout
_CONV = [(9, 32, 32, 40), (9, 32, 32, 40), (9, 32, 64, 20), (9, 64, 64, 20), (9, 64, 128, 10), (9, 128, 128, 10),
         (4, 128, 64, 10), (9, 128, 64, 20), (9, 64, 64, 20), (4, 64, 32, 20), (9, 64, 32, 40), (9, 32, 32, 40),
         (1, 32, 32, 40)]
STAGE_NAMES = ["convin_project", "plane_finalize"] + [
    "unet.down0.conv1", "unet.down0.conv2+pool", "unet.down1.conv1", "unet.down1.conv2+pool", "unet.down2.conv1",
    "unet.down2.conv2", "unet.up0.upconv", "unet.up0.conv1", "unet.up0.conv2", "unet.up1.upconv",
    "unet.up1.conv1", "unet.up1.conv2", "unet.conv_final"]


def stage_flops(stage, B):
    """Algorithmic FLOPs of one launch of encoder stage `stage` for B scenes."""
    if stage == 0:
        return 110_592_000 * B
    if stage == 1:
        return 0
    taps, cin, cout, hw = _CONV[stage - 2]
    return 2 * hw * hw * taps * cin * cout * 3 * B


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="scenes per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--dp-train", action="store_true",
                    help="with --gpus N > 1: also time the data-parallel training step of BASELINE c5 (one RCCL "
                         "all-reduce of the flat gradient bucket per step) and report it under extra")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); no CPU fallback for the timed path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
        # launched by torch.distributed.run: one rank per GPU over RCCL (also at world size 1, so that the collective
        # code path of the multi-GPU runs can be exercised on a single-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")    # one node by contract; the hostname may not resolve
        # Rendezvous and the barriers that bracket the timed region run over gloo (host side, TCP on 127.0.0.1); the
        # RCCL communicator for the collectives proper (MAX of the times, all_gather of the counters, the data-parallel
        # gradient all-reduce) is created AFTER the timed region: the hot path has no collective, and a live RCCL
        # communicator (its streams / hardware queues) was measured to slow the single-GPU step by 2-5 %
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from giga_amd import _capi, networks, synth, weights
    from giga_amd.convonet import decode_heads

    B, M = args.batch, 2048
    sd = weights.make_state_dict(7)
    net = networks.get_network("giga")
    net.load_state_dict(sd)
    net = net.to(dev).eval().set_precision("fp32")
    first = rank * B                                     # each rank owns its own scenes
    x = torch.from_numpy(synth.tsdf_batch(first, B)).to(dev)
    pos = torch.from_numpy(synth.query_points(first, B, 1, stream=2)).to(dev)
    pos_occ = torch.from_numpy(synth.query_points(first, B, M, stream=3)).to(dev)
    blob = net.packed_blob(dev)
    L = _capi.lib()

    def barrier():
        # drain the device first, so that the barrier collective is enqueued on an idle device (dist.barrier() +
        # synchronize then cost 27 us); the order made no measurable difference at world size 1
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # -------- find the dominant kernel of the workload (untimed pre-pass, every stage probed once) ----
    ev_a, ev_b = L.giga_event_create(), L.giga_event_create()
    dec_ev = (L.giga_event_create(), L.giga_event_create())

    def step(probe=None, dec_probe=None):
        """One forward of the workload through the module API (vgn-compatible call)."""
        with torch.no_grad():
            if dec_probe is None:
                return net(x, pos, p_tsdf=pos_occ, _probe=probe)
            nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision="fp32")
            g = decode_heads(nhwc, pos, blob, 7, "fp32", True)
            t = decode_heads(nhwc, pos_occ, blob, 8, "fp32", False, probe=dec_probe)
        return g, t

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    stage_ms = []
    ms = ctypes.c_float()
    for st in range(15):
        reps = []
        for _ in range(5):                 # median of five: two layers are within a few percent of each other
            step(probe=(st, ev_a, ev_b))
            _capi.check(L.giga_event_elapsed_ms(ev_a, ev_b, ctypes.byref(ms)), "event")
            reps.append(ms.value)
        stage_ms.append(float(np.median(reps)))
    step(dec_probe=dec_ev)
    _capi.check(L.giga_event_elapsed_ms(dec_ev[0], dec_ev[1], ctypes.byref(ms)), "event")
    dec_ms_once = ms.value
    dom = int(np.argmax(stage_ms))

    # -------- timed region ----------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    K = args.steps
    evs = [(L.giga_event_create(), L.giga_event_create()) for _ in range(K)]
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(probe=(dom, evs[i][0], evs[i][1]))
    barrier()
    elapsed = time.perf_counter() - t0
    dom_ms = []
    for a, b in evs:
        _capi.check(L.giga_event_elapsed_ms(a, b, ctypes.byref(ms)), "event")
        dom_ms.append(ms.value)
        L.giga_event_destroy(a); L.giga_event_destroy(b)
    t_elapsed = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rccl = None
    if dist is not None:
        rccl = dist.new_group(backend="nccl")                # one rank per GPU over RCCL / xGMI
        dist.all_reduce(t_elapsed, op=dist.ReduceOp.MAX, group=rccl)
        # the north-star's single data-path-free collective: gather per-rank counters
        counters = torch.tensor([float(B * K), elapsed], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(counters) for _ in range(world)]
        dist.all_gather(gathered, counters, group=rccl)
        total_scenes = sum(float(c[0]) for c in gathered)
    else:
        total_scenes = float(B * K)
    T = float(t_elapsed.item())
    scenes_per_s = total_scenes / T

    # -------- opt-in extra at N > 1 (--dp-train): the data-parallel training step of BASELINE c5 (every rank takes part;
    # off by default so that the scaling run consists of the data-path-free forward only) ----------
    dp_train = None
    if dist is not None and world > 1 and args.dp_train:
        try:
            net.enable_data_parallel(group=rccl)
            dp_train = bench_train(net, dev, synth, B, M, rank=rank, world=world)
        except Exception as e:  # noqa: BLE001
            dp_train = {"error": f"{type(e).__name__}: {e}"}
        finally:
            net.enable_data_parallel(enabled=False)
            net.eval().set_precision("fp32")

    def finish():                          # every rank leaves together: no rank tears the communicator down early
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if rank != 0:
        finish()
        return

    # HBM bytes per launch of the dominant kernel: PMC counters cannot be collected from inside this process,
    # so they come from the committed rocprofv3 PMC table of the same workload (profiles/r01_traffic_c2.json,
    # produced by tools/gpu_traffic.sh); null when the table has no entry for this kernel / batch size.
    traffic = None
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic_c2.json")))
        if B == 32 and STAGE_NAMES[dom] in tab["stages"]:
            traffic = tab["stages"][STAGE_NAMES[dom]]["bytes"]
    except (OSError, ValueError, KeyError):
        pass
    dom_avg_ms = float(np.mean(dom_ms))
    dom_flops = stage_flops(dom, B)
    achieved = dom_flops / (dom_avg_ms * 1e-3) / 1e12
    flop_scene = FLOP_ENCODER + FLOP_GRASP3 * 1 + FLOP_HEAD["tsdf"] * M
    points_per_scene = 1 * 3 + M          # head evaluations per scene
    out = {
        "metric": "scenes/sec",
        "value": scenes_per_s,
        "unit": "scenes/s",
        "n_gpus": world,
        "steps": K,
        "warmup": args.warmup,
        "ms_per_step": T / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "c2: batch=32/GPU synthetic 40^3 TSDF, encoder + 3 grasp heads @1 query + "
                               "occupancy head @2048 queries/scene (literal train_giga call), fp32 MFMA",
                   "scenes_per_gpu_per_step": B, "occ_points_per_scene": M, "grasp_points_per_scene": 1,
                   "parallelism": f"scene-sharded x{world} (replicas, one all_gather of counters)"},
        "query_points_per_sec": scenes_per_s * points_per_scene,
        "algorithmic_tflops": scenes_per_s * flop_scene / 1e12,
        "roofline": {
            "kernel": STAGE_NAMES[dom], "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MATRIX_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MATRIX_TFLOPS, "traffic": traffic,
            "frac_of_measured_peak": achieved / MEASURED_F32_MATRIX_TFLOPS,
            "avg_launch_ms": dom_avg_ms, "flops_per_launch": dom_flops,
            "note": "fp32-input MFMA (v_mfma_f32_32x32x2_f32) peak; events on the launch stream inside the timed steps",
        },
        "stage_ms": {STAGE_NAMES[i]: round(v, 4) for i, v in enumerate(stage_ms)},
        "stage_note": "unet.conv_final is not launched in this call: the 1x1 convolution is folded into the heads' fc_c weights "
                      "(GIGA_FOLD_FINAL); its entry is the empty event bracket",
        "decoder_occ_ms": round(dec_ms_once, 4),
    }

    if dp_train is not None:
        out["extra"] = {"c5_train_step_fp32_data_parallel": dp_train}
    # -------- extra: c4 (64 000 grasp queries per scene, f16 MFMA fused decoder); single-GPU runs only ----
    single = world == 1
    if single and not args.no_extra:
        try:
            out["extra"] = {"c4": bench_c4(net, sd, dev, L, _capi, synth, decode_heads)}
            # BASELINE c4 is a sweep over the scenes per GPU; the decoder's MFMA fraction per batch size
            sweep = []
            for bc in (1, 8, 128):
                r = bench_c4(net, sd, dev, L, _capi, synth, decode_heads, Bc=bc, steps=10)
                sweep.append({"scenes": bc, "scenes_per_sec": r["scenes_per_sec"], "ms_per_step": r["ms_per_step"],
                              "decoder_ms": r["roofline"]["avg_launch_ms"], "decoder_tflops": r["roofline"]["achieved"],
                              "decoder_frac_of_f16_mfma_peak": r["roofline"]["frac"]})
            out["extra"]["c4_sweep"] = sweep
        except Exception as e:  # noqa: BLE001
            out["extra"] = {"c4_error": f"{type(e).__name__}: {e}"}

    # -------- extra: c2 (ii), N = M = 2048 queries per scene on all four heads -------------------
    if single and not args.no_extra:
        try:
            out["extra"]["c2_ii"] = bench_c2_ii(net, dev, synth, B)
        except Exception as e:  # noqa: BLE001
            out["extra"]["c2_ii_error"] = f"{type(e).__name__}: {e}"

    # -------- extra: c5-shaped training step (fp32 here; BASELINE c5 names bf16 -- see DESIGN.md) -----------
    if single and not args.no_extra:
        try:
            out["extra"]["c5_train_step_fp32"] = bench_train(net, dev, synth, B, M)
        except Exception as e:  # noqa: BLE001
            out["extra"]["c5_error"] = f"{type(e).__name__}: {e}"

    # -------- CPU baseline: the oracle (port of the reference path) on the host cores -------------------
    if single and not args.no_cpu_baseline:               # rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(sd, synth, M)
    print(json.dumps(out), flush=True)
    finish()


def bench_c4(net, sd, dev, L, _capi, synth, decode_heads, Bc=32, steps=10):
    N = 64000
    net.set_precision("fp16")
    blob = net.packed_blob(dev)
    from giga_amd.detection import query_lattice
    x = torch.from_numpy(synth.tsdf_batch(1000, Bc)).to(dev)
    lat = query_lattice(40, dev)          # the VGNImplicit lattice (detection_implicit.py:28-31), shared by the batch
    ev = [(L.giga_event_create(), L.giga_event_create()) for _ in range(steps)]

    def step(pr=None):
        with torch.no_grad():
            nhwc, _ = net.encoder.encode_nhwc(x, blob=blob, precision="fp16", fold_final=True)
            return decode_heads(nhwc, lat, blob, 7, "fp16", True, probe=pr, folded=True)

    for _ in range(8):                     # also absorbs the allocator's one-off work after a change of batch size
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(ev[i])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = ctypes.c_float()
    dms = []
    for a, b in ev:
        _capi.check(L.giga_event_elapsed_ms(a, b, ctypes.byref(ms)), "event")
        dms.append(ms.value)
        L.giga_event_destroy(a); L.giga_event_destroy(b)
    dec_ms = float(np.mean(dms))
    flops = Bc * N * FLOP_GRASP3
    ach = flops / (dec_ms * 1e-3) / 1e12
    net.set_precision("fp32")
    return {
        "workload": f"c4: batch={Bc} scenes x the 64000-point inference lattice, 3 grasp heads, f16 MFMA decoder "
                    f"(lattice path) + f16 encoder",
        "scenes_per_sec": Bc * steps / el, "query_points_per_sec": Bc * steps * N / el,
        "ms_per_step": el / steps * 1e3, "dtype": "f16 operands / f32 accumulate",
        "roofline": {"kernel": "decoder_f16_kernel", "bound": "mfma", "achieved": ach, "peak": PEAK_F16_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": ach / PEAK_F16_MFMA_TFLOPS, "traffic": None,
                     "frac_of_measured_peak": ach / MEASURED_F16_MFMA_TFLOPS,
                     "avg_launch_ms": dec_ms, "flops_per_launch": flops},
    }


def bench_c2_ii(net, dev, synth, B, steps=20):
    """BASELINE c2 (ii): every scene gets 2048 grasp queries (3 heads) and 2048 occupancy queries, fp32."""
    N = 2048
    net.set_precision("fp32").eval()
    x = torch.from_numpy(synth.tsdf_batch(3000, B)).to(dev)
    p = torch.from_numpy(synth.query_points(3000, B, N, stream=2)).to(dev)
    po = torch.from_numpy(synth.query_points(3000, B, N, stream=3)).to(dev)
    with torch.no_grad():
        for _ in range(5):
            net(x, p, p_tsdf=po)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x, p, p_tsdf=po)
        torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    flop_scene = FLOP_ENCODER + N * (FLOP_GRASP3 + FLOP_HEAD["tsdf"])
    return {"workload": f"c2 (ii): batch={B}, 2048 grasp queries x 3 heads + 2048 occupancy queries per scene, fp32",
            "ms_per_step": el * 1e3, "scenes_per_sec": B / el, "query_points_per_sec": B * 2 * N / el,
            "algorithmic_tflops": B * flop_scene / el / 1e12}


def bench_train(net, dev, synth, B, M, steps=10, rank=0, world=1):
    """One optimisation step of scripts/train_giga.py:198-211 on this rank's scenes.  world > 1: the backward
    all-reduces (means) the flat gradient bucket over RCCL (net.enable_data_parallel), BASELINE config c5."""
    from giga_amd.training import loss_fn, select
    net.set_precision("fp32").train()
    first = 2000 + rank * B
    x = torch.from_numpy(synth.tsdf_batch(first, B)).to(dev)
    pos = torch.from_numpy(synth.query_points(first, B, 1, stream=2)).to(dev)
    pos_occ = torch.from_numpy(synth.query_points(first, B, M, stream=3)).to(dev)
    y = tuple(torch.from_numpy(a).to(dev) for a in synth.train_labels(first, B, M))
    # the reference's optimiser (train_giga.py:49: Adam, lr 2e-4) in torch's single-launch form; the default
    # per-tensor foreach path costs 6 ms of host time per step for the 164 parameter tensors
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, fused=True)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = loss_fn(select(net(x, pos, p_tsdf=pos_occ)), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    net.eval()
    return {"workload": f"joint GIGA training step (train_giga.py:198-211): B={B} scenes/GPU, 1 grasp query + {M} "
                        f"occupancy queries, forward + HIP backward + fused Adam, fp32, {world} GPU"
                        + (", one RCCL all-reduce of the flat gradient bucket per step" if world > 1 else ""),
            "ms_per_step": el * 1e3, "scenes_per_sec": world * B / el, "final_loss": float(loss.detach())}


def cpu_baseline(sd, synth, M, budget_s=20.0):
    """Time the CPU oracle (oracle/giga_oracle.py, a port of the reference's PyTorch path pinned to
    reference-generated goldens) on a bounded sample of the same workload.  torch's intra-op pool
    collapses when oversubscribed on very wide hosts, so the thread count is calibrated first
    (best of a few candidates on a 2-scene pass) and reported as `cores`."""
    from oracle import giga_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    Bs = 8
    x = torch.from_numpy(synth.tsdf_batch(0, Bs))
    pos = torch.from_numpy(synth.query_points(0, Bs, 1, stream=2))
    pos_occ = torch.from_numpy(synth.query_points(0, Bs, M, stream=3))

    def one(n):
        t0 = time.perf_counter()
        O.model_forward(sd, x[:n], pos[:n], p_tsdf=pos_occ[:n])
        return time.perf_counter() - t0

    with torch.no_grad():
        cands = sorted({c for c in (8, 16, 32, 64, avail) if c <= avail})
        best, best_t = cands[0], None
        for c in cands:
            torch.set_num_threads(c)
            one(2)
            t = min(one(2), one(2))
            if best_t is None or t < best_t:
                best, best_t = c, t
            if t > 3.0:
                break
        torch.set_num_threads(best)
        one(Bs)
        times = []
        t_start = time.perf_counter()
        while len(times) < 10 and (not times or time.perf_counter() - t_start < budget_s):
            times.append(one(Bs))
        # SURVEY 8d also asks for the reference-shaped single-scene cases: c1 (1 scene, 2048 grasp + 2048 occupancy
        # queries) and the c4 shape (1 scene, the 64 000-point lattice, three grasp heads); median of three passes each
        also = {}
        x1 = x[:1]
        p1 = torch.from_numpy(synth.query_points(0, 1, 2048, stream=2))
        lat = torch.from_numpy(synth.inference_lattice())
        for name, fn in (("c1_scene_2048pts_ms", lambda: O.model_forward(sd, x1, p1, p_tsdf=p1)),
                         ("c4_scene_64000pts_ms", lambda: O.model_forward(sd, x1, lat))):
            fn()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            also[name] = float(np.median(ts)) * 1e3
    med = float(np.median(times))
    return {"value": Bs / med, "unit": "scenes/s", "cores": best, "kind": "port",
            "sample": f"{len(times)} passes of {Bs} scenes (1 grasp query + {M} occupancy queries each), "
                      f"torch {torch.__version__} CPU fp32, median; {best} intra-op threads chosen from "
                      f"{cands} on a host with {avail} usable cores",
            "ms_per_pass": med * 1e3, "also": also}


if __name__ == "__main__":
    main()
