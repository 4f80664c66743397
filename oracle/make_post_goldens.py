"""tests/golden/g6_postprocess.npz from the REFERENCE's own process/bound/select (container-only).
    python -m oracle.make_post_goldens"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from giga_amd import synth  # noqa: E402
from oracle import ref_bootstrap  # noqa: E402


def main():
    ref_bootstrap.install()
    from vgn import detection_implicit as di
    R = 40
    out = {}
    pos = torch.from_numpy(synth.inference_lattice(R)).view(R, R, R, 3)
    for case, (seed, out_th, qual_th, force) in enumerate(((0, 0.1, 0.9, True), (1, 0.5, 0.8, False), (2, 0.1, 0.99, True))):
        tsdf, qual, rot, width = synth.post_volumes(seed, R)
        q, r, w = di.process(tsdf[None], qual.copy(), rot, width, out_th=out_th)
        q = di.bound(q, 0.3 / R)
        grasps, scores = di.select(q.copy(), pos, r, w, threshold=qual_th, force_detection=force, max_filter_size=4)
        centers = np.array([g.pose.translation for g in grasps], np.float32).reshape(-1, 3)
        widths = np.array([g.width for g in grasps], np.float32)
        quats = np.array([g.pose.rotation.as_quat() for g in grasps], np.float32).reshape(-1, 4)
        out.update({f"c{case}_params": np.array([seed, out_th, qual_th, float(force)]),
                    f"c{case}_qual_s2": q[::2, ::2, ::2].copy(), f"c{case}_qual_sum": np.float64(q.astype(np.float64).sum()),
                    f"c{case}_nonzero": np.int64((q > 0).sum()),
                    f"c{case}_scores": np.asarray(scores, np.float32), f"c{case}_centers": centers,
                    f"c{case}_widths": widths, f"c{case}_quats": quats})
        print(case, "nonzero", (q > 0).sum(), "grasps", len(grasps), scores[:3])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g6_postprocess.npz"), **out)


if __name__ == "__main__":
    main()
