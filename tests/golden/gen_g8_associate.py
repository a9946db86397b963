#!/usr/bin/env python3
"""Golden g8: the reference's ``associate_snippets`` (inference_utils.py:198-339) on seeded random snippet predictions.
Build container only (imports /root/reference with cv2 / matplotlib / imageio / tqdm stubbed: the function itself is
plain numpy).  Stores inputs and outputs as arrays; no reference source travels."""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
for name in ("cv2", "matplotlib", "matplotlib.pyplot", "imageio", "tqdm"):
    m = types.ModuleType(name)
    if name == "tqdm":
        m.tqdm = lambda x, *a, **k: x
    sys.modules.setdefault(name, m)
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
sys.path.insert(0, "/root/reference")
import inference_utils as ref      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make_case(seed, T, gap, n_snippets, Q=9, K=15):
    rng = np.random.RandomState(seed)
    skip = gap * (T - 1) if T > 1 else gap
    frame_indices = [i * skip for i in range(n_snippets)]
    n_files = frame_indices[-1] + gap * (T - 1) + 1
    files = [f"{i:06d}.jpg" for i in range(n_files)]
    base = rng.uniform(100, 800, (Q, 1, 1, 2))             # persistent persons drifting over time
    results = []
    for s, f0 in enumerate(frame_indices):
        perm = rng.permutation(Q)                           # query order differs per snippet
        kpts = (base + rng.normal(0, 30, (Q, 1, K, 2)) * 0 + rng.normal(0, 6, (Q, T, K, 2)) +
                np.linspace(0, 40, K).reshape(1, 1, K, 1) + 3.0 * (f0 + np.arange(T) * gap).reshape(1, T, 1, 1))[perm]
        results.append({
            "human_score": (rng.uniform(0, 1, (Q, T)) * np.where(rng.uniform(size=(Q, 1)) < 0.75, 1.6, 0.3)).clip(0, 1)[perm],
            "pred_kpt_scores": rng.uniform(0.05, 1, (Q, T, K, 1)),
            "pred_kpts": kpts * np.array([0.8, 0.6]),
            "pred_depth": rng.uniform(1, 12, (Q, 1, 1, 1))[perm] + rng.normal(0, 0.2, (Q, T, K, 1)),
            "inv_trans": np.array([[1.25, 0.0, -20.0], [0.0, 1.25, 7.5]]),
            "filenames": [files[f0 + t * gap] for t in range(T)],
            "img_size": np.array([1200.0, 675.0]),
        })
    return results, frame_indices, files


blob = {}
for name, (seed, T, gap, n) in {"t4_gap2": (1, 4, 2, 4), "t1_gap3": (2, 1, 3, 5), "t2_gap1": (3, 2, 1, 6)}.items():
    results, frame_indices, files = make_case(seed, T, gap, n)
    args = types.SimpleNamespace(seq_gap=gap, num_frames=T, num_future_frames=0, max_depth=15.0)
    frames, max_pid = ref.associate_snippets([dict(r) for r in results], frame_indices, files, args)
    blob[f"{name}_cfg"] = np.array([T, gap, n, max_pid])
    blob[f"{name}_frame_indices"] = np.array(frame_indices)
    for i, r in enumerate(results):
        for k in ("human_score", "pred_kpt_scores", "pred_kpts", "pred_depth", "inv_trans", "img_size"):
            blob[f"{name}_in{i}_{k}"] = r[k]
    blob[f"{name}_frames"] = np.array(sorted(frames))
    for f, (pids, data) in frames.items():
        blob[f"{name}_out{f}_pids"] = np.asarray(pids)
        blob[f"{name}_out{f}_data"] = data
np.savez_compressed(os.path.join(OUT, "g8_associate.npz"), **blob)
print("g8_associate.npz", os.path.getsize(os.path.join(OUT, "g8_associate.npz")), {k: blob[k].tolist() for k in blob if k.endswith("_cfg")})
