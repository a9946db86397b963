"""Host-side association of per-snippet predictions into tracks (numpy).

API mirror of ``associate_snippets`` and its helpers in /root/reference/inference_utils.py:97-112 (``transform_pts_np``,
``compute_match_cost``) and :198-339 (``associate_snippets``): consecutive snippets share one frame (or, for
single-frame snippets, are one ``seq_gap`` apart); persons of the new snippet are matched to the previous frame's persons
by a squared distance over (x / w, y / h, depth / max_depth, 0.1 * score) of all key-points -- a mutual-nearest rule: every
previous person proposes its nearest current person, every current person with at least one proposal keeps the closest
proposer -- and unmatched ones get fresh ids; on the shared frame the matched poses are averaged with their scores as
weights.  Not on the hot path (plain numpy, as in the reference); pinned by golden g8 (tests/test_inference_utils.py).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def transform_pts_np(pts: np.ndarray, trans: np.ndarray) -> np.ndarray:
    """[..., 2] points through a 2 x 3 affine map (reference :97-100)."""
    homo = np.concatenate([pts, np.ones_like(pts[..., :1])], axis=-1)
    return homo @ trans.T


def compute_match_cost(pre: np.ndarray, cur: np.ndarray, h: float, w: float, max_depth: float) -> np.ndarray:
    """[m, K, 4] x [n, K, 4] (x, y, depth, score) -> [m, n] squared distances in normalised units (reference :103-110)."""
    diff = pre[:, None] - cur[None]
    diff = diff / np.array([w, h, max_depth, 10.0], dtype=diff.dtype)        # (the score term enters as 0.1 * difference)
    return np.sum(diff ** 2, axis=(-1, -2))


def _frame_data(kpts, depth, scores, keep, inv_trans):
    """[m, K, 4] = (x, y in image coordinates, depth, score) of the kept persons; joint 0 (the root) is replaced by the
    mean of the two hips (joints 9 and 10), as the reference does before matching and storing."""
    data = np.concatenate([transform_pts_np(kpts[keep], inv_trans), depth[keep], scores[keep]], axis=-1)
    data[:, 0, :] = (data[:, 9, :] + data[:, 10, :]) / 2
    return data


def _match(pre_data, cur_data, h, w, max_depth):
    """For every current person the index of its matched previous person, or -1."""
    cost = compute_match_cost(pre_data, cur_data, h, w, max_depth)          # [m, n]
    proposal = np.argmin(cost, axis=1)                                      # previous -> current (may repeat)
    gated = np.full(cost.shape, np.inf)
    gated[np.arange(proposal.shape[0]), proposal] = cost[np.arange(proposal.shape[0]), proposal]
    cur2pre = np.argmin(gated, axis=0)
    cur2pre[np.all(np.isinf(gated), axis=0)] = -1
    return cur2pre


def associate_snippets(results: Sequence[dict], frame_indices: Sequence[int], all_filenames: Sequence[str], args
                       ) -> Tuple[Dict[int, Tuple[np.ndarray, np.ndarray]], int]:
    """-> ({frame index: (person ids [m], poses [m, K, 4])}, number of ids handed out).

    ``results[i]``: 'human_score' [Q, T], 'pred_kpt_scores' [Q, T, K, 1], 'pred_kpts' [Q, T, K, 2], 'pred_depth'
    [Q, T, K, 1], 'inv_trans' [2, 3], 'filenames' [T], 'img_size' (w, h); ``args``: seq_gap, num_frames, max_depth."""
    gap, T, max_depth = args.seq_gap, args.num_frames, args.max_depth
    frames: Dict[int, Tuple[np.ndarray, np.ndarray]] = {}
    next_id = 0
    for si, res in enumerate(results):
        human = res['human_score'] > 0.5
        alive = human.sum(axis=1) > 0                     # queries that are a person in at least one frame
        human = human[alive]
        scores, kpts, depth = res['pred_kpt_scores'][alive], res['pred_kpts'][alive], res['pred_depth'][alive]
        inv_trans = res['inv_trans']
        first = frame_indices[si]
        cur2pre = np.zeros([0], dtype=np.int64)
        pre_data = None
        if si == 0:
            seq_ids = np.arange(human.shape[0])
            next_id += human.shape[0]
        else:
            pre_ids, pre_data = frames[first] if T > 1 else frames[first - gap]
            here = human[:, 0]
            cur_data = _frame_data(kpts[:, 0], depth[:, 0], scores[:, 0], here, inv_trans)
            seq_ids = np.full(human.shape[0], -1, dtype=np.int32)
            if cur_data.shape[0] and pre_data.shape[0]:
                w, h = res['img_size']
                cur2pre = _match(pre_data, cur_data, h, w, max_depth)
                ids = np.empty(cur2pre.shape[0], dtype=np.int32)
                for i, j in enumerate(cur2pre):           # fresh ids in the order of the current persons
                    if j < 0:
                        ids[i], next_id = next_id, next_id + 1
                    else:
                        ids[i] = pre_ids[j]
                seq_ids[here] = ids
            fresh = seq_ids == -1                          # persons absent from the shared frame (or nothing to match)
            seq_ids[fresh] = next_id + np.arange(int(fresh.sum()))
            next_id += int(fresh.sum())
        for t in range(T):
            assert res['filenames'][t] == all_filenames[first + t * gap]
            here = human[:, t]
            data = _frame_data(kpts[:, t], depth[:, t], scores[:, t], here, inv_trans)
            if si > 0 and t == 0 and T > 1 and cur2pre.shape[0]:
                # the shared frame: score-weighted average of the matched poses, mean of their scores
                matched = np.nonzero(cur2pre != -1)[0]
                pre_pose, cur_pose = pre_data[cur2pre[matched]], data[matched]
                ps, cs = pre_pose[:, :, 3:4], cur_pose[:, :, 3:4]
                data[matched, :, 3:4] = (ps + cs) / 2
                data[matched, :, 0:3] = (ps * pre_pose[:, :, 0:3] + cs * cur_pose[:, :, 0:3]) / (ps + cs)
            frames[first + t * gap] = (seq_ids[here], data)
    return frames, next_id
