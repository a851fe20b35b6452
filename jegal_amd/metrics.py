"""Task metrics on the GPU (evaluation/evaluate_{retrieval,spotting,asd}.py of the reference).

The N x N similarity matrix is never materialised: ``jg_sim_rank`` counts, per query row, the
gallery rows that beat / tie the diagonal (exact-fp32 MFMA), which is all ``compute_metrics``
(evaluate_retrieval.py:51-65) uses.  With ``torch.distributed`` initialised, queries are sharded
across ranks (contiguous blocks, the --rank/--nshard rule of extract_gestsync_feats.py:366-370) and
the gallery is assembled with one all-gather (RCCL over xGMI on MI355X).
"""
import ast
import math

import numpy as np
import torch

from ._lib import Engine
from . import dist as jdist


def _tensor(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, np.float32))


def _offsets(mats):
    off = np.zeros(len(mats) + 1, np.int32)
    off[1:] = np.cumsum([m.shape[0] for m in mats])
    return off


def video_level(engine, mats):
    """Temporal mean per clip (evaluate_retrieval.py:30-31): list of (T_i,512) -> (N,512) device."""
    cat = torch.as_tensor(np.concatenate([np.asarray(m, np.float32) for m in mats], 0))
    return engine.pool_mean(cat, _offsets(mats))


def metrics_from_ranks(rank, ties):
    """Rebuild the reference's `ind` vector (one entry per tied position) and its statistics."""
    rank = np.asarray(rank, np.int64)
    ties = np.asarray(ties, np.int64)
    ind = np.concatenate([np.arange(r, r + t) for r, t in zip(rank, ties)]) if len(rank) else np.zeros(0, np.int64)
    n = len(ind)
    out = {f"R{k}": float(np.sum(ind < k)) / n for k in (1, 5, 10, 25, 50)}
    out["MR"] = float(np.median(ind) + 1)
    return out


def retrieval_metrics(emb1, emb2, engine=None):
    """get_similarity_matrix + compute_metrics (evaluate_retrieval.py:38-65) for query set emb1
    against gallery emb2 (both (N,512), un-normalised video-level means).  Sharded over ranks when
    torch.distributed is initialised: every rank passes ITS contiguous block of rows."""
    eng = engine or Engine.get()
    e1 = eng.l2norm(_tensor(emb1))
    e2 = eng.l2norm(_tensor(emb2))
    if jdist.world_size() > 1:
        gallery, row_offset = jdist.all_gather_rows(e2)
    else:
        gallery, row_offset = e2, 0
    rank, ties = eng.sim_rank(e1, gallery, row_offset)
    if jdist.world_size() > 1:
        rank, _ = jdist.all_gather_rows(rank.to(torch.int32).reshape(-1, 1))
        ties, _ = jdist.all_gather_rows(ties.to(torch.int32).reshape(-1, 1))
    return metrics_from_ranks(rank.flatten().cpu().numpy(), ties.flatten().cpu().numpy())


def reduce_counts(counts, device=None):
    """Sum a short list of integer counters over the ranks (spotting / ASD need only this): every rank evaluates ITS
    contiguous block of clips (jdist.shard_range) and the totals are all-reduced -- RCCL on device tensors under nccl, host
    tensors under gloo.  Single process: the counts themselves."""
    counts = [int(c) for c in counts]
    if jdist.world_size() == 1:
        return counts
    t = torch.tensor(counts, dtype=torch.int64, device=device if jdist.backend() == "nccl" else "cpu")
    return [int(v) for v in jdist.all_reduce_sum(t).cpu()]


def spotting_counts(pred, score, word_boundaries, target_idx, thresh=0.5, frame_thresh=9):
    """(correct, total) of get_spotting_acc (evaluate_spotting.py:59-90) from the per-clip argmax frame and its probability:
    correct iff start - frame_thresh <= pred <= end + frame_thresh and score >= thresh (:77-84)."""
    correct = 0
    for i, (wb, t) in enumerate(zip(word_boundaries, target_idx)):
        s = max(0, wb[t][1] - frame_thresh)
        e = wb[t][2] + frame_thresh
        if s <= int(pred[i]) <= e and float(score[i]) >= thresh:
            correct += 1
    return correct, len(word_boundaries)


def spotting_accuracy(gestures, contents, word_boundaries, targets, thresh=0.5, frame_thresh=9, engine=None, offsets=None):
    """get_spotting_acc (evaluate_spotting.py:59-90).  ``targets`` are word indices (the reference
    looks the target boundary up in the clip's list, :70) or [word,start,end] boundaries.
    Sharded when torch.distributed is initialised: every rank passes ITS block of clips, the two counters are all-reduced.
    gestures / contents: lists of per-clip (T_i,512) / (W_i,512) arrays as the evaluators load them from the .pkl files, or -- with
    ``offsets=(g_offsets, c_offsets)`` -- the already concatenated (sum T,512) / (sum W,512) tensors (device-resident galleries)."""
    eng = engine or Engine.get()
    wbs = [ast.literal_eval(w) if isinstance(w, str) else w for w in word_boundaries]
    tidx = []
    for wb, t in zip(wbs, targets):
        if isinstance(t, str):
            t = ast.literal_eval(t)
        tidx.append(wb.index(t) if isinstance(t, (list, tuple)) else int(t))
    correct = 0
    if len(wbs):                                  # (a rank may hold no clip when there are fewer clips than ranks)
        if offsets is not None:
            g, c, (goff, coff) = gestures, contents, offsets
        else:
            g = torch.as_tensor(np.concatenate([np.asarray(x, np.float32) for x in gestures], 0))
            c = torch.as_tensor(np.concatenate([np.asarray(x, np.float32) for x in contents], 0))
            goff, coff = _offsets(gestures), _offsets(contents)
        pred, score = eng.spot(g, c, goff, coff, tidx)
        correct, _ = spotting_counts(pred.cpu().numpy(), score.cpu().numpy(), wbs, tidx, thresh, frame_thresh)
    correct, total = reduce_counts([correct, len(wbs)], eng.device)
    return 100.0 * correct / max(1, total)


def asd_counts(pred):
    """[correct for 2, 4, 6 speakers, queries] from jg_asd's (n,3) argmax indices: the positive is candidate 0 (evaluate_asd.py:101-113)."""
    pred = np.asarray(pred).reshape(-1, 3)
    return [int(np.sum(pred[:, k] == 0)) for k in range(3)] + [int(pred.shape[0])]


def asd_accuracy(query_content, candidate_gestures, engine=None):
    """evaluate_asd.py:94-113: query_content (N,512) video-level; candidate_gestures = list of (P_i,512)
    with the positive at index 0.  Returns accuracies for 2/4/6 speakers.
    Sharded when torch.distributed is initialised: every rank passes ITS block of queries, the four counters are all-reduced."""
    eng = engine or Engine.get()
    counts = [0, 0, 0, 0]
    if len(candidate_gestures):
        cand = torch.cat([_tensor(c).to(eng.device) for c in candidate_gestures], 0)
        counts = asd_counts(eng.asd(_tensor(query_content), cand, _offsets(candidate_gestures)).cpu().numpy())
    c2, c4, c6, n = reduce_counts(counts, eng.device)
    n = max(1, n)
    return c2 / n, c4 / n, c6 / n
