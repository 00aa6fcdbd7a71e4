"""Landmark tracking + temporal smoothing that precede the crop step (CPU pre-processing;
restates reference ``modules/keep_processor.py:33-115`` and ``:216-231``).

Semantics kept: a track is a list of one ``[5,2]`` landmark array per frame, missing frames
are NaN; association is Hungarian assignment on landmark-centroid distance with a hard
threshold (75 px); gaps are linearly interpolated per coordinate and the result is smoothed
with ``gaussian_filter1d(sigma=2)`` along time.
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d
from scipy.optimize import linear_sum_assignment

_NAN_LM = lambda: np.full((5, 2), np.nan)  # noqa: E731


def interpolate_sequence(seq):
    """Fill NaNs of a 1-D series by linear interpolation over the valid samples (KP:33-40)."""
    out = np.copy(seq)
    missing = np.isnan(seq)
    if missing.any():
        x = np.arange(len(seq))
        out[missing] = np.interp(x[missing], x[~missing], seq[~missing])
    return out


def track_faces(all_frames_landmarks, distance_threshold=75.0):
    """dict track_id -> list (one entry per frame) of [5,2] arrays, NaN where unseen (KP:42-115)."""
    n_frames = len(all_frames_landmarks)
    tracks, next_id = {}, 0
    if all_frames_landmarks and all_frames_landmarks[0]:
        for lm in all_frames_landmarks[0]:
            tracks[next_id] = [lm]
            next_id += 1

    for i in range(1, n_frames):
        for data in tracks.values():                       # pad tracks that skipped frame i-1
            if len(data) < i:
                data.append(_NAN_LM())
        active = [(tid, data[-1]) for tid, data in tracks.items()
                  if len(data) == i and not np.all(np.isnan(data[-1]))]
        current = all_frames_landmarks[i]
        matched = set()
        if active and current:
            cost = np.full((len(active), len(current)), np.inf)
            for r, (_, prev) in enumerate(active):
                for c, cur in enumerate(current):
                    d = np.linalg.norm(prev.mean(axis=0) - cur.mean(axis=0))
                    if d < distance_threshold:
                        cost[r, c] = d
            if not np.all(np.isinf(cost)):
                rows, cols = linear_sum_assignment(cost)
                for r, c in zip(rows, cols):
                    if cost[r, c] != np.inf:
                        tracks[active[r][0]].append(current[c])
                        matched.add(c)
        for tid, _ in active:                              # active but unmatched this frame
            if len(tracks[tid]) == i:
                tracks[tid].append(_NAN_LM())
        for c in set(range(len(current))) - matched:      # new faces start new tracks
            tracks[next_id] = [_NAN_LM()] * i + [current[c]]
            next_id += 1

    for data in tracks.values():
        while len(data) < n_frames:
            data.append(_NAN_LM())
    return tracks


def _smooth(landmark_seq):
    """[N,10] with NaN gaps -> interpolated + gaussian(sigma=2) -> [N,5,2]."""
    arr = np.array(landmark_seq, dtype=np.float64)
    for j in range(arr.shape[1]):
        arr[:, j] = interpolate_sequence(arr[:, j])
    return gaussian_filter1d(arr, sigma=2, axis=0).reshape(arr.shape[0], 5, 2)


def smooth_center_face(raw_landmarks_per_frame):
    """only_center_face: at most one face per frame, no association needed (KP:216-222)."""
    seq = [(f[0] if f else _NAN_LM()).reshape(10) for f in raw_landmarks_per_frame]
    return {0: _smooth(seq)}


def smooth_tracked_faces(raw_landmarks_per_frame):
    """multi-face: Hungarian tracks, each smoothed independently (KP:223-231)."""
    out = {}
    if not any(raw_landmarks_per_frame):
        return out
    for tid, lms in track_faces(raw_landmarks_per_frame).items():
        seq = [lm.reshape(10) if not np.all(np.isnan(lm)) else np.full(10, np.nan) for lm in lms]
        out[tid] = _smooth(seq)
    return out
