"""The tree step of HDBSCAN* over the few hundred topic vectors of an ensemble (host NumPy).

Replaces the three hdbscan routines the reference calls on the KL mutual-reachability matrix
(`mst_linkage_core`, `label`, `_tree_to_labels`, enstop/enstop_.py:291-298) and the
`hdbscan.HDBSCAN(metric="precomputed", cluster_selection_method="leaf")` estimator it runs on the
Hellinger matrix (enstop_.py:340-345).  No third-party clustering package and no private
scikit-learn module is involved.

Formulation (not hdbscan's code): the order in which Prim's algorithm adds the points has the
property that every single-linkage cluster is a contiguous INTERVAL of that order, and the
single-linkage dendrogram of the data equals the dendrogram of the path
v_0 - v_1 - ... - v_{t-1} whose edge i carries the distance of v_i to {v_0 .. v_{i-1}}.  So the
dendrogram is a set of nested intervals: a merge joins two adjacent intervals, "all points below a
node" is a slice, and the condensed tree needs no graph traversal per pruned branch.  Conventions
that decide the OUTPUT are hdbscan's (0.8.x, the reference's dependency), because a drop-in must number
the clusters the same way:
  * core distance = the `min_samples`-th smallest entry of a row, the zero self-distance included
    (enstop_.py:284 does the same by hand; scikit-learn's estimator counts one neighbour fewer),
  * Prim starts at point 0 and breaks ties towards the lowest index; edges are ordered with
    `np.argsort` (the reference's own call, enstop_.py:293),
  * cluster ids grow in breadth-first order of the dendrogram, left (the side holding the earlier
    Prim point) before right; labels are the ranks of the selected ids,
  * "leaf" selection: the clusters of the condensed tree that never split again; no split at all
    means no cluster (every label -1), `allow_single_cluster` being False on both reference paths,
  * membership strength = lambda at which the point leaves its cluster / the largest lambda in it.
Checked in tests/test_hdbscan_tree.py against the test tree's routine-by-routine restatement of
hdbscan's published algorithm and against scikit-learn's public HDBSCAN estimator.
"""
import numpy as np


def core_distances(distance_matrix, min_samples):
    D = np.asarray(distance_matrix, dtype=np.float64)
    k = min(D.shape[0] - 1, int(min_samples))
    return np.partition(D, k, axis=0)[k]


def mutual_reachability(distance_matrix, min_samples):
    """max(d(a, b), core(a), core(b)) for a symmetric distance matrix."""
    D = np.asarray(distance_matrix, dtype=np.float64)
    core = core_distances(D, min_samples)
    return np.maximum(np.maximum(D, core[:, None]), core[None, :])


def prim_path(W):
    """Order in which Prim's algorithm (start: point 0) adds the points of the complete graph with
    weight matrix W, and for each added point its distance to the points added before it."""
    W = np.asarray(W, dtype=np.float64)
    t = W.shape[0]
    order = np.zeros(t, dtype=np.intp)
    reach = np.zeros(t)
    outside = np.ones(t, dtype=bool)
    best = np.full(t, np.inf)
    cur = 0
    outside[0] = False
    for i in range(1, t):
        row = W[cur]
        best = np.where(best < row, best, row)          # a NaN weight never wins over a number already held
        cand = np.flatnonzero(outside)
        cur = cand[np.argmin(best[cand])]               # lowest index among equal distances
        order[i] = cur
        reach[i] = best[cur]
        outside[cur] = False
    return order, reach


def interval_dendrogram(reach):
    """Single-linkage dendrogram of the Prim path.  Node ids: 0..t-1 are path POSITIONS (leaves), t.. are
    merges in order of increasing edge weight.  Returns (left, right, dist, lo, hi): children, merge
    distance and the closed interval of path positions every node t + j covers."""
    t = reach.shape[0]
    edges = np.argsort(reach[1:]) + 1                   # edge e joins positions e - 1 and e
    top = np.arange(t, dtype=np.intp)                   # current dendrogram node of the interval starting / ending here
    start_of = np.arange(t, dtype=np.intp)              # for an interval's last position: its first
    end_of = np.arange(t, dtype=np.intp)                # for an interval's first position: its last
    left = np.empty(t - 1, dtype=np.intp); right = np.empty(t - 1, dtype=np.intp)
    lo = np.empty(t - 1, dtype=np.intp); hi = np.empty(t - 1, dtype=np.intp)
    dist = np.empty(t - 1)
    for j, e in enumerate(edges):
        a0 = start_of[e - 1]                            # interval ending at e - 1 is [a0, e - 1]
        b1 = end_of[e]                                  # interval starting at e is [e, b1]
        left[j], right[j] = top[a0], top[e]
        dist[j] = reach[e]
        lo[j], hi[j] = a0, b1
        top[a0] = t + j
        end_of[a0] = b1
        start_of[b1] = a0
    return left, right, dist, lo, hi


def leaf_clusters(reach, min_cluster_size):
    """Condensed tree of the Prim path with "leaf" selection.  Returns per path POSITION the cluster it
    belongs to (-1: noise) and its membership strength."""
    t = reach.shape[0]
    labels = np.full(t, -1, dtype=np.intp)
    strength = np.zeros(t)
    if t < 2:
        return labels, strength
    left, right, dist, lo, hi = interval_dendrogram(reach)

    def size(node):
        return 1 if node < t else hi[node - t] - lo[node - t] + 1

    def span(node):
        return (node, node) if node < t else (lo[node - t], hi[node - t])

    root = 2 * t - 2
    next_id = 1                                          # the root cluster is 0
    fell_from = np.zeros(t, dtype=np.intp)               # cluster a position drops out of ...
    fell_at = np.zeros(t)                                # ... and the lambda at which it does
    splits = set()                                       # clusters that split into two clusters
    level = [(root, 0)]
    while level:                                         # breadth first: cluster ids follow this order
        nxt = []
        for node, cid in level:
            j = node - t
            lam = 1.0 / dist[j] if dist[j] > 0.0 else np.inf
            big = [size(ch) >= min_cluster_size for ch in (left[j], right[j])]
            for ch, is_big in zip((left[j], right[j]), big):
                if is_big and all(big):
                    splits.add(cid)
                    if ch >= t:
                        nxt.append((ch, next_id))
                    else:                                # min_cluster_size == 1: a single point is a cluster
                        fell_from[ch], fell_at[ch] = next_id, np.inf
                    next_id += 1
                elif is_big:                             # the cluster lives on in its large side
                    if ch >= t:
                        nxt.append((ch, cid))
                    else:
                        fell_from[ch], fell_at[ch] = cid, lam
                else:                                    # too small to be a cluster: its points leave `cid` here
                    a, b = span(ch)
                    fell_from[a:b + 1] = cid
                    fell_at[a:b + 1] = lam
        level = nxt
    selected = sorted(c for c in range(1, next_id) if c not in splits)
    if not splits:
        return labels, strength                          # the data never separates into two clusters
    rank = {c: r for r, c in enumerate(selected)}
    for c in selected:
        mine = fell_from == c
        lam = fell_at[mine]
        peak = lam.max() if lam.size else 0.0
        labels[mine] = rank[c]
        with np.errstate(invalid="ignore"):
            s = np.where(np.isfinite(lam), np.minimum(lam, peak) / peak, 1.0) if peak != 0.0 else np.ones_like(lam)
        strength[mine] = s
    return labels, strength


def labels_from_mutual_reachability(mutual_reachability_matrix, min_cluster_size):
    """(labels, membership strengths) of the leaf clusters of HDBSCAN* on a mutual-reachability matrix
    (enstop_.py:291-298: mst_linkage_core -> argsort -> label -> _tree_to_labels(..., "leaf"))."""
    order, reach = prim_path(mutual_reachability_matrix)
    lab_pos, str_pos = leaf_clusters(reach, int(min_cluster_size))
    labels = np.empty_like(lab_pos)
    strength = np.empty_like(str_pos)
    labels[order] = lab_pos
    strength[order] = str_pos
    # ranks were taken over cluster ids in path terms; ids follow the dendrogram, not the point numbering, so
    # nothing is left to translate: `labels[p]` is the cluster of point p
    return labels, strength


def hdbscan_precomputed_leaf(distance_matrix, min_samples, min_cluster_size):
    """hdbscan.HDBSCAN(min_samples=, min_cluster_size=, metric="precomputed", cluster_selection_method="leaf")
    (enstop_.py:340-345) -> (labels_, probabilities_)."""
    D = np.asarray(distance_matrix, dtype=np.float64)
    k = max(1, min(D.shape[0] - 1, int(min_samples)))
    return labels_from_mutual_reachability(mutual_reachability(D, k), min_cluster_size)
