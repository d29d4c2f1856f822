"""CPU restatement of the hdbscan routines the reference's topic combination calls.

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/ may import this module (enstop_amd/ never does; the product's
own formulation is enstop_amd/hdbscan_tree.py).

Third-party dependency, absent from /root/reference and not installable in the build image:
    hdbscan >= 0.8.10   (requirements.txt:5; setup.py:44), imported at enstop/enstop_.py:21-23
    call sites: enstop_.py:291-298 (mst_linkage_core, label, _tree_to_labels on the KL mutual-reachability
    matrix), :340-345 (hdbscan.HDBSCAN(metric="precomputed", cluster_selection_method="leaf").fit_predict on the
    Hellinger matrix), :388-393 (HDBSCAN on the UMAP embedding: labels_ and probabilities_).

Parity status: UNPINNED by a run of hdbscan itself (it cannot be imported here).  The functions below restate
the library's published algorithm, routine by routine, as of the 0.8.x series:
    hdbscan/_hdbscan_reachability.pyx  mutual_reachability
    hdbscan/_hdbscan_linkage.pyx       mst_linkage_core (dense Prim), UnionFind, label
    hdbscan/_hdbscan_tree.pyx          bfs_from_hierarchy, condense_tree, compute_stability, get_cluster_tree_leaves,
                                       do_labelling, max_lambdas, get_probabilities, get_clusters ('leaf' branch,
                                       cluster_selection_epsilon = 0, allow_single_cluster = False)
    hdbscan/hdbscan_.py                _tree_to_labels, _hdbscan_generic (precomputed metric)
and are cross-checked in tests/test_hdbscan_tree.py against scikit-learn's PUBLIC estimator
sklearn.cluster.HDBSCAN (an independent implementation of the same algorithm; note its `min_samples` counts the
point itself: hdbscan's min_samples = s corresponds to scikit-learn's s + 1).
Pure-Python loops: meant for the few hundred topic vectors of an ensemble, not for data sets.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------
# hdbscan/_hdbscan_reachability.pyx :: mutual_reachability(distance_matrix, min_points, alpha=1.0)
# ------------------------------------------------------------------------------------------------
def mutual_reachability(distance_matrix, min_points=5):
    D = np.asarray(distance_matrix, dtype=np.float64)
    size = D.shape[0]
    min_points = min(size - 1, min_points)
    core = np.partition(D, min_points, axis=0)[min_points]      # the point itself sits at position 0
    stage1 = np.where(core > D, core, D)
    return np.where(core > stage1.T, core.T, stage1.T).T


# ------------------------------------------------------------------------------------------------
# hdbscan/_hdbscan_linkage.pyx :: mst_linkage_core -- Prim on a dense matrix; row i-1 is
# (node added in step i-1, node added in step i, its distance to the tree so far)
# ------------------------------------------------------------------------------------------------
def mst_linkage_core(distance_matrix):
    D = np.asarray(distance_matrix, dtype=np.float64)
    n = D.shape[0]
    result = np.zeros((n - 1, 3))
    current_node = 0
    current_distances = np.inf * np.ones(n)
    current_labels = np.arange(n, dtype=np.intp)
    for i in range(1, n):
        keep = current_labels != current_node
        current_labels = current_labels[keep]
        left = current_distances[keep]
        right = D[current_node][current_labels]
        current_distances = np.where(left < right, left, right)
        j = int(np.argmin(current_distances))
        new_node = current_labels[j]
        result[i - 1] = (current_node, new_node, current_distances[j])
        current_node = new_node
    return result


class _UnionFind:
    """hdbscan/_hdbscan_linkage.pyx :: UnionFind -- every union creates the next dendrogram node"""

    def __init__(self, n):
        self.parent = -1 * np.ones(2 * n - 1, dtype=np.intp)
        self.next_label = n
        self.size = np.hstack((np.ones(n, dtype=np.intp), np.zeros(n - 1, dtype=np.intp)))

    def union(self, m, n):
        self.size[self.next_label] = self.size[m] + self.size[n]
        self.parent[m] = self.next_label
        self.parent[n] = self.next_label
        self.next_label += 1

    def fast_find(self, n):
        p = n
        while self.parent[n] != -1:
            n = self.parent[n]
        while self.parent[p] != n and self.parent[p] != -1:      # path compression
            p, self.parent[p] = self.parent[p], n
        return n


def label(L):
    """hdbscan/_hdbscan_linkage.pyx :: label -- sorted MST edges -> scipy-style single-linkage hierarchy"""
    L = np.asarray(L, dtype=np.float64)
    result = np.zeros((L.shape[0], 4))
    U = _UnionFind(L.shape[0] + 1)
    for index in range(L.shape[0]):
        a, b, delta = int(L[index, 0]), int(L[index, 1]), L[index, 2]
        aa, bb = U.fast_find(a), U.fast_find(b)
        result[index] = (aa, bb, delta, U.size[aa] + U.size[bb])
        U.union(aa, bb)
    return result


# ------------------------------------------------------------------------------------------------
# hdbscan/_hdbscan_tree.pyx
# ------------------------------------------------------------------------------------------------
def bfs_from_hierarchy(hierarchy, bfs_root):
    dim = hierarchy.shape[0]
    num_points = dim + 1
    to_process = [int(bfs_root)]
    result = []
    while to_process:
        result.extend(to_process)
        to_process = [x - num_points for x in to_process if x >= num_points]
        if to_process:
            to_process = hierarchy[to_process, :2].flatten().astype(np.intp).tolist()
    return result


CONDENSED = [("parent", np.intp), ("child", np.intp), ("lambda_val", np.float64), ("child_size", np.intp)]


def condense_tree(hierarchy, min_cluster_size=10):
    root = 2 * hierarchy.shape[0]
    num_points = root // 2 + 1
    next_label = num_points + 1
    node_list = bfs_from_hierarchy(hierarchy, root)
    relabel = np.empty(root + 1, dtype=np.intp)
    relabel[root] = num_points
    rows = []
    ignore = np.zeros(len(node_list), dtype=bool)
    for node in node_list:
        if ignore[node] or node < num_points:
            continue
        left, right, dist = int(hierarchy[node - num_points, 0]), int(hierarchy[node - num_points, 1]), \
            hierarchy[node - num_points, 2]
        lam = 1.0 / dist if dist > 0.0 else np.inf
        lc = int(hierarchy[left - num_points, 3]) if left >= num_points else 1
        rc = int(hierarchy[right - num_points, 3]) if right >= num_points else 1
        if lc >= min_cluster_size and rc >= min_cluster_size:
            relabel[left] = next_label; next_label += 1
            rows.append((relabel[node], relabel[left], lam, lc))
            relabel[right] = next_label; next_label += 1
            rows.append((relabel[node], relabel[right], lam, rc))
        elif lc < min_cluster_size and rc < min_cluster_size:
            for side in (left, right):
                for sub in bfs_from_hierarchy(hierarchy, side):
                    if sub < num_points:
                        rows.append((relabel[node], sub, lam, 1))
                    ignore[sub] = True
        elif lc < min_cluster_size:
            relabel[right] = relabel[node]
            for sub in bfs_from_hierarchy(hierarchy, left):
                if sub < num_points:
                    rows.append((relabel[node], sub, lam, 1))
                ignore[sub] = True
        else:
            relabel[left] = relabel[node]
            for sub in bfs_from_hierarchy(hierarchy, right):
                if sub < num_points:
                    rows.append((relabel[node], sub, lam, 1))
                ignore[sub] = True
    return np.array(rows, dtype=CONDENSED)


def compute_stability(tree):
    smallest_cluster = tree["parent"].min()
    largest_child = max(tree["child"].max(), smallest_cluster)
    births = np.nan * np.ones(largest_child + 1)
    for child, lam in zip(tree["child"], tree["lambda_val"]):      # birth = the lambda at which the child appears
        births[child] = lam if np.isnan(births[child]) else min(births[child], lam)
    births[smallest_cluster] = 0.0
    out = {c: 0.0 for c in range(smallest_cluster, tree["parent"].max() + 1)}
    for parent, lam, size in zip(tree["parent"], tree["lambda_val"], tree["child_size"]):
        out[parent] += (lam - births[parent]) * size
    return out


def get_cluster_tree_leaves(cluster_tree):
    if cluster_tree.shape[0] == 0:
        return []

    def recurse(node):
        children = cluster_tree[cluster_tree["parent"] == node]["child"]
        if len(children) == 0:
            return [node]
        return sum([recurse(child) for child in children], [])
    return recurse(cluster_tree["parent"].min())


class _TreeUnionFind:
    def __init__(self, size):
        self.parent = np.arange(size)
        self.rank = np.zeros(size, dtype=np.intp)

    def find(self, x):
        if self.parent[x] != x:
            self.parent[x] = self.find(self.parent[x])
        return self.parent[x]

    def union_(self, x, y):
        xr, yr = self.find(x), self.find(y)
        if self.rank[xr] < self.rank[yr]:
            self.parent[xr] = yr
        elif self.rank[xr] > self.rank[yr]:
            self.parent[yr] = xr
        else:
            self.parent[yr] = xr
            self.rank[xr] += 1


def do_labelling(tree, clusters, cluster_label_map):
    root_cluster = tree["parent"].min()
    result = np.empty(root_cluster, dtype=np.intp)
    uf = _TreeUnionFind(tree["parent"].max() + 1)
    for child, parent in zip(tree["child"], tree["parent"]):
        if child not in clusters:
            uf.union_(parent, child)
    for n in range(root_cluster):
        cluster = uf.find(n)
        result[n] = -1 if cluster <= root_cluster else cluster_label_map[cluster]   # allow_single_cluster = False
    return result


def get_probabilities(tree, reverse_cluster_map, labels):
    result = np.zeros(labels.shape[0])
    deaths = {}
    for parent, lam in zip(tree["parent"], tree["lambda_val"]):      # max_lambdas
        deaths[parent] = max(deaths.get(parent, 0.0), lam)
    root_cluster = tree["parent"].min()
    for point, lam in zip(tree["child"], tree["lambda_val"]):
        if point >= root_cluster or labels[point] == -1:
            continue
        max_lambda = deaths[reverse_cluster_map[labels[point]]]
        if max_lambda == 0.0 or not np.isfinite(lam):
            result[point] = 1.0
        else:
            result[point] = min(lam, max_lambda) / max_lambda
    return result


def get_clusters_leaf(tree, stability):
    """get_clusters(tree, stability, cluster_selection_method='leaf', allow_single_cluster=False)"""
    node_list = sorted(stability.keys(), reverse=True)[:-1]          # the root is never a candidate
    cluster_tree = tree[tree["child_size"] > 1]
    is_cluster = {c: True for c in node_list}
    leaves = set(get_cluster_tree_leaves(cluster_tree))
    if len(leaves) == 0:
        for c in is_cluster:
            is_cluster[c] = False
        is_cluster[tree["parent"].min()] = True
    for c in is_cluster:                                             # selected = the leaves (epsilon = 0)
        is_cluster[c] = c in leaves
    clusters = set(c for c in is_cluster if is_cluster[c])
    cluster_map = {c: n for n, c in enumerate(sorted(clusters))}
    reverse = {n: c for c, n in cluster_map.items()}
    labels = do_labelling(tree, clusters, cluster_map)
    return labels, get_probabilities(tree, reverse, labels)


# ------------------------------------------------------------------------------------------------
# hdbscan/hdbscan_.py
# ------------------------------------------------------------------------------------------------
def tree_to_labels(single_linkage_tree, min_cluster_size=10):
    """_tree_to_labels(X, tree, min_cluster_size, cluster_selection_method='leaf') -> (labels, probabilities)"""
    condensed = condense_tree(single_linkage_tree, min_cluster_size)
    return get_clusters_leaf(condensed, compute_stability(condensed))


def labels_from_mutual_reachability(mr, min_cluster_size):
    """the call sequence of enstop_.py:291-298"""
    mst = mst_linkage_core(mr)
    mst = mst[np.argsort(mst.T[2])]
    return tree_to_labels(label(mst), min_cluster_size)


def hdbscan_precomputed_leaf(distance_matrix, min_samples, min_cluster_size):
    """hdbscan.HDBSCAN(min_samples, min_cluster_size, metric='precomputed', cluster_selection_method='leaf')
    -> _hdbscan_generic: mutual reachability, dense Prim, sort, label, _tree_to_labels (enstop_.py:340-345)"""
    size = np.asarray(distance_matrix).shape[0]
    min_samples = max(1, min(size - 1, min_samples))
    return labels_from_mutual_reachability(mutual_reachability(distance_matrix, min_samples), min_cluster_size)
