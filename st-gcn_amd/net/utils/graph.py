"""Skeleton graph and its partitioned adjacency, as consumed by the graph-convolution kernels.

Same constructor and attributes as the reference Graph (net/utils/graph.py:25-36 of
hensyou/st-gcn): Graph(layout, strategy, max_hop, dilation) with .A (K,V,V) float64, .edge,
.num_node, .center, .hop_dis.  A[k, j, i] weights source joint j into target joint i; columns
are normalised by their degree (A . D^-1, reference graph.py:139-147).
"""
from collections import deque

import numpy as np

_BONES = {
    # reference graph.py:42-72 ; (joints, index base, bones, centre joint)
    "openpose": (18, 0, ((4, 3), (3, 2), (7, 6), (6, 5), (13, 12), (12, 11), (10, 9), (9, 8), (11, 5),
                         (8, 2), (5, 1), (2, 1), (0, 1), (15, 0), (14, 0), (17, 15), (16, 14)), 1),
    "ntu-rgb+d": (25, 1, ((1, 2), (2, 21), (3, 21), (4, 3), (5, 21), (6, 5), (7, 6), (8, 7), (9, 21),
                          (10, 9), (11, 10), (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1),
                          (18, 17), (19, 18), (20, 19), (22, 23), (23, 8), (24, 25), (25, 12)), 20),
    "ntu_edge": (24, 1, ((1, 2), (3, 2), (4, 3), (5, 2), (6, 5), (7, 6), (8, 7), (9, 2), (10, 9),
                         (11, 10), (12, 11), (13, 1), (14, 13), (15, 14), (16, 15), (17, 1), (18, 17),
                         (19, 18), (20, 19), (21, 22), (22, 8), (23, 24), (24, 12)), 2),
}


class Graph:
    def __init__(self, layout="openpose", strategy="uniform", max_hop=1, dilation=1):
        self.max_hop = max_hop
        self.dilation = dilation
        self.get_edge(layout)
        self.hop_dis = _hop_distance(self.num_node, self.edge, max_hop)
        self.get_adjacency(strategy)

    def __str__(self):
        return str(self.A)

    def get_edge(self, layout):
        if layout not in _BONES:
            raise ValueError("Do Not Exist This Layout.")
        n, base, bones, centre = _BONES[layout]
        self.num_node = n
        self.edge = [(i, i) for i in range(n)] + [(a - base, b - base) for a, b in bones]
        self.center = centre

    def get_adjacency(self, strategy):
        n, hop = self.num_node, self.hop_dis
        hops = list(range(0, self.max_hop + 1, self.dilation))
        linked = np.isin(hop, hops).astype(np.float64)
        deg = linked.sum(axis=0)
        norm = linked / np.where(deg > 0, deg, 1.0)[None, :]
        if strategy == "uniform":
            self.A = norm[None].copy()
        elif strategy == "distance":
            self.A = np.stack([norm * (hop == h) for h in hops])
        elif strategy == "spatial":
            d = hop[:, self.center]
            src, dst = d[:, None], d[None, :]          # entry [j, i]: source j, target i
            planes = []
            for h in hops:
                ring = hop == h
                root = norm * (ring & (src == dst))
                closer = norm * (ring & (src > dst))
                further = norm * (ring & (src < dst))
                if h == 0:
                    planes.append(root)
                else:
                    planes.extend((root + closer, further))
            self.A = np.stack(planes)
        else:
            raise ValueError("Do Not Exist This Strategy")


def _hop_distance(n, edge, max_hop):
    adj = [set() for _ in range(n)]
    for i, j in edge:
        if i != j:
            adj[i].add(j)
            adj[j].add(i)
    hop = np.full((n, n), np.inf)
    for s in range(n):
        hop[s, s] = 0
        frontier, seen = deque([s]), {s}
        for d in range(1, max_hop + 1):
            nxt = deque()
            for u in frontier:
                for w in adj[u]:
                    if w not in seen:
                        seen.add(w)
                        hop[s, w] = d
                        nxt.append(w)
            frontier = nxt
    return hop
