"""CPU restatement (test infrastructure) of the producers of the GP / BA inputs:

  TrackEngine::EstablishFullTracks       glomap/controllers/track_establishment.cc:5-17
    BlindConcatenation                   :19-63   union-find over inlier matches
    TrackCollection                      :65-152  collect members per root, discard inconsistent tracks
  TrackEngine::FindTracksForProblem      :154-227 length filter, (length, id)-descending greedy selection
  ViewGraph::KeepLargestConnectedComponents  glomap/scene/view_graph.cc:56-97 (+ :6-54, :160-173)

colmap::UnionFind<T> (colmap/math/union_find.h @ b6b7b54e) is un-vendored: restated from its published
behaviour — Find with path compression, Union(x, y) links root(x) under root(y).

Two restatements live here:
  * `*_literal`: the reference's loops, container by container (python dict / set in place of
    std::unordered_map / unordered_set), for small cases;
  * the vectorised numpy / scipy versions used at benchmark sizes, pinned against the literal ones in
    tests/test_oracle_tracks.py.

What the reference leaves to hash-table iteration order, and the canonical choice made here (and in the HIP
path):
  * the track id is the union-find root, which depends on the order in which image pairs are visited
    -> canonical id = the smallest global feature id (image << 32 | feature) of the track;
  * the order of a track's observations (unordered_set iteration) -> ascending (image, feature);
  * the order of the returned maps -> full tracks ascending by id, selected tracks in selection order
    (descending (length, id));
  * ties between equally large connected components (first one found wins) -> the one holding the
    smallest node index.
Everything else (the partition into tracks, which tracks are discarded, lengths, the selected set given
the ids) is order-independent and reproduced exactly.

PINNED TO REFERENCE CODE (round 5): track_establishment.cc and view_graph.cc are pure container / integer logic and compile
in this image — `make -C oracle ref` builds them FROM /root/reference, unmodified, against the stand-in scene types of
oracle/ref_shim/ into oracle/_ref/libref_glomap.so (oracle/ref_glue.cc, oracle/ref.py), and tests/test_oracle_ref.py holds
both restatements of this file to it: same partition, same discarded count, same selection for every option combination,
same largest component and pair flags.  (Still restated, not compiled: colmap::UnionFind, above.)

Only tests/, smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import connected_components


# ------------------------------------------------------------------------------------------------------
# literal restatements
# ------------------------------------------------------------------------------------------------------
class UnionFind:
    """colmap::UnionFind<T>: lazily created singletons, path-compressing Find, Union(x, y): root(x) -> root(y)."""

    def __init__(self):
        self.parent = {}

    def find(self, x):
        p = self.parent.setdefault(x, x)
        if p == x:
            return x
        # iterative path compression (recursion depth is unbounded in python)
        root = p
        while self.parent[root] != root:
            root = self.parent[root]
        while self.parent[x] != root:
            self.parent[x], x = root, self.parent[x]
        return root

    def union(self, x, y):
        rx, ry = self.find(x), self.find(y)
        if rx != ry:
            self.parent[rx] = ry


def _gid(image, feat):
    return (int(image) << 32) | int(feat)


def establish_full_tracks_literal(pair_image1, pair_image2, pair_valid, pair_offset, match_feat1, match_feat2,
                                  feat_offset, feat_xy, thres_inconsistency=10.0):
    """track_establishment.cc:5-152.  Returns ({track_id: [(image, feature), ...]} with the reference's own ids
    (union-find roots for THIS pair order) and empty lists for the discarded tracks, the discarded count, and the
    member sets {track_id: {global ids}} so that canonicalize() can name the discarded tracks too)."""
    uf = UnionFind()
    npairs = len(pair_image1)
    for p in range(npairs):  # BlindConcatenation :19-63
        if pair_valid is not None and not pair_valid[p]:
            continue
        i1, i2 = int(pair_image1[p]), int(pair_image2[p])
        for m in range(int(pair_offset[p]), int(pair_offset[p + 1])):
            g1, g2 = _gid(i1, match_feat1[m]), _gid(i2, match_feat2[m])
            if g2 < g1:
                uf.union(g1, g2)
            else:
                uf.union(g2, g1)
    track_map = {}
    for p in range(npairs):  # TrackCollection :65-113
        if pair_valid is not None and not pair_valid[p]:
            continue
        i1, i2 = int(pair_image1[p]), int(pair_image2[p])
        for m in range(int(pair_offset[p]), int(pair_offset[p + 1])):
            g1, g2 = _gid(i1, match_feat1[m]), _gid(i2, match_feat2[m])
            s = track_map.setdefault(uf.find(g1), set())
            s.add(g1)
            s.add(g2)
    tracks, discarded = {}, 0
    for tid, members in track_map.items():  # :115-148
        seen = {}
        obs = tracks.setdefault(tid, [])
        for g in members:
            image, feat = g >> 32, g & 0xFFFFFFFF
            xy = feat_xy[int(feat_offset[image]) + feat]
            if image in seen:
                bad = False
                for other in seen[image]:
                    d = other - xy
                    if np.sqrt(d[0] * d[0] + d[1] * d[1]) > thres_inconsistency:
                        bad = True
                        break
                if bad:
                    obs.clear()
                    discarded += 1
                    break
            else:
                seen[image] = []
            seen[image].append(xy)
            obs.append((image, feat))
    return tracks, discarded, track_map


def canonicalize(tracks, members=None):
    """{id: [(image, feat)]} with arbitrary ids / orders  ->  canonical CSR (track_id, track_offset, obs_image,
    obs_feature).  A discarded (empty) track keeps its slot; its canonical id comes from `members`."""
    items = []
    for tid, obs in tracks.items():
        o = sorted(obs)
        if members is not None:
            cid = min(members[tid])
        else:
            cid = min(_gid(i, f) for i, f in o) if o else int(tid)
        items.append((cid, o))
    items.sort(key=lambda t: t[0])
    tid = np.array([t[0] for t in items], dtype=np.int64)
    off = np.zeros(len(items) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(t[1]) for t in items])
    flat = [o for t in items for o in t[1]]
    img = np.array([o[0] for o in flat], dtype=np.int32)
    ft = np.array([o[1] for o in flat], dtype=np.uint32)
    return tid, off, img, ft


def find_tracks_for_problem_literal(track_id, track_offset, obs_image, obs_feature, image_registered,
                                    min_num_tracks_per_view=-1, min_num_view_per_track=3, max_num_view_per_track=100,
                                    max_num_tracks=10000000):
    """track_establishment.cc:154-227, loop for loop.  The comparisons between size_t / track_t (uint64) values and
    the int options follow C++'s usual arithmetic conversions: a negative option compares as 2^64 - |x|."""
    def u64(x):
        return int(x) & 0xFFFFFFFFFFFFFFFF

    T = len(track_id)
    lens = np.diff(track_offset)
    cand = [(int(lens[t]), int(track_id[t]), t) for t in range(T)
            if not (int(lens[t]) < u64(min_num_view_per_track)) and not (int(lens[t]) > u64(max_num_view_per_track))]
    cand.sort(reverse=True)  # std::sort(rbegin, rend) on (length, id)
    per_cam = {int(i): 0 for i in np.nonzero(image_registered)[0]}
    cameras_left = len(per_cam)
    limit = u64(min_num_tracks_per_view)
    sel = []
    for _, _, t in cand:
        temp = [(int(obs_image[k]), int(obs_feature[k])) for k in range(int(track_offset[t]), int(track_offset[t + 1]))
                if int(obs_image[k]) in per_cam]
        if len({i for i, _ in temp}) < u64(min_num_view_per_track):
            continue
        added = False
        for image, _ in temp:
            if per_cam[image] > limit:
                continue
            per_cam[image] += 1
            if per_cam[image] > limit:
                cameras_left -= 1
            if not added:
                sel.append((t, temp))
                added = True
        if cameras_left == 0:
            break
        if len(sel) > u64(max_num_tracks):
            break
    return _pack_selection(track_id, sel)


def _pack_selection(track_id, sel):
    tid = np.array([int(track_id[t]) for t, _ in sel], dtype=np.int64)
    off = np.zeros(len(sel) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(o) for _, o in sel])
    img = np.array([i for _, o in sel for i, _ in o], dtype=np.int32)
    ft = np.array([f for _, o in sel for _, f in o], dtype=np.uint32)
    return tid, off, img, ft


def keep_largest_connected_component_literal(num_nodes, edge_i, edge_j, edge_valid, node_num_images=None):
    """view_graph.cc:6-97 with dense node indices (frames).  Returns (node_registered [N] bool, edge_valid' [E] bool,
    number of registered images).  Nodes are visited in index order, so the first largest component found is the
    one holding the smallest node index (the canonical tie-break)."""
    adj = {}
    for e in range(len(edge_i)):
        if edge_valid[e]:
            adj.setdefault(int(edge_i[e]), set()).add(int(edge_j[e]))
            adj.setdefault(int(edge_j[e]), set()).add(int(edge_i[e]))
    visited, comps = set(), []
    for root in sorted(adj):
        if root in visited:
            continue
        comp, queue = {root}, [root]
        visited.add(root)
        while queue:
            cur = queue.pop(0)
            for nb in adj[cur]:
                if nb not in visited:
                    visited.add(nb)
                    comp.add(nb)
                    queue.append(nb)
        comps.append(comp)
    best, best_size = None, 0
    for c in comps:
        if len(c) > best_size:
            best, best_size = c, len(c)
    reg = np.zeros(num_nodes, dtype=bool)
    ev = np.asarray(edge_valid, dtype=bool).copy()
    if best is None:
        return None, ev, 0  # "return 0" before touching anything (:70)
    reg[list(best)] = True
    ev &= reg[np.asarray(edge_i)] & reg[np.asarray(edge_j)]
    w = np.ones(num_nodes, dtype=np.int64) if node_num_images is None else np.asarray(node_num_images, dtype=np.int64)
    return reg, ev, int(w[reg].sum())


# ------------------------------------------------------------------------------------------------------
# vectorised versions (benchmark sizes)
# ------------------------------------------------------------------------------------------------------
def _ranges(starts, lens):
    """concatenate(arange(s, s + l) for s, l in zip(starts, lens)) without the python loop."""
    starts, lens = np.asarray(starts, dtype=np.int64), np.asarray(lens, dtype=np.int64)
    total = int(lens.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    excl = np.cumsum(lens) - lens
    return np.repeat(starts - excl, lens) + np.arange(total, dtype=np.int64)


def _match_nodes(pair_image1, pair_image2, pair_valid, pair_offset, match_feat1, match_feat2, feat_offset):
    counts = np.diff(pair_offset)
    pid = np.repeat(np.arange(len(counts)), counts)
    u = feat_offset[np.asarray(pair_image1)[pid]] + match_feat1.astype(np.int64)
    v = feat_offset[np.asarray(pair_image2)[pid]] + match_feat2.astype(np.int64)
    if pair_valid is not None:
        ok = np.asarray(pair_valid, dtype=bool)[pid]
        u, v = u[ok], v[ok]
    return u, v


def establish_full_tracks(pair_image1, pair_image2, pair_valid, pair_offset, match_feat1, match_feat2,
                          feat_offset, feat_xy, thres_inconsistency=10.0):
    """Canonical CSR of EstablishFullTracks: (track_id [T], track_offset [T+1], obs_image, obs_feature, discarded).
    Discarded tracks keep their slot with zero observations (the reference keeps them as empty Track objects)."""
    feat_offset = np.asarray(feat_offset, dtype=np.int64)
    F = int(feat_offset[-1])
    u, v = _match_nodes(pair_image1, pair_image2, pair_valid, pair_offset, np.asarray(match_feat1), np.asarray(match_feat2),
                        feat_offset)
    if len(u) == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.uint32), 0
    g = sp.coo_matrix((np.ones(len(u), dtype=np.int8), (u, v)), shape=(F, F)).tocsr()
    _, lab = connected_components(g, directed=False)
    touched = np.zeros(F, dtype=bool)
    touched[u] = True
    touched[v] = True
    nodes = np.nonzero(touched)[0]  # ascending global node index == ascending (image, feature)
    labs = lab[nodes]
    _, first = np.unique(labs, return_index=True)  # nodes ascending => first occurrence = smallest member
    cmin = np.zeros(int(lab.max()) + 1, dtype=np.int64)
    cmin[labs[first]] = nodes[first]
    key = cmin[labs]
    order = np.argsort(key, kind="stable")  # tracks ascending by canonical id, members ascending within
    nodes, key = nodes[order], key[order]
    head = np.ones(len(nodes), dtype=bool)
    head[1:] = key[1:] != key[:-1]
    starts = np.nonzero(head)[0]
    lens = np.diff(np.append(starts, len(nodes)))
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    img = (np.searchsorted(feat_offset, nodes, side="right") - 1).astype(np.int32)
    ft = (nodes - feat_offset[img]).astype(np.uint32)
    tid = (img[off[:-1]].astype(np.int64) << 32) | ft[off[:-1]].astype(np.int64)
    # inconsistency: any two members in one image further apart than the threshold (:126-137; the reference tests
    # each new member against all earlier ones of its image, i.e. all pairs)
    trk = np.repeat(np.arange(len(lens)), lens)
    bad = np.zeros(len(lens), dtype=bool)
    xy = np.asarray(feat_xy, dtype=np.float64)
    shift = 1
    while True:
        a = np.arange(shift, len(nodes))
        same = (trk[a] == trk[a - shift]) & (img[a] == img[a - shift])
        if not same.any():
            break
        a = a[same]
        d = xy[nodes[a]] - xy[nodes[a - shift]]
        far = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) > thres_inconsistency
        bad[trk[a[far]]] = True
        shift += 1
    keep = ~bad[trk]
    lens2 = np.where(bad, 0, lens)
    off2 = np.zeros(len(lens) + 1, dtype=np.int64)
    off2[1:] = np.cumsum(lens2)
    return tid, off2, img[keep], ft[keep], int(bad.sum())


def find_tracks_for_problem(track_id, track_offset, obs_image, obs_feature, image_registered,
                            min_num_tracks_per_view=-1, min_num_view_per_track=3, max_num_view_per_track=100,
                            max_num_tracks=10000000):
    """Vectorised FindTracksForProblem.  The greedy loop is order-dependent only through the per-camera counters, and
    a counter is `min(limit + 1, number of earlier observations of that camera in tracks that pass the filters)`:
    it never depends on whether those tracks were inserted.  Hence track s is inserted iff one of its registered
    observations has rank <= limit among its camera's observations in selection order."""
    def u64(x):
        return int(x) & 0xFFFFFFFFFFFFFFFF

    track_id = np.asarray(track_id, dtype=np.int64)
    track_offset = np.asarray(track_offset, dtype=np.int64)
    obs_image = np.asarray(obs_image)
    reg = np.asarray(image_registered, dtype=bool)
    lens = np.diff(track_offset)
    T = len(lens)
    trk = np.repeat(np.arange(T), lens)
    oreg = reg[obs_image]
    nreg = np.bincount(trk, weights=oreg, minlength=T).astype(np.int64)
    # unique registered images per track
    key = trk.astype(np.int64) * (int(obs_image.max()) + 1 if len(obs_image) else 1) + obs_image
    uniq = np.zeros(T, dtype=np.int64)
    if len(key):
        k = np.unique(key[oreg])
        uniq = np.bincount(k // (int(obs_image.max()) + 1), minlength=T)
    surv = ~(lens.astype(np.uint64) < np.uint64(u64(min_num_view_per_track))) & \
           ~(lens.astype(np.uint64) > np.uint64(u64(max_num_view_per_track))) & \
           ~(uniq.astype(np.uint64) < np.uint64(u64(min_num_view_per_track)))
    cand = np.nonzero(surv)[0]
    order = cand[np.lexsort((track_id[cand].astype(np.uint64), lens[cand]))[::-1]]
    limit = u64(min_num_tracks_per_view)
    if limit >= (1 << 62):
        added = np.ones(len(order), dtype=bool)
    else:
        # registered observations of the surviving tracks, in selection order
        ko = _ranges(track_offset[order], lens[order])
        ks = np.repeat(np.arange(len(order)), lens[order])
        m = oreg[ko]
        ko, ks = ko[m], ks[m]
        cam = obs_image[ko]
        o2 = np.argsort(cam, kind="stable")
        cs = cam[o2]
        first = np.searchsorted(cs, cs, side="left")
        rank = np.arange(len(cs)) - first
        added = np.zeros(len(order), dtype=bool)
        added[ks[o2[rank <= limit]]] = True
    prefix = np.cumsum(added) - added  # exclusive
    chosen = order[added & (prefix.astype(np.uint64) <= np.uint64(u64(max_num_tracks)))]
    rr = _ranges(track_offset[chosen], lens[chosen])
    rr = rr[oreg[rr]]
    tid = track_id[chosen]
    off = np.zeros(len(chosen) + 1, dtype=np.int64)
    off[1:] = np.cumsum(nreg[chosen])
    return tid, off, obs_image[rr].astype(np.int32), np.asarray(obs_feature)[rr].astype(np.uint32)


def keep_largest_connected_component(num_nodes, edge_i, edge_j, edge_valid, node_num_images=None):
    edge_i, edge_j = np.asarray(edge_i), np.asarray(edge_j)
    ev = np.asarray(edge_valid, dtype=bool).copy()
    if not ev.any():
        return None, ev, 0
    g = sp.coo_matrix((np.ones(int(ev.sum()), dtype=np.int8), (edge_i[ev], edge_j[ev])), shape=(num_nodes, num_nodes)).tocsr()
    _, lab = connected_components(g, directed=False)
    touched = np.zeros(num_nodes, dtype=bool)
    touched[edge_i[ev]] = True
    touched[edge_j[ev]] = True
    size = np.bincount(lab[touched], minlength=lab.max() + 1)
    first = np.full(lab.max() + 1, num_nodes, dtype=np.int64)
    np.minimum.at(first, lab[touched], np.nonzero(touched)[0])
    best = np.lexsort((first, -size))[0]
    reg = touched & (lab == best)
    ev &= reg[edge_i] & reg[edge_j]
    w = np.ones(num_nodes, dtype=np.int64) if node_num_images is None else np.asarray(node_num_images, dtype=np.int64)
    return reg, ev, int(w[reg].sum())
