"""oracle/tracks.py against the REFERENCE'S OWN code.

oracle/_ref/libref_glomap.so is glomap/scene/view_graph.cc and glomap/controllers/track_establishment.cc compiled from
/root/reference, unmodified, against the stand-in scene types of oracle/ref_shim/ (recipe: `make -C oracle ref`; flat entry
points: oracle/ref_glue.cc).  These two translation units are pure container / integer logic, which is why they — unlike the
Ceres-backed estimators — can be built in this image.  What is pinned here:

  ViewGraph::KeepLargestConnectedComponents     view_graph.cc:56-97       <->  oracle.tracks.keep_largest_connected_component[_literal]
  TrackEngine::EstablishFullTracks              track_establishment.cc:5-152  <->  oracle.tracks.establish_full_tracks[_literal]
  TrackEngine::FindTracksForProblem             track_establishment.cc:154-227 <->  oracle.tracks.find_tracks_for_problem[_literal]

csrc/tracks.hip is compared bit for bit with oracle/tracks.py on the GPU (tests/test_tracks.py, test_fullsize_gpu.py), so this
chain ends at reference code.  Third-party boundary that remains: colmap::UnionFind (un-vendored) is restated in
oracle/ref_shim/colmap/math/union_find.h; it decides which member names a track (the reference's track ids), not the tracks."""
import numpy as np
import pytest

from glomap_amd import synthetic
from oracle import ref, tracks as ot

pytestmark = pytest.mark.skipif(ref.load() is None, reason="neither /root/reference nor a prebuilt oracle/_ref/libref_glomap.so")


def _random_view_graph(rng, num_frames, images_per_frame, comps, p_invalid):
    """Frames split into components of the given sizes (distinct: no tie for the largest), random spanning edges + extras
    inside each component, a share of extra pairs already invalid; images of one frame share it (rigs)."""
    assert sum(comps) <= num_frames and len(set(comps)) == len(comps)
    frame_of_image = np.repeat(np.arange(num_frames), images_per_frame).astype(np.int32)
    images_of = [np.nonzero(frame_of_image == f)[0] for f in range(num_frames)]
    perm = rng.permutation(num_frames)
    e1, e2, valid = [], [], []
    start = 0
    for size in comps:
        members = perm[start : start + size]
        start += size
        for k in range(1, size):  # a random spanning tree of the component
            a, b = members[k], members[rng.integers(0, k)]
            e1.append(rng.choice(images_of[a]))
            e2.append(rng.choice(images_of[b]))
            valid.append(1)
        for _ in range(2 * size):  # extra pairs inside the component, some invalid
            a, b = rng.choice(members, 2, replace=True)
            if a == b:
                continue
            e1.append(rng.choice(images_of[a]))
            e2.append(rng.choice(images_of[b]))
            valid.append(0 if rng.random() < p_invalid else 1)
    # invalid pairs BETWEEN components must not connect anything
    for _ in range(10):
        a, b = rng.integers(0, num_frames, 2)
        e1.append(rng.choice(images_of[a]))
        e2.append(rng.choice(images_of[b]))
        valid.append(0)
    order = rng.permutation(len(e1))
    return (frame_of_image, np.array(e1, np.int32)[order], np.array(e2, np.int32)[order], np.array(valid, np.uint8)[order])


@pytest.mark.parametrize("seed", range(6))
def test_keep_largest_connected_components_equals_the_reference(seed):
    rng = np.random.default_rng(seed)
    num_frames = 60
    images_per_frame = 1 if seed % 2 == 0 else int(rng.integers(2, 4))
    comps = [27, 14, 9, 5, 1][: int(rng.integers(2, 6))]
    imf, p1, p2, pv = _random_view_graph(rng, num_frames, images_per_frame, comps, p_invalid=0.3)
    reg_r, pv_r, n_r = ref.keep_largest_connected_components(len(imf), imf, num_frames, p1, p2, pv)
    per_frame = np.bincount(imf, minlength=num_frames)
    for fn in (ot.keep_largest_connected_component_literal, ot.keep_largest_connected_component):
        reg_o, pv_o, n_o = fn(num_frames, imf[p1], imf[p2], pv, node_num_images=per_frame)
        assert n_o == n_r == comps[0] * images_per_frame
        assert np.array_equal(reg_o, reg_r) and np.array_equal(pv_o, pv_r)
    # a pair that was valid and joins two registered frames stays valid, everything else is invalid now (view_graph.cc:86-91)
    assert np.array_equal(pv_r, pv.astype(bool) & reg_r[imf[p1]] & reg_r[imf[p2]])


def test_keep_largest_connected_components_without_a_valid_pair():
    imf = np.arange(5, dtype=np.int32)
    p1, p2, pv = np.array([0, 1], np.int32), np.array([1, 2], np.int32), np.zeros(2, np.uint8)
    reg_r, pv_r, n_r = ref.keep_largest_connected_components(5, imf, 5, p1, p2, pv)
    reg_o, pv_o, n_o = ot.keep_largest_connected_component_literal(5, p1, p2, pv)
    assert n_r == n_o == 0 and reg_o is None and not pv_r.any()  # the reference returns before it touches a flag (:70)


def _args(g):
    return (g["pair_image1"], g["pair_image2"], g["pair_valid"], g["pair_offset"], g["match_feat1"], g["match_feat2"],
            g["feat_offset"], g["feat_xy"])


def _as_sets(tracks):
    return {frozenset(o) for o in tracks.values() if o}


@pytest.mark.parametrize("seed", range(4))
def test_establish_full_tracks_equals_the_reference(seed):
    """Same partition into tracks, same observations per kept track, same number of tracks discarded by the same-image
    consistency test — against the literal restatement AND the vectorised oracle the GPU path is pinned to."""
    g = synthetic.make_match_graph(60, 700, seed=seed, false_match_frac=0.02, twin_frac=0.02)
    tr_r, disc_r = ref.establish_full_tracks(*_args(g))
    tr_l, disc_l, members = ot.establish_full_tracks_literal(*_args(g))
    assert disc_r == disc_l > 0 and len(tr_r) == len(tr_l)
    assert _as_sets(tr_r) == _as_sets(tr_l)
    # the reference's track ids: union-find roots — the literal restatement reproduces them for the same pair order only
    # when it walks the pairs like the reference's unordered_map does; what must agree is the canonical form
    tid, off, img, ft, disc_v = ot.establish_full_tracks(*_args(g))
    assert disc_v == disc_r and len(tid) == len(tr_r)
    kept = {frozenset(zip(img[off[t] : off[t + 1]].tolist(), ft[off[t] : off[t + 1]].tolist())) for t in range(len(tid)) if off[t + 1] > off[t]}
    assert kept == _as_sets(tr_r)
    # canonical id of a kept track = its smallest member (image << 32 | feature): DESIGN.md 4.7 / INTEGRATION.md
    by_min = {min((i << 32) | f for i, f in o): o for o in tr_r.values() if o}
    for t in range(len(tid)):
        if off[t + 1] > off[t]:
            assert int(tid[t]) in by_min


@pytest.mark.parametrize("kw", [dict(), dict(min_num_tracks_per_view=5), dict(min_num_tracks_per_view=0),
                                dict(min_num_tracks_per_view=20, max_num_tracks=30), dict(max_num_tracks=10), dict(max_num_tracks=0),
                                dict(min_num_view_per_track=2, max_num_view_per_track=6), dict(min_num_tracks_per_view=3, min_num_view_per_track=4)])
def test_find_tracks_for_problem_equals_the_reference(kw):
    """The greedy selection on the oracle's canonical full tracks (ids = smallest member): the reference's own loop must pick
    the same tracks and keep the same observations as both oracle forms, for every option combination of tests/test_tracks.py."""
    g = synthetic.make_match_graph(50, 600, seed=3)
    tid, off, img, ft, _ = ot.establish_full_tracks(*_args(g))
    reg = np.ones(50, np.uint8)
    reg[::7] = 0
    sel_r, kept_r, n_r = ref.find_tracks_for_problem(50, reg, tid, off, img, ft, **kw)
    for fn in (ot.find_tracks_for_problem_literal, ot.find_tracks_for_problem):
        s_tid, s_off, s_img, s_ft = fn(tid, off, img, ft, reg, **kw)
        assert len(s_tid) == n_r == int(sel_r.sum())
        assert set(s_tid.tolist()) == set(tid[sel_r].tolist())
        # observations of the selected tracks: the registered-image subsequence of the full track
        want = {}
        for t in np.nonzero(sel_r)[0]:
            k = np.arange(off[t], off[t + 1])[kept_r[off[t] : off[t + 1]]]
            want[int(tid[t])] = list(zip(img[k].tolist(), ft[k].tolist()))
        got = {int(s_tid[j]): list(zip(s_img[s_off[j] : s_off[j + 1]].tolist(), s_ft[s_off[j] : s_off[j + 1]].tolist())) for j in range(len(s_tid))}
        assert got == want
