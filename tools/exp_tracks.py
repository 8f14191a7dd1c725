"""Timing of track establishment / selection on device-resident inputs (usage: python tools/exp_tracks.py [n_images n_tracks])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from glomap_amd import _lib, synthetic
from glomap_amd.tracks import MatchGraph, TrackEngine, TrackEstablishmentOptions

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_trk = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
ctx = _lib.Context(-1)
t = time.time()
g = synthetic.make_match_graph(n_img, n_trk, seed=0)
print("gen %.1fs matches %d features %d pairs %d" % (time.time() - t, len(g["match_feat1"]), g["feat_offset"][-1], len(g["pair_image1"])), flush=True)
eng = TrackEngine(MatchGraph.from_dict(g).to_device(ctx), ctx=ctx)
reg = _lib.DeviceArray.from_numpy(ctx, np.ones(n_img, np.uint8))
ctx.profile_enable(True)
for it in range(4):
    t = time.perf_counter(); nt = eng.EstablishFullTracks(fetch=False); te = time.perf_counter() - t
    t = time.perf_counter(); ns = eng.FindTracksForProblem(reg, fetch=False); ts = time.perf_counter() - t
    eng.options = TrackEstablishmentOptions(min_num_tracks_per_view=200)
    t = time.perf_counter(); ns2 = eng.FindTracksForProblem(reg, fetch=False); ts2 = time.perf_counter() - t
    eng.options = TrackEstablishmentOptions()
    print("establish %.2f ms (%d tracks, %d discarded)  select %.2f ms (%d)  select(cap 200) %.2f ms (%d)" % (te * 1e3, nt, eng.num_discarded, ts * 1e3, ns, ts2 * 1e3, ns2), flush=True)
n, ms = ctx.profile_read(7)
M = len(g["match_feat1"])
print("k_uf_hook: %d launches avg %.1f us; %.1f Gmatch/s; algorithmic %.0f MB -> %.0f GB/s" % (n, ms / n * 1e3, M / (ms / n * 1e-3) / 1e9, (8 * M + 16 * M) / 1e6, (8 * M + 16 * M) / (ms / n * 1e-3) / 1e9))
