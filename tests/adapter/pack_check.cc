// Host-only check of the adapter's numbering (include/gsfm_glomap_adapter.hpp: FrameIndex::AddSorted, detail::PackTracks): frames
// and tracks are numbered by ascending id whatever order the caller's hash maps iterate in, and the walk / draw orders that
// are handed to the library reproduce that iteration order.  No GPU, no libgsfm call.  Prints "PACK OK" on success.
#include <cstdio>
#include <random>

#include "gsfm_glomap_adapter.hpp"

using namespace glomap;

int main() {
  std::mt19937 rng(3);
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  std::unordered_map<track_t, Track> tracks;
  const int N = 57, P = 400;
  std::vector<int> ids(N);
  for (int i = 0; i < N; ++i) ids[i] = 1000 - 13 * i;  // descending, gaps: neither insertion nor hash order is ascending
  for (int i = 0; i < N; ++i) {
    Frame fr;
    fr.is_registered = true;
    frames[ids[i]] = fr;
  }
  for (int i = 0; i < N; ++i) {
    Image im;
    im.image_id = ids[i];
    im.frame_id = ids[i];
    im.camera_id = 1;
    im.features_undist.assign(64, mock_eigen::Vector3d(0.0, 0.0, 1.0));
    images[ids[i]] = im;
  }
  for (auto& [id, im] : images) im.frame_ptr = &frames[im.frame_id];
  for (int p = 0; p < P; ++p) {
    Track tr;
    const int len = 2 + (int)(rng() % 5);  // some below min_views = 3
    for (int k = 0; k < len; ++k) tr.observations.emplace_back(ids[rng() % N], (feature_t)(rng() % 64));
    tracks[(track_t)(7 * (P - p) + 1)] = tr;
  }
  // frames
  gsfm_glomap::detail::FrameIndex fidx;
  std::vector<frame_t> walk;
  for (auto& [fid, fr] : frames) walk.push_back(fid);
  const std::vector<int32_t> cam_order = fidx.AddSorted(walk);
  if ((int)fidx.ids.size() != N) return std::printf("frame count\n"), 1;
  for (int i = 1; i < N; ++i)
    if (!(fidx.ids[i - 1] < fidx.ids[i])) return std::printf("frames not ascending\n"), 1;
  for (size_t i = 0; i < walk.size(); ++i)
    if (fidx.ids[cam_order[i]] != walk[i]) return std::printf("cam_draw_order does not reproduce the walk\n"), 1;
  // tracks
  auto keep = [](const Image& im, uint32_t) { return im.IsRegistered(); };
  std::vector<int32_t> draw;
  gsfm_glomap::detail::TrackPack tp = gsfm_glomap::detail::PackTracks(images, tracks, fidx, keep, 3, /*keep_empty=*/true, &draw);
  size_t expect = 0;
  for (auto& [tid, tr] : tracks) expect += tr.observations.size() >= 3 ? 1 : 0;
  if (tp.track_ids.size() != expect || draw.size() != expect) return std::printf("packed track count\n"), 1;
  for (size_t p = 1; p < tp.track_ids.size(); ++p)
    if (!(tp.track_ids[p - 1] < tp.track_ids[p])) return std::printf("tracks not ascending\n"), 1;
  size_t i = 0;
  std::vector<char> seen(expect, 0);
  for (auto& [tid, tr] : tracks) {  // the hash map's own walk over the kept tracks
    if (tr.observations.size() < 3) continue;
    if (tp.track_ids[draw[i]] != tid || seen[draw[i]]) return std::printf("pt_draw_order does not reproduce the walk\n"), 1;
    seen[draw[i]] = 1;
    ++i;
  }
  // observations: every packed track carries its own observations, cameras as dense indices of ascending frame ids
  for (size_t p = 0; p < tp.track_ids.size(); ++p) {
    const Track& tr = tracks.at(tp.track_ids[p]);
    if ((size_t)(tp.pt_offset[p + 1] - tp.pt_offset[p]) != tr.observations.size()) return std::printf("observation count\n"), 1;
    for (size_t k = 0; k < tr.observations.size(); ++k)
      if (fidx.ids[tp.obs_cam[tp.pt_offset[p] + k]] != (frame_t)tr.observations[k].first) return std::printf("obs_cam\n"), 1;
  }
  std::printf("PACK OK %d frames %zu tracks\n", N, tp.track_ids.size());
  return 0;
}
