// ref_glue.cc — flat C entry points around the REFERENCE'S OWN view_graph.cc and track_establishment.cc (compiled from
// /root/reference by `make -C oracle ref` against the stand-in types of oracle/ref_shim/): test infrastructure that pins
// oracle/tracks.py to reference code (tests/test_oracle_ref.py).  Builds the reference's containers from flat arrays, calls
//   ViewGraph::KeepLargestConnectedComponents      glomap/scene/view_graph.cc:56-97
//   TrackEngine::EstablishFullTracks               glomap/controllers/track_establishment.cc:5-152
//   TrackEngine::FindTracksForProblem              glomap/controllers/track_establishment.cc:154-227
//   TrackFilter::FilterTracksByReprojection / ByAngle / FilterTrackTriangulationAngle   glomap/processors/track_filter.cc:7-127
//   NormalizeReconstruction                        glomap/processors/reconstruction_normalizer.cc:5-85
//   RelPoseFilter::FilterRotations                 glomap/processors/relpose_filter.cc:7-33
// and flattens the results.  Images are indices 0..N-1 (= image ids), frames indices 0..F-1.
#include <algorithm>
#include <cstring>
#include <sstream>

#include "glomap/controllers/track_establishment.h"
#include "glomap/processors/reconstruction_normalizer.h"
#include "glomap/processors/relpose_filter.h"
#include "glomap/processors/track_filter.h"
#include "glomap/scene/types_sfm.h"

using namespace glomap;

namespace {
struct Scene {
  std::unordered_map<frame_t, Frame> frames;
  std::unordered_map<image_t, Image> images;
  ViewGraph vg;
};

void make_images(Scene& s, int num_images, const int32_t* image_frame, const uint8_t* frame_registered, int num_frames) {
  s.frames.reserve(num_frames);
  for (int f = 0; f < num_frames; ++f) s.frames[f].is_registered = frame_registered ? frame_registered[f] != 0 : true;
  for (int i = 0; i < num_images; ++i) {
    Image im;
    im.image_id = i;
    im.frame_id = image_frame ? image_frame[i] : i;
    s.images.emplace(i, im);
  }
  for (auto& [id, im] : s.images) im.frame_ptr = &s.frames.at(im.frame_id);  // (after the maps stopped growing)
}

struct QuietCout {  // the reference prints progress to std::cout
  std::streambuf* old;
  std::ostringstream sink;
  QuietCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~QuietCout() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

// Returns the reference's return value (images in the largest component); frame_registered_out [F], pair_valid_inout [E].
int ref_keep_largest_connected_components(int num_images, const int32_t* image_frame, int num_frames, long num_pairs,
                                          const int32_t* pair_image1, const int32_t* pair_image2, uint8_t* pair_valid_inout,
                                          uint8_t* frame_registered_out) {
  Scene s;
  make_images(s, num_images, image_frame, nullptr, num_frames);
  for (long e = 0; e < num_pairs; ++e) {
    ImagePair p;
    p.image_id1 = pair_image1[e];
    p.image_id2 = pair_image2[e];
    p.is_valid = pair_valid_inout[e] != 0;
    s.vg.image_pairs.emplace(static_cast<image_pair_t>(e), p);
  }
  const int n = s.vg.KeepLargestConnectedComponents(s.frames, s.images);
  for (int f = 0; f < num_frames; ++f) frame_registered_out[f] = s.frames.at(f).is_registered ? 1 : 0;
  for (long e = 0; e < num_pairs; ++e) pair_valid_inout[e] = s.vg.image_pairs.at(static_cast<image_pair_t>(e)).is_valid ? 1 : 0;
  return n;
}

// Full tracks of the match graph (every match is an inlier).  Output, caller-allocated with capacity = number of matched
// features: track_id_out [T] (the reference's ids: union-find roots), track_len_out [T] (0: discarded, the reference keeps
// the empty Track), members_out = for every track its member global ids (image << 32 | feature), concatenated in track
// order, member_off_out [T + 1]; obs_out = the kept observations as global ids, concatenated, obs_off_out [T + 1].
// Returns T, or -1 when a buffer is too small.
long ref_establish_full_tracks(int num_images, const long* feat_offset, const double* feat_xy, long num_pairs,
                               const int32_t* pair_image1, const int32_t* pair_image2, const uint8_t* pair_valid,
                               const long* pair_offset, const uint32_t* match_feat1, const uint32_t* match_feat2,
                               double thres_inconsistency, long cap_tracks, long cap_items, uint64_t* track_id_out,
                               long* obs_off_out, uint64_t* obs_out, long* discarded_out) {
  QuietCout quiet;
  Scene s;
  make_images(s, num_images, nullptr, nullptr, num_images);
  for (auto& [id, im] : s.images) {
    const long f0 = feat_offset[id], f1 = feat_offset[id + 1];
    im.features.reserve(f1 - f0);
    for (long f = f0; f < f1; ++f) im.features.emplace_back(feat_xy[2 * f], feat_xy[2 * f + 1]);
  }
  for (long e = 0; e < num_pairs; ++e) {
    ImagePair p;
    p.image_id1 = pair_image1[e];
    p.image_id2 = pair_image2[e];
    p.is_valid = pair_valid ? pair_valid[e] != 0 : true;
    const long m0 = pair_offset[e], m1 = pair_offset[e + 1];
    p.matches.d.reserve(2 * (m1 - m0));
    for (long m = m0; m < m1; ++m) {
      p.matches.d.push_back(static_cast<int>(match_feat1[m]));
      p.matches.d.push_back(static_cast<int>(match_feat2[m]));
      p.inliers.push_back(static_cast<int>(m - m0));
    }
    s.vg.image_pairs.emplace(static_cast<image_pair_t>(e), std::move(p));
  }
  TrackEstablishmentOptions opt;
  opt.thres_inconsistency = thres_inconsistency;
  TrackEngine engine(s.vg, s.images, opt);
  std::unordered_map<track_t, Track> tracks;
  engine.EstablishFullTracks(tracks);
  const long T = static_cast<long>(tracks.size());
  if (T > cap_tracks) return -1;
  std::vector<track_t> ids;
  ids.reserve(T);
  for (auto& [id, tr] : tracks) ids.push_back(id);
  std::sort(ids.begin(), ids.end());
  long n = 0, disc = 0;
  for (long t = 0; t < T; ++t) {
    const Track& tr = tracks.at(ids[t]);
    track_id_out[t] = ids[t];
    obs_off_out[t] = n;
    if (tr.observations.empty()) ++disc;
    if (n + static_cast<long>(tr.observations.size()) > cap_items) return -1;
    for (const auto& [im, ft] : tr.observations) obs_out[n++] = (static_cast<uint64_t>(im) << 32) | ft;
  }
  obs_off_out[T] = n;
  *discarded_out = disc;
  return T;
}

// FindTracksForProblem on the given full tracks (ids, CSR observations as (image, feature)).  selected_out [T]: 1 = the
// reference put the track into tracks_selected; kept_obs_out [M]: 1 = that observation is in the selected track
// (observations of unregistered images are dropped).  Returns the reference's return value.
long ref_find_tracks_for_problem(int num_images, const uint8_t* image_registered, long num_tracks, const uint64_t* track_id,
                                 const long* track_offset, const int32_t* obs_image, const uint32_t* obs_feature,
                                 int min_num_tracks_per_view, int min_num_view_per_track, int max_num_view_per_track,
                                 int max_num_tracks, uint8_t* selected_out, uint8_t* kept_obs_out) {
  QuietCout quiet;
  Scene s;
  make_images(s, num_images, nullptr, image_registered, num_images);
  TrackEstablishmentOptions opt;
  opt.min_num_tracks_per_view = min_num_tracks_per_view;
  opt.min_num_view_per_track = min_num_view_per_track;
  opt.max_num_view_per_track = max_num_view_per_track;
  opt.max_num_tracks = max_num_tracks;
  TrackEngine engine(s.vg, s.images, opt);
  std::unordered_map<track_t, Track> full, sel;
  std::unordered_map<track_t, long> index;
  for (long t = 0; t < num_tracks; ++t) {
    Track tr;
    tr.track_id = track_id[t];
    for (long k = track_offset[t]; k < track_offset[t + 1]; ++k) tr.observations.emplace_back(obs_image[k], obs_feature[k]);
    full.emplace(track_id[t], std::move(tr));
    index.emplace(track_id[t], t);
  }
  const long n = static_cast<long>(engine.FindTracksForProblem(full, sel));
  std::memset(selected_out, 0, static_cast<size_t>(num_tracks));
  std::memset(kept_obs_out, 0, static_cast<size_t>(track_offset[num_tracks]));
  for (auto& [id, tr] : sel) {
    const long t = index.at(id);
    selected_out[t] = 1;
    // the selected track holds the observations of registered images, in the full track's order
    size_t j = 0;
    for (long k = track_offset[t]; k < track_offset[t + 1] && j < tr.observations.size(); ++k)
      if (tr.observations[j].first == static_cast<image_t>(obs_image[k]) && tr.observations[j].second == obs_feature[k]) {
        kept_obs_out[k] = 1;
        ++j;
      }
    if (j != tr.observations.size()) return -2;  // (cannot happen: the selection copies a subsequence)
  }
  return n;
}

}  // extern "C"

namespace {
// cameras with poses (trivial frames: image i = frame i = camera i), tracks from the flat CSR; returns the obs -> slot map
void make_posed_scene(Scene& s, std::unordered_map<camera_t, Camera>& cameras, std::unordered_map<track_t, Track>& tracks, int num_cams,
                      const double* cam_q, const double* cam_t, const uint8_t* cam_calibrated, const uint8_t* cam_registered,
                      long num_pts, const long* pt_offset, const int32_t* obs_cam, const double* obs_undist, const double* pt_xyz) {
  make_images(s, num_cams, nullptr, cam_registered, num_cams);
  for (int n = 0; n < num_cams; ++n) {
    Frame& f = s.frames.at(n);
    f.has_pose = true;
    f.rig_from_world.rotation = Eigen::Quaterniond(cam_q[4 * n], cam_q[4 * n + 1], cam_q[4 * n + 2], cam_q[4 * n + 3]);
    f.rig_from_world.translation = Eigen::Vector3d(cam_t[3 * n], cam_t[3 * n + 1], cam_t[3 * n + 2]);
    s.images.at(n).camera_id = n;
    cameras[n].has_prior_focal_length = cam_calibrated ? cam_calibrated[n] != 0 : true;
  }
  for (long p = 0; p < num_pts; ++p) {
    Track tr;
    tr.track_id = p;
    tr.xyz = Eigen::Vector3d(pt_xyz[3 * p], pt_xyz[3 * p + 1], pt_xyz[3 * p + 2]);
    for (long k = pt_offset[p]; k < pt_offset[p + 1]; ++k) {
      Image& im = s.images.at(obs_cam[k]);
      // the observation's feature id = its slot in the image's feature list (unique per observation: the keep mask is per slot)
      const feature_t fid = static_cast<feature_t>(im.features_undist.size());
      if (obs_undist) im.features_undist.emplace_back(obs_undist[3 * k], obs_undist[3 * k + 1], obs_undist[3 * k + 2]);
      else im.features_undist.emplace_back(0.0, 0.0, 1.0);
      tr.observations.emplace_back(static_cast<image_t>(obs_cam[k]), fid);
    }
    tracks.emplace(static_cast<track_t>(p), std::move(tr));
  }
}
}  // namespace

extern "C" {

// mode 0: FilterTracksByReprojection (normalised image coordinates), 1: FilterTracksByAngle, 2: FilterTrackTriangulationAngle.
// obs_keep_out [M]: 1 = the observation is still in its track afterwards.  Returns the reference's return value (tracks changed).
int ref_filter_tracks(int mode, int num_cams, const double* cam_q, const double* cam_t, const uint8_t* cam_calibrated, long num_pts,
                      const long* pt_offset, const int32_t* obs_cam, const double* obs_undist, const double* pt_xyz, double threshold,
                      uint8_t* obs_keep_out) {
  QuietCout quiet;
  Scene s;
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<track_t, Track> tracks;
  make_posed_scene(s, cameras, tracks, num_cams, cam_q, cam_t, cam_calibrated, nullptr, num_pts, pt_offset, obs_cam, obs_undist, pt_xyz);
  // remember every observation's (image, feature) to find it again afterwards
  std::vector<std::pair<image_t, feature_t>> key(static_cast<size_t>(pt_offset[num_pts]));
  for (long p = 0; p < num_pts; ++p) {
    const Track& tr = tracks.at(p);
    for (long k = pt_offset[p]; k < pt_offset[p + 1]; ++k) key[k] = tr.observations[k - pt_offset[p]];
  }
  int counter = 0;
  if (mode == 0) counter = TrackFilter::FilterTracksByReprojection(s.vg, cameras, s.images, tracks, threshold, true);
  else if (mode == 1) counter = TrackFilter::FilterTracksByAngle(s.vg, cameras, s.images, tracks, threshold);
  else counter = TrackFilter::FilterTrackTriangulationAngle(s.vg, s.images, tracks, threshold);
  for (long p = 0; p < num_pts; ++p) {
    const Track& tr = tracks.at(p);
    size_t j = 0;  // the filters keep a subsequence
    for (long k = pt_offset[p]; k < pt_offset[p + 1]; ++k) {
      const bool kept = j < tr.observations.size() && tr.observations[j] == key[k];
      obs_keep_out[k] = kept ? 1 : 0;
      if (kept) ++j;
    }
    if (j != tr.observations.size()) return -1;
  }
  return counter;
}

// NormalizeReconstruction: cam_t [N][3] and pt_xyz [P][3] transformed in place, sim3_out = {scale, tx, ty, tz}.
int ref_normalize_reconstruction(int num_cams, const double* cam_q, double* cam_t_inout, const uint8_t* cam_registered, long num_pts,
                                 double* pt_xyz_inout, int fixed_scale, double extent, double p0, double p1, double* sim3_out) {
  QuietCout quiet;
  Scene s;
  std::unordered_map<camera_t, Camera> cameras;
  std::unordered_map<track_t, Track> tracks;
  std::unordered_map<rig_t, Rig> rigs;
  std::vector<long> off(static_cast<size_t>(num_pts) + 1, 0);
  make_posed_scene(s, cameras, tracks, num_cams, cam_q, cam_t_inout, nullptr, cam_registered, num_pts, off.data(), nullptr, nullptr, pt_xyz_inout);
  const colmap::Sim3d t = NormalizeReconstruction(rigs, cameras, s.frames, s.images, tracks, fixed_scale != 0, extent, p0, p1);
  for (int n = 0; n < num_cams; ++n)
    for (int j = 0; j < 3; ++j) cam_t_inout[3 * n + j] = s.frames.at(n).rig_from_world.translation(j);
  for (long p = 0; p < num_pts; ++p)
    for (int j = 0; j < 3; ++j) pt_xyz_inout[3 * p + j] = tracks.at(p).xyz(j);
  sim3_out[0] = t.scale;
  for (int j = 0; j < 3; ++j) sim3_out[1 + j] = t.translation(j);
  return 0;
}

}  // extern "C"

// RelPoseFilter::FilterRotations: node_q [N][4] cam_from_world rotations (w, x, y, z), edge_q [E][4] cam2_from_cam1;
// edge_valid_inout [E].  Returns the number of pairs the reference invalidated.
extern "C" long ref_filter_rotations(int num_nodes, const double* node_q, const uint8_t* node_registered, long num_edges, const int32_t* edge_i,
                                     const int32_t* edge_j, const double* edge_q, double max_angle_deg, uint8_t* edge_valid_inout) {
  QuietCout quiet;
  Scene s;
  make_images(s, num_nodes, nullptr, node_registered, num_nodes);
  for (int n = 0; n < num_nodes; ++n) {
    Frame& f = s.frames.at(n);
    f.has_pose = true;
    f.rig_from_world.rotation = Eigen::Quaterniond(node_q[4 * n], node_q[4 * n + 1], node_q[4 * n + 2], node_q[4 * n + 3]);
  }
  long before = 0;
  for (long e = 0; e < num_edges; ++e) {
    ImagePair p;
    p.image_id1 = edge_i[e];
    p.image_id2 = edge_j[e];
    p.is_valid = edge_valid_inout[e] != 0;
    before += p.is_valid ? 1 : 0;
    p.cam2_from_cam1.rotation = Eigen::Quaterniond(edge_q[4 * e], edge_q[4 * e + 1], edge_q[4 * e + 2], edge_q[4 * e + 3]);
    s.vg.image_pairs.emplace(static_cast<image_pair_t>(e), p);
  }
  RelPoseFilter::FilterRotations(s.vg, s.images, max_angle_deg);
  long after = 0;
  for (long e = 0; e < num_edges; ++e) {
    edge_valid_inout[e] = s.vg.image_pairs.at(static_cast<image_pair_t>(e)).is_valid ? 1 : 0;
    after += edge_valid_inout[e];
  }
  return before - after;
}
