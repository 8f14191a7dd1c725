// THE DROP-IN AT THE TOP OF THE HOT PATH: the reference's controller glomap/controllers/global_mapper.cc — GlobalMapper::Solve,
// the caller of all three estimators (global_mapper.cc:92-110 rotation averaging, :152-160 global positioning, :201-223 the
// bundle adjustment rounds) — included from /root/reference, UNMODIFIED, with the estimator names it instantiates switched at
// their use sites to the adapter classes of include/gsfm_glomap_adapter.hpp.  This is INTEGRATION.md section 2's diff, applied
// by the preprocessor instead of by hand:
//     GlobalPositioner gp_engine(options_.opt_gp);      ->  gsfm_glomap::GlobalPositioner gp_engine(options_.opt_gp);
//     BundleAdjuster ba_engine(options_.opt_ba);        ->  gsfm_glomap::BundleAdjuster ba_engine(options_.opt_ba);
//     SolveRotationAveraging(...)                       ->  the reference's rotation_averager.cc on gsfm_glomap::RotationEstimator
//                                                           (oracle/ref_dropin_controller_on_gsfm.cc, round 5)
//     UndistortImages(cameras, images, clean)           ->  gsfm_glomap::UndistortImages (one device sweep)
// and, with -DREF_MAPPER_PROCESSORS_ON_GSFM, the processors between the solves as well (TrackFilter, NormalizeReconstruction,
// RelPoseFilter: filters.hip).  Track establishment stays the reference's in both (its track ids are union-find roots, and the
// ids decide the hash map's walk = the draw order of GlobalPositioner's random start; gsfm_glomap::TrackEngine names a track by
// its smallest member — same tracks, other ids, another start).
// The class is renamed so that this object can sit next to the plain compilation of the same file (the reference's controller
// on the reference's estimators) in one library, oracle/_ref/libref_dropin_mapper.so (oracle/ref_glue_mapper.cc runs either).
#ifndef REF_MAPPER_CLASS
#define REF_MAPPER_CLASS GlobalMapperOnGsfm
#endif
#define GlobalMapper REF_MAPPER_CLASS
#define SolveRotationAveraging SolveRotationAveragingOnGsfm
// every header global_mapper.cc names, under the reference's own names (include guards make its own #includes no-ops)
#include "glomap/controllers/global_mapper.h"
#include "glomap/controllers/rotation_averager.h"
#include "glomap/io/colmap_converter.h"
#include "glomap/processors/image_pair_inliers.h"
#include "glomap/processors/image_undistorter.h"
#include "glomap/processors/reconstruction_normalizer.h"
#include "glomap/processors/reconstruction_pruning.h"
#include "glomap/processors/relpose_filter.h"
#include "glomap/processors/track_filter.h"
#include "glomap/processors/view_graph_manipulation.h"

#include <colmap/util/file.h>
#include <colmap/util/timer.h>

#include "gsfm_glomap_adapter.hpp"

#define GlobalPositioner gsfm_glomap::GlobalPositioner
#define BundleAdjuster gsfm_glomap::BundleAdjuster
#define UndistortImages gsfm_glomap::UndistortImages
#ifdef REF_MAPPER_PROCESSORS_ON_GSFM
#define TrackFilter gsfm_glomap::TrackFilter
#define NormalizeReconstruction gsfm_glomap::NormalizeReconstruction
#define RelPoseFilter gsfm_glomap::RelPoseFilter
#endif
#ifdef REF_MAPPER_TRACKS_ON_GSFM
// Track establishment as well: the same tracks under OTHER ids (smallest member instead of union-find root), hence another walk of
// the `tracks` map and another draw of GlobalPositioner's random start — an equally valid run, compared with ground truth and not
// with the all-reference build.
#define TrackEngine gsfm_glomap::TrackEngine
#endif
#include REF_GLOBAL_MAPPER_CC
