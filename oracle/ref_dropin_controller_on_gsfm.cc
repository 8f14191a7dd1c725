// The reference's rotation-averaging CONTROLLER (glomap/controllers/rotation_averager.cc, included from /root/reference,
// unmodified) compiled against libgsfm's adapter: oracle/ref_shim_dropin/ comes first on the include path and makes
// glomap::RotationEstimator the adapter class.  The entry point is renamed so that this object can sit next to the plain
// compilation of the same file (reference controller + reference estimator) in one library: oracle/ref_glue_dropin.cc.
#define SolveRotationAveraging SolveRotationAveragingOnGsfm
#include REF_ROTATION_AVERAGER_CC
