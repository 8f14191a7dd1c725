// ORACLE (test infrastructure only).  orc_ba.cc compiled a second time with intrinsics blocks of 16 doubles: FULL_OPENCV,
// THIN_PRISM_FISHEYE (12 parameters) and RAD_TAN_THIN_PRISM_FISHEYE (16) — colmap/sensor/models.h, restated as in oracle/ba.py.
// Entry point orc_ba_solve_wide (same arguments as orc_ba_solve; intr_params_inout is [K][16]).
#define ORC_BA_MAXP 16
#define ORC_BA_ENTRY orc_ba_solve_wide
#include "orc_ba.cc"
