// ORACLE (test infrastructure only — never linked into or called by the product path).
//
// orc_gp.cc — multithreaded C++ CPU restatement of GLOMAP's global positioning, ONLY_POINTS with
// trivial rigs (the only mode `glomap mapper` accepts, glomap/controllers/global_mapper.cc:145-149).
// Same algorithm as oracle/gp.py (which it is cross-validated against on small problems), written
// so that it also runs at the full benchmark sizes and can be timed on all host cores:
//
//   residual      BATAPairwiseDirectionError, glomap/estimators/cost_function.h:15-41: r = v - s (X - c)
//   problem       AddTrackToProblem, global_positioning.cc:212-375 (tracks >= min_num_view_per_track :258;
//                 scales start at 1 :298-305, lower bound 1e-5 :373; Huber(0.1) for cameras with a prior
//                 focal length, ScaledLoss(Huber(0.1), 0.5) otherwise :242-255,313-316)
//   random init   global_positioning.cc:123-165,261-264: 100 * U(-1,1)^3 from std::mt19937(seed) and
//                 std::uniform_real_distribution<double>; draw order: constrained cameras by index, then
//                 used tracks by index (the reference's unordered_map order cannot be reproduced)
//   gauge         the first scale is constant, global_positioning.cc:484-489
//   ordering      scales, then points, then camera centres (global_positioning.cc:388-429)
//   solver        Ceres LM (orc_lm.hpp).  The scales (1x1) and the points (3x3) are eliminated in closed
//                 form — the step SPARSE_SCHUR computes — and the 3N camera system is solved by block-Jacobi
//                 PCG to 1e-14.
//
// parity unpinned: Ceres is un-vendored; compared through converged solutions.
#include <random>

#include "orc_lm.hpp"

namespace orc {

struct GpOptionsC {  // mirrors the python-side ctypes struct (oracle/cpu.py)
  int32_t max_num_iterations;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius, min_relative_decrease;
  double min_lm_diagonal, max_lm_diagonal;
  int32_t jacobi_scaling, max_num_consecutive_invalid_steps;
  double pcg_relative_tolerance;
  int32_t pcg_max_iterations, order, verbose, line_search;
  double thres_loss_function;
  int32_t generate_random_positions, generate_random_points, generate_scales;
  int32_t optimize_positions, optimize_points, optimize_scales;
  int32_t min_num_view_per_track;
  uint32_t seed;
};

struct GpReport {
  int32_t iterations, successful_steps, termination, usable;
  i64 linear_iterations;
  double initial_cost, final_cost, max_linear_residual, seconds_total, seconds_linear;
  int32_t threads, line_search_shrunk;
};

namespace {

struct Gp : LmProblem {
  i64 N, P, M;
  std::vector<int32_t> cam, pt;   // [M] track-major
  std::vector<i64> poff;          // [P+1]
  const double* v;                // [M][3] (compacted copy)
  std::vector<double> vbuf;
  std::vector<double> off;        // [M][3] known rigs: R_cw^T t_cam_from_rig of the observation's image (else empty)
  std::vector<uint8_t> cal;
  // unknown cam_from_rig translations (RigUnknownBATAPairwiseDirectionError, cost_function.h:90-136, gp.cc:354-368):
  // S centre blocks c_s behind the N frames in c / the reduced vector; observation k of such a sensor has sblk[k] >= 0 and
  // Rf[k] = R_rig_from_world of its frame:  d = X - c_frame - Rf^T c_s
  i64 S = 0;
  std::vector<int32_t> sblk;
  std::vector<double> Rf;             // [M][9]
  std::vector<int32_t> sobs, sown;
  OwnerLists bysens;
  OwnerLists bycam;
  // camera-to-camera constraints (constraint_type != ONLY_POINTS; AddCameraToCameraConstraints, gp.cc:167-210):
  // r_e = pv_e - s_e (c_j - c_i), a scale of its own per pair (ps), plain Huber; pair 0's scale is the constant one
  // (the pairs are added before the tracks: gp.cc:484-489), no observation's then (fixed_obs = -1)
  i64 E = 0;
  i64 fixed_obs = 0;
  std::vector<int32_t> pi, pj;
  std::vector<double> pv, ps, ps2, pw, pjs, pqa, pqb;
  Huber loss_pair;
  inline V3 pair_d(i64 e, const std::vector<double>& cc) const { return ld3(&cc[3 * (i64)pj[e]]) - ld3(&cc[3 * (i64)pi[e]]); }
  inline double mscale(i64 k) const { return k == fixed_obs ? 0.0 : ms; }
  inline double mpair(i64 e) const { return e == 0 ? 0.0 : ms; }
  bool defl_on = true;  // ORC_DEFLATE experiment: deflate the next reduced solve
  Huber loss_cal, loss_unc;
  double mc, mx, ms;  // 1 / 0: optimize_positions / points / scales
  double lm_lo, lm_hi;
  bool rev;
  double pcg_tol;
  int pcg_max;
  // state
  std::vector<double> c, X, s, c2, X2, s2;
  // linearisation
  std::vector<double> w;        // [M] rho'
  std::vector<double> gc, gX;   // gradients [3N], [3P]
  std::vector<double> hc, hx;   // diag of J^T J per camera / point (one value: same for the 3 components)
  std::vector<double> jc, jx, js;  // Jacobi scales
  bool have_scale = false;
  // per-step
  std::vector<double> qa, qb;   // per observation: Q_k = qa (I - qb d d^T)
  std::vector<double> Hinv;     // [P][6] inverse of H_pp (symmetric, xx xy xz yy yz zz)
  std::vector<double> tp;       // [P][3]
  std::vector<double> Minv;     // [N][9] block-Jacobi
  std::vector<double> dcam;     // [N] LM damping of the camera blocks
  // the step of the last step() (masks applied), for the projected line search of bounds-constrained programs
  std::vector<double> d_c, d_X, d_s, d_ps, gs_all, pgs_all;
  double slope0 = 0.0, dmax = 0.0;
  bool constrained() const override { return ms != 0.0 && M + E > 1; }  // a non-constant scale with its lower bound
  double step_slope() const override { return slope0; }
  double step_max_norm() const override { return dmax; }
  // x_t = Plus(x, t delta): scales projected on their lower bound (ParameterBlock::Plus)
  void point_at(double t, std::vector<double>& ct, std::vector<double>& Xt, std::vector<double>& st, std::vector<double>& pst) const {
    ct.resize(c.size());
    Xt.resize(X.size());
    st.resize(M);
    pst.resize(E);
    for (size_t i = 0; i < c.size(); ++i) ct[i] = c[i] + t * d_c[i];
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < 3 * P; ++i) Xt[i] = X[i] + t * d_X[i];
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) st[k] = std::max(s[k] + t * d_s[k], 1e-5);
    for (i64 e = 0; e < E; ++e) pst[e] = std::max(ps[e] + t * d_ps[e], 1e-5);
  }
  // cost and g(x_t) . delta: per residual block rho' r . (J delta), J delta = s e - ds d with e = dc_cam - dX the
  // direction's share of -(X - c) (LineSearchFunction::Evaluate: direction . gradient at the trial point)
  void ls_eval(double t, double* cost_out, double* slope_out) override {
    std::vector<double> ct, Xt, st, pst;
    point_at(t, ct, Xt, st, pst);
    std::vector<double> rho(M), sl(M);
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) {
      const V3 d = dvec(k, ct, Xt);
      const V3 r = ld3(v + 3 * k) - st[k] * d;
      double r0, r1;
      (cal[k] ? loss_cal : loss_unc).eval(dot(r, r), r0, r1);
      rho[k] = r0;
      const V3 e = zcam(k, d_c) - ld3(&d_X[3 * (i64)pt[k]]);
      const V3 jd = st[k] * e - d_s[k] * d;
      sl[k] = r1 * dot(jd, r);
    }
    double tot = chunked_sum(M, [&](i64 k) { return rho[k]; });
    double slope = chunked_sum(M, [&](i64 k) { return sl[k]; });
    for (i64 e = 0; e < E; ++e) {
      const V3 d = pair_d(e, ct);
      const V3 r = ld3(&pv[3 * e]) - pst[e] * d;
      double r0, r1;
      loss_pair.eval(dot(r, r), r0, r1);
      tot += r0;
      const V3 ee = ld3(&d_c[3 * (i64)pi[e]]) - ld3(&d_c[3 * (i64)pj[e]]);
      const V3 jd = pst[e] * ee - d_ps[e] * d;
      slope += r1 * dot(jd, r);
    }
    *cost_out = 0.5 * tot;
    *slope_out = slope;
  }
  void set_step_size(double t, double* cand_cost, double* step_norm) override {
    for (double& a : d_c) a *= t;
    for (double& a : d_X) a *= t;
    for (double& a : d_s) a *= t;
    for (double& a : d_ps) a *= t;
    make_candidate(cand_cost, step_norm);
  }
  // candidate = Plus(x, delta), its cost and |candidate - x| (ambient space, after the projection)
  void make_candidate(double* cand_cost, double* step_norm) {
    point_at(1.0, c2, X2, s2, ps2);
    double sn = chunked_sum(3 * (N + S), [&](i64 i) { const double d = c2[i] - c[i]; return d * d; });
    sn += chunked_sum(3 * P, [&](i64 i) { const double d = X2[i] - X[i]; return d * d; });
    sn += chunked_sum(M, [&](i64 k) { const double d = s2[k] - s[k]; return d * d; });
    double pair_sn = 0.0;
    for (i64 e = 0; e < E; ++e) pair_sn += (ps2[e] - ps[e]) * (ps2[e] - ps[e]);
    *step_norm = std::sqrt(sn + pair_sn);
    *cand_cost = cost_at(c2, X2, s2, ps2);
  }

  // d = X - c, or X - c_rig + t_rig for an image of a calibrated rig (RigBATAPairwiseDirectionError,
  // cost_function.h:49-82, with the rig scale constant at 1: global_positioning.cc:470-478)
  inline V3 dvec(i64 k, const std::vector<double>& cc, const std::vector<double>& XX) const {
    V3 d = ld3(&XX[3 * (i64)pt[k]]) - ld3(&cc[3 * (i64)cam[k]]);
    if (S > 0 && sblk[k] >= 0) d = d - rot_t(k, ld3(&cc[3 * (N + sblk[k])]));
    return off.empty() ? d : d + ld3(&off[3 * k]);
  }
  inline V3 rot_t(i64 k, const V3& a) const {  // Rf^T a
    const double* R = &Rf[9 * k];
    return V3{R[0] * a.x + R[3] * a.y + R[6] * a.z, R[1] * a.x + R[4] * a.y + R[7] * a.z, R[2] * a.x + R[5] * a.y + R[8] * a.z};
  }
  inline V3 rot(i64 k, const V3& a) const {  // Rf a
    const double* R = &Rf[9 * k];
    return V3{R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z, R[6] * a.x + R[7] * a.y + R[8] * a.z};
  }
  // the camera-side tangent an observation sees: z_frame + Rf^T z_sensor
  inline V3 zcam(i64 k, const std::vector<double>& z) const {
    V3 a = ld3(&z[3 * (i64)cam[k]]);
    if (S > 0 && sblk[k] >= 0) a = a + rot_t(k, ld3(&z[3 * (N + sblk[k])]));
    return a;
  }
  inline double damp(double h, double j, double radius) const {
    const double j2 = j * j;
    return std::min(std::max(j2 * h, lm_lo), lm_hi) / (radius * j2);
  }

  double cost_at(const std::vector<double>& cc, const std::vector<double>& XX, const std::vector<double>& ss,
                 const std::vector<double>& pss) const {
    const Gp* g = this;
    double tot = chunked_sum(M, [=, &cc, &XX, &ss](i64 k) {
      const V3 d = g->dvec(k, cc, XX);
      const V3 r = ld3(g->v + 3 * k) - ss[k] * d;
      double r0, r1;
      (g->cal[k] ? g->loss_cal : g->loss_unc).eval(dot(r, r), r0, r1);
      return r0;
    });
    for (i64 e = 0; e < E; ++e) {
      const V3 r = ld3(&pv[3 * e]) - pss[e] * pair_d(e, cc);
      double r0, r1;
      loss_pair.eval(dot(r, r), r0, r1);
      tot += r0;
    }
    return 0.5 * tot;
  }

  double linearize(double* gmax_out) override {
    w.resize(M);
    gc.assign(3 * (N + S), 0.0);
    gX.assign(3 * P, 0.0);
    hc.assign(N + S, 0.0);
    hx.assign(P, 0.0);
    std::vector<double> rho(M), gs(M);
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) {
      const V3 d = dvec(k, c, X);
      const V3 r = ld3(v + 3 * k) - s[k] * d;
      double r0, r1;
      (cal[k] ? loss_cal : loss_unc).eval(dot(r, r), r0, r1);
      rho[k] = r0;
      w[k] = r1;
      gs[k] = mscale(k) * (-r1 * dot(d, r));  // the first scale is constant
      // bounds-constrained program: the gradient test sees x - Plus(x, -g) (trust_region_minimizer.cc), i.e. the part
      // of -g the lower bound lets through
      if (gs[k] > 0.0 && s[k] - gs[k] < 1e-5) gs[k] = s[k] - 1e-5;
    }
    const double cost = 0.5 * chunked_sum(M, [&](i64 k) { return rho[k]; });
    // point side: g_X = -sum w s r, h_X = sum w s^2  (track-major, serial inside a track)
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double a[4] = {0, 0, 0, 0};
      auto body = [&](i64 k) {
        const V3 d = dvec(k, c, X);
        const V3 r = ld3(v + 3 * k) - s[k] * d;
        const double ws = w[k] * s[k];
        a[0] += ws * s[k];
        a[1] -= ws * r.x;
        a[2] -= ws * r.y;
        a[3] -= ws * r.z;
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      hx[p] = mx * a[0];
      gX[3 * p] = mx * a[1];
      gX[3 * p + 1] = mx * a[2];
      gX[3 * p + 2] = mx * a[3];
    }
    // camera side
    std::vector<double> acc(4 * N);
    bycam.reduce<4>(acc.data(), rev, [&](i64 k, double* a) {
      const V3 d = dvec(k, c, X);
      const V3 r = ld3(v + 3 * k) - s[k] * d;
      const double ws = w[k] * s[k];
      a[0] += ws * s[k];
      a[1] += ws * r.x;
      a[2] += ws * r.y;
      a[3] += ws * r.z;
    });
#pragma omp parallel for schedule(static)
    for (i64 n = 0; n < N; ++n) {
      hc[n] = mc * acc[4 * n];
      gc[3 * n] = mc * acc[4 * n + 1];
      gc[3 * n + 1] = mc * acc[4 * n + 2];
      gc[3 * n + 2] = mc * acc[4 * n + 3];
    }
    if (S > 0) {  // d r / d c_s = s Rf^T: same squared column norms, gradient rotated by Rf
      std::vector<double> accs(4 * S);
      bysens.reduce<4>(accs.data(), rev, [&](i64 e, double* a) {
        const i64 k = sobs[e];
        const V3 d = dvec(k, c, X);
        const V3 r = ld3(v + 3 * k) - s[k] * d;
        const double ws = w[k] * s[k];
        const V3 g = rot(k, ws * r);
        a[0] += ws * s[k];
        a[1] += g.x;
        a[2] += g.y;
        a[3] += g.z;
      });
      for (i64 sb = 0; sb < S; ++sb) {
        hc[N + sb] = mc * accs[4 * sb];
        for (int j = 0; j < 3; ++j) gc[3 * (N + sb) + j] = mc * accs[4 * sb + 1 + j];
      }
    }
    double pair_cost = 0.0, pair_gs = 0.0;
    pw.resize(E);
    for (i64 e = 0; e < E; ++e) {  // few pairs: serial, a fixed summation order
      const V3 d = pair_d(e, c);
      const V3 r = ld3(&pv[3 * e]) - ps[e] * d;
      double r0, r1;
      loss_pair.eval(dot(r, r), r0, r1);
      pw[e] = r1;
      pair_cost += r0;
      {
        double g = mpair(e) * (-r1 * dot(d, r));
        if (g > 0.0 && ps[e] - g < 1e-5) g = ps[e] - 1e-5;
        pair_gs = std::max(pair_gs, std::fabs(g));
      }
      const double ws = r1 * ps[e];
      // d r / d c_i = +s I, d r / d c_j = -s I
      hc[pi[e]] += mc * ws * ps[e];
      hc[pj[e]] += mc * ws * ps[e];
      for (int j = 0; j < 3; ++j) {
        const double g = mc * ws * (&r.x)[j];
        gc[3 * (i64)pi[e] + j] += g;
        gc[3 * (i64)pj[e] + j] -= g;
      }
    }
    double gmax = std::max(pair_gs, chunked_max(M, [&](i64 k) { return std::fabs(gs[k]); }));
    gmax = std::max(gmax, chunked_max(3 * (N + S), [&](i64 i) { return std::fabs(gc[i]); }));
    gmax = std::max(gmax, chunked_max(3 * P, [&](i64 i) { return std::fabs(gX[i]); }));
    *gmax_out = gmax;
    return cost + 0.5 * pair_cost;
  }

  void set_jacobi_scaling(bool enabled) override {
    jc.assign(N + S, 1.0);
    jx.assign(P, 1.0);
    js.assign(M, 1.0);
    if (enabled) {
      for (i64 n = 0; n < N + S; ++n) jc[n] = 1.0 / (1.0 + std::sqrt(hc[n]));
      for (i64 p = 0; p < P; ++p) jx[p] = 1.0 / (1.0 + std::sqrt(hx[p]));
#pragma omp parallel for schedule(static)
      for (i64 k = 0; k < M; ++k) {
        const V3 d = dvec(k, c, X);
        const double h = mscale(k) * w[k] * dot(d, d);
        js[k] = 1.0 / (1.0 + std::sqrt(h));
      }
    }
    pjs.assign(E, 1.0);
    if (enabled)
      for (i64 e = 0; e < E; ++e) {
        const V3 d = pair_d(e, c);
        pjs[e] = 1.0 / (1.0 + std::sqrt(mpair(e) * pw[e] * dot(d, d)));
      }
    have_scale = true;
  }

  // w_out = S z  (S = reduced camera system), two sweeps
  void apply(const std::vector<double>& z, std::vector<double>& out) {
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double a[3] = {0, 0, 0};
      auto body = [&](i64 k) {
        const V3 d = dvec(k, c, X);
        const V3 zc = zcam(k, z);
        const V3 q = qa[k] * (zc - (qb[k] * dot(d, zc)) * d);
        a[0] += q.x;
        a[1] += q.y;
        a[2] += q.z;
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      const double* H = &Hinv[6 * p];
      // t_p = mc mx Hpp^-1 sum Q z   (H_pc = -mc mx Q)
      const double f = mc * mx;
      tp[3 * p] = f * (H[0] * a[0] + H[1] * a[1] + H[2] * a[2]);
      tp[3 * p + 1] = f * (H[1] * a[0] + H[3] * a[1] + H[4] * a[2]);
      tp[3 * p + 2] = f * (H[2] * a[0] + H[4] * a[1] + H[5] * a[2]);
    }
    std::vector<double> acc(3 * N);
    bycam.reduce<3>(acc.data(), rev, [&](i64 k, double* a) {
      const V3 d = dvec(k, c, X);
      const V3 e = mc * zcam(k, z) - mx * ld3(&tp[3 * (i64)pt[k]]);
      const V3 q = qa[k] * (e - (qb[k] * dot(d, e)) * d);
      a[0] += q.x;
      a[1] += q.y;
      a[2] += q.z;
    });
    for (i64 e = 0; e < E; ++e) {  // Q_e (z_i - z_j) on i, minus that on j
      const V3 d = pair_d(e, c);
      const V3 zz = mc * (ld3(&z[3 * (i64)pi[e]]) - ld3(&z[3 * (i64)pj[e]]));
      const V3 q = pqa[e] * (zz - (pqb[e] * dot(d, zz)) * d);
      for (int j = 0; j < 3; ++j) {
        acc[3 * (i64)pi[e] + j] += (&q.x)[j];
        acc[3 * (i64)pj[e] + j] -= (&q.x)[j];
      }
    }
#pragma omp parallel for schedule(static)
    for (i64 n = 0; n < N; ++n)
      for (int j = 0; j < 3; ++j) out[3 * n + j] = mc * acc[3 * n + j] + dcam[n] * z[3 * n + j];
    if (S > 0) {
      std::vector<double> accs(3 * S);
      bysens.reduce<3>(accs.data(), rev, [&](i64 ee, double* a) {
        const i64 k = sobs[ee];
        const V3 d = dvec(k, c, X);
        const V3 e = mc * zcam(k, z) - mx * ld3(&tp[3 * (i64)pt[k]]);
        const V3 q = rot(k, qa[k] * (e - (qb[k] * dot(d, e)) * d));
        a[0] += q.x;
        a[1] += q.y;
        a[2] += q.z;
      });
      for (i64 sb = 0; sb < S; ++sb)
        for (int j = 0; j < 3; ++j) out[3 * (N + sb) + j] = mc * accs[3 * sb + j] + dcam[N + sb] * z[3 * (N + sb) + j];
    }
  }

  bool step(double radius, double* model_change, double* cand_cost, double* step_norm, double* x_norm, i64* lin,
            double* relres) override {
    qa.resize(M);
    qb.resize(M);
    std::vector<double> qg(3 * M);  // q_k = s w (r - beta d (d.r))
    std::vector<double> hss(M);
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) {
      const V3 d = dvec(k, c, X);
      const V3 r = ld3(v + 3 * k) - s[k] * d;
      const double m = mscale(k);
      const double h = m * w[k] * dot(d, d);
      const double ht = h + damp(h, js[k], radius);
      hss[k] = ht;
      const double beta = m * w[k] / ht;
      qa[k] = w[k] * s[k] * s[k];
      qb[k] = beta;
      st3(&qg[3 * k], (s[k] * w[k]) * (r - (beta * dot(d, r)) * d));
    }
    // points: H_pp = mx^2 sum Q + D_X, reduced gradient gX' = -mx sum q
    Hinv.resize(6 * P);
    tp.assign(3 * P, 0.0);
    std::vector<double> gXr(3 * P);
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      double g[3] = {0, 0, 0};
      auto body = [&](i64 k) {
        const V3 d = dvec(k, c, X);
        const double a = qa[k], ab = qa[k] * qb[k];
        H[0] += a - ab * d.x * d.x;
        H[1] += -ab * d.x * d.y;
        H[2] += -ab * d.x * d.z;
        H[4] += a - ab * d.y * d.y;
        H[5] += -ab * d.y * d.z;
        H[8] += a - ab * d.z * d.z;
        g[0] -= qg[3 * k];
        g[1] -= qg[3 * k + 1];
        g[2] -= qg[3 * k + 2];
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      const double dx = damp(hx[p], jx[p], radius);
      H[0] = mx * H[0] + dx;
      H[4] = mx * H[4] + dx;
      H[8] = mx * H[8] + dx;
      H[1] *= mx;
      H[2] *= mx;
      H[5] *= mx;
      H[3] = H[1];
      H[6] = H[2];
      H[7] = H[5];
      if (!spd_inverse(H, 3)) {
#pragma omp atomic write
        ok = false;
      }
      double* o = &Hinv[6 * p];
      o[0] = H[0];
      o[1] = H[1];
      o[2] = H[2];
      o[3] = H[4];
      o[4] = H[5];
      o[5] = H[8];
      for (int j = 0; j < 3; ++j) gXr[3 * p + j] = mx * g[j];
    }
    if (!ok) return false;
    // cameras: damping, rhs = -(g_c' - H_cp Hpp^-1 g_X'), g_c' = mc sum q,  H_cp = -mc mx Q
    dcam.resize(N + S);
    for (i64 n = 0; n < N + S; ++n) dcam[n] = damp(hc[n], jc[n], radius);
    std::vector<double> u(3 * P);  // Hpp^-1 gX'
#pragma omp parallel for schedule(static)
    for (i64 p = 0; p < P; ++p) {
      const double* H = &Hinv[6 * p];
      const double* g = &gXr[3 * p];
      u[3 * p] = H[0] * g[0] + H[1] * g[1] + H[2] * g[2];
      u[3 * p + 1] = H[1] * g[0] + H[3] * g[1] + H[4] * g[2];
      u[3 * p + 2] = H[2] * g[0] + H[4] * g[1] + H[5] * g[2];
    }
    // gradient share and diagonal Schur block of observation k, in world coordinates (a[0..3) | a[3..12))
    auto cam_block = [&](i64 k, double* a) {
      const V3 d = dvec(k, c, X);
      const double qa_ = qa[k], ab = qa[k] * qb[k];
      // gradient share: q_k + mx Q_k u_p   (rhs = -(mc sum q + mc mx sum Q u))
      const V3 up = ld3(&u[3 * (i64)pt[k]]);
      const V3 Qu = qa_ * (up - (qb[k] * dot(d, up)) * d);
      a[0] += qg[3 * k] + mx * Qu.x;
      a[1] += qg[3 * k + 1] + mx * Qu.y;
      a[2] += qg[3 * k + 2] + mx * Qu.z;
      // diagonal block of S: Q - mx^2 Q Hpp^-1 Q
      const double* H = &Hinv[6 * (i64)pt[k]];
      double Q[9] = {qa_ - ab * d.x * d.x, -ab * d.x * d.y, -ab * d.x * d.z, 0, qa_ - ab * d.y * d.y, -ab * d.y * d.z, 0, 0,
                     qa_ - ab * d.z * d.z};
      Q[3] = Q[1];
      Q[6] = Q[2];
      Q[7] = Q[5];
      const double Hf[9] = {H[0], H[1], H[2], H[1], H[3], H[4], H[2], H[4], H[5]};
      double HQ[9];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) HQ[3 * i + j] = Hf[3 * i] * Q[j] + Hf[3 * i + 1] * Q[3 + j] + Hf[3 * i + 2] * Q[6 + j];
      const double f = mx * mx;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
          a[3 + 3 * i + j] += Q[3 * i + j] - f * (Q[3 * i] * HQ[j] + Q[3 * i + 1] * HQ[3 + j] + Q[3 * i + 2] * HQ[6 + j]);
    };
    std::vector<double> acc(12 * (N + S));
    bycam.reduce<12>(acc.data(), rev, cam_block);
    pqa.resize(E);
    pqb.resize(E);
    for (i64 e = 0; e < E; ++e) {
      const V3 d = pair_d(e, c);
      const V3 r = ld3(&pv[3 * e]) - ps[e] * d;
      const double m = mpair(e);
      const double h = m * pw[e] * dot(d, d);
      const double beta = m * pw[e] / (h + damp(h, pjs[e], radius));
      const double a = pw[e] * ps[e] * ps[e];
      pqa[e] = a;
      pqb[e] = beta;
      const V3 q = (ps[e] * pw[e]) * (r - (beta * dot(d, r)) * d);
      const double ab = a * beta;
      const double Q[9] = {a - ab * d.x * d.x, -ab * d.x * d.y, -ab * d.x * d.z, -ab * d.x * d.y, a - ab * d.y * d.y, -ab * d.y * d.z,
                           -ab * d.x * d.z,    -ab * d.y * d.z, a - ab * d.z * d.z};
      for (int j = 0; j < 3; ++j) {
        acc[12 * (i64)pi[e] + j] += (&q.x)[j];
        acc[12 * (i64)pj[e] + j] -= (&q.x)[j];
      }
      for (int j = 0; j < 9; ++j) {
        acc[12 * (i64)pi[e] + 3 + j] += Q[j];
        acc[12 * (i64)pj[e] + 3 + j] += Q[j];
      }
    }
    if (S > 0) {  // sensor blocks: the same quantities through the tangent map Rf^T (gradient Rf g, block Rf B Rf^T)
      std::vector<double> accs(12 * S);
      bysens.reduce<12>(accs.data(), rev, [&](i64 ee, double* a) {
        const i64 k = sobs[ee];
        double b[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        cam_block(k, b);
        const double* R = &Rf[9 * k];
        const V3 g = rot(k, V3{b[0], b[1], b[2]});
        a[0] += g.x;
        a[1] += g.y;
        a[2] += g.z;
        double RB[9];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) RB[3 * i + j] = R[3 * i] * b[3 + j] + R[3 * i + 1] * b[6 + j] + R[3 * i + 2] * b[9 + j];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) a[3 + 3 * i + j] += RB[3 * i] * R[3 * j] + RB[3 * i + 1] * R[3 * j + 1] + RB[3 * i + 2] * R[3 * j + 2];
      });
      std::copy(accs.begin(), accs.end(), acc.begin() + 12 * N);
    }
    std::vector<double> rhs(3 * (N + S));
    Minv.resize(9 * (N + S));
    for (i64 n = 0; n < N + S; ++n) {
      for (int j = 0; j < 3; ++j) rhs[3 * n + j] = -mc * acc[12 * n + j];
      double B[9];
      for (int j = 0; j < 9; ++j) B[j] = mc * mc * acc[12 * n + 3 + j];
      // symmetrise (rounding) and damp
      B[1] = B[3] = 0.5 * (B[1] + B[3]);
      B[2] = B[6] = 0.5 * (B[2] + B[6]);
      B[5] = B[7] = 0.5 * (B[5] + B[7]);
      B[0] += dcam[n];
      B[4] += dcam[n];
      B[8] += dcam[n];
      if (!spd_inverse(B, 3)) return false;
      std::memcpy(&Minv[9 * n], B, sizeof B);
    }
    std::vector<double> dc(3 * (N + S), 0.0);
    *relres = 0.0;
    // experiment (ORC_DEFLATE, tools/exp_deflation.py): deflate the four gauge modes of global positioning — world
    // translation (dc_n = a) and scale (dc_n = c_n) — from the PCG.  Nothing anchors them but the LM damping (no frame is
    // constant, gp.cc:437-439), so the PCG drops them again when W^T A W cannot be inverted.
    const bool deflate = std::getenv("ORC_DEFLATE") != nullptr;  // (read per solve: oracle.cpu toggles it between legs)
    std::vector<std::vector<double>> W;
    if (deflate && S == 0 && E == 0 && mc != 0.0 && defl_on) {  // like gp.hip: short solves run undeflated
      W.assign(4, std::vector<double>(3 * N, 0.0));
      for (i64 n = 0; n < N; ++n)
        for (int a = 0; a < 3; ++a) {
          W[a][3 * n + a] = 1.0;
          W[3][3 * n + a] = c[3 * n + a];
        }
    }
    *lin = solve_reduced(
        3 * (N + S), rhs, dc, pcg_tol, pcg_max, [&](const std::vector<double>& z, std::vector<double>& o) { apply(z, o); },
        [&](const std::vector<double>& r, std::vector<double>& z) {
#pragma omp parallel for schedule(static)
          for (i64 n = 0; n < N + S; ++n) {
            const double* B = &Minv[9 * n];
            for (int i = 0; i < 3; ++i) z[3 * n + i] = B[3 * i] * r[3 * n] + B[3 * i + 1] * r[3 * n + 1] + B[3 * i + 2] * r[3 * n + 2];
          }
        },
        relres, (double)M, nullptr, W.empty() ? nullptr : &W);
    defl_on = W.empty() ? *lin > 3 * 4 : *lin - (i64)W.size() > (i64)W.size();
    // back-substitution: dX_p = Hpp^-1 (-gX' + mc mx sum Q dc) ; ds_k = beta (d.(r + s (mc dc - mx dX)))
    std::vector<double> dX(3 * P);
    apply_points_only(dc);  // tp = mc mx Hpp^-1 sum Q dc
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < 3 * P; ++i) dX[i] = tp[i] - u[i];
    std::vector<double> ds(M);
    std::vector<double> mterm(M), gterm(M);
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) {
      const V3 d = dvec(k, c, X);
      const V3 r = ld3(v + 3 * k) - s[k] * d;
      const V3 e = mc * zcam(k, dc) - mx * ld3(&dX[3 * (i64)pt[k]]);
      const double m = mscale(k);
      ds[k] = qb[k] * dot(d, r + s[k] * e);  // qb = m w / h~ss
      // model: J delta = sqrt(w) (s e - m d ds), r~ = sqrt(w) r
      const V3 jd = s[k] * e - (m * ds[k]) * d;
      mterm[k] = w[k] * (dot(jd, r) + 0.5 * dot(jd, jd));
      gterm[k] = w[k] * dot(jd, r);  // this block's share of g . delta
    }
    double pair_model = 0.0, pair_xn = 0.0, pair_slope = 0.0;
    d_ps.resize(E);
    for (i64 e = 0; e < E; ++e) {
      const V3 d = pair_d(e, c);
      const V3 r = ld3(&pv[3 * e]) - ps[e] * d;
      const V3 ee = mc * (ld3(&dc[3 * (i64)pi[e]]) - ld3(&dc[3 * (i64)pj[e]]));
      const double dse = pqb[e] * dot(d, r + ps[e] * ee);
      const V3 jd = ps[e] * ee - (mpair(e) * dse) * d;
      pair_model += pw[e] * (dot(jd, r) + 0.5 * dot(jd, jd));
      pair_slope += pw[e] * dot(jd, r);
      d_ps[e] = mpair(e) * dse;
      pair_xn += ps[e] * ps[e];
    }
    *model_change = -(chunked_sum(M, [&](i64 k) { return mterm[k]; }) + pair_model);
    slope0 = chunked_sum(M, [&](i64 k) { return gterm[k]; }) + pair_slope;
    // the step in ambient coordinates (constant blocks: zero)
    d_c.resize(3 * (N + S));
    d_X.resize(3 * P);
    d_s.resize(M);
    for (i64 i = 0; i < 3 * (N + S); ++i) d_c[i] = mc * dc[i];
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < 3 * P; ++i) d_X[i] = mx * dX[i];
#pragma omp parallel for schedule(static)
    for (i64 k = 0; k < M; ++k) d_s[k] = mscale(k) * ds[k];
    dmax = chunked_max(3 * (N + S), [&](i64 i) { return std::fabs(d_c[i]); });
    dmax = std::max(dmax, chunked_max(3 * P, [&](i64 i) { return std::fabs(d_X[i]); }));
    dmax = std::max(dmax, chunked_max(M, [&](i64 k) { return std::fabs(d_s[k]); }));
    for (i64 e = 0; e < E; ++e) dmax = std::max(dmax, std::fabs(d_ps[e]));
    // candidate = Plus(x, delta), scales projected on their lower bound
    make_candidate(cand_cost, step_norm);
    double xn = chunked_sum(3 * (N + S), [&](i64 i) { return c[i] * c[i]; });
    xn += chunked_sum(3 * P, [&](i64 i) { return X[i] * X[i]; });
    xn += chunked_sum(M, [&](i64 k) { return s[k] * s[k]; });
    *x_norm = std::sqrt(xn + pair_xn);
    const double sn = *step_norm;
    bool finite = std::isfinite(sn);
    return finite;
  }

  void apply_points_only(const std::vector<double>& z) {
#pragma omp parallel for schedule(dynamic, 256)
    for (i64 p = 0; p < P; ++p) {
      double a[3] = {0, 0, 0};
      auto body = [&](i64 k) {
        const V3 d = dvec(k, c, X);
        const V3 zc = zcam(k, z);
        const V3 q = qa[k] * (zc - (qb[k] * dot(d, zc)) * d);
        a[0] += q.x;
        a[1] += q.y;
        a[2] += q.z;
      };
      if (!rev)
        for (i64 k = poff[p]; k < poff[p + 1]; ++k) body(k);
      else
        for (i64 k = poff[p + 1] - 1; k >= poff[p]; --k) body(k);
      const double* H = &Hinv[6 * p];
      const double f = mc * mx;
      tp[3 * p] = f * (H[0] * a[0] + H[1] * a[1] + H[2] * a[2]);
      tp[3 * p + 1] = f * (H[1] * a[0] + H[3] * a[1] + H[4] * a[2]);
      tp[3 * p + 2] = f * (H[2] * a[0] + H[4] * a[1] + H[5] * a[2]);
    }
  }

  void accept() override {
    c.swap(c2);
    X.swap(X2);
    s.swap(s2);
    ps.swap(ps2);
  }
};

}  // namespace
}  // namespace orc

extern "C" {
using orc::i64;

// Arrays as gsfm_gp_problem (include/gsfm.h).  cam_center_inout [N][3], pt_xyz_inout [P][3].
// Returns 0 when the solution is usable, -5 for an empty problem, -6 when not usable.
// Known rigs: image_frame [I] / image_offset [I][3] (NULL = trivial rigs): obs_cam then indexes images.
// Camera-to-camera constraints (constraint_type 1 = ONLY_CAMERAS, 2 = POINTS_AND_CAMERAS_BALANCED, 3 = POINTS_AND_CAMERAS;
// global_positioning.h:11-20, gp.cc:42-71, 167-255): pair_i / pair_j [E] frame indices, pair_dir [E][3]; trivial frames only.
int orc_gp_solve_pairs(int32_t num_cams, i64 num_pts, const i64* pt_offset, const int32_t* obs_cam, const double* obs_dir,
                       const uint8_t* obs_calibrated, const orc::GpOptionsC* o, double* cam_center_inout, double* pt_xyz_inout,
                       orc::GpReport* rep, int32_t num_threads, const int32_t* image_frame, const double* image_offset,
                       int32_t num_sensors, const int32_t* image_sensor, const double* image_sensor_rot,
                       double* sensor_center_inout, int32_t constraint_type, double constraint_reweight_scale, i64 num_pairs,
                       const int32_t* pair_i, const int32_t* pair_j, const double* pair_dir) {
  using namespace orc;
  const double t0 = omp_get_wtime();
  if (num_threads > 0) omp_set_num_threads(num_threads);
  Gp g;
  g.N = num_cams;
  const bool with_pairs = constraint_type != 0;
  const bool with_points = constraint_type != 1;
  if (with_pairs && (image_frame || num_pairs <= 0 || !pair_i || !pair_j || !pair_dir)) return with_pairs && num_pairs <= 0 ? -5 : -7;
  g.S = (image_frame && num_sensors > 0 && image_sensor && image_sensor_rot && sensor_center_inout) ? num_sensors : 0;
  if (g.S > 0 && !o->optimize_positions) return -7;
  std::vector<i64> used_pts;
  g.poff.push_back(0);
  for (i64 p = 0; p < num_pts; ++p) {
    const i64 len = pt_offset[p + 1] - pt_offset[p];
    if (len < o->min_num_view_per_track) continue;  // gp.cc:258
    const i64 id = (i64)used_pts.size();
    used_pts.push_back(p);
    for (i64 k = pt_offset[p]; k < pt_offset[p + 1]; ++k) {
      if (image_frame) {
        g.cam.push_back(image_frame[obs_cam[k]]);
        for (int j = 0; j < 3; ++j) g.off.push_back(image_offset[3 * (i64)obs_cam[k] + j]);
        if (g.S > 0) {
          const i64 im = obs_cam[k];
          g.sblk.push_back(image_sensor[im]);
          for (int j = 0; j < 9; ++j) g.Rf.push_back(image_sensor_rot[9 * im + j]);
          if (image_sensor[im] >= 0) {
            g.sobs.push_back((int32_t)g.cam.size() - 1);
            g.sown.push_back(image_sensor[im]);
          }
        }
      } else {
        g.cam.push_back(obs_cam[k]);
      }
      g.pt.push_back((int32_t)id);
      g.vbuf.push_back(obs_dir[3 * k]);
      g.vbuf.push_back(obs_dir[3 * k + 1]);
      g.vbuf.push_back(obs_dir[3 * k + 2]);
      g.cal.push_back(obs_calibrated ? obs_calibrated[k] : 1);
    }
    g.poff.push_back((i64)g.cam.size());
  }
  g.P = (i64)used_pts.size();
  g.M = (i64)g.cam.size();
  std::memset(rep, 0, sizeof *rep);
  rep->threads = omp_get_max_threads();
  if (g.M == 0 && with_points) return -5;
  // InitializeRandomPositions (gp.cc:121-163) marks the frames of the valid pairs and of the kept tracks, whatever the type
  std::vector<uint8_t> constrained(g.N, 0);
  for (i64 k = 0; k < g.M; ++k) constrained[g.cam[k]] = 1;
  if (with_pairs) {
    g.E = num_pairs;
    g.fixed_obs = -1;
    g.pi.assign(pair_i, pair_i + num_pairs);
    g.pj.assign(pair_j, pair_j + num_pairs);
    g.pv.assign(pair_dir, pair_dir + 3 * num_pairs);
    g.ps.assign(num_pairs, 1.0);
    for (i64 e = 0; e < num_pairs; ++e) {
      if (pair_i[e] < 0 || pair_i[e] >= num_cams || pair_j[e] < 0 || pair_j[e] >= num_cams) return -1;
      constrained[pair_i[e]] = 1;
      constrained[pair_j[e]] = 1;
    }
  }
  if (!with_points) {  // AddPointToCameraConstraints is skipped (gp.cc:69-71): no draws for the points, no residuals
    g.cam.clear();
    g.pt.clear();
    g.vbuf.clear();
    g.cal.clear();
    g.poff.assign(1, 0);
    used_pts.clear();
    g.P = g.M = 0;
  }
  g.v = g.vbuf.data();
  g.bycam.build(g.N, g.M, g.cam.data());
  if (g.S > 0) g.bysens.build(g.S, (i64)g.sobs.size(), g.sown.data());
  // POINTS_AND_CAMERAS_BALANCED: the point-to-camera losses are scaled by reweight * #pairs / tracks.size() (gp.cc:223-255)
  const double wpt = (constraint_type == 2 && num_pairs > 0 && num_pts > 0) ? constraint_reweight_scale * (double)num_pairs / (double)num_pts : 1.0;
  g.loss_cal = {o->thres_loss_function, wpt};
  g.loss_unc = {o->thres_loss_function, 0.5 * wpt};
  g.loss_pair = {o->thres_loss_function, 1.0};
  g.mc = o->optimize_positions ? 1.0 : 0.0;
  g.mx = o->optimize_points ? 1.0 : 0.0;
  g.ms = o->optimize_scales ? 1.0 : 0.0;
  g.lm_lo = o->min_lm_diagonal;
  g.lm_hi = o->max_lm_diagonal;
  g.rev = o->order != 0;
  g.pcg_tol = o->pcg_relative_tolerance;
  g.pcg_max = o->pcg_max_iterations;
  g.c.assign(cam_center_inout, cam_center_inout + 3 * g.N);
  for (i64 i = 0; i < 3 * g.S; ++i) g.c.push_back(sensor_center_inout[i]);
  g.X.resize(3 * g.P);
  for (i64 i = 0; i < g.P; ++i)
    for (int j = 0; j < 3; ++j) g.X[3 * i + j] = pt_xyz_inout[3 * used_pts[i] + j];
  // random initialisation, gp.cc:123-165,261-264
  std::mt19937 gen(o->seed);
  std::uniform_real_distribution<double> uni(-1.0, 1.0);
  if (o->generate_random_positions && o->optimize_positions) {
    for (i64 n = 0; n < g.N; ++n)
      if (constrained[n])
        for (int j = 0; j < 3; ++j) g.c[3 * n + j] = 100.0 * uni(gen);
  }
  if (o->generate_random_points && o->optimize_points)
    for (i64 i = 0; i < 3 * g.P; ++i) g.X[i] = 100.0 * uni(gen);
  for (i64 i = 0; i < 3 * g.S; ++i) g.c[3 * g.N + i] = uni(gen);  // ParameterizeVariables, gp.cc:442-456 (after every other draw)
  g.s.assign(g.M, 1.0);
  if (!o->generate_scales)
    for (i64 k = 0; k < g.M; ++k) {
      const V3 d = g.dvec(k, g.c, g.X);
      g.s[k] = std::max(1e-5, dot(ld3(g.v + 3 * k), d) / dot(d, d));
    }
  LmOptions lo;
  lo.max_num_iterations = o->max_num_iterations;
  lo.function_tolerance = o->function_tolerance;
  lo.gradient_tolerance = o->gradient_tolerance;
  lo.parameter_tolerance = o->parameter_tolerance;
  lo.initial_trust_region_radius = o->initial_trust_region_radius;
  lo.max_trust_region_radius = o->max_trust_region_radius;
  lo.min_trust_region_radius = o->min_trust_region_radius;
  lo.min_relative_decrease = o->min_relative_decrease;
  lo.min_lm_diagonal = o->min_lm_diagonal;
  lo.max_lm_diagonal = o->max_lm_diagonal;
  lo.jacobi_scaling = o->jacobi_scaling;
  lo.max_num_consecutive_invalid_steps = o->max_num_consecutive_invalid_steps;
  lo.verbose = o->verbose;
  lo.line_search = o->line_search;
  LmSummary s;
  lm_minimize(g, lo, &s);
  std::memcpy(cam_center_inout, g.c.data(), sizeof(double) * 3 * g.N);
  for (i64 i = 0; i < 3 * g.S; ++i) sensor_center_inout[i] = g.c[3 * g.N + i];
  for (i64 i = 0; i < g.P; ++i)
    for (int j = 0; j < 3; ++j) pt_xyz_inout[3 * used_pts[i] + j] = g.X[3 * i + j];
  rep->iterations = s.iterations;
  rep->successful_steps = s.successful_steps;
  rep->termination = s.termination;
  rep->usable = s.usable;
  rep->linear_iterations = s.linear_iterations;
  rep->initial_cost = s.initial_cost;
  rep->final_cost = s.final_cost;
  rep->max_linear_residual = s.max_linear_residual;
  rep->seconds_linear = s.seconds_linear;
  rep->line_search_shrunk = s.line_search_shrunk;
  rep->seconds_total = omp_get_wtime() - t0;
  return s.usable ? 0 : -6;
}

int orc_gp_solve(int32_t num_cams, i64 num_pts, const i64* pt_offset, const int32_t* obs_cam, const double* obs_dir,
                 const uint8_t* obs_calibrated, const orc::GpOptionsC* o, double* cam_center_inout, double* pt_xyz_inout,
                 orc::GpReport* rep, int32_t num_threads, const int32_t* image_frame, const double* image_offset,
                 int32_t num_sensors, const int32_t* image_sensor, const double* image_sensor_rot,
                 double* sensor_center_inout) {
  return orc_gp_solve_pairs(num_cams, num_pts, pt_offset, obs_cam, obs_dir, obs_calibrated, o, cam_center_inout, pt_xyz_inout, rep,
                            num_threads, image_frame, image_offset, num_sensors, image_sensor, image_sensor_rot,
                            sensor_center_inout, 0, 1.0, 0, nullptr, nullptr, nullptr);
}

int orc_num_threads(void) { return omp_get_max_threads(); }

// the LM iterations of the last orc_gp_solve / orc_ba_solve of this process (orc_lm.hpp lm_trace_store): copies up to
// max_rows rows of 7 doubles, returns the number recorded
int32_t orc_lm_trace(double* out, int32_t max_rows) {
  const std::vector<double>& t = orc::lm_trace_store();
  const int32_t rows = (int32_t)(t.size() / 7);
  if (out && max_rows > 0) std::memcpy(out, t.data(), sizeof(double) * 7 * (size_t)std::min(rows, max_rows));
  return rows;
}

// test hook (tests/test_oracle_cpu.py): one interpolation step of the Armijo search — the minimiser over [x_lo, x_hi] of the
// polynomial through n samples (x, value, slope) — so that orc_lm.hpp's fit / root finder can be held to oracle/lm.py's
double orc_ls_interpolate(int32_t n, const double* x, const double* value, const double* slope, double x_lo, double x_hi) {
  std::vector<orc::LsSample> smp;
  for (int32_t i = 0; i < n; ++i) smp.push_back(orc::LsSample{x[i], value[i], slope[i], true});
  return orc::poly_minimize(orc::poly_fit(smp), x_lo, x_hi);
}
}
