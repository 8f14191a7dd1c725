// ra.hip — rotation averaging on MI355X (gfx950).
//
// Replaces RotationEstimator::EstimateRotations (glomap/estimators/global_rotation_averaging.cc:40-85).  The 3-DoF path
// on frames is described here; gravity-aligned frames (use_gravity: 1-DoF unknowns, k_ra_mask3) and cam_from_rig
// rotations among the unknowns (image-level graph + cam blocks, ra_solve_rig_impl) reuse it, see their sections below:
//   host  : maximum-spanning-tree initialisation            gra.cc:87-138, math/tree.cc:78-153
//   device: ComputeResiduals                                gra.cc:696-756      -> k_node_quat, k_edge_residual
//           L1 stage, colmap::LeastAbsoluteDeviationSolver  gra.cc:479-541      -> k_admm_edge, k_node_gather<L1*>
//           IRLS stage (Geman-McClure / half-norm weights)  gra.cc:543-625      -> k_node_gather<IRLS>
//           CHOLMOD solve of A^T W A x = A^T W b            gra.cc:603-611      -> dense direct (ra_dense.hpp, N <= 2048)
//                                                                                  or Jacobi-PCG (cg.hpp), 3 right-hand sides
//           UpdateGlobalRotations / ComputeAverageStepSize  gra.cc:627-644,758-772 -> k_node_update
//
// Structure exploited: A = B (x) I3 with B the signed incidence matrix of the view graph and the
// IRLS weight constant over an edge's three rows (gra.cc:599), so A^T W A = L_w (x) I3 + gauge:
// ONE scalar weighted graph Laplacian with three right-hand sides.  Nothing of size 3E x 3N is
// ever formed; the Laplacian lives as a CSR-by-node incidence list with per-incidence weights.
//
// Data layout in HBM (all f64 unless noted):
//   edge_i/j[E] i32, edge_q[E][4], edge_w[E]           inputs (SoA, read coalesced by the edge sweep)
//   rowptr[N+1] i32, inc[2E] i32 ((eid<<1)|is_j), nbr[2E] i32, inc_w[2E]   CSR-by-node incidence
//   rot[N][3], nq[N][4]                                 node state (angle-axis) + its quaternion
//   res[E+1][3] (row E = gauge rows), wirls[E]          residuals / IRLS weights
//   z,u,dz[E+1][3]                                      ADMM state
//   rhs,x[N][3], lap_diag[N], lap_diag_loc[N]           right-hand side / solution, Laplacian diagonal (global / this rank)
//   cg_*[N][3]                                          PCG vectors (cg.hpp);  dense_a/b[Np][Np] dense Laplacian / inverse
#include <algorithm>
#include <numeric>

#include "cg.hpp"
#include "device.hpp"
#include "dump.hpp"
#include "ra_dense.hpp"
#include "ra_sub.hpp"

namespace gsfm {
namespace {

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// nq[n] = quat(Exp(rot[n])); the thread of the fixed node also evaluates the gauge rows
// Log(Exp(r_fix0)^T Exp(r_fix)) (gra.cc:751-755) into gauge_out[3].
__global__ void __launch_bounds__(kBlock) k_node_quat(int N, const double* __restrict__ rot,
                                                      double* __restrict__ nq, int fixed_node,
                                                      const double* __restrict__ fixed_rot0,
                                                      double* __restrict__ gauge_out, int has_gauge) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const Quat q = aa_to_quat(rot[3 * n], rot[3 * n + 1], rot[3 * n + 2]);
    store_quat(nq + 4 * n, q);
    if (n == fixed_node) {
      double gx = 0, gy = 0, gz = 0;
      if (has_gauge) {
        const Quat q0 = aa_to_quat(fixed_rot0[0], fixed_rot0[1], fixed_rot0[2]);
        quat_to_aa(qmul(qconj(q0), q), gx, gy, gz);
      }
      gauge_out[0] = gx;
      gauge_out[1] = gy;
      gauge_out[2] = gz;
    }
  }
}

// Per edge: b_e = -Log(R_j^T R_rel R_i)  (gra.cc:741-742) and, when wirls != nullptr, the IRLS
// weight of that residual (gra.cc:579-588).  Algorithmic traffic per edge: 8 B indices + 32 B
// q_rel read, 24 B residual + 8 B weight written; the two node quaternions are gathers that hit
// L2 (N*32 B is 32 KB at C2, 320 KB at N = 10k).
__global__ void __launch_bounds__(kBlock)
    k_edge_residual(long E, const int* __restrict__ ei, const int* __restrict__ ej,
                    const double* __restrict__ eq, const double* __restrict__ nq,
                    double* __restrict__ res, double* __restrict__ wirls, int weight_type,
                    double sigma2, int* __restrict__ nan_flag,
                    const unsigned char* __restrict__ grav /* use_gravity: [N] frame has gravity, else null */,
                    const double* __restrict__ rot /* node angle-axis (gravity frames: (0, angle, 0)) */) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
    const Quat qi = load_quat(nq + 4 * (long)ei[e]);
    const Quat qj = load_quat(nq + 4 * (long)ej[e]);
    const Quat qr = load_quat(eq + 4 * e);
    double ax, ay, az, xz = 0.0;
    if (grav != nullptr && grav[ei[e]] && grav[ej[e]]) {
      // both frames gravity aligned: ONE row, RelAngleError of the angles against the y component of the (aligned)
      // relative rotation; its x / z components are a constant of the IRLS weight (gra.cc:19-36, 329-338, 571-573, 709-713)
      double rx, ry, rz;
      quat_to_aa(qr, rx, ry, rz);
      double est = (rot[3 * (long)ej[e] + 1] - rot[3 * (long)ei[e] + 1]) - ry;
      est = fmod(est + M_PI, 2.0 * M_PI);
      if (est < 0.0) est += 2.0 * M_PI;
      est -= M_PI;  // [-pi, pi)
      ax = 0.0;
      ay = -est;
      az = 0.0;
      xz = rx * rx + rz * rz;
    } else {
      const Quat m = qmul(qmul(qconj(qj), qr), qi);
      quat_to_aa(m, ax, ay, az);
    }
    res[3 * e] = -ax;
    res[3 * e + 1] = -ay;
    res[3 * e + 2] = -az;
    if (wirls != nullptr) {
      const double e2 = ax * ax + ay * ay + az * az + xz;
      double w;
      if (weight_type == 0) {  // GEMAN_MCCLURE
        const double t = e2 + sigma2;
        w = sigma2 / (t * t);
      } else {  // HALF_NORM: err^((0.5-2)/2)
        w = pow(e2, -0.75);
      }
      if (isnan(w)) atomicOr(nan_flag, 1);
      wirls[e] = w;
    }
  }
}

// Node-centric gathers over the incidence list (deterministic, no atomics): LPR lanes per node.
//   GATHER_IRLS : rhs_n = sum_k sgn_k c_k b_k,          c = wirls * w        (A^T W_irls W b, gra.cc:603-611)
//                 lap_diag_n = sum_k c_k (+1 on the gauge node), inc_w[k] = c_k
//   GATHER_L1W  : inc_w[k] = w_k^2, lap_diag_n = sum_k w_k^2 (+1)           ((WA)^T (WA), gra.cc:488-491)
//   GATHER_L1RHS: rhs_n = sum_k sgn_k w_k (w_k b_k + z_k - u_k)             (A'^T (b' + z - u))
//                 s_n   = sum_k sgn_k w_k dz_k,  t_n = sum_k sgn_k w_k u_k  (dual residual terms)
enum { GATHER_IRLS = 0, GATHER_L1W = 1, GATHER_L1RHS = 2 };

template <int MODE, int LPR>
__global__ void __launch_bounds__(kBlock)
    k_node_gather(int N, long E, const int* __restrict__ rowptr, const int* __restrict__ inc,
                  const double* __restrict__ res, const double* __restrict__ wirls,
                  const double* __restrict__ ew, const double* __restrict__ z,
                  const double* __restrict__ u, const double* __restrict__ dz,
                  double* __restrict__ inc_w, double* __restrict__ lap_diag, double* __restrict__ lap_diag_loc,
                  double* __restrict__ rhs, double* __restrict__ gat_s, double* __restrict__ gat_t,
                  int fixed_node, int has_gauge, const int* __restrict__ stop) {
  if (stop != nullptr && *stop) return;
  const int gpb = kBlock / LPR;  // groups (nodes) per block per sweep
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  for (int n = blockIdx.x * gpb + g; n < N; n += gridDim.x * gpb) {
    const int k0 = rowptr[n], k1 = rowptr[n + 1];
    double a0 = 0, a1 = 0, a2 = 0, s0 = 0, s1 = 0, s2 = 0, t0 = 0, t1 = 0, t2 = 0, dsum = 0;
    for (int k = k0 + l; k < k1; k += LPR) {
      const int code = inc[k];
      const long e = code >> 1;
      const double sgn = (code & 1) ? 1.0 : -1.0;
      const double w = ew ? ew[e] : 1.0;
      if constexpr (MODE == GATHER_IRLS) {
        const double c = wirls[e] * w;
        inc_w[k] = c;
        dsum += c;
        const double sc = sgn * c;
        a0 += sc * res[3 * e];
        a1 += sc * res[3 * e + 1];
        a2 += sc * res[3 * e + 2];
      } else if constexpr (MODE == GATHER_L1W) {
        inc_w[k] = w * w;
        dsum += w * w;
      } else {
        const double sc = sgn * w;
        a0 += sc * (w * res[3 * e] + z[3 * e] - u[3 * e]);
        a1 += sc * (w * res[3 * e + 1] + z[3 * e + 1] - u[3 * e + 1]);
        a2 += sc * (w * res[3 * e + 2] + z[3 * e + 2] - u[3 * e + 2]);
        s0 += sc * dz[3 * e];
        s1 += sc * dz[3 * e + 1];
        s2 += sc * dz[3 * e + 2];
        t0 += sc * u[3 * e];
        t1 += sc * u[3 * e + 1];
        t2 += sc * u[3 * e + 2];
      }
    }
    a0 = group_sum<LPR>(a0);
    a1 = group_sum<LPR>(a1);
    a2 = group_sum<LPR>(a2);
    if constexpr (MODE != GATHER_L1RHS) dsum = group_sum<LPR>(dsum);
    if constexpr (MODE == GATHER_L1RHS) {
      s0 = group_sum<LPR>(s0);
      s1 = group_sum<LPR>(s1);
      s2 = group_sum<LPR>(s2);
      t0 = group_sum<LPR>(t0);
      t1 = group_sum<LPR>(t1);
      t2 = group_sum<LPR>(t2);
    }
    if (l == 0) {
      const bool gauge = has_gauge && n == fixed_node;
      if (gauge) {  // gauge rows: +I3 at the fixed node, weight 1 (gra.cc:455-460, 560)
        const long ge = 3 * E;
        if constexpr (MODE == GATHER_IRLS) {
          a0 += res[ge];
          a1 += res[ge + 1];
          a2 += res[ge + 2];
        } else if constexpr (MODE == GATHER_L1RHS) {
          a0 += res[ge] + z[ge] - u[ge];
          a1 += res[ge + 1] + z[ge + 1] - u[ge + 1];
          a2 += res[ge + 2] + z[ge + 2] - u[ge + 2];
          s0 += dz[ge];
          s1 += dz[ge + 1];
          s2 += dz[ge + 2];
          t0 += u[ge];
          t1 += u[ge + 1];
          t2 += u[ge + 2];
        }
        dsum += 1.0;
      }
      if constexpr (MODE != GATHER_L1RHS) {
        lap_diag[n] = dsum;      // all-reduced afterwards when the edges are sharded: Jacobi preconditioner
        lap_diag_loc[n] = dsum;  // stays local: diagonal of THIS rank's share of the operator (SpMV)
      }
      if constexpr (MODE != GATHER_L1W) {
        rhs[3 * n] = a0;
        rhs[3 * n + 1] = a1;
        rhs[3 * n + 2] = a2;
      }
      if constexpr (MODE == GATHER_L1RHS) {
        gat_s[3 * n] = s0;
        gat_s[3 * n + 1] = s1;
        gat_s[3 * n + 2] = s2;
        gat_t[3 * n] = t0;
        gat_t[3 * n + 1] = t1;
        gat_t[3 * n + 2] = t2;
      }
    }
  }
}

// (L_w (x) I3 + gauge) v for one node row, LPR lanes cooperating; result valid in all lanes.
template <int LPR>
__device__ __forceinline__ void laplacian_row(int n, int l, const int* __restrict__ rowptr,
                                              const int* __restrict__ nbr,
                                              const double* __restrict__ inc_w,
                                              const double* __restrict__ lap_diag,
                                              const double* __restrict__ v, double& y0, double& y1,
                                              double& y2) {
  const int k0 = rowptr[n], k1 = rowptr[n + 1];
  double a0 = 0, a1 = 0, a2 = 0;
  for (int k = k0 + l; k < k1; k += LPR) {
    const long m = nbr[k];
    const double w = inc_w[k];
    a0 += w * v[3 * m];
    a1 += w * v[3 * m + 1];
    a2 += w * v[3 * m + 2];
  }
  a0 = group_sum<LPR>(a0);
  a1 = group_sum<LPR>(a1);
  a2 = group_sum<LPR>(a2);
  const double d = lap_diag[n];
  y0 = d * v[3 * (long)n] - a0;
  y1 = d * v[3 * (long)n + 1] - a1;
  y2 = d * v[3 * (long)n + 2] - a2;
}

// Stand-alone SpMV  y = (L_w (x) I3 + gauge) v   (multi-rank path, warm starts, public API).
// Algorithmic traffic per launch: 2E*(4 B nbr + 8 B w) + N*(24 in + 24 out + 8 diag + 4 rowptr).
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    k_spmv(int N, const int* __restrict__ rowptr, const int* __restrict__ nbr,
           const double* __restrict__ inc_w, const double* __restrict__ lap_diag,
           const double* __restrict__ v, double* __restrict__ y) {
  const int gpb = kBlock / LPR;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (int n = blockIdx.x * gpb + g; n < N; n += gridDim.x * gpb) {
    double y0, y1, y2;
    laplacian_row<LPR>(n, l, rowptr, nbr, inc_w, lap_diag, v, y0, y1, y2);
    if (l == 0) {
      y[3 * (long)n] = y0;
      y[3 * (long)n + 1] = y1;
      y[3 * (long)n + 2] = y2;
    }
  }
}

// w = (L_w (x) I3 + gauge) u plus this block's share of gamma = r.u and delta = w.u  (block-dense PCG, ra_dense.hpp)
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    k_bd_spmv(int N, const int* __restrict__ rowptr, const int* __restrict__ nbr, const double* __restrict__ inc_w,
              const double* __restrict__ lap_diag, const double* __restrict__ u, const double* __restrict__ r,
              double* __restrict__ w, double* __restrict__ dpart, const DpcgState* __restrict__ st) {
  __shared__ double smem[4 * 2];
  if (st->done) return;
  const int gpb = kBlock / LPR;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  double acc[2] = {0.0, 0.0};
  for (int n = blockIdx.x * gpb + g; n < N; n += gridDim.x * gpb) {
    double y0, y1, y2;
    laplacian_row<LPR>(n, l, rowptr, nbr, inc_w, lap_diag, u, y0, y1, y2);
    if (l == 0) {
      const long o = 3 * (long)n;
      w[o] = y0;
      w[o + 1] = y1;
      w[o + 2] = y2;
      acc[0] += r[o] * u[o] + r[o + 1] * u[o + 1] + r[o + 2] * u[o + 2];
      acc[1] += y0 * u[o] + y1 * u[o + 1] + y2 * u[o + 2];
    }
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    dpart[2 * blockIdx.x] = acc[0];
    dpart[2 * blockIdx.x + 1] = acc[1];
  }
}

// ---- Jacobi-PCG (cg.hpp) on the 3N system  (L_w (x) I3 + gauge) x = rhs ---------------------------
// The three right-hand sides share the operator, so they are solved as ONE SPD system of 3N unknowns
// (block-diagonal in the column index) with the single-reduction PCG of cg.hpp: per iteration the
// SpMV below (w = A z and this block's share of delta = z.w) and k_cg_update<3, false> whose 3 x 3
// "camera blocks" are the Jacobi preconditioner diag(1/d_n) I3.  With edges sharded over ranks the
// driver all-reduces w; the diagonal used here is then the LOCAL one (lap_diag_loc).
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    k_ra_apply(int N, const int* __restrict__ rowptr, const int* __restrict__ nbr,
               const double* __restrict__ inc_w, const double* __restrict__ lap_diag_loc, CgVec v, int it,
               double tol2, const unsigned char* __restrict__ grav) {
  __shared__ double smem[4 * 2 + 2];
  if (cg_converged(v, it, tol2, smem)) return;
  const int gpb = kBlock / LPR;
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  double acc[1] = {0.0};
  for (int n = blockIdx.x * gpb + g; n < N; n += gridDim.x * gpb) {
    double w0, w1, w2;
    laplacian_row<LPR>(n, l, rowptr, nbr, inc_w, lap_diag_loc, v.z, w0, w1, w2);
    if (grav != nullptr && grav[n]) {  // a gravity frame has no x / z unknowns (k_ra_mask3)
      w0 = 0.0;
      w2 = 0.0;
    }
    if (l == 0) {
      const long i = 3 * (long)n;
      v.w[i] = w0;
      v.w[i + 1] = w1;
      v.w[i + 2] = w2;
      acc[0] += v.z[i] * w0 + v.z[i + 1] * w1 + v.z[i + 2] * w2;
    }
  }
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) v.dpart[blockIdx.x] = acc[0];
}

// use_gravity: the x and z components of a gravity-aligned frame are not unknowns (its one column sits in the y rows,
// gra.cc:390-418): every vector of the normal equations carries zeros there.  With them masked the ordinary Laplacian
// machinery solves the reference's mixed 1-DoF / 3-DoF system: the y components see the whole graph, the x / z components
// the graph of the other frames with their edges to gravity frames acting as anchors.
__global__ void __launch_bounds__(kBlock)
    k_ra_mask3(int N, const unsigned char* __restrict__ grav, double* __restrict__ a, double* __restrict__ b,
               double* __restrict__ c) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    if (!grav[n]) continue;
    a[3 * (long)n] = 0.0;
    a[3 * (long)n + 2] = 0.0;
    if (b) {
      b[3 * (long)n] = 0.0;
      b[3 * (long)n + 2] = 0.0;
    }
    if (c) {
      c[3 * (long)n] = 0.0;
      c[3 * (long)n + 2] = 0.0;
    }
  }
}

// Jacobi preconditioner as 3 x 3 blocks for cg.hpp: minv[n] = (1 / d_n) I3
__global__ void __launch_bounds__(kBlock)
    k_ra_minv(int N, const double* __restrict__ lap_diag, double* __restrict__ minv) {
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const double di = 1.0 / lap_diag[n];
    double* m = minv + 9 * (long)n;
    m[0] = di; m[1] = 0; m[2] = 0;
    m[3] = 0; m[4] = di; m[5] = 0;
    m[6] = 0; m[7] = 0; m[8] = di;
  }
}

// x = x0 + dx
__global__ void __launch_bounds__(kBlock)
    k_ra_add(long n, const double* x0, const double* __restrict__ dx, double* x) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = x0[i] + dx[i];
}

// r_n <- Log(Exp(r_n) Exp(-delta_n)) (gra.cc:635-640) + the per-iteration scalars:
// sum |delta_n| (ComputeAverageStepSize, gra.cc:758-772), sum delta^2 (curr_norm, gra.cc:522),
// NaN count (gra.cc:508-512).
__global__ void __launch_bounds__(kBlock)
    k_node_update(int N, double* __restrict__ rot, const double* __restrict__ delta,
                  double* __restrict__ part /* [grid][3] */) {
  __shared__ double smem[4 * 3];
  double acc[3] = {0, 0, 0};
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const double dx = delta[3 * n], dy = delta[3 * n + 1], dz = delta[3 * n + 2];
    const double d2 = dx * dx + dy * dy + dz * dz;
    acc[0] += sqrt(d2);
    acc[1] += d2;
    acc[2] += (isnan(dx) || isnan(dy) || isnan(dz)) ? 1.0 : 0.0;
    const Quat q = aa_to_quat(rot[3 * n], rot[3 * n + 1], rot[3 * n + 2]);
    const Quat d = aa_to_quat(-dx, -dy, -dz);
    double ax, ay, az;
    quat_to_aa(qmul(q, d), ax, ay, az);
    rot[3 * n] = ax;
    rot[3 * n + 1] = ay;
    rot[3 * n + 2] = az;
  }
  block_sum<3>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) part[blockIdx.x * 3 + k] = acc[k];
  }
}

// ADMM z/u update of colmap::LeastAbsoluteDeviationSolver::Solve for A' = W A, b' = W b (row E =
// the three gauge rows).  Writes dz = z - z_old and partial sums |A'x - z - b'|^2, |A'x|^2, |z|^2,
// |b'|^2.
__global__ void __launch_bounds__(kBlock)
    k_admm_edge(long E, int has_gauge, int fixed_node, const int* __restrict__ ei,
                const int* __restrict__ ej, const double* __restrict__ ew,
                const double* __restrict__ res, const double* __restrict__ x,
                double* __restrict__ z, double* __restrict__ u, double* __restrict__ dz, double alpha,
                double inv_rho, double* __restrict__ part /* [grid][4] */, const int* __restrict__ stop) {
  __shared__ double smem[4 * 4];
  if (stop != nullptr && *stop) return;
  double acc[4] = {0, 0, 0, 0};
  const long rows = E + (has_gauge ? 1 : 0);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < rows; e += (long)gridDim.x * blockDim.x) {
    double w = 1.0;
    long i = 0, j = 0;
    const bool gauge = e == E;
    if (!gauge) {
      w = ew ? ew[e] : 1.0;
      i = ei[e];
      j = ej[e];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double Ax = gauge ? x[3 * (long)fixed_node + c] : w * (x[3 * j + c] - x[3 * i + c]);
      const double b = w * res[3 * e + c];
      const double zo = z[3 * e + c];
      const double uo = u[3 * e + c];
      const double Ax_hat = alpha * Ax + (1.0 - alpha) * (zo + b);
      const double v = Ax_hat - b + uo;
      const double zn = fmax(0.0, v - inv_rho) - fmax(0.0, -v - inv_rho);
      z[3 * e + c] = zn;
      dz[3 * e + c] = zn - zo;
      u[3 * e + c] = uo + Ax_hat - zn - b;
      const double rn = Ax - zn - b;
      acc[0] += rn * rn;
      acc[1] += Ax * Ax;
      acc[2] += zn * zn;
      acc[3] += b * b;
    }
  }
  block_sum<4>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) part[blockIdx.x * 4 + k] = acc[k];
  }
}


// ---- fused ADMM sweep of the dense (single-GPU, N <= 2048) L1 stage -------------------------------------
// One launch = k_admm_edge + k_node_gather<GATHER_L1RHS> + k_admm_check.  Node-centric: the wave(s) of node n
// walk its incidence list and evaluate the z / u update of every incident edge — each edge is evaluated by
// BOTH endpoints with the same operand order (bit-identical values), so the gathers need no second pass; the
// endpoint that is image_id1 owns the edge: it stores the new z / u (double-buffered: the other endpoint may
// still be reading the old ones) and accounts the edge in the norm partials.  The block that finishes last
// (ticket counter) re-reduces the partials and evaluates LeastAbsoluteDeviationSolver's stopping test.
template <int LPR>
__global__ void __launch_bounds__(kBlock)
    k_admm_node(int N, long E, const int* __restrict__ rowptr, const int* __restrict__ inc, const int* __restrict__ nbr,
                const double* __restrict__ ew, const double* __restrict__ res, const double* __restrict__ x,
                const double* __restrict__ z_in, const double* __restrict__ u_in, double* __restrict__ z_out,
                double* __restrict__ u_out, double alpha, double inv_rho, double* __restrict__ rhs, int fixed_node,
                int has_gauge, double* part /* [grid][6] */, unsigned* ticket, double rows, double abs_tol, double rel_tol,
                double rho, int* stop, int* count) {
  __shared__ double smem[4 * 6 + 6];
  __shared__ int last_s;
  if (*stop) return;
  const int gpb = kBlock / LPR;
  const int g = threadIdx.x / LPR;
  const int l = threadIdx.x % LPR;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  auto edge = [&](long e, double w, double sgn, const double (&xi)[3], const double (&xj)[3], bool gauge, bool owner,
                  double (&a)[3], double (&sv)[3], double (&tv)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double Ax = gauge ? xj[c] : w * (xj[c] - xi[c]);
      const double b = w * res[3 * e + c];
      const double zo = z_in[3 * e + c];
      const double uo = u_in[3 * e + c];
      const double Ax_hat = alpha * Ax + (1.0 - alpha) * (zo + b);
      const double v = Ax_hat - b + uo;
      const double zn = fmax(0.0, v - inv_rho) - fmax(0.0, -v - inv_rho);
      const double un = uo + Ax_hat - zn - b;
      if (owner) {
        z_out[3 * e + c] = zn;
        u_out[3 * e + c] = un;
        const double rn = Ax - zn - b;
        acc[0] += rn * rn;
        acc[1] += Ax * Ax;
        acc[2] += zn * zn;
        acc[3] += b * b;
      }
      const double sc = sgn * w;
      a[c] += sc * (b + zn - un);
      sv[c] += sc * (zn - zo);
      tv[c] += sc * un;
    }
  };
  for (int n = blockIdx.x * gpb + g; n < N; n += gridDim.x * gpb) {
    const int k0 = rowptr[n], k1 = rowptr[n + 1];
    const double xn[3] = {x[3 * (long)n], x[3 * (long)n + 1], x[3 * (long)n + 2]};
    double a[3] = {0, 0, 0}, sv[3] = {0, 0, 0}, tv[3] = {0, 0, 0};
    for (int k = k0 + l; k < k1; k += LPR) {
      const int code = inc[k];
      const long e = code >> 1;
      const long m = nbr[k];
      const double xm[3] = {x[3 * m], x[3 * m + 1], x[3 * m + 2]};
      const double w = ew ? ew[e] : 1.0;
      if (code & 1)  // this node is image_id2 (+I3): Ax = w (x_n - x_m)
        edge(e, w, 1.0, xm, xn, false, false, a, sv, tv);
      else  // this node is image_id1 (-I3): it owns the edge
        edge(e, w, -1.0, xn, xm, false, true, a, sv, tv);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a[c] = group_sum<LPR>(a[c]);
      sv[c] = group_sum<LPR>(sv[c]);
      tv[c] = group_sum<LPR>(tv[c]);
    }
    if (l == 0) {
      if (has_gauge && n == fixed_node) edge(E, 1.0, 1.0, xn, xn, true, true, a, sv, tv);  // gauge rows (gra.cc:455-460)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        rhs[3 * (long)n + c] = a[c];
        acc[4] += sv[c] * sv[c];
        acc[5] += tv[c] * tv[c];
      }
    }
  }
  block_sum<6>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) part[blockIdx.x * 6 + k] = acc[k];
    __threadfence();
    last_s = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last_s) return;
  __threadfence();
  double h[6];
  reduce_partials<6>(part, gridDim.x, h, smem);
  if (threadIdx.x == 0) {
    const double r_norm = sqrt(h[0]), Ax_norm = sqrt(h[1]), z_norm = sqrt(h[2]), b_norm = sqrt(h[3]);
    const double s_norm = rho * sqrt(h[4]), dual_norm = rho * sqrt(h[5]);
    const double primal_eps = sqrt(rows) * abs_tol + rel_tol * fmax(Ax_norm, fmax(z_norm, b_norm));
    const double dual_eps = sqrt(3.0 * N) * abs_tol + rel_tol * dual_norm;
    *count += 1;
    if (r_norm < primal_eps && s_norm < dual_eps) *stop = 1;
    *ticket = 0u;
  }
}

// partial sums of squares of two N*3 vectors -> part[grid][2]
__global__ void __launch_bounds__(kBlock)
    k_sumsq2(long n, const double* __restrict__ a, const double* __restrict__ b,
             double* __restrict__ part) {
  __shared__ double smem[4 * 2];
  double acc[2] = {0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    acc[0] += a[i] * a[i];
    acc[1] += b[i] * b[i];
  }
  block_sum<2>(acc, smem);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = acc[0];
    part[blockIdx.x * 2 + 1] = acc[1];
  }
}

// Device-side stopping test of colmap::LeastAbsoluteDeviationSolver (primal / dual residual against
// eps_pri / eps_dual): single block.  part = [nb][4] partial sums of k_admm_edge, gat_s / gat_t =
// A'^T (z - z_old) and A'^T u.  Raises *stop when converged, else counts the iteration.
__global__ void __launch_bounds__(1024)
    k_admm_check(const double* __restrict__ part, int nb, const double* __restrict__ gat_s,
                 const double* __restrict__ gat_t, int n3, double rows, double abs_tol, double rel_tol, double rho,
                 int* __restrict__ stop, int* __restrict__ count) {
  __shared__ double smem[16 * 6];
  if (*stop) return;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += part[4 * b + k];
  }
  for (int i = threadIdx.x; i < n3; i += blockDim.x) {
    acc[4] += gat_s[i] * gat_s[i];
    acc[5] += gat_t[i] * gat_t[i];
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) smem[wave * 6 + k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double h[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w)
      for (int k = 0; k < 6; ++k) h[k] += smem[w * 6 + k];
    const double r_norm = sqrt(h[0]), Ax_norm = sqrt(h[1]), z_norm = sqrt(h[2]), b_norm = sqrt(h[3]);
    const double s_norm = rho * sqrt(h[4]), dual_norm = rho * sqrt(h[5]);
    const double primal_eps = sqrt(rows) * abs_tol + rel_tol * fmax(Ax_norm, fmax(z_norm, b_norm));
    const double dual_eps = sqrt((double)n3) * abs_tol + rel_tol * dual_norm;
    *count += 1;
    if (r_norm < primal_eps && s_norm < dual_eps) *stop = 1;
  }
}

// Single-block finaliser: out[k] = sum_b part[b*K + k]  (fixed order).
template <int K>
__global__ void __launch_bounds__(kBlock)
    k_finalize(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
  __shared__ double smem[4 * K + K];
  double tot[K];
  reduce_partials<K>(part, nblocks, tot, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = tot[k];
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct RaWs {
  DevBuf<int> ei, ej, rowptr, inc, nbr, inc_row, flags;
  DevBuf<unsigned char> grav;
  DevBuf<double> dense_a, dense_b, dense_pinv, rot_out;
  DevBuf<float> bd_inv;
  DevBuf<int> order;
  // substructured preconditioner (ra_sub.hpp)
  DevBuf<double> sub_a, sub_b, sub_pinv, sub_work, sub_y, sub_t;
  DevBuf<int> sub_ioff, sub_ipad, sub_rowblk, sub_tz, sub_coff, sub_goff, sub_cloc, sub_gloc, sub_eoff, sub_ec, sub_eg, sub_eslot;
  DevBuf<long> sub_invoff, sub_woff;
  DevBuf<unsigned char> sub_inc;
  DevBuf<double> eq, ew, inc_w, lap_diag, lap_diag_loc, rot, nq, res, wirls, z, u, dz, rhs, x, r, wbuf,
      gat_s, gat_t, fixed_rot0, part_misc, z2, u2, scal, cg_b, cg_x, cg_r, cg_z, cg_p, cg_s, cg_w, cg_minv, vpart, dpart;
  DevBuf<CgStatus> cgst;
  DevBuf<CgScal> cgsc;
  static void destroy(void* p) { delete static_cast<RaWs*>(p); }
};

RaWs* ra_ws(gsfm_ctx* ctx) {
  if (!ctx->ra_ws) {
    ctx->ra_ws = new RaWs();
    ctx->ra_ws_free = &RaWs::destroy;
  }
  return static_cast<RaWs*>(ctx->ra_ws);
}

// Maximum spanning tree on #inliers + BFS propagation (gra.cc:87-138, tree.cc:78-153).
// The root (node 0 = first registered image) gets the identity: the reference never assigns
// cam_from_worlds[root], so it stays the default-constructed Rigid3d.
void mst_init(int N, long E, const int* ei, const int* ej, const double* eq, const int* ninl,
              double* rot /* [N][3] in/out */, int root = 0) {
  // edges by descending inlier count, ties in input order: counting sort when the value range is
  // moderate (inlier counts are), comparison sort otherwise
  std::vector<long> order(E);
  int lo = E ? ninl[0] : 0, hi = lo;
  for (long e = 0; e < E; ++e) {
    lo = std::min(lo, ninl[e]);
    hi = std::max(hi, ninl[e]);
  }
  const long range = (long)hi - lo + 1;
  if (range <= (1L << 22)) {
    std::vector<long> start(range + 1, 0);
    for (long e = 0; e < E; ++e) start[hi - ninl[e] + 1]++;
    for (long v = 0; v < range; ++v) start[v + 1] += start[v];
    for (long e = 0; e < E; ++e) order[start[hi - ninl[e]]++] = e;
  } else {
    std::iota(order.begin(), order.end(), 0L);
    std::stable_sort(order.begin(), order.end(), [&](long a, long b) { return ninl[a] > ninl[b]; });
  }
  std::vector<int> parent(N);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int v) {
    while (parent[v] != v) {
      parent[v] = parent[parent[v]];
      v = parent[v];
    }
    return v;
  };
  std::vector<std::vector<std::pair<int, long>>> adj(N);
  int tree_edges = 0;
  for (long e : order) {
    const int a = ei[e], b = ej[e];
    const int ra = find(a), rb = find(b);
    if (ra != rb) {
      parent[ra] = rb;
      adj[a].emplace_back(b, e);
      adj[b].emplace_back(a, e);
      // spanning: every later edge joins two nodes of the one component and would be rejected (a 10 k / 500 k view graph is
      // spanned after ~ 10 % of its edges; disconnected graphs never get here and are scanned to the end as before)
      if (++tree_edges == N - 1) break;
    }
  }
  struct Q {
    double w, x, y, z;
  };
  auto mul = [](const Q& a, const Q& b) {
    return Q{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  };
  std::vector<Q> q(N, Q{1, 0, 0, 0});
  std::vector<char> vis(N, 0);
  std::vector<int> queue;
  queue.reserve(N);
  queue.push_back(root);
  vis[root] = 1;
  for (size_t h = 0; h < queue.size(); ++h) {
    const int cur = queue[h];
    for (auto [nb, e] : adj[cur]) {
      if (vis[nb]) continue;
      vis[nb] = 1;
      Q qr{eq[4 * e], eq[4 * e + 1], eq[4 * e + 2], eq[4 * e + 3]};
      const double nrm = std::sqrt(qr.w * qr.w + qr.x * qr.x + qr.y * qr.y + qr.z * qr.z);
      qr = Q{qr.w / nrm, qr.x / nrm, qr.y / nrm, qr.z / nrm};
      if (ei[e] == nb) {  // nb is image_id1: 1_R_w = 2_R_1^T * 2_R_w
        q[nb] = mul(Q{qr.w, -qr.x, -qr.y, -qr.z}, q[cur]);
      } else {  // 2_R_w = 2_R_1 * 1_R_w
        q[nb] = mul(qr, q[cur]);
      }
      queue.push_back(nb);
    }
  }
  for (int n = 0; n < N; ++n) {
    if (!vis[n]) continue;  // other components keep their input
    const Q& a = q[n];
    const double vn = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (vn > 0) {
      const double ang = 2.0 * std::atan2(vn, std::fabs(a.w));
      const double k = (a.w < 0 ? -ang : ang) / vn;
      rot[3 * n] = k * a.x;
      rot[3 * n + 1] = k * a.y;
      rot[3 * n + 2] = k * a.z;
    } else {
      rot[3 * n] = rot[3 * n + 1] = rot[3 * n + 2] = 0.0;
    }
  }
}

struct RaDevice {
  gsfm_ctx* ctx;
  RaWs* ws;
  int N;
  long E;
  int fixed;
  int has_gauge;
  int lpr;
  const double* ew;  // device edge weights or nullptr
  const unsigned char* grav = nullptr;  // use_gravity: [N] frame has gravity (1-DoF unknown), else nullptr
  double rows_edges = 0.0, cols = 0.0;  // rows of A contributed by this rank's edges; columns of A
  int gauge_rows = 3;
  int gridN, gridE, gridRow;
  // direct (dense) solves for small graphs, ra_dense.hpp
  bool dense = false;
  bool dense_valid = false;   // dense_inv holds the inverse of the current weighted Laplacian
  bool dense_have = false;    // dense_inv holds the inverse of SOME earlier weighted Laplacian (preconditioner)
  bool dense_refresh = false; // the stale inverse needed too many PCG iterations: re-invert at the next solve
  // 2048 < N <= kBlockDenseMaxN, one rank: PCG preconditioned by dense inverses of index-contiguous diagonal blocks;
  // the node ids of the whole solve are BFS positions (ws->order maps them back)
  bool blockdense = false;
  bool bd_have = false, bd_refresh = false;
  bool bd_fresh = false;   // the factored blocks belong to the weights currently in inc_w / lap_diag
  int bd_base_iters = 0;   // iterations of the first cold solve with a matching preconditioner (stale budget = 2.5x)
  int bd_skip_stale = 0;   // after a stale preconditioner ran out of budget: re-invert directly for this many solves
  int bd_nb = 0, bd_nblk = 0;
  // ... with the interface between the blocks eliminated exactly (ra_sub.hpp) when it is small enough
  bool sub = false;
  SubPlan plan;
  SubDev sd;
  SubCouple sc;
  bool dense_always_factor = false;  // GSFM_RA_DENSE_REFACTOR=1: re-invert for every new weighting (the round-1 behaviour)
  int Np = 0, T = 0;          // padded size, tiles per side
  double* dense_inv = nullptr;
};

template <typename F>
void dispatch_lpr(int lpr, F&& f) {
  switch (lpr) {
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    case 16: f(std::integral_constant<int, 16>{}); break;
    case 32: f(std::integral_constant<int, 32>{}); break;
    default: f(std::integral_constant<int, 64>{}); break;
  }
}

int choose_lpr(long E, int N) {
  const double avg_deg = N > 0 ? 2.0 * (double)E / N : 0.0;
  if (avg_deg >= 48) return 64;
  if (avg_deg >= 24) return 32;
  if (avg_deg >= 12) return 16;
  if (avg_deg >= 6) return 8;
  return 4;
}

// Builds the CSR-by-node incidence structure on the host (counting sort, O(E)) and uploads it.
void build_incidence(RaDevice& d, const int* h_ei, const int* h_ej, std::vector<int>* keep_rowptr = nullptr,
                     std::vector<int>* keep_nbr = nullptr) {
  const int N = d.N;
  const long E = d.E;
  GSFM_REQUIRE(2 * E < (1L << 31), "RA: 2*num_edges must fit int32");
  std::vector<int> rowptr(N + 1, 0);
  for (long e = 0; e < E; ++e) {
    GSFM_REQUIRE(h_ei[e] >= 0 && h_ei[e] < N && h_ej[e] >= 0 && h_ej[e] < N, "RA: edge index out of range");
    rowptr[h_ei[e] + 1]++;
    rowptr[h_ej[e] + 1]++;
  }
  for (int n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
  std::vector<int> fill(rowptr.begin(), rowptr.end() - 1);
  std::vector<int> inc(2 * E), nbr(2 * E), inc_row(2 * E);
  for (long e = 0; e < E; ++e) {
    const int i = h_ei[e], j = h_ej[e];
    int k = fill[i]++;
    inc[k] = (int)(e << 1);  // node is image_id1: -I3
    nbr[k] = j;
    inc_row[k] = i;
    k = fill[j]++;
    inc[k] = (int)(e << 1) | 1;  // node is image_id2: +I3
    nbr[k] = i;
    inc_row[k] = j;
  }
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  GSFM_HIP_CHECK(hipMemcpyAsync(ws->rowptr.ensure(N + 1), rowptr.data(), (N + 1) * sizeof(int), hipMemcpyHostToDevice, s));
  GSFM_HIP_CHECK(hipMemcpyAsync(ws->inc.ensure(2 * E + 1), inc.data(), 2 * E * sizeof(int), hipMemcpyHostToDevice, s));
  GSFM_HIP_CHECK(hipMemcpyAsync(ws->nbr.ensure(2 * E + 1), nbr.data(), 2 * E * sizeof(int), hipMemcpyHostToDevice, s));
  if (d.dense || d.blockdense)
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->inc_row.ensure(2 * E + 1), inc_row.data(), 2 * E * sizeof(int), hipMemcpyHostToDevice, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));  // host vectors go out of scope
  if (keep_rowptr) keep_rowptr->swap(rowptr);
  if (keep_nbr) keep_nbr->swap(nbr);
}

// Solves (L_w (x) I3 + gauge) x = rhs with the weights currently in ws->inc_w / lap_diag.
// warm: keep the current contents of ws->x as initial guess.  Returns PCG iterations.
// (L_w + gauge)^-1 by tiled Gauss-Jordan (ra_dense.hpp): T launches of a T x T grid.
void dense_factor(RaDevice& d) {
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  const int Np = d.Np, T = d.T;
  const size_t nn = (size_t)Np * Np;
  double* cur = ws->dense_a.ensure(nn);
  double* oth = ws->dense_b.ensure(nn);
  double* pinv = ws->dense_pinv.ensure(2 * kTile * kTile);
  GSFM_HIP_CHECK(hipMemsetAsync(cur, 0, nn * sizeof(double), s));
  hipLaunchKernelGGL(k_dense_fill_offdiag, dim3(grid_for(2 * d.E, kBlock)), dim3(kBlock), 0, s, 2 * d.E,
                     ws->inc_row.get(), ws->nbr.get(), ws->inc_w.get(), Np, cur);
  hipLaunchKernelGGL(k_dense_fill_diag, dim3(grid_for(Np, kBlock)), dim3(kBlock), 0, s, d.N, Np, ws->lap_diag.get(), Np, cur);
  hipLaunchKernelGGL(k_gj_pivot0, dim3(1), dim3(kBlock), 0, s, cur, Np, (size_t)0, pinv);
  for (int k = 0; k < T; ++k) {
    const bool timed = d.ctx->prof.begin(s, GSFM_KERNEL_RA_GJ);
    hipLaunchKernelGGL(k_gj_sweep_step, dim3(gj_tiles(T)), dim3(kBlock), 0, s, cur, oth, Np, (size_t)0, (const int*)nullptr, T, k, pinv);
    if (timed) d.ctx->prof.end(s);
    std::swap(cur, oth);
  }
  hipLaunchKernelGGL(k_gj_finish_full, dim3(grid_wide(nn, kBlock, 1 << 12)), dim3(kBlock), 0, s, cur, Np, Np);
  if (d.ctx->prof.enabled) {
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    d.ctx->prof.harvest();
  }
  d.dense_inv = cur;
  d.dense_valid = true;
  d.dense_have = true;
  d.dense_refresh = false;
}

// x = A^-1 rhs, then (refine) one step of iterative refinement against the sparse operator.
int dense_solve(RaDevice& d, bool refine = true, const int* stop = nullptr) {
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  const int N = d.N;
  if (!d.dense_valid) dense_factor(d);
  const int gA = grid_for(N, kBlock / 64);
  hipLaunchKernelGGL(k_dense_apply3, dim3(gA), dim3(kBlock), 0, s, N, d.Np, d.dense_inv, ws->rhs.get(), ws->x.get(), 0, stop);
  if (!refine) return 1;
  dispatch_lpr(d.lpr, [&](auto L) {
    hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, N, ws->rowptr.get(),
                       ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), ws->x.get(), ws->wbuf.get());
  });
  hipLaunchKernelGGL(k_dense_residual, dim3(d.gridN), dim3(kBlock), 0, s, 3L * N, ws->rhs.get(), ws->wbuf.get(), ws->r.get());
  hipLaunchKernelGGL(k_dense_apply3, dim3(gA), dim3(kBlock), 0, s, N, d.Np, d.dense_inv, ws->r.get(), ws->x.get(), 1, stop);
  return 1;
}

// (L_w + gauge) x = rhs by PCG with the stale dense inverse as preconditioner (ra_dense.hpp).  Returns the number
// of iterations, or -1 when it did not converge (the caller re-inverts).
int dense_pcg_solve(RaDevice& d, double tol) {
  RaWs* ws = d.ws;
  gsfm_ctx* ctx = d.ctx;
  hipStream_t s = ctx->stream;
  const int N = d.N, n3 = 3 * N;
  DpcgState* st = reinterpret_cast<DpcgState*>(ws->cgst.get());
  double* u = ws->cg_z.get();
  double* w = ws->cg_w.get();
  hipLaunchKernelGGL(k_dpcg_init, dim3(1), dim3(1024), 0, s, n3, ws->rhs.get(), ws->x.get(), ws->cg_r.get(), ws->cg_p.get(),
                     ws->cg_s.get(), st);
  const int gA = grid_for(N, kBlock / 64);
  constexpr int kBatch = 10, kMaxIters = 40;
  DpcgState h;
  for (int done = 0; done < kMaxIters; done += kBatch) {
    for (int it = 0; it < kBatch; ++it) {
      hipLaunchKernelGGL(k_dense_apply3, dim3(gA), dim3(kBlock), 0, s, N, d.Np, d.dense_inv, ws->cg_r.get(), u, 0, &st->done);
      dispatch_lpr(d.lpr, [&](auto L) {
        hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, N, ws->rowptr.get(),
                           ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), u, w);
      });
      hipLaunchKernelGGL(k_dpcg_update, dim3(1), dim3(1024), 0, s, n3, tol * tol, u, w, ws->x.get(), ws->cg_r.get(),
                         ws->cg_p.get(), ws->cg_s.get(), st);
    }
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 96, st, sizeof(DpcgState), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    std::memcpy(&h, ctx->h_pinned + 96, sizeof(DpcgState));
    if (h.done) break;
  }
  if (!h.done || h.bad) return -1;
  if (h.iters > 24) d.dense_refresh = true;  // the weights have drifted: refresh the preconditioner next time
  return h.iters;
}

// ---- substructured preconditioner (ra_sub.hpp): tables, factorisation --------------------------------------------------
template <typename T>
const T* sub_up(gsfm_ctx* ctx, DevBuf<T>& buf, const std::vector<T>& h) {
  T* p = buf.ensure(h.size() + 1);
  if (!h.empty()) GSFM_HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return p;
}

// Coupling tables from the final CSR, everything to the device.  Falls back to the plain block preconditioner (d.sub =
// false; the node order stays — it is a BFS order with the interface moved to the end) when the coupling is too dense.
void sub_upload(RaDevice& d, const std::vector<int>& h_rowptr, const std::vector<int>& h_nbr) {
  RaWs* ws = d.ws;
  gsfm_ctx* ctx = d.ctx;
  SubPlan& sp = d.plan;
  if (!sub_build_coupling(d.N, h_rowptr.data(), h_nbr.data(), sp)) {
    d.sub = false;
    return;
  }
  std::vector<int> tz(sp.nblk);
  for (int b = 0; b < sp.nblk; ++b) tz[b] = sp.ipad[b] / kTile;
  d.sd.N = d.N;
  d.sd.NI = sp.NI;
  d.sd.nblk = sp.nblk;
  d.sd.ioff = sub_up(ctx, ws->sub_ioff, sp.ioff);
  d.sd.ipad = sub_up(ctx, ws->sub_ipad, sp.ipad);
  d.sd.invoff = sub_up(ctx, ws->sub_invoff, sp.invoff);
  d.sd.row_blk = sub_up(ctx, ws->sub_rowblk, sp.row_blk);
  d.sd.in_c = sub_up(ctx, ws->sub_inc, sp.in_c);
  d.sd.inv = ws->bd_inv.ensure((size_t)sp.inv_total);
  sub_up(ctx, ws->sub_tz, tz);
  d.sc.coff = sub_up(ctx, ws->sub_coff, sp.coff);
  d.sc.goff = sub_up(ctx, ws->sub_goff, sp.goff);
  d.sc.cloc = sub_up(ctx, ws->sub_cloc, sp.cloc);
  d.sc.gloc = sub_up(ctx, ws->sub_gloc, sp.gloc);
  d.sc.eoff = sub_up(ctx, ws->sub_eoff, sp.eoff);
  d.sc.ec = sub_up(ctx, ws->sub_ec, sp.ec);
  d.sc.eg = sub_up(ctx, ws->sub_eg, sp.eg);
  d.sc.eslot = sub_up(ctx, ws->sub_eslot, sp.eslot);
  d.sc.woff = sub_up(ctx, ws->sub_woff, sp.woff);
  d.sc.work = ws->sub_work.ensure((size_t)sp.work_total + 1);
  ws->sub_y.ensure(3 * (size_t)d.N);
  ws->sub_t.ensure(3 * (size_t)d.N);
  GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the host tables may go away
  static const bool trace = getenv("GSFM_RA_TRACE") != nullptr;
  if (trace)
    fprintf(stderr, "[ra] substructured: %d interiors (largest %d padded) + interface %d, boundary layers <= %d, adjacent interface <= %d\n",
            sp.nblk, sp.pmax, sp.G, sp.cmax, sp.gmax);
}

// Inverts every interior block of the CURRENT weighted Laplacian in one batched Gauss-Jordan sweep, assembles the Schur
// complement of the interface from the boundary layers and inverts it.
void sub_factor(RaDevice& d) {
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  const SubPlan& sp = d.plan;
  const int nblk = sp.nblk, ld = sp.pmax, Tmax = ld / kTile;
  const int Gp = sp.ipad[nblk], Tg = Gp / kTile;
  const size_t zs = (size_t)ld * ld;
  const size_t need = std::max(zs * nblk, (size_t)Gp * Gp);
  double* cur = ws->sub_a.ensure(need);
  double* oth = ws->sub_b.ensure(need);
  double* Sa = ws->dense_a.ensure((size_t)Gp * Gp);
  double* Sb = ws->dense_b.ensure((size_t)Gp * Gp);
  double* pinv = ws->sub_pinv.ensure((size_t)(nblk + 1) * 2 * kTile * kTile);
  GSFM_HIP_CHECK(hipMemsetAsync(cur, 0, zs * nblk * sizeof(double), s));
  GSFM_HIP_CHECK(hipMemsetAsync(Sa, 0, (size_t)Gp * Gp * sizeof(double), s));
  GSFM_HIP_CHECK(hipMemsetAsync(d.sc.work, 0, (size_t)sp.work_total * sizeof(double), s));
  hipLaunchKernelGGL(k_sub_fill_offdiag, dim3(grid_for(2 * d.E, kBlock)), dim3(kBlock), 0, s, 2 * d.E, ws->inc_row.get(),
                     ws->nbr.get(), ws->inc_w.get(), d.sd, cur, ld, zs, Sa, Gp);
  hipLaunchKernelGGL(k_sub_fill_diag, dim3(grid_for(std::max(ld, Gp), kBlock), nblk + 1), dim3(kBlock), 0, s, d.sd,
                     ws->lap_diag.get(), cur, ld, zs, Sa, Gp);
  // interiors
  hipLaunchKernelGGL(k_gj_pivot0, dim3(1, 1, nblk), dim3(kBlock), 0, s, cur, ld, zs, pinv);
  for (int k = 0; k < Tmax; ++k) {
    const bool timed = d.ctx->prof.begin(s, GSFM_KERNEL_RA_GJ);
    hipLaunchKernelGGL(k_gj_sweep_step, dim3(gj_tiles(Tmax), 1, nblk), dim3(kBlock), 0, s, cur, oth, ld, zs,
                       (const int*)ws->sub_tz.get(), Tmax, k, pinv);
    if (timed) d.ctx->prof.end(s);
    std::swap(cur, oth);
  }
  hipLaunchKernelGGL(k_sub_to_f32, dim3(grid_wide(zs, kBlock, 1 << 12), nblk), dim3(kBlock), 0, s, d.sd, 0, cur, ld, zs,
                     ws->bd_inv.get());
  // Schur complement of the interface: S = A_GG - sum_b E_b^T (A_bb^-1)[C_b, C_b] E_b
  const int gEM = grid_for((size_t)std::max((long)sp.cmax * sp.cmax, (long)(sp.eoff[nblk] / std::max(1, nblk) + 1)), kBlock);
  hipLaunchKernelGGL(k_sub_couple_EM, dim3(gEM, nblk), dim3(kBlock), 0, s, d.sd, d.sc, ws->inc_w.get(), cur, ld, zs);
  hipLaunchKernelGGL(k_sub_couple_Y, dim3(grid_for((size_t)sp.cmax * sp.gmax, kBlock), nblk), dim3(kBlock), 0, s, d.sc);
  hipLaunchKernelGGL(k_sub_couple_S, dim3(grid_for((size_t)sp.gmax * sp.gmax, kBlock), nblk), dim3(kBlock), 0, s, d.sc, Sa, Gp);
  double* pS = pinv + (size_t)nblk * 2 * kTile * kTile;
  hipLaunchKernelGGL(k_gj_pivot0, dim3(1), dim3(kBlock), 0, s, Sa, Gp, (size_t)0, pS);
  for (int k = 0; k < Tg; ++k) {
    hipLaunchKernelGGL(k_gj_sweep_step, dim3(gj_tiles(Tg)), dim3(kBlock), 0, s, Sa, Sb, Gp, (size_t)0, (const int*)nullptr, Tg, k, pS);
    std::swap(Sa, Sb);
  }
  hipLaunchKernelGGL(k_sub_to_f32, dim3(grid_wide((size_t)Gp * Gp, kBlock, 1 << 12), 1), dim3(kBlock), 0, s, d.sd, nblk, Sa, Gp,
                     (size_t)0, ws->bd_inv.get());
  if (d.ctx->prof.enabled) {
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    d.ctx->prof.harvest();
  }
  d.bd_have = true;
  d.bd_refresh = false;
  d.bd_fresh = true;
  d.bd_base_iters = 0;
}

// u = M^-1 r with the substructured preconditioner: two sweeps over the interior inverses around the interface solve
void sub_apply(RaDevice& d, const double* r, double* u, int it, double tol2, const double* rpart, DpcgState* st) {
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  const int N = d.N, NI = d.plan.NI;
  const int gI = grid_wide(NI, kBlock / 64, 1 << 14), gG = grid_wide(N - NI, kBlock / 64, 1 << 14);
  double* y = ws->sub_y.get();
  double* t = ws->sub_t.get();
  hipLaunchKernelGGL(k_sub_apply3, dim3(gI), dim3(kBlock), 0, s, d.sd, 0, NI, r, y, 1, it, tol2, rpart, st);        // y_I = A_II^-1 r_I
  hipLaunchKernelGGL((k_sub_couple<true>), dim3(gG), dim3(kBlock), 0, s, d.sd, ws->rowptr.get(), ws->nbr.get(),
                     ws->inc_w.get(), r, (const double*)y, t, (const DpcgState*)st);                                        // t_G = r_G - A_GI y_I
  hipLaunchKernelGGL(k_sub_apply3, dim3(gG), dim3(kBlock), 0, s, d.sd, NI, N, (const double*)t, u, 0, it, tol2, rpart, st);  // u_G = S^-1 t_G
  hipLaunchKernelGGL((k_sub_couple<false>), dim3(gI), dim3(kBlock), 0, s, d.sd, ws->rowptr.get(), ws->nbr.get(),
                     ws->inc_w.get(), r, (const double*)u, t, (const DpcgState*)st);                                        // t_I = r_I - A_IG u_G
  hipLaunchKernelGGL(k_sub_apply3, dim3(gI), dim3(kBlock), 0, s, d.sd, 0, NI, (const double*)t, u, 0, it, tol2, rpart, st);  // u_I = A_II^-1 t_I
}

// Inverts the diagonal blocks of the CURRENT weighted Laplacian (ra_dense.hpp): all blocks through one batched sweep.
void bd_factor(RaDevice& d) {
  if (d.sub) return sub_factor(d);
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  const int nb = d.bd_nb, T = nb / kTile, nblk = d.bd_nblk;
  const size_t nn = (size_t)nb * nb;
  float* inv = ws->bd_inv.ensure((size_t)nblk * nn);
  double* pinv = ws->sub_pinv.ensure((size_t)nblk * 2 * kTile * kTile);
  double* cur = ws->sub_a.ensure(nn * nblk);
  double* oth = ws->sub_b.ensure(nn * nblk);
  GSFM_HIP_CHECK(hipMemsetAsync(cur, 0, nn * nblk * sizeof(double), s));
  for (int b = 0; b < nblk; ++b) {
    hipLaunchKernelGGL(k_bd_fill_offdiag, dim3(grid_for(2 * d.E, kBlock)), dim3(kBlock), 0, s, 2 * d.E, ws->inc_row.get(),
                       ws->nbr.get(), ws->inc_w.get(), b * nb, nb, cur + nn * b);
    hipLaunchKernelGGL(k_bd_fill_diag, dim3(grid_for(nb, kBlock)), dim3(kBlock), 0, s, d.N, b * nb, nb, ws->lap_diag.get(), cur + nn * b);
  }
  // all blocks through one batched symmetric sweep
  hipLaunchKernelGGL(k_gj_pivot0, dim3(1, 1, nblk), dim3(kBlock), 0, s, cur, nb, nn, pinv);
  for (int k = 0; k < T; ++k) {
    hipLaunchKernelGGL(k_gj_sweep_step, dim3(gj_tiles(T), 1, nblk), dim3(kBlock), 0, s, cur, oth, nb, nn, (const int*)nullptr, T, k, pinv);
    std::swap(cur, oth);
  }
  hipLaunchKernelGGL(k_bd_to_f32, dim3(grid_wide(nn, kBlock, 1 << 12), nblk), dim3(kBlock), 0, s, nb, cur, nn, inv);
  d.bd_have = true;
  d.bd_refresh = false;
  d.bd_fresh = true;
  d.bd_base_iters = 0;
}

// PCG with the block-diagonal dense preconditioner; same contract as the Jacobi path of pcg_solve.
// A preconditioner factored from EARLIER weights is tried first (the L1 stage never changes its weights, IRLS changes
// them little once it is converging); when that takes more than ~2.5x the iterations of a matching preconditioner —
// the first IRLS systems, whose weights of a few nodes differ by orders of magnitude from the L1 ones, do — the solve is
// abandoned, the blocks are re-inverted with the current weights and the solve restarts.
int bd_pcg_solve(RaDevice& d, bool warm, double tol, int max_iter) {
  RaWs* ws = d.ws;
  gsfm_ctx* ctx = d.ctx;
  hipStream_t s = ctx->stream;
  const int N = d.N, n3 = 3 * N;
  static const bool trace = getenv("GSFM_RA_TRACE") != nullptr;
  if (!d.bd_fresh && d.bd_skip_stale > 0) {  // the weights are still moving fast (the last stale attempt failed)
    --d.bd_skip_stale;
    d.bd_refresh = true;
  }
  // the substructured factorisation is cheap and only pays when it matches the weights (kept through the IRLS weights it
  // needs 1 140 instead of 11 iterations, tools/exp_ra_linear_solves.py): always re-factor for a new weighting
  if (d.sub && !d.bd_fresh) d.bd_refresh = true;
  if (!d.bd_have || d.bd_refresh) bd_factor(d);
  const double* b = ws->rhs.get();
  double* x = ws->x.get();
  if (warm) {  // solve A dx = rhs - A x0 and add (the ADMM iterates of one L1 solve are close)
    dispatch_lpr(d.lpr, [&](auto L) {
      hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, N, ws->rowptr.get(), ws->nbr.get(),
                         ws->inc_w.get(), ws->lap_diag_loc.get(), ws->x.get(), ws->wbuf.get());
    });
    hipLaunchKernelGGL(k_dense_residual, dim3(d.gridN), dim3(kBlock), 0, s, (long)n3, ws->rhs.get(), ws->wbuf.get(), ws->cg_b.get());
    b = ws->cg_b.get();
    x = ws->cg_x.get();
  }
  DpcgState* st = reinterpret_cast<DpcgState*>(ws->cgst.get());
  BdScal* scal = reinterpret_cast<BdScal*>(ws->cgsc.get());
  double* u = ws->cg_z.get();
  double* w = ws->cg_w.get();
  double* rpart = ws->vpart.get();
  double* dpart = ws->dpart.get();
  const int gA = grid_wide(N, kBlock / 64, 1 << 14);
  const int gS = std::min(d.gridRow, kMaxApplySlots / 2);
  // iterations enqueued per status read-back: the substructured preconditioner converges in two or three
  const int batch = d.sub ? 3 : (tol > 1e-6 ? 6 : 12);  // (inexact ADMM solves stop after a handful of iterations)
  DpcgState h{};
  int total = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // a stale preconditioner gets a bounded budget; a fresh one the caller's
    const int budget = d.bd_fresh ? max_iter : std::min(max_iter, std::max(40, (5 * d.bd_base_iters) / 2));
    hipLaunchKernelGGL(k_dpcg_init_mb, dim3(kBdUpdateBlocks), dim3(kBlock), 0, s, n3, b, x, ws->cg_r.get(), ws->cg_p.get(),
                       ws->cg_s.get(), rpart);
    hipLaunchKernelGGL(k_dpcg_init_fin, dim3(1), dim3(64), 0, s, (const double*)rpart, kBdUpdateBlocks, st);
    h = DpcgState{};
    for (int done = 0; done < budget && !h.done; done += batch) {
      for (int it = done; it < done + batch; ++it) {
        if (d.sub)
          sub_apply(d, ws->cg_r.get(), u, it, tol * tol, rpart, st);
        else
          hipLaunchKernelGGL(k_bd_apply3, dim3(gA), dim3(kBlock), 0, s, N, d.bd_nb, ws->bd_inv.get(), ws->cg_r.get(), u, it,
                             tol * tol, rpart, st);
        const bool timed = ctx->prof.begin(s, GSFM_KERNEL_RA_LAPLACIAN);
        dispatch_lpr(d.lpr, [&](auto L) {
          hipLaunchKernelGGL((k_bd_spmv<decltype(L)::value>), dim3(gS), dim3(kBlock), 0, s, N, ws->rowptr.get(), ws->nbr.get(),
                             ws->inc_w.get(), ws->lap_diag_loc.get(), u, ws->cg_r.get(), w, dpart, st);
        });
        if (timed) ctx->prof.end(s);
        hipLaunchKernelGGL(k_bd_update, dim3(kBdUpdateBlocks), dim3(kBlock), 0, s, n3, gS, dpart, u, w, x, ws->cg_r.get(),
                           ws->cg_p.get(), ws->cg_s.get(), it, scal + (it & 1), scal + ((it + 1) & 1), rpart, st);
      }
      // one more convergence test for the last update of the batch (k_bd_apply3 of the next iteration would do it)
      GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 96, st, sizeof(DpcgState), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 104, rpart, kBdUpdateBlocks * sizeof(double), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      std::memcpy(&h, ctx->h_pinned + 96, sizeof(DpcgState));
      if (!h.done) {
        double rr = 0.0;
        for (int i = 0; i < kBdUpdateBlocks; ++i) rr += ctx->h_pinned[104 + i];
        if (rr <= tol * tol * h.bb) {
          h.done = 1;
          h.rr = rr;
        }
      }
    }
    total += h.iters;
    if (trace)
      fprintf(stderr, "[ra] bd_pcg warm=%d tol=%.1e fresh=%d iters=%d done=%d bad=%d rr/bb=%.3e\n", (int)warm, tol, (int)d.bd_fresh,
              h.iters, h.done, h.bad, h.bb > 0 ? h.rr / h.bb : 0.0);
    if (h.done && !h.bad) {
      if (d.bd_fresh && !warm && d.bd_base_iters == 0) d.bd_base_iters = std::max(h.iters, 1);
      break;
    }
    if (d.bd_fresh) break;  // a matching preconditioner did not converge within the caller's limit: report as is
    bd_factor(d);           // stale preconditioner out of budget: re-invert with the current weights and restart
    d.bd_skip_stale = 2;
  }
  if (warm) hipLaunchKernelGGL(k_ra_add, dim3(d.gridN), dim3(kBlock), 0, s, (long)n3, ws->x.get(), ws->cg_x.get(), ws->x.get());
  return total;
}

int pcg_solve(RaDevice& d, bool warm, double tol, int max_iter) {
  if (d.dense) {
    // a stale inverse (same graph, earlier weights) preconditions the new system; fall back to a fresh inversion
    if (!d.dense_valid && d.dense_have && !d.dense_refresh && !d.dense_always_factor) {
      const int it = dense_pcg_solve(d, tol);
      if (it >= 0) return it;
      d.dense_refresh = true;
    }
    return dense_solve(d);
  }
  if (d.blockdense) return bd_pcg_solve(d, warm, tol, max_iter);
  RaWs* ws = d.ws;
  gsfm_ctx* ctx = d.ctx;
  hipStream_t s = ctx->stream;
  const int N = d.N;
  const long n3 = 3L * N;
  // warm start: solve A dx = rhs - A x0 and add (the ADMM iterates of one L1 solve are close)
  const double* b = ws->rhs.get();
  if (warm) {
    dispatch_lpr(d.lpr, [&](auto L) {
      hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, N, ws->rowptr.get(),
                         ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), ws->x.get(), ws->wbuf.get());
    });
    allreduce_sum(ctx, ws->wbuf.get(), (size_t)n3);
    if (d.grav) hipLaunchKernelGGL(k_ra_mask3, dim3(d.gridN), dim3(kBlock), 0, s, N, d.grav, ws->wbuf.get(), (double*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(k_dense_residual, dim3(d.gridN), dim3(kBlock), 0, s, n3, ws->rhs.get(), ws->wbuf.get(), ws->cg_b.get());
    b = ws->cg_b.get();
  }
  hipLaunchKernelGGL(k_ra_minv, dim3(d.gridN), dim3(kBlock), 0, s, N, ws->lap_diag.get(), ws->cg_minv.get());
  CgVec v;
  v.n = (int)n3;
  v.N = N;
  v.K = 0;
  v.nb_update = std::min(kCgUpdateBlocks, grid_for(N, kBlock));
  const int gA = grid_wide(N, kBlock / d.lpr, kMaxApplySlots);
  v.nb_apply = gA;
  v.b = b;
  v.x = warm ? ws->cg_x.get() : ws->x.get();
  v.r = ws->cg_r.get();
  v.z = ws->cg_z.get();
  v.p = ws->cg_p.get();
  v.s = ws->cg_s.get();
  v.w = ws->cg_w.get();
  v.minv = ws->cg_minv.get();
  v.vpart = ws->vpart.get();
  v.dpart = ws->dpart.get();
  v.scal = ws->cgsc.get();
  v.st = ws->cgst.get();
  auto run_cg = [&](double tol_pass) {
    return cg_solve<3, false>(ctx, v, tol_pass, max_iter, [&](int it) {
      const bool timed = ctx->prof.begin(s, GSFM_KERNEL_RA_LAPLACIAN);
      dispatch_lpr(d.lpr, [&](auto L) {
        hipLaunchKernelGGL((k_ra_apply<decltype(L)::value>), dim3(gA), dim3(kBlock), 0, s, N, ws->rowptr.get(),
                           ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), v, it, tol_pass * tol_pass, d.grav);
      });
      if (timed) ctx->prof.end(s);
    });
  };
  long iters = run_cg(tol);
  if (warm) hipLaunchKernelGGL(k_ra_add, dim3(d.gridN), dim3(kBlock), 0, s, n3, ws->x.get(), ws->cg_x.get(), ws->x.get());
  // The recurrence residual of a long Jacobi-PCG run drifts away from b - A x (measured: ring graph, 20k nodes,
  // ~600 iterations per solve: the solves "converged" to 1e-10 while the final rotations differed from the
  // direct-solve oracle on half of the nodes).  So the TRUE residual is verified and, where it misses the tolerance,
  // the correction equation A dx = b - A x is solved and added (at most three times).
  const double* bref = b;  // what `tol` is relative to: rhs, or the warm-start residual
  for (int pass = 0; pass < 3; ++pass) {
    dispatch_lpr(d.lpr, [&](auto L) {
      hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, N, ws->rowptr.get(),
                         ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), ws->x.get(), ws->wbuf.get());
    });
    allreduce_sum(ctx, ws->wbuf.get(), (size_t)n3);
    if (d.grav) hipLaunchKernelGGL(k_ra_mask3, dim3(d.gridN), dim3(kBlock), 0, s, N, d.grav, ws->wbuf.get(), (double*)nullptr, (double*)nullptr);
    hipLaunchKernelGGL(k_dense_residual, dim3(d.gridN), dim3(kBlock), 0, s, n3, ws->rhs.get(), ws->wbuf.get(), ws->r.get());
    hipLaunchKernelGGL(k_sumsq2, dim3(d.gridN), dim3(kBlock), 0, s, n3, ws->r.get(), bref, ws->part_misc.get());
    hipLaunchKernelGGL((k_finalize<2>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), d.gridN, ws->scal.get() + 16);
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 200, ws->scal.get() + 16, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    const double rr = ctx->h_pinned[200], bb = ctx->h_pinned[201];
    if (!(rr > tol * tol * bb) || !(bb > 0.0)) break;  // also leaves on NaN
    const double tol_pass = std::min(0.1, std::max(1e-12, 0.5 * tol * std::sqrt(bb / rr)));
    v.b = ws->r.get();
    v.x = ws->cg_x.get();
    iters += run_cg(tol_pass);
    hipLaunchKernelGGL(k_ra_add, dim3(d.gridN), dim3(kBlock), 0, s, n3, ws->x.get(), ws->cg_x.get(), ws->x.get());
  }
  return (int)iters;
}

void launch_residuals(RaDevice& d, bool with_weights, int weight_type, double sigma2) {
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  hipLaunchKernelGGL(k_node_quat, dim3(d.gridN), dim3(kBlock), 0, s, d.N, ws->rot.get(), ws->nq.get(),
                     d.fixed, ws->fixed_rot0.get(), ws->res.get() + 3 * d.E, d.has_gauge);
  hipLaunchKernelGGL(k_edge_residual, dim3(d.gridE), dim3(kBlock), 0, s, d.E, ws->ei.get(), ws->ej.get(),
                     ws->eq.get(), ws->nq.get(), ws->res.get(), with_weights ? ws->wirls.get() : nullptr,
                     weight_type, sigma2, ws->flags.get(), d.grav, ws->rot.get());
}

// Applies ws->x as the tangent step, returns {mean |delta|, |delta|_2, #NaN}.
void update_rotations(RaDevice& d, double out[3]) {
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  hipLaunchKernelGGL(k_node_update, dim3(d.gridN), dim3(kBlock), 0, s, d.N, ws->rot.get(), ws->x.get(),
                     ws->part_misc.get());
  hipLaunchKernelGGL((k_finalize<3>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), d.gridN, ws->scal.get());
  GSFM_HIP_CHECK(hipMemcpyAsync(d.ctx->h_pinned + 64, ws->scal.get(), 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  out[0] = d.ctx->h_pinned[64] / d.N;
  out[1] = std::sqrt(d.ctx->h_pinned[65]);
  out[2] = d.ctx->h_pinned[66];
}

// BFS order of the view graph from `root` (unreached nodes appended): order[p] = node, pos[node] = p.
void bfs_order(int N, long E, const int* ei, const int* ej, int root, std::vector<int>& order, std::vector<int>& pos) {
  std::vector<int> rowptr(N + 1, 0);
  for (long e = 0; e < E; ++e) {
    rowptr[ei[e] + 1]++;
    rowptr[ej[e] + 1]++;
  }
  for (int n = 0; n < N; ++n) rowptr[n + 1] += rowptr[n];
  std::vector<int> fill(rowptr.begin(), rowptr.end() - 1), adj(2 * E);
  for (long e = 0; e < E; ++e) {
    adj[fill[ei[e]]++] = ej[e];
    adj[fill[ej[e]]++] = ei[e];
  }
  order.clear();
  order.reserve(N);
  pos.assign(N, -1);
  auto visit = [&](int start) {
    pos[start] = (int)order.size();
    order.push_back(start);
    for (size_t h = order.size() - 1; h < order.size(); ++h) {
      const int cur = order[h];
      for (int k = rowptr[cur]; k < rowptr[cur + 1]; ++k) {
        const int nb = adj[k];
        if (pos[nb] < 0) {
          pos[nb] = (int)order.size();
          order.push_back(nb);
        }
      }
    }
  };
  visit(root);
  for (int n = 0; n < N; ++n)
    if (pos[n] < 0) visit(n);
}

// Host-side copies that outlive setup (the rotation upload of finish_init is asynchronous).
struct RaHostInit {
  std::vector<int> h_ei, h_ej, h_ninl, pos;
  std::vector<double> h_eq, h_rot;
  int mst_root = 0;
};

// Phase 1 of the setup: everything the linear algebra needs (edges, incidence structure, workspaces) plus the
// device-to-host copies of what the initialisation needs.  The caller may enqueue the L1-stage factorisation right
// after it; finish_init then computes the maximum-spanning-tree initialisation on the host WHILE the GPU inverts.
void setup_device(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                  const double* rot_in, RaDevice& d, RaHostInit& hi, bool allow_blockdense = false) {
  RaWs* ws = ra_ws(ctx);
  const int N = prob->num_nodes;
  const long E = prob->num_edges;
  d.ctx = ctx;
  d.ws = ws;
  d.N = N;
  d.E = E;
  GSFM_REQUIRE(prob->fixed_node >= 0 && prob->fixed_node < N, "RA: fixed_node out of range");  // (every entry point lands here)
  d.fixed = prob->fixed_node;
  d.has_gauge = ctx->comm.rank == 0 ? 1 : 0;
  d.lpr = choose_lpr(E, N);
  d.grav = nullptr;
  d.rows_edges = 3.0 * (double)E;
  d.cols = 3.0 * (double)N;
  d.gauge_rows = 3;
  std::vector<unsigned char> h_grav;
  if (opt->use_gravity && prob->node_gravity != nullptr) {
    to_host(ctx, h_grav, prob->node_gravity, (size_t)N, prob->mem);
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->grav.ensure(N), h_grav.data(), (size_t)N, hipMemcpyHostToDevice, ctx->stream));
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    d.grav = ws->grav.get();
    d.cols = 0.0;
    for (int n = 0; n < N; ++n) d.cols += h_grav[n] ? 1.0 : 3.0;
    d.gauge_rows = h_grav[prob->fixed_node] ? 1 : 3;
  }
  // mixed 1-DoF / 3-DoF systems are not ONE scalar Laplacian any more: the masked Jacobi-PCG path solves them
  d.dense = d.grav == nullptr && ctx->comm.world == 1 && N <= kDenseMaxN && opt->pcg_max_iterations > 0 && !opt->force_iterative;
  d.dense_valid = false;
  d.dense_have = false;
  d.dense_refresh = false;
  d.blockdense = allow_blockdense && d.grav == nullptr && !d.dense && ctx->comm.world == 1 && N <= kBlockDenseMaxN && opt->pcg_max_iterations > 0 &&
                 !opt->force_iterative && !ctx->knob[GSFM_KNOB_RA_NO_BLOCKDENSE];
  d.bd_have = d.bd_refresh = d.bd_fresh = false;
  d.bd_base_iters = 0;
  d.bd_skip_stale = 0;
  if (d.blockdense) {
    d.bd_nblk = (N + kDenseMaxN - 1) / kDenseMaxN;
    d.bd_nb = (((N + d.bd_nblk - 1) / d.bd_nblk + kTile - 1) / kTile) * kTile;
  }
  {
    const bool always = ctx->knob[GSFM_KNOB_RA_DENSE_REFACTOR] != 0;
    d.dense_always_factor = always;
  }
  d.T = (N + kTile - 1) / kTile;
  d.Np = d.T * kTile;
  d.gridN = grid_for(N, kBlock);
  d.gridE = grid_for(E + 1, kBlock);
  d.gridRow = grid_for(N, kBlock / d.lpr);
  hipStream_t s = ctx->stream;
  const int mem = prob->mem;
  ws->nq.ensure(4 * (size_t)N);
  ws->res.ensure(3 * (size_t)(E + 1));
  ws->wirls.ensure(E + 1);
  ws->inc_w.ensure(2 * E + 1);
  ws->lap_diag.ensure(N);
  ws->lap_diag_loc.ensure(N);
  for (DevBuf<double>* b : {&ws->rhs, &ws->x, &ws->r, &ws->wbuf, &ws->gat_s, &ws->gat_t})
    b->ensure(3 * (size_t)N);
  for (DevBuf<double>* b : {&ws->cg_b, &ws->cg_x, &ws->cg_r, &ws->cg_z, &ws->cg_p, &ws->cg_s})
    b->ensure(3 * (size_t)N);
  ws->cg_w.ensure(3 * (size_t)N + 2);
  ws->cg_minv.ensure(9 * (size_t)N);
  ws->vpart.ensure(2 * kCgMaxBlocks * 2);
  ws->dpart.ensure(kMaxApplySlots);
  ws->cgst.ensure(1);
  ws->cgsc.ensure(2);
  ws->part_misc.ensure(kMaxBlocks * 8);
  ws->scal.ensure(64);
  ws->flags.ensure(4);
  GSFM_HIP_CHECK(hipMemsetAsync(ws->flags.get(), 0, 4 * sizeof(int), s));

  copy_in(ctx, ws->ei.ensure(E + 1), prob->edge_i, E, mem);
  copy_in(ctx, ws->ej.ensure(E + 1), prob->edge_j, E, mem);
  copy_in(ctx, ws->eq.ensure(4 * (E + 1)), prob->edge_q, 4 * E, mem);
  d.ew = nullptr;
  if (opt->use_weight) {
    GSFM_REQUIRE(prob->edge_weight != nullptr, "RA: use_weight requires edge_weight");
    // negative weights mean "unset" -> 1 (gra.cc:417-420)
    std::vector<double> hw;
    to_host(ctx, hw, prob->edge_weight, E, mem);
    for (auto& w : hw)
      if (!(w >= 0)) w = 1.0;
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->ew.ensure(E + 1), hw.data(), E * sizeof(double), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    d.ew = ws->ew.get();
  }
  // host copies of the topology for the incidence build (and the MST)
  std::vector<int>&h_ei = hi.h_ei, &h_ej = hi.h_ej;
  to_host(ctx, h_ei, prob->edge_i, E, mem);
  to_host(ctx, h_ej, prob->edge_j, E, mem);
  if (d.grav != nullptr) {  // a pair of two gravity frames contributes one row (gra.cc:390-397)
    d.rows_edges = 0.0;
    for (long e = 0; e < E; ++e) {
      GSFM_REQUIRE(h_ei[e] >= 0 && h_ei[e] < N && h_ej[e] >= 0 && h_ej[e] < N, "RA: edge index out of range");
      d.rows_edges += (h_grav[h_ei[e]] && h_grav[h_ej[e]]) ? 1.0 : 3.0;
    }
  }
  std::vector<int>& pos = hi.pos;
  pos.clear();
  hi.mst_root = 0;
  static const bool trace_setup = getenv("GSFM_TRACE_SETUP") != nullptr;
  const double ts0 = now_seconds();
  if (d.blockdense) {  // relabel the nodes in BFS order for the whole solve: index-contiguous blocks become graph-local
    for (long e = 0; e < E; ++e)
      GSFM_REQUIRE(h_ei[e] >= 0 && h_ei[e] < N && h_ej[e] >= 0 && h_ej[e] < N, "RA: edge index out of range");
    std::vector<int> order;
    bfs_order(N, E, h_ei.data(), h_ej.data(), d.fixed, order, pos);
    for (long e = 0; e < E; ++e) {
      h_ei[e] = pos[h_ei[e]];
      h_ej[e] = pos[h_ej[e]];
    }
    // substructuring (ra_sub.hpp): interiors of more, smaller BFS blocks first, the interface between them last
    const bool no_sub = ctx->knob[GSFM_KNOB_RA_NO_SUBSTRUCTURE] != 0;  // A/B knob: the plain block preconditioner
    if (!no_sub) sub_plan(N, E, h_ei.data(), h_ej.data(), d.plan);
    d.sub = !no_sub && d.plan.ok;
    if (d.sub) {
      const std::vector<int>& np = d.plan.newpos;
      for (long e = 0; e < E; ++e) {
        h_ei[e] = np[h_ei[e]];
        h_ej[e] = np[h_ej[e]];
      }
      std::vector<int> order2(N);
      for (int p = 0; p < N; ++p) order2[np[p]] = order[p];
      order.swap(order2);
      for (int n = 0; n < N; ++n) pos[n] = np[pos[n]];
    }
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->ei.get(), h_ei.data(), E * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->ej.get(), h_ej.data(), E * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipMemcpyAsync(ws->order.ensure(N), order.data(), N * sizeof(int), hipMemcpyHostToDevice, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    d.fixed = pos[d.fixed];
    hi.mst_root = pos[0];
  }
  const double ts1 = now_seconds();
  if (d.sub) {
    std::vector<int> h_rowptr, h_nbr;
    build_incidence(d, h_ei.data(), h_ej.data(), &h_rowptr, &h_nbr);
    const double ts2 = now_seconds();
    sub_upload(d, h_rowptr, h_nbr);
    if (trace_setup)
      fprintf(stderr, "[gsfm ra setup] BFS + plan %.2f ms, incidence %.2f ms, coupling tables %.2f ms\n", (ts1 - ts0) * 1e3,
              (ts2 - ts1) * 1e3, (now_seconds() - ts2) * 1e3);
  } else {
    build_incidence(d, h_ei.data(), h_ej.data());
  }

  // device-to-host copies for the initialisation, all BEFORE any solver kernel is enqueued
  to_host(ctx, hi.h_rot, rot_in, 3 * (size_t)N, mem);
  if (!opt->skip_initialization && !opt->use_gravity) {
    GSFM_REQUIRE(prob->edge_ninl != nullptr, "RA: MST initialisation requires edge_ninl");
    GSFM_REQUIRE(ctx->comm.world == 1, "RA: MST initialisation needs the whole graph; initialise before sharding");
    to_host(ctx, hi.h_eq, prob->edge_q, 4 * (size_t)E, mem);
    to_host(ctx, hi.h_ninl, prob->edge_ninl, E, mem);
  }
}

// Phase 2: initial rotations (maximum spanning tree, gra.cc:87-138) on the host, uploaded asynchronously.
void finish_init(gsfm_ctx* ctx, const gsfm_ra_options* opt, RaDevice& d, RaHostInit& hi) {
  RaWs* ws = d.ws;
  hipStream_t s = ctx->stream;
  const int N = d.N;
  const long E = d.E;
  std::vector<double>& h_rot = hi.h_rot;
  if (d.blockdense) {
    std::vector<double> tmp(h_rot.size());
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < 3; ++c) tmp[3 * (size_t)hi.pos[n] + c] = h_rot[3 * (size_t)n + c];
    h_rot.swap(tmp);
  }
  if (!opt->skip_initialization && !opt->use_gravity)  // gra.cc:60-62: no spanning-tree start with use_gravity
    mst_init(N, E, hi.h_ei.data(), hi.h_ej.data(), hi.h_eq.data(), hi.h_ninl.data(), h_rot.data(), hi.mst_root);
  GSFM_HIP_CHECK(hipMemcpyAsync(ws->rot.ensure(3 * (size_t)N), h_rot.data(), 3 * (size_t)N * sizeof(double), hipMemcpyHostToDevice, s));
  // the gauge node is held at its (post-initialisation) rotation (gra.cc:248-257)
  GSFM_HIP_CHECK(hipMemcpyAsync(ws->fixed_rot0.ensure(4), h_rot.data() + 3 * (size_t)d.fixed, 3 * sizeof(double), hipMemcpyHostToDevice, s));
}

int read_nan_flag(RaDevice& d) {
  int* h = reinterpret_cast<int*>(d.ctx->h_pinned + 128);
  GSFM_HIP_CHECK(hipMemcpyAsync(h, d.ws->flags.get(), sizeof(int), hipMemcpyDeviceToHost, d.ctx->stream));
  GSFM_HIP_CHECK(hipStreamSynchronize(d.ctx->stream));
  int flag = h[0];
  if (d.ctx->comm.world > 1) {  // every rank must take the same exit
    d.ctx->h_pinned[81] = (double)flag;
    GSFM_HIP_CHECK(hipMemcpyAsync(d.ws->scal.get() + 9, d.ctx->h_pinned + 81, sizeof(double), hipMemcpyHostToDevice, d.ctx->stream));
    allreduce_max(d.ctx, d.ws->scal.get() + 9, 1);
    GSFM_HIP_CHECK(hipMemcpyAsync(d.ctx->h_pinned + 81, d.ws->scal.get() + 9, sizeof(double), hipMemcpyDeviceToHost, d.ctx->stream));
    GSFM_HIP_CHECK(hipStreamSynchronize(d.ctx->stream));
    flag = d.ctx->h_pinned[81] != 0.0;
  }
  return flag;
}

template <int MODE>
void launch_gather(RaDevice& d, const int* stop = nullptr) {
  RaWs* ws = d.ws;
  if (MODE != GATHER_L1RHS) {  // the weighted Laplacian changes
    d.dense_valid = false;
    d.bd_fresh = false;
  }
  dispatch_lpr(d.lpr, [&](auto L) {
    hipLaunchKernelGGL((k_node_gather<MODE, decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0,
                       d.ctx->stream, d.N, d.E, ws->rowptr.get(), ws->inc.get(), ws->res.get(),
                       ws->wirls.get(), d.ew, ws->z.get(), ws->u.get(), ws->dz.get(), ws->inc_w.get(),
                       ws->lap_diag.get(), ws->lap_diag_loc.get(), ws->rhs.get(), ws->gat_s.get(), ws->gat_t.get(),
                       d.fixed, d.has_gauge, stop);
  });
  if (d.grav != nullptr && MODE != GATHER_L1W)  // A^T(...) has no x / z rows at gravity frames
    hipLaunchKernelGGL(k_ra_mask3, dim3(d.gridN), dim3(kBlock), 0, d.ctx->stream, d.N, d.grav, ws->rhs.get(),
                       MODE == GATHER_L1RHS ? ws->gat_s.get() : nullptr, MODE == GATHER_L1RHS ? ws->gat_t.get() : nullptr);
}

int ra_solve_rig_impl(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt, double* rot_inout,
                      gsfm_report* rep);

int ra_solve_impl(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                  double* rot_inout, gsfm_report* rep) {
  GSFM_REQUIRE(prob && opt && rot_inout, "RA: null argument");
  if (opt->use_gravity && prob->num_images > 0)
    throw StatusError(GSFM_ERR_UNSUPPORTED, "RA: use_gravity with cam_from_rig unknowns (the reference refuses it too, gra.cc:47-58)");
  if (prob->num_nodes <= 0) throw StatusError(GSFM_ERR_EMPTY_PROBLEM, "RA: no nodes");
  GSFM_REQUIRE(prob->fixed_node >= 0 && prob->fixed_node < prob->num_nodes, "RA: fixed_node out of range");
  if (prob->num_images > 0) return ra_solve_rig_impl(ctx, prob, opt, rot_inout, rep);  // cam_from_rig rotations unknown
  const double t0 = now_seconds();
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  RaDevice d;
  RaHostInit hi;
  setup_device(ctx, prob, opt, rot_inout, d, hi, /*allow_blockdense=*/true);
  RaWs* ws = d.ws;
  hipStream_t s = ctx->stream;
  const int N = d.N;
  const long E = d.E;
  const bool multi = ctx->comm.world > 1;
  // The L1-stage matrix (WA)^T (WA) depends on the topology and the edge weights only: enqueue its (block)
  // inversion now, so that the GPU factorises while the host computes the maximum-spanning-tree initialisation.
  bool l1w_ready = false;
  if (opt->max_num_l1_iterations > 0 && !multi && (d.dense || d.blockdense)) {
    launch_gather<GATHER_L1W>(d);
    if (d.dense) dense_factor(d); else bd_factor(d);
    l1w_ready = true;
  }
  const double tf0 = now_seconds();
  finish_init(ctx, opt, d, hi);
  const double t1 = now_seconds();
  if (getenv("GSFM_TRACE_SETUP")) fprintf(stderr, "[gsfm ra setup] total before init %.2f ms, spanning tree + upload %.2f ms\n", (tf0 - t0) * 1e3, (t1 - tf0) * 1e3);
  long lin_iters = 0;
  int it_l1 = 0, it_irls = 0;
  double last_step = 0.0;
  double upd[3];

  // ---------------- L1 stage (gra.cc:479-541)
  if (opt->max_num_l1_iterations > 0) {
    const size_t rows3 = 3 * (size_t)(E + 1);
    ws->z.ensure(rows3);
    ws->u.ensure(rows3);
    ws->dz.ensure(rows3);
    if (!l1w_ready) launch_gather<GATHER_L1W>(d);  // (WA)^T (WA): factorised once in the reference (gra.cc:491)
    if (multi) {
      allreduce_sum(ctx, ws->lap_diag.get(), N);  // preconditioner diagonal; inc_w / lap_diag_loc stay local
    }
    double last_norm = 0.0, curr_norm = 0.0;
    launch_residuals(d, false, 0, 0.0);
    // A.rows() incl. the gauge rows — of the WHOLE graph when the edges are sharded over ranks
    double e_glob = d.rows_edges;  // rows of A from the edges: 3 each, 1 for a pair of two gravity frames
    if (multi) {
      ctx->h_pinned[80] = e_glob;
      GSFM_HIP_CHECK(hipMemcpyAsync(ws->scal.get() + 8, ctx->h_pinned + 80, sizeof(double), hipMemcpyHostToDevice, s));
      allreduce_sum(ctx, ws->scal.get() + 8, 1);
      GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 80, ws->scal.get() + 8, sizeof(double), hipMemcpyDeviceToHost, s));
      GSFM_HIP_CHECK(hipStreamSynchronize(s));
      e_glob = ctx->h_pinned[80];
    }
    const double rows_total = e_glob + (double)d.gauge_rows;
    for (int it = 0; it < opt->max_num_l1_iterations; ++it) {
      last_norm = curr_norm;
      // --- colmap::LeastAbsoluteDeviationSolver::Solve(b' = W b, &x), x starts at 0
      GSFM_HIP_CHECK(hipMemsetAsync(ws->z.get(), 0, rows3 * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(ws->u.get(), 0, rows3 * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(ws->dz.get(), 0, rows3 * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(ws->x.get(), 0, 3 * (size_t)N * sizeof(double), s));
      launch_gather<GATHER_L1RHS>(d);  // rhs = A'^T (b' + 0 - 0)
      if (multi) allreduce_sum(ctx, ws->rhs.get(), 3 * (size_t)N);
      double rows_glob = rows_total;
      if (d.dense) {
        // The whole ADMM loop is enqueued without a host round trip: k_admm_check raises the stop
        // flag on the device and every later kernel of this solve returns at once.  The solves of
        // the inner iterations skip the refinement step (ADMM itself stops at 1e-2 relative).
        int* stop = ws->flags.get() + 1;
        int* count = ws->flags.get() + 2;
        GSFM_HIP_CHECK(hipMemsetAsync(stop, 0, 3 * sizeof(int), s));  // stop, count, ticket
        // two launches per ADMM iteration: x = A^-1 rhs, then the fused edge / gather / stopping-test sweep
        double *z_cur = ws->z.get(), *u_cur = ws->u.get();
        double *z_nxt = ws->z2.ensure(rows3), *u_nxt = ws->u2.ensure(rows3);
        unsigned* ticket = reinterpret_cast<unsigned*>(ws->flags.get() + 3);
        for (int a = 0; a < opt->l1_admm_max_num_iterations; ++a) {
          dense_solve(d, /*refine=*/false, stop);
          dispatch_lpr(d.lpr, [&](auto L) {
            hipLaunchKernelGGL((k_admm_node<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, N, E, ws->rowptr.get(),
                               ws->inc.get(), ws->nbr.get(), d.ew, ws->res.get(), ws->x.get(), z_cur, u_cur, z_nxt, u_nxt,
                               opt->l1_admm_alpha, 1.0 / opt->l1_admm_rho, ws->rhs.get(), d.fixed, d.has_gauge,
                               ws->part_misc.get(), ticket, rows_glob, opt->l1_admm_absolute_tolerance,
                               opt->l1_admm_relative_tolerance, opt->l1_admm_rho, stop, count);
          });
          std::swap(z_cur, z_nxt);
          std::swap(u_cur, u_nxt);
        }
        lin_iters += opt->l1_admm_max_num_iterations;
      }
      for (int a = 0; !d.dense && a < opt->l1_admm_max_num_iterations; ++a) {
        // first x-update of a solve: cold start, full tolerance; later ones: warm-started corrections
        lin_iters += pcg_solve(d, a > 0, a > 0 ? opt->pcg_relative_tolerance_admm : opt->pcg_relative_tolerance,
                               opt->pcg_max_iterations);
        hipLaunchKernelGGL(k_admm_edge, dim3(d.gridE), dim3(kBlock), 0, s, E, d.has_gauge, d.fixed,
                           ws->ei.get(), ws->ej.get(), d.ew, ws->res.get(), ws->x.get(), ws->z.get(),
                           ws->u.get(), ws->dz.get(), opt->l1_admm_alpha, 1.0 / opt->l1_admm_rho,
                           ws->part_misc.get(), nullptr);
        hipLaunchKernelGGL((k_finalize<4>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), d.gridE, ws->scal.get());
        launch_gather<GATHER_L1RHS>(d);  // next rhs + the two dual-residual gathers
        if (multi) {
          allreduce_sum(ctx, ws->rhs.get(), 3 * (size_t)N);
          allreduce_sum(ctx, ws->gat_s.get(), 3 * (size_t)N);
          allreduce_sum(ctx, ws->gat_t.get(), 3 * (size_t)N);
          allreduce_sum(ctx, ws->scal.get(), 4);
        }
        hipLaunchKernelGGL(k_sumsq2, dim3(d.gridN), dim3(kBlock), 0, s, 3 * (long)N, ws->gat_s.get(),
                           ws->gat_t.get(), ws->part_misc.get());
        hipLaunchKernelGGL((k_finalize<2>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), d.gridN, ws->scal.get() + 4);
        GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 16, ws->scal.get(), 6 * sizeof(double), hipMemcpyDeviceToHost, s));
        GSFM_HIP_CHECK(hipStreamSynchronize(s));
        const double* h = ctx->h_pinned + 16;
        const double r_norm = std::sqrt(h[0]), Ax_norm = std::sqrt(h[1]), z_norm = std::sqrt(h[2]),
                     b_norm = std::sqrt(h[3]);
        const double rho = opt->l1_admm_rho;
        const double s_norm = rho * std::sqrt(h[4]);
        const double dual_norm = rho * std::sqrt(h[5]);
        const double primal_eps = std::sqrt(rows_glob) * opt->l1_admm_absolute_tolerance +
                                  opt->l1_admm_relative_tolerance * std::max({Ax_norm, z_norm, b_norm});
        const double dual_eps = std::sqrt(d.cols) * opt->l1_admm_absolute_tolerance +
                                opt->l1_admm_relative_tolerance * dual_norm;
        if (r_norm < primal_eps && s_norm < dual_eps) break;
      }
      // --- back in SolveL1Regression
      update_rotations(d, upd);
      it_l1 = it + 1;
      if (upd[2] > 0) {  // NaN in the step (gra.cc:508-512)
        if (rep) rep->iterations_l1 = it_l1;
        return GSFM_ERR_NUMERICAL;
      }
      curr_norm = upd[1];
      last_step = upd[0];
      launch_residuals(d, false, 0, 0.0);
      if (upd[0] < opt->l1_step_convergence_threshold || std::fabs(last_norm - curr_norm) < 1e-12) break;
    }
  }

  // ---------------- IRLS stage (gra.cc:543-625)
  if (opt->max_num_irls_iterations > 0) {
    const double sigma = opt->irls_loss_parameter_sigma * M_PI / 180.0;
    launch_residuals(d, true, opt->weight_type, sigma * sigma);
    for (int it = 0; it < opt->max_num_irls_iterations; ++it) {
      if (read_nan_flag(d)) {  // NaN weight (gra.cc:590-593)
        if (rep) rep->iterations_irls = it_irls;
        return GSFM_ERR_NUMERICAL;
      }
      launch_gather<GATHER_IRLS>(d);
      if (multi) {
        allreduce_sum(ctx, ws->rhs.get(), 3 * (size_t)N);
        allreduce_sum(ctx, ws->lap_diag.get(), N);
      }
      lin_iters += pcg_solve(d, false, opt->pcg_relative_tolerance, opt->pcg_max_iterations);
      update_rotations(d, upd);
      it_irls = it + 1;
      last_step = upd[0];
      launch_residuals(d, true, opt->weight_type, sigma * sigma);
      if (upd[0] < opt->irls_step_convergence_threshold) break;
    }
  }

  if (d.blockdense) {  // back from BFS positions to the caller's node ids
    hipLaunchKernelGGL(k_unpermute3, dim3(d.gridN), dim3(kBlock), 0, s, N, ws->order.get(), ws->rot.get(), ws->rot_out.ensure(3 * (size_t)N));
    copy_out(ctx, rot_inout, ws->rot_out.get(), 3 * (size_t)N, prob->mem);
  } else {
    copy_out(ctx, rot_inout, ws->rot.get(), 3 * (size_t)N, prob->mem);
  }
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  const double t2 = now_seconds();
  if (rep) {
    rep->iterations_l1 = it_l1;
    rep->iterations_irls = it_irls;
    rep->iterations = it_l1 + it_irls;
    rep->linear_iterations = lin_iters;
    rep->termination = GSFM_TERM_CONVERGENCE;
    rep->seconds_total = t2 - t0;
    rep->seconds_solve = t2 - t1;
    rep->last_step_norm = last_step;
  }
  return GSFM_OK;
}


// ------------------------------------------------------------------------------------------
// Rigs with cam_from_rig ROTATIONS among the unknowns (gra.cc:173-191, 396-446, 646-693, 718-739)
// ------------------------------------------------------------------------------------------
// The view graph stays what it is — nodes = IMAGES, two endpoints per edge — and everything above keeps working on it:
// the residual sweep sees image quaternions q_cam * q_frame, the gathers and the Laplacian SpMV run per image.  The
// unknowns are the frames and the C cam blocks; an image's tangent is the SUM of its frame's and its cam's
// (x_image = V x, V = [frame incidence | cam incidence]): exactly the -1 / +1 pattern gra.cc:396-446 writes, with the
// cancellation of equal columns for free.  So  A = A_image V,  A^T W A = V^T L_w V  and the linear solves are a
// Jacobi-PCG on the 3 (N + C) reduced unknowns whose operator is  expand -> image SpMV -> reduce.  The gauge rows sit on
// an image of the gauge frame that has no cam block (a virtual, edge-less image is appended when the frame has none).
struct RigRa {
  int N, C, NI;
  const int* img_frame;  // [NI]
  const int* img_cam;    // [NI] block or -1
  const int* foff;       // [N + 1] frame -> images
  const int* fimg;
  const int* coff;       // [C + 1] cam -> images
  const int* cimg;
};

// nq[i] = quat(Exp(cam)) * quat(Exp(frame))  (gra.cc:729-738); the gauge rows Log(R_fix0^T R_fix) of the gauge FRAME
__global__ void __launch_bounds__(kBlock)
    k_rig_ra_node_quat(RigRa r, const double* __restrict__ rotf, const double* __restrict__ rotc, double* __restrict__ nq,
                       int fixed_img, const double* __restrict__ fixed_rot0, double* __restrict__ gauge_out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < r.NI; i += gridDim.x * blockDim.x) {
    const int f = r.img_frame[i], c = r.img_cam[i];
    const Quat qf = aa_to_quat(rotf[3 * f], rotf[3 * f + 1], rotf[3 * f + 2]);
    Quat q = qf;
    if (c >= 0) q = qmul(aa_to_quat(rotc[3 * c], rotc[3 * c + 1], rotc[3 * c + 2]), qf);
    store_quat(nq + 4 * (long)i, q);
    if (i == fixed_img) {
      const Quat q0 = aa_to_quat(fixed_rot0[0], fixed_rot0[1], fixed_rot0[2]);
      double gx, gy, gz;
      quat_to_aa(qmul(qconj(q0), qf), gx, gy, gz);
      gauge_out[0] = gx;
      gauge_out[1] = gy;
      gauge_out[2] = gz;
    }
  }
}

// dst_image = V src: frame part + cam part.  As the first kernel of a PCG apply it also runs the convergence test.
__global__ void __launch_bounds__(kBlock)
    k_rig_ra_expand(RigRa r, const double* __restrict__ src, double* __restrict__ dst, CgVec v, int it, double tol2,
                    int test) {
  __shared__ double smem[4 * 2 + 2];
  if (test && cg_converged(v, it, tol2, smem)) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < r.NI; i += gridDim.x * blockDim.x) {
    const long f = r.img_frame[i];
    const int c = r.img_cam[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double x = src[3 * f + j];
      if (c >= 0) x += src[3 * (long)(r.N + c) + j];
      dst[3 * (long)i + j] = x;
    }
  }
}

// dst = V^T src (W values per image): frames by one thread each, cam blocks by one workgroup each (fixed order)
template <int W>
__global__ void __launch_bounds__(kBlock)
    k_rig_ra_reduce(RigRa r, const double* __restrict__ src, double* __restrict__ dst) {
  __shared__ double smem[4 * W];
  const int fblocks = (r.N + kBlock - 1) / kBlock;
  if ((int)blockIdx.x < fblocks) {
    const int f = blockIdx.x * kBlock + threadIdx.x;
    if (f < r.N) {
      double acc[W];
#pragma unroll
      for (int j = 0; j < W; ++j) acc[j] = 0.0;
      for (int a = r.foff[f]; a < r.foff[f + 1]; ++a) {
        const double* sp = src + (long)W * r.fimg[a];
#pragma unroll
        for (int j = 0; j < W; ++j) acc[j] += sp[j];
      }
#pragma unroll
      for (int j = 0; j < W; ++j) dst[(long)W * f + j] = acc[j];
    }
    return;
  }
  const int c = blockIdx.x - fblocks;  // gridDim.x = fblocks + C
  double acc[W];
#pragma unroll
  for (int j = 0; j < W; ++j) acc[j] = 0.0;
  for (int a = r.coff[c] + threadIdx.x; a < r.coff[c + 1]; a += blockDim.x) {
    const double* sp = src + (long)W * r.cimg[a];
#pragma unroll
    for (int j = 0; j < W; ++j) acc[j] += sp[j];
  }
  block_sum<W>(acc, smem);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < W; ++j) dst[(long)W * (r.N + c) + j] = acc[j];
  }
}

// delta partials of a PCG apply: dpart[block] = sum z.w over this block's share
__global__ void __launch_bounds__(kBlock) k_rig_ra_dot(CgVec v) {
  __shared__ double smem[4];
  if (v.st->done) return;
  double acc[1] = {0.0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < v.n; i += (long)gridDim.x * blockDim.x) acc[0] += v.z[i] * v.w[i];
  block_sum<1>(acc, smem);
  if (threadIdx.x == 0) v.dpart[blockIdx.x] = acc[0];
}

// UpdateGlobalRotations for the cam blocks (gra.cc:646-690), after the frames were updated: the new cam rotation is the
// quaternion average over the cam's images of  R_cam R_f Exp(-step) R_f^T  (principal eigenvector of sum q q^T, cyclic
// Jacobi on the 4 x 4).  One thread per cam block: there are a handful.
__global__ void __launch_bounds__(64)
    k_rig_ra_cam_update(RigRa r, const double* __restrict__ rotf, double* __restrict__ rotc, const double* __restrict__ step) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= r.C || r.coff[c + 1] == r.coff[c]) return;
  const Quat qc = aa_to_quat(rotc[3 * c], rotc[3 * c + 1], rotc[3 * c + 2]);
  const double* d = step + 3 * (long)(r.N + c);
  const Quat qu = aa_to_quat(-d[0], -d[1], -d[2]);
  double A[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) A[i][j] = 0.0;
  for (int a = r.coff[c]; a < r.coff[c + 1]; ++a) {
    const int f = r.img_frame[r.cimg[a]];
    const Quat qf = aa_to_quat(rotf[3 * f], rotf[3 * f + 1], rotf[3 * f + 2]);
    const Quat q = qmul(qmul(qmul(qc, qf), qu), qconj(qf));
    const double v[4] = {q.w, q.x, q.y, q.z};
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) A[i][j] += v[i] * v[j];
  }
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) off += A[i][j] * A[i][j];
    if (off < 1e-30) break;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double th = 0.5 * atan2(2.0 * A[p][q], A[q][q] - A[p][p]);
        const double cs = cos(th), sn = sin(th);
        for (int k = 0; k < 4; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = cs * akp - sn * akq;
          A[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cs * apk - sn * aqk;
          A[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = cs * vkp - sn * vkq;
          V[k][q] = sn * vkp + cs * vkq;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i)
    if (A[i][i] > A[best][best]) best = i;
  double ax, ay, az;
  quat_to_aa(Quat{V[0][best], V[1][best], V[2][best], V[3][best]}, ax, ay, az);
  rotc[3 * c] = ax;
  rotc[3 * c + 1] = ay;
  rotc[3 * c + 2] = az;
}

struct RigRaWs {
  DevBuf<int> img_frame, img_cam, foff, fimg, coff, cimg;
  DevBuf<double> rotf, rotc, rhs, x, r, wbuf, gat_s, gat_t, diag, b, cx, cr, cz, cp, cs, cw, minv, zimg, tmp_rot;
  static void destroy(void* p) { delete static_cast<RigRaWs*>(p); }
};

struct RigSolve {
  RaDevice* d;
  RigRaWs* rw;
  RigRa r;
  int nred;      // N + C
  int gridRed;   // blocks of the reduce kernels: frames + one per cam
  int gridI, gridR;
};

void rig_expand(RigSolve& g, const double* src_red, double* dst_img) {
  CgVec dummy{};
  hipLaunchKernelGGL(k_rig_ra_expand, dim3(g.gridI), dim3(kBlock), 0, g.d->ctx->stream, g.r, src_red, dst_img, dummy, 0, 0.0, 0);
}
template <int W>
void rig_reduce(RigSolve& g, const double* src_img, double* dst_red) {
  hipLaunchKernelGGL((k_rig_ra_reduce<W>), dim3(g.gridRed), dim3(kBlock), 0, g.d->ctx->stream, g.r, src_img, dst_red);
}

// y_red = V^T (L_w + gauge) V x_red through the image-level SpMV
void rig_operator(RigSolve& g, const double* x_red, double* y_red) {
  RaDevice& d = *g.d;
  RaWs* ws = d.ws;
  rig_expand(g, x_red, g.rw->zimg.get());
  dispatch_lpr(d.lpr, [&](auto L) {
    hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, d.ctx->stream, d.N, ws->rowptr.get(),
                       ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), g.rw->zimg.get(), ws->wbuf.get());
  });
  rig_reduce<3>(g, ws->wbuf.get(), y_red);
}

// Solves V^T (L_w + gauge) V x = rhs_red (weights in ws->inc_w / lap_diag, Jacobi blocks from the reduced diagonal) the
// way pcg_solve does for plain graphs: optional warm start, then true-residual verification passes.
int rig_pcg_solve(RigSolve& g, bool warm, double tol, int max_iter) {
  RaDevice& d = *g.d;
  RaWs* ws = d.ws;
  RigRaWs* rw = g.rw;
  gsfm_ctx* ctx = d.ctx;
  hipStream_t s = ctx->stream;
  const int n = g.nred;
  const long n3 = 3L * n;
  const double* b = rw->rhs.get();
  if (warm) {
    rig_operator(g, rw->x.get(), rw->wbuf.get());
    hipLaunchKernelGGL(k_dense_residual, dim3(g.gridR), dim3(kBlock), 0, s, n3, rw->rhs.get(), rw->wbuf.get(), rw->b.get());
    b = rw->b.get();
  }
  hipLaunchKernelGGL(k_ra_minv, dim3(g.gridR), dim3(kBlock), 0, s, n, rw->diag.get(), rw->minv.get());
  CgVec v;
  v.n = (int)n3;
  v.N = n;
  v.K = 0;
  v.nb_update = std::min(kCgUpdateBlocks, grid_for(n, kBlock));
  const int gD = std::min(64, grid_for((size_t)n3, kBlock));
  v.nb_apply = gD;
  v.b = b;
  v.x = warm ? rw->cx.get() : rw->x.get();
  v.r = rw->cr.get();
  v.z = rw->cz.get();
  v.p = rw->cp.get();
  v.s = rw->cs.get();
  v.w = rw->cw.get();
  v.minv = rw->minv.get();
  v.vpart = ws->vpart.get();
  v.dpart = ws->dpart.get();
  v.scal = ws->cgsc.get();
  v.st = ws->cgst.get();
  auto run_cg = [&](double tol_pass) {
    return cg_solve<3, false>(ctx, v, tol_pass, max_iter, [&](int it) {
      hipLaunchKernelGGL(k_rig_ra_expand, dim3(g.gridI), dim3(kBlock), 0, s, g.r, v.z, rw->zimg.get(), v, it,
                         tol_pass * tol_pass, 1);
      dispatch_lpr(d.lpr, [&](auto L) {
        hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, d.N, ws->rowptr.get(),
                           ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(), rw->zimg.get(), ws->wbuf.get());
      });
      rig_reduce<3>(g, ws->wbuf.get(), v.w);
      hipLaunchKernelGGL(k_rig_ra_dot, dim3(gD), dim3(kBlock), 0, s, v);
    });
  };
  long iters = run_cg(tol);
  if (warm) hipLaunchKernelGGL(k_ra_add, dim3(g.gridR), dim3(kBlock), 0, s, n3, rw->x.get(), rw->cx.get(), rw->x.get());
  const double* bref = b;
  for (int pass = 0; pass < 3; ++pass) {  // true residual, correction solves (see pcg_solve)
    rig_operator(g, rw->x.get(), rw->wbuf.get());
    hipLaunchKernelGGL(k_dense_residual, dim3(g.gridR), dim3(kBlock), 0, s, n3, rw->rhs.get(), rw->wbuf.get(), rw->r.get());
    hipLaunchKernelGGL(k_sumsq2, dim3(g.gridR), dim3(kBlock), 0, s, n3, rw->r.get(), bref, ws->part_misc.get());
    hipLaunchKernelGGL((k_finalize<2>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), g.gridR, ws->scal.get() + 16);
    GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 200, ws->scal.get() + 16, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    const double rr = ctx->h_pinned[200], bb = ctx->h_pinned[201];
    if (!(rr > tol * tol * bb) || !(bb > 0.0)) break;
    const double tol_pass = std::min(0.1, std::max(1e-12, 0.5 * tol * std::sqrt(bb / rr)));
    v.b = rw->r.get();
    v.x = rw->cx.get();
    iters += run_cg(tol_pass);
    hipLaunchKernelGGL(k_ra_add, dim3(g.gridR), dim3(kBlock), 0, s, n3, rw->x.get(), rw->cx.get(), rw->x.get());
  }
  return (int)iters;
}

void rig_residuals(RigSolve& g, bool with_weights, int weight_type, double sigma2) {
  RaDevice& d = *g.d;
  RaWs* ws = d.ws;
  hipStream_t s = d.ctx->stream;
  hipLaunchKernelGGL(k_rig_ra_node_quat, dim3(g.gridI), dim3(kBlock), 0, s, g.r, g.rw->rotf.get(), g.rw->rotc.get(),
                     ws->nq.get(), d.fixed, ws->fixed_rot0.get(), ws->res.get() + 3 * d.E);
  hipLaunchKernelGGL(k_edge_residual, dim3(d.gridE), dim3(kBlock), 0, s, d.E, ws->ei.get(), ws->ej.get(), ws->eq.get(),
                     ws->nq.get(), ws->res.get(), with_weights ? ws->wirls.get() : nullptr, weight_type, sigma2,
                     ws->flags.get(), (const unsigned char*)nullptr, (const double*)nullptr);
}

// frames: r <- Log(Exp(r) Exp(-step)), then the cam blocks from the UPDATED frames.  out = {mean |step| over the FRAMES
// (gra.cc:758-772), |step|_2 over all unknowns, #NaN}
void rig_update(RigSolve& g, double out[3]) {
  RaDevice& d = *g.d;
  RaWs* ws = d.ws;
  RigRaWs* rw = g.rw;
  hipStream_t s = d.ctx->stream;
  const int gF = grid_for(g.r.N, kBlock);
  hipLaunchKernelGGL(k_node_update, dim3(gF), dim3(kBlock), 0, s, g.r.N, rw->rotf.get(), rw->x.get(), ws->part_misc.get());
  hipLaunchKernelGGL((k_finalize<3>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), gF, ws->scal.get());
  if (g.r.C > 0) {
    // |step|^2 and NaN count of the cam part (k_node_update on a scratch copy of the cam rotations: only its sums are used)
    GSFM_HIP_CHECK(hipMemcpyAsync(rw->tmp_rot.get(), rw->rotc.get(), 3 * (size_t)g.r.C * sizeof(double), hipMemcpyDeviceToDevice, s));
    const int gC = grid_for(g.r.C, kBlock);
    hipLaunchKernelGGL(k_node_update, dim3(gC), dim3(kBlock), 0, s, g.r.C, rw->tmp_rot.get(), rw->x.get() + 3 * (size_t)g.r.N,
                       ws->part_misc.get() + 3 * (size_t)gF);
    hipLaunchKernelGGL((k_finalize<3>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get() + 3 * (size_t)gF, gC, ws->scal.get() + 3);
    hipLaunchKernelGGL(k_rig_ra_cam_update, dim3(grid_for(g.r.C, 64)), dim3(64), 0, s, g.r, rw->rotf.get(), rw->rotc.get(),
                       rw->x.get());
  } else {
    GSFM_HIP_CHECK(hipMemsetAsync(ws->scal.get() + 3, 0, 3 * sizeof(double), s));
  }
  GSFM_HIP_CHECK(hipMemcpyAsync(d.ctx->h_pinned + 64, ws->scal.get(), 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  const double* h = d.ctx->h_pinned + 64;
  out[0] = h[0] / g.r.N;
  out[1] = std::sqrt(h[1] + h[4]);
  out[2] = h[2] + h[5];
}

int ra_solve_rig_impl(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt, double* rot_inout,
                      gsfm_report* rep) {
  GSFM_REQUIRE(prob->image_frame && prob->image_cam, "RA: image tables missing");
  GSFM_REQUIRE(prob->num_cams >= 0 && (prob->num_cams == 0 || prob->cam_rot_aa), "RA: cam blocks missing");
  if (ctx->comm.world > 1) throw StatusError(GSFM_ERR_UNSUPPORTED, "RA: cam_from_rig unknowns are solved on one rank");
  if (!opt->skip_initialization)
    throw StatusError(GSFM_ERR_UNSUPPORTED,
                      "RA: cam_from_rig unknowns need skip_initialization (initialise from the image-level spanning tree)");
  const double t0 = now_seconds();
  GSFM_HIP_CHECK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int mem = prob->mem;
  const int N = prob->num_nodes, C = prob->num_cams;
  int NI = prob->num_images;
  const long E = prob->num_edges;
  std::vector<int> h_imf, h_imc;
  to_host(ctx, h_imf, prob->image_frame, (size_t)NI, mem);
  to_host(ctx, h_imc, prob->image_cam, (size_t)NI, mem);
  int fixed_img = -1;
  for (int i = 0; i < NI; ++i) {
    GSFM_REQUIRE(h_imf[i] >= 0 && h_imf[i] < N, "RA: image_frame out of range");
    GSFM_REQUIRE(h_imc[i] >= -1 && h_imc[i] < C, "RA: image_cam out of range");
    if (fixed_img < 0 && h_imf[i] == prob->fixed_node && h_imc[i] < 0) fixed_img = i;
  }
  if (fixed_img < 0) {  // the gauge frame has no image without a cam block: a virtual edge-less one carries the gauge rows
    fixed_img = NI++;
    h_imf.push_back(prob->fixed_node);
    h_imc.push_back(-1);
  }
  // the image-level graph through the plain machinery: Jacobi-PCG structures only, no initialisation
  gsfm_ra_problem ip = *prob;
  ip.num_nodes = NI;
  ip.fixed_node = fixed_img;
  ip.num_images = 0;
  gsfm_ra_options o2 = *opt;
  o2.force_iterative = 1;
  o2.skip_initialization = 1;
  RaDevice d;
  RaHostInit hi;
  std::vector<double> zeros_h;
  DevBuf<double> zeros_d;
  const double* rot_img0 = nullptr;
  if (mem == GSFM_MEM_DEVICE) {
    GSFM_HIP_CHECK(hipMemsetAsync(zeros_d.ensure(3 * (size_t)NI), 0, 3 * (size_t)NI * sizeof(double), s));
    rot_img0 = zeros_d.get();
  } else {
    zeros_h.assign(3 * (size_t)NI, 0.0);
    rot_img0 = zeros_h.data();
  }
  setup_device(ctx, &ip, &o2, rot_img0, d, hi, /*allow_blockdense=*/false);
  finish_init(ctx, &o2, d, hi);
  RaWs* ws = d.ws;
  if (!ctx->ra_rig_ws) {
    ctx->ra_rig_ws = new RigRaWs();
    ctx->ra_rig_ws_free = &RigRaWs::destroy;
  }
  RigRaWs* rw = static_cast<RigRaWs*>(ctx->ra_rig_ws);
  // tables
  std::vector<int> foff((size_t)N + 1, 0), fimg((size_t)NI), coff((size_t)C + 1, 0), cimg;
  for (int i = 0; i < NI; ++i) {
    foff[h_imf[i] + 1]++;
    if (h_imc[i] >= 0) coff[h_imc[i] + 1]++;
  }
  for (int f = 0; f < N; ++f) foff[f + 1] += foff[f];
  for (int c = 0; c < C; ++c) coff[c + 1] += coff[c];
  cimg.resize((size_t)coff[C] + 1);
  {
    std::vector<int> fc(foff.begin(), foff.end() - 1), cc(coff.begin(), coff.end() - 1);
    for (int i = 0; i < NI; ++i) {
      fimg[fc[h_imf[i]]++] = i;
      if (h_imc[i] >= 0) cimg[cc[h_imc[i]]++] = i;
    }
  }
  auto up = [&](DevBuf<int>& b, const std::vector<int>& v) {
    GSFM_HIP_CHECK(hipMemcpyAsync(b.ensure(v.size() + 1), v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s));
  };
  up(rw->img_frame, h_imf);
  up(rw->img_cam, h_imc);
  up(rw->foff, foff);
  up(rw->fimg, fimg);
  up(rw->coff, coff);
  up(rw->cimg, cimg);
  const int nred = N + C;
  for (DevBuf<double>* b : {&rw->rhs, &rw->x, &rw->r, &rw->wbuf, &rw->gat_s, &rw->gat_t, &rw->b, &rw->cx, &rw->cr, &rw->cz,
                            &rw->cp, &rw->cs})
    b->ensure(3 * (size_t)nred);
  rw->cw.ensure(3 * (size_t)nred + 2);
  rw->diag.ensure(nred);
  rw->minv.ensure(9 * (size_t)nred);
  rw->zimg.ensure(3 * (size_t)NI);
  rw->tmp_rot.ensure(3 * (size_t)std::max(C, 1));
  copy_in(ctx, rw->rotf.ensure(3 * (size_t)N), rot_inout, 3 * (size_t)N, mem);
  rw->rotc.ensure(3 * (size_t)std::max(C, 1));
  if (C > 0)
    GSFM_HIP_CHECK(hipMemcpyAsync(rw->rotc.get(), prob->cam_rot_aa, 3 * (size_t)C * sizeof(double), hipMemcpyHostToDevice, s));
  // the gauge frame is held at its initial rotation (gra.cc:248-257)
  GSFM_HIP_CHECK(hipMemcpyAsync(ws->fixed_rot0.get(), rw->rotf.get() + 3 * (size_t)prob->fixed_node, 3 * sizeof(double),
                                hipMemcpyDeviceToDevice, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));  // host tables go out of scope below
  RigSolve g;
  g.d = &d;
  g.rw = rw;
  g.r = RigRa{N, C, NI, rw->img_frame.get(), rw->img_cam.get(), rw->foff.get(), rw->fimg.get(), rw->coff.get(), rw->cimg.get()};
  g.nred = nred;
  g.gridRed = (N + kBlock - 1) / kBlock + C;
  g.gridI = grid_for(NI, kBlock);
  g.gridR = grid_for(nred, kBlock);
  const double t1 = now_seconds();
  long lin_iters = 0;
  int it_l1 = 0, it_irls = 0;
  double last_step = 0.0;
  double upd[3];
  const size_t n3 = 3 * (size_t)nred;

  // ---------------- L1 stage (gra.cc:479-541)
  if (opt->max_num_l1_iterations > 0) {
    const size_t rows3 = 3 * (size_t)(E + 1);
    ws->z.ensure(rows3);
    ws->u.ensure(rows3);
    ws->dz.ensure(rows3);
    launch_gather<GATHER_L1W>(d);
    rig_reduce<1>(g, ws->lap_diag.get(), rw->diag.get());
    double last_norm = 0.0, curr_norm = 0.0;
    rig_residuals(g, false, 0, 0.0);
    const double rows_total = 3.0 * (double)E + 3.0;
    for (int it = 0; it < opt->max_num_l1_iterations; ++it) {
      last_norm = curr_norm;
      GSFM_HIP_CHECK(hipMemsetAsync(ws->z.get(), 0, rows3 * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(ws->u.get(), 0, rows3 * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(ws->dz.get(), 0, rows3 * sizeof(double), s));
      GSFM_HIP_CHECK(hipMemsetAsync(rw->x.get(), 0, n3 * sizeof(double), s));
      launch_gather<GATHER_L1RHS>(d);
      rig_reduce<3>(g, ws->rhs.get(), rw->rhs.get());
      for (int a = 0; a < opt->l1_admm_max_num_iterations; ++a) {
        lin_iters += rig_pcg_solve(g, a > 0, a > 0 ? opt->pcg_relative_tolerance_admm : opt->pcg_relative_tolerance,
                                   opt->pcg_max_iterations);
        rig_expand(g, rw->x.get(), ws->x.get());  // A x per edge from the image-level step
        hipLaunchKernelGGL(k_admm_edge, dim3(d.gridE), dim3(kBlock), 0, s, E, d.has_gauge, d.fixed, ws->ei.get(), ws->ej.get(),
                           d.ew, ws->res.get(), ws->x.get(), ws->z.get(), ws->u.get(), ws->dz.get(), opt->l1_admm_alpha,
                           1.0 / opt->l1_admm_rho, ws->part_misc.get(), nullptr);
        hipLaunchKernelGGL((k_finalize<4>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), d.gridE, ws->scal.get());
        launch_gather<GATHER_L1RHS>(d);
        rig_reduce<3>(g, ws->rhs.get(), rw->rhs.get());
        rig_reduce<3>(g, ws->gat_s.get(), rw->gat_s.get());
        rig_reduce<3>(g, ws->gat_t.get(), rw->gat_t.get());
        hipLaunchKernelGGL(k_sumsq2, dim3(g.gridR), dim3(kBlock), 0, s, (long)n3, rw->gat_s.get(), rw->gat_t.get(),
                           ws->part_misc.get());
        hipLaunchKernelGGL((k_finalize<2>), dim3(1), dim3(kBlock), 0, s, ws->part_misc.get(), g.gridR, ws->scal.get() + 4);
        GSFM_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned + 16, ws->scal.get(), 6 * sizeof(double), hipMemcpyDeviceToHost, s));
        GSFM_HIP_CHECK(hipStreamSynchronize(s));
        const double* h = ctx->h_pinned + 16;
        const double r_norm = std::sqrt(h[0]), Ax_norm = std::sqrt(h[1]), z_norm = std::sqrt(h[2]), b_norm = std::sqrt(h[3]);
        const double rho = opt->l1_admm_rho;
        const double s_norm = rho * std::sqrt(h[4]);
        const double dual_norm = rho * std::sqrt(h[5]);
        const double primal_eps = std::sqrt(rows_total) * opt->l1_admm_absolute_tolerance +
                                  opt->l1_admm_relative_tolerance * std::max({Ax_norm, z_norm, b_norm});
        const double dual_eps = std::sqrt((double)n3) * opt->l1_admm_absolute_tolerance + opt->l1_admm_relative_tolerance * dual_norm;
        if (r_norm < primal_eps && s_norm < dual_eps) break;
      }
      rig_update(g, upd);
      it_l1 = it + 1;
      if (upd[2] > 0) {
        if (rep) rep->iterations_l1 = it_l1;
        return GSFM_ERR_NUMERICAL;
      }
      curr_norm = upd[1];
      last_step = upd[0];
      rig_residuals(g, false, 0, 0.0);
      if (upd[0] < opt->l1_step_convergence_threshold || std::fabs(last_norm - curr_norm) < 1e-12) break;
    }
  }

  // ---------------- IRLS stage (gra.cc:543-625)
  if (opt->max_num_irls_iterations > 0) {
    const double sigma = opt->irls_loss_parameter_sigma * M_PI / 180.0;
    rig_residuals(g, true, opt->weight_type, sigma * sigma);
    for (int it = 0; it < opt->max_num_irls_iterations; ++it) {
      if (read_nan_flag(d)) {
        if (rep) rep->iterations_irls = it_irls;
        return GSFM_ERR_NUMERICAL;
      }
      launch_gather<GATHER_IRLS>(d);
      rig_reduce<3>(g, ws->rhs.get(), rw->rhs.get());
      rig_reduce<1>(g, ws->lap_diag.get(), rw->diag.get());
      lin_iters += rig_pcg_solve(g, false, opt->pcg_relative_tolerance, opt->pcg_max_iterations);
      rig_update(g, upd);
      it_irls = it + 1;
      last_step = upd[0];
      rig_residuals(g, true, opt->weight_type, sigma * sigma);
      if (upd[0] < opt->irls_step_convergence_threshold) break;
    }
  }
  copy_out(ctx, rot_inout, rw->rotf.get(), 3 * (size_t)N, mem);
  if (C > 0)
    GSFM_HIP_CHECK(hipMemcpyAsync(prob->cam_rot_aa, rw->rotc.get(), 3 * (size_t)C * sizeof(double), hipMemcpyDeviceToHost, s));
  GSFM_HIP_CHECK(hipStreamSynchronize(s));
  const double t2 = now_seconds();
  if (rep) {
    rep->iterations_l1 = it_l1;
    rep->iterations_irls = it_irls;
    rep->iterations = it_l1 + it_irls;
    rep->linear_iterations = lin_iters;
    rep->termination = GSFM_TERM_CONVERGENCE;
    rep->seconds_total = t2 - t0;
    rep->seconds_solve = t2 - t1;
    rep->last_step_norm = last_step;
  }
  return GSFM_OK;
}

}  // namespace
}  // namespace gsfm

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
using namespace gsfm;

extern "C" void gsfm_ra_options_default(gsfm_ra_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->max_num_l1_iterations = 5;
  o->l1_step_convergence_threshold = 0.001;
  o->max_num_irls_iterations = 100;
  o->irls_step_convergence_threshold = 0.001;
  o->irls_loss_parameter_sigma = 5.0;
  o->weight_type = 0;
  o->skip_initialization = 0;
  o->use_weight = 0;
  o->use_gravity = 0;
  o->l1_admm_max_num_iterations = 10;
  o->l1_admm_rho = 1.0;
  o->l1_admm_alpha = 1.0;
  o->l1_admm_absolute_tolerance = 1e-4;
  o->l1_admm_relative_tolerance = 1e-2;
  o->pcg_relative_tolerance = 1e-10;
  o->pcg_max_iterations = 2000;
  o->force_iterative = 0;
  o->pcg_relative_tolerance_admm = 1e-10;
}

extern "C" int gsfm_ra_solve(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                             double* rot_aa_inout, gsfm_report* report) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  gsfm_report local{};
  if (!report) report = &local;
  std::memset(report, 0, sizeof(*report));
  FlatDump dump(ctx, "ra");
  if (dump.active() && prob && opt && rot_aa_inout) {
    const int64_t N = prob->num_nodes, E = prob->num_edges;
    dump.scalar("num_nodes", (double)N);
    dump.scalar("fixed_node", prob->fixed_node);
    dump.scalar("comm_rank", ctx->comm.rank);
    dump.scalar("comm_world", ctx->comm.world);
    dump.array("edge_i", prob->edge_i, {E}, prob->mem);
    dump.array("edge_j", prob->edge_j, {E}, prob->mem);
    dump.array("edge_q", prob->edge_q, {E, 4}, prob->mem);
    dump.array("edge_weight", prob->edge_weight, {E}, prob->mem);
    dump.array("edge_ninl", prob->edge_ninl, {E}, prob->mem);
    dump.array("node_aa0", rot_aa_inout, {N, 3}, prob->mem);
    if (prob->num_images > 0 && prob->image_frame && prob->image_cam) {
      dump.array("image_frame", prob->image_frame, {(int64_t)prob->num_images}, prob->mem);
      dump.array("image_cam", prob->image_cam, {(int64_t)prob->num_images}, prob->mem);
      if (prob->num_cams > 0 && prob->cam_rot_aa) dump.array("cam_aa0", prob->cam_rot_aa, {(int64_t)prob->num_cams, 3}, GSFM_MEM_HOST);
    }
    if (prob->node_gravity) dump.array("node_gravity", prob->node_gravity, {N}, prob->mem);
    GSFM_DUMP_OPT(dump, opt, max_num_l1_iterations);
    GSFM_DUMP_OPT(dump, opt, l1_step_convergence_threshold);
    GSFM_DUMP_OPT(dump, opt, max_num_irls_iterations);
    GSFM_DUMP_OPT(dump, opt, irls_step_convergence_threshold);
    GSFM_DUMP_OPT(dump, opt, irls_loss_parameter_sigma);
    GSFM_DUMP_OPT(dump, opt, weight_type);
    GSFM_DUMP_OPT(dump, opt, skip_initialization);
    GSFM_DUMP_OPT(dump, opt, use_weight);
    GSFM_DUMP_OPT(dump, opt, use_gravity);
    GSFM_DUMP_OPT(dump, opt, l1_admm_max_num_iterations);
    GSFM_DUMP_OPT(dump, opt, l1_admm_rho);
    GSFM_DUMP_OPT(dump, opt, l1_admm_alpha);
    GSFM_DUMP_OPT(dump, opt, l1_admm_absolute_tolerance);
    GSFM_DUMP_OPT(dump, opt, l1_admm_relative_tolerance);
    GSFM_DUMP_OPT(dump, opt, pcg_relative_tolerance);
    GSFM_DUMP_OPT(dump, opt, pcg_max_iterations);
    GSFM_DUMP_OPT(dump, opt, force_iterative);
    GSFM_DUMP_OPT(dump, opt, pcg_relative_tolerance_admm);
  }
  const int rc = guarded(ctx, report, [&] { return ra_solve_impl(ctx, prob, opt, rot_aa_inout, report); });
  if (dump.active() && prob && rot_aa_inout) {
    dump.array("out_rot_aa", rot_aa_inout, {(int64_t)prob->num_nodes, 3}, prob->mem);
    if (prob->num_images > 0 && prob->num_cams > 0 && prob->cam_rot_aa)
      dump.array("out_cam_rot_aa", prob->cam_rot_aa, {(int64_t)prob->num_cams, 3}, GSFM_MEM_HOST);
    dump.write(report, rc);
  }
  return rc;
}

extern "C" int gsfm_ra_residuals(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                                 const double* rot_aa, double* residual_out, double* weight_out) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(prob && opt && rot_aa, "RA: null argument");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    gsfm_ra_options o = *opt;
    o.skip_initialization = 1;
    RaDevice d;
    RaHostInit hi;
    setup_device(ctx, prob, &o, rot_aa, d, hi);
    finish_init(ctx, &o, d, hi);
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const double sigma = o.irls_loss_parameter_sigma * M_PI / 180.0;
    launch_residuals(d, true, o.weight_type, sigma * sigma);
    if (residual_out) copy_out(ctx, residual_out, d.ws->res.get(), 3 * (size_t)d.E, prob->mem);
    if (weight_out) copy_out(ctx, weight_out, d.ws->wirls.get(), (size_t)d.E, prob->mem);
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return (int)GSFM_OK;
  });
}

// `repeat` back-to-back launches of the per-edge residual + IRLS-weight sweep (k_edge_residual) between two HIP events on
// the ctx stream: the roofline measurement of SURVEY.md section 8d's K-RA-res on a graph that does not fit the caches.
extern "C" int gsfm_ra_residuals_timed(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const gsfm_ra_options* opt,
                                       const double* rot_aa, int repeat, double* avg_kernel_ms) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(prob && opt && rot_aa, "RA: null argument");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    gsfm_ra_options o = *opt;
    o.skip_initialization = 1;
    RaDevice d;
    RaHostInit hi;
    setup_device(ctx, prob, &o, rot_aa, d, hi);
    finish_init(ctx, &o, d, hi);
    RaWs* ws = d.ws;
    hipStream_t s = ctx->stream;
    const double sigma = o.irls_loss_parameter_sigma * M_PI / 180.0;
    launch_residuals(d, true, o.weight_type, sigma * sigma);  // node quaternions + one warm-up sweep
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    const int reps = repeat > 0 ? repeat : 1;
    GSFM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(k_edge_residual, dim3(d.gridE), dim3(kBlock), 0, s, d.E, ws->ei.get(), ws->ej.get(), ws->eq.get(),
                         ws->nq.get(), ws->res.get(), ws->wirls.get(), o.weight_type, sigma * sigma, ws->flags.get(),
                         (const unsigned char*)nullptr, (const double*)nullptr);
    GSFM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    if (avg_kernel_ms) {
      float ms = 0.f;
      GSFM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      *avg_kernel_ms = ms / reps;
    }
    return (int)GSFM_OK;
  });
}

extern "C" int gsfm_ra_laplacian_apply(gsfm_ctx* ctx, const gsfm_ra_problem* prob, const double* w,
                                       const double* x, double* y, int repeat, double* avg_kernel_ms) {
  if (!ctx) return GSFM_ERR_INVALID_ARGUMENT;
  return guarded(ctx, nullptr, [&] {
    GSFM_REQUIRE(prob && w && x && y, "RA: null argument");
    GSFM_HIP_CHECK(hipSetDevice(ctx->device));
    gsfm_ra_options o;
    gsfm_ra_options_default(&o);
    o.skip_initialization = 1;
    RaDevice d;
    std::vector<double> zeros(3 * (size_t)prob->num_nodes, 0.0);
    gsfm_ra_problem p2 = *prob;
    // rotations are irrelevant here; feed zeros from the host side of the same mem space
    double* rot_tmp = nullptr;
    if (prob->mem == GSFM_MEM_DEVICE) {
      GSFM_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&rot_tmp), zeros.size() * sizeof(double)));
      GSFM_HIP_CHECK(hipMemset(rot_tmp, 0, zeros.size() * sizeof(double)));
    }
    RaHostInit hi;
    setup_device(ctx, &p2, &o, prob->mem == GSFM_MEM_DEVICE ? rot_tmp : zeros.data(), d, hi);
    finish_init(ctx, &o, d, hi);
    GSFM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (rot_tmp) (void)hipFree(rot_tmp);
    RaWs* ws = d.ws;
    hipStream_t s = ctx->stream;
    // IRLS-style system with the given edge weights: inc_w[k] = w[e], diag = sum (+1 gauge)
    copy_in(ctx, ws->wirls.get(), w, (size_t)d.E, prob->mem);
    GSFM_HIP_CHECK(hipMemsetAsync(ws->res.get(), 0, 3 * (size_t)(d.E + 1) * sizeof(double), s));
    launch_gather<GATHER_IRLS>(d);
    copy_in(ctx, ws->x.get(), x, 3 * (size_t)d.N, prob->mem);
    const int reps = repeat > 0 ? repeat : 1;
    GSFM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
    for (int r = 0; r < reps; ++r) {
      dispatch_lpr(d.lpr, [&](auto L) {
        hipLaunchKernelGGL((k_spmv<decltype(L)::value>), dim3(d.gridRow), dim3(kBlock), 0, s, d.N,
                           ws->rowptr.get(), ws->nbr.get(), ws->inc_w.get(), ws->lap_diag_loc.get(),
                           ws->x.get(), ws->wbuf.get());
      });
    }
    GSFM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    copy_out(ctx, y, ws->wbuf.get(), 3 * (size_t)d.N, prob->mem);
    GSFM_HIP_CHECK(hipStreamSynchronize(s));
    if (avg_kernel_ms) {
      float ms = 0.f;
      GSFM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      *avg_kernel_ms = ms / reps;
    }
    return (int)GSFM_OK;
  });
}
