// ra_sub.hpp — substructured preconditioner for the rotation-averaging Laplacian (2048 < N <= 32768, one rank).
//
// The block-diagonal preconditioner of ra_dense.hpp (dense inverses of BFS-contiguous diagonal blocks) leaves the
// coupling BETWEEN the blocks to the PCG: 38 iterations per solve on the configs[3] ring, 2 069 per rotation averaging.
// Those iterations are entirely the coupling (CPU replay tools/exp_ra_linear_solves.py: 2 078 -> 111 with the interface
// treated exactly), so here the nodes are split once per solve into
//     I_b  interior of block b: every neighbour lies in block b          (b = 0 .. nblk-1)
//     G    interface: nodes with a neighbour in another block            (|G| <= kDenseMaxN, else the plain block path runs)
// and numbered [I_0 | I_1 | ... | G].  With A = [[A_II, A_IG], [A_GI, A_GG]], A_II block diagonal,
//     S = A_GG - sum_b A_GI_b A_bb^-1 A_I_bG           (dense |G| x |G| Schur complement)
//     A^-1 r:   y_I = A_II^-1 r_I,   u_G = S^-1 (r_G - A_GI y_I),   u_I = A_II^-1 (r_I - A_IG u_G)
// is an EXACT solve up to the fp32 rounding of the stored inverses: what is left for the PCG is two or three iterations.
// A_GI_b touches only the boundary layer C_b of block b (interior nodes with an interface neighbour), so
// S -= E_b^T (A_bb^-1)[C_b, C_b] E_b with the small dense E_b = -A_{C_b, G_b} — a few MFLOP per block.
// Smaller interiors than ra_dense.hpp's 2 016-node blocks pay twice: the inversions cost sum T_b^3 tile products and all
// blocks go through ONE batched symmetric sweep (k_gj_sweep_step of ra_dense.hpp, grid.z = block).
#pragma once

#include <vector>

#include "ra_dense.hpp"

namespace gsfm {

// Device view.  Block index nblk is the interface (its "inverse" is S^-1).
struct SubDev {
  int N = 0, NI = 0, nblk = 0;
  const int* ioff = nullptr;        // [nblk + 2] node ranges: block b = [ioff[b], ioff[b+1]); interface = [ioff[nblk], N)
  const int* ipad = nullptr;        // [nblk + 1] leading dimension of block b's stored inverse (multiple of kTile)
  const long* invoff = nullptr;     // [nblk + 1] offset of block b's inverse in `inv`
  const int* row_blk = nullptr;     // [N] block of a node (nblk for interface nodes)
  const unsigned char* in_c = nullptr;  // [N] interior node with an interface neighbour
  const float* inv = nullptr;
};

// ---- assembly -------------------------------------------------------------------------------------------------------
// Off-diagonal entries of all diagonal blocks at once: A_b[r - ioff_b][c - ioff_b] -= w for incidence entries inside one
// block (interiors: matrix z = b of `A`, leading dimension ld; interface: `S`, leading dimension lds).
static __global__ void __launch_bounds__(kBlock)
    k_sub_fill_offdiag(long nnz, const int* __restrict__ inc_row, const int* __restrict__ nbr, const double* __restrict__ inc_w,
                       SubDev sd, double* __restrict__ A, int ld, size_t zstride, double* __restrict__ S, int lds) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
    const int r = inc_row[k], c = nbr[k];
    const int b = sd.row_blk[r];
    if (b != sd.row_blk[c]) continue;
    const int o = sd.ioff[b];
    if (b < sd.nblk)
      unsafeAtomicAdd(A + zstride * b + (size_t)(r - o) * ld + (c - o), -inc_w[k]);
    else
      unsafeAtomicAdd(S + (size_t)(r - o) * lds + (c - o), -inc_w[k]);
  }
}
// diagonals (+= keeps self-loop terms) and the identity on the padding rows; one thread per padded row of every block
static __global__ void __launch_bounds__(kBlock)
    k_sub_fill_diag(SubDev sd, const double* __restrict__ lap_diag, double* __restrict__ A, int ld, size_t zstride,
                    double* __restrict__ S, int lds) {
  const int b = blockIdx.y;
  const int o = sd.ioff[b], cnt = (b < sd.nblk ? sd.ioff[b + 1] : sd.N) - o, pad = sd.ipad[b];
  double* M = b < sd.nblk ? A + zstride * b : S;
  const int l = b < sd.nblk ? ld : lds;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pad; i += gridDim.x * blockDim.x)
    M[(size_t)i * l + i] = i < cnt ? M[(size_t)i * l + i] + lap_diag[o + i] : 1.0;
}
// fp32 copy of an inverse (swept buffer of k_gj_sweep_step, leading dimension ld) into its slot of the preconditioner storage
static __global__ void __launch_bounds__(kBlock)
    k_sub_to_f32(SubDev sd, int b0, const double* __restrict__ A, int ld, size_t zstride, float* __restrict__ inv) {
  const int b = b0 + blockIdx.y;
  const int pad = sd.ipad[b];
  const double* M = A + zstride * blockIdx.y;
  float* out = inv + sd.invoff[b];
  const size_t nn = (size_t)pad * pad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / pad, c = i % pad;
    out[i] = (float)gj_inv_at(M, ld, (int)r, (int)c);
  }
}

// Coupling tables of block b = blockIdx.y (host-built, concatenated over the blocks):
//   coff / goff  [nblk + 1]  ranges of the block's boundary layer C_b and of its adjacent interface nodes G_b
//   cloc[c]      local index (node - ioff_b) of boundary node c;   gloc[g]  interface index (node - NI) of G_b's g
//   eoff [nblk + 1], ec / eg / eslot: coupling entries (local c, local g, incidence slot of the weight)
//   woff [nblk + 1]  offsets of the block's dense scratch (E_b | M_b | Y_b) in `work`
struct SubCouple {
  const int *coff = nullptr, *goff = nullptr, *cloc = nullptr, *gloc = nullptr, *eoff = nullptr, *ec = nullptr, *eg = nullptr,
            *eslot = nullptr;
  const long* woff = nullptr;
  double* work = nullptr;
};
__device__ __forceinline__ void sub_scratch(const SubCouple& sc, int b, int& C, int& G, double*& E, double*& M, double*& Y) {
  C = sc.coff[b + 1] - sc.coff[b];
  G = sc.goff[b + 1] - sc.goff[b];
  E = sc.work + sc.woff[b];
  M = E + (size_t)C * G;
  Y = M + (size_t)C * C;
}
// E_b[c][g] = sum of the weights of the edges between boundary node c and interface node g (= -A_{c g}); M_b = (A_bb^-1)[C_b, C_b]
static __global__ void __launch_bounds__(kBlock)
    k_sub_couple_EM(SubDev sd, SubCouple sc, const double* __restrict__ inc_w, const double* __restrict__ Ainv, int ld,
                    size_t zstride) {
  const int b = blockIdx.y;
  int C, G;
  double *E, *M, *Y;
  sub_scratch(sc, b, C, G, E, M, Y);
  const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long)gridDim.x * blockDim.x;
  for (long e = sc.eoff[b] + tid; e < sc.eoff[b + 1]; e += nth)
    unsafeAtomicAdd(E + (size_t)sc.ec[e] * G + sc.eg[e], inc_w[sc.eslot[e]]);
  const double* A = Ainv + zstride * b;
  const int* cl = sc.cloc + sc.coff[b];
  for (long i = tid; i < (long)C * C; i += nth) {
    const int c1 = (int)(i / C), c2 = (int)(i % C);
    M[i] = gj_inv_at(A, ld, cl[c1], cl[c2]);
  }
}
// Y_b = M_b E_b
static __global__ void __launch_bounds__(kBlock) k_sub_couple_Y(SubCouple sc) {
  const int b = blockIdx.y;
  int C, G;
  double *E, *M, *Y;
  sub_scratch(sc, b, C, G, E, M, Y);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)C * G; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i / G), g = (int)(i % G);
    double acc = 0.0;
    for (int c2 = 0; c2 < C; ++c2) acc += M[(size_t)c * C + c2] * E[(size_t)c2 * G + g];
    Y[i] = acc;
  }
}
// S[g1][g2] -= sum_c E_b[c][g1] Y_b[c][g2], all blocks in one launch (blockIdx.y = block).  An interface node touches
// two blocks at most on a banded graph, so an entry of S receives at most two contributions (atomic adds: their order
// does not matter for two terms; S only feeds the preconditioner).
static __global__ void __launch_bounds__(kBlock) k_sub_couple_S(SubCouple sc, double* __restrict__ S, int lds) {
  const int b = blockIdx.y;
  int C, G;
  double *E, *M, *Y;
  sub_scratch(sc, b, C, G, E, M, Y);
  const int* gl = sc.gloc + sc.goff[b];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)G * G; i += (long)gridDim.x * blockDim.x) {
    const int g1 = (int)(i / G), g2 = (int)(i % G);
    double acc = 0.0;
    for (int c = 0; c < C; ++c) acc += E[(size_t)c * G + g1] * Y[(size_t)c * G + g2];
    unsafeAtomicAdd(S + (size_t)gl[g1] * lds + gl[g2], -acc);
  }
}

// ---- application ----------------------------------------------------------------------------------------------------
// y[n] = sum_m inv_b(n)[n][m] v[m] over the nodes n in [n0, n1) (3 right-hand sides; one wave per row; fp32 inverses, f64
// sums).  test: the convergence test of the PCG iteration (ra_dense.hpp k_bd_apply3) runs in the prologue.
static __global__ void __launch_bounds__(kBlock)
    k_sub_apply3(SubDev sd, int n0, int n1, const double* __restrict__ v, double* __restrict__ y, int test, int it,
                 double tol2, const double* __restrict__ rpart, DpcgState* st) {
  __shared__ double smem[5];
  if (st->done) return;
  if (test && it > 0) {
    const double rr = bd_reduce1(rpart, kBdUpdateBlocks, smem);
    if (rr <= tol2 * st->bb) {  // every block takes the same decision from the same slots
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->rr = rr;
        st->done = 1;
      }
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  for (int n = n0 + wave; n < n1; n += nwaves) {
    const int b = sd.row_blk[n];
    const int b0 = sd.ioff[b];
    const int cols = (b < sd.nblk ? sd.ioff[b + 1] : sd.N) - b0;
    const float* row = sd.inv + sd.invoff[b] + (size_t)(n - b0) * sd.ipad[b];  // 128-byte aligned (ipad: multiple of 32)
    const double* vb = v + 3 * (size_t)b0;
    double a0 = 0, a1 = 0, a2 = 0;
    const int cols4 = cols & ~3;
    for (int m = 4 * lane; m < cols4; m += 256) {  // 16-byte loads of the fp32 row: a quarter of the load instructions
      const float4 w = *reinterpret_cast<const float4*>(row + m);
      const double* vm = vb + 3 * (size_t)m;
      a0 += (double)w.x * vm[0] + (double)w.y * vm[3] + (double)w.z * vm[6] + (double)w.w * vm[9];
      a1 += (double)w.x * vm[1] + (double)w.y * vm[4] + (double)w.z * vm[7] + (double)w.w * vm[10];
      a2 += (double)w.x * vm[2] + (double)w.y * vm[5] + (double)w.z * vm[8] + (double)w.w * vm[11];
    }
    for (int m = cols4 + lane; m < cols; m += 64) {
      const double w = (double)row[m];
      a0 += w * vb[3 * m];
      a1 += w * vb[3 * m + 1];
      a2 += w * vb[3 * m + 2];
    }
    a0 = group_sum<64>(a0);
    a1 = group_sum<64>(a1);
    a2 = group_sum<64>(a2);
    if (lane == 0) {
      y[3 * (size_t)n] = a0;
      y[3 * (size_t)n + 1] = a1;
      y[3 * (size_t)n + 2] = a2;
    }
  }
}

// Coupling sweeps over the CSR-by-node incidence list (one wave per row):
//   up   (rows n >= NI): out[n] = r[n] + sum_{nbr < NI}  w y[nbr]        = r_G - A_GI y_I
//   down (rows n <  NI): out[n] = r[n] + sum_{nbr >= NI} w u[nbr]        = r_I - A_IG u_G   (plain copy outside the boundary layer)
template <bool UP>
static __global__ void __launch_bounds__(kBlock)
    k_sub_couple(SubDev sd, const int* __restrict__ rowptr, const int* __restrict__ nbr, const double* __restrict__ inc_w,
                 const double* __restrict__ r, const double* __restrict__ src, double* __restrict__ out, const DpcgState* st) {
  if (st->done) return;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (kBlock / 64);
  const int n0 = UP ? sd.NI : 0, n1 = UP ? sd.N : sd.NI;
  for (int n = n0 + wave; n < n1; n += nwaves) {
    double a0 = 0, a1 = 0, a2 = 0;
    if (UP || sd.in_c[n]) {
      for (int k = rowptr[n] + lane; k < rowptr[n + 1]; k += 64) {
        const int m = nbr[k];
        if (UP ? m < sd.NI : m >= sd.NI) {
          const double w = inc_w[k];
          a0 += w * src[3 * (size_t)m];
          a1 += w * src[3 * (size_t)m + 1];
          a2 += w * src[3 * (size_t)m + 2];
        }
      }
      a0 = group_sum<64>(a0);
      a1 = group_sum<64>(a1);
      a2 = group_sum<64>(a2);
    }
    if (lane == 0) {
      out[3 * (size_t)n] = r[3 * (size_t)n] + a0;
      out[3 * (size_t)n + 1] = r[3 * (size_t)n + 1] + a1;
      out[3 * (size_t)n + 2] = r[3 * (size_t)n + 2] + a2;
    }
  }
}

// ---- host-side plan ---------------------------------------------------------------------------------------------------
struct SubPlan {
  bool ok = false;
  int nblk = 0, NI = 0, G = 0;
  int pmax = 0;                    // largest padded interior size
  std::vector<int> newpos;         // BFS position -> final position
  std::vector<int> ioff, ipad;     // see SubDev
  std::vector<long> invoff;
  long inv_total = 0;
  std::vector<int> row_blk;
  std::vector<unsigned char> in_c;
  // coupling tables (SubCouple), filled by sub_build_coupling from the final CSR
  std::vector<int> coff, goff, cloc, gloc, eoff, ec, eg, eslot;
  std::vector<long> woff;
  long work_total = 0;
  int cmax = 0, gmax = 0;
};

// Interface size and interior sizes when the BFS order is cut into blocks of `nbn` positions.  Returns false as soon as the
// interface exceeds `gcap`.
inline bool sub_count(int N, long E, const int* ei, const int* ej, int nbn, int gcap, std::vector<unsigned char>& isif, int& G) {
  isif.assign(N, 0);
  G = 0;
  for (long e = 0; e < E; ++e) {
    const int p = ei[e], q = ej[e];
    if (p / nbn != q / nbn) {
      if (!isif[p]) {
        isif[p] = 1;
        ++G;
      }
      if (!isif[q]) {
        isif[q] = 1;
        ++G;
      }
      if (G > gcap) return false;
    }
  }
  return true;
}

// Chooses the number of blocks (cost model: sum of T^3 tile products over the interior inversions and the Schur
// complement's) and numbers the nodes [I_0 | ... | I_{nblk-1} | G].  ei / ej: edges in BFS positions.
inline void sub_plan(int N, long E, const int* ei, const int* ej, SubPlan& sp) {
  sp.ok = false;
  const int gcap = kDenseMaxN;
  const int nblk0 = (N + kDenseMaxN - 1) / kDenseMaxN;
  std::vector<unsigned char> isif;
  int G0 = 0;
  if (nblk0 < 2 || !sub_count(N, E, ei, ej, (N + nblk0 - 1) / nblk0, gcap, isif, G0)) return;
  // interface nodes per cut, then the block count that minimises the modelled inversion work
  const double per_cut = (double)G0 / (nblk0 - 1);
  auto cost = [&](int nb) {
    const double G = per_cut * (nb - 1);
    if (G > 0.9 * gcap || G >= N) return 1e300;
    const double ti = ((N - G) / nb) / kTile, tg = G / kTile;
    return nb * ti * ti * ti + tg * tg * tg;
  };
  int best = nblk0;
  for (int nb = nblk0 + 1; nb <= 64; ++nb)
    if (cost(nb) < cost(best)) best = nb;
  int nblk = best, G = G0;
  while (nblk > nblk0 && !sub_count(N, E, ei, ej, (N + nblk - 1) / nblk, gcap, isif, G)) --nblk;
  if (nblk == nblk0) sub_count(N, E, ei, ej, (N + nblk0 - 1) / nblk0, gcap, isif, G);
  const int nbn = (N + nblk - 1) / nblk;
  nblk = (N + nbn - 1) / nbn;
  sp.nblk = nblk;
  sp.G = G;
  sp.NI = N - G;
  sp.newpos.assign(N, -1);
  sp.ioff.assign(nblk + 2, 0);
  sp.row_blk.assign(N, 0);
  int next = 0;
  for (int b = 0; b < nblk; ++b) {
    sp.ioff[b] = next;
    for (int p = b * nbn; p < std::min(N, (b + 1) * nbn); ++p)
      if (!isif[p]) {
        sp.row_blk[next] = b;
        sp.newpos[p] = next++;
      }
  }
  sp.ioff[nblk] = next;
  for (int p = 0; p < N; ++p)
    if (isif[p]) {
      sp.row_blk[next] = nblk;
      sp.newpos[p] = next++;
    }
  sp.ioff[nblk + 1] = N;
  sp.ipad.assign(nblk + 1, kTile);
  sp.invoff.assign(nblk + 1, 0);
  sp.pmax = kTile;
  long off = 0;
  for (int b = 0; b <= nblk; ++b) {
    const int cnt = sp.ioff[b + 1] - sp.ioff[b];
    sp.ipad[b] = std::max(kTile, ((cnt + kTile - 1) / kTile) * kTile);
    if (b < nblk) sp.pmax = std::max(sp.pmax, sp.ipad[b]);
    sp.invoff[b] = off;
    off += (long)sp.ipad[b] * sp.ipad[b];
  }
  sp.inv_total = off;
  sp.ok = sp.G > 0 && sp.NI > 0;
}

// Coupling tables from the final CSR (rowptr / nbr in final node numbers).  Returns false when the dense coupling work of
// some block would be out of proportion (graphs without locality: the caller keeps the plain block preconditioner).
inline bool sub_build_coupling(int N, const int* rowptr, const int* nbr, SubPlan& sp) {
  const int nblk = sp.nblk, NI = sp.NI;
  sp.in_c.assign(N, 0);
  std::vector<std::vector<int>> cn(nblk), gn(nblk);   // boundary nodes / adjacent interface nodes per block (ascending)
  std::vector<int> gmark(nblk, -1);
  for (int j = NI; j < N; ++j) {
    for (int k = rowptr[j]; k < rowptr[j + 1]; ++k) {
      const int q = nbr[k];
      if (q >= NI) continue;
      const int b = sp.row_blk[q];
      if (gmark[b] != j) {
        gmark[b] = j;
        gn[b].push_back(j);
      }
      if (!sp.in_c[q]) {
        sp.in_c[q] = 1;
        cn[b].push_back(q);
      }
    }
  }
  std::vector<int> cidx(N, -1);
  sp.coff.assign(nblk + 1, 0);
  sp.goff.assign(nblk + 1, 0);
  sp.eoff.assign(nblk + 1, 0);
  sp.woff.assign(nblk + 1, 0);
  sp.cloc.clear();
  sp.gloc.clear();
  sp.cmax = sp.gmax = 0;
  double work = 0.0;
  for (int b = 0; b < nblk; ++b) {
    std::sort(cn[b].begin(), cn[b].end());
    for (size_t c = 0; c < cn[b].size(); ++c) {
      cidx[cn[b][c]] = (int)c;
      sp.cloc.push_back(cn[b][c] - sp.ioff[b]);
    }
    for (int j : gn[b]) sp.gloc.push_back(j - NI);
    const long C = (long)cn[b].size(), Gb = (long)gn[b].size();
    sp.coff[b + 1] = sp.coff[b] + (int)C;
    sp.goff[b + 1] = sp.goff[b] + (int)Gb;
    sp.woff[b + 1] = sp.woff[b] + 2 * C * Gb + C * C;
    sp.cmax = std::max(sp.cmax, (int)C);
    sp.gmax = std::max(sp.gmax, (int)Gb);
    work += (double)C * C * Gb + (double)Gb * Gb * C;
  }
  sp.work_total = sp.woff[nblk];
  if (work > 2e10 || sp.work_total > (1L << 28)) return false;
  // entries grouped by block, interface nodes ascending inside a block
  sp.ec.clear();
  sp.eg.clear();
  sp.eslot.clear();
  for (int b = 0; b < nblk; ++b) {
    for (size_t g = 0; g < gn[b].size(); ++g) {
      const int j = gn[b][g];
      for (int k = rowptr[j]; k < rowptr[j + 1]; ++k) {
        const int q = nbr[k];
        if (q < NI && sp.row_blk[q] == b) {
          sp.ec.push_back(cidx[q]);
          sp.eg.push_back((int)g);
          sp.eslot.push_back(k);
        }
      }
    }
    sp.eoff[b + 1] = (int)sp.ec.size();
  }
  return true;
}

}  // namespace gsfm
