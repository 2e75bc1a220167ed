// Jacobi-preconditioned CG of the PD global step, two launches per iteration (k_cg_ap, k_cg_update): the form the
// contact-heavy graph variant and the SELL experiments use; pd_cg1_kernels.hip holds the one-launch-per-iteration form.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "pd_cg_device.h"

namespace pies {

// (K + C) x for the nodes with contacts, before k_cg_init (the residual needs the complete row; inside the CG iterations
// the contact rows are summed by extra blocks of k_cg_ap itself)
__global__ void __launch_bounds__(kBlock) k_contact_rows(CgArrays A, const float4* __restrict__ x) {
  const uint32_t used = *A.tUsedCount, lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, nwaves = (gridDim.x * kBlock) >> 6;
  for (uint32_t u = wave; u < used; u += nwaves) {
    const uint32_t node = A.tUsed[u];
    float sx, sy, sz;
    contact_rows_of_node(A, node, lane, [&](uint32_t j, float& qx, float& qy, float& qz) { const float4 v = x[j]; qx = v.x; qy = v.y; qz = v.z; }, sx, sy, sz);
    if (lane == 0) A.cAp[node] = make_float4(sx, sy, sz, 0.f);
  }
}

// r = f - (K + C) x ; z = D^-1 r ; partB = {rz, rr} ; partI = {bb}.   LPR lanes per row (sliced ELL, see CgArrays).
// prevPartB != nullptr: an extra block closes the previous solve's statistics (its scal[] entries are still intact:
// this solve's k_cg_ap(0) is the first kernel to overwrite them).
template <int LPR>
__global__ void __launch_bounds__(kBlock) k_cg_init(CgArrays A, const float4* __restrict__ x, const float4* __restrict__ f,
                                                    const float* __restrict__ prevPartB) {
  if (blockIdx.x == A.nparts) {  // one block behind the SpMV blocks: bookkeeping only (inside block 0 it delayed that block's rows by 2-3 us)
    if (prevPartB) solve_statistics(A, prevPartB);
    if (threadIdx.x == 0) {
      A.scal[10] = 0.0f;  // this solve has not converged yet (read by k_cg_ap / k_cg_update)
      A.ticket[0] = 0u;   // grid barrier counter of the solve's last k_cg_update
      A.ticket[1] = 0u;   // its abort word
    }
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const SliceSweep sw = slice_sweep<LPR>(A.n, A.nparts);
  float acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
    const uint32_t i = sl * (64u / LPR) + lane / LPR;
    // (K + C) x in double, like k_cg1_init: the row's products cancel against f to a residual five orders smaller
    double dx = 0.0, dy = 0.0, dz = 0.0;
    if (LPR == 1 && A.rowStencil) {  // row dictionary: the row's (column - row, value) pairs, shared by every row like it
      if (i < A.n) {
        const uint32_t rs = A.rowStencil[i];
        const uint32_t b = rs & 0xffffffu, e = b + (rs >> 24);
#pragma unroll 4
        for (uint32_t k = b; k < e; ++k) {
          const int2 p = A.stencil[k];
          const float a = __int_as_float(p.y);
          const float4 xj = x[static_cast<uint32_t>(static_cast<int>(i) + p.x)];
          dx = fma(static_cast<double>(a), static_cast<double>(xj.x), dx);
          dy = fma(static_cast<double>(a), static_cast<double>(xj.y), dy);
          dz = fma(static_cast<double>(a), static_cast<double>(xj.z), dz);
        }
      }
    } else {
      const uint32_t off = A.sliceOff[sl], width = (A.sliceOff[sl + 1] - off) >> 6;
#pragma unroll 4
      for (uint32_t k = 0; k < width; ++k) {
        const uint32_t at = off + (k << 6) + lane;
        const float a = A.val[at];
        const float4 xj = x[A.col[at]];
        dx = fma(static_cast<double>(a), static_cast<double>(xj.x), dx);
        dy = fma(static_cast<double>(a), static_cast<double>(xj.y), dy);
        dz = fma(static_cast<double>(a), static_cast<double>(xj.z), dz);
      }
    }
    float sx = 0.f, sy = 0.f, sz = 0.f;  // the contact part of the row, and (several lanes per row: the SELL experiments) the lanes' shares
    if (LPR > 1) {
      sx = static_cast<float>(dx); sy = static_cast<float>(dy); sz = static_cast<float>(dz);
      dx = dy = dz = 0.0;
      row_combine<LPR>(sx, sy, sz);
    }
    if (i < A.n && lane % LPR == 0u) {
      if (A.useCAp) {
        if (*A.tUsedCount != 0u && A.tIncCnt[i]) {
          const float4 c = A.cAp[i]; sx += c.x; sy += c.y; sz += c.z;
        }
      } else {
        contact_row(A, i, [&](uint32_t j, float& px, float& py, float& pz) { const float4 v = x[j]; px = v.x; py = v.y; pz = v.z; }, sx, sy, sz);
      }
      const float4 xi = x[i], fi = f[i];
      const float cd = A.cdiag[i], di = A.dinv[i];
      const double cdd = static_cast<double>(cd);
      const float rx = static_cast<float>(static_cast<double>(fi.x) - (fma(cdd, static_cast<double>(xi.x), dx) + static_cast<double>(sx)));
      const float ry = static_cast<float>(static_cast<double>(fi.y) - (fma(cdd, static_cast<double>(xi.y), dy) + static_cast<double>(sy)));
      const float rz = static_cast<float>(static_cast<double>(fi.z) - (fma(cdd, static_cast<double>(xi.z), dz) + static_cast<double>(sz)));
      const float zx = di * rx, zy = di * ry, zz = di * rz;
      A.r[i] = make_float4(rx, ry, rz, 0.f);
      A.z[i] = make_float4(zx, zy, zz, 0.f);
      acc9[0] += rx * zx; acc9[1] += ry * zy; acc9[2] += rz * zz;
      acc9[3] += rx * rx; acc9[4] += ry * ry; acc9[5] += rz * rz;
      acc9[6] += fi.x * fi.x; acc9[7] += fi.y * fi.y; acc9[8] += fi.z * fi.z;
    }
  }
  block_write_partial<9>(acc9, A.partI, 9);
}

// The SpMV rows of iteration k for this workgroup's slices: p = z + beta p_old (written), Ap = (K + C) p (written; the contact
// part only when inlineContacts, i.e. summed by the row's lane), acc += p.Ap.
template <int LPR> PIES_DEV void cg_ap_rows(const CgArrays& A, int k, const float beta[3], bool inlineContacts, float acc[3]) {
  const float4* __restrict__ pold = A.p[(k + 1) & 1];
  float4* __restrict__ pnew = A.p[k & 1];
  const uint32_t lane = threadIdx.x & 63u;
  const SliceSweep sw = slice_sweep<LPR>(A.n, A.nparts);
  for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
    const uint32_t i = sl * (64u / LPR) + lane / LPR;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (LPR == 1 && A.rowStencil) {  // row dictionary (see k_cg_init)
      if (i < A.n) {
        const uint32_t rs = A.rowStencil[i];
        const uint32_t sb = rs & 0xffffffu, se = sb + (rs >> 24);
        if (k > 0) {
#pragma unroll 4
          for (uint32_t q = sb; q < se; ++q) {
            const int2 pr = A.stencil[q];
            const float a = __int_as_float(pr.y);
            const uint32_t j = static_cast<uint32_t>(static_cast<int>(i) + pr.x);
            const float4 zj = A.z[j], pj = pold[j];
            sx = fmaf(a, fmaf(beta[0], pj.x, zj.x), sx);
            sy = fmaf(a, fmaf(beta[1], pj.y, zj.y), sy);
            sz = fmaf(a, fmaf(beta[2], pj.z, zj.z), sz);
          }
        } else {
#pragma unroll 4
          for (uint32_t q = sb; q < se; ++q) {
            const int2 pr = A.stencil[q];
            const float a = __int_as_float(pr.y);
            const float4 zj = A.z[static_cast<uint32_t>(static_cast<int>(i) + pr.x)];
            sx = fmaf(a, zj.x, sx);
            sy = fmaf(a, zj.y, sy);
            sz = fmaf(a, zj.z, sz);
          }
        }
      }
    } else {
    const uint32_t off = A.sliceOff[sl], width = (A.sliceOff[sl + 1] - off) >> 6;
    if (k > 0) {
#pragma unroll 4
      for (uint32_t kk = 0; kk < width; ++kk) {
        const uint32_t at = off + (kk << 6) + lane;
        const float a = A.val[at];
        const uint32_t j = A.col[at];
        const float4 zj = A.z[j], pj = pold[j];
        sx = fmaf(a, fmaf(beta[0], pj.x, zj.x), sx);
        sy = fmaf(a, fmaf(beta[1], pj.y, zj.y), sy);
        sz = fmaf(a, fmaf(beta[2], pj.z, zj.z), sz);
      }
    } else {
#pragma unroll 4
      for (uint32_t kk = 0; kk < width; ++kk) {
        const uint32_t at = off + (kk << 6) + lane;
        const float a = A.val[at];
        const float4 zj = A.z[A.col[at]];
        sx = fmaf(a, zj.x, sx);
        sy = fmaf(a, zj.y, sy);
        sz = fmaf(a, zj.z, sz);
      }
    }
    }
    row_combine<LPR>(sx, sy, sz);
    if (i < A.n && lane % LPR == 0u) {
      if (inlineContacts) {
        contact_row(A, i, [&](uint32_t j, float& qx, float& qy, float& qz) {
          const float4 zj = A.z[j];
          qx = zj.x; qy = zj.y; qz = zj.z;
          if (k > 0) {
            const float4 pj = pold[j];
            qx = fmaf(beta[0], pj.x, qx); qy = fmaf(beta[1], pj.y, qy); qz = fmaf(beta[2], pj.z, qz);
          }
        }, sx, sy, sz);
      }
      const float4 zi = A.z[i];
      float px = zi.x, py = zi.y, pz = zi.z;
      if (k > 0) {
        const float4 pi = pold[i];
        px = fmaf(beta[0], pi.x, px);
        py = fmaf(beta[1], pi.y, py);
        pz = fmaf(beta[2], pi.z, pz);
      }
      const float cd = A.cdiag[i];
      const float ax = fmaf(cd, px, sx), ay = fmaf(cd, py, sy), az = fmaf(cd, pz, sz);
      pnew[i] = make_float4(px, py, pz, 0.f);
      A.ap[i] = make_float4(ax, ay, az, 0.f);
      acc[0] += px * ax;
      acc[1] += py * ay;
      acc[2] += pz * az;
    }
  }
}

// iteration k:  beta = rz_k / rz_{k-1} (0 for k = 0) ; p = z + beta p_old ; Ap = (K + C) p ; partA = {pAp}
// With useCAp the launch carries kCgRowBlocks extra blocks behind the nparts SpMV blocks: they sum the contact rows of p
// (one wavefront per node, written to cAp) and add their share of p.Ap to partA; k_cg_update adds cAp to Ap.  (Round 2: a
// launch of their own before every k_cg_ap, 100 launches per substep of a contact scene.)
template <int LPR> __global__ void __launch_bounds__(kBlock) k_cg_ap(CgArrays A, int k, float tol2) {
  // The solve converged in an earlier iteration: nothing to do, and nothing to read but this word (without it every block
  // of the remaining captured launches re-reduced the residual partials to find that out: 4.7 instead of 2.5 us per launch
  // at 100k rows, and half of config 3's CG launches are such exits).
  if (A.scal[10] != 0.0f) return;
  float red[9];
  float rz[3], rr[3], bb[3];
  if (k == 0) {
    block_reduce_partials<9>(A.partI, 9, A.nparts, red);
#pragma unroll
    for (int c = 0; c < 3; ++c) { rz[c] = red[c]; rr[c] = red[3 + c]; bb[c] = red[6 + c]; }
  } else {
    block_reduce_partials<6>(A.partB, 6, A.nparts, red);
#pragma unroll
    for (int c = 0; c < 3; ++c) { rz[c] = red[c]; rr[c] = red[3 + c]; bb[c] = A.scal[6 + c]; }
  }
  if (k == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    A.scal[6] = bb[0];
    A.scal[7] = bb[1];
    A.scal[8] = bb[2];
    A.scal[9] = 0.0f;  // iterations started in this solve
  }
  if (all_converged(rr, bb, tol2)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      A.scal[11] = k == 0 ? 0.0f : static_cast<float>(1 + (k & 1));  // where the final residual partials are (solve_statistics)
      A.scal[10] = 1.0f;
    }
    return;
  }
  float beta[3] = {0.f, 0.f, 0.f};
  if (k > 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float old = A.scal[3 * ((k - 1) & 1) + c];
      beta[c] = old > 0.0f ? rz[c] / old : 0.0f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) A.scal[3 * (k & 1) + c] = rz[c];
    A.scal[9] = static_cast<float>(k + 1);
  }
  const float4* __restrict__ pold = A.p[(k + 1) & 1];
  const uint32_t lane = threadIdx.x & 63u;
  float acc[3] = {0, 0, 0};
  if (blockIdx.x >= A.nparts) {  // contact rows of p = z + beta p_old
    auto fetch = [&](uint32_t j, float& qx, float& qy, float& qz) {
      const float4 zj = A.z[j];
      qx = zj.x; qy = zj.y; qz = zj.z;
      if (k > 0) {
        const float4 pj = pold[j];
        qx = fmaf(beta[0], pj.x, qx); qy = fmaf(beta[1], pj.y, qy); qz = fmaf(beta[2], pj.z, qz);
      }
    };
    const uint32_t used = *A.tUsedCount;
    const uint32_t wave = ((blockIdx.x - A.nparts) * kBlock + threadIdx.x) >> 6, nwaves = ((gridDim.x - A.nparts) * kBlock) >> 6;
    for (uint32_t u = wave; u < used; u += nwaves) {
      const uint32_t node = A.tUsed[u];
      float sx, sy, sz;
      contact_rows_of_node(A, node, lane, fetch, sx, sy, sz);
      if (lane == 0) {
        A.cAp[node] = make_float4(sx, sy, sz, 0.f);
        float px, py, pz;
        fetch(node, px, py, pz);
        acc[0] += px * sx; acc[1] += py * sy; acc[2] += pz * sz;
      }
    }
    block_write_partial<3>(acc, A.partA, 3);
    return;
  }
  cg_ap_rows<LPR>(A, k, beta, !A.useCAp, acc);
  block_write_partial<3>(acc, A.partA, 3);
}

// x += alpha p ; r -= alpha Ap ; z = D^-1 r for this workgroup's rows; acc += {r.z, r.r} per column
PIES_DEV void cg_update_rows(const CgArrays& A, float4* __restrict__ x, int k, const float alpha[3], bool addCAp, float acc[6]) {
  const float4* __restrict__ p = A.p[k & 1];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < A.n; i += gridDim.x * kBlock) {
    const float4 pi = p[i];
    float4 api = A.ap[i];
    if (addCAp && *A.tUsedCount != 0u && A.tIncCnt[i]) {
      const float4 c = A.cAp[i];
      api.x += c.x; api.y += c.y; api.z += c.z;
    }
    float4 xi = x[i], ri = A.r[i];
    xi.x = fmaf(alpha[0], pi.x, xi.x);
    xi.y = fmaf(alpha[1], pi.y, xi.y);
    xi.z = fmaf(alpha[2], pi.z, xi.z);
    ri.x = fmaf(-alpha[0], api.x, ri.x);
    ri.y = fmaf(-alpha[1], api.y, ri.y);
    ri.z = fmaf(-alpha[2], api.z, ri.z);
    const float di = A.dinv[i];
    const float zx = di * ri.x, zy = di * ri.y, zz = di * ri.z;
    x[i] = xi;
    A.r[i] = ri;
    A.z[i] = make_float4(zx, zy, zz, 0.f);
    acc[0] += ri.x * zx; acc[1] += ri.y * zy; acc[2] += ri.z * zz;
    acc[3] += ri.x * ri.x; acc[4] += ri.y * ri.y; acc[5] += ri.z * ri.z;
  }
}

// The continuation's grid barrier needs every workgroup of the launch resident at once: how many k_cg_update workgroups the
// device holds (pd_setup.cpp sizes the CG kernels' grid below it)
uint32_t cg_update_resident_blocks(int device);

// alpha = rz_k / pAp ; x += alpha p ; r -= alpha Ap ; z = D^-1 r ; partB = {rz_{k+1}, rr_{k+1}}.
// overflow > 0: this is the solve's last captured iteration.  If the residual is still above the tolerance after it (new
// contacts stiffened the system since the budget was chosen, and the host has not looked yet), the launch goes on: its
// workgroups run up to `overflow` more iterations themselves, a grid barrier where the captured path has a kernel boundary
// (k_cg_ap's rows, barrier, these rows, barrier), the contact rows summed lane by lane.  An iteration costs about what a
// captured one does, so neither pies_tick nor a blind queue of pies_tick_async calls feeds an unconverged solve into the
// next substep, and the host raises the captured budget at its next look.  (The first version let the last workgroup to
// finish go on alone: 4 ms per iteration at 125k rows, a second per frame at a contact onset.)
__global__ void __launch_bounds__(kBlock) k_cg_update(CgArrays A, float4* __restrict__ x, int k, float tol2, int overflow) {
  // k_cg_ap(k) has looked at the residual of iteration k: had the solve converged, it would have set the flag and produced
  // nothing.  (Until the flag existed this kernel re-reduced the residual partials to take the same decision: one block-wide
  // reduction per launch for nothing.)
  if (A.scal[10] != 0.0f) return;
  float red[9];
  float rr[3], bb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) bb[c] = A.scal[6 + c];  // (written by k_cg_ap(0) of this solve)
  const bool rows = A.useCAp && A.tIncCnt;
  float pap[3];
  block_reduce_partials<3>(A.partA, 3, A.nparts + (rows ? kCgRowBlocks : 0u), pap);
  float alpha[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float rzk = A.scal[3 * (k & 1) + c];
    alpha[c] = pap[c] > 0.0f ? rzk / pap[c] : 0.0f;
  }
  float acc[6] = {0, 0, 0, 0, 0, 0};
  // blocks run concurrently, so the new residual partials go to the other half of a ping-pong pair
  cg_update_rows(A, x, k, alpha, rows, acc);
  block_write_partial<6>(acc, A.partBnext, 6);
  if (overflow <= 0 || A.lanesPerRow != 1u) return;
  // ---- the iterations beyond the captured ones --------------------------------------------------------------------
  float* const pb[2] = {A.partB0, A.partB1};
  uint32_t passed = 0;
  int kk = k + 1;
  float rzOld[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) rzOld[c] = A.scal[3 * (k & 1) + c];  // (kept in registers: block 0 does not have to publish them)
  for (;;) {
    if (!grid_barrier(A.ticket, gridDim.x, passed)) return;  // the partials of iteration kk - 1 are complete
    float rz[3];
    block_reduce_partials<6>(pb[kk & 1], 6, A.nparts, red);
#pragma unroll
    for (int c = 0; c < 3; ++c) { rz[c] = red[c]; rr[c] = red[3 + c]; }
    if (all_converged(rr, bb, tol2) || kk >= k + 1 + overflow) break;
    float beta[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) beta[c] = rzOld[c] > 0.0f ? rz[c] / rzOld[c] : 0.0f;
    float a3[3] = {0, 0, 0};
    cg_ap_rows<1>(A, kk, beta, true, a3);
    block_write_partial<3>(a3, A.partA, 3);
    if (!grid_barrier(A.ticket, gridDim.x, passed)) return;
    block_reduce_partials<3>(A.partA, 3, A.nparts, pap);
#pragma unroll
    for (int c = 0; c < 3; ++c) alpha[c] = pap[c] > 0.0f ? rz[c] / pap[c] : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = 0.0f;
    cg_update_rows(A, x, kk, alpha, false, acc);
    block_write_partial<6>(acc, pb[(kk + 1) & 1], 6);
#pragma unroll
    for (int c = 0; c < 3; ++c) rzOld[c] = rz[c];
    ++kk;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // where the final residual partials are, and how many iterations it took
    A.scal[9] = static_cast<float>(kk);
    A.scal[11] = static_cast<float>(1 + (kk & 1));
    A.scal[10] = 1.0f;
  }
}

// end of the last solve of a substep: its statistics
__global__ void __launch_bounds__(kBlock) k_cg_finish(CgArrays A) { solve_statistics(A, A.partB); }
void launch_cg_finish(hipStream_t st, const CgArrays& A) { hipLaunchKernelGGL(k_cg_finish, dim3(1), dim3(kBlock), 0, st, A); }

uint32_t cg_update_resident_blocks(int device) {
  int perCu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_cg_update, kBlock, 0) != hipSuccess) return 0;
  return static_cast<uint32_t>(std::max(0, perCu)) * static_cast<uint32_t>(std::max(0, prop.multiProcessorCount));
}

void launch_pd_solve(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, int maxIters, float tol, int part, bool first, bool last,
                     bool neverExit, void (*hook)(void*, int), void* hookCtx, int overflowIters) {
  if (nd.n == 0) return;
  CgArrays A = pd.cg;
  const dim3 grid(A.nparts), block(kBlock);
  const bool rows = A.useCAp && A.tIncCnt;
  const dim3 agrid(A.nparts + (rows ? kCgRowBlocks : 0u));  // k_cg_ap: SpMV blocks + contact-row blocks
  if (part >= 0) {  // profile pass: one kind of kernel only, never taking the converged early exit
    (void)hipMemsetAsync(A.scal + 10, 0, sizeof(float), st);  // (the last real solve may have left "converged" behind)
    for (int k = 0; k < maxIters; ++k) {
      if (part == 1) {
        if (A.lanesPerRow == 4) hipLaunchKernelGGL(k_cg_ap<4>, agrid, block, 0, st, A, k, -1.0f);
        else if (A.lanesPerRow == 2) hipLaunchKernelGGL(k_cg_ap<2>, agrid, block, 0, st, A, k, -1.0f);
        else if (A.lanesPerRow == 8) hipLaunchKernelGGL(k_cg_ap<8>, agrid, block, 0, st, A, k, -1.0f);
        else hipLaunchKernelGGL(k_cg_ap<1>, agrid, block, 0, st, A, k, -1.0f);
      }
      else hipLaunchKernelGGL(k_cg_update, grid, block, 0, st, A, nd.pos, k, -1.0f, 0);
    }
    return;
  }
  const float tol2 = neverExit ? -1.0f : tol * tol;
  A.tol2 = tol * tol;
  float* pb[2] = {pd.cg.partB, pd.cg.partBnext};
  // every solve of a substep runs the same number of iterations, so the previous solve left its final partials here
  if (rows) hipLaunchKernelGGL(k_contact_rows, dim3(256), block, 0, st, A, nd.pos);
  const float* prevB = first ? nullptr : pb[maxIters & 1];
  const dim3 igrid(A.nparts + 1u);  // k_cg_init: SpMV blocks + the bookkeeping block
  if (A.lanesPerRow == 4) hipLaunchKernelGGL(k_cg_init<4>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  else if (A.lanesPerRow == 2) hipLaunchKernelGGL(k_cg_init<2>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  else if (A.lanesPerRow == 8) hipLaunchKernelGGL(k_cg_init<8>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  else hipLaunchKernelGGL(k_cg_init<1>, igrid, block, 0, st, A, nd.pos, pd.rhs, prevB);
  for (int k = 0; k < maxIters; ++k) {
    A.partB = pb[k & 1];       // residual partials of iteration k (k = 0 reads partI instead)
    A.partBnext = pb[(k + 1) & 1];
    if (hook) hook(hookCtx, 14);  // PIES_KERNEL_PD_SPMV
    if (A.lanesPerRow == 4) hipLaunchKernelGGL(k_cg_ap<4>, agrid, block, 0, st, A, k, tol2);
    else if (A.lanesPerRow == 2) hipLaunchKernelGGL(k_cg_ap<2>, agrid, block, 0, st, A, k, tol2);
    else if (A.lanesPerRow == 8) hipLaunchKernelGGL(k_cg_ap<8>, agrid, block, 0, st, A, k, tol2);
    else hipLaunchKernelGGL(k_cg_ap<1>, agrid, block, 0, st, A, k, tol2);
    if (hook) { hook(hookCtx, 14); hook(hookCtx, 15); }  // PIES_KERNEL_PD_CG_UPDATE
    hipLaunchKernelGGL(k_cg_update, grid, block, 0, st, A, nd.pos, k, tol2, k + 1 == maxIters && !neverExit ? overflowIters : 0);
    if (hook) hook(hookCtx, 15);
  }
  if (!last) return;
  A.partB = pb[maxIters & 1];
  hipLaunchKernelGGL(k_cg_finish, dim3(1), block, 0, st, A);
}

}  // namespace pies
