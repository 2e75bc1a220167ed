// Jacobi-preconditioned CG of the PD global step (Solver.cpp:356 solves with a sparse Cholesky), ONE launch per
// iteration.
//
// The two-launch form (pd_cg_kernels.hip) pays two kernel boundaries per iteration because standard CG has two global
// reductions with a vector update between them: alpha = r.z / p.Ap, then beta = r'.z' / r.z.  At 100k rows a boundary costs as
// much as the rows (4.5 us per launch, profiles/r03_moving_config3.txt: 350 of a 780-us substep), so the iteration is
// rearranged (Chronopoulos & Gear 1989): with u = D^-1 r, w = (K + C) u,
//     gamma_i = r_i.u_i, delta_i = w_i.u_i              (one reduction)
//     beta_i = gamma_i / gamma_{i-1}, alpha_i = gamma_i / (delta_i - beta_i gamma_i / alpha_{i-1})
//     p_i = u_i + beta_i p_{i-1};  s_i = w_i + beta_i s_{i-1}   (s = (K + C) p by recurrence)
//     x_{i+1} = x_i + alpha_i p_i;  r_{i+1} = r_i - alpha_i s_i;  u_{i+1} = D^-1 r_{i+1};  w_{i+1} = (K + C) u_{i+1}
// which has ONE synchronisation point per iteration.  The remaining dependency - a row of w_{i+1} needs u_{i+1} at its
// neighbours - is removed by recomputing the neighbour's update while it is gathered: u_{i+1,j} depends on j's own old vectors
// and the two scalars only.  The kernel keeps the PRECONDITIONED vectors t = D^-1 r, c = D^-1 s, a = D^-1 w (12-byte records),
// so that u_{i+1,j} = t_j - alpha (a_j + beta c_j) is three gathers and two fused multiply-adds, and row j itself stores
// exactly that value as its new t: what a neighbour used and what the row keeps are the same bits.  New t / c / a go to the
// other half of a ping-pong pair (other workgroups are still gathering the old ones).
//
// Launches of a solve: k_cg1_init (r = b - (K + C) x, with the right-hand side evaluated in place when it is only a few tile
// sums per node), k_cg1_first (w_0), then one k_cg1_iter per iteration; a launch that finds the residual of the previous
// one below the tolerance sets the "done" word and every later launch of the solve returns on it.  Dot products are
// per-workgroup partial sums that every workgroup of the next launch re-reduces in a fixed order (deterministic, no atomics).
// The last captured launch goes on by itself behind a grid barrier when the solve needs more iterations than were captured
// (see grid_barrier).  Contact rows are summed inline by the row's lane: this is the contact-light graph variant; the
// contact-heavy one keeps the two-launch form.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>

#include "pd_cg_device.h"
#include "pd_rhs_device.h"


namespace pies {

// In-kernel time stamps of a diagnostic build (python -m pies_amd.build --exp; tools/cg_timeline.py): lane 0 of every workgroup of
// k_cg1_iter notes s_memtime at the points marked CG_STAMP.  Never part of the product build: there CgStamp is empty.
#ifdef PIES_EXPERIMENTS
__device__ unsigned long long* g_cg_stamps = nullptr;  // [iteration & 7][workgroup < 4096][32]
struct CgStamp {
  unsigned long long* base = nullptr;
  int idx = 0;
  PIES_DEV void open(int it) {
    if (g_cg_stamps && threadIdx.x == 0 && blockIdx.x < 4096u) {
      base = g_cg_stamps + (static_cast<size_t>(it & 7) * 4096u + blockIdx.x) * 32u;
      base[idx++] = __builtin_amdgcn_s_memrealtime();
    }
  }
  PIES_DEV void mark() { if (base && idx < 31) base[idx++] = __builtin_amdgcn_s_memtime(); }
  PIES_DEV void close() { if (base) base[31] = __builtin_amdgcn_s_memrealtime(); }
};
extern "C" int pies_exp_cg_stamps(unsigned long long* deviceBuffer) {  // 8 x 4096 x 32 x 8 bytes, or nullptr
  return hipMemcpyToSymbol(HIP_SYMBOL(g_cg_stamps), &deviceBuffer, sizeof(deviceBuffer)) == hipSuccess ? 0 : 1;
}
#else
struct CgStamp {
  PIES_DEV void open(int) {}
  PIES_DEV void mark() {}
  PIES_DEV void close() {}
};
#endif

// f(value, column) for every stored entry of row i (lane `lane` of slice `sl`): the row dictionary or the SELL arrays
template <class F> PIES_DEV void row_entries(const CgArrays& A, uint32_t sl, uint32_t lane, uint32_t i, F f) {
  if (A.rowStencil) {
    if (i < A.n) {
      const uint32_t rs = A.rowStencil[i];
      const uint32_t b = rs & 0xffffffu, e = b + (rs >> 24);
#pragma unroll 4
      for (uint32_t q = b; q < e; ++q) {
        const int2 pr = A.stencil[q];
        f(__int_as_float(pr.y), static_cast<uint32_t>(static_cast<int>(i) + pr.x));
      }
    }
  } else {
    const uint32_t off = A.sliceOff[sl], width = (A.sliceOff[sl + 1] - off) >> 6;
#pragma unroll 4
    for (uint32_t kk = 0; kk < width; ++kk) {
      const uint32_t at = off + (kk << 6) + lane;
      f(A.val[at], A.col[at]);  // (padding entries: the row itself with value 0)
    }
  }
}

// ---- windowed rows (CgArrays: wRows ...) -----------------------------------------------------------------------------------
// The chunks a workgroup sweeps: workgroups that share an XCD (equal blockIdx % 8) take one contiguous part of the chunks, like
// slice_sweep's wavefronts.
struct ChunkSweep {
  uint32_t begin, end, step;
};
PIES_DEV ChunkSweep chunk_sweep(uint32_t nchunks, uint32_t nblocks) {
  const uint32_t labels = nblocks < 8u ? nblocks : 8u;
  const uint32_t x = blockIdx.x % labels, xb = blockIdx.x / labels;
  const uint32_t nbx = (nblocks - x + labels - 1u) / labels;
  const uint32_t segBeg = static_cast<uint32_t>((static_cast<uint64_t>(nchunks) * x) / labels);
  const uint32_t segEnd = static_cast<uint32_t>((static_cast<uint64_t>(nchunks) * (x + 1u)) / labels);
  return {segBeg + xb, segEnd, nbx};
}

// One sweep over the matrix in its windowed form, chunk by chunk:
//   own(i, aux)         the row's own streaming work, in the natural order (coalesced); returns what its window slot holds and
//                       may leave 16 bytes for the row's turn behind the sum in `aux` (AUX)
//   halo(j)             what the slot of column j outside the chunk holds
//   row(i, sx, sy, sz, u, aux)   what the row does with its sum (u = the row's own slot); the rows of a chunk may come in any
//                       order here (sorted by length)
// ACC: float or double sums.  All threads of the workgroup must call (barriers).
// (Measured and dropped: the first 8-16 entries of a wavefront's first slice requested before the window is filled - they do not
// depend on it.  The entries in registers across the fill cost 40-60 registers and scalar spills: 12.4 against 10.4 us per launch at
// 100k rows, 14.1 against 12.7 on the unstructured beam.)
template <class ACC, bool AUX, class Own, class Halo, class Row>
PIES_DEV void window_rows(const CgArrays& A, uint32_t nblocks, Own own, Halo halo, Row row, CgStamp* stamp = nullptr) {
  CgStamp none;
  CgStamp& st = stamp ? *stamp : none;
  extern __shared__ float4 pies_window[];
  float4* __restrict__ win = pies_window;
  float4* __restrict__ auxv = pies_window + A.wLdsSlots;  // (wRows slots behind the largest window)
  const ChunkSweep cs = chunk_sweep(A.wChunks, nblocks);
  const uint32_t R = A.wRows, spc = R >> 6;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (uint32_t c = cs.begin; c < cs.end; c += cs.step) {
    const uint32_t r0 = c * R;
    const uint2 ch = A.wChunk[c];
    for (uint32_t lr = threadIdx.x; lr < R; lr += kBlock) {
      const uint32_t i = r0 + lr;
      if (i < A.n) {
        float4 aux = make_float4(0.f, 0.f, 0.f, 0.f);
        win[lr] = own(i, aux);
        if (AUX) auxv[lr] = aux;
      }
    }
    st.mark();  // the own rows are done (their stores issued)
    // (Measured and dropped in round 6: the halo's column indices requested before the own rows' loads, without a branch, and the
    // gathers of eight columns per lane in flight together.  In the in-kernel stamps of tools/cg_timeline.py the halo phase falls from
    // 2.65 to 1.6 us on the unstructured beam (6.4 halo columns per row) and the own rows' rises from 1.6 to 1.85; a launch as a whole
    // - 391 workgroups on 256 compute units, the last one ends 3 us after the median - and the substep do not change.)
    if (A.wHalo16) {
      const uint32_t base = A.wBase[c];
      const uint16_t* __restrict__ h = A.wHalo16 + ch.x;
      for (uint32_t k = threadIdx.x; k < ch.y; k += kBlock)
        if (PIES_IN_BOUNDS(R + k < A.wLdsSlots && base + h[k] < A.n, 32u)) win[R + k] = halo(base + h[k]);
    } else if (A.wHalo32) {
      const uint32_t* __restrict__ h = A.wHalo32 + ch.x;
      for (uint32_t k = threadIdx.x; k < ch.y; k += kBlock)
        if (PIES_IN_BOUNDS(R + k < A.wLdsSlots && h[k] < A.n, 33u)) win[R + k] = halo(h[k]);
    }
    st.mark();  // the halo is requested and stored
    __syncthreads();
    st.mark();  // the window is complete
    for (uint32_t sc = wave; sc < spc; sc += kBlock / 64u) {
      const uint32_t sl = c * spc + sc;
      const uint32_t lr = A.wPerm ? A.wPerm[static_cast<size_t>(sl) * 64u + lane] : sc * 64u + lane;
      const uint32_t i = r0 + lr;
      const uint32_t off = A.wSliceOff[sl], width = (A.wSliceOff[sl + 1] - off) >> 6;
      const float* __restrict__ wv = A.wVal + off + lane;
      const uint16_t* __restrict__ wi = A.wIdx + off + lane;
      ACC sx = 0, sy = 0, sz = 0;
#pragma unroll 4  // (8 and 16 measure the same: round 6)
      for (uint32_t kk = 0; kk < width; ++kk) {
        const ACC a = static_cast<ACC>(wv[kk << 6]);
        if (!PIES_IN_BOUNDS(wi[kk << 6] < A.wLdsSlots, 31u)) continue;
        const float4 q = win[wi[kk << 6]];
        sx = fma(a, static_cast<ACC>(q.x), sx);
        sy = fma(a, static_cast<ACC>(q.y), sy);
        sz = fma(a, static_cast<ACC>(q.z), sz);
      }
#ifdef PIES_EXPERIMENTS
      asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));
      st.mark();  // the row's sum
#endif
      if (i < A.n) row(i, sx, sy, sz, win[lr], AUX ? auxv[lr] : make_float4(0.f, 0.f, 0.f, 0.f));
      st.mark();  // the row's turn behind the sum
    }
    __syncthreads();  // the next chunk's window overwrites this one
    st.mark();
  }
}

// r = f - (K + C) x ; t = D^-1 r ; partI = {r.t, r.r, f.f}.  RHS: f is evaluated here (Msn_h2 + the node's records + contacts +
// shape / goal terms + floor: rhs_of_node with one lane per node) instead of being read from the array k_pd_rhs wrote.
// prevPart != nullptr: an extra block closes the previous solve's statistics (see k_cg_init).
template <bool RHS, bool WIN>
__global__ void __launch_bounds__(kBlock) k_cg1_init(CgArrays A, const float4* __restrict__ x, const float4* __restrict__ f, RhsArrays R,
                                                     const float* __restrict__ prevPart) {
  if (blockIdx.x == A.npartsI) {
    if (prevPart) solve_statistics(A, prevPart);
    if (threadIdx.x == 0) {
      A.scal[10] = 0.0f;
      A.ticket[0] = 0u;
      A.ticket[1] = 0u;
    }
    return;
  }
  float acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  Vec3f* __restrict__ t0 = A.t1[0];
  // what a row does with (K x)_i (fp64 sums, see below), its own x and its right-hand side
  auto finish = [&](uint32_t i, double sx, double sy, double sz, const float4 xi, float4 fi) {
    double cx = 0.0, cy = 0.0, cz = 0.0;
    contact_row<4, double>(A, i, [&](uint32_t j, float& px, float& py, float& pz) { const float4 v = x[j]; px = v.x; py = v.y; pz = v.z; }, cx, cy, cz);
    if (!RHS) fi = f[i];
    const float cd = A.cdiag[i], di = A.dinv[i];
    const double cdd = static_cast<double>(cd);
    const float rx = static_cast<float>(static_cast<double>(fi.x) - (fma(cdd, static_cast<double>(xi.x), sx) + cx));
    const float ry = static_cast<float>(static_cast<double>(fi.y) - (fma(cdd, static_cast<double>(xi.y), sy) + cy));
    const float rz = static_cast<float>(static_cast<double>(fi.z) - (fma(cdd, static_cast<double>(xi.z), sz) + cz));
    const float tx = di * rx, ty = di * ry, tz = di * rz;
    t0[i] = Vec3f{tx, ty, tz};
    acc9[0] += rx * tx; acc9[1] += ry * ty; acc9[2] += rz * tz;
    acc9[3] += rx * rx; acc9[4] += ry * ry; acc9[5] += rz * rz;
    acc9[6] += fi.x * fi.x; acc9[7] += fi.y * fi.y; acc9[8] += fi.z * fi.z;
  };
  if (WIN) {
    // (the right-hand side's gather is issued with the window's loads, in the natural order, and waits in LDS for the row's turn)
    window_rows<double, RHS>(
        A, A.npartsI,
        [&](uint32_t i, float4& aux) {
          const float4 xi = x[i];
          if (RHS) aux = rhs_of_node<1>(R, i, 0u, true);
          return xi;
        },
        [&](uint32_t j) { return x[j]; },
        [&](uint32_t i, double sx, double sy, double sz, const float4 xi, const float4 fi) { finish(i, sx, sy, sz, xi, fi); });
    block_write_partial<9>(acc9, A.partI, 9);
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const SliceSweep sw = slice_sweep<1>(A.n, A.npartsI);
  for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
    const uint32_t i = sl * 64u + lane;
    const bool live = i < A.n;
    float4 fi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RHS) fi = rhs_of_node<1>(R, i, 0u, live);  // (issued first: its gathers are in flight beside the row's)
    // (K + C) x in DOUBLE: the products are ~ (m / h^2) |x| ~ 1e6 and cancel against f to a residual five orders smaller; summed
    // in fp32 the row carries a noise of an ulp of 1e6 (0.1: 2e-5 in x per solve, a random walk over the local/global
    // iterations - measured against the oracle's fp64 solve: 2.3 x the deviation of the reference's own fp32 arithmetic, in
    // every variant of the local step and of the CG).  fp64 fused multiply-adds cost this memory-bound launch nothing.
    double sx = 0.0, sy = 0.0, sz = 0.0;
    row_entries(A, sl, lane, i, [&](float a, uint32_t j) {
      const float4 xj = x[j];
      const double ad = static_cast<double>(a);
      sx = fma(ad, static_cast<double>(xj.x), sx);
      sy = fma(ad, static_cast<double>(xj.y), sy);
      sz = fma(ad, static_cast<double>(xj.z), sz);
    });
    if (live) finish(i, sx, sy, sz, x[i], fi);
  }
  block_write_partial<9>(acc9, A.partI, 9);
}

// a_0 = D^-1 w_0, w_0 = (K + C) t_0 ; part1[0] = {., w_0.t_0, .}.  Also the solve's first look at the residual.
template <bool WIN> __global__ void __launch_bounds__(kBlock) k_cg1_first(CgArrays A, float tol2) {
  float red[9];
  block_reduce_partials<9>(A.partI, 9, A.npartsI, red);
  const float rr[3] = {red[3], red[4], red[5]}, bb[3] = {red[6], red[7], red[8]};
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A.scal[c] = red[c];  // gamma_0
      A.scal[6 + c] = bb[c];
    }
    A.scal[9] = 0.0f;  // iterations of this solve
  }
  if (all_converged(rr, bb, tol2)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      // the final residual norms themselves, for the statistics: partI is overwritten by the next solve's k_cg1_init while the
      // extra workgroup of that very launch closes this solve's statistics (a mix of two solves could read as "short")
#pragma unroll
      for (int c = 0; c < 3; ++c) A.scal[18 + c] = rr[c];
      A.scal[11] = 0.0f;  // converged at the first look
      A.scal[10] = 1.0f;
    }
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const Vec3f* __restrict__ t0 = A.t1[0];
  Vec3f* __restrict__ a0 = A.a1[0];
  float acc[3] = {0, 0, 0};
  auto finish = [&](uint32_t i, float sx, float sy, float sz, float tix, float tiy, float tiz) {
    contact_row(A, i, [&](uint32_t j, float& px, float& py, float& pz) { const Vec3f v = t0[j]; px = v.x; py = v.y; pz = v.z; }, sx, sy, sz);
    const float cd = A.cdiag[i], di = A.dinv[i];
    const float wx = fmaf(cd, tix, sx), wy = fmaf(cd, tiy, sy), wz = fmaf(cd, tiz, sz);
    a0[i] = Vec3f{di * wx, di * wy, di * wz};
    acc[0] += wx * tix; acc[1] += wy * tiy; acc[2] += wz * tiz;
  };
  if (WIN) {
    auto slot = [&](uint32_t j) { const Vec3f v = t0[j]; return make_float4(v.x, v.y, v.z, 0.f); };
    window_rows<float, false>(
        A, gridDim.x, [&](uint32_t i, float4&) { return slot(i); }, slot,
        [&](uint32_t i, float sx, float sy, float sz, const float4 ti, const float4) { finish(i, sx, sy, sz, ti.x, ti.y, ti.z); });
  } else {
    const SliceSweep sw = slice_sweep<1>(A.n, gridDim.x);
    for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
      const uint32_t i = sl * 64u + lane;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      row_entries(A, sl, lane, i, [&](float a, uint32_t j) {
        const Vec3f tj = t0[j];
        sx = fmaf(a, tj.x, sx);
        sy = fmaf(a, tj.y, sy);
        sz = fmaf(a, tj.z, sz);
      });
      if (i < A.n) {
        const Vec3f ti = t0[i];
        finish(i, sx, sy, sz, ti.x, ti.y, ti.z);
      }
    }
  }
  block_write_partial<3>(acc, A.part1[0] + 3, 9);
  if (blockIdx.x == 0 && threadIdx.x == 0) A.partCount[2] = gridDim.x;
}

// The rows of iteration `it` (the it-th update of x, it >= 1) for this workgroup's slices, given alpha_{it-1} and beta_{it-1}:
// reads the vectors of parity (it - 1) & 1, writes those of parity it & 1; acc += {r.t, w.t, r.r} of the new vectors.
// FIRST (it == 1): beta = 0 and there is no c / p yet.
template <bool FIRST, bool WIN>
PIES_DEV void cg1_rows(const CgArrays& A, float4* __restrict__ x, int it, const float alpha[3], const float beta[3], float acc[9], CgStamp* stamp = nullptr) {
  const Vec3f* __restrict__ tO = A.t1[(it - 1) & 1];
  const Vec3f* __restrict__ aO = A.a1[(it - 1) & 1];
  const Vec3f* __restrict__ cO = A.c1[(it - 1) & 1];
  Vec3f* __restrict__ tN = A.t1[it & 1];
  Vec3f* __restrict__ aN = A.a1[it & 1];
  Vec3f* __restrict__ cN = A.c1[it & 1];
  Vec3f* __restrict__ p = A.p1;
  // the new preconditioned residual of row j, and (own row) the new c: the SAME two fused multiply-adds wherever it is evaluated
  auto renew = [&](uint32_t j, float& ux, float& uy, float& uz, float& cx, float& cy, float& cz) {
    const Vec3f tj = tO[j], aj = aO[j];
    cx = aj.x; cy = aj.y; cz = aj.z;
    if (!FIRST) {
      const Vec3f cj = cO[j];
      cx = fmaf(beta[0], cj.x, cx); cy = fmaf(beta[1], cj.y, cy); cz = fmaf(beta[2], cj.z, cz);
    }
    ux = fmaf(-alpha[0], cx, tj.x); uy = fmaf(-alpha[1], cy, tj.y); uz = fmaf(-alpha[2], cz, tj.z);
  };
  auto fetch = [&](uint32_t j, float& ux, float& uy, float& uz) {
    float cx, cy, cz;
    renew(j, ux, uy, uz, cx, cy, cz);
  };
  if (WIN) {
    // The row's own updates - everything that does not wait for the row's sum - happen while its window slot is filled, in the
    // natural order: x, p, the new t and c, r.t and r.r.  Behind the sum only w = (K + C) t, a = D^-1 w and w.t are left.
    window_rows<float, false>(
        A, gridDim.x,
        [&](uint32_t i, float4&) {
          float tx, ty, tz, cx, cy, cz;
          renew(i, tx, ty, tz, cx, cy, cz);
          const Vec3f told = tO[i];
          float px = told.x, py = told.y, pz = told.z;
          if (!FIRST) {
            const Vec3f po = p[i];
            px = fmaf(beta[0], po.x, px); py = fmaf(beta[1], po.y, py); pz = fmaf(beta[2], po.z, pz);
          }
          float4 xi = x[i];
          xi.x = fmaf(alpha[0], px, xi.x);
          xi.y = fmaf(alpha[1], py, xi.y);
          xi.z = fmaf(alpha[2], pz, xi.z);
          const float kd = A.kdiag[i] + A.cdiag[i];
          const float rx = kd * tx, ry = kd * ty, rz = kd * tz;
          x[i] = xi;
          p[i] = Vec3f{px, py, pz};
          tN[i] = Vec3f{tx, ty, tz};
          cN[i] = Vec3f{cx, cy, cz};
          acc[0] += rx * tx; acc[1] += ry * ty; acc[2] += rz * tz;
          acc[6] += rx * rx; acc[7] += ry * ry; acc[8] += rz * rz;
          return make_float4(tx, ty, tz, 0.f);
        },
        [&](uint32_t j) {
          float ux, uy, uz;
          fetch(j, ux, uy, uz);
          return make_float4(ux, uy, uz, 0.f);
        },
        [&](uint32_t i, float sx, float sy, float sz, const float4 t, const float4) {
          contact_row<1>(A, i, fetch, sx, sy, sz);
          const float cd = A.cdiag[i], di = A.dinv[i];
          const float wx = fmaf(cd, t.x, sx), wy = fmaf(cd, t.y, sy), wz = fmaf(cd, t.z, sz);
          aN[i] = Vec3f{di * wx, di * wy, di * wz};
          acc[3] += wx * t.x; acc[4] += wy * t.y; acc[5] += wz * t.z;
        }, stamp);
    return;
  }
  const uint32_t lane = threadIdx.x & 63u;
  const SliceSweep sw = slice_sweep<1>(A.n, gridDim.x);
  for (uint32_t sl = sw.begin; sl < sw.end; sl += sw.step) {
    const uint32_t i = sl * 64u + lane;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    row_entries(A, sl, lane, i, [&](float a, uint32_t j) {
      float ux, uy, uz;
      fetch(j, ux, uy, uz);
      sx = fmaf(a, ux, sx);
      sy = fmaf(a, uy, sy);
      sz = fmaf(a, uz, sz);
    });
    if (i < A.n) {
      contact_row<1>(A, i, fetch, sx, sy, sz);
      float tx, ty, tz, cx, cy, cz;
      renew(i, tx, ty, tz, cx, cy, cz);
      const Vec3f told = tO[i];
      float px = told.x, py = told.y, pz = told.z;  // p = u + beta p_old, u = the OLD t
      if (!FIRST) {
        const Vec3f po = p[i];
        px = fmaf(beta[0], po.x, px); py = fmaf(beta[1], po.y, py); pz = fmaf(beta[2], po.z, pz);
      }
      float4 xi = x[i];
      xi.x = fmaf(alpha[0], px, xi.x);
      xi.y = fmaf(alpha[1], py, xi.y);
      xi.z = fmaf(alpha[2], pz, xi.z);
      const float cd = A.cdiag[i], di = A.dinv[i], kd = A.kdiag[i] + cd;
      const float wx = fmaf(cd, tx, sx), wy = fmaf(cd, ty, sy), wz = fmaf(cd, tz, sz);
      const float rx = kd * tx, ry = kd * ty, rz = kd * tz;
      x[i] = xi;
      p[i] = Vec3f{px, py, pz};
      tN[i] = Vec3f{tx, ty, tz};
      cN[i] = Vec3f{cx, cy, cz};
      aN[i] = Vec3f{di * wx, di * wy, di * wz};
      acc[0] += rx * tx; acc[1] += ry * ty; acc[2] += rz * tz;
      acc[3] += wx * tx; acc[4] += wy * ty; acc[5] += wz * tz;
      acc[6] += rx * rx; acc[7] += ry * ry; acc[8] += rz * rz;
    }
  }
}

// three wave-uniform values moved to scalar registers
PIES_DEV void uniform3(float v[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v[c])));
}

// alpha_i and beta_i from {gamma_i, delta_i} and the previous iteration's {gamma_{i-1}, alpha_{i-1}}; a column that has nothing
// left to do (gamma = 0) or whose recurrence breaks down (a non-positive denominator) stands still
PIES_DEV void cg1_scalars(const float gam[3], const float del[3], const float gamOld[3], const float alphaOld[3], float alpha[3], float beta[3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    beta[c] = gamOld[c] > 0.0f ? gam[c] / gamOld[c] : 0.0f;
    const float den = alphaOld[c] > 0.0f ? del[c] - (beta[c] * gam[c]) / alphaOld[c] : del[c];
    alpha[c] = den > 0.0f ? gam[c] / den : 0.0f;
  }
}

// Launch `it` of a solve's iterations (it >= 1): scalars from the partial sums of the launch before, then the rows.
// overflow > 0: the solve's last captured launch; when the residual is still above the tolerance after its rows, its
// workgroups (all resident, see cg_update_resident_blocks) run up to `overflow` more iterations themselves, one grid barrier
// per iteration where the captured path has a kernel boundary.
// (Measured and dropped: register targets for this kernel.  __launch_bounds__(kBlock, 5) - five wavefronts per SIMD, the rows fit 85-92
// registers without spills: 41.1 against 39.5 us per launch with the row dictionary at 1M rows; amdgpu_waves_per_eu(4, 4): 43.9
// against 39.6 with the dictionary, 69.3 against 57.1 windowed behind a SELL first product.  What the windowed iteration's time
// depends on at 1M rows is whether the launch before it streamed the same matrix arrays - see launch_pd_solve1.)
template <bool FIRST, bool WIN>
__global__ void __launch_bounds__(kBlock) k_cg1_iter(CgArrays A, float4* __restrict__ x, int it, float tol2, int overflow) {
  if (A.scal[10] != 0.0f) return;  // the solve converged in an earlier launch
  CgStamp st;
  st.open(it);
  st.mark();
  // (Measured and dropped in round 6: the flag, the scalars, the count and the first two partial sums of every lane requested
  // together - one round trip instead of three.  The in-kernel stamps of tools/cg_timeline.py show the prologue at 2.2 instead of
  // 3.0 us and an isolated launch is 0.8 us shorter, but config 3 as a whole runs 1 % slower, 2 197 against 2 225 substeps/s in
  // three A/B pairs: two of a solve's three captured launches are early exits, which then carry the requests for nothing.)
  float red[9];
  float alpha[3], beta[3] = {0.f, 0.f, 0.f}, gam[3], bb[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) bb[c] = A.scal[6 + c];
  if (FIRST) {
    block_reduce_partials<3>(A.part1[0] + 3, 9, A.partCount[2], red + 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gam[c] = A.scal[c];
      alpha[c] = red[3 + c] > 0.0f ? gam[c] / red[3 + c] : 0.0f;
    }
  } else {
    block_reduce_partials<9>(A.part1[(it - 1) & 1], 9, A.partCount[(it - 1) & 1], red);
    const float rr[3] = {red[6], red[7], red[8]};
    if (all_converged(rr, bb, tol2)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        A.scal[11] = static_cast<float>(1 + ((it - 1) & 1));  // where the final residual partials are
        A.scal[10] = 1.0f;
      }
      return;
    }
    float gamOld[3], alphaOld[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      gam[c] = red[c];
      gamOld[c] = A.scal[3 * (it & 1) + c];         // gamma_{it-2}
      alphaOld[c] = A.scal[12 + 3 * (it & 1) + c];  // alpha_{it-2}
    }
    cg1_scalars(gam, red + 3, gamOld, alphaOld, alpha, beta);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      A.scal[3 * ((it - 1) & 1) + c] = gam[c];
      A.scal[12 + 3 * ((it - 1) & 1) + c] = alpha[c];
    }
    A.scal[9] = static_cast<float>(it);
  }
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  uniform3(alpha);  // (the same in every lane: scalar registers for the rows' sweep)
  uniform3(beta);
  st.mark();  // the scalars of this iteration
  cg1_rows<FIRST, WIN>(A, x, it, alpha, beta, acc, &st);
  st.mark();
  block_write_partial<9>(acc, A.part1[it & 1], 9);
  if (blockIdx.x == 0 && threadIdx.x == 0) A.partCount[it & 1] = gridDim.x;
  st.mark();
  st.close();
  if (overflow <= 0) return;
  // ---- the iterations beyond the captured ones ----------------------------------------------------------------------
  uint32_t passed = 0;
  int kk = it;
  float gamOld[3], alphaOld[3];
  for (;;) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { gamOld[c] = gam[c]; alphaOld[c] = alpha[c]; }
    if (!grid_barrier(A.ticket, gridDim.x, passed)) return;  // the partials of iteration kk are complete
    block_reduce_partials<9>(A.part1[kk & 1], 9, gridDim.x, red);
    const float rr[3] = {red[6], red[7], red[8]};
    if (all_converged(rr, bb, tol2) || kk >= it + overflow) break;
#pragma unroll
    for (int c = 0; c < 3; ++c) gam[c] = red[c];
    cg1_scalars(gam, red + 3, gamOld, alphaOld, alpha, beta);
    ++kk;
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = 0.0f;
    uniform3(alpha);
    uniform3(beta);
    cg1_rows<false, WIN>(A, x, kk, alpha, beta, acc);
    block_write_partial<9>(acc, A.part1[kk & 1], 9);
    if (blockIdx.x == 0 && threadIdx.x == 0) A.partCount[kk & 1] = gridDim.x;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    A.scal[9] = static_cast<float>(kk);
    A.scal[11] = static_cast<float>(1 + (kk & 1));
    A.scal[10] = 1.0f;
  }
}

// The windowed kernels take their window as dynamic LDS: above 64 KB a kernel has to be told - per DEVICE (the attribute belongs to
// the function on the device that is current when it is set), so the limit that has been raised is remembered per device, under a
// mutex (handles on different devices finalise from different threads).
static bool window_lds_ok(uint32_t bytes) {
  if (bytes <= 64u * 1024u) return true;
  static std::mutex mu;
  static std::map<int, uint32_t> allowed;  // device -> the limit its five windowed kernels have been raised to
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  const auto it = allowed.find(dev);
  if (it != allowed.end() && bytes <= it->second) return true;
  const void* fns[] = {reinterpret_cast<const void*>(&k_cg1_init<true, true>),  reinterpret_cast<const void*>(&k_cg1_init<false, true>),
                       reinterpret_cast<const void*>(&k_cg1_first<true>), reinterpret_cast<const void*>(&k_cg1_iter<true, true>),
                       reinterpret_cast<const void*>(&k_cg1_iter<false, true>)};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)) != hipSuccess) return false;
  allowed[dev] = bytes;
  return true;
}

PIES_BOUNDS_REPORT(cg1)
// workgroups of k_cg1_iter the device holds at once (its continuation's grid barrier needs all of a launch resident)
uint32_t cg1_iter_resident_blocks(int device, uint32_t windowLdsBytes) {
  int perCu = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
  if (windowLdsBytes) {
    if (!window_lds_ok(windowLdsBytes)) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_cg1_iter<false, true>, kBlock, windowLdsBytes) != hipSuccess) return 0;
  } else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_cg1_iter<false, false>, kBlock, 0) != hipSuccess) return 0;
  return static_cast<uint32_t>(std::max(0, perCu)) * static_cast<uint32_t>(std::max(0, prop.multiProcessorCount));
}

void launch_pd_solve1(hipStream_t st, const NodeArrays& nd, const PdArrays& pd, int iters, float tol, bool first, bool last, bool fuseRhs,
                      bool neverExit, void (*hook)(void*, int), void* hookCtx, int overflowIters) {
  if (nd.n == 0) return;
  CgArrays A = pd.cg;
  A.single = 1;
  A.tol2 = tol * tol;
  const float tol2 = neverExit ? -1.0f : tol * tol;
  // The launches of a solve are as wide as the rows ask for (npartsI) - except the last one: its continuation synchronises its
  // workgroups with a grid barrier and may only be as wide as the device holds at once (nparts).  Every launch notes its width with
  // its partial sums (partCount).
  const dim3 wide(A.npartsI), narrow(A.nparts), block(kBlock);
  // every solve of a substep captures the same number of launches, so the previous solve's last one left its partials here
  const float* prev = first ? nullptr : A.part1[iters & 1];
  const RhsArrays R = rhs_arrays(nd, pd);
  // Which launches of a solve take the windowed matrix where it was built (PIES_PD_WINDOW_KERNELS: 1 the iterations, 2 the first
  // product, 4 the residual; default 3).  The iterations gather three vectors per column and gain from the window by themselves
  // (unstructured 100k beam: 12.7 against 14.3 us per launch).  At 1M rows a second effect is larger: the launches of a solve that
  // stream the SAME arrays find them in the 256 MB Infinity Cache - the iteration behind a windowed first product takes 46.5 us,
  // behind one that streamed the SELL arrays 56-57 (its own matrix has to come from HBM then).  The residual kernel (fp64 sums,
  // the right-hand side's gather) is slower with the window everywhere measured (60.3 against 57.1 us at 1M, 13.0 against 10.8 at
  // 100k) and keeps the SELL rows.
  const uint32_t lds = A.wRows ? window_lds_bytes(A) : 0u;
  const bool win = A.wRows != 0 && window_lds_ok(lds);
  uint32_t which = 3u;
  if (const char* e = tuning_env("PIES_PD_WINDOW_KERNELS")) which = static_cast<uint32_t>(std::atoi(e)) & 7u;
  const bool winIter = win && (which & 1u), winFirst = win && (which & 2u), winAll = win && (which & 4u);
  if (fuseRhs) {
    if (hook) hook(hookCtx, 13);  // PIES_KERNEL_PD_RHS: the residual kernel that evaluates the right-hand side
    if (winAll) hipLaunchKernelGGL((k_cg1_init<true, true>), dim3(A.npartsI + 1u), block, lds, st, A, nd.pos, pd.rhs, R, prev);
    else hipLaunchKernelGGL((k_cg1_init<true, false>), dim3(A.npartsI + 1u), block, 0, st, A, nd.pos, pd.rhs, R, prev);
    if (hook) hook(hookCtx, 13);
  } else {
    if (winAll) hipLaunchKernelGGL((k_cg1_init<false, true>), dim3(A.npartsI + 1u), block, lds, st, A, nd.pos, pd.rhs, R, prev);
    else hipLaunchKernelGGL((k_cg1_init<false, false>), dim3(A.npartsI + 1u), block, 0, st, A, nd.pos, pd.rhs, R, prev);
  }
  if (winFirst) hipLaunchKernelGGL(k_cg1_first<true>, wide, block, lds, st, A, tol2);
  else hipLaunchKernelGGL(k_cg1_first<false>, wide, block, 0, st, A, tol2);
  for (int it = 1; it <= iters; ++it) {
    if (hook) hook(hookCtx, 14);  // PIES_KERNEL_PD_SPMV
    const int overflow = it == iters && !neverExit ? overflowIters : 0;
    const dim3 grid = overflow > 0 ? narrow : wide;
    if (winIter) {
      if (it == 1) hipLaunchKernelGGL((k_cg1_iter<true, true>), grid, block, lds, st, A, nd.pos, it, tol2, overflow);
      else hipLaunchKernelGGL((k_cg1_iter<false, true>), grid, block, lds, st, A, nd.pos, it, tol2, overflow);
    } else {
      if (it == 1) hipLaunchKernelGGL((k_cg1_iter<true, false>), grid, block, 0, st, A, nd.pos, it, tol2, overflow);
      else hipLaunchKernelGGL((k_cg1_iter<false, false>), grid, block, 0, st, A, nd.pos, it, tol2, overflow);
    }
    if (hook) hook(hookCtx, 14);
  }
  if (!last) return;
  A.partB = A.part1[iters & 1];
  launch_cg_finish(st, A);
}

}  // namespace pies
