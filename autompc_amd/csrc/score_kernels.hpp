// Trajectory scoring on the device: Cost.__call__ (autompc/costs/cost.py:27-41) for a batch of
// finished closed-loop trajectories,
//     score = sum_t [ eval_obs_cost(obs_t) + eval_ctrl_cost(ctrl_t) ] + eval_term_obs_cost(obs_last)
// where the task cost is a sum of terms (SumCost._sum_results, sum_cost.py:49-54):
//   kind 0  quadratic   (x-g)'Q(x-g), u'Ru, terminal (x-g)'F(x-g)       (quad_cost.py:7-51)
//   kind 1  threshold   1 if max_{lo<=i<hi} |x_i - g_i| > threshold      (thresh_cost.py:27-32)
//   kind 2  box         1 if any x_i < lo_i or x_i > hi_i                (thresh_cost.py:73-77)
// Threshold and box terms have no control or terminal part (thresh_cost.py:34-38, 79-83).
// HBM-bound and tiny (B*T1*(nx+nu) values read once); one workgroup per trajectory, rows strided
// over threads, one block reduction.
#pragma once
#include <hip/hip_runtime.h>

#include "mppi_kernels.hpp"

namespace ampc {

enum ScoreKind : int { SCORE_QUAD = 0, SCORE_THRESHOLD = 1, SCORE_BOX = 2 };

// Parameter block sizes (in values) of one term; layouts:
//   quad       Q[no*no] R[nu*nu] F[no*no] goal[no]
//   threshold  goal[no] lo hi threshold
//   box        lo[no] hi[no]
__host__ __device__ inline int score_term_size(int kind, int no, int nu) {
  switch (kind) {
    case SCORE_QUAD: return 2 * no * no + nu * nu + no;
    case SCORE_THRESHOLD: return no + 3;
    case SCORE_BOX: return 2 * no;
    default: return -1;
  }
}

template <typename T>
__device__ inline T score_quad_form(const T* __restrict__ M, const T* __restrict__ v,
                                    const T* __restrict__ g, int n) {
  T acc = T(0);
  for (int i = 0; i < n; ++i) {
    T s = T(0);
    for (int j = 0; j < n; ++j) s += M[i * n + j] * (v[j] - (g ? g[j] : T(0)));
    acc += (v[i] - (g ? g[i] : T(0))) * s;
  }
  return acc;
}

template <typename T>
__device__ inline T score_stage(int kind, const T* __restrict__ par, const T* __restrict__ x,
                                const T* __restrict__ u, int no, int nu) {
  if (kind == SCORE_QUAD) {
    const T* Q = par;
    const T* R = par + no * no;
    const T* g = par + 2 * no * no + nu * nu;
    return score_quad_form(Q, x, g, no) + score_quad_form(R, u, (const T*)nullptr, nu);
  }
  if (kind == SCORE_THRESHOLD) {
    const int lo = (int)par[no], hi = (int)par[no + 1];
    const T thr = par[no + 2];
    bool out = false, nan_in = false;       // (norm(diff, inf) is NaN as soon as one in-range entry is: not charged)
    for (int i = lo; i < hi; ++i) {
      const T d = x[i] - par[i];
      out = out || (d > thr) || (-d > thr);
      nan_in = nan_in || d != d;
    }
    return (out && !nan_in) ? T(1) : T(0);
  }
  bool out = false;
  for (int i = 0; i < no; ++i) out = out || (x[i] < par[i]) || (x[i] > par[no + i]);
  return out ? T(1) : T(0);
}

// obs [B][T1][nx] (the first `no` entries of a row are the observation), ctrls [B][T1][nu],
// scores [B].  kinds/offs [n_terms]: term kind and offset of its parameter block in `par`.
template <typename T>
__global__ __launch_bounds__(kWG) void score_trajectories_kernel(
    const T* __restrict__ obs, const T* __restrict__ ctrls, int T1, int nx, int nu, int no,
    int n_terms, const int* __restrict__ kinds, const int* __restrict__ offs,
    const T* __restrict__ par, T* __restrict__ scores) {
  __shared__ T scratch[kWaves];
  const int b = blockIdx.x;
  T acc = T(0);
  for (int t = threadIdx.x; t < T1; t += kWG) {
    const T* x = obs + ((size_t)b * T1 + t) * nx;
    const T* u = ctrls + ((size_t)b * T1 + t) * nu;
    for (int k = 0; k < n_terms; ++k) acc += score_stage(kinds[k], par + offs[k], x, u, no, nu);
  }
  acc = block_sum(acc, scratch);
  if (threadIdx.x == 0) {
    const T* x = obs + ((size_t)b * T1 + (T1 - 1)) * nx;
    for (int k = 0; k < n_terms; ++k)
      if (kinds[k] == SCORE_QUAD) {
        const T* p = par + offs[k];
        acc += score_quad_form(p + no * no + nu * nu, x, p + 2 * no * no + nu * nu, no);
      }
    scores[b] = acc;
  }
}

}  // namespace ampc
