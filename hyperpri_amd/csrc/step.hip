// The caller-side tail of one training / validation step: everything RootLightningModel does
// with the logits after the network returns them (PLTrainer.py:79-98, 100-140, 171-183, 530-562):
//   * nn.BCEWithLogitsLoss() mean (params_HyperPRI.py:60) forward + gradient,
//   * seg = sigmoid(pred) > threshold and the TP/FP/FN/TN counts torchmetrics' Accuracy / JaccardIndex / Dice reduce to,
//   * the binned PrecisionRecallCurve('binary', thresholds=500) histogram (PLTrainer.py:542-543),
//   * optim.Adam / optim.SGD over all parameter tensors in one launch (multi-tensor, PLTrainer.py:171-181).
// All of it is HBM-bound (1 channel of logits; 125 MB of parameters): one pass per tensor, fixed-order reductions
// (fp64 partials; integer atomics only), no host synchronisation (scalars stay on the device).
#include "common.h"

#define STEP_THREADS 256
#define STEP_MAX_BLOCKS 1024

// ------------------------------------------------------------------------------------------------
// BCE-with-logits, mean reduction:  l(x, y) = max(x, 0) - x*y + log1p(exp(-|x|))
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_elem(float x, float y) {
  return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

__device__ __forceinline__ float sigmoid_f32(float x) { return 1.f / (1.f + expf(-x)); }

// partial[b] = fp64 sum of this block's grid-stride slice (fixed slice -> fixed result)
__global__ __launch_bounds__(STEP_THREADS) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               long long n, double* __restrict__ partial) {
  __shared__ double red[STEP_THREADS];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * STEP_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * STEP_THREADS)
    s += (double)bce_elem(x[i], y[i]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = STEP_THREADS / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(STEP_THREADS) void bce_finalize_kernel(const double* __restrict__ partial, int nblk, long long n,
                                                                    float* __restrict__ loss) {
  __shared__ double red[STEP_THREADS];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += STEP_THREADS) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = STEP_THREADS / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

// dx = (sigmoid(x) - y) * g / n   (g: upstream scalar gradient on the device, nullptr = 1), evaluated as
// (1 - y) * sigmoid(x) - y * sigmoid(-x) so that a confident correct pixel keeps its relative accuracy
__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n,
                               const float* __restrict__ gout, float* __restrict__ dx) {
  const float g = (gout ? gout[0] : 1.f) / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
  {
    const float xi = x[i], yi = y[i];
    const float e = expf(-fabsf(xi)), r = 1.f / (1.f + e);
    const float sp = xi >= 0.f ? r : e * r, sn = xi >= 0.f ? e * r : r;     // sigmoid(x), sigmoid(-x)
    dx[i] = ((1.f - yi) * sp - yi * sn) * g;
  }
}

static inline int step_blocks(long long n) {
  long long b = (n + STEP_THREADS * 4 - 1) / (STEP_THREADS * 4);
  if (b > STEP_MAX_BLOCKS) b = STEP_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t hpri_bce_workspace_doubles(long long n) { return (size_t)step_blocks(n); }

extern "C" int hpri_bce_logits_fwd(const float* logits, const float* target, long long n, float* loss, double* workspace,
                                   size_t ws_doubles, hipStream_t stream) {
  HPRI_REQUIRE(logits && target && loss && workspace && n > 0, "bce_logits_fwd: bad arguments");
  const int nb = step_blocks(n);
  if ((size_t)nb > ws_doubles) return hpri_set_error(HPRI_ERR_WORKSPACE, "bce_logits_fwd: workspace too small");
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(nb), dim3(STEP_THREADS), 0, stream, logits, target, n, workspace);
  HPRI_CHECK_LAUNCH();
  hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(STEP_THREADS), 0, stream, workspace, nb, n, loss);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// loss = sum of `nblk` fp64 partial sums / n: the second half of hpri_bce_logits_fwd for partial sums another kernel produced
// (hpri_outconv_fwd_bce: the loss computed inside the network's last layer)
extern "C" int hpri_bce_finish(const double* partial, int nblk, long long n, float* loss, hipStream_t stream) {
  HPRI_REQUIRE(partial && loss && nblk > 0 && n > 0, "bce_finish: bad arguments");
  hipLaunchKernelGGL(bce_finalize_kernel, dim3(1), dim3(STEP_THREADS), 0, stream, partial, nblk, n, loss);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

extern "C" int hpri_bce_logits_bwd(const float* logits, const float* target, long long n, const float* grad_out,
                                   float* dlogits, hipStream_t stream) {
  HPRI_REQUIRE(logits && target && dlogits && n > 0, "bce_logits_bwd: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(step_blocks(n)), dim3(STEP_THREADS), 0, stream, logits, target, n, grad_out,
                     dlogits);
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ------------------------------------------------------------------------------------------------
// seg = sigmoid(logit) > thr (fp32, as the reference evaluates it); counts[0..3] += TP, FP, FN, TN.
// The target is the float mask the batch carries; positive = its int32 truncation is non-zero
// (PLTrainer.py:80 `batch['mask'].to(torch.int32)`).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(STEP_THREADS) void seg_counts_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  long long n, float thr, int is_logits,
                                                                  unsigned long long* __restrict__ counts) {
  __shared__ unsigned int red[4];
  if (threadIdx.x < 4) red[threadIdx.x] = 0;
  __syncthreads();
  unsigned int c[4] = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * STEP_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * STEP_THREADS) {
    const float p = is_logits ? sigmoid_f32(x[i]) : x[i];
    const int seg = p > thr, pos = ((int)y[i]) != 0;
    c[(seg ? 0 : 2) + (pos ? 0 : 1)] += 1;   // seg&pos -> 0 (TP); seg&!pos -> 1 (FP); !seg&pos -> 2 (FN); !seg&!pos -> 3 (TN)
  }
  for (int k = 0; k < 4; ++k) {
    unsigned int v = c[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&red[k], v);
  }
  __syncthreads();
  if (threadIdx.x < 4 && red[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)red[threadIdx.x]);
}

extern "C" int hpri_seg_counts(const float* pred, const float* target, long long n, float threshold, int is_logits,
                               long long* counts, hipStream_t stream) {
  HPRI_REQUIRE(pred && target && counts && n > 0, "seg_counts: bad arguments");
  hipLaunchKernelGGL(seg_counts_kernel, dim3(step_blocks(n)), dim3(STEP_THREADS), 0, stream, pred, target, n, threshold,
                     is_logits, reinterpret_cast<unsigned long long*>(counts));
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ------------------------------------------------------------------------------------------------
// Binned precision-recall curve (torchmetrics 1.2.0 BinaryPrecisionRecallCurve with `thresholds=T`: the state is,
// per threshold t_k, the 2x2 confusion matrix of (pred >= t_k) vs target).  Instead of T comparisons per pixel, each
// pixel is dropped into the bin b = #{k : t_k <= p} of its class; the confusion matrices are suffix sums of the two
// histograms (host side).  hist layout: [2 classes][T + 1 bins] int64, accumulated across calls.
// ------------------------------------------------------------------------------------------------
#define PR_MAX_T 4096
__global__ __launch_bounds__(STEP_THREADS) void pr_hist_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               long long n, const float* __restrict__ thr, int T,
                                                               int is_logits, unsigned long long* __restrict__ hist) {
  extern __shared__ unsigned int lh[];   // [2][T + 1] counts, then T thresholds
  float* lt = reinterpret_cast<float*>(lh + 2 * (T + 1));
  for (int i = threadIdx.x; i < 2 * (T + 1); i += STEP_THREADS) lh[i] = 0;
  for (int i = threadIdx.x; i < T; i += STEP_THREADS) lt[i] = thr[i];
  __syncthreads();
  const float scale = (float)(T - 1);
  for (long long i = (long long)blockIdx.x * STEP_THREADS + threadIdx.x; i < n; i += (long long)gridDim.x * STEP_THREADS) {
    const float p = is_logits ? sigmoid_f32(x[i]) : x[i];
    int g = (int)(fminf(fmaxf(p, 0.f), 1.f) * scale);       // guess, then make it exact against the table
    if (g > T - 1) g = T - 1;
    if (g < 0) g = 0;
    while (g + 1 < T && lt[g + 1] <= p) ++g;
    while (g >= 0 && !(lt[g] <= p)) --g;                    // NaN ends in bin 0 (no threshold passes)
    const int pos = ((int)y[i]) != 0;
    atomicAdd(&lh[pos * (T + 1) + g + 1], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * (T + 1); i += STEP_THREADS)
    if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

extern "C" int hpri_pr_curve_hist(const float* pred, const float* target, long long n, const float* thresholds, int T,
                                  int is_logits, long long* hist, hipStream_t stream) {
  HPRI_REQUIRE(pred && target && thresholds && hist && n > 0, "pr_curve_hist: bad arguments");
  HPRI_REQUIRE(T >= 2 && T <= PR_MAX_T, "pr_curve_hist: 2 <= thresholds <= 4096");
  const size_t lds = (size_t)(2 * (T + 1)) * sizeof(unsigned int) + (size_t)T * sizeof(float);
  int nb = step_blocks(n);
  if (nb > 512) nb = 512;
  hipLaunchKernelGGL(pr_hist_kernel, dim3(nb), dim3(STEP_THREADS), lds, stream, pred, target, n, thresholds, T, is_logits,
                     reinterpret_cast<unsigned long long*>(hist));
  HPRI_CHECK_LAUNCH();
  return HPRI_OK;
}

// ------------------------------------------------------------------------------------------------
// Multi-tensor optimizer step.  Up to OPT_CHUNK tensors per launch, descriptors by value in the kernel arguments
// (no device-side table to keep in sync with p.grad pointers that change every step).
//   Adam (torch.optim.Adam, amsgrad=False, maximize=False):
//     g += wd * p;  m += (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;
//     p += (-(lr / bc1) * m) / (sqrt(v) / sqrt(bc2) + eps)          bc_i = 1 - b_i^step
//   SGD (torch.optim.SGD, dampening=0, nesterov=False):
//     g += wd * p;  buf = first ? g : momentum * buf + g;  p -= lr * buf
// ------------------------------------------------------------------------------------------------
#define OPT_CHUNK 48
#define OPT_ELEMS_PER_BLOCK (STEP_THREADS * 16)
struct OptTensors {
  float* p[OPT_CHUNK];
  const float* g[OPT_CHUNK];
  float* s0[OPT_CHUNK];          // Adam: exp_avg      SGD: momentum buffer
  float* s1[OPT_CHUNK];          // Adam: exp_avg_sq
  long long n[OPT_CHUNK];
  int blk_end[OPT_CHUNK];        // exclusive prefix of blocks per tensor
  int count;
};

struct AdamScalars { float b1, b2, eps, wd, step_size, inv_sqrt_bc2; };

__global__ __launch_bounds__(STEP_THREADS) void adam_kernel(OptTensors t, AdamScalars a, const float* __restrict__ gscale) {
  int k = 0;
  while (k < t.count - 1 && (int)blockIdx.x >= t.blk_end[k]) ++k;
  const int b0 = k ? t.blk_end[k - 1] : 0;
  const long long base = (long long)(blockIdx.x - b0) * OPT_ELEMS_PER_BLOCK;
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.s0[k];
  float* __restrict__ v = t.s1[k];
  const long long n = t.n[k];
  const float gs = gscale ? gscale[0] : 1.f;
  const float omb1 = 1.f - a.b1, omb2 = 1.f - a.b2;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const long long i = base + j * STEP_THREADS + threadIdx.x;
    if (i < n) {
      const float pi = p[i];
      float gi = g[i] * gs;
      if (a.wd != 0.f) gi = gi + a.wd * pi;
      float mi = m[i];
      mi = mi + (gi - mi) * omb1;
      const float vi = v[i] * a.b2 + (omb2 * gi) * gi;
      const float denom = sqrtf(vi) * a.inv_sqrt_bc2 + a.eps;
      p[i] = pi + (-a.step_size * mi) / denom;
      m[i] = mi;
      v[i] = vi;
    }
  }
}

__global__ __launch_bounds__(STEP_THREADS) void sgd_kernel(OptTensors t, float lr, float momentum, float wd, int first,
                                                           const float* __restrict__ gscale) {
  int k = 0;
  while (k < t.count - 1 && (int)blockIdx.x >= t.blk_end[k]) ++k;
  const int b0 = k ? t.blk_end[k - 1] : 0;
  const long long base = (long long)(blockIdx.x - b0) * OPT_ELEMS_PER_BLOCK;
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ buf = t.s0[k];
  const long long n = t.n[k];
  const float gs = gscale ? gscale[0] : 1.f;
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const long long i = base + j * STEP_THREADS + threadIdx.x;
    if (i < n) {
      const float pi = p[i];
      float gi = g[i] * gs;
      if (wd != 0.f) gi = gi + wd * pi;
      if (buf != nullptr) {
        const float bi = first ? gi : buf[i] * momentum + gi;
        buf[i] = bi;
        gi = bi;
      }
      p[i] = pi - lr * gi;
    }
  }
}

template <class Launch>
static int opt_for_chunks(float* const* p, const float* const* g, float* const* s0, float* const* s1, const long long* n,
                          int ntensors, Launch&& launch) {
  for (int c0 = 0; c0 < ntensors; c0 += OPT_CHUNK) {
    OptTensors t;
    t.count = ntensors - c0 < OPT_CHUNK ? ntensors - c0 : OPT_CHUNK;
    long long blocks = 0;
    for (int k = 0; k < t.count; ++k) {
      t.p[k] = p[c0 + k]; t.g[k] = g[c0 + k];
      t.s0[k] = s0 ? s0[c0 + k] : nullptr;
      t.s1[k] = s1 ? s1[c0 + k] : nullptr;
      t.n[k] = n[c0 + k];
      blocks += (n[c0 + k] + OPT_ELEMS_PER_BLOCK - 1) / OPT_ELEMS_PER_BLOCK;
      if (blocks > 0x7fffffffLL) return hpri_set_error(HPRI_ERR_ARG, "optimizer step: too many elements in one chunk");
      t.blk_end[k] = (int)blocks;
    }
    for (int k = t.count; k < OPT_CHUNK; ++k) {
      t.p[k] = nullptr; t.g[k] = nullptr; t.s0[k] = nullptr; t.s1[k] = nullptr; t.n[k] = 0; t.blk_end[k] = (int)blocks;
    }
    if (blocks == 0) continue;
    const int rc = launch(t, (int)blocks);
    if (rc != HPRI_OK) return rc;
  }
  return HPRI_OK;
}

extern "C" int hpri_adam_step(float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const long long* numel, int ntensors, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                              hipStream_t stream) {
  HPRI_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && ntensors > 0 && step >= 1, "adam_step: bad arguments");
  for (int k = 0; k < ntensors; ++k)
    HPRI_REQUIRE(numel[k] >= 0 && (numel[k] == 0 || (params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k])),
                 "adam_step: null tensor pointer");
  // the scalar prefactors in double, as torch's _single_tensor_adam computes them on the host
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  AdamScalars a;
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.step_size = (float)((double)lr / bc1);
  a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  return opt_for_chunks(params, grads, exp_avg, exp_avg_sq, numel, ntensors, [&](const OptTensors& t, int blocks) {
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(STEP_THREADS), 0, stream, t, a, grad_scale);
    HPRI_CHECK_LAUNCH();
    return HPRI_OK;
  });
}

__global__ __launch_bounds__(STEP_THREADS) void scale_kernel(OptTensors t, float scale) {
  int k = 0;
  while (k < t.count - 1 && (int)blockIdx.x >= t.blk_end[k]) ++k;
  const int b0 = k ? t.blk_end[k - 1] : 0;
  const long long base = (long long)(blockIdx.x - b0) * OPT_ELEMS_PER_BLOCK;
  float* __restrict__ p = t.p[k];
  const long long n = t.n[k];
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const long long i = base + j * STEP_THREADS + threadIdx.x;
    if (i < n) p[i] *= scale;
  }
}

// tensors[k][0 .. numel[k]) *= scale, all tensors in one launch per 48 (the half-precision mode takes its loss scale out of the
// parameter gradients with this: hyperpri_amd/engine.py)
extern "C" int hpri_scale_tensors(float* const* tensors, const long long* numel, int ntensors, float scale, hipStream_t stream) {
  HPRI_REQUIRE(tensors && numel && ntensors > 0, "scale_tensors: bad arguments");
  for (int k = 0; k < ntensors; ++k) HPRI_REQUIRE(numel[k] >= 0 && (numel[k] == 0 || tensors[k]), "scale_tensors: null tensor pointer");
  return opt_for_chunks(tensors, reinterpret_cast<const float* const*>(tensors), nullptr, nullptr, numel, ntensors,
                        [&](const OptTensors& t, int blocks) {
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(STEP_THREADS), 0, stream, t, scale);
    HPRI_CHECK_LAUNCH();
    return HPRI_OK;
  });
}

extern "C" int hpri_sgd_step(float* const* params, const float* const* grads, float* const* momentum_buf,
                             const long long* numel, int ntensors, float lr, float momentum, float weight_decay,
                             int first_step, const float* grad_scale, hipStream_t stream) {
  HPRI_REQUIRE(params && grads && numel && ntensors > 0, "sgd_step: bad arguments");
  HPRI_REQUIRE(momentum == 0.f || momentum_buf != nullptr, "sgd_step: momentum needs buffers");
  for (int k = 0; k < ntensors; ++k)
    HPRI_REQUIRE(numel[k] >= 0 && (numel[k] == 0 || (params[k] && grads[k])), "sgd_step: null tensor pointer");
  return opt_for_chunks(params, grads, momentum != 0.f ? momentum_buf : nullptr, nullptr, numel, ntensors,
                        [&](const OptTensors& t, int blocks) {
    hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(STEP_THREADS), 0, stream, t, lr, momentum, weight_decay, first_step,
                       grad_scale);
    HPRI_CHECK_LAUNCH();
    return HPRI_OK;
  });
}
