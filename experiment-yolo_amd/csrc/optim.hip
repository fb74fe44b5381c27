// Flat-buffer optimizer step: unscale + inf check + global-norm clip + SGD(nesterov) | AdamW + EMA in two launches
// over ONE contiguous fp32 parameter buffer laid out as [bias group | decayed weights | norm weights].
// Replaces BaseTrainer.optimizer_step (reference engine/trainer.py:949-957: GradScaler.unscale_, clip_grad_norm_(10),
// optimizer.step over ~190 tensors, scaler.update) and ModelEMA.update's Python loop over the state_dict
// (utils/torch_utils.py:447-458).  The GradScaler policy (halve on inf/nan, double after 2000 clean steps) runs on
// device, so no host synchronisation is needed to decide whether to skip a step.
#include "common.h"
#include "dealyolo_hip.h"

// hyper[] (float, written by the host before each step): 0..2 lr per group, 3 momentum/beta1, 4..6 weight decay per
// group, 7 ema decay, 8 max grad norm, 9 adam beta2, 10 adam eps, 11 dynamic loss scale on (1) / off (0: amp=False -- the
// scale stays where it is; a non-finite step is still skipped and counted, and the trainer raises on it)
// state[] (float, device-resident): 0 loss scale, 1 growth tracker, 2 found_inf (this step), 3 grad norm (unscaled),
// 4 clip coefficient, 5 optimizer steps taken, 6 skipped steps
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* g, long n, float* partials) {
  float s = 0.f;
  int bad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = g[i];
    if (!(fabsf(v) <= 3.0e38f)) bad = 1;  // inf or nan
    s += v * v;
  }
  __shared__ float red[256];
  __shared__ int rbad[256];
  red[threadIdx.x] = s;
  rbad[threadIdx.x] = bad;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      rbad[threadIdx.x] |= rbad[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partials[blockIdx.x * 2] = red[0];
    partials[blockIdx.x * 2 + 1] = (float)rbad[0];
  }
}

__global__ __launch_bounds__(256) void grad_norm_final_kernel(const float* partials, int np, const float* hyper,
                                                              float* state) {
  __shared__ double red[256];
  __shared__ int rbad[256];
  double s = 0.0;
  int bad = 0;
  for (int i = threadIdx.x; i < np; i += 256) {
    s += partials[i * 2];
    bad |= partials[i * 2 + 1] != 0.f;
  }
  red[threadIdx.x] = s;
  rbad[threadIdx.x] = bad;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      rbad[threadIdx.x] |= rbad[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float scale = state[0];
    const int found = rbad[0] || !(red[0] <= 1.0e300);
    const float norm = found ? 0.f : (float)(sqrt(red[0]) / (double)scale);
    state[2] = (float)found;
    state[3] = norm;
    const float coef = hyper[8] / (norm + 1e-6f);  // clip_grad_norm_: clamp(max_norm / (total + 1e-6), max=1)
    state[4] = coef < 1.f ? coef : 1.f;
  }
}

struct StepArgs {
  float* p;
  const float* g;
  float* m;   // momentum buffer / adam exp_avg
  float* v;   // adam exp_avg_sq (null for SGD)
  float* ema; // may be null
  const float* hyper;
  float* state;
  long n;
  long b[5];  // ends of the first five of six segments; segment k holds optimizer group k % 3 (bias, decayed weights, norm weights):
              // [neck + head: g0 | g1 | g2][backbone: g0 | g1 | g2] -- or, from dy_optimizer_step, [g0 | g1 | g2] and three empty ones
  const uint8_t* frozen;   // optional per-element mask (1 = requires_grad False)
  int adam;                // 0 SGD-nesterov, 1 Adam (L2 decay), 2 AdamW (decoupled), 3 RMSprop(alpha .99, momentum), 4 RAdam, 5 Adamax, 6 NAdam
};

__global__ __launch_bounds__(256) void optim_step_kernel(StepArgs a) {
  const float found = a.state[2];
  const float inv_scale = 1.f / a.state[0], coef = a.state[4];
  const float mom = a.hyper[3], d = a.hyper[7];
  const bool first = a.state[5] == 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
    const int seg = (i >= a.b[0]) + (i >= a.b[1]) + (i >= a.b[2]) + (i >= a.b[3]) + (i >= a.b[4]);
    const int grp = seg >= 3 ? seg - 3 : seg;
    float p = a.p[i];
    if (found == 0.f && !(a.frozen && a.frozen[i])) {
      const float lr = a.hyper[grp], wd = a.hyper[4 + grp];
      float g = a.g[i] * inv_scale * coef;
      if (a.adam == 0) {
        if (wd != 0.f) g += wd * p;
        float buf = first ? g : mom * a.m[i] + g;
        a.m[i] = buf;
        p -= lr * (g + mom * buf);
      } else if (a.adam == 3) {  // torch.optim.RMSprop(lr, alpha=0.99, eps=1e-8, momentum): engine/trainer.py:1161-1162
        if (wd != 0.f) g += wd * p;
        const float sq = 0.99f * (first ? 0.f : a.v[i]) + 0.01f * g * g;
        a.v[i] = sq;
        const float step = g / (sqrtf(sq) + 1e-8f);
        const float buf = mom * (first ? 0.f : a.m[i]) + step;
        a.m[i] = buf;
        p -= lr * (mom > 0.f ? buf : step);
      } else if (a.adam == 6) {  // torch.optim.NAdam (betas = (momentum, 0.999), eps 1e-8, momentum_decay 4e-3, L2 decay)
        if (wd != 0.f) g += wd * p;
        const float b2 = a.hyper[9], t = a.state[5] + 1.f;
        const float mu = mom * (1.f - 0.5f * powf(0.96f, t * 0.004f)), mu_next = mom * (1.f - 0.5f * powf(0.96f, (t + 1.f) * 0.004f));
        const float mu_prod = (first ? 1.f : a.state[7]) * mu;  // state[7]: product of the mu of the steps taken so far
        const float m = mom * (first ? 0.f : a.m[i]) + (1.f - mom) * g;
        const float v = b2 * (first ? 0.f : a.v[i]) + (1.f - b2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const float denom = sqrtf(v / (1.f - powf(b2, t))) + 1e-8f;
        p -= lr * (1.f - mu) / (1.f - mu_prod) * g / denom;
        p -= lr * mu_next / (1.f - mu_prod * mu_next) * m / denom;
      } else if (a.adam == 4 || a.adam == 5) {  // torch.optim.RAdam / Adamax (betas = (momentum, 0.999), eps 1e-8, L2 decay)
        if (wd != 0.f) g += wd * p;
        const float b2 = a.hyper[9], t = a.state[5] + 1.f;
        const float m = mom * (first ? 0.f : a.m[i]) + (1.f - mom) * g;
        a.m[i] = m;
        const float bc1 = 1.f - powf(mom, t);
        if (a.adam == 5) {
          const float u = fmaxf(b2 * (first ? 0.f : a.v[i]), fabsf(g) + 1e-8f);
          a.v[i] = u;
          p -= lr / bc1 * m / u;
        } else {
          const float v = b2 * (first ? 0.f : a.v[i]) + (1.f - b2) * g * g;
          a.v[i] = v;
          const float b2t = powf(b2, t), bc2 = 1.f - b2t;
          const float rho_inf = 2.f / (1.f - b2) - 1.f, rho = rho_inf - 2.f * t * b2t / bc2;
          float upd = m / bc1;
          if (rho > 5.f)
            upd *= sqrtf((rho - 4.f) * (rho - 2.f) * rho_inf / ((rho_inf - 4.f) * (rho_inf - 2.f) * rho)) * (sqrtf(bc2) / (sqrtf(v) + 1e-8f));
          p -= lr * upd;
        }
      } else {
        if (a.adam == 2) p *= 1.f - lr * wd;
        else if (wd != 0.f) g += wd * p;
        const float b2 = a.hyper[9];
        const float m = mom * (first ? 0.f : a.m[i]) + (1.f - mom) * g;
        const float v = b2 * (first ? 0.f : a.v[i]) + (1.f - b2) * g * g;
        a.m[i] = m;
        a.v[i] = v;
        const float t = a.state[5] + 1.f;  // steps actually taken (skipped steps do not advance Adam's clock)
        const float bc1 = 1.f - powf(mom, t), bc2 = 1.f - powf(b2, t);
        p -= lr / bc1 * m / (sqrtf(v) / sqrtf(bc2) + a.hyper[10]);
      }
      a.p[i] = p;
    }
    if (a.ema) a.ema[i] = d * a.ema[i] + (1.f - d) * p;  // ModelEMA.update runs whether or not the step was skipped
  }
}

// EMA of the floating-point buffers (BN running statistics) + GradScaler.update bookkeeping
__global__ __launch_bounds__(256) void ema_buffers_kernel(const float* s, float* es, long n, const float* hyper,
                                                          float* state, int update_scaler, int mode) {
  const float d = hyper[7], found = state[2];
  if (es)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
      es[i] = d * es[i] + (1.f - d) * s[i];
  if (update_scaler && blockIdx.x == 0 && threadIdx.x == 0) {
    const bool dynamic = hyper[11] != 0.f;
    if (found != 0.f) {
      if (dynamic) state[0] *= 0.5f;
      state[1] = 0.f;
      state[6] += 1.f;
    } else {
      if (mode == 6) {  // NAdam's running mu product (the parameter kernel of this step has already read the old value)
        const float t = state[5] + 1.f;
        state[7] = (state[5] == 0.f ? 1.f : state[7]) * hyper[3] * (1.f - 0.5f * powf(0.96f, t * 0.004f));
      }
      state[5] += 1.f;
      state[1] += 1.f;
      if (dynamic && state[1] >= 2000.f) {
        state[0] *= 2.f;
        state[1] = 0.f;
      }
    }
  }
}

static int optimizer_step_impl(float* params, const float* grads, float* mom, float* adam_v, float* ema, long n, const long* b5,
                               const unsigned char* frozen, const float* buffers, float* ema_buffers, long n_buffers, const float* hyper,
                               float* state, float* partials, int mode, hipStream_t stream);
extern "C" int dy_optimizer_step(float* params, const float* grads, float* mom, float* adam_v, float* ema, long n,
                                 long g0_end, long g1_end, const unsigned char* frozen, const float* buffers,
                                 float* ema_buffers, long n_buffers, const float* hyper, float* state,
                                 float* partials, int mode, hipStream_t stream) {
  const long b5[5] = {g0_end, g1_end, n, n, n};
  return optimizer_step_impl(params, grads, mom, adam_v, ema, n, b5, frozen, buffers, ema_buffers, n_buffers, hyper, state, partials, mode, stream);
}
// The same step over a flat buffer laid out in SIX segments [bucket A: g0 | g1 | g2][bucket B: g0 | g1 | g2] (seg_ends: the ends of
// the first five): the layout that makes each data-parallel gradient bucket ONE contiguous slice (hip/runtime.py; reference
// engine/trainer.py:694-695: DDP's reducer works on contiguous buckets).
extern "C" int dy_optimizer_step_seg(float* params, const float* grads, float* mom, float* adam_v, float* ema, long n, const long* seg_ends,
                                     const unsigned char* frozen, const float* buffers, float* ema_buffers, long n_buffers,
                                     const float* hyper, float* state, float* partials, int mode, hipStream_t stream) {
  if (!seg_ends) return DY_ERR_ARG;
  for (int k = 0; k < 5; ++k)
    if (seg_ends[k] < (k ? seg_ends[k - 1] : 0) || seg_ends[k] > n) return DY_ERR_ARG;
  return optimizer_step_impl(params, grads, mom, adam_v, ema, n, seg_ends, frozen, buffers, ema_buffers, n_buffers, hyper, state, partials, mode, stream);
}
static int optimizer_step_impl(float* params, const float* grads, float* mom, float* adam_v, float* ema, long n, const long* b5,
                               const unsigned char* frozen, const float* buffers, float* ema_buffers, long n_buffers, const float* hyper,
                               float* state, float* partials, int mode, hipStream_t stream) {
  if (n <= 0 || mode < 0 || mode > 6 || (mode > 0 && !adam_v)) return DY_ERR_ARG;
  int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, stream, grads, n, partials);
  hipLaunchKernelGGL(grad_norm_final_kernel, dim3(1), dim3(256), 0, stream, partials, blocks, hyper, state);
  StepArgs a{params, grads, mom, adam_v, ema, hyper, state, n, {b5[0], b5[1], b5[2], b5[3], b5[4]}, frozen, mode};
  hipLaunchKernelGGL(optim_step_kernel, dim3(blocks), dim3(256), 0, stream, a);
  int b2 = (int)((n_buffers + 255) / 256);
  if (b2 < 1) b2 = 1;
  if (b2 > 256) b2 = 256;
  hipLaunchKernelGGL(ema_buffers_kernel, dim3(b2), dim3(256), 0, stream, buffers, ema_buffers, n_buffers, hyper, state,
                     1, mode);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// The per-step hyper-parameters travel as KERNEL ARGUMENTS (copied at enqueue time), not through a reused pinned host buffer:
// the host may be many steps ahead of the device, and an asynchronous copy from a buffer the host has since overwritten would
// hand step i the learning rate of step i+k.
struct Hyper16 { float v[16]; };
__global__ void set_hyper_kernel(float* dst, Hyper16 h) {
  if (threadIdx.x < 16) dst[threadIdx.x] = h.v[threadIdx.x];
}
extern "C" int dy_set_hyper(float* hyper_dev, const float* host_values16, hipStream_t stream) {
  if (!hyper_dev || !host_values16) return DY_ERR_ARG;
  Hyper16 h;
  for (int i = 0; i < 16; ++i) h.v[i] = host_values16[i];
  hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(64), 0, stream, hyper_dev, h);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// y += a * x (fp32), used for gradient accumulation across micro-steps
__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, float a, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += a * x[i];
}
extern "C" int dy_axpy_f32(float* y, const float* x, float a, long n, hipStream_t stream) {
  int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, stream, y, x, a, n);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
