// rau_dev.hip -- device-tensor helpers of the C ABI (include/rau.h "device tensors").
//
// A host that keeps the reference's own feval loops and calls the clones one by one
// (experiments/Ours_SS/LstmAttCtrlGradNoiseDontSelect.lua:443-596) does a little tensor algebra
// of its own between those calls: `rnn_out[k] = lst[k]` row by row where x_len[k] == t
// (SS:455-461, 585-591), `uni_pred:add(pred[1])` (SS:522-526), `torch.max(pred[1], 2)` and
// `ans:eq(y):sum()` (SS:488-492), zero-filled state tensors (SS:357-413).  On the reference's
// CUDA box cutorch does that; on an MI355X host there is no cutorch, so the same handful of
// operations is exported here on plain device pointers, enqueued on the ctx stream like everything
// else.  bindings/rau.lua wraps them in a tensor-shaped object.
#include <cmath>

#include "rau_ctx.h"

namespace {

__global__ void k_fill(size_t n, float v, float* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] = v;
}
__global__ void k_axpy(size_t n, float a, const float* __restrict__ x, float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}
__global__ void k_scale(size_t n, float a, float* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] *= a;
}
// dst[r,:] = src[r,:] for the rows with key[r] == value
__global__ void k_select_rows(int rows, int cols, const int32_t* __restrict__ key, int32_t value,
                              const float* __restrict__ src, float* __restrict__ dst) {
  const size_t n = (size_t)rows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    if (key[i / cols] == value) dst[i] = src[i];
}
// torch.max(x, 2): per row the maximum and the FIRST index attaining it, 1-based
__global__ void k_rowmax(int cols, const float* __restrict__ x, float* __restrict__ mv,
                         int32_t* __restrict__ mi) {
  const float* r = x + (size_t)blockIdx.x * cols;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = threadIdx.x; k < cols; k += 64) {
    const float v = r[k];
    if (v > best) { best = v; bi = k; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (threadIdx.x == 0) {
    if (mv) mv[blockIdx.x] = best;
    if (mi) mi[blockIdx.x] = bi + 1;
  }
}
// one block: out[0] = sum x (fixed order: lane-strided partials, wave tree, waves in order)
__global__ void k_sum(size_t n, const float* __restrict__ x, double* __restrict__ out) {
  __shared__ double s[4];
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += 256) acc += (double)x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ void k_count_eq(int n, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                           int32_t* __restrict__ out) {
  __shared__ int s[4];
  int acc = 0;
  for (int i = threadIdx.x; i < n; i += 256) acc += a[i] == b[i] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s[0] + s[1]) + (s[2] + s[3]);
}

// y += a * x1 * x2   /   y += a * x1 / x2   (torch addcmul / addcdiv, utils/optim_updates.lua:80-87)
__global__ void k_addcmul(size_t n, float a, const float* __restrict__ x1, const float* __restrict__ x2,
                          float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] += a * x1[i] * x2[i];
}
__global__ void k_addcdiv(size_t n, float a, const float* __restrict__ x1, const float* __restrict__ x2,
                          float* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] += a * x1[i] / x2[i];
}
__global__ void k_sqrt(size_t n, float* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] = sqrtf(x[i]);
}
__global__ void k_add_scalar(size_t n, float v, float* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] += v;
}
// adam(x, dx, ..) of utils/optim_updates.lua:59-87 on one flat vector, its five tensor statements
// fused into one pass: m = b1 m + (1-b1) dx; v = b2 v + (1-b2) dx^2; x -= step m / (sqrt(v) + eps)
__global__ void k_adam_vec(size_t n, float b1, float b2, float eps, float step, const float* __restrict__ dx,
                           float* __restrict__ m, float* __restrict__ v, float* __restrict__ x) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float g = dx[i];
    const float mi = m[i] * b1 + (1.f - b1) * g;
    const float vi = v[i] * b2 + (1.f - b2) * g * g;
    m[i] = mi;
    v[i] = vi;
    x[i] -= step * mi / (sqrtf(vi) + eps);
  }
}

inline int blocks_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 2048); }

}  // namespace

extern "C" {

int rau_dev_alloc(rau_ctx* ctx, size_t n, float** out) {
  NEED(ctx && out, "null argument");
  return dalloc(ctx, out, n);            // zero-filled, owned by the ctx
}
int rau_dev_free(rau_ctx* ctx, float* p) {
  NEED(ctx, "null ctx");
  if (!p) return RAU_OK;
  auto it = std::find(ctx->allocs.begin(), ctx->allocs.end(), (void*)p);
  NEED(it != ctx->allocs.end(), "rau_dev_free: pointer was not allocated by rau_dev_alloc");
  HIPC(hipStreamSynchronize(ctx->st));
  HIPC(hipFree(p));
  ctx->allocs.erase(it);
  return RAU_OK;
}
int rau_dev_fill(rau_ctx* ctx, float* dst, size_t n, float value) {
  NEED(ctx && (dst || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_fill, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, value, dst);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_copy(rau_ctx* ctx, float* dst, const float* src, size_t n) {
  NEED(ctx && ((dst && src) || !n), "null argument");
  if (n) HIPC(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->st));
  return RAU_OK;
}
int rau_dev_axpy(rau_ctx* ctx, float* y, const float* x, size_t n, float alpha) {
  NEED(ctx && ((y && x) || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_axpy, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, alpha, x, y);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_scale(rau_ctx* ctx, float* x, size_t n, float alpha) {
  NEED(ctx && (x || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_scale, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, alpha, x);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_addcmul(rau_ctx* ctx, float* y, float alpha, const float* x1, const float* x2, size_t n) {
  NEED(ctx && ((y && x1 && x2) || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_addcmul, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, alpha, x1, x2, y);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_addcdiv(rau_ctx* ctx, float* y, float alpha, const float* x1, const float* x2, size_t n) {
  NEED(ctx && ((y && x1 && x2) || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_addcdiv, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, alpha, x1, x2, y);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_sqrt(rau_ctx* ctx, float* x, size_t n) {
  NEED(ctx && (x || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_sqrt, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, x);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_add_scalar(rau_ctx* ctx, float* x, size_t n, float value) {
  NEED(ctx && (x || !n), "null argument");
  if (n) hipLaunchKernelGGL(k_add_scalar, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, value, x);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_adam(rau_ctx* ctx, float* x, const float* dx, float* m, float* v, size_t n, float lr,
                 float beta1, float beta2, float eps, int32_t t) {
  NEED(ctx && ((x && dx && m && v) || !n), "null argument");
  NEED(t >= 1, "rau_dev_adam: t = %d (the step counter AFTER its increment, >= 1)", t);
  // step size on the host in double like LuaJIT's numbers (optim_updates.lua:80-83)
  const double bc1 = 1.0 - std::pow((double)beta1, (double)t), bc2 = 1.0 - std::pow((double)beta2, (double)t);
  const float step = (float)((double)lr * std::sqrt(bc2) / bc1);
  if (n)
    hipLaunchKernelGGL(k_adam_vec, dim3(blocks_for(n)), dim3(256), 0, ctx->st, n, beta1, beta2, eps, step,
                       dx, m, v, x);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_select_rows(rau_ctx* ctx, float* dst, const float* src, int32_t rows, int32_t cols,
                        const int32_t* key_dev, int32_t value) {
  NEED(ctx && dst && src && key_dev, "null argument");
  NEED(rows >= 0 && cols > 0, "rau_dev_select_rows: bad shape %d x %d", rows, cols);
  const size_t n = (size_t)rows * cols;
  if (n)
    hipLaunchKernelGGL(k_select_rows, dim3(blocks_for(n)), dim3(256), 0, ctx->st, rows, cols, key_dev,
                       value, src, dst);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_rowmax(rau_ctx* ctx, const float* x, int32_t rows, int32_t cols, float* max_dev,
                   int32_t* argmax_dev) {
  NEED(ctx && x, "null argument");
  NEED(rows >= 0 && cols > 0, "rau_dev_rowmax: bad shape %d x %d", rows, cols);
  if (rows) hipLaunchKernelGGL(k_rowmax, dim3(rows), dim3(64), 0, ctx->st, cols, x, max_dev, argmax_dev);
  HIPC(hipGetLastError());
  return RAU_OK;
}
int rau_dev_sum(rau_ctx* ctx, const float* x, size_t n, double* out_host) {
  NEED(ctx && out_host && (x || !n), "null argument");
  double* tmp = reinterpret_cast<double*>(ctx->norms_d);   // 4 floats = 2 doubles of scratch
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(256), 0, ctx->st, n, x, tmp);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(out_host, tmp, sizeof(double), hipMemcpyDeviceToHost, ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  return RAU_OK;
}
int rau_dev_count_eq(rau_ctx* ctx, const int32_t* a_dev, const int32_t* b_dev, int32_t n,
                     int32_t* count_host) {
  NEED(ctx && a_dev && b_dev && count_host, "null argument");
  int32_t* tmp = reinterpret_cast<int32_t*>(ctx->norms_d);
  hipLaunchKernelGGL(k_count_eq, dim3(1), dim3(256), 0, ctx->st, n, a_dev, b_dev, tmp);
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(count_host, tmp, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->st));
  HIPC(hipStreamSynchronize(ctx->st));
  return RAU_OK;
}
int rau_dev_upload(rau_ctx* ctx, void* dst_dev, const void* host, size_t bytes) {
  NEED(ctx && ((dst_dev && host) || !bytes), "null argument");
  if (bytes) {
    HIPC(hipMemcpyAsync(dst_dev, host, bytes, hipMemcpyHostToDevice, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));   // the host buffer may be reused right away
  }
  return RAU_OK;
}
int rau_dev_download(rau_ctx* ctx, void* host, const void* src_dev, size_t bytes) {
  NEED(ctx && ((host && src_dev) || !bytes), "null argument");
  if (bytes) {
    HIPC(hipMemcpyAsync(host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->st));
    HIPC(hipStreamSynchronize(ctx->st));
  }
  return RAU_OK;
}

}  // extern "C"
