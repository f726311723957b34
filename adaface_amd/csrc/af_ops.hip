// Operator-level C-ABI entry points used by the parity tests (tests/test_ops_gpu.py).
// They take fp32 device tensors in the REFERENCE's layouts (NCHW, [rows, C],
// [B, N, heads*dh]), convert to the internal NHWC storage dtype, run exactly the
// kernels the UNet / VAE executors use, and convert back.  Temporary buffers are
// hipMalloc'd per call (test path only, never on the hot path).
#include "../../include/adaface_hip.h"
#include "af_kernels.h"

#include <algorithm>
#include <cstring>
#include <vector>

namespace {
struct Tmp {
  std::vector<void*> bufs;
  void* get(size_t bytes, bool zero = false) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    if (zero) hipMemset(p, 0, bytes);
    bufs.push_back(p);
    return p;
  }
  ~Tmp() {
    hipDeviceSynchronize();
    for (void* p : bufs) hipFree(p);
  }
};
inline size_t esz(int dtype) { return dtype == AF_DTYPE_BF16 ? 2 : 4; }
inline int bk(int dtype) { return dtype == AF_DTYPE_BF16 ? 64 : 32; }
inline int rup(int a, int b) { return (a + b - 1) / b * b; }
}  // namespace

#define OP_TRY(expr)          \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)
#define DISP(dtype, A, B) ((dtype) == AF_DTYPE_BF16 ? (A) : (B))
#define OP_ALLOC(var, bytes, zero)                               \
  void* var = tmp.get((bytes), (zero));                          \
  if (!var) { af_set_error_msg("hipMalloc failed in op"); return AF_ERR_HIP; }

// ---- device clock probe (bench.py stamps every line with it: devices of one pool hold different clocks under MFMA load, so
// whole-path numbers from different boxes are only comparable through a number like this) ----
// Every wave runs `iters` rounds of four independent v_mfma_f32_32x32x16_bf16 on pseudo-random register operands (no memory
// traffic, no LDS) between two (s_memtime, s_memrealtime) stamp pairs: shader cycles / 100 MHz reference ticks = the clock
// the chip holds under a dense MFMA stream.  The stamps go to a buffer nothing else reads; the accumulators to a sink.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ stamps, float* __restrict__ sink,
                                                          int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned hsh = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hsh = hsh * 1664525u + 1013904223u;
    a[e] = (bf16)(((int)(hsh >> 16) & 0xFF) * (1.0f / 128.0f) - 1.0f);
    hsh = hsh * 1664525u + 1013904223u;
    b[e] = (bf16)(((int)(hsh >> 16) & 0xFF) * (1.0f / 128.0f) - 1.0f);
  }
  f32x16 c0, c1, c2, c3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
  }
  asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    stamps[((size_t)blockIdx.x * 4 + wave) * 2 + 0] = t1 - t0;
    stamps[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
  }
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc += c0[r] + c1[r] + c2[r] + c3[r];
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

extern "C" {

int af_clock_probe(void* stream, int iters, double* mfma_mhz, double* mfma_tflops) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (iters <= 0) iters = 40000;
  hipDeviceProp_t prop;
  int dev = 0;
  HIP_CHECK_RET(hipGetDevice(&dev));
  HIP_CHECK_RET(hipGetDeviceProperties(&prop, dev));
  const int blocks = 2 * prop.multiProcessorCount;    // two waves per SIMD: every matrix pipe is kept busy
  Tmp tmp;
  OP_ALLOC(stamps, (size_t)blocks * 4 * 2 * sizeof(unsigned long long), true);
  OP_ALLOC(sink, (size_t)blocks * 256 * sizeof(float), false);
  struct Ev {   // destroyed on every exit path
    hipEvent_t e = nullptr;
    ~Ev() { if (e) hipEventDestroy(e); }
  } ev0, ev1;
  HIP_CHECK_RET(hipEventCreate(&ev0.e));
  HIP_CHECK_RET(hipEventCreate(&ev1.e));
  hipEvent_t e0 = ev0.e, e1 = ev1.e;
  // one short launch to wake the device, then the timed one
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, s, (unsigned long long*)stamps, (float*)sink, 2000);
  HIP_CHECK_RET(hipEventRecord(e0, s));
  hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, s, (unsigned long long*)stamps, (float*)sink, iters);
  HIP_CHECK_RET(hipEventRecord(e1, s));
  HIP_CHECK_RET(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)blocks * 8);
  HIP_CHECK_RET(hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  for (size_t i = 0; i < h.size(); i += 2)
    if (h[i + 1] > 0) mhz.push_back((double)h[i] / (double)h[i + 1] * 100.0);
  if (mhz.empty()) { af_set_error_msg("af_clock_probe: no stamps"); return AF_ERR_STATE; }
  std::sort(mhz.begin(), mhz.end());
  if (mfma_mhz) *mfma_mhz = mhz[mhz.size() / 2];
  if (mfma_tflops) *mfma_tflops = (double)blocks * 4 * (double)iters * 4 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
  return AF_OK;
}

int af_op_conv2d(int dtype, const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                 float* y_dev, int B, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int upsample,
                 void* stream) {
  if ((ks != 1 && ks != 3) || pad != ks / 2) { af_set_error_msg("af_op_conv2d: ks must be 1 or 3 with pad ks/2"); return AF_ERR_INVALID; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int cin_pad = rup(Cin, bk(dtype));
  const int Hi = H << upsample, Wi = W << upsample;
  const int Ho = (Hi + 2 * pad - ks) / stride + 1, Wo = (Wi + 2 * pad - ks) / stride + 1;
  const int co4 = rup(Cout, 4), rows_pad = rup(Cout, 128), ldw = ks * ks * cin_pad;
  OP_ALLOC(xn, (size_t)B * H * W * cin_pad * esz(dtype), false);
  OP_ALLOC(wn, (size_t)rows_pad * ldw * esz(dtype), true);
  OP_ALLOC(yn, (size_t)B * Ho * Wo * co4 * esz(dtype), true);
  void* rn = nullptr;
  float* bn = nullptr;
  OP_TRY(DISP(dtype, af_launch_nchw_to_nhwc<bf16>(x_dev, xn, B, Cin, H * W, cin_pad, 1.f, s),
              af_launch_nchw_to_nhwc<float>(x_dev, xn, B, Cin, H * W, cin_pad, 1.f, s)));
  OP_TRY(DISP(dtype, af_launch_repack_weight<bf16>(w_dev, wn, Cout, Cin, cin_pad, ks, ldw, 0, 0, s),
              af_launch_repack_weight<float>(w_dev, wn, Cout, Cin, cin_pad, ks, ldw, 0, 0, s)));
  if (bias_dev) {
    bn = reinterpret_cast<float*>(tmp.get((size_t)rup(Cout, 128) * 4, true));
    if (!bn) return AF_ERR_HIP;
    if (hipMemcpyAsync(bn, bias_dev, (size_t)Cout * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return AF_ERR_HIP;
  }
  if (residual_dev) {
    rn = tmp.get((size_t)B * Ho * Wo * co4 * esz(dtype), false);
    if (!rn) return AF_ERR_HIP;
    OP_TRY(DISP(dtype, af_launch_nchw_to_nhwc<bf16>(residual_dev, rn, B, Cout, Ho * Wo, co4, 1.f, s),
                af_launch_nchw_to_nhwc<float>(residual_dev, rn, B, Cout, Ho * Wo, co4, 1.f, s)));
  }
  ConvGemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = xn; p.src_batch_stride = (long)H * W * cin_pad; p.ldc = cin_pad; p.Cin = cin_pad;
  p.Hs = H; p.Ws = W; p.up = upsample; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo;
  p.ks = ks; p.stride = stride; p.pad = pad;
  p.W = wn; p.ldw = ldw; p.Wrows = rows_pad;
  p.M = B * Ho * Wo; p.N = co4; p.K = ldw;
  p.bias = bn; p.residual = rn; p.ldr = co4; p.out = yn; p.ldo = co4; p.alpha = 1.f;
  p.k_logical = ks * ks * Cin;
  if (upsample && ks == 3 && dtype == AF_DTYPE_BF16 && cin_pad % 64 == 0) {
    // as the model does for its Upsample layers: phase weights next to the 3x3 ones, the launcher decides (AF_CONV_UP_PHASE4)
    OP_ALLOC(w4, (size_t)4 * rows_pad * 4 * cin_pad * 2, false);
    OP_TRY(af_launch_up_phase4_weights(wn, rows_pad, cin_pad, ldw, w4, s));
    p.W_up4 = w4;
  }
  const AfGemmPlan pl = af_plan_conv_gemm(p, 1, (int)esz(dtype));
  void* ws = nullptr;
  if (pl.splitk > 1) { ws = tmp.get(pl.ws_bytes, false); if (!ws) return AF_ERR_HIP; }
  OP_TRY(DISP(dtype, af_launch_conv_gemm<bf16>(p, 1, s, &pl, ws), af_launch_conv_gemm<float>(p, 1, s, &pl, ws)));
  OP_TRY(DISP(dtype, af_launch_nhwc_to_nchw<bf16>(yn, y_dev, B, Cout, Ho * Wo, co4, s),
              af_launch_nhwc_to_nchw<float>(yn, y_dev, B, Cout, Ho * Wo, co4, s)));
  return 0;
}

// fp8 convolution as the UNet's fp8 mode runs it (bf16 storage): x is cast to bf16, multiplied by 2^act_shift and stored
// as e4m3 (what the GroupNorm kernels write), the weight is repacked to bf16 and quantised per output row
// (af_launch_quant_weight_fp8), the ping-pong kernel multiplies on the block-scaled fp8 MFMA; bias / residual / output bf16.
// Returns AF_ERR_INVALID when the shape has no fp8 plan (the model keeps such layers on bf16).
int af_op_conv2d_fp8(const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev, float* y_dev,
                     int B, int Cin, int H, int W, int Cout, int ks, int stride, int pad, int upsample, int act_shift,
                     void* stream) {
  if ((ks != 1 && ks != 3) || pad != ks / 2 || Cin % 64 != 0) { af_set_error_msg("af_op_conv2d_fp8: ks 1|3, pad ks/2, Cin%%64==0"); return AF_ERR_INVALID; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int Hi = H << upsample, Wi = W << upsample;
  const int Ho = (Hi + 2 * pad - ks) / stride + 1, Wo = (Wi + 2 * pad - ks) / stride + 1;
  const int co4 = rup(Cout, 4), rows_pad = rup(Cout, 128), ldw = ks * ks * Cin, k8 = rup(ldw, 128);
  const size_t nx = (size_t)B * H * W * Cin;
  OP_ALLOC(xn, nx * 2, false);
  OP_ALLOC(x8, nx, false);
  OP_ALLOC(wn, (size_t)rows_pad * ldw * 2, true);
  OP_ALLOC(w8, (size_t)rows_pad * k8 + rows_pad, true);
  OP_ALLOC(yn, (size_t)B * Ho * Wo * co4 * 2, true);
  unsigned char* sc = reinterpret_cast<unsigned char*>(w8) + (size_t)rows_pad * k8;
  void* rn = nullptr;
  float* bn = nullptr;
  OP_TRY(af_launch_nchw_to_nhwc<bf16>(x_dev, xn, B, Cin, H * W, Cin, 1.f, s));
  OP_TRY(af_launch_cast_fp8(xn, x8, (long)nx, (float)(1 << act_shift), s));
  OP_TRY(af_launch_repack_weight<bf16>(w_dev, wn, Cout, Cin, Cin, ks, ldw, 0, 0, s));
  OP_TRY(af_launch_quant_weight_fp8(wn, rows_pad, ldw, Cin, ks, w8, k8, sc, s));
  if (bias_dev) {
    bn = reinterpret_cast<float*>(tmp.get((size_t)rows_pad * 4, true));
    if (!bn) return AF_ERR_HIP;
    if (hipMemcpyAsync(bn, bias_dev, (size_t)Cout * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return AF_ERR_HIP;
  }
  if (residual_dev) {
    rn = tmp.get((size_t)B * Ho * Wo * co4 * 2, false);
    if (!rn) return AF_ERR_HIP;
    OP_TRY(af_launch_nchw_to_nhwc<bf16>(residual_dev, rn, B, Cout, Ho * Wo, co4, 1.f, s));
  }
  ConvGemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = x8; p.src_batch_stride = (long)H * W * Cin; p.ldc = Cin; p.Cin = Cin;
  p.Hs = H; p.Ws = W; p.up = upsample; p.Hi = Hi; p.Wi = Wi; p.Ho = Ho; p.Wo = Wo;
  p.ks = ks; p.stride = stride; p.pad = pad;
  p.W = w8; p.ldw = k8; p.Wrows = rows_pad;
  p.M = B * Ho * Wo; p.N = co4; p.K = k8;
  p.bias = bn; p.residual = rn; p.ldr = co4; p.out = yn; p.ldo = co4; p.alpha = 1.f;
  p.k_logical = ks * ks * Cin;
  p.fp8 = 1; p.w_scale = sc; p.x_scale_e8 = 127 - act_shift;
  const AfGemmPlan pl = af_plan_conv_gemm(p, 1, 2);
  if (pl.tile < 4) { af_set_error_msg("af_op_conv2d_fp8: no fp8 plan for M=%d N=%d K=%d", p.M, p.N, p.K); return AF_ERR_INVALID; }
  void* ws = nullptr;
  if (pl.splitk > 1) { ws = tmp.get(pl.ws_bytes, false); if (!ws) return AF_ERR_HIP; }
  OP_TRY(af_launch_conv_gemm<bf16>(p, 1, s, &pl, ws));
  OP_TRY(af_launch_nhwc_to_nchw<bf16>(yn, y_dev, B, Cout, Ho * Wo, co4, s));
  return 0;
}

// GroupNorm (+SiLU) with the e4m3 output the fp8 convolutions read: y8_dev [B][H*W][C] bytes of result * 2^act_shift
int af_op_groupnorm_fp8(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, int silu,
                        unsigned char* y8_dev, int B, int C, int H, int W, int act_shift, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int HW = H * W;
  OP_ALLOC(xn, (size_t)B * HW * C * 2, false);
  OP_ALLOC(ws, af_gn_workspace_bytes(B, HW), false);
  OP_TRY(af_launch_nchw_to_nhwc<bf16>(x_dev, xn, B, C, HW, C, 1.f, s));
  OP_TRY(af_launch_groupnorm<bf16>(xn, (long)HW * C, C, B, HW, C, gamma_dev, beta_dev, eps, silu, y8_dev, (long)HW * C, C, ws, s,
                                   (float)(1 << act_shift)));
  return 0;
}

// LayerNorm with the e4m3 output the fp8 q / k / v projection reads: y8_dev [rows][C] bytes of result * 2^act_shift
int af_op_layernorm_fp8(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, unsigned char* y8_dev,
                        int64_t rows, int C, int act_shift, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  OP_ALLOC(xn, (size_t)rows * C * 2, false);
  OP_TRY(af_launch_cast_f32<bf16>(x_dev, xn, rows * C, s));
  OP_TRY(af_launch_layernorm<bf16>(xn, C, rows, C, gamma_dev, beta_dev, eps, y8_dev, C, s, (float)(1 << act_shift)));
  return 0;
}

int af_op_linear(int dtype, const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                 float* y_dev, int64_t M, int K, int N, int geglu, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int kp = rup(K, bk(dtype));
  const int rows = geglu ? 2 * N : N;  // weight rows
  const int rows_pad = rup(rows, 128);
  const int no4 = rup(N, 4);
  OP_ALLOC(xn, (size_t)M * kp * esz(dtype), false);
  OP_ALLOC(wn, (size_t)rows_pad * kp * esz(dtype), true);
  OP_ALLOC(yn, (size_t)M * no4 * esz(dtype), true);
  float* bn = nullptr;
  void* rn = nullptr;
  // [M,K] rows == NCHW with C=K, HW=1 per "sample": reuse the NCHW converter with B=M, HW=1
  OP_TRY(DISP(dtype, af_launch_nchw_to_nhwc<bf16>(x_dev, xn, (int)M, K, 1, kp, 1.f, s),
              af_launch_nchw_to_nhwc<float>(x_dev, xn, (int)M, K, 1, kp, 1.f, s)));
  OP_TRY(DISP(dtype, af_launch_repack_weight<bf16>(w_dev, wn, rows, K, kp, 1, kp, 0, geglu ? 1 : 0, s),
              af_launch_repack_weight<float>(w_dev, wn, rows, K, kp, 1, kp, 0, geglu ? 1 : 0, s)));
  if (bias_dev) {
    bn = reinterpret_cast<float*>(tmp.get((size_t)rows_pad * 4, true));
    if (!bn) return AF_ERR_HIP;
    OP_TRY(af_launch_permute_bias(bias_dev, bn, rows, geglu ? 1 : 0, s));
  }
  if (residual_dev) {
    rn = tmp.get((size_t)M * no4 * esz(dtype), false);
    if (!rn) return AF_ERR_HIP;
    OP_TRY(DISP(dtype, af_launch_nchw_to_nhwc<bf16>(residual_dev, rn, (int)M, N, 1, no4, 1.f, s),
                af_launch_nchw_to_nhwc<float>(residual_dev, rn, (int)M, N, 1, no4, 1.f, s)));
  }
  ConvGemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = xn; p.src_batch_stride = (long)M * kp; p.ldc = kp; p.Cin = kp;
  p.Hs = 1; p.Ws = (int)M; p.Hi = 1; p.Wi = (int)M; p.Ho = 1; p.Wo = (int)M;
  p.ks = 1; p.stride = 1; p.pad = 0;
  p.W = wn; p.ldw = kp; p.Wrows = rows_pad;
  p.M = (int)M; p.N = geglu ? 2 * N : no4; p.K = kp;
  p.bias = bn; p.residual = rn; p.ldr = no4; p.out = yn; p.ldo = no4;
  p.epilogue = geglu ? AF_EPI_GEGLU : AF_EPI_NONE;
  p.alpha = 1.f;
  p.k_logical = K;
  const AfGemmPlan pl = af_plan_conv_gemm(p, 1, (int)esz(dtype));
  void* ws = nullptr;
  if (pl.splitk > 1) { ws = tmp.get(pl.ws_bytes, false); if (!ws) return AF_ERR_HIP; }
  OP_TRY(DISP(dtype, af_launch_conv_gemm<bf16>(p, 1, s, &pl, ws), af_launch_conv_gemm<float>(p, 1, s, &pl, ws)));
  OP_TRY(DISP(dtype, af_launch_nhwc_to_nchw<bf16>(yn, y_dev, (int)M, N, 1, no4, s),
              af_launch_nhwc_to_nchw<float>(yn, y_dev, (int)M, N, 1, no4, s)));
  return 0;
}

// GroupNorm (no SiLU) followed by a 1x1 convolution -- SpatialTransformer.norm + proj_in (attention.py:325-326) -- both ways
// on the same bf16 operands: y_plain = the apply pass + the GEMM, y_fused = the row-panel GEMM normalising its rows in its
// prologue (ConvGemmParams::gn_ab).  Outputs [B, N, H, W] fp32.  Fails when the shape has no row-panel launch.
int af_op_gn_conv1x1(const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, const float* w_dev,
                     const float* bias_dev, float* y_plain_dev, float* y_fused_dev, int B, int C, int H, int W, int N,
                     void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int HW = H * W, rows_pad = rup(N, 128);
  if (C % 64 || N % 4) { af_set_error_msg("af_op_gn_conv1x1: C %% 64 and N %% 4"); return AF_ERR_INVALID; }
  OP_ALLOC(xn, (size_t)B * HW * C * 2, false);
  OP_ALLOC(gn, (size_t)B * HW * C * 2, false);
  OP_ALLOC(wn, (size_t)rows_pad * C * 2, true);
  OP_ALLOC(y0, (size_t)B * HW * N * 2, true);
  OP_ALLOC(y1, (size_t)B * HW * N * 2, true);
  OP_ALLOC(ws, af_gn_workspace_bytes(B, HW), false);
  OP_ALLOC(ab, (size_t)B * 2 * C * sizeof(float), false);
  float* bn = nullptr;
  if (bias_dev) {
    bn = reinterpret_cast<float*>(tmp.get((size_t)rows_pad * 4, true));
    if (!bn) return AF_ERR_HIP;
    if (hipMemcpyAsync(bn, bias_dev, (size_t)N * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return AF_ERR_HIP;
  }
  OP_TRY(af_launch_nchw_to_nhwc<bf16>(x_dev, xn, B, C, HW, C, 1.f, s));
  OP_TRY(af_launch_repack_weight<bf16>(w_dev, wn, N, C, C, 1, C, 0, 0, s));
  ConvGemmParams p;
  memset(&p, 0, sizeof(p));
  p.src_batch_stride = (long)HW * C; p.ldc = C; p.Cin = C;
  p.Hs = H; p.Ws = W; p.Hi = H; p.Wi = W; p.Ho = H; p.Wo = W;
  p.ks = 1; p.stride = 1; p.pad = 0;
  p.W = wn; p.ldw = C; p.Wrows = rows_pad;
  p.M = B * HW; p.N = N; p.K = C; p.k_logical = C;
  p.bias = bn; p.ldo = N; p.alpha = 1.f;
  // plain: GroupNorm apply pass, then the GEMM as the planner places it
  OP_TRY(af_launch_groupnorm<bf16>(xn, (long)HW * C, C, B, HW, C, gamma_dev, beta_dev, eps, 0, gn, (long)HW * C, C, ws, s));
  {
    ConvGemmParams q = p;
    q.src = gn; q.out = y0;
    const AfGemmPlan pl = af_plan_conv_gemm(q, 1, 2);
    void* wsk = nullptr;
    if (pl.splitk > 1) { wsk = tmp.get(pl.ws_bytes, false); if (!wsk) return AF_ERR_HIP; }
    OP_TRY(af_launch_conv_gemm<bf16>(q, 1, s, &pl, wsk));
  }
  // fused: statistics + fold, the GEMM reads the un-normalised rows
  OP_TRY(af_launch_groupnorm_fold<bf16>(xn, (long)HW * C, C, B, HW, C, gamma_dev, beta_dev, eps, ws, s, nullptr, 0, (float*)ab));
  {
    ConvGemmParams q = p;
    q.src = xn; q.out = y1;
    q.gn_ab = (const float*)ab; q.gn_hw = HW;
    AfGemmPlan pl = af_plan_conv_gemm(q, 1, 2);
    q.splitk = pl.splitk;
    if (!af_conv_rowpanel_kind(q, 1)) { af_set_error_msg("af_op_gn_conv1x1: M=%d K=%d N=%d has no row-panel launch", q.M, q.K, q.N); return AF_ERR_INVALID; }
    q.splitk = 0;
    OP_TRY(af_launch_conv_gemm<bf16>(q, 1, s, &pl, nullptr));
  }
  OP_TRY(af_launch_nhwc_to_nchw<bf16>(y0, y_plain_dev, B, N, HW, N, s));
  OP_TRY(af_launch_nhwc_to_nchw<bf16>(y1, y_fused_dev, B, N, HW, N, s));
  return 0;
}

int af_op_groupnorm(int dtype, const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps, int silu,
                    float* y_dev, int B, int C, int H, int W, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int HW = H * W;
  OP_ALLOC(xn, (size_t)B * HW * C * esz(dtype), false);
  OP_ALLOC(yn, (size_t)B * HW * C * esz(dtype), false);
  OP_ALLOC(ws, af_gn_workspace_bytes(B, HW), false);
  OP_TRY(DISP(dtype, af_launch_nchw_to_nhwc<bf16>(x_dev, xn, B, C, HW, C, 1.f, s),
              af_launch_nchw_to_nhwc<float>(x_dev, xn, B, C, HW, C, 1.f, s)));
  OP_TRY(DISP(dtype,
              af_launch_groupnorm<bf16>(xn, (long)HW * C, C, B, HW, C, gamma_dev, beta_dev, eps, silu, yn, (long)HW * C, C, ws, s),
              af_launch_groupnorm<float>(xn, (long)HW * C, C, B, HW, C, gamma_dev, beta_dev, eps, silu, yn, (long)HW * C, C, ws, s)));
  OP_TRY(DISP(dtype, af_launch_nhwc_to_nchw<bf16>(yn, y_dev, B, C, HW, C, s),
              af_launch_nhwc_to_nchw<float>(yn, y_dev, B, C, HW, C, s)));
  return 0;
}

// 3x3 / stride-1 convolution (bf16) followed by GroupNorm(32) (+SiLU) as a ResBlock runs the pair: the convolution also
// writes the GroupNorm partial sums of its output (ConvGemmParams::gn_stats_out) and the GroupNorm makes no statistics pass.
// h_dev: the convolution's output, y_dev: the GroupNorm's; AF_ERR_INVALID when the shape has no producer plan.
int af_op_conv_gn(const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                  const float* gamma_dev, const float* beta_dev, float eps, int silu, float* h_dev, float* y_dev, int B,
                  int Cin, int H, int W, int Cout, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  if (Cin % 64 != 0 || Cout % 32 != 0) { af_set_error_msg("af_op_conv_gn: Cin%%64==0 and Cout%%32==0"); return AF_ERR_INVALID; }
  const int HW = H * W, rows_pad = rup(Cout, 128), ldw = 9 * Cin;
  OP_ALLOC(xn, (size_t)B * HW * Cin * 2, false);
  OP_ALLOC(wn, (size_t)rows_pad * ldw * 2, true);
  OP_ALLOC(hn, (size_t)B * HW * Cout * 2, true);
  OP_ALLOC(yn, (size_t)B * HW * Cout * 2, false);
  OP_ALLOC(ws, af_gn_workspace_bytes(B, HW), false);
  OP_ALLOC(st, (size_t)B * (HW / 64 + 1) * 32 * 2 * sizeof(float), true);
  void* rn = nullptr;
  float* bn = nullptr;
  OP_TRY(af_launch_nchw_to_nhwc<bf16>(x_dev, xn, B, Cin, HW, Cin, 1.f, s));
  OP_TRY(af_launch_repack_weight<bf16>(w_dev, wn, Cout, Cin, Cin, 3, ldw, 0, 0, s));
  if (bias_dev) {
    bn = reinterpret_cast<float*>(tmp.get((size_t)rows_pad * 4, true));
    if (!bn) return AF_ERR_HIP;
    if (hipMemcpyAsync(bn, bias_dev, (size_t)Cout * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return AF_ERR_HIP;
  }
  if (residual_dev) {
    rn = tmp.get((size_t)B * HW * Cout * 2, false);
    if (!rn) return AF_ERR_HIP;
    OP_TRY(af_launch_nchw_to_nhwc<bf16>(residual_dev, rn, B, Cout, HW, Cout, 1.f, s));
  }
  ConvGemmParams p;
  memset(&p, 0, sizeof(p));
  p.src = xn; p.src_batch_stride = (long)HW * Cin; p.ldc = Cin; p.Cin = Cin;
  p.Hs = H; p.Ws = W; p.Hi = H; p.Wi = W; p.Ho = H; p.Wo = W;
  p.ks = 3; p.stride = 1; p.pad = 1;
  p.W = wn; p.ldw = ldw; p.Wrows = rows_pad;
  p.M = B * HW; p.N = Cout; p.K = ldw;
  p.bias = bn; p.residual = rn; p.ldr = Cout; p.out = hn; p.ldo = Cout; p.alpha = 1.f;
  p.k_logical = 9 * Cin;
  const AfGemmPlan pl = af_plan_conv_gemm(p, 1, 2);
  if (!af_conv_gn_stats_ok(p, pl, Cout / 32)) {
    af_set_error_msg("af_op_conv_gn: no GroupNorm-statistics producer plan for M=%d N=%d K=%d", p.M, p.N, p.K);
    return AF_ERR_INVALID;
  }
  p.gn_stats_out = reinterpret_cast<float*>(st);
  p.gn_cpg = Cout / 32;
  OP_TRY(af_launch_conv_gemm<bf16>(p, 1, s, &pl, nullptr));
  OP_TRY(af_launch_groupnorm<bf16>(hn, (long)HW * Cout, Cout, B, HW, Cout, gamma_dev, beta_dev, eps, silu, yn, (long)HW * Cout,
                                   Cout, ws, s, 0.f, reinterpret_cast<const float*>(st), HW / 64));
  OP_TRY(af_launch_nhwc_to_nchw<bf16>(hn, h_dev, B, Cout, HW, Cout, s));
  OP_TRY(af_launch_nhwc_to_nchw<bf16>(yn, y_dev, B, Cout, HW, Cout, s));
  return 0;
}

int af_op_layernorm(int dtype, const float* x_dev, const float* gamma_dev, const float* beta_dev, float eps,
                    float* y_dev, int64_t rows, int C, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  OP_ALLOC(xn, (size_t)rows * C * esz(dtype), false);
  OP_ALLOC(yn, (size_t)rows * C * esz(dtype), false);
  OP_TRY(DISP(dtype, af_launch_cast_f32<bf16>(x_dev, xn, rows * C, s), af_launch_cast_f32<float>(x_dev, xn, rows * C, s)));
  OP_TRY(DISP(dtype, af_launch_layernorm<bf16>(xn, C, rows, C, gamma_dev, beta_dev, eps, yn, C, s),
              af_launch_layernorm<float>(xn, C, rows, C, gamma_dev, beta_dev, eps, yn, C, s)));
  OP_TRY(DISP(dtype, af_launch_cast_to_f32<bf16>(yn, y_dev, rows * C, s), af_launch_cast_to_f32<float>(yn, y_dev, rows * C, s)));
  return 0;
}

int af_op_attention(int dtype, const float* q_dev, const float* k_dev, const float* v_dev, float* o_dev, int B, int Nq,
                    int Nk, int heads, int dh, float scale, int causal, void* stream) {   // causal = flags: bit 0 causal, bit 1 NaN guard rows
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  const int C = heads * dh;
  const long nq = (long)B * Nq * C, nk = (long)B * Nk * C;
  OP_ALLOC(qn, (size_t)nq * esz(dtype), false);
  // flags bit 1 (test hook): K and V are the heads of larger allocations whose 128 tail rows hold NaNs, so a kernel that
  // reads rows >= Nk of the last sample's last key tile (instead of having them zero-filled) shows up as NaN in O
  const bool guard = (causal & 2) != 0;
  causal &= 1;
  const size_t tail = guard ? (size_t)128 * C * esz(dtype) : 0;
  OP_ALLOC(kn, (size_t)nk * esz(dtype) + tail, false);
  OP_ALLOC(vn, (size_t)nk * esz(dtype) + tail, false);
  if (guard) {
    if (hipMemsetAsync((char*)kn + (size_t)nk * esz(dtype), 0xFF, tail, s) != hipSuccess ||   // 0xFFFF / 0xFFFFFFFF: NaN
        hipMemsetAsync((char*)vn + (size_t)nk * esz(dtype), 0xFF, tail, s) != hipSuccess) return AF_ERR_HIP;
  }
  OP_ALLOC(on, (size_t)nq * esz(dtype), true);
  OP_TRY(DISP(dtype, af_launch_cast_f32<bf16>(q_dev, qn, nq, s), af_launch_cast_f32<float>(q_dev, qn, nq, s)));
  OP_TRY(DISP(dtype, af_launch_cast_f32<bf16>(k_dev, kn, nk, s), af_launch_cast_f32<float>(k_dev, kn, nk, s)));
  OP_TRY(DISP(dtype, af_launch_cast_f32<bf16>(v_dev, vn, nk, s), af_launch_cast_f32<float>(v_dev, vn, nk, s)));
  AttnParams p;
  p.q = qn; p.k = kn; p.v = vn; p.o = on; p.lse = nullptr; p.causal = causal;
  p.ldq = p.ldk = p.ldv = p.ldo = C;
  p.bsq = (long)Nq * C; p.bso = (long)Nq * C; p.bsk = (long)Nk * C; p.bsv = (long)Nk * C;
  p.Nq = Nq; p.Nk = Nk; p.H = heads; p.scale = scale;
  p.vt_pack = nullptr;
  if (dtype == AF_DTYPE_BF16 && !causal) {   // as af_set_context does for the cached cross-attention K / V
    const long pe = af_attn_short_pack_elems<bf16>(B, heads, dh, Nk);
    if (pe > 0) {
      OP_ALLOC(vt, (size_t)pe * 2, false);
      OP_TRY(af_launch_attn_short_pack<bf16>(vn, C, (long)Nk * C, Nk, heads, dh, B, vt, s));
      p.vt_pack = vt;
    }
  }
  OP_TRY(DISP(dtype, af_launch_attention<bf16>(p, B, dh, s), af_launch_attention<float>(p, B, dh, s)));
  OP_TRY(DISP(dtype, af_launch_cast_to_f32<bf16>(on, o_dev, nq, s), af_launch_cast_to_f32<float>(on, o_dev, nq, s)));
  return 0;
}

// One cross-attention layer through xattn_fused_kernel (bf16), prepared as the model prepares it: LayerNorm folded into to_q
// (af_launch_ln_fold), K / V packed per head pair, to_out's K dimension permuted.  ln_parts_out_dev (optional): [4][B * N][2]
// partial sums of the stored rows, as the next LayerNorm's consumer reads them.
int af_op_xattn_fused(const float* x_dev, const float* ln_stats_dev, const float* gamma_dev, const float* beta_dev,
                      const float* wq_dev, const float* kv_dev, const float* wo_dev, const float* bo_dev, float* y_dev,
                      float* ln_parts_out_dev, int B, int N, int S, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  constexpr int C = 320, Hh = 8, dh = 40;
  const int M = B * N, rows_pad = rup(C, 128);
  if (!af_xattn_fused_ok(M, N, C, Hh, dh, S)) { af_set_error_msg("af_op_xattn_fused: B=%d N=%d S=%d has no fused kernel", B, N, S); return AF_ERR_INVALID; }
  OP_ALLOC(xn, (size_t)M * C * 2, false);
  OP_ALLOC(yn, (size_t)M * C * 2, true);
  OP_ALLOC(wq, (size_t)rows_pad * C * 2, true);
  OP_ALLOC(wqf, (size_t)rows_pad * C * 2, true);
  OP_ALLOC(wo, (size_t)rows_pad * C * 2, true);
  OP_ALLOC(wop, (size_t)rows_pad * C * 2, true);
  OP_ALLOC(cs, (size_t)rows_pad * 4, true);
  OP_ALLOC(bf, (size_t)rows_pad * 4, true);
  OP_ALLOC(kvn, (size_t)B * S * 2 * C * 2, false);
  const long pe = af_xattn_fused_pack_elems(B, Hh, dh, S);
  OP_ALLOC(pack, (size_t)pe * 2, false);
  OP_ALLOC(parts, (size_t)4 * M * 2 * sizeof(float), true);
  OP_TRY(af_launch_cast_f32<bf16>(x_dev, xn, (long)M * C, s));
  OP_TRY(af_launch_cast_f32<bf16>(kv_dev, kvn, (long)B * S * 2 * C, s));
  OP_TRY(af_launch_repack_weight<bf16>(wq_dev, wq, C, C, C, 1, C, 0, 0, s));
  OP_TRY(af_launch_repack_weight<bf16>(wo_dev, wo, C, C, C, 1, C, 0, 0, s));
  OP_TRY(af_launch_ln_fold<bf16>(wq, wqf, gamma_dev, beta_dev, nullptr, (float*)cs, (float*)bf, rows_pad, C, C, s));
  OP_TRY(af_launch_xattn_fused_permute_wo(wo, C, C, wop, C, s));
  OP_TRY(af_launch_xattn_fused_pack(kvn, 2 * C, (long)S * 2 * C, S, B, pack, s));
  AfXattnFusedParams a;
  memset(&a, 0, sizeof(a));
  a.x = xn; a.ldx = C; a.M = M; a.rows_per_sample = N;
  a.ln_stats = ln_stats_dev; a.ln_parts_n = 0;
  a.wq = wqf; a.ldwq = C; a.q_colsum = (const float*)cs; a.q_bias = (const float*)bf;
  a.kvpack = pack;
  a.wo = wop; a.ldwo = C; a.o_bias = bo_dev;
  a.out = yn; a.ldo = C;
  a.ln_stats_out = (float*)parts;
  a.Nk = S;
  OP_TRY(af_launch_xattn_fused(a, s));
  OP_TRY(af_launch_cast_to_f32<bf16>(yn, y_dev, (long)M * C, s));
  if (ln_parts_out_dev && hipMemcpyAsync(ln_parts_out_dev, parts, (size_t)4 * M * 2 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return AF_ERR_HIP;
  return 0;
}

int af_op_timestep_embedding(int dtype, const int64_t* t_dev, float* y_dev, int B, int dim, void* stream) {
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tmp tmp;
  OP_ALLOC(yn, (size_t)B * dim * esz(dtype), false);
  OP_TRY(DISP(dtype, af_launch_timestep_embedding<bf16>((const long long*)t_dev, yn, B, dim, s),
              af_launch_timestep_embedding<float>((const long long*)t_dev, yn, B, dim, s)));
  OP_TRY(DISP(dtype, af_launch_cast_to_f32<bf16>(yn, y_dev, (long)B * dim, s),
              af_launch_cast_to_f32<float>(yn, y_dev, (long)B * dim, s)));
  return 0;
}

}  // extern "C"
