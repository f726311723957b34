// GroupNorm(32)(+SiLU) and LayerNorm kernels for gfx950, NHWC activations.
//
// Replaces GroupNorm32 / normalization (ldm/modules/diffusionmodules/util.py:202-219,
// eps 1e-5, fp32 math), Normalize (attention.py:71-72 and model.py:39-40, eps 1e-6),
// the nn.SiLU / nonlinearity that always follows them in ResBlock / ResnetBlock
// (openaimodel.py:205-207,229-231; model.py:34-36) and nn.LayerNorm
// (attention.py:267-269).  All are HBM-bound: every access is a 16-byte vector,
// pixel-major so a wave reads contiguous memory; statistics are fp32 partial sums
// combined in fp64 (biased variance, as torch).
//
//   gn_stats    grid (chunks, B): per-(sample, pixel-chunk, group) sum / sumsq
//   gn_finalize grid (B)        : fp64 combine -> mean, rstd per (sample, group)
//   gn_apply    grid (chunks, B): y = (x-mean)*rstd*gamma+beta, optional SiLU
//   layernorm   one wave per row, row held in registers, two-pass variance
#include "af_common.h"
#include <cstdlib>

#define GN_GROUPS 32
#define GN_MAX_C 2560

// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, long batch_stride, int ldc,
                                                        int HW, int Cn, int P, float* __restrict__ partial,
                                                        int nchunk) {
  constexpr int EPC = 16 / sizeof(T);
  __shared__ float s_sum[GN_MAX_C];
  __shared__ float s_sq[GN_MAX_C];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int NV = Cn / EPC;  // 16-byte vectors per pixel
  for (int c = tid; c < Cn; c += 256) { s_sum[c] = 0.f; s_sq[c] = 0.f; }
  __syncthreads();

  const int p0 = chunk * P;
  const int p1 = min(HW, p0 + P);
  const T* xb = x + (long)b * batch_stride;

  if (NV <= 256) {
    // PL pixel lanes x NV vector columns; per-lane partials go to LDS slots and are summed in a
    // fixed order (no atomics: results are bitwise reproducible run to run)
    const int PL = 256 / NV;
    const int v = tid % NV, pl = tid / NV;
    float a[EPC], q[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { a[e] = 0.f; q[e] = 0.f; }
    if (pl < PL) {
#pragma unroll 4
      for (int pix = p0 + pl; pix < p1; pix += PL) {
        Vec16<T> vv;
        vv.u = *reinterpret_cast<const uint4*>(xb + (long)pix * ldc + v * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float f = to_f32<T>(vv.e[e]);
          a[e] += f;
          q[e] += f * f;
        }
      }
    }
    for (int lane_p = 0; lane_p < PL; ++lane_p) {
      if (pl == lane_p) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          s_sum[v * EPC + e] += a[e];
          s_sq[v * EPC + e] += q[e];
        }
      }
      __syncthreads();
    }
  } else {
    for (int v = tid; v < NV; v += 256) {
      float a[EPC], q[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) { a[e] = 0.f; q[e] = 0.f; }
      for (int pix = p0; pix < p1; ++pix) {
        Vec16<T> vv;
        vv.u = *reinterpret_cast<const uint4*>(xb + (long)pix * ldc + v * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          float f = to_f32<T>(vv.e[e]);
          a[e] += f;
          q[e] += f * f;
        }
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        s_sum[v * EPC + e] = a[e];
        s_sq[v * EPC + e] = q[e];
      }
    }
  }
  __syncthreads();
  if (tid < GN_GROUPS) {
    const int cpg = Cn / GN_GROUPS;
    float a = 0.f, q = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += s_sum[c]; q += s_sq[c]; }
    float* dst = partial + (((long)b * nchunk + chunk) * GN_GROUPS + tid) * 2;
    dst[0] = a;
    dst[1] = q;
  }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk,
                                                           double count, float eps,
                                                           float* __restrict__ stats) {
  // 256 threads = 32 groups x 8 slices over the chunks
  __shared__ double s_a[256], s_q[256];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int g = tid & 31, sl = tid >> 5;
  double a = 0.0, q = 0.0;
  // (eight loads in flight, added in the order of the plain loop: the VAE decoder's 512 x 512 maps bring > 1000 chunks per
  // sample and one load latency per chunk made this 27 us per launch)
  int c = sl;
  for (; c + 56 < nchunk; c += 64) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float2*>(partial + (((long)b * nchunk + c + 8 * u) * GN_GROUPS + g) * 2);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += (double)v[u].x; q += (double)v[u].y; }
  }
  for (; c < nchunk; c += 8) {
    const float* src = partial + (((long)b * nchunk + c) * GN_GROUPS + g) * 2;
    a += (double)src[0];
    q += (double)src[1];
  }
  s_a[tid] = a;
  s_q[tid] = q;
  __syncthreads();
  if (tid < 32) {
    for (int s = 1; s < 8; ++s) { a += s_a[tid + 32 * s]; q += s_q[tid + 32 * s]; }
    double mean = a / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((long)b * GN_GROUPS + tid) * 2 + 0] = (float)mean;
    stats[((long)b * GN_GROUPS + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// fp8 output (the consumer is a convolution on the block-scaled fp8 MFMA): OCP e4m3 bytes of value * mul, saturating
__device__ __forceinline__ unsigned gn_pack4_e4m3(float a, float b, float c, float d, float mul) {
  a = __builtin_amdgcn_fmed3f(a * mul, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b * mul, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c * mul, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d * mul, -448.f, 448.f);
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, long batch_stride, int ldc,
                                                        int HW, int Cn, int P,
                                                        const float* __restrict__ stats,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int silu,
                                                        T* __restrict__ y, long y_batch_stride, int ldy,
                                                        const float* __restrict__ partial, int nchunk, double count,
                                                        float eps, float fp8_mul, int npart) {
  constexpr int EPC = 16 / sizeof(T);
  __shared__ __attribute__((aligned(16))) float s_a[GN_MAX_C];
  __shared__ __attribute__((aligned(16))) float s_b[GN_MAX_C];
  __shared__ double s_ra[256], s_rq[256];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int cpg = Cn / GN_GROUPS;
  if (partial) {
    // finalize folded in (few chunks): every block combines its sample's partial sums itself, in the same fixed
    // order as gn_finalize_kernel, so the statistics are bitwise identical across blocks and launches
    // (npart: partial sums per sample -- the chunks of gn_stats_kernel, or the 64-row slabs of a producer convolution)
    const int g = tid & 31, sl = tid >> 5;
    double a = 0.0, q = 0.0;
    for (int c = sl; c < npart; c += 8) {
      const float* src = partial + (((long)b * npart + c) * GN_GROUPS + g) * 2;
      a += (double)src[0];
      q += (double)src[1];
    }
    s_ra[tid] = a;
    s_rq[tid] = q;
    __syncthreads();
    if (tid < 32) {
      for (int s2 = 1; s2 < 8; ++s2) { a += s_ra[tid + 32 * s2]; q += s_rq[tid + 32 * s2]; }
      const double mean = a / count;
      double var = q / count - mean * mean;
      if (var < 0.0) var = 0.0;
      s_mean[tid] = (float)mean;
      s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  for (int c = tid; c < Cn; c += 256) {
    const int g = c / cpg;
    const float mean = partial ? s_mean[g] : stats[((long)b * GN_GROUPS + g) * 2 + 0];
    const float rstd = partial ? s_rstd[g] : stats[((long)b * GN_GROUPS + g) * 2 + 1];
    const float a = gamma[c] * rstd;
    s_a[c] = a;
    s_b[c] = beta[c] - mean * a;
  }
  __syncthreads();
  // (pixel, 16-byte vector) items in flat order, 256 apart per thread: the item -> (pixel, vector) split is carried
  // incrementally (one division per thread, not per item); the folded scale / shift of a vector's channels come from
  // LDS as 16-byte reads; SiLU = f * rcp(1 + exp2(-f log2 e)) on the fast transcendental units.  Before this the
  // kernel was VALU-bound at ~2 TB/s (a division, 16 scalar LDS reads and a full-precision expf + divide per item).
  const int NV = Cn / EPC;
  const int p0 = chunk * P;
  const int p1 = min(HW, p0 + P);
  const int nitems = (p1 - p0) * NV;
  const T* xb = x + (long)b * batch_stride;
  T* yb = y + (long)b * y_batch_stride;
  const int dpix = 256 / NV, dv = 256 - dpix * NV;
  int pix = p0 + tid / NV, v = tid - (tid / NV) * NV;
  // UN items per thread in flight: with one load per thread outstanding (round 2) the pass ran at the memory latency,
  // 2.7-3.3 TB/s at 64x64; the loads of a group are all issued before the first of them is consumed
  constexpr int UN = 4;
  for (int it = tid; it < nitems; it += 256 * UN) {
    Vec16<T> vv[UN];
    int pixs[UN], vs[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      pixs[u] = pix;
      vs[u] = v;
      vv[u].u = make_uint4(0, 0, 0, 0);
      if (it + 256 * u < nitems) vv[u].u = *reinterpret_cast<const uint4*>(xb + (long)pix * ldc + v * EPC);
      pix += dpix;
      v += dv;
      if (v >= NV) { v -= NV; ++pix; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (it + 256 * u >= nitems) break;
      Vec16<T> oo;
      float ca[EPC], cb[EPC];
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q) {
        *reinterpret_cast<float4*>(ca + 4 * q) = *reinterpret_cast<const float4*>(s_a + vs[u] * EPC + 4 * q);
        *reinterpret_cast<float4*>(cb + 4 * q) = *reinterpret_cast<const float4*>(s_b + vs[u] * EPC + 4 * q);
      }
      float ff[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float f = fmaf(to_f32<T>(vv[u].e[e]), ca[e], cb[e]);
        if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * f));
        ff[e] = f;
        oo.e[e] = from_f32<T>(f);
      }
      if (fp8_mul != 0.f) {   // (y_batch_stride, ldy in bytes)
        unsigned char* y8 = reinterpret_cast<unsigned char*>(y) + (long)b * y_batch_stride + (long)pixs[u] * ldy + vs[u] * EPC;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<unsigned*>(y8 + e) = gn_pack4_e4m3(ff[e], ff[e + 1], ff[e + 2], ff[e + 3], fp8_mul);
      } else {
        *reinterpret_cast<uint4*>(yb + (long)pixs[u] * ldy + vs[u] * EPC) = oo.u;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// LayerNorm: one wave per row; C*sizeof(T)/16 <= 64*LN_MAXV vectors.
// ---------------------------------------------------------------------------
#define LN_MAXV 5
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, int ldx, long rows, int Cn,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         T* __restrict__ y, int ldy, float fp8_mul) {
  constexpr int EPC = 16 / sizeof(T);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int NV = Cn / EPC;
  Vec16<T> v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < NV) {
      v[i].u = *reinterpret_cast<const uint4*>(x + row * ldx + vi * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += to_f32<T>(v[i].e[e]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)Cn;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < NV) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float d = to_f32<T>(v[i].e[e]) - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)Cn + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int vi = lane + 64 * i;
    if (vi < NV) {
      Vec16<T> o;
      float ff[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = vi * EPC + e;
        ff[e] = (to_f32<T>(v[i].e[e]) - mean) * rstd * gamma[c] + beta[c];
        o.e[e] = from_f32<T>(ff[e]);
      }
      if (fp8_mul != 0.f) {   // e4m3 output (ldy in bytes)
        unsigned char* y8 = reinterpret_cast<unsigned char*>(y) + row * ldy + vi * EPC;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<unsigned*>(y8 + e) = gn_pack4_e4m3(ff[e], ff[e + 1], ff[e + 2], ff[e + 3], fp8_mul);
      } else {
        *reinterpret_cast<uint4*>(y + row * ldy + vi * EPC) = o.u;
      }
    }
  }
}

// Small feature maps (16x16, 8x8): one workgroup per (sample, group) holds the group's HW x C/32 elements in
// registers, so GroupNorm (+SiLU) is ONE launch that reads the data once instead of stats + apply (two launches of
// ~7-9 us each, both launch-latency bound at these sizes).  Sums in fp32 per lane, wave shuffles, the four wave
// partials combined in fp64 in a fixed order (bitwise reproducible).
#define GNS_MAXV 10
template <typename T>
__global__ __launch_bounds__(256) void gn_small_kernel(const T* __restrict__ x, long batch_stride, int ldc, int HW,
                                                        int Cn, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, int silu,
                                                        T* __restrict__ y, long y_batch_stride, int ldy, float fp8_mul) {
  constexpr int EPC = 16 / sizeof(T);
  __shared__ float s_ra[4], s_rq[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int cpg = Cn / GN_GROUPS;
  const int VP = cpg / EPC;                 // 16-byte vectors per pixel in this group
  const int nitems = HW * VP;
  const T* xb = x + (long)b * batch_stride + (long)g * cpg;
  T* yb = y + (long)b * y_batch_stride + (long)g * cpg;
  Vec16<T> v[GNS_MAXV];
  float a = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < GNS_MAXV; ++i) {
    const int it = tid + i * 256;
    if (it < nitems) {
      const int pix = it / VP, vv = it - pix * VP;
      v[i].u = *reinterpret_cast<const uint4*>(xb + (long)pix * ldc + vv * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float f = to_f32<T>(v[i].e[e]);
        a += f;
        q += f * f;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
  if (lane == 0) { s_ra[wave] = a; s_rq[wave] = q; }
  __syncthreads();
  const double count = (double)HW * (double)cpg;
  const double sa = (double)s_ra[0] + (double)s_ra[1] + (double)s_ra[2] + (double)s_ra[3];
  const double sq = (double)s_rq[0] + (double)s_rq[1] + (double)s_rq[2] + (double)s_rq[3];
  const double mean_d = sa / count;
  double var = sq / count - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
#pragma unroll
  for (int i = 0; i < GNS_MAXV; ++i) {
    const int it = tid + i * 256;
    if (it < nitems) {
      const int pix = it / VP, vv = it - pix * VP;
      const int c0 = g * cpg + vv * EPC;
      Vec16<T> o;
      float ff[EPC];
#pragma unroll
      for (int t = 0; t < EPC / 4; ++t) {
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c0 + 4 * t);
        const float4 bt = *reinterpret_cast<const float4*>(beta + c0 + 4 * t);
        const float* gp = reinterpret_cast<const float*>(&gm);
        const float* bp = reinterpret_cast<const float*>(&bt);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = gp[e] * rstd;
          float f = fmaf(to_f32<T>(v[i].e[4 * t + e]), sc, bp[e] - mean * sc);
          if (silu) f = f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * f));
          ff[4 * t + e] = f;
          o.e[4 * t + e] = from_f32<T>(f);
        }
      }
      if (fp8_mul != 0.f) {   // (y_batch_stride, ldy in bytes)
        unsigned char* y8 = reinterpret_cast<unsigned char*>(y) + (long)b * y_batch_stride + (long)g * cpg + (long)pix * ldy + vv * EPC;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<unsigned*>(y8 + e) = gn_pack4_e4m3(ff[e], ff[e + 1], ff[e + 2], ff[e + 3], fp8_mul);
      } else {
        *reinterpret_cast<uint4*>(yb + (long)pix * ldy + vv * EPC) = o.u;
      }
    }
  }
}

// Row-group LayerNorm: RL lanes share a row (RL = 8 for bf16, 16 for f32), lane j holding vectors j, j+RL, ... of the
// row in registers, so all 64 lanes are busy at C = 320 (the wave-per-row kernel above leaves 24 of 64 idle there and
// fetches gamma / beta with 16 scalar loads per vector; here they sit in LDS and are read as 16-byte vectors).
// One load instruction covers 64/RL rows x 128 contiguous bytes.  Same two-pass variance in fp32.
#define LNG_MAXV 10
template <typename T, int RL>
__global__ __launch_bounds__(256) void layernorm_rowgroup_kernel(const T* __restrict__ x, int ldx, long rows, int Cn,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps,
                                                                  T* __restrict__ y, int ldy, float fp8_mul) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int RPB = 256 / RL;   // rows per block
  __shared__ __attribute__((aligned(16))) float s_g[1280 * 2];
  float* s_b = s_g + Cn;
  for (int c = threadIdx.x; c < Cn; c += 256) { s_g[c] = gamma[c]; s_b[c] = beta[c]; }
  __syncthreads();
  const int j = threadIdx.x % RL;
  const long row = (long)blockIdx.x * RPB + threadIdx.x / RL;
  const bool live = row < rows;   // (no early return: the shuffles below want whole waves)
  const int nv = Cn / EPC / RL;   // vectors per lane
  Vec16<T> v[LNG_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LNG_MAXV; ++i)
    if (i < nv) {
      v[i].u = live ? *reinterpret_cast<const uint4*>(x + row * ldx + (j + i * RL) * EPC) : uint4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < EPC; ++e) s += to_f32<T>(v[i].e[e]);
    }
#pragma unroll
  for (int o = RL / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)Cn;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LNG_MAXV; ++i)
    if (i < nv) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float d = to_f32<T>(v[i].e[e]) - mean;
        q += d * d;
      }
    }
#pragma unroll
  for (int o = RL / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)Cn + eps);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < LNG_MAXV; ++i)
    if (i < nv) {
      const int c0 = (j + i * RL) * EPC;
      float g[EPC], bb[EPC];
#pragma unroll
      for (int t = 0; t < EPC / 4; ++t) {
        *reinterpret_cast<float4*>(g + 4 * t) = *reinterpret_cast<const float4*>(s_g + c0 + 4 * t);
        *reinterpret_cast<float4*>(bb + 4 * t) = *reinterpret_cast<const float4*>(s_b + c0 + 4 * t);
      }
      Vec16<T> o;
      float ff[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        ff[e] = (to_f32<T>(v[i].e[e]) - mean) * rstd * g[e] + bb[e];
        o.e[e] = from_f32<T>(ff[e]);
      }
      if (fp8_mul != 0.f) {   // e4m3 output (ldy in bytes)
        unsigned char* y8 = reinterpret_cast<unsigned char*>(y) + row * ldy + c0;
#pragma unroll
        for (int e = 0; e < EPC; e += 4) *reinterpret_cast<unsigned*>(y8 + e) = gn_pack4_e4m3(ff[e], ff[e + 1], ff[e + 2], ff[e + 3], fp8_mul);
      } else {
        *reinterpret_cast<uint4*>(y + row * ldy + c0) = o.u;
      }
    }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
// pixels per block: aim for ~1024 blocks over (chunks x samples) so that 8x8 / 16x16 feature maps still fill the
// chip, at least 4 pixels per block, at most 1024 chunks per sample (the finalize kernel loops over them)
int af_gn_chunking(int HW, int B, int* P_out) {
  int P = 4;
  while (P < 64 && (long)((HW + P - 1) / P) * B > 1024) P *= 2;
  while ((HW + P - 1) / P > 1024) P *= 2;
  *P_out = P;
  return (HW + P - 1) / P;
}

// workspace: partial [B][nchunk][32][2] floats + stats [B][32][2] floats
size_t af_gn_workspace_bytes(int B, int HW) {
  int P;
  int nchunk = af_gn_chunking(HW, B, &P);
  return ((size_t)B * nchunk * GN_GROUPS * 2 + (size_t)B * GN_GROUPS * 2) * sizeof(float);
}

template <typename T>
int af_launch_groupnorm(const void* x, long x_bs, int ldx, int B, int HW, int Cn, const float* gamma,
                        const float* beta, float eps, int silu, void* y, long y_bs, int ldy, void* workspace,
                        hipStream_t stream, float fp8_mul, const float* pre_partial, int pre_npart) {
  constexpr int EPC = 16 / sizeof(T);
  if (fp8_mul != 0.f && sizeof(T) != 2) { af_set_error_msg("groupnorm: fp8 output needs the bf16 storage mode"); return -1; }
  if (Cn % GN_GROUPS != 0 || Cn % EPC != 0 || Cn > GN_MAX_C || ldx % EPC != 0 || ldy % EPC != 0) {
    af_set_error_msg("groupnorm: unsupported C=%d (need C%%32==0, C%%%d==0, C<=%d)", Cn, EPC, GN_MAX_C);
    return -1;
  }
  AfProfScope prof(AF_K_GROUPNORM, stream, 0.0, 2.0 * B * HW * (double)Cn * sizeof(T));
  {
    // small maps: single-launch register-resident kernel (group channels must be whole 16-byte vectors)
    const int cpg = Cn / GN_GROUPS;
    const bool small_ok = g_af_knobs.gn_small != 0;
    if (small_ok && cpg % EPC == 0 && (long)HW * (cpg / EPC) <= 256 * GNS_MAXV && Cn % 4 == 0) {
      hipLaunchKernelGGL((gn_small_kernel<T>), dim3(GN_GROUPS, B), dim3(256), 0, stream, reinterpret_cast<const T*>(x),
                         x_bs, ldx, HW, Cn, gamma, beta, eps, silu, reinterpret_cast<T*>(y), y_bs, ldy, fp8_mul);
      HIP_CHECK_RET(hipGetLastError());
      return 0;
    }
  }
  int P;
  const int nchunk = af_gn_chunking(HW, B, &P);
  const float* partial = reinterpret_cast<float*>(workspace);
  float* stats = reinterpret_cast<float*>(workspace) + (size_t)B * nchunk * GN_GROUPS * 2;
  int npart = nchunk;
  if (pre_partial) {
    // the producer convolution wrote [B][pre_npart][32][2] partial sums of the values it stored (ConvGemmParams::gn_stats_out):
    // no statistics pass over x at all
    partial = pre_partial;
    npart = pre_npart;
  } else {
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nchunk, B), dim3(256), 0, stream,
                       reinterpret_cast<const T*>(x), x_bs, ldx, HW, Cn, P, reinterpret_cast<float*>(workspace), nchunk);
  }
  const double count = (double)HW * (double)(Cn / GN_GROUPS);
  const bool fold = npart <= 64;  // few chunks: the apply blocks finalize the statistics themselves
  if (!fold) hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, stream, partial, npart, count, eps, stats);
  hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nchunk, B), dim3(256), 0, stream,
                     reinterpret_cast<const T*>(x), x_bs, ldx, HW, Cn, P, stats, gamma, beta, silu,
                     reinterpret_cast<T*>(y), y_bs, ldy, fold ? partial : nullptr, nchunk, count, eps, fp8_mul, npart);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// GroupNorm as a per-sample affine map for a CONSUMER that applies it itself (the row-panel proj_in GEMM of a
// SpatialTransformer normalises its activation rows in registers: ConvGemmParams::gn_ab): ab[b][0][c] = gamma[c] * rstd,
// ab[b][1][c] = beta[c] - mean * gamma[c] * rstd, from the same partial sums, combined in the same order and precision, as
// gn_apply_kernel's own fold -- y = fma(x, a, b) rounded to bf16 is then bit for bit what the apply pass would have stored.
__global__ __launch_bounds__(256) void gn_fold_kernel(const float* __restrict__ partial, int npart, double count, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int Cn, float* __restrict__ ab) {
  __shared__ double s_ra[256], s_rq[256];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int g = tid & 31, sl = tid >> 5;
  double a = 0.0, q = 0.0;
  for (int c = sl; c < npart; c += 8) {
    const float* src = partial + (((long)b * npart + c) * GN_GROUPS + g) * 2;
    a += (double)src[0];
    q += (double)src[1];
  }
  s_ra[tid] = a;
  s_rq[tid] = q;
  __syncthreads();
  if (tid < 32) {
    for (int s2 = 1; s2 < 8; ++s2) { a += s_ra[tid + 32 * s2]; q += s_rq[tid + 32 * s2]; }
    const double mean = a / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int cpg = Cn / GN_GROUPS;
  for (int c = tid; c < Cn; c += 256) {
    const int gg = c / cpg;
    const float sc = gamma[c] * s_rstd[gg];
    ab[((long)b * 2 + 0) * Cn + c] = sc;
    ab[((long)b * 2 + 1) * Cn + c] = beta[c] - s_mean[gg] * sc;
  }
}

template <typename T>
int af_launch_groupnorm_fold(const void* x, long x_bs, int ldx, int B, int HW, int Cn, const float* gamma, const float* beta,
                             float eps, void* workspace, hipStream_t stream, const float* pre_partial, int pre_npart,
                             float* ab_out) {
  constexpr int EPC = 16 / sizeof(T);
  if (Cn % GN_GROUPS != 0 || Cn % EPC != 0 || Cn > GN_MAX_C || ldx % EPC != 0) {
    af_set_error_msg("groupnorm fold: unsupported C=%d", Cn);
    return -1;
  }
  int P;
  const int nchunk = af_gn_chunking(HW, B, &P);
  const float* partial = reinterpret_cast<float*>(workspace);
  int npart = nchunk;
  if (pre_partial) {
    partial = pre_partial;
    npart = pre_npart;
  } else {
    AfProfScope prof(AF_K_GROUPNORM, stream, 0.0, 1.0 * B * HW * (double)Cn * sizeof(T));
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nchunk, B), dim3(256), 0, stream,
                       reinterpret_cast<const T*>(x), x_bs, ldx, HW, Cn, P, reinterpret_cast<float*>(workspace), nchunk);
  }
  hipLaunchKernelGGL(gn_fold_kernel, dim3(B), dim3(256), 0, stream, partial, npart, (double)HW * (double)(Cn / GN_GROUPS), eps,
                     gamma, beta, Cn, ab_out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <typename T>
int af_launch_layernorm(const void* x, int ldx, long rows, int Cn, const float* gamma, const float* beta,
                        float eps, void* y, int ldy, hipStream_t stream, float fp8_mul) {
  constexpr int EPC = 16 / sizeof(T);
  if (fp8_mul != 0.f && sizeof(T) != 2) { af_set_error_msg("layernorm: fp8 output needs the bf16 storage mode"); return -1; }
  if (Cn % EPC != 0 || Cn / EPC > 64 * LN_MAXV || ldx % EPC != 0 || ldy % EPC != 0) {
    af_set_error_msg("layernorm: unsupported C=%d", Cn);
    return -1;
  }
  if (rows <= 0) return 0;
  AfProfScope prof(AF_K_LAYERNORM, stream, 0.0, 2.0 * rows * (double)Cn * sizeof(T));
  constexpr int RL = sizeof(T) == 2 ? 8 : 16;
  const int NV = Cn / EPC;
  constexpr int RPB = 256 / RL;
  // (few rows of many vectors -- [4096, 1280] -- fill the chip better with one wave per row: measured 10 vs 19 us)
  if (NV % RL == 0 && NV / RL <= 10 && Cn <= 1280 && Cn % 4 == 0 && rows / RPB >= 256) {
    hipLaunchKernelGGL((layernorm_rowgroup_kernel<T, RL>), dim3((unsigned)((rows + RPB - 1) / RPB)), dim3(256), 0, stream,
                       reinterpret_cast<const T*>(x), ldx, rows, Cn, gamma, beta, eps, reinterpret_cast<T*>(y), ldy, fp8_mul);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL((layernorm_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream,
                     reinterpret_cast<const T*>(x), ldx, rows, Cn, gamma, beta, eps,
                     reinterpret_cast<T*>(y), ldy, fp8_mul);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template int af_launch_groupnorm<bf16>(const void*, long, int, int, int, int, const float*, const float*, float,
                                       int, void*, long, int, void*, hipStream_t, float, const float*, int);
template int af_launch_groupnorm<float>(const void*, long, int, int, int, int, const float*, const float*, float,
                                        int, void*, long, int, void*, hipStream_t, float, const float*, int);
template int af_launch_groupnorm_fold<bf16>(const void*, long, int, int, int, int, const float*, const float*, float, void*,
                                            hipStream_t, const float*, int, float*);
template int af_launch_groupnorm_fold<float>(const void*, long, int, int, int, int, const float*, const float*, float, void*,
                                             hipStream_t, const float*, int, float*);
template int af_launch_layernorm<bf16>(const void*, int, long, int, const float*, const float*, float, void*, int,
                                       hipStream_t, float);
template int af_launch_layernorm<float>(const void*, int, long, int, const float*, const float*, float, void*,
                                        int, hipStream_t, float);
