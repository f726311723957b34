// Implicit-GEMM convolution / linear kernel for gfx950 (MFMA 32x32, wave64).
//
// Replaces, on the hot path, every nn.Conv2d (3x3 stride 1/2, 1x1) and nn.Linear
// the reference executes through torch (SURVEY.md §2.2 K1/K1'/K2):
//   ResBlock convs            openaimodel.py:208,234     Downsample.op  :155-157
//   Upsample.conv (+nearest)  openaimodel.py:111,120-122 skip 1x1       :245
//   proj_in/proj_out          attention.py:302-317       to_q/k/v/out   :157-165
//   GEGLU / FeedForward       attention.py:32-59         VAE convs      model.py:83-142
//
// Design (MI355X-first, not a translation of a cuDNN call):
//   * activations NHWC, weights repacked to [Cout][ky][kx][Cin] so K is contiguous
//     for both MFMA operands; one K tile = 128 bytes per row (64 bf16 / 32 f32)
//     and lies inside a single filter tap, so the im2col gather is one predicated
//     16-byte load per lane (zero padding, stride, nearest-2x upsample folded in).
//   * 256 threads = 4 waves as 2(M) x 2(N); each wave owns (BM/2)x(BN/2) of the
//     tile as 32x32 MFMA blocks.  The WEIGHT tile is the MFMA A operand and the
//     ACTIVATION tile the B operand, so the accumulator has the output pixel on
//     the lane and 4 consecutive output channels in consecutive registers ->
//     8/16-byte epilogue stores and vector bias/residual loads.
//   * LDS tiles are [rows][128 B] with a 16-byte-chunk XOR swizzle
//     chunk ^= (row>>1)&7, conflict-free for ds_read_b128's 16-lane groups.
//   * register-staged double buffering: tile k+1's global loads are issued before
//     tile k's MFMAs and written to the other LDS buffer after them; one barrier
//     per K tile.
//   * fused epilogues: alpha, bias, per-sample time-embedding bias, residual add,
//     GEGLU (value*gelu(gate) with value/gate rows interleaved in 32-row groups).
#include "af_common.h"

template <int BM, int BN> struct TileCfg {
  static constexpr int XR = BM / 32, WR = BN / 32;
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int MI = WM / 32, NI = WN / 32;
  static constexpr int TILE_BYTES = (BM + BN) * 128;
  static constexpr int LDS_BYTES = 2 * TILE_BYTES;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
  using C = TileCfg<BM, BN>;
  constexpr int EPC = 16 / sizeof(T);
  constexpr int BK = 128 / sizeof(T);
  constexpr int XR = C::XR, WR = C::WR, MI = C::MI, NI = C::NI;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const long z = blockIdx.z;

  const T* __restrict__ src = reinterpret_cast<const T*>(p.src) + z * p.bs_src;
  const T* __restrict__ Wt = reinterpret_cast<const T*>(p.W) + z * p.bs_w;

  const int chunk = tid & 7, r0 = tid >> 3;
  const int HoWo = p.Ho * p.Wo;

  // --- per-thread gather state for the activation rows it stages ---
  int x_iy0[XR], x_ix0[XR];
  long x_base[XR];
  bool x_ok[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    int m = m0 + r0 + 32 * i;
    bool ok = m < p.M;
    int mm = ok ? m : 0;
    int b = mm / HoWo;
    int rem = mm - b * HoWo;
    int oy = rem / p.Wo;
    int ox = rem - oy * p.Wo;
    x_iy0[i] = oy * p.stride - p.pad;
    x_ix0[i] = ox * p.stride - p.pad;
    x_base[i] = (long)b * p.src_batch_stride;
    x_ok[i] = ok;
  }
  const T* w_ptr[WR];
  bool w_ok[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = n0 + r0 + 32 * i;
    w_ok[i] = n < p.Wrows;
    w_ptr[i] = Wt + (long)(w_ok[i] ? n : 0) * p.ldw + chunk * EPC;
  }

  uint4 xr[XR], wr[WR];
  int ky = 0, kx = 0, c0 = 0;  // filter tap and channel offset of the NEXT tile to load

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
      bool ok = x_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      int sy = ok ? (iy >> p.up) : 0, sx = ok ? (ix >> p.up) : 0;
      const T* ptr = src + x_base[i] + ((long)(sy * p.Ws + sx)) * p.ldc + c0 + chunk * EPC;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *reinterpret_cast<const uint4*>(ptr);
      xr[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (w_ok[i]) v = *reinterpret_cast<const uint4*>(w_ptr[i] + k0);
      wr[i] = v;
    }
    // advance tap state
    c0 += BK;
    if (c0 >= p.Cin) {
      c0 = 0;
      if (++kx >= p.ks) { kx = 0; ++ky; }
    }
  };
  auto lstore = [&](int buf) {
    char* xs = smem + buf * C::TILE_BYTES;
    char* ws = xs + BM * 128;
#pragma unroll
    for (int i = 0; i < XR; ++i)
      *reinterpret_cast<uint4*>(xs + lds_off(r0 + 32 * i, chunk)) = xr[i];
#pragma unroll
    for (int i = 0; i < WR; ++i)
      *reinterpret_cast<uint4*>(ws + lds_off(r0 + 32 * i, chunk)) = wr[i];
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int b = 0; b < MI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int KT = p.K / BK;
  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) gload((kt + 1) * BK);
    const char* xs = smem + cur * C::TILE_BYTES;
    const char* ws = xs + BM * 128;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 xf[MI], wf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        xf[mi] = *reinterpret_cast<const uint4*>(xs + lds_off(wm * C::WM + mi * 32 + l31, 2 * s + h));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        wf[ni] = *reinterpret_cast<const uint4*>(ws + lds_off(wn * C::WN + ni * 32 + l31, 2 * s + h));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) Mma<T>::step(wf[ni], xf[mi], acc[ni][mi]);
    }
    if (kt + 1 < KT) lstore(cur ^ 1);
    __syncthreads();
  }

  // ------------------------------- epilogue -------------------------------
  T* __restrict__ out = reinterpret_cast<T*>(p.out) + z * p.bs_out;
  const T* __restrict__ res = p.residual ? reinterpret_cast<const T*>(p.residual) + z * p.bs_res : nullptr;
  const T* __restrict__ rowb = reinterpret_cast<const T*>(p.rowbias);

  if (p.epilogue == AF_EPI_GEGLU) {
    if constexpr (NI == 2) {
      // rows [g*64, g*64+32) of the packed weight are "value", [g*64+32, g*64+64) "gate"
      const int g = (n0 + wn * 64) >> 6;
      const int Nout = p.N >> 1;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * C::WM + mi * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 8 * q + 4 * h;           // row inside the 32-row block
          const int nv = n0 + wn * 64 + j;       // packed row of the value
          const int no = g * 32 + j;             // output column
          if (no >= Nout) continue;
          Quad<T> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float val = acc[0][mi][4 * q + e] * p.alpha;
            float gat = acc[1][mi][4 * q + e] * p.alpha;
            if (p.bias) { val += p.bias[nv + e]; gat += p.bias[nv + 32 + e]; }
            o.e[e] = from_f32<T>(val * gelu_erf_f(gat));
          }
          o.store(out + (long)m * p.ldo + no);
        }
      }
    }
    return;
  }

#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * C::WM + mi * 32 + l31;
    if (m >= p.M) continue;
    const int b = rowb ? (m / HoWo) : 0;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * C::WN + ni * 32 + 8 * q + 4 * h;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[ni][mi][4 * q + e] * p.alpha;
        if (p.bias) {
          const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
        }
        if (rowb) {
          Quad<T> rb;
          rb.load(rowb + (long)b * p.ldrb + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(rb.e[e]);
        }
        if (res) {
          Quad<T> rv;
          rv.load(res + (long)m * p.ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(rv.e[e]);
        }
        Quad<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.e[e] = from_f32<T>(v[e]);
        o.store(out + (long)m * p.ldo + n);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host launcher
// ---------------------------------------------------------------------------
template <typename T, int BM, int BN>
static int launch_cfg(const ConvGemmParams& p, int batch, hipStream_t stream) {
  using C = TileCfg<BM, BN>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, batch);
  hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN>), grid, dim3(256), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <typename T> int af_launch_conv_gemm(const ConvGemmParams& p, int batch, hipStream_t stream) {
  constexpr int BK = 128 / sizeof(T);
  if (p.K % BK != 0 || p.Cin % BK != 0 || p.K != p.ks * p.ks * p.Cin) {
    af_set_error_msg("conv_gemm: K=%d Cin=%d ks=%d must satisfy K==ks*ks*Cin and Cin%%%d==0", p.K, p.Cin, p.ks, BK);
    return -1;
  }
  if (p.ldw % (16 / (int)sizeof(T)) != 0 || p.ldc % (16 / (int)sizeof(T)) != 0) {
    af_set_error_msg("conv_gemm: ldw/ldc must be multiples of 16 bytes");
    return -1;
  }
  if (p.N % 4 != 0 || p.ldo % 4 != 0 || (p.residual && p.ldr % 4 != 0) || (p.rowbias && p.ldrb % 4 != 0)) {
    af_set_error_msg("conv_gemm: N/ldo/ldr/ldrb must be multiples of 4 (N=%d ldo=%d)", p.N, p.ldo);
    return -1;
  }
  if (p.M <= 0 || p.N <= 0) return 0;
  AfProfScope prof(AF_K_CONV_GEMM, stream, 2.0 * p.M * (double)p.N * (p.k_logical ? p.k_logical : p.K) * batch,
                   ((double)p.M * p.K / (p.ks * p.ks) + (double)p.N * p.K + (double)p.M * p.N) * batch * sizeof(T));
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * batch;
  if (p.epilogue == AF_EPI_GEGLU) {
    if (p.N % 64 != 0) { af_set_error_msg("conv_gemm: GEGLU needs N%%64==0"); return -1; }
    return t128 >= 256 ? launch_cfg<T, 128, 128>(p, batch, stream) : launch_cfg<T, 64, 128>(p, batch, stream);
  }
  const bool n128 = (p.N % 128) == 0;
  if (n128) {
    return t128 >= 256 ? launch_cfg<T, 128, 128>(p, batch, stream) : launch_cfg<T, 64, 128>(p, batch, stream);
  } else {
    const long t = (long)((p.M + 127) / 128) * ((p.N + 63) / 64) * batch;
    return t >= 256 ? launch_cfg<T, 128, 64>(p, batch, stream) : launch_cfg<T, 64, 64>(p, batch, stream);
  }
}

template int af_launch_conv_gemm<bf16>(const ConvGemmParams&, int, hipStream_t);
template int af_launch_conv_gemm<float>(const ConvGemmParams&, int, hipStream_t);
