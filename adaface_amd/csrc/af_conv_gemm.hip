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
//     the lane and 4 consecutive output channels in consecutive registers.
//   * LDS tiles are [rows][128 B] with a 16-byte-chunk XOR swizzle
//     chunk ^= (row>>1)&7, conflict-free for ds_read_b128's 16-lane groups.
//   * register-staged double buffering: tile k+1's global loads are issued before
//     tile k's MFMAs and written to the other LDS buffer after them; one barrier
//     per K tile.
//   * epilogue through LDS: the fp32 accumulator tile is transposed in LDS so that
//     every lane then handles 8 consecutive output channels of one pixel: bias /
//     time-embedding / residual are read and the output written as whole 16-byte
//     vectors, a wave covering full contiguous rows (the per-lane 8-byte scattered
//     stores this replaces were 2-3x slower on the K<=640 linears).
//   * 1-D grid with a bijective XCD-aware remap: the N tiles that share one
//     activation tile are adjacent on one XCD, so the tile is fetched into that
//     XCD's L2 once; weights (<= a few MB) stay L2-resident everywhere.
//   * split-K (deep-K, small-M layers at 8x8 / 16x16): K slices write fp32 slabs,
//     a second kernel reduces them and applies the epilogue.
//   * fused epilogues: alpha, bias, per-sample time-embedding bias, residual add,
//     GEGLU (value*gelu(gate) with value/gate rows interleaved in 16-row groups).
#include "af_kernels.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

template <int BM, int BN> struct TileCfg {
  static constexpr int XR = BM / 32, WR = BN / 32;
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int MI = WM / 32, NI = WN / 32;
  static constexpr int TILE_BYTES = (BM + BN) * 128;
  static constexpr int EPI_LD = BN + 4;  // floats; +4 keeps float4 alignment and spreads rows over banks
  static constexpr int EPI_BYTES = BM * EPI_LD * 4;
  static constexpr int LDS_BYTES = (2 * TILE_BYTES > EPI_BYTES) ? 2 * TILE_BYTES : EPI_BYTES;
};

// LDS-DMA: 16 bytes per lane from a buffer resource straight into LDS at (wave-uniform base) + lane*16.
// (The builtin only exists in the device pass; the host pass of this translation unit must not see it, or clang
// silently drops the kernel's host stub.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset,
                                           0, 0);
#endif
}

// Workgroup id -> (M tile, N tile).  First the bijective XCD remap (blocks b, b+8, ... share an XCD and its L2:
// give each XCD one contiguous range of ids), then grouped ordering: ids sweep `gm` M tiles before moving to the next
// N tile, so the ~64 workgroups co-resident on one XCD form a gm x (64/gm) patch of the output — the activation
// K-slices they stream are shared by 64/gm of them and the weight K-slices by gm of them instead of every workgroup
// re-fetching its operands past the 4 MiB L2 (measured: up to 35x the footprint on the N>=1920 projections).
__device__ __forceinline__ void tile_coords(int wg, int nwg, int ntm, int ntn, int gm, int& tm, int& tn) {
  const int xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
  wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  const int per_group = gm * ntn;
  const int gid = wg / per_group;
  const int first_m = gid * gm;
  const int gsz = min(ntm - first_m, gm);
  const int in_g = wg - gid * per_group;
  tn = in_g / gsz;
  tm = first_m + (in_g - tn * gsz);
}

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// 8 consecutive outputs of one row: bias / rowbias / residual / store as 16-byte vectors
template <typename T>
__device__ __forceinline__ void epi_store8(const ConvGemmParams& p, float (&v)[8], int m, int b, int n, int nvalid,
                                           T* __restrict__ out, const T* __restrict__ res,
                                           const T* __restrict__ rowb, const float* __restrict__ bias) {
  if (bias) {
#pragma unroll
    for (int hq = 0; hq < 2; ++hq)
      if (4 * hq < nvalid) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n + 4 * hq);
        v[4 * hq + 0] += bv.x; v[4 * hq + 1] += bv.y; v[4 * hq + 2] += bv.z; v[4 * hq + 3] += bv.w;
      }
  }
  if (rowb) {
#pragma unroll
    for (int hq = 0; hq < 2; ++hq)
      if (4 * hq < nvalid) {
        Quad<T> rb;
        rb.load(rowb + (long)b * p.ldrb + n + 4 * hq);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * hq + e] += to_f32<T>(rb.e[e]);
      }
  }
  if (res) {
#pragma unroll
    for (int hq = 0; hq < 2; ++hq)
      if (4 * hq < nvalid) {
        Quad<T> rv;
        rv.load(res + (long)m * p.ldr + n + 4 * hq);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * hq + e] += to_f32<T>(rv.e[e]);
      }
  }
#pragma unroll
  for (int hq = 0; hq < 2; ++hq)
    if (4 * hq < nvalid) {
      Quad<T> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.e[e] = from_f32<T>(v[4 * hq + e]);
      o.store(out + (long)m * p.ldo + n + 4 * hq);
    }
}

template <typename T, int BM, int BN, bool DMA>
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

  // ---- block -> tile: XCD-aware remap + grouped ordering ----
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, ntm, ntn, p.group_m, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const long z = blockIdx.z;
  const long zb = p.splitk > 1 ? 0 : z;   // batch index
  const int zk = p.splitk > 1 ? (int)z : 0;  // K slice

  // Buffer resources (wave-uniform base, 32-bit per-lane byte offsets).  Lanes that must read zero (image
  // padding, rows beyond M / Wrows) get voffset 0xFFFFFFFF: the hardware range check returns 0 for them, so the
  // gather needs neither branches nor 64-bit address arithmetic.
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.src) + zb * p.bs_src), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.W) + zb * p.bs_w), 0, (int)0xFFFFFFF0u, 0x00020000);

  const int chunk = tid & 7, r0 = tid >> 3;
  // DMA variant: LDS-DMA writes lane-linearly (slot = lane&7 of row r0+32i), so the XOR swizzle moves to the
  // SOURCE address: slot c' of row r receives data chunk c' ^ ((r>>1)&7) (bits 1-3 of r do not depend on i)
  const int schunk = DMA ? (chunk ^ ((r0 >> 1) & 7)) : chunk;
  const int HoWo = p.Ho * p.Wo;
  const unsigned ldcb = (unsigned)p.ldc * (unsigned)sizeof(T);

  // --- per-thread gather state for the activation rows it stages ---
  int x_iy0[XR], x_ix0[XR];
  unsigned x_off[XR];
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
    x_off[i] = (unsigned)((long)b * p.src_batch_stride * (long)sizeof(T)) + (unsigned)(schunk * 16);
    x_ok[i] = ok;
  }
  unsigned w_off[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = n0 + r0 + 32 * i;
    w_off[i] = n < p.Wrows ? (unsigned)(((long)n * p.ldw) * (long)sizeof(T)) + (unsigned)(schunk * 16) : 0xFFFFFFFFu;
  }

  // K range of this block
  const int KT_all = p.K / BK;
  const int kt_per = (KT_all + p.splitk - 1) / p.splitk;
  const int kt_begin = zk * kt_per;
  const int kt_end = min(KT_all, kt_begin + kt_per);
  const int KT = kt_end - kt_begin;

  int ky, kx, c0;  // filter tap and channel offset of the NEXT tile to load
  {
    const int k0 = kt_begin * BK;
    const int tap = k0 / p.Cin;
    c0 = k0 - tap * p.Cin;
    ky = tap / p.ks;
    kx = tap - ky * p.ks;
  }

  auto gload = [&](uint4 (&xr)[XR], uint4 (&wr)[WR], int k0) {
    const unsigned c0b = (unsigned)c0 * (unsigned)sizeof(T);
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
      const bool ok = x_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      const unsigned pix = (unsigned)((iy >> p.up) * p.Ws + (ix >> p.up));
      const unsigned vo = ok ? x_off[i] + pix * ldcb : 0xFFFFFFFFu;
      xr[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, c0b, 0));
    }
    const unsigned k0b = (unsigned)k0 * (unsigned)sizeof(T);
#pragma unroll
    for (int i = 0; i < WR; ++i)
      wr[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[i], k0b, 0));
    c0 += BK;
    if (c0 >= p.Cin) {
      c0 = 0;
      if (++kx >= p.ks) { kx = 0; ++ky; }
    }
  };
  auto lstore = [&](int buf, const uint4 (&xr)[XR], const uint4 (&wr)[WR]) {
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

  // one K tile from LDS buffer `buf`; the fragments of k-step s+1 are read while step s multiplies
  auto compute = [&](int buf) {
    const char* xs = smem + buf * C::TILE_BYTES;
    const char* ws = xs + BM * 128;
    uint4 xf[2][MI], wf[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
      xf[0][mi] = *reinterpret_cast<const uint4*>(xs + lds_off(wm * C::WM + mi * 32 + l31, h));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      wf[0][ni] = *reinterpret_cast<const uint4*>(ws + lds_off(wn * C::WN + ni * 32 + l31, h));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          xf[(s + 1) & 1][mi] =
              *reinterpret_cast<const uint4*>(xs + lds_off(wm * C::WM + mi * 32 + l31, 2 * (s + 1) + h));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          wf[(s + 1) & 1][ni] =
              *reinterpret_cast<const uint4*>(ws + lds_off(wn * C::WN + ni * 32 + l31, 2 * (s + 1) + h));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) Mma<T>::step(wf[s & 1][ni], xf[s & 1][mi], acc[ni][mi]);
    }
  };

  if constexpr (!DMA) {
    // Two register sets: while tile kt is multiplied from LDS, tile kt+1 sits in one set (loaded a whole
    // iteration ago, so its ds_write does not wait) and tile kt+2's loads are issued into the other.
    uint4 xa[XR], wa[WR], xb[XR], wb[WR];
    if (KT > 0) gload(xa, wa, kt_begin * BK);
    if (KT > 1) gload(xb, wb, (kt_begin + 1) * BK);
    if (KT > 0) lstore(0, xa, wa);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += 2) {
      if (kt + 2 < KT) gload(xa, wa, (kt_begin + kt + 2) * BK);
      compute(0);
      if (kt + 1 < KT) lstore(1, xb, wb);
      __syncthreads();
      if (kt + 1 >= KT) break;
      if (kt + 3 < KT) gload(xb, wb, (kt_begin + kt + 3) * BK);
      compute(1);
      if (kt + 2 < KT) lstore(0, xa, wa);
      __syncthreads();
    }
  } else {
    // LDS-DMA staging (buffer_load ... lds): no VGPR round trip and no ds_write; tile kt+1 lands in the other
    // buffer while tile kt is multiplied.  Out-of-range lanes (padding / tails) are written as zeros by the DMA.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto gdma = [&](int buf, int k0) {
      char* xs = smem + buf * C::TILE_BYTES + wv * 1024;
      char* ws = xs + BM * 128;
      const unsigned c0b = (unsigned)c0 * (unsigned)sizeof(T);
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        const int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
        const bool ok = x_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const unsigned pix = (unsigned)((iy >> p.up) * p.Ws + (ix >> p.up));
        const unsigned vo = ok ? x_off[i] + pix * ldcb : 0xFFFFFFFFu;
        lds_dma16(rs_x, xs + i * 4096, vo, c0b);
      }
      const unsigned k0b = (unsigned)k0 * (unsigned)sizeof(T);
#pragma unroll
      for (int i = 0; i < WR; ++i)
        lds_dma16(rs_w, ws + i * 4096, w_off[i], k0b);
      c0 += BK;
      if (c0 >= p.Cin) {
        c0 = 0;
        if (++kx >= p.ks) { kx = 0; ++ky; }
      }
    };
    if (KT > 0) gdma(0, kt_begin * BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < KT) gdma(cur ^ 1, (kt_begin + kt + 1) * BK);
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ------------------------------- epilogue -------------------------------
  // phase 1: accumulators -> fp32 LDS tile [BM][BNo (+4)]
  float* et = reinterpret_cast<float*>(smem);
  const bool geglu = p.epilogue == AF_EPI_GEGLU;
  int BNo = BN;  // columns of the staged tile
  if (geglu) {
    if constexpr (NI == 2) {
      BNo = BN / 2;
      // packed weight rows [32k, 32k+16) are "value", [32k+16, 32k+32) "gate": within a 32-column MFMA block the
      // value sits in accumulator registers 0-7 (columns 8q+4h+e, q < 2) and its gate in registers 8-15 of the SAME lane
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * C::WM + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int j = 8 * q + 4 * h;
            const int nv = n0 + wn * 64 + ni * 32 + j;  // packed row index of the value (bias is packed the same way)
            float4 o;
            float* op = reinterpret_cast<float*>(&o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float val = acc[ni][mi][4 * q + e] * p.alpha;
              float gat = acc[ni][mi][4 * (q + 2) + e] * p.alpha;
              if (p.bias) { val += p.bias[nv + e]; gat += p.bias[nv + 16 + e]; }
              op[e] = val * (sizeof(T) == 2 ? gelu_bf16out_f(gat) : gelu_erf_f(gat));
            }
            *reinterpret_cast<float4*>(et + row * C::EPI_LD + wn * 32 + ni * 16 + j) = o;
          }
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row = wm * C::WM + mi * 32 + l31;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 o;
          o.x = acc[ni][mi][4 * q + 0] * p.alpha; o.y = acc[ni][mi][4 * q + 1] * p.alpha;
          o.z = acc[ni][mi][4 * q + 2] * p.alpha; o.w = acc[ni][mi][4 * q + 3] * p.alpha;
          *reinterpret_cast<float4*>(et + row * C::EPI_LD + wn * C::WN + ni * 32 + 8 * q + 4 * h) = o;
        }
    }
  }
  __syncthreads();

  // phase 2: each thread owns 8 consecutive columns of a row; a wave covers whole contiguous rows
  const int tpr = BNo >> 3;            // threads per row
  const int rpp = 256 / tpr;           // rows per pass
  const int trow = tid / tpr, c8 = (tid - trow * tpr) * 8;
  const int ncol0 = geglu ? (n0 >> 1) : n0;
  const int Nvalid = geglu ? (p.N >> 1) : p.N;
  const int n = ncol0 + c8;

  if (p.splitk > 1) {
    // raw fp32 partial sums into this slice's slab [M][N]
    float* slab = reinterpret_cast<float*>(p.ws) + (long)zk * p.M * p.N;
    for (int row = trow; row < BM; row += rpp) {
      const int m = m0 + row;
      if (m >= p.M || n >= Nvalid) continue;
      const float4 a = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8);
      *reinterpret_cast<float4*>(slab + (long)m * p.N + n) = a;
      if (n + 4 < Nvalid) {
        const float4 b4 = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8 + 4);
        *reinterpret_cast<float4*>(slab + (long)m * p.N + n + 4) = b4;
      }
    }
    return;
  }

  T* __restrict__ out = reinterpret_cast<T*>(p.out) + zb * p.bs_out;
  const T* __restrict__ res = p.residual ? reinterpret_cast<const T*>(p.residual) + zb * p.bs_res : nullptr;
  const T* __restrict__ rowb = reinterpret_cast<const T*>(p.rowbias);
  const float* __restrict__ bias = geglu ? nullptr : p.bias;
  if (n < Nvalid) {
    const int nvalid = min(8, Nvalid - n);
    for (int row = trow; row < BM; row += rpp) {
      const int m = m0 + row;
      if (m >= p.M) break;
      float v[8];
      const float4 a = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8);
      const float4 b4 = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
      epi_store8<T>(p, v, m, rowb ? m / HoWo : 0, n, nvalid, out, res, rowb, bias);
    }
  }
}

// ---------------------------------------------------------------------------
// "Ping-pong" tile: 256 x BN x 64 (BN = 160 or 128), 512 threads, bf16, LDS-DMA ring of three K tiles.
//
// The 128x128 / 4-wave structure above tops out near 900 TFLOP/s: every wave alternates LDS reads, MFMAs and a
// barrier, so the matrix pipe idles while fragments are fetched.  Here waves 0-3 (group 0, one per SIMD) and waves
// 4-7 (group 1, their SIMD partners) run the SAME program one barrier apart (lab: scripts/lab/gemm_pp_lab.hip):
//
//     interval 2t   : group 0  D(t)  issue LDS-DMA              | group 1  C(t-1) MFMAs
//     interval 2t+1 : group 0  C(t)  MFMAs + fragment reads     | group 1  D(t)
//
//   * D(t): group 0 issues its pieces of K tile t+1, group 1 its pieces of tile t+2 (a piece = 8 rows x 128 B = one
//     buffer_load_dwordx4 ... lds; 7 / 6 pieces per wave at BN = 160).  Nothing else: a DMA costs the issuing wave
//     60-75 cycles, so the fragment reads do not live here.
//   * C(t): the MFMAs run half a tile behind the reads: [read unit 2t | MFMA unit 2t-1] [read unit 2t+1 | MFMA
//     unit 2t] (unit = 32 of the tile's 64 K values), one ds_read_b128 issued behind each of the first nine MFMAs of
//     a half, so fragment registers stay at one tile (72 VGPRs) and the reads' latency is covered by MFMAs.
//   * counted s_waitcnt vmcnt(N) (never 0 inside the loop) + raw s_barrier: a tile is waited for by the waves that
//     issued it one barrier before its first reader, and a slot is refilled only after the barrier that follows its
//     last reader's lgkmcnt(0).  Fragment reads are inline asm: hipcc puts s_waitcnt vmcnt(0) in front of any LDS
//     read it can see after an LDS-DMA, which would drain the ring every tile.
//   * each wave owns 64 rows x BN/2 columns as 4 x (BN/32) MFMA 16x16x32 blocks; the weight tile is the MFMA A
//     operand, so a lane holds 4 consecutive output channels of one output pixel.
//   * swizzle on the DMA SOURCE address (LDS-DMA writes lane-linearly): slot c' of row r holds chunk c' ^ (r & 7),
//     conflict-free for the 16-lane groups of ds_read_b128 on 16-row fragments.
//   * epilogue through LDS in two 128-row passes (fp32), then the same 16-byte row stores as conv_gemm_kernel.
// Measured on plain GEMMs (lab): 1.15-1.27 PFLOP/s where the 128x128 kernel gives 0.6-0.9.
// ---------------------------------------------------------------------------
template <int BN> struct PpCfg {
  static constexpr int BM = 256, BK = 64;
  static constexpr int HN = BN / 2;                 // columns per wave group
  static constexpr int NI = HN / 16, MI = 4;        // 16x16 blocks per wave
  static constexpr int XBYTES = BM * 128, WBYTES = BN * 128, SLOT = XBYTES + WBYTES;
  static constexpr int XP = BM / 8, WP = BN / 8;    // 1-KiB pieces per K tile
  // (splitting the gathers evenly between the two groups -- 4 + 2..3 pieces per wave each -- measured 5 % SLOWER on the
  // bf16 3x3 convolutions and the same on fp8: the group that computes first after a barrier is better left with the
  // cheap weight pieces)
  static constexpr int XP0 = ((XP + WP + 7) / 8) * 4;  // activation pieces staged by group 0
  static constexpr int NP0 = XP0 / 4;               // pieces per wave, group 0 (all activation)
  static constexpr int NX1 = (XP - XP0) / 4;        // activation pieces per wave, group 1
  static constexpr int NW1 = WP / 4;                // weight pieces per wave, group 1
  static constexpr int NP1 = NX1 + NW1;
  static constexpr int NPMAX = NP0 > NP1 ? NP0 : NP1;
  static constexpr int EPI_LD = BN + 4;
  static constexpr int EPI_BYTES = 128 * EPI_LD * 4;
  static constexpr int LDS_BYTES = 3 * SLOT;
  static_assert(XP0 % 4 == 0 && (XP - XP0) % 4 == 0 && WP % 4 == 0 && HN % 16 == 0, "piece split");
  static_assert(EPI_BYTES <= LDS_BYTES && NP0 >= NX1, "epilogue tile / gather state");
};

typedef __attribute__((ext_vector_type(4))) unsigned pp_u32x4;
template <int I, int N, typename F> __device__ __forceinline__ void pp_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pp_static_for<I + 1, N>(f);
  }
}
template <int OFF> __device__ __forceinline__ pp_u32x4 pp_lds_read128(unsigned addr) {
  pp_u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_wait_lgkm0() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);   // hipcc hoists register-only MFMAs over an asm wait without this
}

// One output quad of a sliced-K launch: out[m][n .. n+3] = T(sum_z slab[z][m][n..] + bias + rowbias + residual), slabs summed
// in slice order (splitk_reduce_kernel).
template <typename T>
__device__ __forceinline__ void splitk_reduce_quad(const ConvGemmParams& p, int m, int n) {
  const float* ws = reinterpret_cast<const float*>(p.ws);
  f32x4 a;
  a = *reinterpret_cast<const f32x4*>(ws + (long)m * p.N + n);
  for (int zz = 1; zz < p.splitk; ++zz) a += *reinterpret_cast<const f32x4*>(ws + ((long)zz * p.M + m) * p.N + n);
  float v[4] = {a[0], a[1], a[2], a[3]};
  if (p.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
  if (p.rowbias) {
    Quad<T> rb;
    rb.load(reinterpret_cast<const T*>(p.rowbias) + (long)(m / (p.Ho * p.Wo)) * p.ldrb + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(rb.e[e]);
  }
  if (p.residual) {
    Quad<T> rv;
    rv.load(reinterpret_cast<const T*>(p.residual) + (long)m * p.ldr + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += to_f32<T>(rv.e[e]);
  }
  Quad<T> o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o.e[e] = from_f32<T>(v[e]);
  o.store(reinterpret_cast<T*>(p.out) + (long)m * p.ldo + n);
}
// Epilogue of the eight-wave kernels (shared by conv_gemm_pp_kernel and conv3x3_halo8_kernel): bias / LayerNorm / GEGLU on
// the accumulators, then either straight from the registers or through an fp32 LDS tile in two 128-row passes.
// Accumulator layout: wave (g, wq) holds rows m0 + wq * 64 + j * 16 + (lane & 15), columns n0 + g * HN + i * 16 +
// 4 * (lane >> 4) + e in acc[i][j][e].
template <int BN, int LNMODE>
__device__ __forceinline__ void pp_epilogue(const ConvGemmParams& p, f32x4 (&acc)[PpCfg<BN>::NI][PpCfg<BN>::MI],
                                            float4 (&bias_r)[PpCfg<BN>::NI], float4 (&ln_cs)[LNMODE == 1 ? PpCfg<BN>::NI : 1],
                                            float (&ln_mu)[LNMODE == 1 ? PpCfg<BN>::MI : 1],
                                            float (&ln_rs)[LNMODE == 1 ? PpCfg<BN>::MI : 1], char* smem, int tid, int lane,
                                            int g, int wq, int m0, int n0, int tn, int zk) {
  using C = PpCfg<BN>;
  typedef bf16 T;
  constexpr int NI = C::NI, MI = C::MI;
  const int cl = 4 * (lane >> 4);
  const int HoWo = p.Ho * p.Wo;
  // ------------------------------- epilogue -------------------------------
  // GEGLU is evaluated in registers by all eight waves first (value block 2k, gate block 2k+1 of the same lane);
  // then two passes of 128 rows: waves with (wq >> 1) == pass put their accumulators into an fp32 LDS tile and all
  // 512 threads walk it in 8-column vectors (a wave covers whole contiguous rows).  Every load of a pass (LDS tile,
  // time-embedding row, residual) is issued before the first store so their latencies overlap.
  float* et = reinterpret_cast<float*>(smem);
  const bool geglu = p.epilogue == AF_EPI_GEGLU;
  if (geglu) {
    if constexpr ((NI & 1) == 0) {
      // packed weight rows [32k, 32k+16) are "value", [32k+16, 32k+32) "gate" (bias packed alike)
#pragma unroll
      for (int k2 = 0; k2 < NI / 2; ++k2) {
        const float* bvp = reinterpret_cast<const float*>(&bias_r[2 * k2]);
        const float* bgp = reinterpret_cast<const float*>(&bias_r[2 * k2 + 1]);
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {      // pairs: the packed f32 instructions do two values each
            f32x2 val, gat;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if constexpr (LNMODE == 1) {
                val[h] = (acc[2 * k2][j][e + h] - ln_mu[j] * reinterpret_cast<const float*>(&ln_cs[2 * k2])[e + h]) * ln_rs[j] + bvp[e + h];
                gat[h] = (acc[2 * k2 + 1][j][e + h] - ln_mu[j] * reinterpret_cast<const float*>(&ln_cs[2 * k2 + 1])[e + h]) * ln_rs[j] + bgp[e + h];
              } else {
                val[h] = acc[2 * k2][j][e + h] * p.alpha + bvp[e + h];
                gat[h] = acc[2 * k2 + 1][j][e + h] * p.alpha + bgp[e + h];
              }
            }
            const f32x2 r = val * gelu_bf16out_f2(gat);
            acc[k2][j][e] = r.x;                 // block k2 <= 2 k2: already consumed
            acc[k2][j][e + 1] = r.y;
          }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float* bp = reinterpret_cast<const float*>(&bias_r[i]);
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (LNMODE == 1)
            acc[i][j][e] = (acc[i][j][e] - ln_mu[j] * reinterpret_cast<const float*>(&ln_cs[i])[e]) * ln_rs[j] + bp[e];
          else
            acc[i][j][e] = acc[i][j][e] * p.alpha + bp[e];
        }
    }
  }
  const int BNo = geglu ? BN / 2 : BN;              // columns of the staged tile
  const int HNo = geglu ? C::HN / 2 : C::HN;        // ... per wave group
  const int tpr = BNo >> 3;
  const int ncol0 = geglu ? (n0 >> 1) : n0;
  const int Nvalid = geglu ? (p.N >> 1) : p.N;
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  const T* __restrict__ rowb = reinterpret_cast<const T*>(p.rowbias);
  float* slab = p.splitk > 1 ? reinterpret_cast<float*>(p.ws) + (long)zk * p.M * p.N : nullptr;
  if (LNMODE == 2 || p.pp_epilogue == 2 || (p.pp_epilogue == 0 && (geglu || slab))) {
    // Direct epilogue: every lane stores its 4 consecutive output channels of a pixel straight from the accumulator
    // (8-byte stores, four lanes covering a 32-byte run of the row; fp32 split-K slabs: 16-byte stores).  No LDS pass,
    // no workgroup barriers; the time-bias / residual quads of a row group are fetched before its first store.
    // Measured against the two-pass LDS transposition below: GEGLU -4...-6 %, split-K slabs -2...-5 %, everything else
    // within +-1 %, so it is the default for those two and AF_PP_DIRECT = 0 / 1 forces either.
    const int nblk = geglu ? NI / 2 : NI;
    const int cbase = ncol0 + g * HNo + cl;
    // bf16 outputs leave through a WAVE-PRIVATE LDS tile (64 rows x HNo columns, no barriers) as 16-byte stores of whole
    // row segments: straight from the accumulators a store instruction covers 32 bytes of each of 16 rows, and that
    // pattern, not the arithmetic, was the cost of this epilogue (row-panel GEGLU: 160 -> 110 us from this change alone)
    const int wpitch = HNo * 2 + 16;                  // bytes; 80 / 144 / 176: 8-byte writes of 16 rows hit distinct banks
    char* wtile = smem + (g * 4 + wq) * (64 * (C::HN * 2 + 16));
    auto wput = [&](int j, int i, const Quad<T>& o) { o.store(reinterpret_cast<T*>(wtile + (j * 16 + (lane & 15)) * wpitch) + i * 16 + cl); };
    auto wflush = [&]() {
      const int cpr = HNo >> 3;                       // 16-byte chunks per row segment (4 / 8 / 10)
      const int total = 64 * cpr;
#pragma unroll
      for (int t = 0; t < C::HN / 8; ++t) {
        const int c = lane + 64 * t;
        if (c >= total) break;
        const int row = c / cpr, ch = c - row * cpr;
        const int m = m0 + wq * 64 + row;
        const uint4 v = *reinterpret_cast<const uint4*>(wtile + row * wpitch + ch * 16);
        long orow = m;
        if (p.phase4) {   // row m = pixel (b, y, x) of the stored map -> pixel (2 y + dy, 2 x + dx) of the upsampled output
          const int ph = blockIdx.y;
          const int b = m >> p.howo_shift, rem = m & ((1 << p.howo_shift) - 1);
          const int y = rem >> p.wo_shift, x = rem & ((1 << p.wo_shift) - 1);
          orow = ((long)b << (p.howo_shift + 2)) + ((long)(2 * y + (ph >> 1)) << (p.wo_shift + 1)) + 2 * x + (ph & 1);
        }
        if (m < p.M) *reinterpret_cast<uint4*>(out + orow * p.ldo + ncol0 + g * HNo + ch * 8) = v;
      }
    };
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const int m = m0 + wq * 64 + j * 16 + (lane & 15);
      if constexpr (LNMODE == 2) {
        // statistics producer: every lane takes part in the row reduction, rows >= M contribute nothing and store nothing
        const bool mok = m < p.M;
        Quad<T> rq[NI], bq[NI];
        if (rowb && mok) {
          const T* rp = rowb + (long)(p.howo_shift >= 0 ? m >> p.howo_shift : m / HoWo) * p.ldrb + cbase;
#pragma unroll
          for (int i = 0; i < NI; ++i) bq[i].load(rp + i * 16);
        }
        if (res && mok) {
          const T* rp = res + (long)m * p.ldr + cbase;
#pragma unroll
          for (int i = 0; i < NI; ++i) rq[i].load(rp + i * 16);
        }
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          Quad<T> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][e];
            if (rowb && mok) v += to_f32<T>(bq[i].e[e]);
            if (res && mok) v += to_f32<T>(rq[i].e[e]);
            o.e[e] = from_f32<T>(v);
            const float vr = to_f32<T>(o.e[e]);   // the value the consumer will read
            ps += vr;
            pq += vr * vr;
          }
          wput(j, i, o);
        }
        ps += __shfl_xor(ps, 16, 64); pq += __shfl_xor(pq, 16, 64);
        ps += __shfl_xor(ps, 32, 64); pq += __shfl_xor(pq, 32, 64);
        if (mok && (lane >> 4) == 0)
          *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(tn * 2 + g) * p.M + m) * 2) = float2{ps, pq};
        continue;
      }
      if (slab) {
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (i >= nblk) continue;
          *reinterpret_cast<f32x4*>(slab + (long)m * p.N + cbase + i * 16) = acc[i][j];
        }
        continue;
      }
      const bool mok2 = m < p.M;
      Quad<T> rq[NI], bq[NI];
      if (rowb && mok2) {
        const T* rp = rowb + (long)(p.howo_shift >= 0 ? m >> p.howo_shift : m / HoWo) * p.ldrb + cbase;
#pragma unroll
        for (int i = 0; i < NI; ++i) if (i < nblk) bq[i].load(rp + i * 16);
      }
      if (res && mok2) {
        const T* rp = res + (long)m * p.ldr + cbase;
#pragma unroll
        for (int i = 0; i < NI; ++i) if (i < nblk) rq[i].load(rp + i * 16);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        if (i >= nblk) continue;
        Quad<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e];
          if (rowb && mok2) v += to_f32<T>(bq[i].e[e]);
          if (res && mok2) v += to_f32<T>(rq[i].e[e]);
          o.e[e] = from_f32<T>(v);
        }
        wput(j, i, o);
      }
    }
    if (!slab) wflush();
    if (p.gn_stats_out && !slab && !geglu) {
      // GroupNorm statistics of the producer (ConvGemmParams::gn_stats_out): the wave's 64 x HNo outputs still sit, bf16-rounded
      // as stored, in its transposition tile.  Lane = a pair of adjacent columns (never straddles a group: gn_cpg is even),
      // 64 conflict-free 4-byte reads down the rows; then gn_cpg / 2 consecutive lanes per group are summed through the tile.
      const int mslab = m0 + wq * 64;
      if (mslab < p.M) {              // (M % 64 == 0: a slab is all rows or none)
        float s = 0.f, q = 0.f;
        if (lane < HNo / 2) {
#pragma unroll 8
          for (int r = 0; r < 64; ++r) {
            const unsigned w = *reinterpret_cast<const unsigned*>(wtile + r * wpitch + lane * 4);
            const float a = __uint_as_float(w << 16), b = __uint_as_float(w & 0xffff0000u);
            s += a + b;
            q += a * a + b * b;
          }
        }
        float2* scr = reinterpret_cast<float2*>(wtile);   // (LDS operations of one wave execute in order)
        scr[lane] = float2{s, q};
        const int hpg = p.gn_cpg >> 1, ng = (HNo / 2) / hpg;
        if (lane < ng) {
          float a = 0.f, b = 0.f;
          for (int t = 0; t < hpg; ++t) {
            const float2 v = scr[lane * hpg + t];
            a += v.x;
            b += v.y;
          }
          const int hw = 1 << p.howo_shift;
          const long part = ((long)(mslab >> p.howo_shift) * (hw >> 6) + ((mslab & (hw - 1)) >> 6)) * 32 + (ncol0 + g * HNo) / p.gn_cpg + lane;
          *reinterpret_cast<float2*>(p.gn_stats_out + part * 2) = float2{a, b};
        }
      }
    }
    return;
  }
  constexpr int ITEMS = BN / 32;                    // 8-column vectors per thread and pass (GEGLU: half of them)
  const int nitems = geglu ? ITEMS / 2 : ITEMS;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
    if ((wq >> 1) == pass) {
      const int rbase = (wq & 1) * 64 + (lane & 15);
      const int nblk = geglu ? NI / 2 : NI;
#pragma unroll
      for (int j = 0; j < MI; ++j)
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (i < nblk)
            *reinterpret_cast<f32x4*>(et + (rbase + j * 16) * C::EPI_LD + g * HNo + i * 16 + cl) = acc[i][j];
    }
    __syncthreads();
    // ---- gather phase: everything this thread needs for its ITEMS vectors ----
    float v[ITEMS][8];
    int im[ITEMS], in_[ITEMS];
    Quad<T> rq[ITEMS][2], bq[ITEMS][2];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 512;
      const int row = geglu ? idx / (BN / 16) : idx / (BN / 8);   // constant divisors: no runtime integer division
      const int c8 = (idx - row * tpr) * 8;
      const int m = m0 + pass * 128 + row, n = ncol0 + c8;
      const bool ok = it < nitems && m < p.M && n < Nvalid;
      im[it] = ok ? m : -1;
      in_[it] = n;
      if (!ok) continue;
      const float4 a = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8);
      const float4 b4 = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8 + 4);
      v[it][0] = a.x; v[it][1] = a.y; v[it][2] = a.z; v[it][3] = a.w;
      v[it][4] = b4.x; v[it][5] = b4.y; v[it][6] = b4.z; v[it][7] = b4.w;
      if (slab) continue;
      if (rowb) {
        const T* rp = rowb + (long)(p.howo_shift >= 0 ? m >> p.howo_shift : m / HoWo) * p.ldrb + n;
        bq[it][0].load(rp);
        bq[it][1].load(rp + 4);
      }
      if (res) {
        const T* rp = res + (long)m * p.ldr + n;
        rq[it][0].load(rp);
        rq[it][1].load(rp + 4);
      }
    }
    // ---- combine + store ----
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int m = im[it], n = in_[it];
      if (m < 0) continue;
      if (slab) {
        *reinterpret_cast<float4*>(slab + (long)m * p.N + n) = float4{v[it][0], v[it][1], v[it][2], v[it][3]};
        *reinterpret_cast<float4*>(slab + (long)m * p.N + n + 4) = float4{v[it][4], v[it][5], v[it][6], v[it][7]};
        continue;
      }
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        float* vv = v[it] + 4 * hq;
        if (rowb) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] += to_f32<T>(bq[it][hq].e[e]);
        }
        if (res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] += to_f32<T>(rq[it][hq].e[e]);
        }
        Quad<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.e[e] = from_f32<T>(vv[e]);
        o.store(out + (long)m * p.ldo + n + 4 * hq);
      }
    }
  }
}

template <int N> __device__ __forceinline__ void pp_wait_lgkm() {   // counted: LDS reads return in order
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
typedef __attribute__((ext_vector_type(8))) int pp_i32x8;
typedef __attribute__((ext_vector_type(4))) int pp_i32x4;

// LNMODE: 0 = plain, 1 = LayerNorm consumer (A = un-normalised rows, epilogue applies mu / rstd), 2 = LayerNorm-statistics
// producer (direct epilogue + per-row partial sums); separate instantiations so that the plain kernel's code is untouched
//
// FP8: both operands are OCP e4m3 bytes and the K tile is 128 values (the same 128-byte LDS rows, pieces, swizzle and
// fragment reads as bf16), multiplied by v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate) with power-of-two
// scales per weight row / per activation tensor in the instruction's E8M0 operands.  A lane's 32 operand bytes are the
// 16-byte chunks q and q + 4 of its row (q = lane >> 4): any 32 of the row's 128 K values will do as long as the weight
// lane and the activation lane of a lane group hold the same ones, and (q, q + 4) is the conflict-free ds_read_b128 pair
// the bf16 kernel already uses.  A K tile is two 64-channel UNITS (unit u = channel chunk u / taps, tap u % taps), so
// that channel counts that are multiples of 64 but not of 128 (320, 960) waste nothing: chunks 0-3 of an LDS row come
// from unit 2t, chunks 4-7 from unit 2t + 1, each lane of the staging waves walks the units of ITS half.
//
// SCHED: 0 = the round-1 compute phase (two K halves, a full LDS drain after each); 1 = block-ordered compute phase (see
// "FP8 fragments and compute phase" below), staging still in its own phase; 2 = MERGED: no staging phase at all -- every
// wave stages its share of tile t + 2 (one LDS-DMA piece behind every other MFMA of tile t) while it computes, ONE barrier
// per K tile, and both waves of a SIMD always have MFMAs to offer the matrix pipe.
template <int BN, bool GATHER, int LNMODE = 0, bool FP8 = false, int SCHED = FP8 ? 2 : 0>
__global__ __launch_bounds__(512) void conv_gemm_pp_kernel(const ConvGemmParams p) {
  constexpr bool NS = SCHED >= 1, MG = SCHED == 2;
  static_assert(MG || !FP8, "fp8 operands exist on the merged schedule only");
  using C = PpCfg<BN>;
  typedef bf16 T;
  static_assert(!FP8 || LNMODE == 0, "fp8 operands: plain epilogue only");
  constexpr unsigned XE = FP8 ? 1u : 2u;   // bytes per operand element
  constexpr int NI = C::NI, MI = C::MI, SLOT = C::SLOT, XBYTES = C::XBYTES;
  // piece split between the two wave groups.  Phased schedules: group 0 all-activation (7 of 8), group 1 the rest and
  // every weight piece.  Merged: every wave stages while it computes, so the gathers are split evenly.
  constexpr int XP0 = MG ? C::XP / 2 : C::XP0;                       // activation pieces of group 0
  constexpr int NX0 = XP0 / 4, NX1 = (C::XP - XP0) / 4;              // ... per wave, group 0 / 1
  constexpr int WP0 = MG ? (C::WP / 8) * 4 : 0;                      // weight pieces of group 0
  constexpr int NW0 = WP0 / 4, NW1 = (C::WP - WP0) / 4;
  constexpr int NP0 = NX0 + NW0, NP1 = NX1 + NW1, NPMAX = NP0 > NP1 ? NP0 : NP1;
  constexpr int NXM = NX0 > NX1 ? NX0 : NX1, NWM = NW0 > NW1 ? NW0 : NW1;
  static_assert(XP0 % 4 == 0 && WP0 % 4 == 0 && NW1 >= 1, "piece split");
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wid >> 2, wq = wid & 3;
  const int ntm = (p.M + 255) / 256, ntn = p.N / BN;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, ntm, ntn, p.group_m, tm, tn);
  const int m0 = tm * 256, n0 = tn * BN;
  const int zk = blockIdx.z;
  // phase-decomposed upsampled convolution (ConvGemmParams::W_up4): blockIdx.y = output phase, its own padding and weights
  const int ph4 = (GATHER && p.phase4) ? (int)blockIdx.y : 0;
  const int pad_y = (GATHER && p.phase4) ? 1 - (ph4 >> 1) : p.pad, pad_x = (GATHER && p.phase4) ? 1 - (ph4 & 1) : p.pad;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.src)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.W)), 0, (int)0xFFFFFFF0u, 0x00020000);

  // K range of this block (split-K slices)
  const int KT_all = p.K / (FP8 ? 128 : 64);
  const int kt_per = (KT_all + p.splitk - 1) / p.splitk;
  const int kt_begin = zk * kt_per;
  const int KT = min(KT_all, kt_begin + kt_per) - kt_begin;

  // ---- staging state.  Piece q of this wave: group 0: activation piece wq + 4q (q < NP0); group 1: activation
  // piece XP0 + wq + 4q (q < NX1), then weight piece wq + 4(q - NX1).  Lane: row lane>>3 of the piece, LDS slot
  // lane&7, which receives data chunk (lane&7) ^ (row&7).
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);            // data chunk of the 128-byte row this lane fills
  const unsigned lchunk = (FP8 ? (dchunk & 3u) : dchunk) * 16u;     // ... its byte offset inside the 64-channel run
  const int HoWo = p.Ho * p.Wo;
  const unsigned ldcb = (unsigned)p.ldc * XE;
  const bool fast_taps = GATHER && ((p.up == 0 && p.ks * p.ks <= 31 && (p.fast_taps & 1)) || MG);
  unsigned x_off[NXM];
  int x_yx[GATHER ? NXM : 1];   // (iy0 << 16) | (ix0 & 0xffff): input coordinate of tap (0,0); fast taps: validity mask
  pp_static_for<0, NXM>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    const int piece = g == 0 ? wq + 4 * q : XP0 + wq + 4 * q;
    const int m = m0 + piece * 8 + srow;
    const bool ok = m < p.M && (g == 0 ? q < NX0 : q < NX1);
    const int mm = ok ? m : 0;
    // (b, oy, ox) of the output pixel: shifts when the map sizes are powers of two (every SD-1.5 / VAE level) -- the
    // seven integer divisions per lane cost a few thousand cycles per SIMD at the head of a short-K workgroup
    int b, oy, ox;
    if (p.howo_shift >= 0 && p.wo_shift >= 0) {
      b = mm >> p.howo_shift;
      const int rem = mm & (HoWo - 1);
      oy = rem >> p.wo_shift;
      ox = rem & (p.Wo - 1);
    } else {
      b = mm / HoWo;
      const int rem = mm - b * HoWo;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
    unsigned off = (unsigned)((long)b * p.src_batch_stride * XE) + lchunk;
    if constexpr (GATHER) {
      const int y0 = oy * p.stride - pad_y, x0 = ox * p.stride - pad_x;
      if (fast_taps) {
        // no upsample: the address of tap (ky, kx) is the tap-(0,0) address plus a wave-uniform delta, and whether the tap
        // falls inside the image is one bit of a mask made here (row bits x column bits) -- the staging phase then spends
        // 4 vector instructions per piece instead of 12, issue slots it competes for with the partner wave's MFMAs
        int vy = 0, vx = 0;
        for (int k = 0; k < p.ks; ++k) {
          vy |= ((unsigned)(y0 + k) < (unsigned)p.Hi ? 1 : 0) << (k * p.ks);
          vx |= ((unsigned)(x0 + k) < (unsigned)p.Wi ? 1 : 0) << k;
        }
        int msk = 0;
        for (int k = 0; k < p.ks; ++k) msk |= ((vy >> (k * p.ks)) & 1) ? (vx << (k * p.ks)) : 0;
        x_yx[q] = ok ? msk : 0;
        off += (unsigned)(y0 * p.Ws + x0) * ldcb;     // (wraps for border pixels; only in-image taps are fetched)
      } else {
        x_yx[q] = ok ? ((y0 << 16) | (x0 & 0xffff)) : (int)0xC0000000;
      }
    } else {
      off += (unsigned)(((oy * p.stride) >> p.up) * p.Ws + ((ox * p.stride) >> p.up)) * ldcb;
      if (!ok) off = 0xFFFFFFFFu;
    }
    x_off[q] = off;
  });
  unsigned w_off[NWM];
#pragma unroll
  for (int q = 0; q < NWM; ++q) {
    const int n = n0 + ((g == 0 ? 0 : WP0) + wq + 4 * q) * 8 + srow;
    w_off[q] = n < p.Wrows ? (unsigned)(((long)ph4 * p.Wrows + n) * p.ldw * XE) + dchunk * 16u : 0xFFFFFFFFu;
  }
  // K order.  The sum over (tap, channel chunk) can be walked either way; with the TAP innermost (k_tap_inner) the nine
  // taps of a channel chunk re-read the same few input rows back to back, so eight of nine gathers hit the XCD's L2
  // instead of each tap streaming the whole input slice again (PMC: 225 MB fetched per launch against ~50 MB of input
  // with the tap outermost).  The weight tile of (tap, chunk) is the 128-byte run at column tap * Cin + c0 either way.
  // K-walk state of the NEXT tile this wave stages: a plain struct handed around BY VALUE (as by-reference lambda captures
  // mutated inside the merged schedule's compute phase these scalars ended up in scratch, came back as VGPRs, and every
  // LDS-DMA grew a waterfall loop around its scalar offset)
  struct KWalk {
    int ky, kx, c0, ktile;   // filter tap / channel offset / K tile (bf16)
    int u_tap, u_c0;         // fp8, per lane: tap and channel offset of the unit its half of the tile comes from
  };
  KWalk kw;
  kw.ktile = kt_begin;
  kw.u_tap = 0; kw.u_c0 = 0;
  if (p.k_tap_inner) {
    const int taps = p.ks * p.ks;
    const int cc = kt_begin / taps, tap = kt_begin - cc * taps;
    kw.c0 = cc * 64;
    kw.ky = tap / p.ks;
    kw.kx = tap - kw.ky * p.ks;
  } else {
    const int k0 = kt_begin * 64;
    const int tap = k0 / p.Cin;
    kw.c0 = k0 - tap * p.Cin;
    kw.ky = tap / p.ks;
    kw.kx = tap - kw.ky * p.ks;
  }
  const int taps8 = p.ks * p.ks;
  const int adv_q = 2 / taps8, adv_r = 2 - adv_q * taps8;   // fp8: two units further = adv_q chunks + adv_r taps
  if constexpr (FP8) {
    const int u = 2 * kt_begin + (int)(dchunk >> 2);
    const int cc = u / taps8;
    kw.u_tap = u - cc * taps8;
    kw.u_c0 = cc * 64;
  }
  // per-tile staging values, passed BY VALUE from stage_begin to the pieces (as captured variables they are written and
  // read across the "memory"-clobbering fragment-read asm of the merged schedule and end up in scratch)
  struct StageCtx {
    unsigned c0b, k0b;      // scalar byte offsets of the activation channel chunk / the weight K tile
    int live;               // merged schedule: the tile exists (pieces past the K range are issued out of bounds: no
                            // memory traffic, zeros into a slot nobody reads, the vmcnt counts stay fixed)
    int tapbit;             // bf16 fast taps: mask bit of the tap, byte delta of its pixel
    unsigned delta;
    int u_ok;               // fp8 fast taps (per lane): unit inside K, byte delta of its tap + channel offset
    unsigned u_delta;
    int ky, kx, u_tap, u_c0;  // copies for the generic (per-tap bounds arithmetic) address path
  };
  auto x_addr = [&](auto qc, auto ftc, const StageCtx& sc) -> unsigned {
    constexpr int q = decltype(qc)::value;
    constexpr bool FT = decltype(ftc)::value;   // fast taps (compile-time here: one uniform branch per stage() call, not per piece)
    if constexpr (FP8) {
      if constexpr (GATHER) {
        if constexpr (FT) return ((x_yx[q] >> sc.u_tap) & sc.u_ok) ? x_off[q] + sc.u_delta : 0xFFFFFFFFu;
        const int kyl = p.ks == 3 ? (sc.u_tap * 11) >> 5 : 0, kxl = p.ks == 3 ? sc.u_tap - 3 * kyl : 0;
        const int iy = (x_yx[q] >> 16) + kyl, ix = ((x_yx[q] << 16) >> 16) + kxl;
        const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && sc.u_c0 < p.Cin;
        const unsigned pix = (unsigned)((iy >> p.up) * p.Ws + (ix >> p.up));
        return ok ? x_off[q] + pix * ldcb + (unsigned)sc.u_c0 : 0xFFFFFFFFu;
      } else {
        return (sc.u_c0 < p.Cin && x_off[q] != 0xFFFFFFFFu) ? x_off[q] + (unsigned)sc.u_c0 : 0xFFFFFFFFu;
      }
    } else if constexpr (GATHER) {
      if constexpr (FT) return (x_yx[q] & sc.tapbit) ? x_off[q] + sc.delta : 0xFFFFFFFFu;
      const int iy = (x_yx[q] >> 16) + sc.ky, ix = ((x_yx[q] << 16) >> 16) + sc.kx;
      const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      const unsigned pix = (unsigned)((iy >> p.up) * p.Ws + (ix >> p.up));
      return ok ? x_off[q] + pix * ldcb : 0xFFFFFFFFu;
    } else {
      return x_off[q];
    }
  };
  // staging of one K tile = stage_begin (wave-uniform / per-lane tap state), one stage_piece per LDS-DMA piece, stage_end
  // (advance to the next tile).  The phased schedules run the three back to back (stage); the merged one spreads the
  // pieces over the compute phase.
  auto stage_begin = [&](const KWalk& k) -> StageCtx {
    StageCtx sc;
    sc.tapbit = 0; sc.delta = 0u; sc.u_ok = 0; sc.u_delta = 0u;
    sc.ky = k.ky; sc.kx = k.kx; sc.u_tap = k.u_tap; sc.u_c0 = k.u_c0;
    if (fast_taps) {
      if constexpr (FP8) {
        const int kyl = p.ks == 3 ? (k.u_tap * 11) >> 5 : 0, kxl = p.ks == 3 ? k.u_tap - 3 * kyl : 0;
        sc.u_ok = k.u_c0 < p.Cin ? 1 : 0;
        sc.u_delta = (unsigned)(kyl * p.Ws + kxl) * ldcb + (unsigned)k.u_c0;
      } else {
        sc.tapbit = 1 << (k.ky * p.ks + k.kx);
#ifdef AF_LAB_ABLATE
        if ((p.fast_taps & 0x100) && (k.ky | k.kx)) sc.tapbit = 0;   // only tap (0,0) fetches activations
#endif
        sc.delta = (unsigned)(k.ky * p.Ws + k.kx) * ldcb;
      }
    }
    sc.c0b = FP8 ? 0u : (unsigned)k.c0 * 2u;
    sc.k0b = (!FP8 && p.k_tap_inner) ? (unsigned)((k.ky * p.ks + k.kx) * p.Cin + k.c0) * 2u : (unsigned)k.ktile * 128u;
    sc.live = (!MG || k.ktile < kt_begin + KT) ? 1 : 0;
    return sc;
  };
  auto stage_piece = [&](auto qc, auto ftc, char* base, const StageCtx& sc) {
    constexpr int q = decltype(qc)::value;
    auto fix = [&](unsigned a) -> unsigned { if constexpr (MG) return sc.live ? a : 0xFFFFFFFFu; else return a; };
    if (g == 0) {
      if constexpr (q < NX0) lds_dma16(rs_x, base + (wq + 4 * q) * 1024, fix(x_addr(qc, ftc, sc)), sc.c0b);
      else if constexpr (q < NP0) lds_dma16(rs_w, base + XBYTES + (wq + 4 * (q - NX0)) * 1024, fix(w_off[q - NX0]), sc.k0b);
    } else {
      if constexpr (q < NX1) lds_dma16(rs_x, base + (XP0 + wq + 4 * q) * 1024, fix(x_addr(qc, ftc, sc)), sc.c0b);
      else if constexpr (q < NP1) lds_dma16(rs_w, base + XBYTES + (WP0 + wq + 4 * (q - NX1)) * 1024, fix(w_off[q - NX1]), sc.k0b);
    }
  };
  auto kw_next = [&](KWalk k) -> KWalk {
    ++k.ktile;
    if constexpr (FP8) {
      k.u_tap += adv_r;
      k.u_c0 += 64 * adv_q;
      if (k.u_tap >= taps8) { k.u_tap -= taps8; k.u_c0 += 64; }
    } else if (p.k_tap_inner) {
      if (++k.kx >= p.ks) {
        k.kx = 0;
        if (++k.ky >= p.ks) { k.ky = 0; k.c0 += 64; }
      }
    } else {
      k.c0 += 64;
      if (k.c0 >= p.Cin) {
        k.c0 = 0;
        if (++k.kx >= p.ks) { k.kx = 0; ++k.ky; }
      }
    }
    return k;
  };
  auto stage = [&](int slot_off) {
    const StageCtx sc = stage_begin(kw);
    char* base = smem + slot_off;
    auto issue = [&](auto ftc) { pp_static_for<0, NPMAX>([&](auto qc) { stage_piece(qc, ftc, base, sc); }); };
    if constexpr (GATHER && MG) {
      issue(std::true_type{});
    } else if constexpr (GATHER) {
      if (fast_taps) issue(std::true_type{}); else issue(std::false_type{});
    } else {
      issue(std::false_type{});
    }
    kw = kw_next(kw);
  };
  auto wait_keep1 = [&]() { if (g == 0) pp_wait_vm<NP0>(); else pp_wait_vm<NP1>(); };

  // ---- fragments: row (lane & 15) of a 16-row block, chunk ((lane >> 4) + 4 u) ^ (row & 7), u = K half ----
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)((lane >> 4) ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)(((lane >> 4) + 4) ^ (lane & 7)) * 16u;
  const unsigned x_base = (unsigned)((wq * 64 + (lane & 15)) * 128);
  const unsigned w_base = (unsigned)(XBYTES + (g * C::HN + (lane & 15)) * 128);

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  pp_u32x4 xf[MI][2], wf[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i) wf[i][1] = pp_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < MI; ++j) xf[j][1] = pp_u32x4{0u, 0u, 0u, 0u};

  // one half of a compute segment: MFMAs of unit UM, the NI + MI fragment reads of unit UR one behind each of the
  // first MFMAs (in-order issue: reads placed after the MFMAs would not overlap them)
  auto half = [&](int slot_off, auto rc, auto mc) {
    constexpr int ur = decltype(rc)::value, um = decltype(mc)::value;
    const unsigned b = lds0 + (unsigned)slot_off + (ur ? fch1 : fch0);
    pp_static_for<0, NI * MI>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int i = n / MI, j = n % MI;
      if constexpr (n < NI) wf[n][ur] = pp_lds_read128<n * 2048>(b + w_base);
      else if constexpr (n < NI + MI) xf[n - NI][ur] = pp_lds_read128<(n - NI) * 2048>(b + x_base);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][um]),
                                                          __builtin_bit_cast(bf16x8, xf[j][um]), acc[i][j], 0, 0, 0);
      if constexpr (n < NI + MI) __builtin_amdgcn_sched_barrier(0);
    });
  };
  std::integral_constant<int, 0> U0;
  std::integral_constant<int, 1> U1;

  // bias of this lane's output channels (GEGLU: value and gate rows), fetched now so the main loop hides the latency
  const int cl = 4 * (lane >> 4);
  float4 bias_r[NI];
  auto load_bias = [&]() {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bias_r[i] = float4{0.f, 0.f, 0.f, 0.f};
      if (p.bias && p.splitk <= 1) bias_r[i] = *reinterpret_cast<const float4*>(p.bias + n0 + g * C::HN + i * 16 + cl);
    }
  };
  if constexpr (!NS) load_bias();   // (block-ordered schedule: 104 fragment registers in the loop; the bias is fetched after it)

  // LayerNorm consumer: column sums of W * gamma for this lane's channels and mu / rstd of its four rows
  float4 ln_cs[LNMODE == 1 ? NI : 1];
  float ln_mu[LNMODE == 1 ? MI : 1], ln_rs[LNMODE == 1 ? MI : 1];
  if constexpr (LNMODE == 1) {
#pragma unroll
    for (int i = 0; i < NI; ++i) ln_cs[i] = *reinterpret_cast<const float4*>(p.ln_colsum + n0 + g * C::HN + i * 16 + cl);
#pragma unroll
    for (int j = 0; j < MI; ++j) {   // (mu, rstd) per row, finalised by ln_finalize_kernel; consumed in the epilogue only
      const int m = m0 + wq * 64 + j * 16 + (lane & 15);
      float2 st = float2{0.f, 0.f};
      if (m < p.M) st = *reinterpret_cast<const float2*>(p.ln_stats + (long)m * 2);
      ln_mu[j] = st.x;
      ln_rs[j] = st.y;
    }
  }

  // ---- FP8 fragments and compute phase ----
  // Order of a tile's 20 (16) MFMAs: weight block outermost, and the LAST weight block's four MFMAs are held back to the
  // head of the next tile's compute phase, where they cover the latency of that tile's first fragment reads (nothing of
  // a tile can be read before the barrier that opens its phase).  They need the previous tile's activation fragments, so
  // those are double-buffered (xa8 / xb8 alternate per tile).  Reads are issued in the order W0 X0 X1 X2 X3 W1 .. W(NI-1):
  // two blocks up front, then one block (two ds_read_b128) behind each MFMA; every MFMA waits with a counted lgkmcnt for
  // exactly the blocks it needs (LDS reads return in order).
  pp_u32x4 w8[NS ? NI : 1][2], xa8[NS ? MI : 1][2], xb8[NS ? MI : 1][2];
  int wsc8[FP8 ? NI : 1];
  int xsc8 = p.x_scale_e8;
  if constexpr (NS && !FP8) {
#pragma unroll
    for (int i = 0; i < NI; ++i) w8[i][0] = w8[i][1] = pp_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < MI; ++j) xa8[j][0] = xa8[j][1] = xb8[j][0] = xb8[j][1] = pp_u32x4{0u, 0u, 0u, 0u};
  }
  if constexpr (FP8) {
    // the activation scale as a VGPR made HERE: first used inside the loop, the kernel-argument load behind it would get
    // its s_waitcnt lgkmcnt(0) in front of the first MFMA of every other tile, draining the fragment reads just issued
    asm volatile("" : "+v"(xsc8));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      w8[i][0] = w8[i][1] = pp_u32x4{0u, 0u, 0u, 0u};
      wsc8[i] = (int)p.w_scale[n0 + g * C::HN + i * 16 + (lane & 15)];
    }
#pragma unroll
    for (int j = 0; j < MI; ++j) xa8[j][0] = xa8[j][1] = xb8[j][0] = xb8[j][1] = pp_u32x4{0u, 0u, 0u, 0u};
  }
  auto mfma8 = [&](f32x4& c, const pp_u32x4 (&wv)[2], const pp_u32x4 (&xv)[2], int wscale) {
    if constexpr (!FP8) {   // bf16: the two K halves of the 64-value tile (chunks q and q + 4)
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[0]), __builtin_bit_cast(bf16x8, xv[0]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[1]), __builtin_bit_cast(bf16x8, xv[1]), c, 0, 0, 0);
      asm volatile("" : "+v"(c));
      return;
    }
    const pp_i32x8 a = __builtin_shufflevector(__builtin_bit_cast(pp_i32x4, wv[0]), __builtin_bit_cast(pp_i32x4, wv[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    const pp_i32x8 b = __builtin_shufflevector(__builtin_bit_cast(pp_i32x4, xv[0]), __builtin_bit_cast(pp_i32x4, xv[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, wscale, 0, xsc8);
    asm volatile("" : "+v"(c));   // a use at this point: hipcc otherwise SINKS the whole tile's MFMAs below the last wait
  };
  // counted wait that hands the guarded fragment block through (a data dependency: the MFMA cannot be hoisted over it)
  auto wait_block = [&](auto nc, pp_u32x4 (&blk)[2]) {
    constexpr int n = decltype(nc)::value;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(blk[0]), "+v"(blk[1]) : "n"(n) : "memory");
  };
  // (ftc / stage_off: merged schedule only -- the LDS-DMA pieces of tile t + 2 go out one behind every other MFMA)
  // Timing ablations (a separate lab build with -DAF_LAB_ABLATE, scripts/lab/ablate_conv.sh; results are WRONG): bits
  // 4.. of the conv_fast_taps knob: 0x10 no LDS-DMA in the loop, 0x20 no fragment reads, 0x40 no MFMAs, 0x100 see above
#ifdef AF_LAB_ABLATE
  const int lab = p.fast_taps >> 4;
#else
  constexpr int lab = 0;
#endif
  auto cphase8 = [&](int slot_off, pp_u32x4 (&xc)[NS ? MI : 1][2], pp_u32x4 (&xp)[NS ? MI : 1][2], auto ftc, int stage_off,
                     const StageCtx& sc) {
    if constexpr (NS) {
      constexpr int NB = NI + MI;
      char* sbase = smem + stage_off;
      const unsigned b0 = lds0 + (unsigned)slot_off + fch0, b1 = lds0 + (unsigned)slot_off + fch1;
      auto rd_block = [&](auto bc) {
        constexpr int bi = decltype(bc)::value;
        if (lab & 2) return;
        if constexpr (bi == 0) {
          w8[0][0] = pp_lds_read128<0>(b0 + w_base);
          w8[0][1] = pp_lds_read128<0>(b1 + w_base);
        } else if constexpr (bi <= MI) {
          xc[bi - 1][0] = pp_lds_read128<(bi - 1) * 2048>(b0 + x_base);
          xc[bi - 1][1] = pp_lds_read128<(bi - 1) * 2048>(b1 + x_base);
        } else {
          w8[bi - MI][0] = pp_lds_read128<(bi - MI) * 2048>(b0 + w_base);
          w8[bi - MI][1] = pp_lds_read128<(bi - MI) * 2048>(b1 + w_base);
        }
      };
      rd_block(std::integral_constant<int, 0>{});
      rd_block(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      pp_static_for<0, MI * NI>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < MI) {
          if (!(lab & 4)) mfma8(acc[NI - 1][m], w8[NI - 1], xp[m], wsc8[FP8 ? NI - 1 : 0]);   // held back from the previous tile
        } else {
          constexpr int i = (m - MI) / MI, j = (m - MI) % MI;
          constexpr int issued = (2 + m) < NB ? (2 + m) : NB;               // blocks issued before this MFMA
          constexpr int need = i == 0 ? 1 + j : (j == 0 ? MI + i : -1);     // youngest block it reads
          if constexpr (need >= 0) {
            std::integral_constant<int, 2 * (issued - need - 1)> cnt;
            if constexpr (i == 0) wait_block(cnt, xc[j]); else wait_block(cnt, w8[i]);
            if constexpr (i == 0 && j == 0) wait_block(cnt, w8[0]);
          }
          if (!(lab & 4)) mfma8(acc[i][j], w8[i], xc[j], wsc8[FP8 ? i : 0]);
        }
        if constexpr (2 + m < NB) rd_block(std::integral_constant<int, 2 + m>{});
        if constexpr (MG && m >= MI && (m - MI) / 2 < NPMAX) {
          // (the two waves of a SIMD issue their pieces behind alternate MFMAs: group 0 behind the even ones, group 1 odd)
          if (!(lab & 1) && (g == ((m - MI) & 1) || !p.pp_stagger)) {
            if (p.pp_stagger || ((m - MI) & 1) == 0) stage_piece(std::integral_constant<int, (m - MI) / 2>{}, ftc, sbase, sc);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      static_assert(!MG || MI + 2 * (NPMAX - 1) + 1 < MI * NI, "merged schedule: a staging slot behind an MFMA for every piece");
      pp_wait_lgkm0();       // every read of the tile is back (the last weight block included)
    }
  };

  int rd = 0, w0 = SLOT, w1 = 2 * SLOT;       // slots of tiles t, t+1, t+2
  if constexpr (MG) {
    // ---- merged schedule: tiles 0 and 1 in flight (every wave its own pieces), then per tile: own pieces of tile t
    // landed (tile t + 1's stay in flight) -> barrier (tile t complete and visible; everyone done reading tile t - 1) ->
    // compute tile t while staging tile t + 2 into the slot of tile t - 1
    stage(0);
    stage(SLOT);
    auto tile = [&](auto& xc, auto& xp) {
      if (g == 0) pp_wait_vm<NP0>(); else pp_wait_vm<NP1>();
      __builtin_amdgcn_s_barrier();
      // (gathers: fast taps only -- the launcher sends upsampled convolutions to the phased schedule)
      const StageCtx sc = stage_begin(kw);
      cphase8(rd, xc, xp, std::integral_constant<bool, GATHER>{}, w1, sc);
      kw = kw_next(kw);
      const int tmp = rd; rd = w0; w0 = w1; w1 = tmp;
    };
    for (int t = 0; t + 1 < KT; t += 2) {   // (pairs: no branch inside the body, so no register shuffles where paths merge)
      tile(xa8, xb8);
      tile(xb8, xa8);
    }
    if (KT & 1) tile(xa8, xb8);
    pp_wait_vm<0>();          // the out-of-range pieces of tiles KT, KT + 1 (zero fill) before the epilogue reuses the LDS
    if (KT & 1) {
#pragma unroll
      for (int j = 0; j < MI; ++j) mfma8(acc[NI - 1][j], w8[NI - 1], xa8[j], wsc8[FP8 ? NI - 1 : 0]);
    } else {
#pragma unroll
      for (int j = 0; j < MI; ++j) mfma8(acc[NI - 1][j], w8[NI - 1], xb8[j], wsc8[FP8 ? NI - 1 : 0]);
    }
    __builtin_amdgcn_s_barrier();
  }
  // ---- phased schedules, prologue: tile 0 (group 1 also tile 1) in flight; group 1's part of tile 0 landed ----
  if constexpr (!MG) {
  stage(0);
  if (g == 1) {
    if (KT > 1) { stage(SLOT); wait_keep1(); } else pp_wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  if (g == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind
  }
  auto dphase = [&](int t) {
    if (g == 0) {
      if (t + 1 < KT) { stage(w0); wait_keep1(); } else pp_wait_vm<0>();   // own part of tile t landed
    } else {
      if (t + 2 < KT) stage(w1);
    }
    __builtin_amdgcn_s_barrier();
  };
  auto cend = [&](int t) {
    if (g == 1) { if (t + 2 < KT) wait_keep1(); else pp_wait_vm<0>(); }    // own part of tile t+1 landed
    __builtin_amdgcn_s_barrier();
    const int tmp = rd; rd = w0; w0 = w1; w1 = tmp;
  };
  if constexpr (MG) {
  } else if constexpr (NS) {
    auto tile1 = [&](int t, auto& xc, auto& xp) {
      dphase(t);
      __builtin_amdgcn_s_setprio(1);   // (no priorities at all, or the staging phase raised instead: no difference, measured)
      cphase8(rd, xc, xp, std::false_type{}, 0, StageCtx{});
      __builtin_amdgcn_s_setprio(0);
      cend(t);
    };
    int t = 0;
    for (; t + 1 < KT; t += 2) {
      tile1(t, xa8, xb8);
      tile1(t + 1, xb8, xa8);
    }
    if (KT & 1) tile1(t, xa8, xb8);
    // the last tile's held-back MFMAs
    if (KT & 1) {
#pragma unroll
      for (int j = 0; j < MI; ++j) mfma8(acc[NI - 1][j], w8[NI - 1], xa8[j], wsc8[FP8 ? NI - 1 : 0]);
    } else {
#pragma unroll
      for (int j = 0; j < MI; ++j) mfma8(acc[NI - 1][j], w8[NI - 1], xb8[j], wsc8[FP8 ? NI - 1 : 0]);
    }
  } else {
    for (int t = 0; t < KT; ++t) {
      // ---------------- D(t) ----------------
      dphase(t);
      // ---------------- C(t) ----------------
      __builtin_amdgcn_s_setprio(1);
      half(rd, U0, U1);      // (t == 0: MFMAs on the zeroed fragments of "unit -1")
      pp_wait_lgkm0();
      half(rd, U1, U0);
      pp_wait_lgkm0();       // every read of tile t is back: the barrier below releases its slot
      __builtin_amdgcn_s_setprio(0);
      cend(t);
    }
    // trailing half tile (unit 2 KT - 1), registers only
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][1]),
                                                            __builtin_bit_cast(bf16x8, xf[j][1]), acc[i][j], 0, 0, 0);
  }
  if constexpr (!MG) { if (g == 0) __builtin_amdgcn_s_barrier(); }
  if constexpr (NS) load_bias();

  pp_epilogue<BN, LNMODE>(p, acc, bias_r, ln_cs, ln_mu, ln_rs, smem, tid, lane, g, wq, m0, n0, tn, zk);
}

// ---------------------------------------------------------------------------
// Eight-wave 3x3 / stride 1 / pad 1 convolution with an LDS-resident input HALO (bf16, 256 x 160 tile, merged schedule).
//
// The implicit-GEMM kernel above stages every input pixel of its tile nine times (once per filter tap): 32 of the 52
// LDS-DMA pieces of a K step are activations, and it is the number of those pieces a wave has to issue between its MFMAs,
// not the MFMA rate, that sets the step time (timing ablations: scripts/lab/ablate_conv.sh).  Here a workgroup's 256
// output pixels are R = 256 / Wo whole rows of one image; for each 64-channel chunk their (R + 2) x (Wo + 2) input halo
// is staged ONCE (<= 50 pieces, spread over the nine tap steps of the PREVIOUS chunk: at most one per wave and step) and
// the nine taps read their MFMA operand from it at shifted pixel addresses; only the 160 x 64 weight tile streams per
// step (20 pieces).  A wave issues 2-4 pieces per step instead of 6-7 and the L2 -> LDS traffic halves.
//   LDS (all 160 KiB): two halo buffers of 400 pixels x 128 B (chunk c / chunk c + 1) + three weight slots of 20 KiB.
//   Halo pixel hp = hy * (Wo + 2) + hx holds input pixel (row0 - 1 + hy, hx - 1); its 16-byte chunk c sits in slot
//   c ^ (hp & 7) (the swizzle of the other kernels: 16 consecutive pixels of a row are conflict-free for ds_read_b128).
//   The per-lane source address of every halo piece is computed once (validity included: pixels outside the image are
//   out-of-range addresses = zeros); per chunk only the scalar channel offset changes.
//   Output rows, accumulators and the epilogue are those of conv_gemm_pp_kernel (the tile is 256 consecutive pixels in
//   (b, y, x) raster order), K is walked (channel chunk, tap) and a K slice (split-K) is a whole number of chunks.
// ---------------------------------------------------------------------------
#ifdef AF_LAB_ABLATE
// lab builds only: per-workgroup s_memrealtime stamps (100 MHz) of the halo kernel -- entry, loop start, loop end, stores
// landed -- and the XCC / CU the workgroup ran on (scripts/lab/halo_stamps.py)
__device__ unsigned long long g_lab_stamps[2048 * 5];
extern "C" int af_lab_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lab_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
__device__ __forceinline__ unsigned long long lab_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#endif
struct Halo8Cfg {
  static constexpr int BN = 160;
  static constexpr int HPIX = 400, HBUF = HPIX * 128, WBYTES = BN * 128;
  static constexpr int W_BASE = 2 * HBUF, LDS_BYTES = 2 * HBUF + 3 * WBYTES;   // 163840 = the whole LDS
  static constexpr int NHQ = 7;   // halo pieces per wave and chunk, at most (50 pieces over 8 waves)
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__global__ __launch_bounds__(512) void conv3x3_halo8_kernel(const ConvGemmParams p) {
  using C = PpCfg<160>;
  using H = Halo8Cfg;
  typedef bf16 T;
  constexpr int BN = 160, NI = C::NI, MI = C::MI, NHQ = H::NHQ;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wid >> 2, wq = wid & 3;
  const int ntm = p.M / 256, ntn = p.N / BN;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, ntm, ntn, p.group_m, tm, tn);
  const int m0 = tm * 256, n0 = tn * BN;
  const int zk = blockIdx.z;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.src)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.W)), 0, (int)0xFFFFFFF0u, 0x00020000);

  // K range: steps s = (chunk, tap), nine per chunk; a slice is a whole number of chunks (the launcher checks)
  const int KT_all = p.K / 64;
  const int kt_per = (KT_all + p.splitk - 1) / p.splitk;
  const int kt_begin = zk * kt_per;
#ifdef AF_LAB_ABLATE
  // timing ablations (wrong results; scripts/lab/ablate_conv.sh halo): bits 4.. of the conv_fast_taps knob: 1 no LDS-DMA in
  // the loop, 2 no fragment reads, 4 no MFMAs, 8 no epilogue, 16 one K step only, 32 no prologue staging
  const int lab = p.fast_taps >> 4;
  const int KT = (lab & 16) ? 1 : min(KT_all, kt_begin + kt_per) - kt_begin;
#else
  constexpr int lab = 0;
  const int KT = min(KT_all, kt_begin + kt_per) - kt_begin;
#endif
  const int chunk0 = kt_begin / 9;
#ifdef AF_LAB_ABLATE
  unsigned long long st0 = 0, st1 = 0, st2 = 0;
  if (lab & 64) st0 = lab_now();
#endif

  // ---- geometry: the tile is R rows of image b starting at row y0 ----
  const int Wo = p.Wo, HW = Wo + 2, wsh = p.wo_shift;
  const int R = 256 >> wsh;
  const int HP = (R + 2) * HW;                  // halo pixels (<= 400)
  const int img = m0 >> p.howo_shift;
  const int y0 = (m0 & (p.Ho * Wo - 1)) >> wsh;
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  const unsigned ldcb = (unsigned)p.ldc * 2u;

  // halo pieces of this wave: piece wid + 8 q covers halo pixels 8 * piece .. + 7 (lane >> 3 = pixel, lane & 7 = slot)
  unsigned h_off[NHQ];
  const int nhq = (H::HPIX / 8 - 1 - wid) / 8 + 1;   // pieces that exist for this wave in a 400-pixel buffer (7 or 6)
#pragma unroll
  for (int q = 0; q < NHQ; ++q) {
    const int hp = (wid + 8 * q) * 8 + srow;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = y0 - 1 + hy, ix = hx - 1;
    const bool ok = hp < HP && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    // (nearest-2x upsampling folded in: the halo lives in the UPSAMPLED image, each of its pixels is fetched from the source
    // pixel it replicates -- the per-lane address is made once per kernel either way)
    h_off[q] = ok ? (unsigned)((long)img * p.src_batch_stride * 2) + (unsigned)((iy >> p.up) * p.Ws + (ix >> p.up)) * ldcb + dchunk * 16u
                  : 0xFFFFFFFFu;
  }
  // weight pieces of this wave: piece wid + 8 q, q < 3 (waves 0-3) or 2 (waves 4-7)
  unsigned w_off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int n = n0 + (wid + 8 * q) * 8 + srow;
    w_off[q] = (n < p.Wrows && (wid + 8 * q) < BN / 8) ? (unsigned)((long)n * p.ldw * 2) + dchunk * 16u : 0xFFFFFFFFu;
  }
  const int nwq = g == 0 ? 3 : 2;

  // ---- fragment addressing ----
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)((lane >> 4) ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)(((lane >> 4) + 4) ^ (lane & 7)) * 16u;
  const unsigned w_base = (unsigned)((g * C::HN + (lane & 15)) * 128);
  // halo pixel of tap (0, 0) for this lane's output pixel of block j: tile pixel t = wq * 64 + j * 16 + (lane & 15)
  int xhp[MI];
#pragma unroll
  for (int j = 0; j < MI; ++j) {
    const int t = wq * 64 + j * 16 + (lane & 15);
    xhp[j] = (t >> wsh) * HW + (t & (Wo - 1));
  }
  const unsigned qsel = (unsigned)(lane >> 4);

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  pp_u32x4 w8[NI][2], xa8[MI][2], xb8[MI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i) w8[i][0] = w8[i][1] = pp_u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < MI; ++j) xa8[j][0] = xa8[j][1] = xb8[j][0] = xb8[j][1] = pp_u32x4{0u, 0u, 0u, 0u};
  auto mfma2 = [&](f32x4& c, const pp_u32x4 (&wv)[2], const pp_u32x4 (&xv)[2]) {
    if (lab & 4) return;
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[0]), __builtin_bit_cast(bf16x8, xv[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[1]), __builtin_bit_cast(bf16x8, xv[1]), c, 0, 0, 0);
    asm volatile("" : "+v"(c));
  };
  auto wait_block = [&](auto nc, pp_u32x4 (&blk)[2]) {
    constexpr int n = decltype(nc)::value;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(blk[0]), "+v"(blk[1]) : "n"(n) : "memory");
  };

  // ---- staging (everything by value: see KWalk in conv_gemm_pp_kernel) ----
  struct Step { int tap, chunk; };             // the step being STAGED (weights) / the chunk whose halo is being staged
  bool in_loop = false;
  auto stage_w = [&](auto qc, int wslot, Step st, int live) {
    constexpr int q = decltype(qc)::value;
    if ((lab & 1) && in_loop) return;
    if ((lab & 32) && !in_loop) return;
    const unsigned k0b = (unsigned)(st.tap * p.Cin + st.chunk * 64) * 2u;
    const unsigned a = live ? w_off[q] : 0xFFFFFFFFu;
    lds_dma16(rs_w, smem + H::W_BASE + wslot * H::WBYTES + (wid + 8 * q) * 1024, a, k0b);
  };
  auto stage_h = [&](auto qc, int hbuf, int chunk, int live) {
    constexpr int q = decltype(qc)::value;
    if ((lab & 1) && in_loop) return;
    if ((lab & 32) && !in_loop) return;
    const unsigned a = live ? h_off[q] : 0xFFFFFFFFu;
    lds_dma16(rs_x, smem + hbuf * H::HBUF + (wid + 8 * q) * 1024, a, (unsigned)chunk * 128u);
  };
  const int nchunks_all = p.Cin >> 6;
  const int step_end = kt_begin + KT;          // first step (global index) that does not exist
  const bool stg = p.pp_stagger != 0;

  // one K step: compute (chunk, tap) from halo buffer hb and weight slot ws while staging the weights of step + 2 into
  // slot ws2 and, during the first nhq taps, one halo piece of chunk + 1 into the other halo buffer
  auto step = [&](int s_glob, int tap, int hb, int ws, int ws2, pp_u32x4 (&xc)[MI][2], pp_u32x4 (&xp)[MI][2]) {
    constexpr int NB = NI + MI;
    const int chunk = s_glob / 9;               // (scalar division by a constant)
    // what this step stages
    const int s2 = s_glob + 2;
    Step st2;
    st2.chunk = s2 / 9;
    st2.tap = s2 - st2.chunk * 9;
    const int live_w = s2 < step_end ? 1 : 0;
    const int live_h = ((chunk + 1) * 9 < step_end && chunk + 1 < nchunks_all) ? 1 : 0;
    // operand addresses of this tap
    const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
    const int tapoff = ky * HW + kx;
    unsigned xa0[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) {
      const unsigned hp = (unsigned)(xhp[j] + tapoff);
      xa0[j] = lds0 + (unsigned)(hb * H::HBUF) + (hp << 7) + (((qsel ^ hp) & 7u) << 4);
    }
    const unsigned wb0 = lds0 + (unsigned)(H::W_BASE + ws * H::WBYTES) + w_base + fch0;
    const unsigned wb1 = lds0 + (unsigned)(H::W_BASE + ws * H::WBYTES) + w_base + fch1;
    auto rd_block = [&](auto bc) {
      constexpr int bi = decltype(bc)::value;
      if (lab & 2) return;
      if constexpr (bi == 0) {
        w8[0][0] = pp_lds_read128<0>(wb0);
        w8[0][1] = pp_lds_read128<0>(wb1);
      } else if constexpr (bi <= MI) {
        xc[bi - 1][0] = pp_lds_read128<0>(xa0[bi - 1]);
        xc[bi - 1][1] = pp_lds_read128<0>(xa0[bi - 1] ^ 64u);
      } else {
        w8[bi - MI][0] = pp_lds_read128<(bi - MI) * 2048>(wb0);
        w8[bi - MI][1] = pp_lds_read128<(bi - MI) * 2048>(wb1);
      }
    };
    rd_block(std::integral_constant<int, 0>{});
    rd_block(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    pp_static_for<0, MI * NI>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m < MI) {
        mfma2(acc[NI - 1][m], w8[NI - 1], xp[m]);          // held back from the previous step
      } else {
        constexpr int i = (m - MI) / MI, j = (m - MI) % MI;
        constexpr int issued = (2 + m) < NB ? (2 + m) : NB;
        constexpr int need = i == 0 ? 1 + j : (j == 0 ? MI + i : -1);
        if constexpr (need >= 0) {
          std::integral_constant<int, 2 * (issued - need - 1)> cnt;
          if constexpr (i == 0) wait_block(cnt, xc[j]); else wait_block(cnt, w8[i]);
          if constexpr (i == 0 && j == 0) wait_block(cnt, w8[0]);
        }
        mfma2(acc[i][j], w8[i], xc[j]);
      }
      if constexpr (2 + m < NB) rd_block(std::integral_constant<int, 2 + m>{});
      // staging: weight pieces behind MFMAs 4, 6, (8) and one halo piece behind MFMA 10 for waves 0-3; their SIMD partners
      // (waves 4-7) one MFMA later each, so that the two waves of a SIMD do not stall on their LDS-DMA issue together
      if constexpr (m == MI) { if (g == 0 || !stg) stage_w(std::integral_constant<int, 0>{}, ws2, st2, live_w); }
      if constexpr (m == MI + 1) { if (g == 1 && stg) stage_w(std::integral_constant<int, 0>{}, ws2, st2, live_w); }
      if constexpr (m == MI + 2) { if (g == 0 || !stg) stage_w(std::integral_constant<int, 1>{}, ws2, st2, live_w); }
      if constexpr (m == MI + 3) { if (g == 1 && stg) stage_w(std::integral_constant<int, 1>{}, ws2, st2, live_w); }
      if constexpr (m == MI + 4) { if (g == 0) stage_w(std::integral_constant<int, 2>{}, ws2, st2, live_w); }
      if constexpr (m == MI + 6 || m == MI + 7) {
        if (tap < nhq && (stg ? g == (m - MI - 6) : m == MI + 6)) {
          switch (tap) {
            case 0: stage_h(std::integral_constant<int, 0>{}, hb ^ 1, chunk + 1, live_h); break;
            case 1: stage_h(std::integral_constant<int, 1>{}, hb ^ 1, chunk + 1, live_h); break;
            case 2: stage_h(std::integral_constant<int, 2>{}, hb ^ 1, chunk + 1, live_h); break;
            case 3: stage_h(std::integral_constant<int, 3>{}, hb ^ 1, chunk + 1, live_h); break;
            case 4: stage_h(std::integral_constant<int, 4>{}, hb ^ 1, chunk + 1, live_h); break;
            case 5: stage_h(std::integral_constant<int, 5>{}, hb ^ 1, chunk + 1, live_h); break;
            default: stage_h(std::integral_constant<int, 6>{}, hb ^ 1, chunk + 1, live_h); break;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    pp_wait_lgkm0();
  };

  // ---- prologue: halo of the first chunk, weights of steps 0 and 1 ----
  pp_static_for<0, NHQ>([&](auto qc) {
    if (decltype(qc)::value < nhq) stage_h(qc, 0, chunk0, 1);
  });
  {
    Step s0{0, chunk0}, s1{1, chunk0};
    stage_w(std::integral_constant<int, 0>{}, 0, s0, 1);
    stage_w(std::integral_constant<int, 1>{}, 0, s0, 1);
    if (g == 0) stage_w(std::integral_constant<int, 2>{}, 0, s0, 1);
    stage_w(std::integral_constant<int, 0>{}, 1, s1, 1);
    stage_w(std::integral_constant<int, 1>{}, 1, s1, 1);
    if (g == 0) stage_w(std::integral_constant<int, 2>{}, 1, s1, 1);
  }
  // vmcnt at the head of a step: everything but what the PREVIOUS step issued (its nwq weight pieces + its halo piece)
  auto wait_head = [&](int prev_halo) {
    const int n = nwq + prev_halo;
    if (n == 2) pp_wait_vm<2>(); else if (n == 3) pp_wait_vm<3>(); else pp_wait_vm<4>();
  };
  int tap = 0, hb = 0, ws = 0, ws1 = 1, ws2 = 2, s_glob = kt_begin;
  int prev_halo = 0;                            // the prologue's last issue is a weight set: keep nwq in flight
  auto one = [&](pp_u32x4 (&xc)[MI][2], pp_u32x4 (&xp)[MI][2]) {
    wait_head(prev_halo);
    __builtin_amdgcn_s_barrier();
    step(s_glob, tap, hb, ws, ws2, xc, xp);
    prev_halo = tap < nhq ? 1 : 0;
    ++s_glob;
    if (++tap == 9) { tap = 0; hb ^= 1; }
    const int t3 = ws; ws = ws1; ws1 = ws2; ws2 = t3;
  };
  int t = 0;
  in_loop = true;
#ifdef AF_LAB_ABLATE
  if (lab & 64) st1 = lab_now();
#endif
  for (; t + 1 < KT; t += 2) {
    one(xa8, xb8);
    one(xb8, xa8);
  }
  if (KT & 1) one(xa8, xb8);
  pp_wait_vm<0>();
  if (KT & 1) {
#pragma unroll
    for (int j = 0; j < MI; ++j) mfma2(acc[NI - 1][j], w8[NI - 1], xa8[j]);
  } else {
#pragma unroll
    for (int j = 0; j < MI; ++j) mfma2(acc[NI - 1][j], w8[NI - 1], xb8[j]);
  }
  __builtin_amdgcn_s_barrier();
#ifdef AF_LAB_ABLATE
  if (lab & 64) st2 = lab_now();
#endif
  if (lab & 8) {   // keep the accumulators alive behind a condition that never holds
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678f) reinterpret_cast<float*>(p.out)[tid] = sum;
    return;
  }

  const int cl = 4 * (lane >> 4);
  float4 bias_r[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    bias_r[i] = float4{0.f, 0.f, 0.f, 0.f};
    if (p.bias && p.splitk <= 1) bias_r[i] = *reinterpret_cast<const float4*>(p.bias + n0 + g * C::HN + i * 16 + cl);
  }
  float4 ln_cs[1];
  float ln_mu[1], ln_rs[1];
  pp_epilogue<160, 0>(p, acc, bias_r, ln_cs, ln_mu, ln_rs, smem, tid, lane, g, wq, m0, n0, tn, zk);
#ifdef AF_LAB_ABLATE
  if (lab & 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long st3 = lab_now();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (tid == 0 && blockIdx.x < 2048 && blockIdx.z == 0) {
      unsigned long long* o = g_lab_stamps + (long)blockIdx.x * 5;
      o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3; o[4] = ((unsigned long long)xcc << 32) | hwid;
    }
  }
#endif
}

// ---------------------------------------------------------------------------
// Row-panel GEGLU GEMM for K = 320 (FeedForward.net[0] of the 64x64-level transformers: [65536, 320] x [2560, 320]^T, 5 per
// forward): a workgroup owns 256 rows for ALL of N.
//
// On the tiled kernel these launches are 5120 workgroups of five K steps each, and the prologue / epilogue around so short
// a loop is most of their time (~500 TFLOP/s).  Here the activation rows never touch LDS: each wave keeps its 32 rows x 320
// K values as MFMA B fragments in 80 VGPRs for the whole launch, and only the weight tiles stream (128 packed columns x
// 64 K = 16 pieces per step, two per wave) through a three-slot LDS ring that runs on across the twenty column tiles --
// no pipeline refill between tiles.  Per step a wave issues 2 LDS-DMA pieces, 16 fragment reads and 32 MFMAs; after
// every fifth step the 32 x 64 GEGLU outputs of the wave are transposed through a wave-private LDS tile and leave as four
// 16-byte buffer stores of whole 128-byte row segments (8-byte stores straight from the accumulators -- 32 bytes per row and
// instruction -- made the stores, not the GELU arithmetic, the cost of the epilogue).  The stores are ALWAYS issued -- rows
// past M by an out-of-range offset -- so the vmcnt arithmetic of the staging stays exact: stores and LDS-DMA share that
// counter on gfx9.
// LNMODE 1: LayerNorm-consumer epilogue as conv_gemm_pp_kernel (the rows are the un-normalised activations).
// ---------------------------------------------------------------------------
// Generalised (round 2, later): GEGLU = false runs the other K = 320 GEMMs of the 64x64-level transformers on the same
// structure with 160-column tiles -- proj_in / to_q,k,v / attn2.to_q / to_out / proj_out -- with the LayerNorm-consumer
// (LNMODE 1) or statistics-producer (LNMODE 2) epilogue and the residual added in the accumulator layout (one rounding, as
// the tiled kernel).
// KC_ = 10, MJ_ = 1: the K = 640 form for the 32x32-level transformers (16 rows per wave keep the fragment registers at 80;
// 128-row panels, and the column tiles of a panel split over gridDim.y workgroups so that 16384 rows still fill 256 CUs).
// D_ = prefetch distance of the weight ring in K steps (the ring has D_ + 1 slots).  Round 2 shipped D = 2: a step then
// waits for an LDS-DMA issued two steps earlier, and with 16-32 MFMAs per wave and step (0.12-0.25 us of matrix time) the
// ~1.1 us a DMA takes from issue to landing set the step time (0.55 us at K = 640).  Where the LDS allows it the ring is
// now deeper: GEGLU forms 6 slots (D = 5), plain forms with 16 rows per wave 5 slots (D = 4); the plain K = 320 form keeps
// D = 2 (its 32-row output transposition tiles leave room for three 20 KB slots only).  D <= KC: at most one tile's output
// stores are then younger than the pieces a wait covers.
template <bool GEGLU, int KC_ = 5, int MJ_ = 2, int D_ = 2> struct RowPanelCfgT {
  static constexpr int KC = KC_, MJ = MJ_, K = KC * 64, D = D_, NS = D_ + 1;
  static constexpr int WROWS = 16 * MJ, BM = 8 * WROWS;             // rows per wave / per workgroup
  static constexpr int BN = GEGLU ? 128 : 160;
  static constexpr int WBYTES = BN * 128, WPIECES = BN / 8;
  static constexpr int NIW = BN / 16;              // weight blocks per column tile
  static constexpr int OCOLS = GEGLU ? BN / 2 : BN;                 // output columns per tile
  static constexpr int OPITCH = OCOLS * 2 + 16, OBYTES = WROWS * OPITCH;   // per-wave output transposition tile (bf16)
  static constexpr int LDS_BYTES = NS * WBYTES + 8 * OBYTES;
  static constexpr int CPR = OCOLS / 8;            // 16-byte chunks per output row segment
  static constexpr int NST = WROWS * CPR / 64;     // 16-byte buffer stores per lane and column tile
  static_assert(LDS_BYTES <= 160 * 1024 && (WROWS * CPR) % 64 == 0 && (MJ * KC == 10 || (MJ == 1 && KC == 20)) && D >= 2 && D <= KC,
                "row-panel LDS / store mapping / 80 (K = 1280, 16 rows per wave: 160) fragment registers / ring depth");
};
typedef RowPanelCfgT<true> RowPanelCfg;

template <bool GEGLU, int LNMODE, bool RES, int KC_ = 5, int MJ_ = 2, int D_ = 2>
__global__ __launch_bounds__(512) void rowpanel_kernel(const ConvGemmParams p) {
  using C = RowPanelCfgT<GEGLU, KC_, MJ_, D_>;
  typedef bf16 T;
  constexpr int KC = C::KC, NIW = C::NIW, MJ = C::MJ, NST = C::NST, CPR = C::CPR, D = C::D, NS = C::NS;
  static_assert(!(GEGLU && (RES || LNMODE == 2)), "GEGLU: plain or LayerNorm-consumer epilogue, no residual");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * C::BM;
  const int r0 = m0 + wid * C::WROWS;
  const int ntn_all = p.N / C::BN;                 // column tiles (GEGLU: value | gate interleaved in 16-row groups)
  const int nt_begin = (int)((long)blockIdx.y * ntn_all / gridDim.y), nt_end = (int)((long)(blockIdx.y + 1) * ntn_all / gridDim.y);
  const int ntn = nt_end - nt_begin;               // ... of this workgroup
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.src)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.W)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(p.out), 0, (int)0xFFFFFFF0u, 0x00020000);

  // ---- the bias (and, LayerNorm consumer, the column sums) of this workgroup's columns -> LDS, once.  Fetched from memory in
  // every tile's epilogue they were the YOUNGEST vector-memory operations of the wave: waiting for them waits for every LDS-DMA
  // piece of the D steps in flight in front of them (vmcnt counts in order) -- one memory latency per column tile with nothing
  // to compute (round 4: 20 tiles per workgroup in the GEGLU forms) ----
  float* vecs = reinterpret_cast<float*>(smem + C::LDS_BYTES);
  const int nvcol = ntn * C::BN;                    // columns of this workgroup (<= 6144: three float4 per thread and vector)
  float4 vb[3], vc[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {                     // (issued here, written to LDS behind the row loads below: one memory round trip)
    const int c4 = tid + 512 * r;
    const int col = nt_begin * C::BN + 4 * c4;
    vb[r] = vc[r] = float4{0.f, 0.f, 0.f, 0.f};
    if (c4 < nvcol / 4) {
      if (p.bias) vb[r] = *reinterpret_cast<const float4*>(p.bias + col);
      if constexpr (LNMODE == 1) vc[r] = *reinterpret_cast<const float4*>(p.ln_colsum + col);
    }
  }

  // ---- the wave's 32 activation rows as MFMA B fragments: block j, K chunk kc, half u -> k = 64 kc + 32 u + 8 (lane >> 4) ----
  pp_u32x4 xr[MJ][KC][2];
  // store mapping: chunk c = lane + 64 i of the wave's 32 x CPR chunks -> row c / CPR, 16-byte chunk c % CPR of its segment
  unsigned orow[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int c = lane + 64 * i;
    const int row = c / CPR, ch = c - row * CPR;
    const int m = r0 + row;
    orow[i] = m < p.M ? (unsigned)((long)m * p.ldo * 2) + (unsigned)ch * 16u : 0xFFFFFFFFu;
  }
  float ln_mu[LNMODE == 1 ? MJ : 1], ln_rs[LNMODE == 1 ? MJ : 1];
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int m = r0 + j * 16 + (lane & 15);
    const bool ok = m < p.M;
    const unsigned xo = ok ? (unsigned)((long)m * p.ldc * 2) + (unsigned)(lane >> 4) * 16u : 0xFFFFFFFFu;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        xr[j][kc][u] = __builtin_bit_cast(pp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xo, (kc * 64 + u * 32) * 2, 0));
    if constexpr (LNMODE == 1) {
      float2 st = float2{0.f, 0.f};
      if (ok && p.ln_parts_n > 0) {
        // un-finalised statistics: ln_stats = the producer's partial sums [parts][M][2]; this workgroup owns its rows for all
        // of N, so it sums them itself (same order and arithmetic as ln_finalize_kernel) -- no finalize launch in between
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < p.ln_parts_n; ++q) {
          const float2 pq2 = *reinterpret_cast<const float2*>(p.ln_stats + ((long)q * p.M + m) * 2);
          s1 += pq2.x;
          s2 += pq2.y;
        }
        const float mu = s1 * p.ln_inv_count;
        const float var = fmaxf(s2 * p.ln_inv_count - mu * mu, 0.f);
        st = float2{mu, __builtin_amdgcn_rsqf(var + p.ln_eps)};
      } else if (ok) {
        st = *reinterpret_cast<const float2*>(p.ln_stats + (long)m * 2);
      }
      ln_mu[j] = st.x;
      ln_rs[j] = st.y;
    }
  }

#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int c4 = tid + 512 * r;
    if (c4 < nvcol / 4) {
      *reinterpret_cast<float4*>(vecs + 4 * c4) = vb[r];
      if constexpr (LNMODE == 1) *reinterpret_cast<float4*>(vecs + nvcol + 4 * c4) = vc[r];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (visible to the other waves behind the first step's barrier)

  if (p.gn_ab) {
    // GroupNorm of the consumer's input (see ConvGemmParams::gn_ab), after every row load of the wave has been issued: the
    // panel lies inside one sample, so all rows share the (scale, shift) vectors of the channels this lane holds
    const float* ga = p.gn_ab + (long)(m0 / p.gn_hw) * 2 * C::K + 8 * (lane >> 4);
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* gp = ga + kc * 64 + u * 32;
        float sc[8], sh[8];
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(gp);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(gp + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(gp + C::K);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(gp + C::K + 4);
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          Vec16<T> v;
          v.u = __builtin_bit_cast(uint4, xr[j][kc][u]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v.e[e] = from_f32<T>(fmaf(to_f32<T>(v.e[e]), sc[e], sh[e]));
          xr[j][kc][u] = __builtin_bit_cast(pp_u32x4, v.u);
        }
      }
  }

  // ---- weight staging: tile (nt, kc) = rows nt * BN .. of W, bytes kc * 128 .. + 127 of each; piece = 8 rows ----
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  constexpr int NWQ = (C::WPIECES + 7) / 8;        // pieces per wave (2; 3 for waves 0-3 at BN = 160)
  const int nwq = (C::WPIECES - wid + 7) / 8;
  unsigned w_off[NWQ];
#pragma unroll
  for (int q = 0; q < NWQ; ++q) w_off[q] = (unsigned)(((wid + 8 * q) * 8 + srow) * p.ldw * 2) + dchunk * 16u;
  const int T_all = ntn * KC;
  const unsigned tile_stride = (unsigned)(C::BN * p.ldw * 2);
  auto stage = [&](int t, int slot) {              // (by value: see KWalk)
    const int ntl = t / KC, kc = t - ntl * KC;
    const bool live = t < T_all;
    const unsigned so = (unsigned)(nt_begin + ntl) * tile_stride + (unsigned)kc * 128u;
#pragma unroll
    for (int q = 0; q < NWQ; ++q)
      if (q < 2 || wid + 8 * q < C::WPIECES)
        lds_dma16(rs_w, smem + slot * C::WBYTES + (wid + 8 * q) * 1024, live ? w_off[q] : 0xFFFFFFFFu, so);
  };

  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)((lane >> 4) ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)(((lane >> 4) + 4) ^ (lane & 7)) * 16u;
  const unsigned w_base = (unsigned)((lane & 15) * 128);
  const int cl = 4 * (lane >> 4);

#ifdef AF_LAB_ABLATE
  const int lab = p.fast_taps >> 4;   // timing ablations (wrong results): 1 no LDS-DMA, 2 no fragment reads, 4 no MFMAs, 8 no epilogue
#else
  constexpr int lab = 0;
#endif
  char* otile = smem + NS * C::WBYTES + wid * C::OBYTES;
  f32x4 acc[NIW][MJ];
  pp_u32x4 wf[NIW][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NIW; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  auto wait_block = [&](auto nc, pp_u32x4 (&blk)[2]) {
    constexpr int n = decltype(nc)::value;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(blk[0]), "+v"(blk[1]) : "n"(n) : "memory");
  };
  // one K step of one column tile: weight blocks read two ahead of their MFMAs, counted waits; DMA for step t + D
  auto step = [&](auto kcc, int slot, int t, int stage_slot) {
    constexpr int kc = decltype(kcc)::value;
    const unsigned b0 = lds0 + (unsigned)(slot * C::WBYTES) + w_base + fch0;
    const unsigned b1 = lds0 + (unsigned)(slot * C::WBYTES) + w_base + fch1;
    auto rd = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (lab & 2) return;
      wf[i][0] = pp_lds_read128<i * 2048>(b0);
      wf[i][1] = pp_lds_read128<i * 2048>(b1);
    };
    rd(std::integral_constant<int, 0>{});
    rd(std::integral_constant<int, 1>{});
    rd(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    pp_static_for<0, NIW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int issued = (i + 3) < NIW ? (i + 3) : NIW;          // blocks issued before block i's MFMAs
      wait_block(std::integral_constant<int, 2 * (issued - i - 1)>{}, wf[i]);
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        if (lab & 4) continue;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][0]), __builtin_bit_cast(bf16x8, xr[j][kc][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][1]), __builtin_bit_cast(bf16x8, xr[j][kc][1]), acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(acc[i][j]));
      }
      if constexpr (i + 3 < NIW) rd(std::integral_constant<int, i + 3>{});
      if constexpr (i == 1) { if (!(lab & 1)) stage(t + D, stage_slot); }
      __builtin_amdgcn_sched_barrier(0);
    });
    pp_wait_lgkm0();
  };

  // ---- main loop: steps t = (column tile, K chunk); steps 0 .. D-1 of the ring in flight; step t sits in slot t % NS and
  // the DMA for step t + D goes to slot (t + D) % NS = the slot step t - 1 has just left (every wave is past this step's barrier)
#pragma unroll
  for (int i = 0; i < D; ++i) stage(i, i);
  int cur = 0;
  int after_epi = 0;                               // heads that still see the previous tile's stores in the counter
  for (int ntl = 0; ntl < ntn; ++ntl) {
    const int nt = nt_begin + ntl;
    pp_static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      // own pieces of this step landed: everything but the pieces of the next D - 1 steps (and, for D heads after an epilogue,
      // its NST stores, which are YOUNGER than the pieces waited for).  (BN = 160: waves 0-3 stage three pieces a step.)
      if (after_epi > 0) {
        if (nwq == 3) pp_wait_vm<3 * (D - 1) + NST>(); else pp_wait_vm<2 * (D - 1) + NST>();
        --after_epi;
      } else {
        if (nwq == 3) pp_wait_vm<3 * (D - 1)>(); else pp_wait_vm<2 * (D - 1)>();
      }
      __builtin_amdgcn_s_barrier();
      const int prev = cur == 0 ? NS - 1 : cur - 1;
      step(kcc, cur, ntl * KC + kc, prev);
      cur = cur + 1 == NS ? 0 : cur + 1;
    });
    // ---- epilogue of the column tile (every bias / column-sum / residual vector is fetched first: one memory round trip) ----
    if (lab & 8) continue;
    const int ncol = nt * C::OCOLS;                // first output column of the tile
    if constexpr (GEGLU) {
      float4 bvec[NIW], cvec[LNMODE == 1 ? NIW : 1];
      {
        const unsigned va = lds0 + (unsigned)C::LDS_BYTES + (unsigned)((ntl * C::BN + cl) * 4);
        pp_u32x4 tb[NIW], tc[LNMODE == 1 ? NIW : 1];
        pp_static_for<0, NIW>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          tb[i] = pp_lds_read128<i * 64>(va);
          if constexpr (LNMODE == 1) tc[i] = pp_lds_read128<i * 64>(va + (unsigned)nvcol * 4u);
        });
        pp_wait_lgkm0();
#pragma unroll
        for (int i = 0; i < NIW; ++i) {
          asm volatile("" : "+v"(tb[i]));
          bvec[i] = __builtin_bit_cast(float4, tb[i]);
          if constexpr (LNMODE == 1) { asm volatile("" : "+v"(tc[i])); cvec[i] = __builtin_bit_cast(float4, tc[i]); }
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < NIW / 2; ++k2) {
        const float* bvp = reinterpret_cast<const float*>(&bvec[2 * k2]);
        const float* bgp = reinterpret_cast<const float*>(&bvec[2 * k2 + 1]);
        const float* cvp = reinterpret_cast<const float*>(&cvec[LNMODE == 1 ? 2 * k2 : 0]);
        const float* cgp = reinterpret_cast<const float*>(&cvec[LNMODE == 1 ? 2 * k2 + 1 : 0]);
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          Quad<T> o;
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            f32x2 val, gat;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              if constexpr (LNMODE == 1) {
                val[h] = (acc[2 * k2][j][e + h] - ln_mu[j] * cvp[e + h]) * ln_rs[j] + bvp[e + h];
                gat[h] = (acc[2 * k2 + 1][j][e + h] - ln_mu[j] * cgp[e + h]) * ln_rs[j] + bgp[e + h];
              } else {
                val[h] = acc[2 * k2][j][e + h] * p.alpha + bvp[e + h];
                gat[h] = acc[2 * k2 + 1][j][e + h] * p.alpha + bgp[e + h];
              }
            }
            const f32x2 r = val * gelu_bf16out_f2(gat);
            o.e[e] = from_f32<T>(r.x);
            o.e[e + 1] = from_f32<T>(r.y);
          }
          o.store(reinterpret_cast<T*>(otile + (j * 16 + (lane & 15)) * C::OPITCH) + k2 * 16 + cl);
        }
      }
    } else {
      // two halves of five 16-column blocks: the bias / column-sum / residual vectors of ten blocks next to 80 accumulator
      // and 80 fragment registers do not fit 256 VGPRs (scratch traffic, and its vmcnt drains, in every epilogue)
      constexpr int NH = NIW / 2;
      static_assert(NIW % 2 == 0, "row-panel: even number of 16-column blocks");
      float ps[MJ], pq[MJ];
#pragma unroll
      for (int j = 0; j < MJ; ++j) ps[j] = pq[j] = 0.f;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float4 bvec[NH], cvec[LNMODE == 1 ? NH : 1];
        {
          const unsigned va = lds0 + (unsigned)C::LDS_BYTES + (unsigned)((ntl * C::BN + hh * NH * 16 + cl) * 4);
          pp_u32x4 tb[NH], tc[LNMODE == 1 ? NH : 1];
          pp_static_for<0, NH>([&](auto ic) {
            constexpr int ii = decltype(ic)::value;
            tb[ii] = pp_lds_read128<ii * 64>(va);
            if constexpr (LNMODE == 1) tc[ii] = pp_lds_read128<ii * 64>(va + (unsigned)nvcol * 4u);
          });
          pp_wait_lgkm0();
#pragma unroll
          for (int ii = 0; ii < NH; ++ii) {
            asm volatile("" : "+v"(tb[ii]));
            bvec[ii] = __builtin_bit_cast(float4, tb[ii]);
            if constexpr (LNMODE == 1) { asm volatile("" : "+v"(tc[ii])); cvec[ii] = __builtin_bit_cast(float4, tc[ii]); }
          }
        }
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int m = r0 + j * 16 + (lane & 15);
          const bool mok = m < p.M;
          Quad<T> rq[RES ? NH : 1];
          if constexpr (RES) {
            const T* rp = reinterpret_cast<const T*>(p.residual) + (long)(mok ? m : 0) * p.ldr + ncol + hh * NH * 16 + cl;
#pragma unroll
            for (int ii = 0; ii < NH; ++ii) rq[ii].load(rp + ii * 16);
          }
#pragma unroll
          for (int ii = 0; ii < NH; ++ii) {
            const int i = hh * NH + ii;
            const float* bp = reinterpret_cast<const float*>(&bvec[ii]);
            const float* cp = reinterpret_cast<const float*>(&cvec[LNMODE == 1 ? ii : 0]);
            Quad<T> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v;
              if constexpr (LNMODE == 1) v = (acc[i][j][e] - ln_mu[j] * cp[e]) * ln_rs[j] + bp[e];
              else v = acc[i][j][e] * p.alpha + bp[e];
              if constexpr (RES) v += to_f32<T>(rq[ii].e[e]);
              o.e[e] = from_f32<T>(v);
              if constexpr (LNMODE == 2) {
                const float vr = mok ? to_f32<T>(o.e[e]) : 0.f;     // the value the consumer will read
                ps[j] += vr;
                pq[j] += vr * vr;
              }
            }
            o.store(reinterpret_cast<T*>(otile + (j * 16 + (lane & 15)) * C::OPITCH) + i * 16 + cl);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (LNMODE == 2) {
        // per-row partial sums of this 160-column tile: statistics slab 2 nt (slab 2 nt + 1, which the tiled kernel's
        // second wave group would fill, is written as zero so that ln_finalize_kernel sums the same number of parts)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int m = r0 + j * 16 + (lane & 15);
          float s1 = ps[j], s2 = pq[j];
          s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (m < p.M && (lane >> 4) == 0) {
            *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(2 * nt) * p.M + m) * 2) = float2{s1, s2};
            *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(2 * nt + 1) * p.M + m) * 2) = float2{0.f, 0.f};
          }
        }
      }
    }
    // (wave-private tile: no barrier; the reads below are ordinary LDS loads, hipcc orders them after the writes)
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int c = lane + 64 * i;
      const int row = c / CPR, ch = c - row * CPR;
      const pp_u32x4 v = *reinterpret_cast<const pp_u32x4*>(otile + row * C::OPITCH + ch * 16);
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, orow[i], ncol * 2, 0);
    }
    zero_acc();
    after_epi = D;
  }
  pp_wait_vm<0>();
}

template <bool GEGLU, int LNMODE, bool RES, int KC, int MJ, int D>
static int launch_rowpanel_depth(const ConvGemmParams& p, hipStream_t stream) {
  using C = RowPanelCfgT<GEGLU, KC, MJ, D>;
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&rowpanel_kernel<GEGLU, LNMODE, RES, KC, MJ, D>), 160 * 1024)) return rc;
  const int panels = (p.M + C::BM - 1) / C::BM, ntn = p.N / C::BN;
  int ny = 1;                                       // column tiles of a panel over ny workgroups until ~256 exist
  while (panels * ny < 192 && ny * 2 <= ntn && ntn % (ny * 2) == 0) ny *= 2;
  // behind the ring and the transposition tiles: the bias (+ column sums) of a workgroup's columns, fp32
  const int lds_bytes = C::LDS_BYTES + (LNMODE == 1 ? 2 : 1) * ((ntn + ny - 1) / ny) * C::BN * 4;
  if (lds_bytes > 160 * 1024 || ((ntn + ny - 1) / ny) * C::BN > 6144) { af_set_error_msg("row-panel GEMM: N = %d does not leave room for its bias vectors in LDS", p.N); return -1; }
  dim3 grid(panels, ny, 1);
  hipLaunchKernelGGL((rowpanel_kernel<GEGLU, LNMODE, RES, KC, MJ, D>), grid, dim3(512), lds_bytes, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// ring depth: the deepest the form's LDS allows (see RowPanelCfgT).  (Measured against the round-2 depth of 2: nothing, 15.596 vs
// 15.631 ms per forward -- the knob that selected between them left in round 4.)
template <bool GEGLU, int LNMODE, bool RES, int KC = 5, int MJ = 2>
static int launch_rowpanel_one(const ConvGemmParams& p, hipStream_t stream) {
  constexpr int DEEP = GEGLU ? 5 : (MJ == 1 ? 4 : 2);
  return launch_rowpanel_depth<GEGLU, LNMODE, RES, KC, MJ, DEEP>(p, stream);
}
// (k640: the 32x32-level form, K = 640, 16 rows per wave)
static int launch_geglu_rowpanel(const ConvGemmParams& p, hipStream_t stream, bool k640 = false) {
  if (k640) return p.ln_stats ? launch_rowpanel_one<true, 1, false, 10, 1>(p, stream) : launch_rowpanel_one<true, 0, false, 10, 1>(p, stream);
  return p.ln_stats ? launch_rowpanel_one<true, 1, false>(p, stream) : launch_rowpanel_one<true, 0, false>(p, stream);
}
// K = 1280 (16x16 level, 4096 rows): 16 rows per wave x 1280 K = 160 fragment registers; N = 1280 only (eight column tiles
// over eight workgroups per 128-row panel -- one tile of 20 steps each; longer N gains nothing over the tiled kernel)
static int launch_plain_rowpanel_k1280(const ConvGemmParams& p, hipStream_t stream) {
  return p.residual ? launch_rowpanel_one<false, 0, true, 20, 1>(p, stream) : launch_rowpanel_one<false, 0, false, 20, 1>(p, stream);
}
static int launch_plain_rowpanel(const ConvGemmParams& p, hipStream_t stream, bool k640 = false) {
  const bool res = p.residual != nullptr;
  if (k640) {   // (at K = 640 only the LayerNorm-consumer q / k / v projection is long enough to gain -- and proj_in when it
                // applies the GroupNorm of its input itself: statistics-producer epilogue)
    if (p.ln_stats_out) return launch_rowpanel_one<false, 2, false, 10, 1>(p, stream);
    return p.ln_stats ? launch_rowpanel_one<false, 1, false, 10, 1>(p, stream) : launch_rowpanel_one<false, 0, false, 10, 1>(p, stream);
  }
  if (p.ln_stats) return res ? launch_rowpanel_one<false, 1, true>(p, stream) : launch_rowpanel_one<false, 1, false>(p, stream);
  if (p.ln_stats_out) return res ? launch_rowpanel_one<false, 2, true>(p, stream) : launch_rowpanel_one<false, 2, false>(p, stream);
  return res ? launch_rowpanel_one<false, 0, true>(p, stream) : launch_rowpanel_one<false, 0, false>(p, stream);
}

// ---------------------------------------------------------------------------
// 128 x 160 tile GEMM for FEW rows (round 3): the plain 1x1 GEMMs of the 16x16 level, [4096, 1280] -> 1280 and
// [4096, 5120] -> 1280 (36 per forward).
//
// 4096 rows are sixteen 256-row tiles: the 256 x 160 ping-pong tile leaves half of the chip idle (128 workgroups, 30 us; or two
// K slices and a reduce launch) and the K = 1280 row-panel form fills it only by having eight workgroups re-load each 128-row
// panel into registers (320 KB per workgroup in fragment-shaped 16-byte loads: its texture-address FIFO is full 13-16 % of the
// time, 28 us).  Here a workgroup owns 128 rows x 160 columns (32 x 8 = 256 workgroups, ONE K slice), BOTH operands stream
// through an LDS ring of D + 1 slots in whole 128-byte lines by LDS-DMA (16 + 20 pieces of 1 KiB per K step, 4-5 per wave),
// prefetch distance D = 2 steps (3 measured the same), one barrier per step; eight waves as 2 column halves x 4 row quarters,
// 32 rows x 80 columns each (2 x 5 MFMA 16x16x32 blocks, 20 MFMAs per step).  Swizzle, fragment addressing and the transposed
// epilogue are those of rowpanel_kernel.  21.5 us ([4096, 1280] -> 1280), 60.8 us (K = 5120).
//
// The epilogue's 16-byte stores: all chunks and offsets first, then the stores back to back (see the comment there and
// scripts/check_isa_hazards.py) -- the first version let hipcc put the next chunk's address arithmetic between them and stored
// rewritten registers.
// ---------------------------------------------------------------------------
template <int D_> struct M128CfgT {
  static constexpr int BM = 128, BN = 160, D = D_, NS = D + 1;
  static constexpr int XBYTES = BM * 128, WBYTES = BN * 128, SLOT = XBYTES + WBYTES;     // 16 + 20 KiB
  static constexpr int XP = BM / 8, WP = BN / 8;                                         // 16 + 20 pieces per step
  static constexpr int NI = 5, MJ = 2;                                                    // 16-wide blocks per wave: 80 columns, 32 rows
  static constexpr int OCOLS = 80, OPITCH = OCOLS * 2 + 16, OBYTES = 32 * OPITCH;         // per-wave output transposition tile
  static constexpr int CPR = OCOLS / 8, NST = 32 * CPR / 64;                              // 10 chunks per row segment, 5 stores per lane
  // output tiles behind the ring while both fit (D = 2: 108 + 44 KiB), else in the ring's first slots after the last step
  static constexpr bool OWN_OTILE = D >= 2 && NS * SLOT + 8 * OBYTES <= 160 * 1024;   // (D = 1: 72 KiB in all, two workgroups per CU)
  static constexpr int OTILE0 = OWN_OTILE ? NS * SLOT : 0;
  static constexpr int LDS_BYTES = OWN_OTILE ? NS * SLOT + 8 * OBYTES : NS * SLOT;
  static_assert(LDS_BYTES <= 160 * 1024 && 8 * OBYTES <= NS * SLOT && (32 * CPR) % 64 == 0, "m128 LDS / store mapping");
};

// LNMODE as in conv_gemm_pp_kernel / rowpanel_kernel: 1 = LayerNorm consumer (rows un-normalised, W * gamma; the epilogue
// applies rstd * (acc - mu * colsum) + bias; (mu, rstd) from ln_stats or, ln_parts_n > 0, summed here from the producer's
// partial sums like ln_finalize_kernel does), 2 = statistics producer (per-row (sum, sum of squares) of the bf16 values stored,
// slab tn * 2 + g of [N / 80][M][2] -- the tiled kernel's layout)
template <bool RES, int DEPTH, int LNMODE>
__global__ __launch_bounds__(512) void gemm_m128_kernel(const ConvGemmParams p) {
  using C = M128CfgT<DEPTH>;
  typedef bf16 T;
  constexpr int NI = C::NI, MJ = C::MJ, D = C::D, NS = C::NS, NST = C::NST, CPR = C::CPR;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wid >> 2, wq = wid & 3;
  const int ntm = p.M / C::BM, ntn = p.N / C::BN;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, ntm, ntn, p.group_m, tm, tn);
  const int m0 = tm * C::BM, n0 = tn * C::BN;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.src)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.W)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(p.out), 0, (int)0xFFFFFFF0u, 0x00020000);

  // ---- staging: piece = 8 rows x 128 B; lane -> row lane >> 3, LDS slot lane & 7 <- data chunk (lane & 7) ^ row (the swizzle
  // lives in the SOURCE address).  Activations: pieces wid, wid + 8 (all waves two); weights: wid, wid + 8, and wid + 16 for
  // waves 0-3 (20 pieces) ----
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  unsigned x_off[2], w_off[3];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = m0 + (wid + 8 * q) * 8 + srow;
    x_off[q] = m < p.M ? (unsigned)((long)m * p.ldc * 2) + dchunk * 16u : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int n = n0 + (wid + 8 * q) * 8 + srow;
    w_off[q] = (n < p.Wrows && wid + 8 * q < C::WP) ? (unsigned)((long)n * p.ldw * 2) + dchunk * 16u : 0xFFFFFFFFu;
  }
  const int nwq = wid < 4 ? 3 : 2;                  // weight pieces of this wave per step (vmcnt: + 2 activation pieces)
  const int KT = p.K / 64;
  // (steps past the end are not staged; the tail of the loop waits for everything outstanding instead of counting)
  auto stage = [&](int t, int slot) {               // (by value: see KWalk in conv_gemm_pp_kernel)
    if (t >= KT) return;
    const unsigned so = (unsigned)t * 128u;
    char* base = smem + slot * C::SLOT;
#pragma unroll
    for (int q = 0; q < 2; ++q) lds_dma16(rs_x, base + (wid + 8 * q) * 1024, x_off[q], so);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (q < 2 || wid < 4) lds_dma16(rs_w, base + C::XBYTES + (wid + 8 * q) * 1024, w_off[q], so);
  };

  // ---- fragment addressing (rows of 128 B; 16-byte chunk c of row r sits in slot c ^ (r & 7)) ----
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)((lane >> 4) ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)(((lane >> 4) + 4) ^ (lane & 7)) * 16u;
  const unsigned x_base = (unsigned)((wq * 32 + (lane & 15)) * 128);                   // + j * 2048
  const unsigned w_base = (unsigned)(C::XBYTES + (g * 80 + (lane & 15)) * 128);        // + i * 2048
  const int cl = 4 * (lane >> 4);

  f32x4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  pp_u32x4 xf[MJ][2], wf[NI][2];
  auto wait_block = [&](auto nc, pp_u32x4 (&blk)[2]) {
    constexpr int n = decltype(nc)::value;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(blk[0]), "+v"(blk[1]) : "n"(n) : "memory");
  };
  // one K step: the two activation blocks and the first weight block are read first; weight block i + 2 is read behind the
  // MFMAs of block i; the LDS-DMA of step t + D is issued behind the second weight block's MFMAs
  auto step = [&](int slot, int t, int stage_slot) {
    const unsigned sb = lds0 + (unsigned)(slot * C::SLOT);
    auto rd_x = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      xf[j][0] = pp_lds_read128<j * 2048>(sb + x_base + fch0);
      xf[j][1] = pp_lds_read128<j * 2048>(sb + x_base + fch1);
    };
    auto rd_w = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      wf[i][0] = pp_lds_read128<i * 2048>(sb + w_base + fch0);
      wf[i][1] = pp_lds_read128<i * 2048>(sb + w_base + fch1);
    };
    rd_x(std::integral_constant<int, 0>{});
    rd_x(std::integral_constant<int, 1>{});
    rd_w(std::integral_constant<int, 0>{});
    rd_w(std::integral_constant<int, 1>{});
    __builtin_amdgcn_sched_barrier(0);
    pp_static_for<0, NI>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      // reads outstanding behind block i's: the blocks issued after it (two ahead, until the last ones)
      constexpr int issued = (i + 2) < NI ? (i + 2) : NI;          // weight blocks issued before block i's MFMAs
      wait_block(std::integral_constant<int, 2 * (issued - i - 1)>{}, wf[i]);
      if constexpr (i == 0) { wait_block(std::integral_constant<int, 2 * (issued - 1)>{}, xf[0]); wait_block(std::integral_constant<int, 2 * (issued - 1)>{}, xf[1]); }
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][0]), __builtin_bit_cast(bf16x8, xf[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i][1]), __builtin_bit_cast(bf16x8, xf[j][1]), acc[i][j], 0, 0, 0);
        asm volatile("" : "+v"(acc[i][j]));
      }
      if constexpr (i + 2 < NI) rd_w(std::integral_constant<int, i + 2>{});
      if constexpr (i == 1) stage(t + D, stage_slot);
      __builtin_amdgcn_sched_barrier(0);
    });
    pp_wait_lgkm0();
  };

  // LayerNorm consumer: (mu, rstd) of this lane's two rows, fetched BEFORE the first LDS-DMA is issued (the loop's vmcnt
  // arithmetic counts DMA pieces only; hipcc's own waits for these loads then leave the pieces in flight)
  float ln_mu[LNMODE == 1 ? MJ : 1], ln_rs[LNMODE == 1 ? MJ : 1];
  constexpr int LNP = 16;                       // partial-sum slabs fetched at once (N = 1280: 16; more: the loop below)
  float2 ln_part[LNMODE == 1 ? MJ : 1][LNMODE == 1 ? LNP : 1];
  if constexpr (LNMODE == 1) {
    // every slab of both rows in flight at once (out-of-range offsets = zeros, no branches): summed one after the other in a
    // loop these 8-byte loads cost 16 memory latencies per launch
    const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_stats), 0, (int)0xFFFFFFF8u, 0x00020000);
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int m = m0 + wq * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int q = 0; q < LNP; ++q) {
        const bool live = m < p.M && (p.ln_parts_n > 0 ? q < p.ln_parts_n : q == 0);
        const unsigned off = live ? (unsigned)(((long)q * p.M + m) * 8) : 0xFFFFFFFFu;
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs_s, off, 0, 0);
        ln_part[j][q] = float2{__uint_as_float(v.x), __uint_as_float(v.y)};
      }
    }
  }
  auto ln_finish = [&]() {
    if constexpr (LNMODE == 1) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const int m = m0 + wq * 32 + j * 16 + (lane & 15);
        if (p.ln_parts_n > 0) {                 // same order and arithmetic as ln_finalize_kernel
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int q = 0; q < LNP; ++q)
            if (q < p.ln_parts_n) { s1 += ln_part[j][q].x; s2 += ln_part[j][q].y; }
          for (int q = LNP; q < p.ln_parts_n; ++q) {
            const float2 pq2 = m < p.M ? *reinterpret_cast<const float2*>(p.ln_stats + ((long)q * p.M + m) * 2) : float2{0.f, 0.f};
            s1 += pq2.x;
            s2 += pq2.y;
          }
          const float mu = s1 * p.ln_inv_count;
          const float var = fmaxf(s2 * p.ln_inv_count - mu * mu, 0.f);
          ln_mu[j] = mu;
          ln_rs[j] = __builtin_amdgcn_rsqf(var + p.ln_eps);
        } else {
          ln_mu[j] = ln_part[j][0].x;
          ln_rs[j] = ln_part[j][0].y;
        }
      }
      // (all of them consumed before the loop: its vmcnt arithmetic counts LDS-DMA pieces only)
      asm volatile("s_nop 0" : "+v"(ln_mu[0]), "+v"(ln_rs[0]), "+v"(ln_mu[MJ - 1]), "+v"(ln_rs[MJ - 1])::"memory");
    }
  };

  // ---- main loop: step t in slot t % NS; steps 0 .. D-1 in flight ----
#pragma unroll
  for (int i = 0; i < D; ++i) stage(i, i);
  ln_finish();                                  // (the statistics loads were issued before the pieces: older, so already waited for)
  int cur = 0;
  for (int t = 0; t < KT; ++t) {
    // own pieces of step t landed: everything but the pieces of the next D - 1 steps (2 + nwq per step); in the last D - 1
    // steps fewer are in flight: wait for all of them
    if (t + D > KT) pp_wait_vm<0>();
    else if (nwq == 3) pp_wait_vm<5 * (D - 1)>(); else pp_wait_vm<4 * (D - 1)>();
    __builtin_amdgcn_s_barrier();
    const int prev = cur == 0 ? NS - 1 : cur - 1;
    step(cur, t, prev);
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  pp_wait_vm<0>();
  __builtin_amdgcn_s_barrier();   // every wave has read its last step: the ring is free for the output tiles
  asm volatile("" ::: "memory");  // (s_barrier is no memory operation for hipcc: keeps the tile stores below it)

  // ---- epilogue: bias (+ residual) in the accumulator layout, bf16 through the wave-private tile, 16-byte row stores ----
  char* otile = smem + C::OTILE0 + wid * C::OBYTES;
  const int r0 = m0 + wq * 32, ncol = n0 + g * 80;
  float4 bvec[NI], cvec[LNMODE == 1 ? NI : 1];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    bvec[i] = p.bias ? *reinterpret_cast<const float4*>(p.bias + ncol + i * 16 + cl) : float4{0.f, 0.f, 0.f, 0.f};
    if constexpr (LNMODE == 1) cvec[i] = *reinterpret_cast<const float4*>(p.ln_colsum + ncol + i * 16 + cl);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int m = r0 + j * 16 + (lane & 15);
    const bool mok = m < p.M;
    Quad<T> rq[RES ? NI : 1];
    if constexpr (RES) {
      const T* rp = reinterpret_cast<const T*>(p.residual) + (long)(mok ? m : 0) * p.ldr + ncol + cl;
#pragma unroll
      for (int i = 0; i < NI; ++i) rq[i].load(rp + i * 16);
    }
    float ps = 0.f, pq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float* bp = reinterpret_cast<const float*>(&bvec[i]);
      const float* cp = reinterpret_cast<const float*>(&cvec[LNMODE == 1 ? i : 0]);
      Quad<T> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v;
        if constexpr (LNMODE == 1) v = (acc[i][j][e] - ln_mu[j] * cp[e]) * ln_rs[j] + bp[e];
        else v = acc[i][j][e] * p.alpha + bp[e];
        if constexpr (RES) v += to_f32<T>(rq[i].e[e]);
        o.e[e] = from_f32<T>(v);
        if constexpr (LNMODE == 2) {
          const float vr = mok ? to_f32<T>(o.e[e]) : 0.f;     // the value the consumer will read
          ps += vr;
          pq += vr * vr;
        }
      }
      o.store(reinterpret_cast<T*>(otile + (j * 16 + (lane & 15)) * C::OPITCH) + i * 16 + cl);
    }
    if constexpr (LNMODE == 2) {
      ps += __shfl_xor(ps, 16, 64); pq += __shfl_xor(pq, 16, 64);
      ps += __shfl_xor(ps, 32, 64); pq += __shfl_xor(pq, 32, 64);
      if (mok && (lane >> 4) == 0)
        *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(tn * 2 + g) * p.M + m) * 2) = float2{ps, pq};
    }
  }
  // All chunks and offsets first, then the stores back to back with nothing between them: a 16-byte buffer store whose data
  // register a VALU instruction rewrites in the next slot was seen to store the NEW value on gfx950 (hipcc pads that hazard only
  // for stores without an SGPR offset, and this one has ncol * 2 there).
  pp_u32x4 ov[NST];
  unsigned ooff[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int c = lane + 64 * i;
    const int row = c / CPR, ch = c - row * CPR;
    const int m = r0 + row;
    ov[i] = *reinterpret_cast<const pp_u32x4*>(otile + row * C::OPITCH + ch * 16);
    ooff[i] = m < p.M ? (unsigned)((long)m * p.ldo * 2) + (unsigned)ch * 16u : 0xFFFFFFFFu;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b128(ov[i], rs_o, ooff[i], ncol * 2, 0);
}

template <bool RES, int DEPTH, int LNMODE> static int launch_gemm_m128_one(const ConvGemmParams& p, hipStream_t stream) {
  using C = M128CfgT<DEPTH>;
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&gemm_m128_kernel<RES, DEPTH, LNMODE>), C::LDS_BYTES)) return rc;
  dim3 grid((p.M / C::BM) * (p.N / C::BN), 1, 1);
  hipLaunchKernelGGL((gemm_m128_kernel<RES, DEPTH, LNMODE>), grid, dim3(512), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <int DEPTH> static int launch_gemm_m128_depth(const ConvGemmParams& p, hipStream_t stream) {
  if (p.ln_stats) return p.residual ? launch_gemm_m128_one<true, DEPTH, 1>(p, stream) : launch_gemm_m128_one<false, DEPTH, 1>(p, stream);
  if (p.ln_stats_out) return p.residual ? launch_gemm_m128_one<true, DEPTH, 2>(p, stream) : launch_gemm_m128_one<false, DEPTH, 2>(p, stream);
  return p.residual ? launch_gemm_m128_one<true, DEPTH, 0>(p, stream) : launch_gemm_m128_one<false, DEPTH, 0>(p, stream);
}
static int launch_gemm_m128(const ConvGemmParams& p, hipStream_t stream) {
  // more than one round of tiles and few K steps each: two-slot ring (72 KiB), two workgroups per CU cover each other's
  // prologue and epilogue
  const long nb128 = (long)(p.M / 128) * (p.N / 160);
  if (nb128 > 256 && p.K <= 1280 && p.M > 8192) return launch_gemm_m128_depth<1>(p, stream);
  return launch_gemm_m128_depth<2>(p, stream);
}

// ---------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS-resident input halo.
//
// The implicit-GEMM kernel above re-gathers every input pixel 9 times (once per filter tap); at the 64x64 and
// 32x32 levels that gather, not the MFMA, sets the speed (L2 -> LDS traffic per FLOP).  Here one workgroup owns a
// TH x TW patch of output pixels of one image (128 pixels): for each 128-byte channel chunk the (TH+2)x(TW+2) input
// halo is staged ONCE and the nine taps read their MFMA operand from it at shifted row addresses; only the weight
// tile streams per tap.  Activation traffic drops ~7x (halo overhead 1.4-1.6x instead of 9x).
// K is walked as (channel chunk, tap) instead of (tap, channel chunk) — the same sum.
// ---------------------------------------------------------------------------
template <int TW, int BN> struct HaloCfg {
  static constexpr int TH = 128 / TW;
  static constexpr int HWD = TW + 2, HHT = TH + 2, HROWS = HHT * HWD;
  static constexpr int HALO_BYTES = HROWS * 128;
  static constexpr int W_BYTES = BN * 128;
  static constexpr int NHL = (HROWS * 8 + 255) / 256;  // halo 16-byte chunks per thread
  static constexpr int WR = BN / 32;
  static constexpr int WN = BN / 2, NI = WN / 32, MI = 2;
  static constexpr int EPI_LD = BN + 4;
  static constexpr int EPI_BYTES = 128 * EPI_LD * 4;
  static constexpr int MAIN_BYTES = HALO_BYTES + 2 * W_BYTES;
  static constexpr int LDS_BYTES = MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES;
};

template <typename T, int TW, int BN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvGemmParams p) {
  using C = HaloCfg<TW, BN>;
  constexpr int BK = 128 / sizeof(T);
  constexpr int MI = C::MI, NI = C::NI, WR = C::WR, NHL = C::NHL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* wt = smem + C::HALO_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wm = wave & 1, wn = wave >> 1;

  const int H = p.Ho, W = p.Wo;
  const int tiles_x = W / TW, tpi = tiles_x * (H / C::TH);
  const int ntn = (p.N + BN - 1) / BN;
  int tm, tn;
  tile_coords(blockIdx.x, gridDim.x, p.M / 128, ntn, p.group_m, tm, tn);
  const int b = tm / tpi, tt = tm - b * tpi;
  const int ty0 = (tt / tiles_x) * C::TH, tx0 = (tt - (tt / tiles_x) * tiles_x) * TW;
  const int n0 = tn * BN;

  const __amdgpu_buffer_rsrc_t rs_x =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, (int)0xFFFFFFF0u, 0x00020000);
  const unsigned ldcb = (unsigned)p.ldc * (unsigned)sizeof(T);
  const unsigned img_off = (unsigned)((long)b * p.src_batch_stride * (long)sizeof(T));

  // byte offsets of the halo chunks this thread stages (independent of the channel chunk)
  unsigned h_off[NHL];
#pragma unroll
  for (int j = 0; j < NHL; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx >> 3, ch = idx & 7;
    const int hy = row / C::HWD, hx = row - hy * C::HWD;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = idx < C::HROWS * 8 && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    h_off[j] = ok ? img_off + (unsigned)(y * W + x) * ldcb + (unsigned)(ch * 16) : 0xFFFFFFFFu;
  }
  const int chunk = tid & 7, r0 = tid >> 3;
  unsigned w_off[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int n = n0 + r0 + 32 * i;
    w_off[i] = n < p.Wrows ? (unsigned)(((long)n * p.ldw) * (long)sizeof(T)) + (unsigned)(chunk * 16) : 0xFFFFFFFFu;
  }
  // halo row of each output pixel this lane feeds to the MFMA (tap (0,0) position)
  int hrow0[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int ml = wm * 64 + mi * 32 + l31;
    hrow0[mi] = (ml / TW) * C::HWD + (ml % TW);
  }

  const int NC = p.Cin / BK, NS = NC * 9;
  int l_tap = 0, l_cc = 0;  // load-side position in the (cc, tap) walk
  int c_tap = 0, c_cc = 0;  // compute-side position

  auto gloadW = [&](uint4 (&wr)[WR]) {
    const unsigned k0b = (unsigned)(l_tap * p.Cin + l_cc * BK) * (unsigned)sizeof(T);
#pragma unroll
    for (int i = 0; i < WR; ++i)
      wr[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[i], k0b, 0));
    if (++l_tap == 9) { l_tap = 0; ++l_cc; }
  };
  auto lstoreW = [&](int buf, const uint4 (&wr)[WR]) {
    char* ws = wt + buf * C::W_BYTES;
#pragma unroll
    for (int i = 0; i < WR; ++i) *reinterpret_cast<uint4*>(ws + lds_off(r0 + 32 * i, chunk)) = wr[i];
  };
  uint4 hreg[NHL];
  auto hload = [&](int cc) {
    const unsigned cb = (unsigned)(cc * BK) * (unsigned)sizeof(T);
#pragma unroll
    for (int j = 0; j < NHL; ++j)
      hreg[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, h_off[j], cb, 0));
  };
  auto hstore = [&]() {
#pragma unroll
    for (int j = 0; j < NHL; ++j) {
      const int idx = tid + 256 * j;
      if (idx < C::HROWS * 8) *reinterpret_cast<uint4*>(halo + lds_off(idx >> 3, idx & 7)) = hreg[j];
    }
  };

  f32x16 acc[NI][MI];
#pragma unroll
  for (int a = 0; a < NI; ++a)
#pragma unroll
    for (int bb = 0; bb < MI; ++bb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][bb][r] = 0.f;

  auto compute = [&](int buf) {
    const char* ws = wt + buf * C::W_BYTES;
    const int ky = c_tap / 3, kx = c_tap - ky * 3;
    const int tapoff = ky * C::HWD + kx;
    const char* xrow[MI];
    int xsw[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int hr = hrow0[mi] + tapoff;
      xrow[mi] = halo + hr * 128;
      xsw[mi] = (hr >> 1) & 7;
    }
    uint4 xf[2][MI], wf[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[0][mi] = *reinterpret_cast<const uint4*>(xrow[mi] + ((h ^ xsw[mi]) << 4));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      wf[0][ni] = *reinterpret_cast<const uint4*>(ws + lds_off(wn * C::WN + ni * 32 + l31, h));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          xf[(s + 1) & 1][mi] = *reinterpret_cast<const uint4*>(xrow[mi] + (((2 * (s + 1) + h) ^ xsw[mi]) << 4));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          wf[(s + 1) & 1][ni] =
              *reinterpret_cast<const uint4*>(ws + lds_off(wn * C::WN + ni * 32 + l31, 2 * (s + 1) + h));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) Mma<T>::step(wf[s & 1][ni], xf[s & 1][mi], acc[ni][mi]);
    }
  };
  // end of a step: publish the next weight tile; at a channel-chunk boundary swap in the prefetched halo
  auto step_end = [&]() {
    const bool boundary = (c_tap == 8) && (c_cc + 1 < NC);
    __syncthreads();
    if (boundary) {
      hstore();
      __syncthreads();
    }
    if (++c_tap == 9) { c_tap = 0; ++c_cc; }
  };

  uint4 wa[WR], wb[WR];
  hload(0);
  gloadW(wa);
  if (NS > 1) gloadW(wb);
  hstore();
  lstoreW(0, wa);
  __syncthreads();
  for (int st = 0; st < NS; st += 2) {
    if (c_tap == 0 && c_cc + 1 < NC) hload(c_cc + 1);
    if (st + 2 < NS) gloadW(wa);
    compute(0);
    if (st + 1 < NS) lstoreW(1, wb);
    step_end();
    if (st + 1 >= NS) break;
    if (c_tap == 0 && c_cc + 1 < NC) hload(c_cc + 1);
    if (st + 3 < NS) gloadW(wb);
    compute(1);
    if (st + 2 < NS) lstoreW(0, wa);
    step_end();
  }

  // ---- epilogue (same two-phase scheme as conv_gemm_kernel; no GEGLU / split-K here) ----
  float* et = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * 64 + mi * 32 + l31;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 o;
        o.x = acc[ni][mi][4 * q + 0] * p.alpha; o.y = acc[ni][mi][4 * q + 1] * p.alpha;
        o.z = acc[ni][mi][4 * q + 2] * p.alpha; o.w = acc[ni][mi][4 * q + 3] * p.alpha;
        *reinterpret_cast<float4*>(et + row * C::EPI_LD + wn * C::WN + ni * 32 + 8 * q + 4 * h) = o;
      }
  }
  __syncthreads();
  constexpr int tpr = BN >> 3, rpp = 256 / tpr;
  const int trow = tid / tpr, c8 = (tid - trow * tpr) * 8;
  const int n = n0 + c8;
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const T* __restrict__ res = reinterpret_cast<const T*>(p.residual);
  const T* __restrict__ rowb = reinterpret_cast<const T*>(p.rowbias);
  if (n < p.N) {
    const int nvalid = min(8, p.N - n);
    for (int row = trow; row < 128; row += rpp) {
      const int py = row / TW, px = row - py * TW;
      const int m = (b * H + ty0 + py) * W + tx0 + px;
      float v[8];
      const float4 a = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8);
      const float4 b4 = *reinterpret_cast<const float4*>(et + row * C::EPI_LD + c8 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
      epi_store8<T>(p, v, m, b, n, nvalid, out, res, rowb, p.bias);
    }
  }
}

// split-K reduce + epilogue: out[m][n] = T(sum_z slab[z][m][n] + bias + rowbias + residual)
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvGemmParams p) {
  const long nq = (long)p.M * (p.N >> 2);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
    const int m = (int)(i / (p.N >> 2));
    const int n = (int)(i - (long)m * (p.N >> 2)) * 4;
    splitk_reduce_quad<T>(p, m, n);
  }
}

// LayerNorm row statistics: partial sums [parts][M][2] (sum, sum of squares; written by the LNMODE 2 epilogue) ->
// [M][2] = (mu, rstd), summed in slab order (deterministic).  One thread per row.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float* __restrict__ part, int parts, int M, float inv_count,
                                                          float eps, float* __restrict__ out) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float s1 = 0.f, s2 = 0.f;
  for (int q = 0; q < parts; ++q) {
    const float2 st = *reinterpret_cast<const float2*>(part + ((long)q * M + m) * 2);
    s1 += st.x;
    s2 += st.y;
  }
  const float mu = s1 * inv_count;
  const float var = fmaxf(s2 * inv_count - mu * mu, 0.f);
  *reinterpret_cast<float2*>(out + (long)m * 2) = float2{mu, __builtin_amdgcn_rsqf(var + eps)};
}
int af_launch_ln_finalize(const float* part, int parts, int M, int count, float eps, float* out, hipStream_t s) {
  if (M <= 0) return 0;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3((M + 255) / 256), dim3(256), 0, s, part, parts, M, 1.0f / (float)count, eps, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// planning (tile shape + split-K) and launch
// ---------------------------------------------------------------------------
// fp8 (OCP e4m3) twin of a repacked bf16 weight [rows][taps * cin_pad] (K order tap, channel): K re-ordered into
// 64-channel units (unit u = channel chunk u / taps, tap u % taps; zero units up to k8, a multiple of 128) and every row
// scaled by the power of two that brings its largest magnitude into (224, 448]; sc[row] = E8M0 byte of the inverse.
__device__ __forceinline__ unsigned pack4_e4m3(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
  b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
  d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}
__global__ __launch_bounds__(256) void quant_weight_fp8_kernel(const bf16* __restrict__ w, int ldw, int cin_pad, int taps,
                                                                unsigned char* __restrict__ w8, int k8,
                                                                unsigned char* __restrict__ sc) {
  __shared__ float s_max[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const bf16* wr = w + (long)row * ldw;
  float mx = 0.f;
  for (int k = tid; k < ldw; k += 256) mx = fmaxf(mx, fabsf(to_f32<bf16>(wr[k])));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((tid & 63) == 0) s_max[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  int e = 0;
  if (mx > 0.f) {
    int k;
    const float m = frexpf(mx, &k);          // mx = m * 2^k, m in [0.5, 1)
    e = (m <= 0.875f ? 9 : 8) - k;           // largest e with mx * 2^e <= 448 = 0.875 * 2^9
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
  }
  const float mul = ldexpf(1.f, e);
  if (tid == 0) sc[row] = (unsigned char)(127 - e);
  unsigned* dst = reinterpret_cast<unsigned*>(w8 + (long)row * k8);
  for (int q = tid; q < k8 / 4; q += 256) {
    const int idx = q * 4, u = idx >> 6, i = idx & 63;
    const int cc = u / taps, tap = u - cc * taps;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (cc * 64 < cin_pad) {
      const bf16* sp = wr + (long)tap * cin_pad + cc * 64 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = to_f32<bf16>(sp[j]) * mul;
    }
    dst[q] = pack4_e4m3(v[0], v[1], v[2], v[3]);
  }
}
int af_launch_quant_weight_fp8(const void* w, int rows, int ldw, int cin_pad, int ks, void* w8, int k8, unsigned char* sc,
                               hipStream_t stream) {
  if (cin_pad % 64 != 0 || k8 % 128 != 0 || k8 < ks * ks * cin_pad || ldw != ks * ks * cin_pad) {
    af_set_error_msg("quant_weight_fp8: cin_pad=%d k8=%d ldw=%d ks=%d", cin_pad, k8, ldw, ks);
    return -1;
  }
  hipLaunchKernelGGL(quant_weight_fp8_kernel, dim3(rows), dim3(256), 0, stream, reinterpret_cast<const bf16*>(w), ldw,
                     cin_pad, ks * ks, reinterpret_cast<unsigned char*>(w8), k8, sc);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// bf16 -> e4m3 of x * mul, saturating (operator-level tests; the model's fp8 activations come from the GroupNorm kernels)
__global__ __launch_bounds__(256) void cast_fp8_kernel(const bf16* __restrict__ x, unsigned* __restrict__ y, long n4, float mul) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    Quad<bf16> q;
    q.load(x + i * 4);
    y[i] = pack4_e4m3(to_f32<bf16>(q.e[0]) * mul, to_f32<bf16>(q.e[1]) * mul, to_f32<bf16>(q.e[2]) * mul, to_f32<bf16>(q.e[3]) * mul);
  }
}
int af_launch_cast_fp8(const void* x, void* y, long n, float mul, hipStream_t stream) {
  if (n % 4 != 0) { af_set_error_msg("cast_fp8: n must be a multiple of 4"); return -1; }
  unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(cast_fp8_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const bf16*>(x),
                     reinterpret_cast<unsigned*>(y), n / 4, mul);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static std::mutex g_last_plan_mu;
static AfGemmPlan g_af_last_plan = {0, 1, 0, 0, 1};
void af_set_last_plan(const AfGemmPlan& pl) { std::lock_guard<std::mutex> lk(g_last_plan_mu); g_af_last_plan = pl; }
AfGemmPlan af_get_last_plan() { std::lock_guard<std::mutex> lk(g_last_plan_mu); return g_af_last_plan; }
std::atomic<long> g_af_gn_consumer_launches{0};
// launches since af_gemm_plan_counts_reset: [0..5] by tile (implicit-GEMM / ping-pong kernels), [6] LDS-halo kernel,
// [7] launches that sliced K (counted in their tile's slot as well), [8] / [9] ping-pong launches with the LayerNorm
// consumer / statistics-producer epilogue, [10] ping-pong launches with fp8 operands, [11] eight-wave halo launches
// (counted under tile 5 as well), [12] row-panel GEGLU launches (counted under their planned tile as well)
std::atomic<long> g_af_plan_counts[15] = {};   // [13]: phase-decomposed upsampled convolutions, [14]: GroupNorm-statistics producers


// tile: 0 = 128x128, 1 = 64x128, 2 = 128x64, 3 = 64x64
static void plan_group_m(AfGemmPlan& pl, const ConvGemmParams& p);

// fp8 operands: the ping-pong kernel or nothing (tile -1: the caller keeps the layer on the bf16 path)
static AfGemmPlan plan_fp8(const ConvGemmParams& p, int batch) {
  AfGemmPlan pl;
  pl.tile = -1;
  pl.splitk = 1;
  pl.ws_bytes = 0;
  pl.halo_tw = 0;
  pl.group_m = 1;
  const int cand = p.N % 160 == 0 ? 5 : (p.N % 128 == 0 ? 4 : -1);
  if (batch != 1 || p.epilogue == AF_EPI_GEGLU || cand < 0 || p.K % 128 != 0 || p.Cin % 64 != 0 || (p.ks != 1 && p.ks != 3) ||
      p.up != 0 || p.M < 512 || !g_af_knobs.gemm_pp)
    return pl;
  const int KT = p.K / 128;
  const long nb = (long)((p.M + 255) / 256) * (p.N / (cand == 5 ? 160 : 128));
  int s = 1;
  if (nb < 208) {   // slices of >= 8 tiles (a tile is 128 K values here)
    s = (int)((256 + nb / 2) / nb);
    if (s > KT / 8) s = KT / 8;
    if (s > 16) s = 16;
    if (s < 1) s = 1;
  }
  const long nbs = nb * s;
  const double fill = (double)nbs / (double)(((nbs + 255) / 256) * 256);
  if (fill < g_af_knobs.gemm_pp_minfill * 0.01) return pl;
  pl.tile = cand;
  pl.splitk = s;
  if (s > 1) pl.ws_bytes = (size_t)s * p.M * p.N * sizeof(float);
  plan_group_m(pl, p);
  return pl;
}

AfGemmPlan af_plan_conv_gemm(const ConvGemmParams& p, int batch, int elem_size) {
  if (p.fp8) return plan_fp8(p, batch);
  AfGemmPlan pl;
  pl.tile = 0;
  pl.splitk = 1;
  pl.ws_bytes = 0;
  const bool geglu = p.epilogue == AF_EPI_GEGLU;
  const bool n128 = geglu || (p.N % 128) == 0 || p.N > 640;
  static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 128, 64, 64};
  static const double eff[4] = {1.0, 0.88, 0.88, 0.72};
  // cost model: rounds over the 256 CUs x tile area / tile efficiency (per unit of K)
  double best = 1e30;
  for (int t = 0; t < 4; ++t) {
    if (n128 && bn[t] != 128) continue;
    if (!n128 && bn[t] != 64) continue;
    const long nb = (long)((p.M + bm[t] - 1) / bm[t]) * ((p.N + bn[t] - 1) / bn[t]) * batch;
    const double rounds = (double)((nb + 255) / 256);
    const double cost = rounds * bm[t] * bn[t] / eff[t];
    if (cost < best) { best = cost; pl.tile = t; }
  }
  const int BK = 128 / elem_size;
  const int KT = p.K / BK;
  const long nb = (long)((p.M + bm[pl.tile] - 1) / bm[pl.tile]) * ((p.N + bn[pl.tile] - 1) / bn[pl.tile]) * batch;
  if (batch == 1 && !geglu && nb < 256 && KT >= 16) {
    // deep-K, few tiles: slice K so that ~2 blocks per CU exist, each slice >= 8 K tiles
    const int target = g_af_knobs.splitk_target;
    int s = (int)((target + nb - 1) / nb);
    if (s > KT / 8) s = KT / 8;
    if (s > 16) s = 16;
    if (s >= 2) {
      // prefer the big tile when slicing K
      pl.tile = n128 ? 0 : 2;
      const long nb2 = (long)((p.M + bm[pl.tile] - 1) / bm[pl.tile]) * ((p.N + bn[pl.tile] - 1) / bn[pl.tile]);
      s = (int)((target + nb2 - 1) / nb2);
      if (s > KT / 8) s = KT / 8;
      if (s > 16) s = 16;
      if (s >= 2) pl.splitk = s;
    }
  }
  // few rows, short K (the 8x8-level transformer GEMMs [1024, 1280] -> 1280 / 3840, the time-embedding GEMMs with M = Bf): these
  // launches are latency-bound, and 64 x 64 tiles in ONE K slice (a few hundred small workgroups, no slabs, no reduce launch)
  // measure ahead of the 128 x 128 tile over two slices the cost model picks: 23.2 -> 18.9 us and 22.7 -> 18.9 us
  if (g_af_knobs.small_m_tile64 && batch == 1 && !geglu && p.M <= 1024 && p.N % 64 == 0 && KT <= 32) {
    const long nb64 = (long)((p.M + 63) / 64) * (p.N / 64);
    if (nb64 >= 64 && nb64 <= 1024) { pl.tile = 3; pl.splitk = 1; }
  }
  // 3x3 / stride 1 / no upsample on maps that tile by 4x32 or 8x16 pixels: LDS-halo kernel (tile 2 or 0 = BN 64 / 128)
  pl.halo_tw = 0;
  if (p.ks == 3 && p.stride == 1 && p.up == 0 && p.pad == 1 && batch == 1 && !geglu && pl.splitk == 1 &&
      p.Ho == p.Hi && p.Wo == p.Wi && p.ldc >= p.Cin && g_af_knobs.conv_halo) {
    if (p.Wo % 32 == 0 && p.Ho % 4 == 0) pl.halo_tw = 32;
    else if (p.Wo % 16 == 0 && p.Ho % 8 == 0) pl.halo_tw = 16;
    if (pl.halo_tw) {
      const long nb128 = (long)(p.M / 128) * ((p.N + 127) / 128), nb64 = (long)(p.M / 128) * ((p.N + 63) / 64);
      const double c128 = (double)((nb128 + 255) / 256) * 128, c64 = (double)((nb64 + 255) / 256) * 64 / 0.9;
      pl.tile = (n128 && c128 <= c64) ? 0 : 2;
      if (!n128 && (p.N % 64) != 0) pl.tile = 2;
    }
  }
  // ping-pong 256 x {160,128} tiles (bf16): preferred wherever the grid fills the chip.  tile 5 = BN 160 (divides
  // every SD-1.5 channel count), tile 4 = BN 128 (GEGLU needs an even number of 16-column blocks per wave; VAE widths)
  if (elem_size == 2 && batch == 1 && p.K % 64 == 0 && p.Cin % 64 == 0 && g_af_knobs.gemm_pp) {
    int cand = -1;
    if (!geglu && p.N % 160 == 0) cand = 5;
    else if (p.N % 128 == 0) cand = 4;
    // (GEGLU with few K tiles is bound by its epilogue -- ~6k VALU cycles of erf per SIMD against 7.5k cycles of main
    // loop at K = 320 -- which one workgroup per CU cannot overlap with another tile's MFMAs; with the register-phase
    // epilogue it still measures 10-18 % ahead of the four-wave kernel, so the threshold defaults to 0)
    if (cand >= 0) {
      const int tbn = cand == 5 ? 160 : 128;
      const long nb = (long)((p.M + 255) / 256) * (p.N / tbn);
      int s = 1;
      if (!geglu && nb < 208) {
        // slice K until the 256 CUs are covered; every slice keeps >= 16 K tiles (shorter slices lose more in the
        // pipeline prologue and the reduce pass than the extra workgroups gain)
        s = (int)((256 + nb / 2) / nb);
        if (s > KT / 16) s = KT / 16;
        if (s > 16) s = 16;
        if (s < 1) s = 1;
      }
      const long nbs = nb * s;
      const double fill = (double)nbs / (double)(((nbs + 255) / 256) * 256);
      if (fill >= g_af_knobs.gemm_pp_minfill * 0.01 && p.M >= 512) {
        pl.tile = cand;
        pl.splitk = s;
        pl.halo_tw = 0;
      }
    }
  }
  // eight-wave LDS-halo kernel for the 3x3 / stride-1 convolutions the 256 x 160 ping-pong tile was chosen for: whole
  // image rows per tile (Wo 16 / 32 / 64), K slices of whole channel chunks
  // (the kernel folds nearest-2x upsampling into its halo fetch as well, but on the three upsampled convolutions of a forward
  // the phased gathering kernel measured 3 % ahead: 388 vs 401 us)
  if (pl.tile == 5 && pl.halo_tw == 0 && p.ks == 3 && p.stride == 1 && p.pad == 1 && p.up == 0 && p.Ho == p.Hi && p.Wo == p.Wi &&
      (p.Wo == 16 || p.Wo == 32 || p.Wo == 64) && (p.Ho & (p.Ho - 1)) == 0 && p.Ho >= 256 / p.Wo && p.M % 256 == 0 &&
      p.K == 9 * p.Cin && p.ldc >= p.Cin && (g_af_knobs.conv_halo8 & 1) && !p.ln_stats && !p.ln_stats_out) {
    pl.halo_tw = 256;
    while (pl.splitk > 1 && (p.Cin / 64) % pl.splitk != 0) --pl.splitk;   // a K slice = whole channel chunks
  }
  // 8 x 8 maps (round 4): tiles of four whole images x 80 columns over four K slices, the images' halos resident in LDS
  // (conv3x3_s8_kernel, af_conv_s8.hip); halo_tw = 8 names it
  // ... and the 16 x 16 maps: one whole image x 80 columns per tile, ONE K slice (256 tiles at Bf = 16)
  if (elem_size == 2 && (g_af_knobs.conv_halo8 & 2) && g_af_knobs.gemm_pp && af_conv_s8_ok(p, batch) && !p.gn_stats_out &&
      (p.Wo <= 16 || (g_af_knobs.conv_halo8 & 4))) {
    pl.tile = 5;
    pl.halo_tw = 8;
    pl.splitk = af_conv_s8_slices(p, batch);
  }
  const int ft = g_af_knobs.gemm_tile;
  if (ft >= 0 && ft < 4 && !(geglu && bn[ft] != 128)) pl.tile = ft;
  const int fs = g_af_knobs.gemm_splitk;
  if (fs >= 1 && batch == 1 && !geglu) pl.splitk = fs > KT ? KT : fs;
  if (pl.halo_tw == 8 && (pl.tile != 5 || pl.splitk != af_conv_s8_slices(p, batch))) pl.halo_tw = 0;   // (a forced tile / slice count: the generic kernels)
  if (pl.splitk > 1 && pl.halo_tw != 256 && pl.halo_tw != 8) pl.halo_tw = 0;
  if (pl.halo_tw == 256 && (pl.tile != 5 || (p.Cin / 64) % pl.splitk != 0)) pl.halo_tw = 0;
  if (pl.splitk > 1) pl.ws_bytes = (size_t)pl.splitk * p.M * p.N * sizeof(float);
  plan_group_m(pl, p);
  return pl;
}

static void plan_group_m(AfGemmPlan& pl, const ConvGemmParams& p) {
  static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 128, 64, 64};
  {
    // grouped tile order: minimise  X_bytes * (NT / gn) + W_bytes * (MT / gm)  with gm * gn = workgroups resident
    // per XCD (32 CUs x blocks per CU)
    const bool h4 = pl.halo_tw != 0 && pl.halo_tw != 256 && pl.halo_tw != 8;   // the four-wave halo kernel (128-pixel patches)
    const int tbm = h4 ? 128 : (pl.tile >= 4 ? 256 : bm[pl.tile]);
    const int tbn = h4 ? ((pl.tile == 0 || pl.tile == 1) ? 128 : 64) : (pl.tile == 5 ? 160 : pl.tile == 4 ? 128 : bn[pl.tile]);
    const int MT = (p.M + tbm - 1) / tbm, NT = (p.N + tbn - 1) / tbn;
    const int resident = pl.tile >= 4 ? 32 : 32 * ((tbm * tbn >= 128 * 128) ? 2 : 3);
    // activation bytes a column of tiles streams per unit of M: every tap re-reads the input unless the taps of a channel
    // chunk follow each other (LDS halo kernel; ping-pong kernel with the tap innermost: ~1.5x halo rows at stride 1)
    const bool taps_reuse = pl.halo_tw || (pl.tile >= 4 && p.ks > 1 && (g_af_knobs.conv_tap_inner || p.fp8));
    const double xb = (double)p.M * (p.K / (p.ks * p.ks)) * (taps_reuse ? (p.stride == 1 ? 1.5 : 1.0) : (double)(p.ks * p.ks) / (p.stride * p.stride));
    const double wb = (double)p.N * p.K;
    double bestc = 1e300;
    int bestg = 1;
    for (int gmm = 1; gmm <= 64; gmm *= 2) {
      const int gm_eff = gmm < MT ? gmm : MT;
      int gn_eff = resident / gm_eff;
      if (gn_eff < 1) gn_eff = 1;
      if (gn_eff > NT) gn_eff = NT;
      const double c = xb * ((double)NT / gn_eff) + wb * ((double)MT / gm_eff);
      if (c < bestc * 0.999) { bestc = c; bestg = gmm; }
    }
    pl.group_m = g_af_knobs.gemm_groupm >= 1 ? g_af_knobs.gemm_groupm : bestg;
  }
}

template <typename T, int BM, int BN, bool DMA>
static int launch_cfg2(const ConvGemmParams& p, int batch, hipStream_t stream) {
  using C = TileCfg<BM, BN>;
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&conv_gemm_kernel<T, BM, BN, DMA>), C::LDS_BYTES)) return rc;
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  dim3 grid(ntm * ntn, 1, p.splitk > 1 ? p.splitk : batch);
  hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, DMA>), grid, dim3(256), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <typename T, int BM, int BN>
static int launch_cfg(const ConvGemmParams& p, int batch, hipStream_t stream) {
  // LDS-DMA staging wins on deep-K problems (fewer VGPRs, no ds_write); with few K tiles the 2-deep register
  // prefetch hides the first loads better.  AF_GEMM_DMA = 0 / 1 forces one variant, default: by K depth.
  const int force = g_af_knobs.gemm_dma;
  const int kt_per_slice = p.K / (128 / (int)sizeof(T)) / (p.splitk > 1 ? p.splitk : 1);
  const bool use_dma = force >= 0 ? force != 0 : kt_per_slice >= 32;
  return use_dma ? launch_cfg2<T, BM, BN, true>(p, batch, stream) : launch_cfg2<T, BM, BN, false>(p, batch, stream);
}

template <int BN, int LNMODE, bool GATHER, bool FP8, int SCHED>
static int launch_pp_one(const ConvGemmParams& p, dim3 grid, hipStream_t stream) {
  using C = PpCfg<BN>;
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&conv_gemm_pp_kernel<BN, GATHER, LNMODE, FP8, SCHED>), C::LDS_BYTES)) return rc;
  hipLaunchKernelGGL((conv_gemm_pp_kernel<BN, GATHER, LNMODE, FP8, SCHED>), grid, dim3(512), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <int BN, int LNMODE, bool GATHER> static int launch_pp_sched(const ConvGemmParams& p, dim3 grid, hipStream_t stream) {
  int sched = g_af_knobs.pp_sched;
  if (sched >= 2 && GATHER && !(p.up == 0 && p.ks * p.ks <= 31)) sched = 1;   // merged gathers exist with tap masks only
  // GEGLU launches (K = 320 .. 1280, an epilogue of ~1/3 of the workgroup's time that wants its bias and LayerNorm operands
  // fetched before the loop): the round-1 schedule measures 3-5 % ahead of both newer ones there
  if (p.epilogue == AF_EPI_GEGLU && g_af_knobs.pp_sched == 2) sched = 0;
  switch (sched) {
    case 0: return launch_pp_one<BN, LNMODE, GATHER, false, 0>(p, grid, stream);
    case 1: return launch_pp_one<BN, LNMODE, GATHER, false, 1>(p, grid, stream);
    default: return launch_pp_one<BN, LNMODE, GATHER, false, 2>(p, grid, stream);
  }
}
template <int BN> static int launch_pp(const ConvGemmParams& p, hipStream_t stream) {
  const bool gather = !(p.ks == 1 && p.pad == 0);
  dim3 grid(((p.M + 255) / 256) * (p.N / BN), 1, p.splitk > 1 ? p.splitk : 1);
  if (p.ln_stats || p.ln_stats_out) {
    // LayerNorm-fused variants: plain (1x1) GEMMs on one K slice only; the caller (af_model.hip) asks the planner first
    if (gather || p.splitk > 1 || (p.ln_stats && p.ln_stats_out) || (p.ln_stats_out && p.epilogue == AF_EPI_GEGLU)) {
      af_set_error_msg("conv_gemm: LayerNorm-fused launch needs a 1x1 GEMM without split-K");
      return -1;
    }
    return p.ln_stats ? launch_pp_sched<BN, 1, false>(p, grid, stream) : launch_pp_sched<BN, 2, false>(p, grid, stream);
  }
  return gather ? launch_pp_sched<BN, 0, true>(p, grid, stream) : launch_pp_sched<BN, 0, false>(p, grid, stream);
}

template <int BN> static int launch_pp8(const ConvGemmParams& p, hipStream_t stream) {
  const bool gather = !(p.ks == 1 && p.pad == 0);
  dim3 grid(((p.M + 255) / 256) * (p.N / BN), 1, p.splitk > 1 ? p.splitk : 1);
  // (fp8 operands exist on the merged schedule only: its gathers need the tap masks, i.e. no upsampling -- plan_fp8 refuses)
  return gather ? launch_pp_one<BN, 0, true, true, 2>(p, grid, stream) : launch_pp_one<BN, 0, false, true, 2>(p, grid, stream);
}

// fp8-operand launch (bf16 everywhere else): validated apart from the bf16 / f32 path, ping-pong kernel only
static int launch_conv_gemm_fp8(ConvGemmParams p, hipStream_t stream, const AfGemmPlan* plan, void* ws) {
  if (p.K % 128 != 0 || p.Cin % 64 != 0 || (p.ks != 1 && p.ks != 3) || p.K < p.ks * p.ks * p.Cin || p.ldw < p.K ||
      p.ldw % 16 != 0 || p.ldc % 16 != 0 || p.ldc < p.Cin || !p.w_scale || p.epilogue == AF_EPI_GEGLU || p.ln_stats || p.ln_stats_out) {
    af_set_error_msg("conv_gemm fp8: K=%d Cin=%d ks=%d ldw=%d ldc=%d (need K%%128==0, Cin%%64==0, ks 1|3, 16-byte pitches, row scales)",
                     p.K, p.Cin, p.ks, p.ldw, p.ldc);
    return -1;
  }
  if (p.N % 4 != 0 || p.ldo % 4 != 0 || (p.residual && p.ldr % 4 != 0) || (p.rowbias && p.ldrb % 4 != 0)) {
    af_set_error_msg("conv_gemm fp8: N/ldo/ldr/ldrb must be multiples of 4 (N=%d ldo=%d)", p.N, p.ldo);
    return -1;
  }
  if (p.M <= 0 || p.N <= 0) return 0;
  {
    const int HoWo = p.Ho * p.Wo > 0 ? p.Ho * p.Wo : 1;
    const double nb = (double)((p.M + HoWo - 1) / HoWo);
    if (nb * (double)p.src_batch_stride >= 4294967280.0 || (double)p.Wrows * p.ldw >= 4294967280.0) {
      af_set_error_msg("conv_gemm fp8: operand exceeds the 4 GB range of the 32-bit gather offsets (split the batch)");
      return -1;
    }
  }
  AfGemmPlan pl = plan ? *plan : af_plan_conv_gemm(p, 1, 2);
  if (pl.tile != 4 && pl.tile != 5) {
    af_set_error_msg("conv_gemm fp8: shape M=%d N=%d K=%d has no fp8 plan (ask af_plan_conv_gemm first)", p.M, p.N, p.K);
    return -1;
  }
  if (pl.splitk > 1 && !ws) pl.splitk = 1;
  p.splitk = pl.splitk;
  p.ws = ws;
  af_set_last_plan(pl);
  g_af_plan_counts[10] += 1;
  if (pl.splitk > 1) g_af_plan_counts[7] += 1;
  {
    auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (v > 0 && (1 << s) == v) ? s : -1; };
    p.howo_shift = lg2(p.Ho * p.Wo);
    p.wo_shift = lg2(p.Wo);
  }
  p.group_m = pl.group_m > 0 ? pl.group_m : 1;
  p.pp_epilogue = g_af_knobs.pp_direct < 0 ? 0 : (g_af_knobs.pp_direct ? 2 : 1);
  if (p.gn_stats_out) {
    if (!af_conv_gn_stats_ok(p, pl, p.gn_cpg)) {
      af_set_error_msg("conv_gemm fp8: GroupNorm partial sums asked of a launch that cannot write them");
      return -1;
    }
    p.pp_epilogue = 2;
    g_af_plan_counts[14] += 1;
  }
  p.k_tap_inner = 1;
  p.fast_taps = g_af_knobs.conv_fast_taps;
  p.pp_stagger = g_af_knobs.pp_stagger;
  AfProfScope prof(AF_K_PP_FP8, stream, 2.0 * p.M * (double)p.N * (p.k_logical ? p.k_logical : p.K),
                   (double)p.M * p.Cin + (double)p.N * p.K + (double)p.M * p.N * 2.0);
  const int rc = pl.tile == 4 ? launch_pp8<128>(p, stream) : launch_pp8<160>(p, stream);
  if (rc) return rc;
  if (p.splitk > 1) {
    const long nq = (long)p.M * (p.N >> 2);
    unsigned blocks = (unsigned)((nq + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((splitk_reduce_kernel<bf16>), dim3(blocks), dim3(256), 0, stream, p);
    HIP_CHECK_RET(hipGetLastError());
  }
  return 0;
}

static int launch_halo8(const ConvGemmParams& p, hipStream_t stream) {
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&conv3x3_halo8_kernel), Halo8Cfg::LDS_BYTES)) return rc;
  dim3 grid((p.M / 256) * (p.N / 160), 1, p.splitk > 1 ? p.splitk : 1);
  hipLaunchKernelGGL(conv3x3_halo8_kernel, grid, dim3(512), Halo8Cfg::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <typename T, int TW, int BN> static int launch_halo(const ConvGemmParams& p, hipStream_t stream) {
  using C = HaloCfg<TW, BN>;
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&conv3x3_halo_kernel<T, TW, BN>), C::LDS_BYTES)) return rc;
  dim3 grid((p.M / 128) * ((p.N + BN - 1) / BN), 1, 1);
  hipLaunchKernelGGL((conv3x3_halo_kernel<T, TW, BN>), grid, dim3(256), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// Nearest-2x upsample + 3x3 convolution in four phases (ConvGemmParams::W_up4; openaimodel.py:107-113 Upsample.forward,
// model.py:43-48 for the VAE): F.interpolate(nearest, 2x) followed by a 3x3 / pad-1 convolution reads, for output pixel
// (2 y + dy, 2 x + dx), the stored rows {y - 1, y} (dy = 0) or {y, y + 1} (dy = 1) -- two of the three window rows are the
// same stored row -- and likewise for columns.  Summing the weights of the taps that coincide gives an exact (up to the one
// bf16 rounding of the summed weight) 2x2 convolution per phase on the STORED map: 4 taps instead of 9, and a gather
// without the >> 1 address arithmetic, so the merged schedule with tap masks applies.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void up_phase4_weights_kernel(const bf16* __restrict__ w3, bf16* __restrict__ w4, long total,
                                                                int rows, int cin, int ldw3) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx / cin), c = (int)(idx - (long)n * cin);
  float w[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) w[ky][kx] = to_f32<bf16>(w3[(long)n * ldw3 + (ky * 3 + kx) * cin + c]);
  const long ld4 = 4L * cin;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    const int dy = ph >> 1, dx = ph & 1;
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
      for (int tx = 0; tx < 2; ++tx) {
        // window rows of tap ty: dy = 0: {0} | {1, 2};  dy = 1: {0, 1} | {2}
        const int ylo = ty == 0 ? 0 : (dy == 0 ? 1 : 2), yhi = ty == 0 ? (dy == 0 ? 0 : 1) : 2;
        const int xlo = tx == 0 ? 0 : (dx == 0 ? 1 : 2), xhi = tx == 0 ? (dx == 0 ? 0 : 1) : 2;
        float sum = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            if (ky >= ylo && ky <= yhi && kx >= xlo && kx <= xhi) sum += w[ky][kx];
        w4[((long)ph * rows + n) * ld4 + (ty * 2 + tx) * cin + c] = from_f32<bf16>(sum);
      }
  }
}
// w3: bf16 [rows][ldw3 >= 9 * cin], K = (ky, kx, c);  w4: bf16 [4][rows][4 * cin], K = (ty, tx, c)
int af_launch_up_phase4_weights(const void* w3, int rows, int cin, int ldw3, void* w4, hipStream_t stream) {
  const long total = (long)rows * cin;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(up_phase4_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const bf16*>(w3), reinterpret_cast<bf16*>(w4), total, rows, cin, ldw3);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// does the launcher take the four-phase form for this (validated, bf16) upsampled convolution?
static bool up_phase4_ok(const ConvGemmParams& p, int batch) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  return p.W_up4 && g_af_knobs.conv_up_phase4 && g_af_knobs.gemm_pp && batch == 1 && p.up == 1 && p.ks == 3 && p.stride == 1 &&
         p.pad == 1 && p.Cin % 64 == 0 && p.ldc >= p.Cin && p.K == 9 * p.Cin && (p.N % 160 == 0 || p.N % 128 == 0) &&
         pow2(p.Hs) && pow2(p.Ws) && p.Ho == 2 * p.Hs && p.Wo == 2 * p.Ws && p.M == (p.M / (p.Ho * p.Wo)) * p.Ho * p.Wo &&
         p.M / 4 >= 1024 && !p.residual && !p.rowbias && !p.ln_stats && !p.ln_stats_out && p.epilogue != AF_EPI_GEGLU &&
         (double)4 * p.Wrows * 4 * p.Cin * 2 < 4294967280.0;
}
static int launch_up_phase4(ConvGemmParams p, hipStream_t stream) {
  auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return s; };
  const int bn = p.N % 160 == 0 ? 160 : 128;
  p.W = p.W_up4;
  p.ks = 2; p.up = 0; p.pad = 1;               // (the kernel takes the padding of its phase: 1 - dy, 1 - dx)
  p.Hi = p.Ho = p.Hs; p.Wi = p.Wo = p.Ws;      // GEMM rows = pixels of the stored map
  p.M /= 4;
  p.K = 4 * p.Cin; p.ldw = p.K;
  p.k_logical = p.K;                           // FLOPs actually spent (4/9 of the nine-tap form)
  p.phase4 = 1;
  p.splitk = 1; p.ws = nullptr;
  p.howo_shift = lg2(p.Ho * p.Wo);
  p.wo_shift = lg2(p.Wo);
  p.pp_epilogue = 2;                           // wave-private transposition: the only epilogue that knows the phase's row map
  p.k_tap_inner = 1;
  p.fast_taps = 1;
  p.pp_stagger = g_af_knobs.pp_stagger;
  AfGemmPlan pl;
  pl.tile = bn == 160 ? 5 : 4; pl.splitk = 1; pl.ws_bytes = 0; pl.halo_tw = 0; pl.group_m = 1;
  plan_group_m(pl, p);
  p.group_m = pl.group_m > 0 ? pl.group_m : 1;
  af_set_last_plan(pl);
  g_af_plan_counts[13] += 1;
  AfProfScope prof(bn == 160 ? AF_K_PP160_GATHER : AF_K_PP128, stream, 2.0 * p.M * (double)p.N * p.K * 4,
                   ((double)p.M * p.Cin + 4.0 * p.N * p.K + 4.0 * p.M * p.N) * 2);
  dim3 grid(((p.M + 255) / 256) * (p.N / bn), 4, 1);
  return bn == 160 ? launch_pp_one<160, 0, true, false, 2>(p, grid, stream) : launch_pp_one<128, 0, true, false, 2>(p, grid, stream);
}

// GroupNorm partial sums from the epilogue: the eight-wave kernels (gathering / LDS-halo / fp8) with bf16 outputs in ONE K
// slice, 3x3 / stride 1 / no upsampling (so neither the row-panel nor the phase-decomposed launches apply), whole 64-row
// slabs inside one sample, every group inside one wave's column half
bool af_conv_gn_stats_ok(const ConvGemmParams& p, const AfGemmPlan& pl, int cpg) {
  auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
  const int hn = pl.tile == 5 ? 80 : 64;
  return g_af_knobs.gn_producer && pl.tile >= 4 && pl.splitk <= 1 && pl.halo_tw != 8 && p.ks == 3 && p.stride == 1 && p.up == 0 &&
         p.epilogue != AF_EPI_GEGLU && !p.ln_stats && !p.ln_stats_out && cpg >= 2 && cpg % 2 == 0 && hn % cpg == 0 &&
         p.N == 32 * cpg && pow2(p.Ho * p.Wo) && p.Ho * p.Wo >= 64 && p.M % 64 == 0;
}

// Which row-panel kernel a bf16 launch with these (validated) parameters takes in ONE K slice: 0 none, 1 GEGLU K = 320,
// 2 plain K = 320 (LayerNorm consumer / producer, residual), 3 plain K = 1280 -> 1280, 4 GEGLU K = 640, 5 plain K = 640
// (N >= 1920), 6 the 128 x 160 tile GEMM for few rows (not a row-panel kernel: LayerNorm / GroupNorm consumers never get it).
// The model asks before it decides who finalises LayerNorm statistics (ConvGemmParams::ln_parts_n).
int af_conv_rowpanel_kind(const ConvGemmParams& p, int batch) {
  const int lvl = g_af_knobs.geglu_rowpanel;
  if (!lvl || batch != 1 || p.ks != 1 || p.pad != 0 || p.stride != 1 || p.up != 0 || p.splitk > 1 || p.rowbias || p.fp8 ||
      p.Cin != p.K || p.ldc < p.Cin)
    return 0;
  const bool geglu = p.epilogue == AF_EPI_GEGLU;
  // (the two-slot 128 x 160 form on [65536, 320] -> 320 instead of the row-panel kernel: 27.8 -> 26.0 us alone, but the forward,
  // where most of these launches carry LayerNorm epilogues, 15.30 -> 15.37 ms: not taken)
  // (a consumer-side GroupNorm needs whole panels inside one sample: 256-row panels at K = 320, 128-row ones at K = 640)
  if (p.K == RowPanelCfg::K && p.M >= 128 * RowPanelCfg::BM) {
    if (p.gn_ab && (p.gn_hw <= 0 || p.gn_hw % RowPanelCfg::BM != 0)) return 0;
    if (geglu) return (p.N % RowPanelCfg::BN == 0 && !p.residual && !p.ln_stats_out) ? 1 : 0;
    return (lvl >= 2 && p.N % 160 == 0 && p.alpha == 1.0f && !(p.ln_stats && p.ln_stats_out) && (!p.residual || p.ldr % 4 == 0)) ? 2 : 0;
  }
  // few rows, plain epilogue: the 128 x 160 tile GEMM where it puts 128 .. 512 workgroups on the chip and the 256 x 160 tile
  // would leave a third of it idle (16x16 level: [4096, 1280] -> 1280)
  if (g_af_knobs.gemm_m128 && !geglu && !(p.ln_stats && p.ln_stats_out) && (!p.ln_stats || p.alpha == 1.0f) && !p.gn_ab && p.M % 128 == 0 &&
      p.M <= 8192 && p.N % 160 == 0 &&
      p.K % 64 == 0 && p.K >= 256 && (!p.residual || p.ldr % 4 == 0) && p.ldo % 8 == 0 && ((__UINTPTR_TYPE__)p.out & 15) == 0) {
    const long nb128 = (long)(p.M / 128) * (p.N / 160), nb256 = (long)((p.M + 255) / 256) * (p.N / 160);
    if (nb128 >= 128 && nb128 <= 512 && nb256 <= 170) return 6;
    // ... and where 128-row tiles come out as whole rounds of 256 workgroups while 256-row tiles do not ([4096, 1280] -> 3840:
    // 768 against 384 = one and a half rounds; 59 -> 52 us)
    if (nb128 % 256 == 0 && nb128 <= 1024 && nb256 % 256 != 0) return 6;
  }
  // [16384, 640] -> 640 of the 32x32 level (25 per forward, LayerNorm epilogues included): 512 half-size tiles on a two-slot
  // ring, two workgroups per CU covering each other's prologue and epilogue, instead of 256 tiles of ten K steps each
  // (25.1 -> 21.4 us; forward 15.45 -> 15.37 ms)
  if (g_af_knobs.gemm_m128 && !geglu && !(p.ln_stats && p.ln_stats_out) && (!p.ln_stats || p.alpha == 1.0f) && !p.gn_ab && p.M % 128 == 0 &&
      p.M > 8192 && p.M <= 16384 && p.N % 160 == 0 && p.K % 64 == 0 && p.K >= 256 && p.K <= 1280 && (!p.residual || p.ldr % 4 == 0) &&
      p.ldo % 8 == 0 && ((__UINTPTR_TYPE__)p.out & 15) == 0 && (long)(p.M / 128) * (p.N / 160) == 512)
    return 6;
  // (longer K or more rounds of this form measured slower: [16384, 2560] -> 640 58 -> 62-67 us, [65536, 1280] -> 320 72 -> 78 us)
  if (p.K == 1280)
    return (lvl >= 4 && p.N == 1280 && p.M >= 4096 && !geglu && p.alpha == 1.0f && !p.ln_stats && !p.ln_stats_out && !p.gn_ab &&
            (!p.residual || p.ldr % 4 == 0)) ? 3 : 0;
  if (p.K == 640 && lvl >= 3 && p.M >= 16384 && !p.residual) {
    // (proj_in of a 32x32-level transformer with its GroupNorm applied in this kernel's prologue was built in round 3 and measured
    // slower -- N = 640 is short for this kernel, the tiled one is ahead on the bare GEMM by what the pass costs: 15.64 vs 15.60 ms)
    if (p.gn_ab) return 0;
    if (p.ln_stats_out) return 0;
    if (geglu) return p.N % 128 == 0 ? 4 : 0;
    return (p.N % 160 == 0 && p.N >= 1920 && p.alpha == 1.0f) ? 5 : 0;
  }
  return 0;
}

template <typename T>
int af_launch_conv_gemm(const ConvGemmParams& p_in, int batch, hipStream_t stream, const AfGemmPlan* plan, void* ws) {
  constexpr int BK = 128 / sizeof(T);
  ConvGemmParams p = p_in;
  if (p.fp8) {
    if constexpr (sizeof(T) == 2) return batch == 1 ? launch_conv_gemm_fp8(p, stream, plan, ws) : (af_set_error_msg("conv_gemm fp8: no batched form"), -1);
    else { af_set_error_msg("conv_gemm: fp8 operands need the bf16 storage mode"); return -1; }
  }
  if (p.gn_ab && sizeof(T) != 2) { af_set_error_msg("conv_gemm: consumer-side GroupNorm exists on the bf16 row-panel kernels only"); return -1; }
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
  if (p.epilogue == AF_EPI_GEGLU && p.N % 64 != 0) { af_set_error_msg("conv_gemm: GEGLU needs N%%64==0"); return -1; }
  if (p.M <= 0 || p.N <= 0) return 0;
  {
    // the gathers address activations and weights with 32-bit byte offsets from one buffer base: refuse operands that
    // do not fit instead of wrapping around silently (SD-1.5 at batch 8 is ~0.25 GB per activation, the VAE ~1.1 GB)
    const int HoWo = p.Ho * p.Wo > 0 ? p.Ho * p.Wo : 1;
    const double nb = (double)((p.M + HoWo - 1) / HoWo);
    const double src_bytes = nb * (double)p.src_batch_stride * sizeof(T), w_bytes = (double)p.Wrows * p.ldw * sizeof(T);
    if (src_bytes >= 4294967280.0 || w_bytes >= 4294967280.0) {
      af_set_error_msg("conv_gemm: operand of %.2f GB exceeds the 4 GB range of the 32-bit gather offsets (split the batch)",
                       (src_bytes > w_bytes ? src_bytes : w_bytes) / 1e9);
      return -1;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (up_phase4_ok(p, batch)) return launch_up_phase4(p, stream);
  }
  AfGemmPlan pl = plan ? *plan : af_plan_conv_gemm(p, batch, (int)sizeof(T));
  if (pl.splitk > 1 && !ws) pl.splitk = 1;  // no workspace supplied: fall back to one slice
  if constexpr (sizeof(T) == 2) {
    // the 128 x 160 tile GEMM fills the chip in one K slice where the 256-row tile was planned over two
    if (pl.splitk > 1) {
      ConvGemmParams q = p;
      q.splitk = 1;
      if (af_conv_rowpanel_kind(q, batch) == 6) pl.splitk = 1;
    }
  }
  p.splitk = pl.splitk;
  p.ws = ws;
  af_set_last_plan(pl);
  if (g_af_knobs.plan_log)
    fprintf(stderr, "[af plan] M=%ld N=%d K=%d ks=%d stride=%d up=%d HoWo=%dx%d batch=%d tile=%d halo=%d splitk=%d rowpanel=%d res=%d geglu=%d ln=%d\n",
            (long)p.M, p.N, p.K, p.ks, p.stride, p.up, p.Ho, p.Wo, batch, pl.tile, pl.halo_tw, pl.splitk, af_conv_rowpanel_kind(p, batch),
            p.residual ? 1 : 0, p.epilogue == AF_EPI_GEGLU ? 1 : 0, (p.ln_stats || p.ln_stats_out) ? 1 : 0);
  g_af_plan_counts[(pl.halo_tw && pl.halo_tw != 256 && pl.halo_tw != 8) ? 6 : (pl.tile >= 0 && pl.tile < 6 ? pl.tile : 0)] += 1;
  if (pl.halo_tw == 256) g_af_plan_counts[11] += 1;
  if (pl.splitk > 1) g_af_plan_counts[7] += 1;
  if (p.ln_stats) g_af_plan_counts[8] += 1;
  if (p.ln_stats_out) g_af_plan_counts[9] += 1;
  {
    auto lg2 = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (v > 0 && (1 << s) == v) ? s : -1; };
    p.howo_shift = lg2(p.Ho * p.Wo);
    p.wo_shift = lg2(p.Wo);
  }
  p.group_m = pl.group_m > 0 ? pl.group_m : 1;
  p.pp_epilogue = g_af_knobs.pp_direct < 0 ? 0 : (g_af_knobs.pp_direct ? 2 : 1);
  if (p.gn_stats_out) {
    if (sizeof(T) != 2 || batch != 1 || !af_conv_gn_stats_ok(p, pl, p.gn_cpg)) {
      af_set_error_msg("conv_gemm: GroupNorm partial sums asked of a launch that cannot write them (ask af_conv_gn_stats_ok first)");
      return -1;
    }
    p.pp_epilogue = 2;   // they are summed from the wave-private transposition tile of the direct epilogue
    g_af_plan_counts[14] += 1;
  }
  p.k_tap_inner = (p.ks > 1 && g_af_knobs.conv_tap_inner) ? 1 : 0;
  p.fast_taps = g_af_knobs.conv_fast_taps;
  p.pp_stagger = g_af_knobs.pp_stagger;
  const int prof_cls = pl.halo_tw == 256 ? AF_K_HALO8
                       : pl.tile == 5 ? ((p.ks == 1 && p.pad == 0) ? AF_K_PP160_PLAIN : AF_K_PP160_GATHER)
                                      : (pl.tile == 4 ? AF_K_PP128 : AF_K_CONV_GEMM);
  AfProfScope prof(prof_cls, stream, 2.0 * p.M * (double)p.N * (p.k_logical ? p.k_logical : p.K) * batch,
                   ((double)p.M * p.K / (p.ks * p.ks) + (double)p.N * p.K + (double)p.M * p.N) * batch * sizeof(T));
  int rc;
  int rk_pre = 0;
  if constexpr (sizeof(T) == 2) rk_pre = af_conv_rowpanel_kind(p, batch);
  if ((p.ln_stats || p.ln_stats_out) && rk_pre != 6 && !(pl.tile >= 4 && !pl.halo_tw)) {   // (the 128 x 160 GEMM has both epilogues whatever was planned)
    af_set_error_msg("conv_gemm: LayerNorm-fused launch planned on a kernel without that epilogue (tile %d)", pl.tile);
    return -1;
  }
  // the row-panel kernels (activation rows resident in registers): K = 320 / 640 / 1280 GEMMs with enough rows
  if constexpr (sizeof(T) == 2) {
    const int rk = rk_pre;
    if (p.ln_parts_n > 0 && !rk) {
      af_set_error_msg("conv_gemm: un-finalised LayerNorm statistics handed to a launch that is not a row-panel one");
      return -1;
    }
    if (p.gn_ab && !rk) {
      af_set_error_msg("conv_gemm: consumer-side GroupNorm asked of a launch that is not a row-panel one (ask af_conv_rowpanel_kind first)");
      return -1;
    }
    if (p.gn_ab) g_af_gn_consumer_launches += 1;
    if (rk) g_af_plan_counts[12] += 1;
    switch (rk) {
      case 1: return launch_geglu_rowpanel(p, stream);
      case 2: return launch_plain_rowpanel(p, stream);
      case 3: return launch_plain_rowpanel_k1280(p, stream);
      case 4: return launch_geglu_rowpanel(p, stream, true);
      case 5: return launch_plain_rowpanel(p, stream, true);
      case 6: return launch_gemm_m128(p, stream);
      default: break;
    }
  }
  if (pl.halo_tw == 8) {
    if constexpr (sizeof(T) == 2) {
      rc = af_launch_conv_s8(p, stream);
    } else {
      af_set_error_msg("conv_gemm: the 8 x 8-map kernel is bf16 only");
      return -1;
    }
    if (rc) return rc;
    if (p.splitk > 1) {
      const long nq = (long)p.M * (p.N >> 2);
      unsigned blocks = (unsigned)((nq + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, stream, p);
      HIP_CHECK_RET(hipGetLastError());
    }
    return 0;
  }
  if (pl.halo_tw == 256) {
    if constexpr (sizeof(T) == 2) {
      rc = launch_halo8(p, stream);
    } else {
      af_set_error_msg("conv_gemm: the eight-wave halo kernel is bf16 only");
      return -1;
    }
    if (rc) return rc;
    if (p.splitk > 1) {
      const long nq = (long)p.M * (p.N >> 2);
      unsigned blocks = (unsigned)((nq + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, stream, p);
      HIP_CHECK_RET(hipGetLastError());
    }
    return 0;
  }
  if (pl.halo_tw) {
    const bool bn128 = pl.tile == 0 || pl.tile == 1;
    if (pl.halo_tw == 32) rc = bn128 ? launch_halo<T, 32, 128>(p, stream) : launch_halo<T, 32, 64>(p, stream);
    else rc = bn128 ? launch_halo<T, 16, 128>(p, stream) : launch_halo<T, 16, 64>(p, stream);
    return rc;
  }
  switch (pl.tile) {
    case 4:
    case 5:
      if constexpr (sizeof(T) == 2) {
        rc = pl.tile == 4 ? launch_pp<128>(p, stream) : launch_pp<160>(p, stream);
      } else {
        af_set_error_msg("conv_gemm: ping-pong tiles are bf16 only");
        return -1;
      }
      break;
    case 0: rc = launch_cfg<T, 128, 128>(p, batch, stream); break;
    case 1: rc = launch_cfg<T, 64, 128>(p, batch, stream); break;
    case 2: rc = launch_cfg<T, 128, 64>(p, batch, stream); break;
    default: rc = launch_cfg<T, 64, 64>(p, batch, stream); break;
  }
  if (rc) return rc;
  if (p.splitk > 1) {
    const long nq = (long)p.M * (p.N >> 2);
    unsigned blocks = (unsigned)((nq + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, stream, p);
    HIP_CHECK_RET(hipGetLastError());
  }
  return 0;
}

template int af_launch_conv_gemm<bf16>(const ConvGemmParams&, int, hipStream_t, const AfGemmPlan*, void*);
template int af_launch_conv_gemm<float>(const ConvGemmParams&, int, hipStream_t, const AfGemmPlan*, void*);
