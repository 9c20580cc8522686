// gemm.hip -- C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias): the encoder's dense contractions on bf16 MFMA
// (rows B1-B4: QKV / attention-output / FFN / spec-head projections).
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights (out, in)), which is exactly the
// lane layout of v_mfma_f32_16x16x32_bf16 (lane l: row l&15, k = 8 (l>>4) .. +7).
//
//   tile      : 128 x 128 x 64 per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles
//   staging   : global -> registers -> LDS, double-buffered, loads of tile t+1 issued before the MFMAs of tile t and
//               written after them (one barrier per K-tile)
//   LDS image : [row][8 x 16-B chunks], chunk' = chunk ^ ((row >> 1) & 7): conflict-free ds_read_b128 fragments
//   epilogue  : accumulators -> LDS (fp32) -> coalesced 16-B row stores with bias / GELU / fp32 residual fused,
//               bf16 and / or fp32 output
//   grid      : XCD-aware bijective remap so the blocks that share an A panel run on one XCD back to back
// Bound: MFMA (2 M N K flop against the 2.5 PFLOP/s dense bf16 peak).
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"

namespace se {

constexpr int kBM = 128, kBN = 128, kBK = 64;
constexpr int kGThreads = 256;
constexpr int kTileBytes = kBM * kBK * 2;      // 16 KiB per operand tile

__device__ __forceinline__ int swz_off(int row, int chunk) {      // byte offset inside a [128][64] bf16 tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == SE_ACT_GELU) return gelu_erf(v);
  if (act == SE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SE_ACT_EXP) return __expf(v);
  if (act == SE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

__global__ __launch_bounds__(kGThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, int M, int N, int K, int act, uint16_t* __restrict__ out_bf16,
    float* __restrict__ out_f32, int ldc, int tiles_m, int tiles_n, int vec_ok) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A tile + B tile) = 64 KiB; reused by the epilogue

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap (blocks b and b+8 share an XCD): each XCD walks a contiguous range of tile ids,
  // n fastest, so an A panel is fetched from HBM once per XCD and the weights stay L2-resident.
  const int nwg = tiles_m * tiles_n;
  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = id / tiles_n, tn = id - tm * tiles_n;
  const int m0 = tm * kBM, n0 = tn * kBN;

  // staging map: 4 x 16-B chunks of A and 4 of W per thread per K-tile; chunk index = tid + 256 i -> row = idx>>3
  // Rows past M / N are CLAMPED to the last valid row instead of zero-filled: they only feed accumulator rows /
  // columns that the epilogue never stores, and a clamped global_load needs no branch or select.
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const int srow = tid >> 3, sch = tid & 7;          // rows srow + 32 i
  const uint16_t* a_p0 = A + (size_t)min(m0 + srow, M - 1) * lda + sch * 8;
  const uint16_t* a_p1 = A + (size_t)min(m0 + srow + 32, M - 1) * lda + sch * 8;
  const uint16_t* a_p2 = A + (size_t)min(m0 + srow + 64, M - 1) * lda + sch * 8;
  const uint16_t* a_p3 = A + (size_t)min(m0 + srow + 96, M - 1) * lda + sch * 8;
  const uint16_t* w_p0 = W + (size_t)min(n0 + srow, N - 1) * ldw + sch * 8;
  const uint16_t* w_p1 = W + (size_t)min(n0 + srow + 32, N - 1) * ldw + sch * 8;
  const uint16_t* w_p2 = W + (size_t)min(n0 + srow + 64, N - 1) * ldw + sch * 8;
  const uint16_t* w_p3 = W + (size_t)min(n0 + srow + 96, N - 1) * ldw + sch * 8;
#define SE_ISSUE_LOADS(k0)                                   \
  do {                                                       \
    ra0 = *reinterpret_cast<const uint4*>(a_p0 + (k0));      \
    ra1 = *reinterpret_cast<const uint4*>(a_p1 + (k0));      \
    ra2 = *reinterpret_cast<const uint4*>(a_p2 + (k0));      \
    ra3 = *reinterpret_cast<const uint4*>(a_p3 + (k0));      \
    rb0 = *reinterpret_cast<const uint4*>(w_p0 + (k0));      \
    rb1 = *reinterpret_cast<const uint4*>(w_p1 + (k0));      \
    rb2 = *reinterpret_cast<const uint4*>(w_p2 + (k0));      \
    rb3 = *reinterpret_cast<const uint4*>(w_p3 + (k0));      \
  } while (0)
  // rows srow + 32 i share (row >> 1) & 7 parity pattern only through srow: compute the 4 swizzled offsets once
  const int so0 = swz_off(srow, sch), so1 = swz_off(srow + 32, sch), so2 = swz_off(srow + 64, sch), so3 = swz_off(srow + 96, sch);
#define SE_WRITE_LDS(buf)                                                   \
  do {                                                                      \
    char* a_w = smem + (buf) * 2 * kTileBytes;                              \
    char* b_w = a_w + kTileBytes;                                           \
    *reinterpret_cast<uint4*>(a_w + so0) = ra0;                             \
    *reinterpret_cast<uint4*>(a_w + so1) = ra1;                             \
    *reinterpret_cast<uint4*>(a_w + so2) = ra2;                             \
    *reinterpret_cast<uint4*>(a_w + so3) = ra3;                             \
    *reinterpret_cast<uint4*>(b_w + so0) = rb0;                             \
    *reinterpret_cast<uint4*>(b_w + so1) = rb1;                             \
    *reinterpret_cast<uint4*>(b_w + so2) = rb2;                             \
    *reinterpret_cast<uint4*>(b_w + so3) = rb3;                             \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / kBK;
  SE_ISSUE_LOADS(0);
  SE_WRITE_LDS(0);
  __syncthreads();

  const int frow = lane & 15, fch = lane >> 4;
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) SE_ISSUE_LOADS((t + 1) * kBK);
    const char* a_s = smem + cur * 2 * kTileBytes;
    const char* b_s = a_s + kTileBytes;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(a_s + swz_off(wm * 64 + i * 16 + frow, s * 4 + fch));
        bfr[i] = *reinterpret_cast<const bf16x8*>(b_s + swz_off(wn * 64 + i * 16 + frow, s * 4 + fch));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) SE_WRITE_LDS(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS fp32 [128][128] (C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
  float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + i * 16 + (lane >> 4) * 4 + r, col = wn * 64 + j * 16 + (lane & 15);
        cs[row * kBN + col] = acc[i][j][r];
      }
  __syncthreads();
  // 16 threads per row, 8 columns each; 16 rows per pass
  const int cg = (tid & 15) * 8;
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 16 + (tid >> 4);
    const int gm = m0 + row, gn = n0 + cg;
    if (gm >= M || gn >= N) continue;
    const float4 c0 = *reinterpret_cast<const float4*>(cs + row * kBN + cg);
    const float4 c1 = *reinterpret_cast<const float4*>(cs + row * kBN + cg + 4);
    float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const bool full = (gn + 8 <= N) && vec_ok;
    if (full) {
      if (bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + gn), b1 = *reinterpret_cast<const float4*>(bias + gn + 4);
        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], act);
      if (residual) {
        const float* rp = residual + (size_t)gm * ldc + gn;
        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      }
      if (out_f32) {
        float* op = out_f32 + (size_t)gm * ldc + gn;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      if (out_bf16) {
        uint4 p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        *reinterpret_cast<uint4*>(out_bf16 + (size_t)gm * ldc + gn) = p;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (gn + e < N) {
          float x = cs[row * kBN + cg + e] + (bias ? bias[gn + e] : 0.f);
          x = act_apply(x, act);
          if (residual) x += residual[(size_t)gm * ldc + gn + e];
          if (out_f32) out_f32[(size_t)gm * ldc + gn + e] = x;
          if (out_bf16) out_bf16[(size_t)gm * ldc + gn + e] = f2bf(x);
        }
      }
    }
  }
}

}  // namespace se

extern "C" int se_gemm2_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, int variant, void* stream);
extern "C" int se_gemm3_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, void* stream);
extern "C" int se_gemm6_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, void* stream);
extern "C" int se_gemm5_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, void* stream);
extern "C" int se_gemm7_plain_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32, int M, int K,
                                     uint16_t* out_bf16, float* out_f32, void* stream);      // gemm4.hip
static int g_gemm_variant = -1;   // -1: read SE_AMD_GEMM once (1 = register-staged 128x128 kernel of this file, 2 = gemm2.hip lockstep, 3 = gemm2.hip ping-pong, 4 = gemm2.hip 128x128 x 2 workgroups / CU, 5 = gemm3.hip 256x256 [default])

extern "C" int se_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                            const float* residual_f32, int M, int N, int K, int act,
                            uint16_t* out_bf16, float* out_f32, int ldc, void* stream) {
  SE_REQUIRE(A && W && (out_bf16 || out_f32), "se_gemm_bf16: null argument");
  SE_REQUIRE(M > 0 && N > 0 && K > 0 && K % se::kBK == 0, "se_gemm_bf16: K=%d must be a positive multiple of %d", K, se::kBK);
  SE_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "se_gemm_bf16: lda/ldw must be >= K and multiples of 8 (16-B rows)");
  SE_REQUIRE(ldc >= N, "se_gemm_bf16: ldc < N");
  SE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "se_gemm_bf16: operands must be 16-B aligned");
  // vector epilogue needs 16-B aligned rows of every tensor it touches
  const int vec_ok = (ldc % 8 == 0) && (((uintptr_t)out_bf16 | (uintptr_t)out_f32 | (uintptr_t)residual_f32 | (uintptr_t)bias) % 16 == 0);
  if (g_gemm_variant < 0) {
    const char* e = getenv("SE_AMD_GEMM");
    g_gemm_variant = (e && e[0] >= '1' && e[0] <= '5') ? (e[0] - '0') : 5;
  }
  if (g_gemm_variant == 5) {      // 256 x 256 kernel for wide outputs (enough tiles to fill 256 CUs several times), 256 x 128 ping-pong otherwise
    // serving-size batches (M = B*T <= 8 utterances of 10 s): a 1 001-row GEMM makes 4 row tiles of 256 -- 36 workgroups for the QKV
    // projection on 256 CUs.  128 x 128 tiles at two workgroups per CU fill the chip 4x better: 0.85 vs 1.85 ms per utterance pass
    // at B = 1, 1.81 vs 2.61 ms at B = 8 (tools/small_batch_sweep.sh); from B = 16 on the large tiles win again.
    // Round 2, with the 64-deep kernels: the crossover moved down to ~4 000 rows (ms per pass, small / large tiles: B = 4 1.14 / 1.14,
    // B = 5 1.30 / 1.22, B = 6 1.51 / 1.37, B = 8 1.69 / 1.57), so the threshold is 4 096 rows (was 8 192).
    static int use5 = -1, use6 = 0, small_m = 4096, min_n6 = 1536;
    if (use5 < 0) {
      const char* e5 = getenv("SE_AMD_GEMM5");       // developer switch: 1 = the one-wave-per-SIMD 256 x 256 kernel (gemm5.hip) for the bf16-output wide GEMMs
      use5 = e5 ? atoi(e5) : 0;
      const char* e6 = getenv("SE_AMD_GEMM6");       // 1 = the 64-deep eight-phase 256 x 256 kernel (gemm6.hip) for the wide GEMMs
      use6 = e6 ? atoi(e6) : 1;                      // default since round 2: +5-8 % over gemm3 on every wide shape (profiles/README.md)
      const char* sm = getenv("SE_AMD_GEMM_SMALL_M");   // row count up to which the 128 x 128 kernel is used (kernel benchmarks set 0)
      small_m = sm ? atoi(sm) : 4096;
      if (const char* mn = getenv("SE_AMD_GEMM6_MIN_N")) min_n6 = atoi(mn);      // A/B: output width from which the 256 x 256 x 64 kernel is used
    }
    if (M <= small_m) return se_gemm2_launch(A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, 4, stream);
    // A/B (SE_AMD_GEMM_TAILSPLIT=1): the persistent 256 x 256 kernel deals tiles_m x tiles_n tiles to 256 workgroups; when the count is r + f rounds with a
    // small fraction f (QKV at B = 32: 1 134 tiles = 4.43 rounds, 110 workgroups own five tiles and 146 four), the last rows go to the 128 x 128
    // two-workgroups-per-CU kernel instead: whole rounds of large tiles, then one round of small ones
    static int tailsplit = -1;
    if (tailsplit < 0) { const char* e = getenv("SE_AMD_GEMM_TAILSPLIT"); tailsplit = e ? atoi(e) : 0; }
    if (tailsplit && use6 && N >= min_n6 && N % 256 == 0 && out_bf16 && !out_f32 && !residual_f32) {
      const int tiles_n = N / 256, tiles_m = (M + 255) / 256, tiles = tiles_m * tiles_n, rounds = tiles / 256, rest = tiles - rounds * 256;
      const int m1 = (rounds * 256 / tiles_n) * 256;                     // rows of the whole rounds
      if (rounds >= 3 && rest > 0 && rest <= 160 && m1 > 0 && M - m1 > 0 && M - m1 <= small_m) {
        const int rc1 = se_gemm_bf16(A, lda, W, ldw, bias, nullptr, m1, N, K, act, out_bf16, nullptr, ldc, stream);
        if (rc1 != SE_OK) return rc1;
        return se_gemm2_launch(A + (size_t)m1 * lda, lda, W, ldw, bias, nullptr, M - m1, N, K, act, out_bf16 + (size_t)m1 * ldc, nullptr, ldc, vec_ok, 4, stream);
      }
    }
    // round 4: N = 768 outputs with a long reduction (the training path's FFN2 forward and FFN1 input gradient, K = 3072: 190 vs 205 us) also
    // run faster on the 256 x 256 x 64 kernel in spite of its 1.48-round tile count; at K = 768 the 256 x 128 ping-pong kernel keeps its lead (66 vs 70 us)
    // round 4 (later): N = 768 with K >= 1536 goes to the row-complete kernel without its LayerNorm (gemm4.hip: 251 tiles of 128 x 768 = ONE round,
    // where 256 x 256 tiles make 1.48).  Same box, interleaved (profiles/r04_revalidate.txt): fine-tune step 18.59 / 18.70 -> 18.30 / 18.28 ms.
    // SE_AMD_GEMM7_PLAIN=0 restores the 256 x 256 x 64 kernel.  (A first "A/B" of this switch read 19.55 against 19.44 ms -- both arms had run a stale
    // library whose rebuild had failed behind a `| tail`; build.py now deletes the library when a rebuild fails.)
    static int plain7 = -1;
    static int plain7_mink = 1536;
    if (plain7 < 0) { const char* e = getenv("SE_AMD_GEMM7_PLAIN"); plain7 = e ? atoi(e) : 1; if (const char* k = getenv("SE_AMD_GEMM7_PLAIN_MINK")) plain7_mink = atoi(k); }
    // ... from 160 row tiles on (M > 20 352: the x3 batch sweep, profiles/r04_x3_rowln_batch.txt); below that the 256 x 256 / 256 x 128 kernels cover the chip better
    if (plain7 && N == 768 && ldc == 768 && K >= plain7_mink && act == SE_ACT_IDENTITY && vec_ok && (M + 127) / 128 >= 160) {
      const int rc7 = se_gemm7_plain_launch(A, lda, W, ldw, bias, residual_f32, M, K, out_bf16, out_f32, stream);
      if (rc7 <= 0) return rc7;
    }
    if (use6 && (N >= min_n6 || (N >= 768 && K >= 1536))) {
      const int rc6 = se_gemm6_launch(A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, stream);
      if (rc6 <= 0) return rc6;
    }
#ifdef SE_AMD_EXPERIMENTS
    if (use5 && N >= 1536) {
      const int rc5 = se_gemm5_launch(A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, stream);
      if (rc5 <= 0) return rc5;
    }
#else
    (void)use5;
#endif
    const int rc = (N >= 1536) ? se_gemm3_launch(A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, stream) : 1;
    if (rc <= 0) return rc;
    return se_gemm2_launch(A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, 3, stream);
  }
  if (g_gemm_variant >= 2)
    return se_gemm2_launch(A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, g_gemm_variant, stream);
  const int tiles_m = (M + se::kBM - 1) / se::kBM, tiles_n = (N + se::kBN - 1) / se::kBN;
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    attr_set = true;
  }
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, se::as_stream(stream));
  hipLaunchKernelGGL(se::gemm_bf16_kernel, dim3(tiles_m * tiles_n), dim3(se::kGThreads), 65536, se::as_stream(stream), A, lda, W,
                     ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, tiles_m, tiles_n, vec_ok);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
