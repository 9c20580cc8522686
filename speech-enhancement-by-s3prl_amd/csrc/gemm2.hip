// gemm2.hip -- the production bf16 GEMM: C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)   (see gemm.hip for the contract)
//
//   tile     : 256 (M) x 128 (N) x 64 (K) per 512-thread workgroup, 8 waves as 4 x 2, wave tile 64 x 64
//              (4 x 4 v_mfma_f32_16x16x32_bf16, 64 accumulator registers); one workgroup per CU, 2 waves per SIMD
//   staging  : global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip) into a 3-deep LDS ring (3 x 48 KiB);
//              tiles t+1 and t+2 are in flight while tile t is multiplied: counted s_waitcnt vmcnt(6) + ONE raw
//              s_barrier per K-tile (never __syncthreads: its fence would drain the DMA queue)
//   LDS image: DMA writes are lane-linear (wave base + 16 lane), so the XOR swizzle chunk' = chunk ^ ((row>>1)&7) is
//              applied to the per-lane SOURCE address and again on the ds_read_b128 side (both-sides-or-neither)
//   epilogue : MFMA operands swapped (C^T accumulators): each lane owns 4 consecutive output columns and stores them
//              straight from registers (8 / 16 B per lane, whole 128-B lines per 4 tiles); bias / GELU / residual fused
//   grid     : XCD-aware bijective remap, n fastest inside an XCD
#include <stdlib.h>
#include "common.h"
#include "bf16.h"
#include "prof.h"

// The raised wave priority around the MFMA clusters (s_setprio 1 ... 0) is OFF: A/B on one box, fine-tune step 19.73 -> 19.52 ms without it
// (the same finding as for the row-complete and the persistent kernels).  SE_AMD_EXTRA_DEFINES=-DSE_AMD_SETPRIO python build.py --force brings it back.
#ifdef SE_AMD_SETPRIO
#define SE_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define SE_SETPRIO(x) ((void)0)
#endif

namespace se {

constexpr int k2BN = 128, k2BK = 64;
// Tile configuration: WR = wave rows (BM = 64 WR, threads = 128 WR), STAGES = LDS ring depth.
//   <4, 3>: 256 x 128 tile, 8 waves, 3 x 48 KiB ring, ONE workgroup per CU (ping-pong capable)
//   <2, 2>: 128 x 128 tile, 4 waves, 2 x 32 KiB ring, TWO independent workgroups per CU: one workgroup's epilogue
//           (pure HBM traffic) overlaps the other's MFMA main loop
// diagnostic stamps (dbg bit 4): lane 0 of every wave of workgroup `blockIdx.x < 8` appends s_memtime values to
// the buffer passed as `residual` (timing-only build; results are garbage)
// Compiled in only with -DSE_AMD_STAMPS (SE_AMD_BUILD_STAMPS=1 python build.py): even switched off, the exec-masked stamp sites cost the
// hot loops measurably (13 % in gemm4.hip's sub-steps).
__device__ __forceinline__ void stamp(unsigned long long* buf, int& idx, bool on) {
#ifdef SE_AMD_STAMPS
  if (on) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    buf[idx++] = t;
  }
#endif
}

template <int WR, int STAGES>
struct G2 {
  static constexpr int BM = 64 * WR, NW = 2 * WR, Threads = 64 * NW;
  static constexpr int ABytes = BM * k2BK * 2, BBytes = k2BN * k2BK * 2, Stage = ABytes + BBytes;
  static constexpr int Lds = STAGES * Stage;
  static constexpr int NA = 4, NB = 8 / WR;          // LDS-DMA instructions per wave per K-tile (A, B)
  static constexpr int NDma = NA + NB;
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ int swz2(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float act2(float v, int act) {
  if (act == SE_ACT_GELU) return gelu_erf(v);
  if (act == SE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SE_ACT_EXP) return __expf(v);
  if (act == SE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

// PINGPONG = 1: waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA cluster while
// its partner fetches the next fragments from LDS (4 raw barriers per K-tile, 16 MFMAs between two of them).
// Epilogue specialisation: ACT = compile-time activation (-1: runtime `act`), EF = flags (-1: all runtime)
//   EF bit 0: fp32 residual added, bit 1: bf16 output, bit 2: fp32 output, bit 3: vector path legal (N % 4 == 0, 16-B rows)
// The specialised forms issue every bias / residual load up front and have no branches, so the compiler emits one
// wait instead of one per access (the all-runtime form is a 15 000-line branch tree that also thrashes the I-cache).
template <int WR, int STAGES, int PINGPONG, int ACT, int EF>
__global__ __launch_bounds__(128 * WR) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm2_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, int M, int N, int K, int act, uint16_t* __restrict__ out_bf16,
    float* __restrict__ out_f32, int ldc, int tiles_m, int tiles_n, int vec_ok, int dbg, int group_m, size_t split_out_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = G2<WR, STAGES>;
  constexpr int k2BM = C::BM, k2ABytes = C::ABytes, k2Stage = C::Stage, k2Stages = STAGES, NW = C::NW;
  static_assert(PINGPONG == 0 || (WR == 4 && STAGES == 3), "ping-pong needs the 8-wave 3-stage configuration");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // split-K (weight-gradient GEMMs reduce over B*T ~ 32 000 rows into a small output): blockIdx.y walks K chunks of
  // length K (the kernel's K argument is the CHUNK length); each split writes its own fp32 partial slab
  if (gridDim.y > 1) {
    A += (size_t)blockIdx.y * K;
    W += (size_t)blockIdx.y * K;
    if (out_f32) out_f32 += (size_t)blockIdx.y * split_out_stride;
  }

  const int nwg = tiles_m * tiles_n;
  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;          // grouped order inside the XCD's id range (see gemm3.hip)
  {
    const int per_group = group_m * tiles_n, grp = id / per_group, first_m = grp * group_m;
    const int gsz = min(tiles_m - first_m, group_m), in = id - grp * per_group;
    tn = in / gsz;
    tm = first_m + (in - tn * gsz);
  }
  const int m0 = tm * k2BM, n0 = tn * k2BN;

  // ---- DMA source pointers.  A stage = BM/8 chunks of 1 KiB (8 rows x 128 B); wave w issues chunks w, w+NW, w+2NW, w+3NW.
  //      lane -> row 8 c + (lane >> 3), LDS position lane & 7 holds logical 16-B chunk (lane & 7) ^ ((row >> 1) & 7).
  const int r8 = lane >> 3, pos = lane & 7;
  const uint16_t* a_src[C::NA];
  const uint16_t* b_src[C::NB];
#pragma unroll
  for (int i = 0; i < C::NA; ++i) {
    const int row = 8 * (i * NW + wave) + r8;
    a_src[i] = A + (size_t)min(((dbg & 8) ? 0 : m0) + row, M - 1) * lda + ((pos ^ ((row >> 1) & 7)) << 3);   // dbg bit 3: every tile loads A panel 0 (L2-resident)
  }
#pragma unroll
  for (int i = 0; i < C::NB; ++i) {
    const int row = 8 * (i * NW + wave) + r8;
    b_src[i] = W + (size_t)min(n0 + row, N - 1) * ldw + ((pos ^ ((row >> 1) & 7)) << 3);
  }
#define SE2_ISSUE(kt, st)                                                                                              \
  do {                                                                                                                 \
    char* sb = smem + (st) * k2Stage + wave * 1024;                                                                    \
    _Pragma("unroll") for (int i = 0; i < C::NA; ++i)                                                                  \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(a_src[i] + (kt) * k2BK), (lds_ptr_t)(sb + i * NW * 1024), 16, 0, 0);  \
    _Pragma("unroll") for (int i = 0; i < C::NB; ++i)                                                                  \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(b_src[i] + (kt) * k2BK), (lds_ptr_t)(sb + k2ABytes + i * NW * 1024), 16, 0, 0); \
  } while (0)

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (dbg & 2) ? 2 : K / k2BK;      // dbg bit 1: timing-only build, main loop cut to 2 K-tiles
  SE2_ISSUE(0, 0);
  if (STAGES == 3 && nk > 1) SE2_ISSUE(1, 1);

  const int frow = lane & 15, fch = lane >> 4;
  int st = 0;
  if constexpr (PINGPONG == 0) {
    for (int t = 0; t < nk; ++t) {
      // tile t landed (this wave's pieces): with a 3-deep ring leave the DMAs of tile t+1 in flight
      if (STAGES == 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NDma) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // everyone's pieces landed AND everyone is done reading the stage refilled next
      if (t + STAGES - 1 < nk) {
        const int st2 = (st + STAGES - 1 >= k2Stages) ? st + STAGES - 1 - k2Stages : st + STAGES - 1;
        SE2_ISSUE(t + STAGES - 1, st2);
      }
      const char* a_s = smem + st * k2Stage;
      const char* b_s = a_s + k2ABytes;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = *reinterpret_cast<const bf16x8*>(a_s + swz2(wm * 64 + i * 16 + frow, s * 4 + fch));
          bfr[i] = *reinterpret_cast<const bf16x8*>(b_s + swz2(wn * 64 + i * 16 + frow, s * 4 + fch));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      st = (st + 1 == k2Stages) ? 0 : st + 1;
    }
    __builtin_amdgcn_s_barrier();
  } else {
    // byte offsets of this lane's fragment rows inside a stage (k-step 0; k-step 1 = chunk + 4 -> XOR-ed again below)
    int a_off[2][4], b_off[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_off[s][i] = swz2(wm * 64 + i * 16 + frow, s * 4 + fch);
        b_off[s][i] = k2ABytes + swz2(wn * 64 + i * 16 + frow, s * 4 + fch);
      }
    const bool late = wave >= 4;       // wave-uniform (readfirstlane above)
    const bool st_on = (dbg & 16) && lane == 0 && blockIdx.x < 8;
    unsigned long long* st_buf = reinterpret_cast<unsigned long long*>(const_cast<float*>(residual)) + ((size_t)blockIdx.x * 8 + wave) * 256;
    int st_i = 0;
    stamp(st_buf, st_i, st_on);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // tile 0 landed (nk >= 2 is guaranteed by the launcher; NDma = 6)
    __builtin_amdgcn_s_barrier();
    if (late) __builtin_amdgcn_s_barrier();                    // stagger: waves 4-7 run one barrier behind
    stamp(st_buf, st_i, st_on);
    for (int t = 0; t < nk; ++t) {
      const char* sb = smem + st * k2Stage;
      bf16x8 af[4], bfr[4];
      // ---------------- k-step 0
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[0][i]);
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + b_off[0][i]);
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(st_buf, st_i, st_on);          // [0] reads(k0) issued
      __builtin_amdgcn_s_barrier();
      stamp(st_buf, st_i, st_on);          // [1] past B0
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nk) {       // ring slot of tile t-1: both groups finished reading it before this barrier
        const int st2 = (st + 2 >= k2Stages) ? st + 2 - k2Stages : st + 2;
        SE2_ISSUE(t + 2, st2);
      }
      SE_SETPRIO(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      SE_SETPRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      stamp(st_buf, st_i, st_on);          // [2] MFMA(k0) issued
      __builtin_amdgcn_s_barrier();
      stamp(st_buf, st_i, st_on);          // [3] past B1
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- k-step 1
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[1][i]);
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + b_off[1][i]);
      }
      // tile t+1 must have landed (every wave's pieces) before the barrier two ahead of its first read
      stamp(st_buf, st_i, st_on);          // [4] reads(k1) issued
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      stamp(st_buf, st_i, st_on);          // [5] DMA wait done
      __builtin_amdgcn_s_barrier();
      stamp(st_buf, st_i, st_on);          // [6] past B0'
      __builtin_amdgcn_sched_barrier(0);
      SE_SETPRIO(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      SE_SETPRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      stamp(st_buf, st_i, st_on);          // [7] MFMA(k1) issued
      __builtin_amdgcn_s_barrier();
      stamp(st_buf, st_i, st_on);          // [8] past B1'
      __builtin_amdgcn_sched_barrier(0);
      st = (st + 1 == k2Stages) ? 0 : st + 1;
    }
    if (!late) __builtin_amdgcn_s_barrier();                   // re-align the two groups
    __builtin_amdgcn_s_barrier();
  }

  // ---- epilogue (all DMAs retired by the vmcnt(0) of the last iteration; all waves past the last reads)
  if (dbg & 1) {          // timing-only build: no epilogue; keep the accumulators live
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 1.2345678e-30f && out_f32) out_f32[0] = sacc;
    return;
  }
  // Operands are swapped (W fragment = MFMA A operand), so the accumulator tile is C^T: col = lane & 15 -> output ROW m,
  // row = 4 (lane >> 4) + r -> 4 CONSECUTIVE output columns n.  Each lane therefore stores 8 B (bf16) / 16 B (fp32)
  // contiguous, and the four j-tiles of one i fill whole 128-B lines of 16 rows: no LDS round trip, no barrier.
  const int mrow = lane & 15, ncol = 4 * (lane >> 4);
  if constexpr (EF >= 0 && (EF & 8)) {
    constexpr bool RES = EF & 1, OBF = EF & 2, OF32 = EF & 4;
    const int a_ = (ACT >= 0) ? ACT : act;
    // N % 4 == 0 here, so a 4-column group is entirely inside or outside N; rows / groups outside are clamped for the
    // loads and predicated for the stores.
    float4 bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = min(n0 + wn * 64 + j * 16 + ncol, N - 4);
      bb[j] = bias ? *reinterpret_cast<const float4*>(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const size_t dbg_mask = (dbg & 4) ? 255 : ~(size_t)0;     // dbg bit 2: all tiles store to rows 0..255 (L2-resident)
    // Interior tiles (all but the last tile row / column) take a branch-free body: with exec-masked stores the
    // compiler cannot count the outstanding memory operations and drains (vmcnt(0)) before every single store.
    const bool interior = (m0 + k2BM <= M) && (n0 + k2BN <= N);      // wave-uniform
#define SE2_EPILOGUE_BODY(PRED)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                          \
    const int gm = m0 + wm * 64 + i * 16 + mrow;                                                                           \
    const bool mok = gm < M;                                                                                               \
    const size_t orow = ((size_t)min(gm, M - 1) & dbg_mask) * ldc;                                                         \
    float4 rr[4];                                                                                                          \
    if constexpr (RES) {                                                                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
        const int gn = min(n0 + wn * 64 + j * 16 + ncol, N - 4);                                                           \
        rr[j] = *reinterpret_cast<const float4*>(residual + orow + gn);                                                    \
      }                                                                                                                    \
    }                                                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
      const int gn = n0 + wn * 64 + j * 16 + ncol;                                                                         \
      float v0 = acc[i][j][0] + bb[j].x, v1 = acc[i][j][1] + bb[j].y, v2 = acc[i][j][2] + bb[j].z, v3 = acc[i][j][3] + bb[j].w; \
      v0 = act2(v0, a_); v1 = act2(v1, a_); v2 = act2(v2, a_); v3 = act2(v3, a_);                                          \
      if constexpr (RES) { v0 += rr[j].x; v1 += rr[j].y; v2 += rr[j].z; v3 += rr[j].w; }                                   \
      if (!(PRED) || (mok && gn < N)) {                                                                                    \
        if constexpr (OF32) *reinterpret_cast<float4*>(out_f32 + orow + gn) = make_float4(v0, v1, v2, v3);                 \
        if constexpr (OBF) *reinterpret_cast<uint2*>(out_bf16 + orow + gn) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)); \
      }                                                                                                                    \
    }                                                                                                                      \
  }
    if (interior) {
      SE2_EPILOGUE_BODY(false)
    } else {
      SE2_EPILOGUE_BODY(true)
    }
#undef SE2_EPILOGUE_BODY
  } else {
    // generic path: any activation / output combination, scalar tail handling (e.g. the 201-column spec-head output)
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + wm * 64 + i * 16 + mrow;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gn = n0 + wn * 64 + j * 16 + ncol;
        const float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        const size_t o = (size_t)min(gm, M - 1) * ldc + gn;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (gm < M && gn + e < N) {
            float x = vv[e] + (bias ? bias[gn + e] : 0.f);
            x = act2(x, act);
            if (residual) x += residual[o + e];
            if (out_f32) out_f32[o + e] = x;
            if (out_bf16) out_bf16[o + e] = f2bf(x);
          }
        }
      }
    }
  }
}

}  // namespace se

// internal launcher (gemm.hip's se_gemm_bf16 dispatches here); arguments already validated.
// variant: 2 = 256x128 lockstep, 3 = 256x128 ping-pong, 4 = 128x128 two-workgroups-per-CU
namespace {
struct GArgs {
  const uint16_t* A; int lda; const uint16_t* W; int ldw; const float* bias; const float* residual; int M, N, K, act;
  uint16_t* out_bf16; float* out_f32; int ldc, vec_ok, dbg; hipStream_t st; int group_m; int splits; size_t split_stride;
};

template <int WR, int STAGES, int PP, int ACT, int EF>
int launch_one(const GArgs& g) {
  using C = se::G2<WR, STAGES>;
  const int tiles_m = (g.M + C::BM - 1) / C::BM, tiles_n = (g.N + se::k2BN - 1) / se::k2BN;
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm2_bf16_kernel<WR, STAGES, PP, ACT, EF>), hipFuncAttributeMaxDynamicSharedMemorySize, C::Lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((se::gemm2_bf16_kernel<WR, STAGES, PP, ACT, EF>), dim3(tiles_m * tiles_n, g.splits), dim3(C::Threads), C::Lds, g.st, g.A, g.lda, g.W,
                     g.ldw, g.bias, g.residual, g.M, g.N, g.K, g.act, g.out_bf16, g.out_f32, g.ldc, tiles_m, tiles_n, g.vec_ok, g.dbg, g.group_m, g.split_stride);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// picks the specialised epilogue when the call matches one of the encoder's forms
template <int WR, int STAGES, int PP>
int launch_cfg(const GArgs& g) {
  const bool vec = g.vec_ok && (g.N % 4 == 0) && (g.ldc % 4 == 0);
  const bool one_out = (g.out_bf16 != nullptr) != (g.out_f32 != nullptr);
  if (vec && one_out && (g.act == SE_ACT_IDENTITY || g.act == SE_ACT_GELU)) {
    const bool gelu = g.act == SE_ACT_GELU, res = g.residual != nullptr, obf = g.out_bf16 != nullptr;
    if (!gelu && !res && obf) return launch_one<WR, STAGES, PP, SE_ACT_IDENTITY, 8 | 2>(g);       // QKV
    if (gelu && !res && obf) return launch_one<WR, STAGES, PP, SE_ACT_GELU, 8 | 2>(g);            // FFN1
    if (!gelu && res && !obf) return launch_one<WR, STAGES, PP, SE_ACT_IDENTITY, 8 | 4 | 1>(g);   // out-proj, FFN2
    if (!gelu && !res && !obf) return launch_one<WR, STAGES, PP, SE_ACT_IDENTITY, 8 | 4>(g);      // input projection
    if (gelu && !res && !obf) return launch_one<WR, STAGES, PP, SE_ACT_GELU, 8 | 4>(g);           // spec-head dense
  }
  return launch_one<WR, STAGES, PP, -1, -1>(g);
}
}  // namespace

extern "C" int se_gemm2_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, int variant, void* stream) {
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("SE_AMD_GEMM_DBG");     // developer ablation switch (timing only, wrong results)
    dbg = e ? atoi(e) : 0;
  }
  static int group_m = 0;
  if (group_m == 0) {
    const char* gm = getenv("SE_AMD_GEMM_GROUPM2");
    group_m = gm ? atoi(gm) : 1;
    if (group_m < 1) group_m = 1;
  }
  GArgs g{A, lda, W, ldw, bias, residual_f32, M, N, K, act, out_bf16, out_f32, ldc, vec_ok, dbg, se::as_stream(stream), group_m, 1, 0};
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, g.st);
  if (variant == 4) return launch_cfg<2, 2, 0>(g);
  if (variant == 3 && K >= 2 * se::k2BK) return launch_cfg<4, 3, 1>(g);
  return launch_cfg<4, 3, 0>(g);
}

// split-K launcher for the weight-gradient GEMM (bwd.hip): C_s[M,N] = A[:, s*Kc:(s+1)*Kc] . W[:, s*Kc:(s+1)*Kc]^T, fp32 slabs
extern "C" int se_gemm2_splitk_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int Kc, int splits,
                                      float* partials, void* stream) {
  GArgs g{A, lda, W, ldw, nullptr, nullptr, M, N, Kc, SE_ACT_IDENTITY, nullptr, partials, N, (N % 4 == 0) ? 1 : 0, 0,
          se::as_stream(stream), 1, splits, (size_t)M * N};
  if (M <= 8192) return launch_cfg<2, 2, 0>(g);      // 128 x 128 tiles, two workgroups per CU: the small-batch (serving) path
  if (Kc >= 2 * se::k2BK) return launch_cfg<4, 3, 1>(g);
  return launch_cfg<4, 3, 0>(g);
}
