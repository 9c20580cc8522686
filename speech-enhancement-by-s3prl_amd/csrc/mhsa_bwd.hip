// mhsa_bwd.hip -- backward of the flash MHSA core (row E2 through B2): given dO = d(ctx) it produces d(qkv) in the
// fused (B*T, 3H) layout the QKV projection's weight / input gradients consume.  Nothing (T x T) is materialised: the
// probabilities are recomputed per tile from the log-sum-exp the training forward stored (se_mhsa_fwd_lse_bf16),
//     P = exp2(c S - lse),  D = rowsum(dO . O),  dS = P (dP - D) / 8,   with c = log2(e) / 8, S = Q K^T, dP = dO V^T
//     dV = P^T dO      dK = dS^T Q      dQ = dS K
// Two kernels, both built like the forward (4 waves, 32 rows of the stationary operand per wave, 64-row tiles of
// the streamed operands double-buffered in LDS with the same swizzle, v_mfma_f32_32x32x16_bf16):
//   dq kernel   query-stationary.  S^T = K Q^T and dP^T = V dO^T put one query on each lane, so lse / D are lane
//               scalars and the dS^T accumulator registers are, as in the forward, directly the B operand of
//               dQ^T += K^T dS^T (K^T fragments through ds_read_b64_tr_b16).  Also computes D and stores it.
//   dkv kernel  key-stationary.  S = Q K^T and dP = dO V^T put one key on each lane; P and dS registers are the B
//               operands of dV^T += dO^T P and dK^T += Q^T dS (dO^T / Q^T through tr_b16 reads of the streamed tiles).
// Recomputing S twice (7 tile products against the forward's 2) keeps both kernels free of atomics and the result
// deterministic.  Keys >= lengths[b] carry P = 0 (their dK / dV rows are written as zeros), as the forward excludes them.
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"
#include "dropout.h"

namespace se {

static constexpr f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define SE_TR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ptr))

// XCD-aware work mapping shared with the forward: all tiles of one (utterance, head) on ONE XCD (they re-read the same
// streamed operands from its L2), consecutive in its dispatch order.
__device__ __forceinline__ void pair_mapping(int& b, int& head, int& tile) {
  const int nt = gridDim.x, pairs = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + nt * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((pairs & 7) == 0) {
    const int xcd = lin & 7, i = lin >> 3;
    const int pair = 8 * (i / nt) + xcd;
    tile = i % nt;
    head = pair % gridDim.y;
    b = pair / gridDim.y;
  } else {
    tile = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// dQ (and D): workgroup = 128 queries of one (utterance, head); streams (K, V) tiles of 64 keys.
// DROP: the forward dropped the probabilities (mhsa.hip); with mask m in {0, 1/(1-p)}: dV = (P m)^T dO, dS = P (m dP - D)
template <int OCC, int DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mhsa_bwd_dq_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ o, const uint16_t* __restrict__ d_o, const float* __restrict__ lse,
    const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ dqkv, float* __restrict__ dvec, uint32_t dkey, uint32_t thr16,
    float dscale) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * kAK * kHD * 2];   // 2 buffers x (K, V) x 8 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int b, head, qt;
  pair_mapping(b, head, qt);
  const int heads = H / kHD;
  const int q0 = qt * kAQ + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  // ---- stationary fragments: lane -> query q0 + l31, d = 16 s + 8 hh .. +7
  bf16x8 qf[4], dof[4];
  float dsum = 0.f;
  const bool qvalid = q0 + l31 < T;
  const int qc = min(q0 + l31, T - 1);
  {
    const uint16_t* qp = base + (size_t)qc * ld + 8 * hh;
    const size_t orow = ((size_t)b * T + qc) * H + head * kHD + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      dof[s] = *reinterpret_cast<const bf16x8*>(d_o + orow + 16 * s);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + orow + 16 * s);
#pragma unroll
      for (int j = 0; j < 8; ++j) dsum = fmaf((float)dof[s][j], (float)of[j], dsum);
    }
  }
  dsum += __shfl_xor(dsum, 32);
  const size_t stat = ((size_t)b * heads + head) * T + qc;
  const float L2 = qvalid ? lse[stat] : INFINITY;       // exp2(x - inf) = 0: rows past T contribute nothing
  if (!qvalid) dsum = 0.f;
  if (qvalid && hh == 0) dvec[stat] = dsum;
  const uint32_t drop_row = (uint32_t)stat * (uint32_t)((T + 1) >> 1);      // pair index base of this lane's query row
  // DROP == 2: the mask as the query-major bit matrix of dropmask.hip (its address in dkey / thr16, as in the forward); one 8-byte load per lane
  // and tile, fetched one tile ahead; the 1 / (1 - p) of the kept probabilities is a factor of the fma below
  const uint32_t* mrow = nullptr;
  uint2 mw_nxt = make_uint2(0u, 0u);
  if (DROP == 2) {
    const uint32_t* dmask = reinterpret_cast<const uint32_t*>(((uint64_t)thr16 << 32) | (uint64_t)dkey);
    mrow = dmask + stat * (size_t)(4 * ((T + 127) >> 7));
    mw_nxt = *reinterpret_cast<const uint2*>(mrow);
  }

  // ---- staging of the (K, V) tiles: 64 rows x 128 B each; 256 threads x 16 B = 32 rows per pass
  const int srow = tid >> 3, sch = tid & 7;
  const uint16_t* kp = base + H + sch * 8;
  const uint16_t* vp = base + 2 * H + sch * 8;
  uint4 rk0, rk1, rv0, rv1;
  const int so0 = kv_off(srow, sch), so1 = kv_off(srow + 32, sch);
#define SE_Q_ISSUE(kt)                                                                       \
  do {                                                                                       \
    const size_t r0 = (size_t)min((kt) * kAK + srow, T - 1) * ld;                            \
    const size_t r1 = (size_t)min((kt) * kAK + srow + 32, T - 1) * ld;                       \
    rk0 = *reinterpret_cast<const uint4*>(kp + r0);                                          \
    rk1 = *reinterpret_cast<const uint4*>(kp + r1);                                          \
    rv0 = *reinterpret_cast<const uint4*>(vp + r0);                                          \
    rv1 = *reinterpret_cast<const uint4*>(vp + r1);                                          \
  } while (0)
#define SE_Q_WRITE(buf)                                                 \
  do {                                                                  \
    char* k_w = smem + (buf) * 16384;                                   \
    char* v_w = k_w + 8192;                                             \
    *reinterpret_cast<uint4*>(k_w + so0) = rk0;                         \
    *reinterpret_cast<uint4*>(k_w + so1) = rk1;                         \
    *reinterpret_cast<uint4*>(v_w + so0) = rv0;                         \
    *reinterpret_cast<uint4*>(v_w + so1) = rv1;                         \
  } while (0)

  f32x16 dq0, dq1;                    // dQ^T d-blocks 0 / 1: col = query (lane & 31), row = d
#pragma unroll
  for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }
  const float c = 0.125f * 1.44269504088896340736f;

  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = kv_off(l31, 2 * s + hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int toff[2][2];                     // transposed-read offsets inside a tile: [dblk][lo / hi]
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    toff[dblk][0] = kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    toff[dblk][1] = kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }

  SE_Q_ISSUE(0);
  SE_Q_WRITE(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) SE_Q_ISSUE(kt + 1);
    const uint2 mw_cur = mw_nxt;
    if (DROP == 2 && kt + 1 < nkt) mw_nxt = *reinterpret_cast<const uint2*>(mrow + 2 * (kt + 1));
    const char* t_s = smem + cur * 16384;
    f32x16 s0, s1, p0, p1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(t_s + koff[s]);
      const bf16x8 kb_ = *reinterpret_cast<const bf16x8*>(t_s + koff[s] + 4096);
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(t_s + 8192 + koff[s]);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(t_s + 8192 + koff[s] + 4096);
      // the first product of each chain takes the inline constant 0 as C (no v_mov per register to clear the accumulators)
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s == 0 ? kZero16 : s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_, qf[s], s == 0 ? kZero16 : s1, 0, 0, 0);
      p0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[s], s == 0 ? kZero16 : p0, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, dof[s], s == 0 ? kZero16 : p1, 0, 0, 0);
    }
    if ((kt + 1) * kAK > len) {
      const int kbase = kt * kAK + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        if (key >= len) s0[r] = -INFINITY;
        if (key + 32 >= len) s1[r] = -INFINITY;
      }
    }
    // dS^T = P^T (m dP^T - D)   (the 1/8 of the score scale is applied once, in the epilogue)
    if (DROP == 2) {
      const uint32_t we_ = mw_cur.x >> (2 * hh), wo_ = mw_cur.y >> (2 * hh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // key 4 hh + (r & 3) + 8 (r >> 2) (+ 32): pair 2 hh + ((r & 3) >> 1) + 4 (r >> 2) (+ 16) of the tile, parity r & 1 (mhsa.hip)
        const int pos = ((r & 3) >> 1) + 4 * (r >> 2);
        const uint32_t w_ = (r & 1) ? wo_ : we_;
        p0[r] = __uint_as_float(__float_as_uint(p0[r]) & (uint32_t)((int32_t)(w_ << (31 - pos)) >> 31));
        p1[r] = __uint_as_float(__float_as_uint(p1[r]) & (uint32_t)((int32_t)(w_ << (15 - pos)) >> 31));
      }
    } else if (DROP) {
      const uint32_t pb = drop_row + (uint32_t)((kt * kAK + 4 * hh) >> 1);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t off = (uint32_t)(((r & 3) + 8 * (r >> 2)) >> 1);
        const uint32_t b0 = dropout_bits(dkey, pb + off), b1 = dropout_bits(dkey, pb + off + 16);
        p0[r] *= dropout_mul(b0, 0, thr16, dscale); p0[r + 1] *= dropout_mul(b0, 1, thr16, dscale);
        p1[r] *= dropout_mul(b1, 0, thr16, dscale); p1[r + 1] *= dropout_mul(b1, 1, thr16, dscale);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e0 = __builtin_amdgcn_exp2f(fmaf(s0[r], c, -L2));
      const float e1 = __builtin_amdgcn_exp2f(fmaf(s1[r], c, -L2));
      s0[r] = e0 * (DROP == 2 ? fmaf(p0[r], dscale, -dsum) : p0[r] - dsum);
      s1[r] = e1 * (DROP == 2 ? fmaf(p1[r], dscale, -dsum) : p1[r] - dsum);
    }
    bf16x8 df[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        df[0][s][j] = (__bf16)s0[8 * s + j];
        df[1][s][j] = (__bf16)s1[8 * s + j];
      }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const bf16x4 lo = SE_TR(t_s + toff[dblk][0] + kb * 4096 + s * 2048);
          const bf16x4 hi = SE_TR(t_s + toff[dblk][1] + kb * 4096 + s * 2048);
          const bf16x8 kt_ = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (dblk == 0) dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_, df[kb][s], dq0, 0, 0, 0);
          else dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_, df[kb][s], dq1, 0, 0, 0);
        }
      }
    if (kt + 1 < nkt) SE_Q_WRITE(cur ^ 1);
    __syncthreads();
  }
#undef SE_Q_ISSUE
#undef SE_Q_WRITE

  // ---- epilogue: dQ = dQ^T / 8 ; lane holds query q0 + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh
  if (qvalid) {
    uint16_t* op = dqkv + ((size_t)b * T + q0 + l31) * ld + head * kHD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w0 = make_uint2(pack_bf16x2(dq0[4 * g] * 0.125f, dq0[4 * g + 1] * 0.125f), pack_bf16x2(dq0[4 * g + 2] * 0.125f, dq0[4 * g + 3] * 0.125f));
      uint2 w1 = make_uint2(pack_bf16x2(dq1[4 * g] * 0.125f, dq1[4 * g + 1] * 0.125f), pack_bf16x2(dq1[4 * g + 2] * 0.125f, dq1[4 * g + 3] * 0.125f));
      *reinterpret_cast<uint2*>(op + 8 * g) = w0;
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 128 keys of one (utterance, head); streams (Q, dO) tiles of 64 queries (+ their lse / D).
constexpr int kBufKV = 16384 + 512;   // Q tile, dO tile, lse[64], D[64]

template <int OCC, int DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mhsa_bwd_dkv_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ d_o, const float* __restrict__ lse, const float* __restrict__ dvec,
    const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ dqkv, uint32_t dkey, uint32_t thr16, float dscale) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kBufKV];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int b, head, ktile;
  pair_mapping(b, head, ktile);
  const int heads = H / kHD;
  const int k0 = ktile * kAQ + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;
  const int key = k0 + l31;
  uint16_t* okp = dqkv + ((size_t)b * T + min(key, T - 1)) * ld + H + head * kHD + 4 * hh;

  if (ktile * kAQ >= len) {
    // the whole workgroup owns padded keys only: dK = dV = 0 (uniform branch, before any barrier)
    if (key < T) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<uint2*>(okp + 8 * g) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(okp + 32 + 8 * g) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(okp + H + 8 * g) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(okp + H + 32 + 8 * g) = make_uint2(0u, 0u);
      }
    }
    return;
  }

  // ---- stationary fragments (B operands): lane -> key k0 + l31, d = 16 s + 8 hh .. +7
  bf16x8 kf[4], vf[4];
  {
    const uint16_t* kp = base + (size_t)min(key, T - 1) * ld + H + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
      vf[s] = *reinterpret_cast<const bf16x8*>(kp + H + 16 * s);
    }
  }

  // ---- staging of the (Q, dO) tiles + per-row statistics
  const int nqt = (T + kAK - 1) / kAK;
  const int srow = tid >> 3, sch = tid & 7;
  const uint16_t* qp = base + sch * 8;
  const uint16_t* dop = d_o + (size_t)b * T * H + head * kHD + sch * 8;
  const float* lsep = lse + ((size_t)b * heads + head) * T;
  const float* dvp = dvec + ((size_t)b * heads + head) * T;
  uint4 rq0, rq1, rd0, rd1;
  float rstat = 0.f;
  const int so0 = kv_off(srow, sch), so1 = kv_off(srow + 32, sch);
#define SE_K_ISSUE(qt)                                                                       \
  do {                                                                                       \
    const int a0 = min((qt) * kAK + srow, T - 1), a1 = min((qt) * kAK + srow + 32, T - 1);    \
    rq0 = *reinterpret_cast<const uint4*>(qp + (size_t)a0 * ld);                             \
    rq1 = *reinterpret_cast<const uint4*>(qp + (size_t)a1 * ld);                             \
    rd0 = *reinterpret_cast<const uint4*>(dop + (size_t)a0 * H);                             \
    rd1 = *reinterpret_cast<const uint4*>(dop + (size_t)a1 * H);                             \
    if (tid < 128) {                                                                         \
      const int qq = (qt) * kAK + (tid & 63);                                                \
      rstat = tid < 64 ? (qq < T ? lsep[qq] : INFINITY) : (qq < T ? dvp[qq] : 0.f);          \
    }                                                                                        \
  } while (0)
#define SE_K_WRITE(buf)                                                 \
  do {                                                                  \
    char* q_w = smem + (buf) * kBufKV;                                  \
    char* d_w = q_w + 8192;                                             \
    *reinterpret_cast<uint4*>(q_w + so0) = rq0;                         \
    *reinterpret_cast<uint4*>(q_w + so1) = rq1;                         \
    *reinterpret_cast<uint4*>(d_w + so0) = rd0;                         \
    *reinterpret_cast<uint4*>(d_w + so1) = rd1;                         \
    if (tid < 128) *reinterpret_cast<float*>(q_w + 16384 + tid * 4) = rstat; \
  } while (0)

  f32x16 dk0, dk1, dv0, dv1;          // dK^T / dV^T d-blocks: col = key (lane & 31), row = d
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }
  const float c = 0.125f * 1.44269504088896340736f;

  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = kv_off(l31, 2 * s + hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int toff[2][2];
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    toff[dblk][0] = kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    toff[dblk][1] = kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }

  // dropout indexing (see the DROP branch below)
  const uint32_t drop_row0 = (uint32_t)(b * heads + head) * (uint32_t)T, drop_ppr = (uint32_t)((T + 1) >> 1);
  const uint32_t drop_k2 = (uint32_t)(min(key, T - 1) >> 1);
  const int drop_h = key & 1;
  // DROP == 2: the key-major bit matrix of dropmask.hip (row = this lane's key, bit j of word w = query 32 w + j); one 8-byte load per 64-query
  // tile; the 1 / (1 - p) factor of P m goes into the dV epilogue and into the fma of dS
  const uint32_t* mrow = nullptr;
  uint2 mw_nxt = make_uint2(0u, 0u);
  if (DROP == 2) {
    const uint32_t* dmask = reinterpret_cast<const uint32_t*>(((uint64_t)thr16 << 32) | (uint64_t)dkey);
    mrow = dmask + ((size_t)(b * heads + head) * (size_t)(128 * ((T + 127) >> 7)) + key) * (size_t)(8 * ((T + 255) >> 8));
    mw_nxt = *reinterpret_cast<const uint2*>(mrow);
  }

  SE_K_ISSUE(0);
  SE_K_WRITE(0);
  __syncthreads();

  for (int qt = 0; qt < nqt; ++qt) {
    const int cur = qt & 1;
    if (qt + 1 < nqt) SE_K_ISSUE(qt + 1);
    const uint2 mw_cur = mw_nxt;
    if (DROP == 2 && qt + 1 < nqt) mw_nxt = *reinterpret_cast<const uint2*>(mrow + 2 * (qt + 1));
    const char* t_s = smem + cur * kBufKV;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      // S = Q K^T, dP = dO V^T for 32 queries x this wave's 32 keys: lane = key, register r -> query
      // 32 qb + (r&3) + 8 (r>>2) + 4 hh
      f32x16 sa, da;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(t_s + koff[s] + qb * 4096);
        const bf16x8 oa = *reinterpret_cast<const bf16x8*>(t_s + 8192 + koff[s] + qb * 4096);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], s == 0 ? kZero16 : sa, 0, 0, 0);
        da = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[s], s == 0 ? kZero16 : da, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l2 = *reinterpret_cast<const float4*>(t_s + 16384 + (32 * qb + 4 * hh + 8 * g) * 4);
        const float4 dd = *reinterpret_cast<const float4*>(t_s + 16384 + 256 + (32 * qb + 4 * hh + 8 * g) * 4);
        const float e0 = __builtin_amdgcn_exp2f(fmaf(sa[4 * g + 0], c, -l2.x));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(sa[4 * g + 1], c, -l2.y));
        const float e2 = __builtin_amdgcn_exp2f(fmaf(sa[4 * g + 2], c, -l2.z));
        const float e3 = __builtin_amdgcn_exp2f(fmaf(sa[4 * g + 3], c, -l2.w));
        if (DROP == 2) {
          // query 32 qb + 4 hh + 8 g + j of the tile: word qb, bit 4 hh + 8 g + j.  All-ones / zero masks (v_bfe_i32) applied to the bit patterns:
          // P m is accumulated unscaled (the dV epilogue multiplies by 1 / (1 - p)), m dP takes the scale inside the fma
          const uint32_t w_ = (qb ? mw_cur.y : mw_cur.x) >> (4 * hh);
          const uint32_t k0 = (uint32_t)((int32_t)(w_ << (31 - 8 * g)) >> 31), k1 = (uint32_t)((int32_t)(w_ << (30 - 8 * g)) >> 31);
          const uint32_t k2 = (uint32_t)((int32_t)(w_ << (29 - 8 * g)) >> 31), k3 = (uint32_t)((int32_t)(w_ << (28 - 8 * g)) >> 31);
          sa[4 * g + 0] = __uint_as_float(__float_as_uint(e0) & k0); sa[4 * g + 1] = __uint_as_float(__float_as_uint(e1) & k1);
          sa[4 * g + 2] = __uint_as_float(__float_as_uint(e2) & k2); sa[4 * g + 3] = __uint_as_float(__float_as_uint(e3) & k3);
          da[4 * g + 0] = e0 * fmaf(__uint_as_float(__float_as_uint(da[4 * g + 0]) & k0), dscale, -dd.x);
          da[4 * g + 1] = e1 * fmaf(__uint_as_float(__float_as_uint(da[4 * g + 1]) & k1), dscale, -dd.y);
          da[4 * g + 2] = e2 * fmaf(__uint_as_float(__float_as_uint(da[4 * g + 2]) & k2), dscale, -dd.z);
          da[4 * g + 3] = e3 * fmaf(__uint_as_float(__float_as_uint(da[4 * g + 3]) & k3), dscale, -dd.w);
        } else {
        float m0 = 1.f, m1 = 1.f, m2 = 1.f, m3 = 1.f;
        if (DROP) {
          // lane = key, registers walk the queries: one hash per element (its pair partner is the neighbouring key, i.e. the
          // neighbouring lane); pair = row_id(query) * ceil(T / 2) + key / 2, 16-bit half by the key's parity
          const uint32_t q_first = (uint32_t)(qt * kAK + 32 * qb + 4 * hh + 8 * g);
          const uint32_t pr = (drop_row0 + q_first) * drop_ppr + drop_k2;
          m0 = dropout_mul(dropout_bits(dkey, pr), drop_h, thr16, dscale);
          m1 = dropout_mul(dropout_bits(dkey, pr + drop_ppr), drop_h, thr16, dscale);
          m2 = dropout_mul(dropout_bits(dkey, pr + 2 * drop_ppr), drop_h, thr16, dscale);
          m3 = dropout_mul(dropout_bits(dkey, pr + 3 * drop_ppr), drop_h, thr16, dscale);
        }
        sa[4 * g + 0] = e0 * m0; sa[4 * g + 1] = e1 * m1; sa[4 * g + 2] = e2 * m2; sa[4 * g + 3] = e3 * m3;
        da[4 * g + 0] = e0 * (m0 * da[4 * g + 0] - dd.x);
        da[4 * g + 1] = e1 * (m1 * da[4 * g + 1] - dd.y);
        da[4 * g + 2] = e2 * (m2 * da[4 * g + 2] - dd.z);
        da[4 * g + 3] = e3 * (m3 * da[4 * g + 3] - dd.w);
        }
      }
      bf16x8 pf[2], df[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[s][j] = (__bf16)sa[8 * s + j];
          df[s][j] = (__bf16)da[8 * s + j];
        }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const int off_lo = toff[dblk][0] + qb * 4096 + s * 2048, off_hi = toff[dblk][1] + qb * 4096 + s * 2048;
          const bf16x4 olo = SE_TR(t_s + 8192 + off_lo);
          const bf16x4 ohi = SE_TR(t_s + 8192 + off_hi);
          const bf16x4 qlo = SE_TR(t_s + off_lo);
          const bf16x4 qhi = SE_TR(t_s + off_hi);
          const bf16x8 ot = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
          const bf16x8 qt_ = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
          if (dblk == 0) {
            dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, pf[s], dv0, 0, 0, 0);
            dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt_, df[s], dk0, 0, 0, 0);
          } else {
            dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ot, pf[s], dv1, 0, 0, 0);
            dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt_, df[s], dk1, 0, 0, 0);
          }
        }
      }
    }
    if (qt + 1 < nqt) SE_K_WRITE(cur ^ 1);
    __syncthreads();
  }
#undef SE_K_ISSUE
#undef SE_K_WRITE

  // ---- epilogue: lane holds key k0 + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh ; padded keys get zeros
  if (key < T) {
    const float sk = key < len ? 0.125f : 0.f, sv = key < len ? (DROP == 2 ? dscale : 1.f) : 0.f;
    const bool live = key < len;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 a0 = make_uint2(pack_bf16x2(dk0[4 * g] * sk, dk0[4 * g + 1] * sk), pack_bf16x2(dk0[4 * g + 2] * sk, dk0[4 * g + 3] * sk));
      uint2 a1 = make_uint2(pack_bf16x2(dk1[4 * g] * sk, dk1[4 * g + 1] * sk), pack_bf16x2(dk1[4 * g + 2] * sk, dk1[4 * g + 3] * sk));
      uint2 b0 = make_uint2(pack_bf16x2(dv0[4 * g] * sv, dv0[4 * g + 1] * sv), pack_bf16x2(dv0[4 * g + 2] * sv, dv0[4 * g + 3] * sv));
      uint2 b1 = make_uint2(pack_bf16x2(dv1[4 * g] * sv, dv1[4 * g + 1] * sv), pack_bf16x2(dv1[4 * g + 2] * sv, dv1[4 * g + 3] * sv));
      if (!live) { a0 = a1 = b0 = b1 = make_uint2(0u, 0u); }      // select, not multiply: a padded key's lane may hold inf / nan
      *reinterpret_cast<uint2*>(okp + 8 * g) = a0;
      *reinterpret_cast<uint2*>(okp + 32 + 8 * g) = a1;
      *reinterpret_cast<uint2*>(okp + H + 8 * g) = b0;
      *reinterpret_cast<uint2*>(okp + H + 32 + 8 * g) = b1;
    }
  }
}

}  // namespace se

extern "C" int se_mhsa_bwd_bf16(const uint16_t* qkv, const uint16_t* ctx, const uint16_t* d_ctx, const float* lse, const int32_t* lengths,
                                int B, int T, int heads, uint16_t* dqkv, float* dvec, float dropout_p, uint64_t seed, uint32_t site,
                                void* stream) {
  SE_REQUIRE(qkv && ctx && d_ctx && lse && dqkv && dvec, "se_mhsa_bwd_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_bwd_bf16: bad shape B=%d T=%d heads=%d", B, T, heads);
  const int H = heads * se::kHD;
  hipStream_t st = se::as_stream(stream);
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  se::ProfScope prof(se::kProfMhsaBwd, 14.0 * B * (double)heads * T * (double)T * se::kHD, st);
  const se::DropoutCfg d = se::make_dropout(dropout_p, seed);
  if (d.thr16) {
    SE_REQUIRE((double)B * heads * T * ((T + 1) / 2) < 4294967296.0, "se_mhsa_bwd_bf16: dropout pair index exceeds 32 bits");
    const uint32_t dkey = se::dropout_key(seed, site);
    hipLaunchKernelGGL((se::mhsa_bwd_dq_kernel<2, 1>), grid, dim3(256), 0, st, qkv, ctx, d_ctx, lse, lengths, T, H, dqkv, dvec, dkey, d.thr16, d.scale);
    SE_LAUNCH_CHECK();
    hipLaunchKernelGGL((se::mhsa_bwd_dkv_kernel<2, 1>), grid, dim3(256), 0, st, qkv, d_ctx, lse, dvec, lengths, T, H, dqkv, dkey, d.thr16, d.scale);
    SE_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL((se::mhsa_bwd_dq_kernel<2, 0>), grid, dim3(256), 0, st, qkv, ctx, d_ctx, lse, lengths, T, H, dqkv, dvec, 0u, 0u, 1.f);
    SE_LAUNCH_CHECK();
    hipLaunchKernelGGL((se::mhsa_bwd_dkv_kernel<2, 0>), grid, dim3(256), 0, st, qkv, d_ctx, lse, dvec, lengths, T, H, dqkv, 0u, 0u, 1.f);
    SE_LAUNCH_CHECK();
  }
  return SE_OK;
}

#ifdef SE_AMD_EXPERIMENTS
// Backward with the dropout mask of dropmask.hip (se_mhsa_dropmask) instead of the in-kernel hash: mask_r query-major (dQ kernel), mask_c key-major
// (dK / dV kernel).  Same mask and the same gradients up to where the 1 / (1 - p) factor is applied.
extern "C" int se_mhsa_bwd_masked_bf16(const uint16_t* qkv, const uint16_t* ctx, const uint16_t* d_ctx, const float* lse, const int32_t* lengths,
                                       int B, int T, int heads, uint16_t* dqkv, float* dvec, const uint32_t* mask_r, const uint32_t* mask_c,
                                       float dropout_p, void* stream) {
  SE_REQUIRE(qkv && ctx && d_ctx && lse && dqkv && dvec && mask_r && mask_c, "se_mhsa_bwd_masked_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_bwd_masked_bf16: bad shape B=%d T=%d heads=%d", B, T, heads);
  SE_REQUIRE((((uintptr_t)mask_r | (uintptr_t)mask_c) % 16) == 0, "se_mhsa_bwd_masked_bf16: masks must be 16-B aligned");
  const se::DropoutCfg d = se::make_dropout(dropout_p, 0);
  SE_REQUIRE(d.thr16 != 0, "se_mhsa_bwd_masked_bf16: dropout_p must be > 0");
  const int H = heads * se::kHD;
  hipStream_t st = se::as_stream(stream);
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  se::ProfScope prof(se::kProfMhsaBwd, 14.0 * B * (double)heads * T * (double)T * se::kHD, st);
  const uintptr_t pr = (uintptr_t)mask_r, pc = (uintptr_t)mask_c;
  hipLaunchKernelGGL((se::mhsa_bwd_dq_kernel<2, 2>), grid, dim3(256), 0, st, qkv, ctx, d_ctx, lse, lengths, T, H, dqkv, dvec, (uint32_t)(pr & 0xffffffffu),
                     (uint32_t)(pr >> 32), d.scale);
  SE_LAUNCH_CHECK();
  hipLaunchKernelGGL((se::mhsa_bwd_dkv_kernel<2, 2>), grid, dim3(256), 0, st, qkv, d_ctx, lse, dvec, lengths, T, H, dqkv, (uint32_t)(pc & 0xffffffffu),
                     (uint32_t)(pc >> 32), d.scale);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
#endif
