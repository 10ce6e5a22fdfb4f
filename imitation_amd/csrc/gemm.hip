// fp32 MFMA GEMM for gfx950: the contraction engine behind every dense layer of the
// discriminator / reward nets (forward, input-gradient and split-K weight-gradient forms).
//
//   NT : C[M,N] = act(A[M,K] . B[N,K]^T + bias)          -- Linear forward   (networks.py:262-277)
//   NN : C[M,N] = (A[M,K] . B[K,N]) * act'(P[M,N])        -- grad wrt layer input
//   TN : C_s[M,N] = A[Ks,M]^T . B[Ks,N]  per K-split s     -- grad wrt weights (+ column sums of A)
//
// One wave owns TM x TN tiles of v_mfma_f32_32x32x2_f32 (lane l feeds A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]); operands are staged through LDS in whichever of two layouts makes the
// fragment read bank-conflict free for ds_read_b32 (32 consecutive lanes -> 32 distinct banks):
//   "MK": [row][BK+1]  (odd stride)  for operands that are k-contiguous in memory,
//   "KM": [k][rows]                  for operands that are row-contiguous in memory.
// Global->LDS goes through registers (float4 loads issued one chunk ahead, written to the
// other LDS buffer after the MFMAs: one barrier per 32-deep K chunk).
#include "common.h"
#include "disc_reduce.h"

namespace {

constexpr int BK = 32;
constexpr int LDP = BK + 1;

// Register staging uses a first-class 4-wide vector (not the float4 struct): aggregate copies of
// struct elements into LDS keep the staging arrays in scratch memory.
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 load4_guard(const float* __restrict__ p, int n_valid, bool vec_ok) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (n_valid >= 4 && vec_ok) {
    v = *reinterpret_cast<const f4*>(p);
  } else if (n_valid > 0) {
    v.x = p[0];
    if (n_valid > 1) v.y = p[1];
    if (n_valid > 2) v.z = p[2];
    if (n_valid > 3) v.w = p[3];
  }
  return v;
}

__device__ __forceinline__ int im_div(int n, int d, unsigned long long magic) {
  return d == 1 ? n : (int)__umul64hi((unsigned long long)(unsigned)n, magic);
}
__device__ __forceinline__ int im_rowoff(const IaIm& im, int m) {
  const int b = im_div(m, im.OHW, im.mOHW), p = m - b * im.OHW;
  const int oh = im_div(p, im.OW, im.mOW), ow = p - oh * im.OW;
  return b * im.HWC + (oh * im.S * im.W + ow * im.S) * im.C;
}
__device__ __forceinline__ int im_kmap(const IaIm& im, int k) {
  const int i = im_div(k, im.seg, im.mseg);
  return i * im.rstride + (k - i * im.seg);
}

// Padded view: element offset of (row m, column k) with the tap clamped into the image, and whether it is inside
__device__ __forceinline__ int im_addr_pad(const IaIm& im, int m, int k, bool& valid) {
  const int b = im_div(m, im.OHW, im.mOHW), p = m - b * im.OHW;
  const int oh = im_div(p, im.OW, im.mOW), ow = p - oh * im.OW;
  const int i = im_div(k, im.seg, im.mseg), r = k - i * im.seg;
  const int j = im_div(r, im.C, im.mC), c = r - j * im.C;
  const int y = oh * im.S - im.pad + i, x = ow * im.S - im.pad + j;
  valid = (unsigned)y < (unsigned)im.H && (unsigned)x < (unsigned)im.W;
  const int yc = min(max(y, 0), im.H - 1), xc = min(max(x, 0), im.W - 1);
  return b * im.HWC + (yc * im.W + xc) * im.C + c;
}
__device__ __forceinline__ f4 im_load_pad(const float* __restrict__ src, const IaIm& im, int m, int k) {
  bool valid;
  const int off = im_addr_pad(im, m, k, valid);
  const f4 v = *reinterpret_cast<const f4*>(src + off);   // (unconditional, clamped address; masked below)
  const f4 z = {0.f, 0.f, 0.f, 0.f};
  return valid ? v : z;
}
__device__ __forceinline__ long long im_crow(const IaIm& im, int m, int py, int px) {
  if (!im.cm_on) return m;
  const int b = im_div(m, im.cm_OHW, im.cm_mOHW), p = m - b * im.cm_OHW;
  const int yy = im_div(p, im.cm_OW, im.cm_mOW), xx = p - yy * im.cm_OW;
  return (long long)b * im.cm_HW + (yy * im.cm_S + py) * im.cm_W + xx * im.cm_S + px;
}

// Tile loaders. ROWS = tile extent in the non-reduction index.
template <int ROWS, int NT, bool KM>
struct TileIO {
  static constexpr int NV = ROWS * (BK / 4) / NT;  // float4 per thread
  static_assert(ROWS * (BK / 4) % NT == 0, "tile not divisible among threads");

  // src is [rows_total, K] (k contiguous) when !KM, or [K, rows_total] (row contiguous) when KM.
  __device__ static void load(f4 (&r)[NV], const float* __restrict__ src, int ld, int row0, int rows_total,
                              int k0, int k_end, bool vec_ok, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if (!KM) {
        const int rr = f >> 3, kq = f & 7;
        const int gr = row0 + rr, gk = k0 + kq * 4;
        const int nv = (gr < rows_total) ? (k_end - gk) : 0;
        r[i] = load4_guard(src + (long long)gr * ld + gk, nv, vec_ok);
      } else {
        constexpr int QPR = ROWS / 4;
        const int kr = f / QPR, mq = f % QPR;
        const int gk = k0 + kr, gm = row0 + mq * 4;
        const int nv = (gk < k_end) ? (rows_total - gm) : 0;
        r[i] = load4_guard(src + (long long)gk * ld + gm, nv, vec_ok);
      }
    }
  }
  // Interior tiles (whole tile in range, K range a multiple of BK, 16-byte aligned rows): plain
  // float4 loads in straight-line code, so the compiler can keep them in flight behind counted
  // `s_waitcnt vmcnt(N)` instead of draining the queue around every guarded load.
  __device__ __forceinline__ static const float* fast_base(const float* __restrict__ src, int ld, int row0, int tid) {
    if (!KM) return src + (long long)(row0 + (tid >> 3)) * ld + (tid & 7) * 4;
    constexpr int QPR = ROWS / 4;
    return src + (long long)(tid / QPR) * ld + row0 + (tid % QPR) * 4;
  }
  __device__ __forceinline__ static void load_fast(f4 (&r)[NV], const float* __restrict__ base, int ld, int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (!KM) {
        r[i] = *reinterpret_cast<const f4*>(base + (long long)(i * (NT / 8)) * ld + k0);
      } else {
        constexpr int QPR = ROWS / 4;
        static_assert(NT % QPR == 0, "thread count must be a multiple of quads per row");
        r[i] = *reinterpret_cast<const f4*>(base + (long long)(k0 + i * (NT / QPR)) * ld);
      }
    }
  }
  // ---- the same two loaders for an operand given as an implicit im2col view (IaIm): element (row, k) of the
  // [rows_total, K] matrix lives at src + im_rowoff(row) + im_kmap(k); quads of four consecutive k never straddle a
  // kernel row (seg % 4 == 0). !KM: rows are the tile rows; KM: the reduction index walks the rows of the view.
  __device__ static void load_im(f4 (&r)[NV], const float* __restrict__ src, const IaIm& im, int row0, int rows_total,
                                 int k0, int k_end, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (!KM) {
        const int rr = f >> 3, kq = f & 7;
        const int gr = row0 + rr, gk = k0 + kq * 4;
        if (gr < rows_total && gk + 4 <= k_end)
          v = im.pad ? im_load_pad(src, im, gr, gk) : *reinterpret_cast<const f4*>(src + im_rowoff(im, gr) + im_kmap(im, gk));
      } else {
        constexpr int QPR = ROWS / 4;
        const int kr = f / QPR, mq = f % QPR;
        const int gk = k0 + kr, gm = row0 + mq * 4;
        if (gk < k_end && gm + 4 <= rows_total)
          v = im.pad ? im_load_pad(src, im, gk, gm) : *reinterpret_cast<const f4*>(src + im_rowoff(im, gk) + im_kmap(im, gm));
      }
      r[i] = v;
    }
  }
  // interior tiles: !KM -> `off[i]` = row offsets of this thread's NV rows (+ its k quad), fixed for the tile;
  // KM -> `off[0]` = im_kmap of this thread's column quad, the row offsets follow the reduction index
  __device__ __forceinline__ static void fast_init_im(int (&off)[NV], const IaIm& im, int row0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (!KM) off[i] = im_rowoff(im, row0 + (tid >> 3) + i * (NT / 8)) + (tid & 7) * 4;
      else off[i] = im_kmap(im, row0 + (tid % (ROWS / 4)) * 4);
    }
  }
  __device__ __forceinline__ static void load_fast_im(f4 (&r)[NV], const float* __restrict__ src, const IaIm& im,
                                                      const int (&off)[NV], int k0, int tid) {
    if (!KM) {
      const int kb = im_kmap(im, k0);   // block-uniform: a 32-deep chunk lies inside one kernel row (seg % 32 == 0)
#pragma unroll
      for (int i = 0; i < NV; ++i) r[i] = *reinterpret_cast<const f4*>(src + off[i] + kb);
    } else {
      constexpr int QPR = ROWS / 4;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        r[i] = *reinterpret_cast<const f4*>(src + im_rowoff(im, k0 + tid / QPR + i * (NT / QPR)) + off[0]);
    }
  }
  // interior tiles of a PADDED view. !KM: the NV rows of a thread are fixed for the tile -- their image base and
  // top-left tap (y0, x0) are decoded once (`fast_init_pad`), a load then needs the chunk's kernel row (uniform) and
  // the thread's tap column. KM: every load decodes its row. Loads are unconditional at clamped addresses, then masked.
  struct PadRow { int base, y0, x0; };
  __device__ __forceinline__ static void fast_init_pad(PadRow (&pr)[NV], const IaIm& im, int row0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int m = row0 + (tid >> 3) + i * (NT / 8);
      const int b = im_div(m, im.OHW, im.mOHW), p = m - b * im.OHW;
      const int oh = im_div(p, im.OW, im.mOW), ow = p - oh * im.OW;
      pr[i].base = b * im.HWC;
      pr[i].y0 = oh * im.S - im.pad;
      pr[i].x0 = ow * im.S - im.pad;
    }
  }
  __device__ __forceinline__ static void load_fast_im_pad_rows(f4 (&r)[NV], const float* __restrict__ src, const IaIm& im,
                                                               const PadRow (&pr)[NV], int k0, int tid) {
    const int ki = im_div(k0, im.seg, im.mseg);            // kernel row of this chunk (block-uniform)
    const int rr = k0 - ki * im.seg + (tid & 7) * 4;       // position inside the kernel row: tap column * C + channel
    const int j = im_div(rr, im.C, im.mC), c = rr - j * im.C;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int y = pr[i].y0 + ki, x = pr[i].x0 + j;
      const bool valid = (unsigned)y < (unsigned)im.H && (unsigned)x < (unsigned)im.W;
      const int yc = min(max(y, 0), im.H - 1), xc = min(max(x, 0), im.W - 1);
      const f4 v = *reinterpret_cast<const f4*>(src + pr[i].base + (yc * im.W + xc) * im.C + c);
      r[i] = valid ? v : z;
    }
  }
  __device__ __forceinline__ static void load_fast_im_pad(f4 (&r)[NV], const float* __restrict__ src, const IaIm& im,
                                                          int row0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (!KM) {
        r[i] = im_load_pad(src, im, row0 + (tid >> 3) + i * (NT / 8), k0 + (tid & 7) * 4);
      } else {
        constexpr int QPR = ROWS / 4;
        r[i] = im_load_pad(src, im, k0 + tid / QPR + i * (NT / QPR), row0 + (tid % QPR) * 4);
      }
    }
  }
  __device__ static void store(const f4 (&r)[NV], float* __restrict__ S, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * NT;
      if (!KM) {
        const int rr = f >> 3, kq = f & 7;
        float* d = S + rr * LDP + kq * 4;
        d[0] = r[i].x; d[1] = r[i].y; d[2] = r[i].z; d[3] = r[i].w;
      } else {
        constexpr int QPR = ROWS / 4;
        const int kr = f / QPR, mq = f % QPR;
        *reinterpret_cast<f4*>(S + kr * ROWS + mq * 4) = r[i];
      }
    }
  }
  __device__ __forceinline__ static float frag(const float* __restrict__ S, int i0, int kk, int li, int lh) {
    return KM ? S[(kk + lh) * ROWS + i0 + li] : S[(i0 + li) * LDP + kk + lh];
  }
  static constexpr int ELEMS = KM ? BK * ROWS : ROWS * LDP;
};

// Tile coordinates of one workgroup.
struct TileCtx {
  int bm0, bn0, split, k_begin, k_end;
  bool a_vec, b_vec;
};

// K loop + epilogue of one output tile. FAST = interior tile (see TileIO::load_fast).
template <int WM, int WN, int TM, int TN, int MODE, bool FAST, bool IM>
__device__ __forceinline__ void gemm_tile(const IaGemm& g, const TileCtx& tc, float* __restrict__ smem) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  using AIO = TileIO<BM, NT, MODE == IA_GEMM_TN>;
  using BIO = TileIO<BN, NT, MODE != IA_GEMM_NT>;
  constexpr int STAGE = AIO::ELEMS + BIO::ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int bm0 = tc.bm0, bn0 = tc.bn0, k_begin = tc.k_begin, k_end = tc.k_end;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Two register stages: while chunk c is multiplied out of LDS, chunk c+1 sits in one register
  // set (to be written to the other LDS buffer after the MFMAs) and chunk c+2 is in flight into
  // the second set, so every global load has two chunk-times to land.
  f4 ra0[AIO::NV], rb0[BIO::NV], ra1[AIO::NV], rb1[BIO::NV];
  const int n_chunks = (k_end - k_begin + BK - 1) / BK;
  const bool do_db = (MODE == IA_GEMM_TN) && (g.dbias != nullptr) && (bn0 == 0);
  float dbacc = 0.f;

  constexpr bool IM_A = IM && MODE == IA_GEMM_NT, IM_B = IM && MODE == IA_GEMM_TN;
  const float* fa = AIO::fast_base(g.A, g.lda, bm0, tid);
  const float* fb = BIO::fast_base(g.B, g.ldb, bn0, tid);
  int ima[AIO::NV], imb[BIO::NV];
  typename AIO::PadRow pra[AIO::NV];
  if (FAST && IM_A) {
    if (g.im.pad) AIO::fast_init_pad(pra, g.im, bm0, tid);
    else AIO::fast_init_im(ima, g.im, bm0, tid);
  }
  if (FAST && IM_B) BIO::fast_init_im(imb, g.im, bn0, tid);
  auto gload = [&](f4 (&ra)[AIO::NV], f4 (&rb)[BIO::NV], int c) {
    const int k0 = k_begin + c * BK;
    if (FAST) {
      if (IM_A) {
        if (g.im.pad) AIO::load_fast_im_pad_rows(ra, g.A, g.im, pra, k0, tid);
        else AIO::load_fast_im(ra, g.A, g.im, ima, k0, tid);
      } else {
        AIO::load_fast(ra, fa, g.lda, k0);
      }
      if (IM_B) {
        if (g.im.pad) BIO::load_fast_im_pad(rb, g.B, g.im, bn0, k0, tid);
        else BIO::load_fast_im(rb, g.B, g.im, imb, k0, tid);
      } else {
        BIO::load_fast(rb, fb, g.ldb, k0);
      }
    } else {
      if (IM_A) AIO::load_im(ra, g.A, g.im, bm0, g.M, k0, k_end, tid);
      else AIO::load(ra, g.A, g.lda, bm0, g.M, k0, k_end, tc.a_vec, tid);
      if (IM_B) BIO::load_im(rb, g.B, g.im, bn0, g.N, k0, k_end, tid);
      else BIO::load(rb, g.B, g.ldb, bn0, g.N, k0, k_end, tc.b_vec, tid);
    }
  };
  auto lstore = [&](const f4 (&ra)[AIO::NV], const f4 (&rb)[BIO::NV], int c) {
    float* S = smem + (c & 1) * STAGE;
    AIO::store(ra, S, tid);
    BIO::store(rb, S + AIO::ELEMS, tid);
  };
  auto compute = [&](int c) {
    const float* As = smem + (c & 1) * STAGE;
    const float* Bs = As + AIO::ELEMS;
    // Fragment reads run one stage (Q k-steps) ahead of the MFMAs that consume them, pinned with
    // scheduling barriers: left alone, the scheduler sinks every ds_read next to its MFMA and each
    // k-step then waits a full LDS round trip.
    constexpr int Q = 4, NQ = BK / 2 / Q;
    float af[2][Q][TM], bf[2][Q][TN];
    auto rd = [&](int q) {
#pragma unroll
      for (int s = 0; s < Q; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[q & 1][s][i] = AIO::frag(As, (wm * TM + i) * 32, 2 * (q * Q + s), li, lh);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[q & 1][s][j] = BIO::frag(Bs, (wn * TN + j) * 32, 2 * (q * Q + s), li, lh);
      }
    };
    rd(0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q + 1 < NQ) rd(q + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < Q; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q & 1][s][i], bf[q & 1][s][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (do_db && tid < BM) {
#pragma unroll 8
      for (int kk = 0; kk < BK; ++kk) dbacc += As[kk * BM + tid];
    }
  };

  if (n_chunks > 0) {
    gload(ra0, rb0, 0);
    lstore(ra0, rb0, 0);
    if (n_chunks > 1) gload(ra1, rb1, 1);
    if (n_chunks > 2) gload(ra0, rb0, 2);
  }
  __syncthreads();
  // even c: set1 = chunk c+1 (landed), set0 = chunk c+2 (in flight); odd c: the sets swap roles.
  int c = 0;
  // Steady state: every load is unconditional, so the waits before the LDS stores are counted
  // (`vmcnt(N)` with the younger set still in flight) rather than full drains.
  for (; c + 4 < n_chunks; c += 2) {
    compute(c);
    lstore(ra1, rb1, c + 1);
    gload(ra1, rb1, c + 3);
    __syncthreads();
    compute(c + 1);
    lstore(ra0, rb0, c + 2);
    gload(ra0, rb0, c + 4);
    __syncthreads();
  }
  // NN epilogue operand (post-activation values for act'): on interior tiles request it before
  // the tail chunks so the 16 loads per tile land behind the remaining MFMAs.
  constexpr bool PREP = FAST && MODE == IA_GEMM_NN && TM * TN <= 2;
  float pv[PREP ? TM : 1][PREP ? TN : 1][16];
  if (PREP && g.P != nullptr) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float* pp = g.P + (long long)(bm0 + (wm * TM + i) * 32 + 4 * lh) * g.ldp + bn0 + (wn * TN + j) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) pv[i][j][r] = pp[(long long)((r & 3) + 8 * (r >> 2)) * g.ldp];
      }
  }
  for (; c < n_chunks; c += 2) {  // tail (at most four chunks)
    compute(c);
    if (c + 1 < n_chunks) lstore(ra1, rb1, c + 1);
    if (c + 3 < n_chunks) gload(ra1, rb1, c + 3);
    __syncthreads();
    if (c + 1 >= n_chunks) break;
    compute(c + 1);
    if (c + 2 < n_chunks) lstore(ra0, rb0, c + 2);
    __syncthreads();
  }

  // scattered output rows (transposed-convolution forms): the class-independent part of every tile row's target row is
  // decoded ONCE per tile into LDS (two reciprocal divisions per row instead of per output element)
  int* rowmap = reinterpret_cast<int*>(smem);
  const bool scatter = IM && MODE == IA_GEMM_NT && g.im.cm_on != 0;
  if (scatter) {
    __syncthreads();   // the last chunk's fragments have been read
    if (tid < BM) {
      const int m = min(bm0 + tid, g.M - 1);
      const int b = im_div(m, g.im.cm_OHW, g.im.cm_mOHW), p = m - b * g.im.cm_OHW;
      const int yy = im_div(p, g.im.cm_OW, g.im.cm_mOW), xx = p - yy * g.im.cm_OW;
      rowmap[tid] = b * g.im.cm_HW + yy * g.im.cm_S * g.im.cm_W + xx * g.im.cm_S;
    }
    __syncthreads();
  }
  float* C = g.C;
  const bool nt_part = MODE == IA_GEMM_NT && !IM && g.nt_split != 0;   // (block-uniform) slab of a split-K forward product
  if (MODE == IA_GEMM_TN || nt_part) C += (long long)tc.split * g.c_split_stride;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = bn0 + (wn * TN + j) * 32 + li;
      const int rbase = bm0 + (wm * TM + i) * 32 + 4 * lh;
      float bcol = 0.f;
      if (MODE == IA_GEMM_NT && g.bias != nullptr && (FAST || col < g.N)) bcol = g.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (FAST || (row < g.M && col < g.N)) {
          float v = acc[i][j][r];
          if (MODE == IA_GEMM_NT) {
            if (!nt_part) v = ia_apply_act(v + bcol, g.act);
          } else if (MODE == IA_GEMM_NN) {
            if (g.P != nullptr)
              v *= ia_act_grad_from_post(PREP ? pv[PREP ? i : 0][PREP ? j : 0][r] : g.P[(long long)row * g.ldp + col], g.act);
          }
          if (IM) {   // implicit-view forms: optional output-row scatter and ReLU mask of the tensor being written to
            int py = g.im.cm_py, px = g.im.cm_px, ccol = col;
            if (g.im.cm_on == 2) {   // sub-pixel classes fused along the columns: class = col / cm_C
              const int cls = im_div(col, g.im.cm_C, g.im.cm_mC);
              ccol = col - cls * g.im.cm_C;
              py = im_div(cls, g.im.cm_S, ~0ull / (unsigned long long)g.im.cm_S + 1ull);
              px = cls - py * g.im.cm_S;
            }
            const long long crow = scatter ? (long long)rowmap[row - bm0] + py * g.im.cm_W + px : (long long)row;
            if (MODE == IA_GEMM_NT && g.P != nullptr && !(g.P[crow * g.ldp + ccol] > 0.f)) v = 0.f;
            C[crow * g.ldc + ccol] = v;
          } else {
            C[(long long)row * g.ldc + col] = v;
          }
        }
      }
    }
  }
  if (do_db && tid < BM && bm0 + tid < g.M) g.dbias[(long long)tc.split * g.dbias_split_stride + bm0 + tid] = dbacc;
}

// Workgroup `b` of `nb` of one GEMM (the body of ia_gemm_kernel; also run per problem by the grouped launch below).
template <int WM, int WN, int TM, int TN, int MODE, bool IM = false>
__device__ __forceinline__ void gemm_block(const IaGemm& g, float* smem, const int nb, const int b) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;

  // XCD-aware block order: hardware places block b on XCD b%8. Give each XCD a CONTIGUOUS run of
  // (split, tile) work items: column tiles that share an A row-block -- and, for split-K, all the
  // tiles of one K-slab -- then run on the same XCD and are served by its L2 instead of being
  // re-fetched into eight different L2s (PMC: 98 MB -> ~algorithmic for the 256x256 wgrad).
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tiles_m = (g.M + BM - 1) / BM;
  const int tiles = tiles_m * tiles_n;
  const int q = nb >> 3, rmd = nb & 7, xcd = b & 7;
  const int lin = (xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q) + (b >> 3);
  TileCtx tc;
  tc.split = lin / tiles;
  const int t = lin - tc.split * tiles;
  tc.bm0 = (t / tiles_n) * BM;
  tc.bn0 = (t % tiles_n) * BN;
  tc.k_begin = 0;
  tc.k_end = g.K;
  if (MODE == IA_GEMM_TN || (MODE == IA_GEMM_NT && !IM && g.nt_split != 0)) {
    tc.k_begin = tc.split * g.k_per_split;
    tc.k_end = min(g.K, tc.k_begin + g.k_per_split);
  }
  tc.a_vec = (g.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  tc.b_vec = (g.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);
  // Block-uniform choice of the unguarded path: whole tile in range, full K chunks, aligned rows.
  const bool fast = tc.a_vec && tc.b_vec && (tc.bm0 + BM <= g.M) && (tc.bn0 + BN <= g.N) &&
                    (tc.k_end > tc.k_begin) && ((tc.k_end - tc.k_begin) % BK == 0);
  if (fast) {
    gemm_tile<WM, WN, TM, TN, MODE, true, IM>(g, tc, smem);
  } else {
    gemm_tile<WM, WN, TM, TN, MODE, false, IM>(g, tc, smem);
  }
}

template <int WM, int WN, int TM, int TN, int MODE, bool IM = false>
__global__ __launch_bounds__(WM* WN * 64) void ia_gemm_kernel(IaGemm g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gemm_block<WM, WN, TM, TN, MODE, IM>(g, smem, gridDim.x, blockIdx.x);
}

// The split-K TN product of the fused discriminator update with a SIDE JOB in front: workgroups [0, side_blocks) run the
// part of the update's slab reduction + Adam step that does not depend on this product (disc_reduce.h) -- ~7 MB of slab
// reads that used to wait for the product to finish, now under its MFMAs; the rest are the product's own workgroups.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void ia_gemm_tn_side_kernel(IaGemm g, ReduceArgs side, int side_blocks) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(WM * WN * 64 == 256, "disc_reduce_block is written for 256 threads");
  if ((int)blockIdx.x < side_blocks) {
    disc_reduce_block(side, blockIdx.x, side_blocks);
    return;
  }
  gemm_block<WM, WN, TM, TN, IA_GEMM_TN>(g, smem, gridDim.x - side_blocks, blockIdx.x - side_blocks);
}

// Up to six independent split-K TN GEMMs in ONE launch (32-row outputs: the hidden-layer weight gradients of the small
// AIRL stacks, 10-16 us each when launched one after the other -- latency-bound with 128 workgroups; together they
// fill the chip): workgroups [0, nb0) run problem 0, [nb0, nb0 + nb1) problem 1, ... (three problems: an update's weight
// gradients; six: those and the gradient penalty's, whose row kernel ran in the same launch as the update's).
constexpr int GEMM_GROUP_MAX = 6;
struct IaGemmGroup { IaGemm p[GEMM_GROUP_MAX]; int nb[GEMM_GROUP_MAX]; };
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(WM* WN * 64) void ia_gemm_group_tn_kernel(IaGemmGroup gs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int b = blockIdx.x;
#pragma unroll
  for (int i = 0; i < GEMM_GROUP_MAX; ++i) {   // (workgroup-uniform)
    if (b < gs.nb[i]) {
      gemm_block<WM, WN, TM, TN, IA_GEMM_TN>(gs.p[i], smem, gs.nb[i], b);
      return;
    }
    b -= gs.nb[i];
  }
}

// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline) ----
struct ProfSlot { double ms; double flops; long long launches; };
constexpr int PROF_KERNELS = 12;          // 3 modes x 4 tile configs
constexpr int PROF_POOL = 8192;           // event pairs per collection window
bool g_prof_on = false;
hipEvent_t g_prof_ev[PROF_POOL][2];
int g_prof_kid[PROF_POOL];
double g_prof_fl[PROF_POOL];
int g_prof_n = 0;
bool g_prof_init = false;
ProfSlot g_prof_acc[PROF_KERNELS];

template <int WM, int WN, int TM, int TN, int MODE, bool IM = false>
int launch_cfg(const IaGemm& g, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  using AIO = TileIO<BM, NT, MODE == IA_GEMM_TN>;
  using BIO = TileIO<BN, NT, MODE != IA_GEMM_NT>;
  constexpr size_t smem = 2 * (AIO::ELEMS + BIO::ELEMS) * sizeof(float);
  auto kern = ia_gemm_kernel<WM, WN, TM, TN, MODE, IM>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  dim3 grid(tiles * ((MODE == IA_GEMM_TN || (MODE == IA_GEMM_NT && !IM && g.nt_split != 0)) ? g.splits : 1));
  const bool prof = g_prof_on && g_prof_n < PROF_POOL;
  if (prof) (void)hipEventRecord(g_prof_ev[g_prof_n][0], stream);
  hipLaunchKernelGGL(kern, grid, dim3(NT), smem, stream, g);
  IA_CHECK_LAUNCH();
  if (prof) {
    (void)hipEventRecord(g_prof_ev[g_prof_n][1], stream);
    constexpr int cfg = (BM == 128 && BN == 128) ? 0 : ((BM == 64 && BN == 64) ? 1 : ((BM == 128 && BN == 32) ? 2 : 3));
    g_prof_kid[g_prof_n] = MODE * 4 + cfg;
    g_prof_fl[g_prof_n] = 2.0 * (double)g.M * (double)g.N * (double)g.K;
    ++g_prof_n;
  }
  return IA_OK;
}

int g_force_cfg = -1;  // tuning override (ia_gemm_set_config)

template <int MODE>
int launch_mode(const IaGemm& g, hipStream_t stream) {
  switch (g_force_cfg) {
    case 0: return launch_cfg<2, 2, 2, 2, MODE>(g, stream);  // 128 x 128
    case 1: return launch_cfg<2, 2, 1, 1, MODE>(g, stream);  // 64 x 64
    case 2: return launch_cfg<4, 1, 1, 1, MODE>(g, stream);  // 128 x 32
    case 3: return launch_cfg<1, 4, 1, 1, MODE>(g, stream);  // 32 x 128
    case 4: return launch_cfg<2, 2, 1, 2, MODE>(g, stream);  // 64 x 128
    case 5: return launch_cfg<2, 2, 2, 1, MODE>(g, stream);  // 128 x 64
    default: break;
  }
  if (g.M <= 32 && g.N > 32) return launch_cfg<1, 4, 1, 1, MODE>(g, stream);  // 32 x 128
  if (g.N <= 32) return launch_cfg<4, 1, 1, 1, MODE>(g, stream);              // 128 x 32
  // 64 x 64 (34 KB LDS, 4 workgroups per CU): this kernel is load-latency bound per workgroup, so
  // residency beats tile size on every discriminator shape measured (tools/gemm_bench.py).
  return launch_cfg<2, 2, 1, 1, MODE>(g, stream);
}

}  // namespace

int ia_launch_gemm(int mode, const IaGemm& g, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || g.K < 0) return IA_ERR_ARG;
  if (g.im.on) {   // convolution forms: 64 x 64 tiles (the shapes of the NatureCNN layers 2 and 3)
    if (mode == IA_GEMM_NT && g.N <= 32) return launch_cfg<4, 1, 1, 1, IA_GEMM_NT, true>(g, stream);   // 128 x 32
    if (mode == IA_GEMM_NT) return launch_cfg<2, 2, 1, 1, IA_GEMM_NT, true>(g, stream);
    if (mode == IA_GEMM_TN) return launch_cfg<2, 2, 1, 1, IA_GEMM_TN, true>(g, stream);
    return IA_ERR_ARG;
  }
  switch (mode) {
    case IA_GEMM_NT: return launch_mode<IA_GEMM_NT>(g, stream);
    case IA_GEMM_NN: return launch_mode<IA_GEMM_NN>(g, stream);
    case IA_GEMM_TN: return launch_mode<IA_GEMM_TN>(g, stream);
  }
  return IA_ERR_ARG;
}

int ia_launch_gemm_tn_side(const IaGemm& g, const ReduceArgs& side, hipStream_t stream) {
  if (g.M <= 0 || g.N <= 0 || g.K < 0 || g.im.on || g.splits < 1) return IA_ERR_ARG;
  constexpr int WM = 2, WN = 2, TM = 1, TN = 1, NT = 256, BM = 64, BN = 64;   // launch_mode's choice for these shapes
  using AIO = TileIO<BM, NT, true>;
  using BIO = TileIO<BN, NT, true>;
  constexpr size_t smem = 2 * (AIO::ELEMS + BIO::ELEMS) * sizeof(float);
  auto kern = ia_gemm_tn_side_kernel<WM, WN, TM, TN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int side_blocks = disc_reduce_blocks(side);
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  const bool prof = g_prof_on && g_prof_n < PROF_POOL;
  if (prof) (void)hipEventRecord(g_prof_ev[g_prof_n][0], stream);
  hipLaunchKernelGGL(kern, dim3(side_blocks + tiles * g.splits), dim3(NT), smem, stream, g, side, side_blocks);
  IA_CHECK_LAUNCH();
  if (prof) {
    (void)hipEventRecord(g_prof_ev[g_prof_n][1], stream);
    g_prof_kid[g_prof_n] = IA_GEMM_TN * 4 + 1;
    g_prof_fl[g_prof_n] = 2.0 * (double)g.M * (double)g.N * (double)g.K;
    ++g_prof_n;
  }
  return IA_OK;
}

// n <= 6 split-K TN GEMMs with M <= 32 in one launch (32 x 128 tiles)
int ia_launch_gemm_group_tn(const IaGemm* gs, int n, hipStream_t stream) {
  if (n < 1 || n > GEMM_GROUP_MAX) return IA_ERR_ARG;
  constexpr int WM = 1, WN = 4, TM = 1, TN = 1, NT = WM * WN * 64, BM = 32, BN = 128;
  using AIO = TileIO<BM, NT, true>;
  using BIO = TileIO<BN, NT, true>;
  constexpr size_t smem = 2 * (AIO::ELEMS + BIO::ELEMS) * sizeof(float);
  IaGemmGroup a{};
  int total = 0;
  for (int i = 0; i < GEMM_GROUP_MAX; ++i) {
    if (i < n) {
      const IaGemm& g = gs[i];
      if (g.M <= 0 || g.M > BM || g.N <= 0 || g.K < 0 || g.im.on || g.splits < 1) return IA_ERR_ARG;
      a.p[i] = g;
      a.nb[i] = ((g.N + BN - 1) / BN) * g.splits;
    } else {
      a.nb[i] = 0;
    }
    total += a.nb[i];
  }
  auto kern = ia_gemm_group_tn_kernel<WM, WN, TM, TN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(total), dim3(NT), smem, stream, a);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

extern "C" int ia_gemm_set_config(int cfg) { g_force_cfg = cfg; return IA_OK; }

// Profiling window: ia_prof_enable(1) .. launches .. ia_prof_collect(ms, flops, launches) [12 each].
// Kernel id = mode*4 + tile config {0:128x128, 1:64x64, 2:128x32, 3:32x128}.
extern "C" int ia_prof_enable(int on) {
  if (on && !g_prof_init) {
    for (int i = 0; i < PROF_POOL; ++i)
      for (int j = 0; j < 2; ++j)
        if (hipEventCreate(&g_prof_ev[i][j]) != hipSuccess) return IA_ERR_ARG;
    g_prof_init = true;
  }
  if (on) {
    g_prof_n = 0;
    for (int k = 0; k < PROF_KERNELS; ++k) g_prof_acc[k] = ProfSlot{0.0, 0.0, 0};
  }
  g_prof_on = on != 0;
  return IA_OK;
}

extern "C" int ia_prof_collect(double* ms, double* flops, long long* launches) {
  for (int i = 0; i < g_prof_n; ++i) {
    if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) return IA_ERR_ARG;
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof_ev[i][0], g_prof_ev[i][1]) != hipSuccess) return IA_ERR_ARG;
    ProfSlot& a = g_prof_acc[g_prof_kid[i]];
    a.ms += t; a.flops += g_prof_fl[i]; a.launches += 1;
  }
  g_prof_n = 0;
  for (int k = 0; k < PROF_KERNELS; ++k) {
    ms[k] = g_prof_acc[k].ms; flops[k] = g_prof_acc[k].flops; launches[k] = g_prof_acc[k].launches;
  }
  return IA_OK;
}

// C-ABI test/bench entry for the raw contraction (device pointers, row-major fp32).
extern "C" int ia_gemm_f32(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                           int N, int K, const float* bias, int act, const float* P, int ldp, int splits,
                           float* dbias, void* stream) {
  IaGemm g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.bias = bias; g.act = act; g.P = P; g.ldp = ldp; g.dbias = dbias;
  g.splits = splits > 0 ? splits : 1;
  g.k_per_split = ((K + g.splits - 1) / g.splits + BK - 1) / BK * BK;
  g.c_split_stride = (long long)M * ldc;
  g.dbias_split_stride = M;
  return ia_launch_gemm(mode, g, (hipStream_t)stream);
}

// Forward product of a layer with FEW output tiles and a long K (the NatureCNN's 3 136 -> 512 linear layer at rollout / minibatch
// sizes: 8 - 32 tiles of 64 x 64, 98 K chunks each on 8 - 32 of 256 compute units): split along K into `splits` slabs of
// `partials` [splits][M][N] (one launch), then C = act(sum of the slabs in split order + bias) (`reduce_bias_act_kernel`).
// Deterministic; the sum order differs from the unsplit product's (fp32 rounding).
namespace {
__global__ void reduce_bias_act_kernel(const float* __restrict__ partials, int splits, int M, int N, const float* __restrict__ bias,
                                       int act, float* __restrict__ C, int ldc) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // element quad
  const long long n = (long long)M * N;
  if (4 * i >= n) return;
  const int row = (int)((4 * i) / N), col = (int)((4 * i) - (long long)row * N);
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < splits; ++k) {
    const f4 t = *reinterpret_cast<const f4*>(partials + (long long)k * n + 4 * i);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  if (bias != nullptr) {
    const f4 b = *reinterpret_cast<const f4*>(bias + col);
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
  }
  f4 o = {ia_apply_act(s.x, act), ia_apply_act(s.y, act), ia_apply_act(s.z, act), ia_apply_act(s.w, act)};
  *reinterpret_cast<f4*>(C + (long long)row * ldc + col) = o;
}
}  // namespace

extern "C" int ia_gemm_f32_nt_splitk(const float* A, int lda, const float* B, int ldb, float* partials, float* C, int ldc, int M,
                                     int N, int K, const float* bias, int act, int splits, void* stream) {
  if (splits < 1 || M <= 0 || N <= 0 || K <= 0 || (N & 3) || (ldc & 3) || partials == nullptr ||
      ((reinterpret_cast<uintptr_t>(partials) | reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15))
    return IA_ERR_ARG;
  IaGemm g{};
  g.A = A; g.B = B; g.C = partials; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.bias = nullptr; g.act = 0;
  g.nt_split = 1;
  g.splits = splits;
  g.k_per_split = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  g.c_split_stride = (long long)M * N;
  int rc = ia_launch_gemm(IA_GEMM_NT, g, (hipStream_t)stream);
  if (rc) return rc;
  const long long quads = ((long long)M * N) >> 2;
  hipLaunchKernelGGL(reduce_bias_act_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials,
                     splits, M, N, bias, act, C, ldc);
  IA_CHECK_LAUNCH();
  return IA_OK;
}

// ia_gemm_f32 with the k-contiguous [rows, KH*KW*Cin] operand of a convolution given IMPLICITLY as the im2col view of
// the channel-last activation tensor x[Bn, H, W, Cin] (no column buffer is ever written or read):
//   NT: C[M, N] = act(view(x)[M, K] . Wt[N, K]^T + bias)    M = Bn*OH*OW rows, K = KH*KW*Cin   (convolution forward)
//   TN: C_s[M, N] = dout[Ks, M]^T . view(x)[Ks, N]           M = Cout, N = KH*KW*Cin, Ks = rows (weight gradient)
// Requirements: Cin % 4 == 0 and (KW*Cin) % 32 == 0 (a 32-deep K chunk stays inside one kernel row).
namespace {
int gemm_im2col(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                const float* bias, int act, int splits, float* dbias, int H, int W, int Cin, int KH, int KW, int S, int P,
                const int* cmap, const float* relu_mask, void* stream);
}
extern "C" int ia_gemm_f32_im2col(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                                  int N, int K, const float* bias, int act, int splits, float* dbias, int H, int W,
                                  int Cin, int KH, int KW, int S, void* stream) {
  return gemm_im2col(mode, A, lda, B, ldb, C, ldc, M, N, K, bias, act, splits, dbias, H, W, Cin, KH, KW, S, 0, nullptr, nullptr,
                     stream);
}
// The same with zero padding P on every side of the image (taps outside read as 0) and, for mode NT, an optional scatter
// of the output rows: cmap = {S_out, py, px, H_out, W_out} sends row (b, y', x') of the [OH, OW] output grid to row
// (b, y'*S_out + py, x'*S_out + px) of a [H_out, W_out] grid in C -- the input gradient of a strided convolution is one
// such padded stride-1 convolution over dout per sub-pixel class (py, px). cmap = NULL: rows in place.
// relu_mask (mode NT, nullable): a tensor laid out like C; outputs are zeroed where it is <= 0 (the ReLU of the layer below).
extern "C" int ia_gemm_f32_im2col_pad(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M,
                                      int N, int K, const float* bias, int act, int splits, float* dbias, int H, int W,
                                      int Cin, int KH, int KW, int S, int P, const int* cmap, const float* relu_mask,
                                      void* stream) {
  if (P < 0) return IA_ERR_ARG;
  return gemm_im2col(mode, A, lda, B, ldb, C, ldc, M, N, K, bias, act, splits, dbias, H, W, Cin, KH, KW, S, P, cmap, relu_mask,
                     stream);
}
namespace {
int gemm_im2col(int mode, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                const float* bias, int act, int splits, float* dbias, int H, int W, int Cin, int KH, int KW, int S, int P,
                const int* cmap, const float* relu_mask, void* stream) {
  if (H + 2 * P < KH || W + 2 * P < KW || S <= 0 || Cin % 4 != 0 || (KW * Cin) % 32 != 0) return IA_ERR_ARG;
  const int OH = (H + 2 * P - KH) / S + 1, OW = (W + 2 * P - KW) / S + 1;
  auto magic = [](int d) { return d <= 1 ? 0ull : (~0ull / (unsigned long long)d) + 1ull; };
  IaGemm g{};
  g.im.on = 1; g.im.OW = OW; g.im.OHW = OH * OW; g.im.W = W; g.im.C = Cin; g.im.S = S; g.im.HWC = H * W * Cin;
  g.im.seg = KW * Cin; g.im.rstride = W * Cin;
  g.im.mOW = magic(OW); g.im.mOHW = magic(OH * OW); g.im.mseg = magic(KW * Cin);
  g.im.pad = P; g.im.H = H; g.im.mC = magic(Cin);
  if (cmap != nullptr) {
    if (mode != IA_GEMM_NT || cmap[0] <= 0) return IA_ERR_ARG;
    g.im.cm_on = 1; g.im.cm_OW = OW; g.im.cm_OHW = OH * OW; g.im.cm_S = cmap[0]; g.im.cm_py = cmap[1]; g.im.cm_px = cmap[2];
    g.im.cm_W = cmap[4]; g.im.cm_HW = cmap[3] * cmap[4];
    g.im.cm_mOW = g.im.mOW; g.im.cm_mOHW = g.im.mOHW;
    if (cmap[1] < 0) {   // py < 0: all S_out^2 classes at once, N = S_out^2 * (channels per class)
      if (N % (cmap[0] * cmap[0]) != 0) return IA_ERR_ARG;
      g.im.cm_on = 2; g.im.cm_C = N / (cmap[0] * cmap[0]); g.im.cm_mC = magic(g.im.cm_C);
      ldc = g.im.cm_C;
    }
  }
  const long long rows = mode == IA_GEMM_NT ? M : K;
  if (rows % g.im.OHW != 0 || (mode == IA_GEMM_NT ? K : N) != KH * KW * Cin) return IA_ERR_ARG;
  if ((rows / g.im.OHW) * (long long)g.im.HWC >= (1ll << 31)) return IA_ERR_UNSUPPORTED;   // 32-bit element offsets
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.bias = bias; g.act = act; g.dbias = dbias;
  g.P = relu_mask; g.ldp = ldc;
  g.splits = splits > 0 ? splits : 1;
  g.k_per_split = ((K + g.splits - 1) / g.splits + BK - 1) / BK * BK;
  g.c_split_stride = (long long)M * ldc;
  g.dbias_split_stride = M;
  return ia_launch_gemm(mode, g, (hipStream_t)stream);
}
}  // namespace
