// Operator apply on a lattice (format 4 with offsets -b, -a, -1, +1, +a, +b): the tiled kernel and the z-marching fused
// CG step, with the geometry tests that decide when they apply.
#include <algorithm>

#include "spmv_device.hpp"

namespace storm {

template <bool DOT, bool WLOAD, int TZ, int HL, bool FUSE = false>
__global__ __launch_bounds__(kBlock) void spmv_canon_tile_kernel(SellArgs A, CanonTileArgs T, Scal alpha_s, Scal beta_s,
                                                                 const double *__restrict__ x, double *__restrict__ y,
                                                                 DotArgs dot, const int *done, IpcSendArgs S, CgFuseArgs F) {
  if (!FUSE && (int)blockIdx.x < S.sp.n_blocks) {  // the first blocks of a partitioned operator's interior launch send its rows
    ipc_halo_send_block(S.w, S.sp, x, (int)blockIdx.x);
    return;
  }
  if (FUSE && *F.iteration < F.my_iteration) return;  // enqueued past convergence: that iteration never ran
  const int done_flag = done ? *done : 0;
  const double cg_a = FUSE ? *F.ca : 0.0, cg_b = FUSE ? *F.cb : 0.0;
  const char *rb_ = reinterpret_cast<const char *>(F.r);
  extern __shared__ __attribute__((aligned(16))) double tile_sh[];  // [TZ][a + kTileRun + a]
  __shared__ double dict_sh[32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_tiles = (int)gridDim.x - S.sp.n_blocks, tb = (int)blockIdx.x - S.sp.n_blocks;
  const int bidx = T.reverse ? n_tiles - 1 - tb : tb;
  int zc, yt;
  if (T.per_xcd > 0) {
    const int xcd = bidx & (kNumXcd - 1), j = bidx >> 3;
    zc = j / T.per_xcd;
    yt = xcd * T.per_xcd + (j - zc * T.per_xcd);
  } else {
    zc = bidx / T.tiles_per_plane;
    yt = bidx - zc * T.tiles_per_plane;
  }
  const int a = T.a, b = T.b;
  const int p0 = yt * kTileRun, z0 = T.plane0 + zc * TZ;
  const int ldw = kTileRun + 2 * a;  // doubles per plane of the LDS copy
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const bool w_is_x = DOT && dot.w == x;
  const char *xb = reinterpret_cast<const char *>(x);
  const char *xg_base = xb - (size_t)kVecGuard * 8;
  char *yb = reinterpret_cast<char *>(y);
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

  const double dict_word = A.dict[lane & 31];
  // ---- everything this wave reads from memory, issued back to back: own rows first (the LDS copy waits for them only)
  bool valid_a[TZ][2], valid_b[TZ][2];
  uint32_t rc[TZ][2];
  double2v xi[TZ][2];
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int q = p0 + 256 * wave + 128 * g + 2 * lane;                 // row of the plane
      const int64_t row = (int64_t)(z0 + t) * b + q;
      const bool in_plane = q < b && z0 + t < T.plane_end;
      valid_a[t][g] = in_plane && row <= (int64_t)last_row, valid_b[t][g] = in_plane && row + 1 <= (int64_t)last_row;
      rc[t][g] = row <= (int64_t)last_row ? (uint32_t)row : (last_row & ~1u);  // pairs past the end re-read the last pair
      xi[t][g] = *reinterpret_cast<const double2v *>(xb + (size_t)(rc[t][g] << 3));
    }
  if (FUSE) {
    // x += alpha p (the OLD direction), then p' = r + beta p takes p's place in the registers
#pragma unroll
    for (int t = 0; t < TZ; ++t)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        double2v *xp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.x) + (size_t)(rc[t][g] << 3));
        const double2v xv = __builtin_nontemporal_load(xp_);
        const double2v rv = *reinterpret_cast<const double2v *>(rb_ + (size_t)(rc[t][g] << 3));
        double2v xn, pn;
        xn.x = __builtin_fma(cg_a, xi[t][g].x, xv.x), xn.y = __builtin_fma(cg_a, xi[t][g].y, xv.y);
        pn.x = __builtin_fma(cg_b, xi[t][g].x, rv.x), pn.y = __builtin_fma(cg_b, xi[t][g].y, rv.y);
        if (valid_b[t][g]) __builtin_nontemporal_store(xn, xp_);
        else if (valid_a[t][g]) F.x[rc[t][g]] = xn.x;
        xi[t][g] = pn;
        if (!done_flag) {
          double2v *pp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.p_out) + (size_t)(rc[t][g] << 3));
          if (valid_b[t][g]) __builtin_nontemporal_store(pn, pp_);
          else if (valid_a[t][g]) F.p_out[rc[t][g]] = pn.x;
        }
      }
    if (done_flag) return;  // converged in that iteration: x is final, no new direction, no apply
  }
  const char *rg_base = FUSE ? rb_ - (size_t)kVecGuard * 8 : nullptr;
  double2v halo[HL];
  int halo_at[HL];  // LDS index (doubles) of the pair, -1: none
#pragma unroll
  for (int i = 0; i < HL; ++i) {
    const unsigned h = threadIdx.x + (unsigned)kBlock * i;                // pair h of the tile's TZ * a halo pairs
    const unsigned t = __umulhi(h, T.a_magic), u = h - t * (unsigned)a;   // plane, pair within the plane's halo
    const bool on = t < (unsigned)TZ;
    const int jj = (int)(2 * u) < a ? (int)(2 * u) - a : kTileRun + (int)(2 * u) - a;  // tile-relative row: [-a, 0) or [1024, 1024 + a)
    int64_t gi = (int64_t)(z0 + (int)t) * b + p0 + jj + kVecGuard;        // guard-relative, clamped like every gather
    gi = gi < 0 ? 0 : gi;
    gi = gi > (int64_t)T.max_gather ? (int64_t)T.max_gather : gi;
    halo_at[i] = on ? (int)t * ldw + a + jj : -1;
    halo[i] = double2v{0.0, 0.0};
    if (on) halo[i] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)gi << 3));
    if (FUSE && on) {  // the halo row's new direction, with its owner's expression
      const double2v rv = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)gi << 3));
      halo[i].x = __builtin_fma(cg_b, halo[i].x, rv.x), halo[i].y = __builtin_fma(cg_b, halo[i].y, rv.y);
    }
  }
  u64x2 vw[TZ][2];
  double2v wi[WLOAD ? TZ : 1][2];
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      vw[t][g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(A.pack + (size_t)(rc[t][g] << 3)));
      if (WLOAD) wi[t][g] = *reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(dot.w) + (size_t)(rc[t][g] << 3));
    }
  double2v xlo[2], xhi[2];  // the planes below the first and above the last one of the tile
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int lo = (int)rc[0][g] - b + kVecGuard, hi = (int)rc[TZ - 1][g] + b + kVecGuard;
    lo = lo < 0 ? 0 : lo;
    hi = hi > T.max_gather ? T.max_gather : hi;
    xlo[g] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)lo << 3));
    xhi[g] = *reinterpret_cast<const double2v *>(xg_base + (size_t)((uint32_t)hi << 3));
    if (FUSE) {
      const double2v rl = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)lo << 3));
      const double2v rh = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)hi << 3));
      xlo[g].x = __builtin_fma(cg_b, xlo[g].x, rl.x), xlo[g].y = __builtin_fma(cg_b, xlo[g].y, rl.y);
      xhi[g].x = __builtin_fma(cg_b, xhi[g].x, rh.x), xhi[g].y = __builtin_fma(cg_b, xhi[g].y, rh.y);
    }
  }
  // ---- the LDS copy of the tile's x (own rows + halo rows), one barrier
  if (lane < 32) dict_sh[lane] = dict_word;  // every wave stores the same words
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g)
      *reinterpret_cast<double2v *>(&tile_sh[t * ldw + a + 256 * wave + 128 * g + 2 * lane]) = xi[t][g];
#pragma unroll
  for (int i = 0; i < HL; ++i)
    if (halo_at[i] >= 0) *reinterpret_cast<double2v *>(&tile_sh[halo_at[i]]) = halo[i];
  __syncthreads();
  double dot_a = 0.0, dot_b = 0.0;
#pragma unroll
  for (int t = 0; t < TZ; ++t)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int at = t * ldw + a + 256 * wave + 128 * g + 2 * lane;  // this pair in the LDS copy
      double2v xg[6];
      xg[0] = t == 0 ? xlo[g] : xi[t == 0 ? 0 : t - 1][g];
      xg[5] = t == TZ - 1 ? xhi[g] : xi[t == TZ - 1 ? t : t + 1][g];
      xg[1] = *reinterpret_cast<const double2v *>(&tile_sh[at - a]);
      xg[4] = *reinterpret_cast<const double2v *>(&tile_sh[at + a]);
      double el = 0.0;
      if (lane == 0) el = tile_sh[at - 1];
      if (lane == kWave - 1) el = tile_sh[at + 2];
      const double left = dpp_shift<0x138>(xi[t][g].y);   // wave_shr:1 -- lane i receives lane i - 1
      const double right = dpp_shift<0x130>(xi[t][g].x);  // wave_shl:1 -- lane i receives lane i + 1
      xg[2].x = lane == 0 ? el : left;
      xg[2].y = xi[t][g].x;
      xg[3].x = xi[t][g].y;
      xg[3].y = lane == kWave - 1 ? el : right;
      double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const unsigned ba = (unsigned)(vw[t][g].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(vw[t][g].y >> (8 * (k + 1))) & 0xffu;
        acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[k].x - xi[t][g].x);
        acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[k].y - xi[t][g].y);
      }
      const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[t][g].x & 0xffu));
      const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)vw[t][g].y & 0xffu));
      // (spelled out: beta x rounded on its own, then the two FMAs the plain kernel's expression contracts to --
      //  `(accumulate ? y : beta x) + alpha (acc + ext x)` -- so that both kernels round alike)
      double2v yi;
      yi.x = __builtin_fma(alpha, __builtin_fma(ext_a, xi[t][g].x, acc_a), beta * xi[t][g].x);
      yi.y = __builtin_fma(alpha, __builtin_fma(ext_b, xi[t][g].y, acc_b), beta * xi[t][g].y);
      if (!done_flag) {
        double2v *yp = reinterpret_cast<double2v *>(yb + (size_t)(rc[t][g] << 3));
        if (valid_b[t][g]) *yp = yi;
        else if (valid_a[t][g]) y[rc[t][g]] = yi.x;  // the odd last row
      }
      if (DOT) {
        yi.x = valid_a[t][g] ? yi.x : 0.0;
        yi.y = valid_b[t][g] ? yi.y : 0.0;
        const double2v wv = WLOAD ? wi[WLOAD ? t : 0][g] : xi[t][g];
        const double pa = dot.w ? wv.x * yi.x + wv.y * yi.y : 0.0;
        const double pb = yi.x * yi.x + yi.y * yi.y;
        dot_a = (t == 0 && g == 0) ? pa : dot_a + pa;
        dot_b = (t == 0 && g == 0) ? pb : dot_b + pb;
      }
    }
  (void)w_is_x;
  if (done_flag) return;
  if (DOT) {
    dot_a = wave_sum_to_lane63(dot_a);
    if (dot.yy) dot_b = wave_sum_to_lane63(dot_b);
    if (dot.tickets == nullptr) {
      if (lane == kWave - 1) {
        const int slot = dot.block_offset + bidx * (kBlock / kWave) + wave;
        dot.partials[slot] = dot_a;
        if (dot.yy) dot.partials[dot.nblocks_total + slot] = dot_b;
      }
    } else {  // the reduction finishes here: block partial, then two levels of tickets
      __shared__ double wave_part[2 * (kBlock / kWave)];
      if (lane == kWave - 1) wave_part[wave] = dot_a, wave_part[kBlock / kWave + wave] = dot.yy ? dot_b : 0.0;
      __syncthreads();
      if (wave != 0) return;
      const double mine[2] = {(wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]),
                              (wave_part[4] + wave_part[5]) + (wave_part[6] + wave_part[7])};
      double total[2];
      const TicketArgs tk{dot.tickets, dot.partials, dot.part2};
      if (ticket_reduce_wave0<2>(tk, mine, dot.yy ? 2 : 1, (unsigned)bidx, (unsigned)n_tiles, total) && lane == 0) {
        *dot.out0 = total[0];
        if (dot.yy) *dot.out1 = total[1];
      }
    }
  }
}

template <int HLP>  // HLP: halo pairs per thread and plane: ceil(a / 256)
__global__ __launch_bounds__(kBlock) void cg_step_march_kernel(SellArgs A, MarchArgs M, Scal alpha_s, Scal beta_s,
                                                               const double *__restrict__ p_in, double *__restrict__ z_out,
                                                               DotArgs dot, const int *done, CgFuseArgs F, IpcSendArgs S) {
  if ((int)blockIdx.x < S.sp.n_blocks) {
    // a partitioned operator: the first blocks send the NEW direction's boundary rows (whatever the iteration gate
    // below says: every rank enqueues the same exchanges, and the receivers poll for them)
    ipc_halo_send_block(S.w, S.sp, p_in, (int)blockIdx.x, F.r, *F.cb);
    return;
  }
  if (F.iteration != nullptr && *F.iteration < F.my_iteration) return;  // enqueued past convergence: that iteration never ran
  const int done_flag = done ? *done : 0;
  const CanonTileArgs &T = M.T;
  extern __shared__ __attribute__((aligned(16))) double tile_sh[];  // [3][a + kTileRun + a]
  __shared__ double dict_sh[32];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_march = (int)gridDim.x - S.sp.n_blocks, mb = (int)blockIdx.x - S.sp.n_blocks;
  const int bidx = T.reverse ? n_march - 1 - mb : mb;
  int zc, yt;
  if (T.per_xcd > 0) {
    const int xcd = bidx & (kNumXcd - 1), j = bidx >> 3;
    zc = j / T.per_xcd;
    yt = xcd * T.per_xcd + (j - zc * T.per_xcd);
  } else {
    zc = bidx / T.tiles_per_plane;
    yt = bidx - zc * T.tiles_per_plane;
  }
  const int a = T.a, b = T.b;
  const int p0 = yt * kTileRun;
  const int z_begin = zc * M.zc_planes, z_end = min(z_begin + M.zc_planes, T.plane_end);
  const int ldw = kTileRun + 2 * a;
  const double alpha = ld_scal2(alpha_s), beta = ld_scal2(beta_s);
  const double cg_a = F.ca ? *F.ca : F.ca_imm, cg_b = F.cb ? *F.cb : F.cb_imm;
  const uint32_t last_row = (uint32_t)(A.n_rows - 1);
  const char *pb = reinterpret_cast<const char *>(p_in), *rb = reinterpret_cast<const char *>(F.r);
  const char *pg_base = pb - (size_t)kVecGuard * 8, *rg_base = rb - (size_t)kVecGuard * 8;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const double dict_word = A.dict[lane & 31];
  if (lane < 32) dict_sh[lane] = dict_word;  // every wave stores the same words; the first barrier below covers them

  // what is in flight for ONE plane: the own rows' p, r, x and record, and this thread's share of the halo lines
  struct Flight {
    double2v p[2], r[2], x[2], hp[HLP], hr[HLP];
    u64x2 w[2];
    uint32_t rc[2];
    bool va[2], vb[2];
    int hat[HLP];
  };
  auto issue = [&](int zp, bool own, Flight &f) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int q = p0 + 256 * wave + 128 * g + 2 * lane;
      const int64_t row = (int64_t)zp * b + q;
      const bool in_plane = q < b && own;
      f.va[g] = in_plane && row <= (int64_t)last_row, f.vb[g] = in_plane && row + 1 <= (int64_t)last_row;
      int64_t gi = row + kVecGuard;  // guard-relative, clamped: a plane below the first / above the last reads zeros or x's last rows, weight 0
      gi = gi < 0 ? 0 : gi;
      gi = gi > (int64_t)T.max_gather ? (int64_t)T.max_gather : gi;
      f.rc[g] = (row >= 0 && row <= (int64_t)last_row) ? (uint32_t)row : (last_row & ~1u);
      f.p[g] = *reinterpret_cast<const double2v *>(pg_base + (size_t)((uint32_t)gi << 3));
      f.r[g] = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)gi << 3));
      if (own) {
        f.x[g] = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(reinterpret_cast<const char *>(F.x) + (size_t)(f.rc[g] << 3)));
        f.w[g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(A.pack + (size_t)(f.rc[g] << 3)));
      }
    }
#pragma unroll
    for (int i = 0; i < HLP; ++i) {
      const int u = (int)threadIdx.x + kBlock * i;  // pair u of the plane's a halo pairs
      const bool on = own && u < a;
      const int jj = 2 * u < a ? 2 * u - a : kTileRun + 2 * u - a;
      int64_t gi = (int64_t)zp * b + p0 + jj + kVecGuard;
      gi = gi < 0 ? 0 : gi;
      gi = gi > (int64_t)T.max_gather ? (int64_t)T.max_gather : gi;
      f.hat[i] = on ? a + jj : -1;
      f.hp[i] = f.hr[i] = double2v{0.0, 0.0};
      if (on) {
        f.hp[i] = *reinterpret_cast<const double2v *>(pg_base + (size_t)((uint32_t)gi << 3));
        f.hr[i] = *reinterpret_cast<const double2v *>(rg_base + (size_t)((uint32_t)gi << 3));
      }
    }
  };
  // the plane has arrived: p' of the own rows (-> out), x and p' stored, the LDS copy of the plane filled
  auto consume = [&](bool own, const Flight &f, double2v (&out)[2], double *buf) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      double2v pn;
      pn.x = __builtin_fma(cg_b, f.p[g].x, f.r[g].x), pn.y = __builtin_fma(cg_b, f.p[g].y, f.r[g].y);
      out[g] = pn;
      if (own) {
        double2v *pp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.p_out) + (size_t)(f.rc[g] << 3));
        double2v xn;
        xn.x = __builtin_fma(cg_a, f.p[g].x, f.x[g].x), xn.y = __builtin_fma(cg_a, f.p[g].y, f.x[g].y);
        double2v *xp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.x) + (size_t)(f.rc[g] << 3));
        if (f.vb[g]) __builtin_nontemporal_store(xn, xp_), __builtin_nontemporal_store(pn, pp_);
        else if (f.va[g]) F.x[f.rc[g]] = xn.x, F.p_out[f.rc[g]] = pn.x;
        *reinterpret_cast<double2v *>(&buf[a + 256 * wave + 128 * g + 2 * lane]) = pn;
      }
    }
    if (own) {
#pragma unroll
      for (int i = 0; i < HLP; ++i)
        if (f.hat[i] >= 0) {
          double2v hn;
          hn.x = __builtin_fma(cg_b, f.hp[i].x, f.hr[i].x), hn.y = __builtin_fma(cg_b, f.hp[i].y, f.hr[i].y);
          *reinterpret_cast<double2v *>(&buf[f.hat[i]]) = hn;
        }
    }
  };

  if (done_flag) {  // converged in that iteration: only x += alpha p is left to do
    for (int zp = z_begin; zp < z_end; ++zp) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int q = p0 + 256 * wave + 128 * g + 2 * lane;
        const int64_t row = (int64_t)zp * b + q;
        if (q < b && row <= (int64_t)last_row) {
          double2v *xp_ = reinterpret_cast<double2v *>(reinterpret_cast<char *>(F.x) + (size_t)((uint32_t)row << 3));
          if (row + 1 <= (int64_t)last_row) {
            const double2v pv = *reinterpret_cast<const double2v *>(pb + (size_t)((uint32_t)row << 3));
            double2v xv = *xp_;
            xv.x = __builtin_fma(cg_a, pv.x, xv.x), xv.y = __builtin_fma(cg_a, pv.y, xv.y);
            *xp_ = xv;
          } else {
            F.x[row] = __builtin_fma(cg_a, p_in[row], F.x[row]);
          }
        }
      }
    }
    return;
  }

  double2v pm[2], pc[2], pn[2];  // p' of the plane behind / at / ahead IN MARCHING ORDER
  u64x2 wc[2];
  uint32_t rcc[2];
  bool vac[2], vbc[2];
  Flight fl;
  const bool down = M.alternate != 0 && (zc & 1) != 0;  // (block-uniform)
  const int nz = z_end - z_begin;
  auto plane = [&](int s) { return down ? z_end - 1 - s : z_begin + s; };  // s = -1 and s = nz: the planes next to the chunk
  auto lds_of = [&](int zp) { return tile_sh + ((zp % 3 + 3) % 3) * ldw; };
  issue(plane(-1), false, fl);
  consume(false, fl, pm, nullptr);
  issue(plane(0), true, fl);
  consume(true, fl, pc, lds_of(plane(0)));
#pragma unroll
  for (int g = 0; g < 2; ++g) wc[g] = fl.w[g], rcc[g] = fl.rc[g], vac[g] = fl.va[g], vbc[g] = fl.vb[g];
  issue(plane(1), 1 < nz, fl);
  double dot_a = 0.0, dot_b = 0.0;
  for (int s = 0; s < nz; ++s) {
    const int zp = plane(s);
    const bool next_own = s + 1 < nz;
    u64x2 wn[2];
    uint32_t rcn[2];
    bool van[2], vbn[2];
    consume(next_own, fl, pn, lds_of(plane(s + 1)));
#pragma unroll
    for (int g = 0; g < 2; ++g) wn[g] = fl.w[g], rcn[g] = fl.rc[g], van[g] = fl.va[g], vbn[g] = fl.vb[g];
    if (s + 2 <= nz) issue(plane(s + 2), s + 2 < nz, fl);  // (one plane ahead of the one consumed next)
    __syncthreads();  // the LDS copy of plane zp is complete; the buffer two planes back is free again
    const double *buf = lds_of(zp);
    const bool applies = zp >= M.apply_begin && zp < M.apply_end;  // (block-uniform)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (!applies) break;
      const int at = a + 256 * wave + 128 * g + 2 * lane;
      double2v xg[6];
      xg[0] = down ? pn[g] : pm[g], xg[5] = down ? pm[g] : pn[g];  // the planes below / above, whichever way the block marches
      xg[1] = *reinterpret_cast<const double2v *>(&buf[at - a]);
      xg[4] = *reinterpret_cast<const double2v *>(&buf[at + a]);
      double el = 0.0;
      if (lane == 0) el = buf[at - 1];
      if (lane == kWave - 1) el = buf[at + 2];
      const double left = dpp_shift<0x138>(pc[g].y);
      const double right = dpp_shift<0x130>(pc[g].x);
      xg[2].x = lane == 0 ? el : left;
      xg[2].y = pc[g].x;
      xg[3].x = pc[g].y;
      xg[3].y = lane == kWave - 1 ? el : right;
      double acc_a = 0.0, acc_b = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const unsigned ba = (unsigned)(wc[g].x >> (8 * (k + 1))) & 0xffu, bb = (unsigned)(wc[g].y >> (8 * (k + 1))) & 0xffu;
        acc_a += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ba) * (xg[k].x - pc[g].x);
        acc_b += *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + bb) * (xg[k].y - pc[g].y);
      }
      const double ext_a = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)wc[g].x & 0xffu));
      const double ext_b = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(dict_sh) + ((unsigned)wc[g].y & 0xffu));
      double2v yi;
      yi.x = __builtin_fma(alpha, __builtin_fma(ext_a, pc[g].x, acc_a), beta * pc[g].x);
      yi.y = __builtin_fma(alpha, __builtin_fma(ext_b, pc[g].y, acc_b), beta * pc[g].y);
      double2v *yp = reinterpret_cast<double2v *>(reinterpret_cast<char *>(z_out) + (size_t)(rcc[g] << 3));
      if (vbc[g]) *yp = yi;
      else if (vac[g]) z_out[rcc[g]] = yi.x;
      yi.x = vac[g] ? yi.x : 0.0;
      yi.y = vbc[g] ? yi.y : 0.0;
      dot_a += pc[g].x * yi.x + pc[g].y * yi.y;
      dot_b += yi.x * yi.x + yi.y * yi.y;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
      pm[g] = pc[g], pc[g] = pn[g], wc[g] = wn[g], rcc[g] = rcn[g], vac[g] = van[g], vbc[g] = vbn[g];
  }
  dot_a = wave_sum_to_lane63(dot_a);
  if (dot.yy) dot_b = wave_sum_to_lane63(dot_b);
  if (dot.tickets == nullptr) {
    if (lane == kWave - 1) {
      const int slot = dot.block_offset + bidx * (kBlock / kWave) + wave;
      dot.partials[slot] = dot_a;
      if (dot.yy) dot.partials[dot.nblocks_total + slot] = dot_b;
    }
    return;
  }
  // the reduction finishes here (one rank, unsplit): block partial, then two levels of tickets -- no final-pass launch
  __shared__ double wave_part[2 * (kBlock / kWave)];
  if (lane == kWave - 1) wave_part[wave] = dot_a, wave_part[kBlock / kWave + wave] = dot.yy ? dot_b : 0.0;
  __syncthreads();
  if (wave != 0) return;
  const double mine[2] = {(wave_part[0] + wave_part[1]) + (wave_part[2] + wave_part[3]),
                          (wave_part[4] + wave_part[5]) + (wave_part[6] + wave_part[7])};
  double total[2];
  const TicketArgs tk{dot.tickets, dot.partials, dot.part2};
  if (ticket_reduce_wave0<2>(tk, mine, dot.yy ? 2 : 1, (unsigned)bidx, (unsigned)n_march, total) && lane == 0) {
    *dot.out0 = total[0];
    if (dot.yy) *dot.out1 = total[1];
  }
}


// The tiled format-4 kernel applies to an UNSPLIT, non-accumulating launch of an operator whose common offsets are
// (-b, -a, -1, +1, +a, +b) with a, b even, a <= 512, b >= 2 a, and enough planes to fill tiles.
int canon_tile_planes(const storm_hip_op *op) {
  const int64_t tz = op->ctx->opt_spmv_canon_tile;
  return tz == 4 ? 4 : 2;
}
// interior = true: the launch over a partitioned (mixed) operator's interior groups, which must be whole planes
// [int_plane0, int_plane1) (op_upload_slice_lists checks that).
bool canon_tile_geometry(const storm_hip_op *op, CanonTileArgs *T, int *n_blocks, bool interior) {
  if (op->ctx->opt_spmv_canon_tile == 0 || op->pair != 2 || op->canon_k != 6 || op->canon_m1 != 2) return false;
  if (interior != (op->d_bnd_pack != nullptr)) return false;  // (a mixed operator always runs as its two lists)
  if (interior && op->int_plane1 <= op->int_plane0) return false;
  const int *o = op->canon_off;
  const int a = o[4], b = o[5];
  if (o[0] != -b || o[1] != -a || o[2] != -1 || o[3] != 1) return false;
  if (a < 2 || a > 512 || (a & 1) || (b & 1) || b < 2 * a) return false;
  const int tz = canon_tile_planes(op);
  if ((int64_t)sizeof(double) * tz * (kTileRun + 2 * a) > 60 * 1024) return false;  // the LDS copy of a tile (64 KiB per block)
  const int64_t plane0 = interior ? op->int_plane0 : 0, plane1 = interior ? op->int_plane1 : (op->n_rows + b - 1) / b;
  const int64_t planes = plane1 - plane0;
  if (planes < 2 * tz || op->n_rows < op->ctx->opt_spmv_canon_tile_min_rows) return false;  // small operators: the plain kernel (or the latency path)
  T->a = a, T->b = b;
  T->a_magic = (unsigned)((((uint64_t)1 << 32) + (uint64_t)a - 1) / (uint64_t)a);
  T->tiles_per_plane = (b + kTileRun - 1) / kTileRun;
  T->per_xcd = (T->tiles_per_plane % kNumXcd == 0 && op->ctx->opt_spmv_xcd_remap != 0) ? T->tiles_per_plane / kNumXcd : 0;
  T->max_gather = (int)(op->n_rows + op->n_halo) + kVecGuard + 2;
  T->reverse = op->ctx->spmv_reverse;
  T->plane0 = (int)plane0, T->plane_end = (int)plane1;
  *n_blocks = (int)(((planes + tz - 1) / tz) * T->tiles_per_plane);
  return true;
}

// The z-marching form of the fused CG step: blocks of 1024 rows x opt_cg_march planes.
// partitioned: a mixed operator (interior planes on format 4, boundary groups on format 3): the march covers ALL owned
// planes for x and p', applies the operator to the interior ones.
bool cg_march_geometry(const storm_hip_op *op, MarchArgs *M, int *n_blocks, bool partitioned) {
  const int64_t zc = op->ctx->opt_cg_march;
  int nbt = 0;
  if (zc < 2 || !canon_tile_geometry(op, &M->T, &nbt, partitioned)) return false;
  if ((int64_t)sizeof(double) * 3 * (kTileRun + 2 * M->T.a) > 60 * 1024) return false;
  if (partitioned && op->n_rows % M->T.b != 0) return false;  // (whole planes only)
  const int64_t planes = (op->n_rows + M->T.b - 1) / M->T.b;
  // (option cg_march is the chunk of a large lattice; a smaller one marches fewer planes per block, so that the grid
  //  still holds ~2 blocks per resident slot: 192^3 with 8-plane chunks is 864 blocks for 1 024 slots -- 124 us per CG
  //  iteration against 111 with 4-plane chunks)
  //  (option cg_march_fill: the block count aimed at; 0 = cg_march whatever the size)
  const int64_t want = op->ctx->opt_cg_march_fill;
  const int64_t fill = want > 0 ? planes * M->T.tiles_per_plane / want : zc;
  M->zc_planes = (int)std::min<int64_t>(std::min<int64_t>(zc, std::max<int64_t>(2, fill)), planes);
  M->alternate = (int)(op->ctx->opt_cg_march_alternate != 0);
  M->apply_begin = partitioned ? (int)op->int_plane0 : 0;
  M->apply_end = partitioned ? (int)op->int_plane1 : (int)planes;
  M->T.plane0 = 0, M->T.plane_end = (int)planes;
  const int64_t chunks = (planes + M->zc_planes - 1) / M->zc_planes;
  *n_blocks = (int)(chunks * M->T.tiles_per_plane);
  return true;
}

template <bool DOT>
static bool launch_tile(const RangeLaunch &L, SellArgs A) {
  const storm_hip_op *op = L.op;
  const int nb = L.nb;
  hipStream_t st = op->ctx->stream;
  const Scal alpha = L.alpha, beta = L.beta;
  const double *x = L.x;
  double *y = L.y;
  const DotArgs dot = L.dot;
  const int *done = L.done;
  hipEvent_t ev0 = L.ev0, ev1 = L.ev1;
  const CgFuseArgs *cg_fuse = L.cg_fuse;
  CanonTileArgs T;
  int tile_blocks = 0;
  const bool interior_list = L.slice_list != nullptr && L.slice_list == op->d_interior;
  IpcSendArgs S{};
  if (L.fused != nullptr && interior_list) S.w = L.fused->w, S.sp = L.fused->sp;  // the interior launch sends
  if (!(op->pair == 2 && !boundary_of_mixed(L) && (L.slice_list == nullptr || interior_list) && !L.accumulate &&
        canon_tile_geometry(op, &T, &tile_blocks, interior_list) && tile_blocks + S.sp.n_blocks == nb))
    return false;
  const int tz = canon_tile_planes(op);
  const int hl_need = (tz * T.a + kBlock - 1) / kBlock;
  const size_t lds = sizeof(double) * (size_t)tz * (size_t)(kTileRun + 2 * T.a) + (size_t)op->ctx->opt_spmv_tile_lds_pad;
  const bool wload = DOT && dot.w != nullptr && dot.w != x;
#define TILE_GO3(WL_, TZ_, HL_)                                                                                              \
  hipExtLaunchKernelGGL((spmv_canon_tile_kernel<DOT, WL_, TZ_, HL_>), dim3(nb), dim3(kBlock), lds, st, ev0, ev1, 0, A, T, alpha, \
                        beta, x, y, dot, done, S, CgFuseArgs{})
#define TILE_GO2(TZ_, HL_)                                                                                                   \
  do {                                                                                                                       \
    if (cg_fuse != nullptr) {                                                                                                \
      if constexpr (DOT)                                                                                                     \
        hipExtLaunchKernelGGL((spmv_canon_tile_kernel<true, false, TZ_, HL_, true>), dim3(nb), dim3(kBlock), lds, st, ev0, ev1, \
                              0, A, T, alpha, beta, x, y, dot, done, S, *cg_fuse);                                            \
    } else if (wload) TILE_GO3(true, TZ_, HL_);                                                                              \
    else TILE_GO3(false, TZ_, HL_);                                                                                          \
  } while (0)
#define TILE_GO(TZ_)                       \
  do {                                     \
    if (hl_need <= 1) TILE_GO2(TZ_, 1);     \
    else if (hl_need <= 2) TILE_GO2(TZ_, 2); \
    else if (hl_need <= 4) TILE_GO2(TZ_, 4); \
    else TILE_GO2(TZ_, 8);                  \
  } while (0)
  if (tz == 2) TILE_GO(2);
  else TILE_GO(4);
#undef TILE_GO
#undef TILE_GO2
#undef TILE_GO3
  return true;
}

bool spmv_tile_run(const RangeLaunch &L) {
  if (L.op->pair != 2) return false;
  int width = 0;
  const SellArgs A = paired_args(L, &width);
  return L.want_dot ? launch_tile<true>(L, A) : launch_tile<false>(L, A);
}

int spmv_march_run(const storm_hip_op *op, const MarchArgs &M, int n_blocks, Scal alpha, Scal beta, const double *x, double *y,
                   const DotArgs &dot, const int *done, const CgFuseArgs &cgf, const IpcSendArgs &S) {
  storm_hip_ctx *c = op->ctx;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (c->opt_profile_spmv != 0) {
    while (c->prof_events.size() < c->prof_used + 2) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreate(&ev));
      c->prof_events.push_back(ev);
    }
    ev0 = c->prof_events[c->prof_used], ev1 = c->prof_events[c->prof_used + 1];
    c->prof_used += 2;
  }
  SellArgs A{op->d_pack, op->d_slice_off, op->n_rows, op->uniform_width, 0, op->d_dict, op->dict_size, op->d_offs, op->offs_size, 0};
  const size_t lds = sizeof(double) * 3 * (size_t)(kTileRun + 2 * M.T.a);
  const int nb = n_blocks + S.sp.n_blocks;
#define MARCH_GO(HLP_)                                                                                                \
  hipExtLaunchKernelGGL((cg_step_march_kernel<HLP_>), dim3(nb), dim3(kBlock), lds, c->stream, ev0, ev1, 0, A, M, alpha, \
                        beta, x, y, dot, done, cgf, S)
  STORM_REQUIRE(cgf.x != nullptr, "spmv: the marching step updates x");
  if (M.T.a <= kBlock) MARCH_GO(1);
  else MARCH_GO(2);
#undef MARCH_GO
  HIP_TRY(hipGetLastError());
  return STORM_HIP_OK;
}

}  // namespace storm
