// m x m element-wise finalisation kernels (negligible cost next to the n x m x m contractions).
#include "kernels.h"

namespace gprhip {

// dst = base + sum_z slices[z] on upper tiles (row tile <= column tile); dst is the mp x mp square (0 elsewhere) or,
// packed, the upper tiles alone (packed_upper_off: the layout of the exchange buffers).
// Each thread owns 16 bytes of a row (2 doubles / 4 floats: one vector load per slice), four slices in flight.
template <typename TS>
__global__ __launch_bounds__(256) void sum_slices_kernel(const double* __restrict__ base,
                                                         const TS* __restrict__ slices, int nslices,
                                                         int nslices_diag, int64_t stride, int mp, int packed,
                                                         double* __restrict__ dst) {
  constexpr int V = 16 / (int)sizeof(TS);
  typedef TS vec_t __attribute__((ext_vector_type(V)));
  const int c = (blockIdx.x * 256 + threadIdx.x) * V;  // mp is a multiple of 128: a vector never straddles a tile
  const int r = blockIdx.y;
  if (c >= mp) return;
  const int64_t off = (int64_t)r * mp + c;
  const bool upper = r / TILE <= c / TILE;
  double acc[V];
#pragma unroll
  for (int v = 0; v < V; ++v) acc[v] = 0.0;
  if (upper) {
    if (base) {
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = base[off + v];
    }
    const int nz = (r / TILE == c / TILE) ? nslices_diag : nslices;  // diagonal tiles of a SYRK launch: fewer slices
    const TS* sp = slices + off;
    int z = 0;
    // sixteen slices' loads in flight, then four (round 6: with four alone a 64-slice sum was sixteen memory round trips long);
    // the additions keep the order z = 0, 1, 2, ...
    for (; z + 16 <= nz; z += 16) {
      vec_t a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = *reinterpret_cast<const vec_t*>(sp + (int64_t)(z + u) * stride);
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = acc[v] + (double)a[u][v];
    }
    for (; z + 4 <= nz; z += 4) {
      const vec_t a0 = *reinterpret_cast<const vec_t*>(sp + (int64_t)z * stride);
      const vec_t a1 = *reinterpret_cast<const vec_t*>(sp + (int64_t)(z + 1) * stride);
      const vec_t a2 = *reinterpret_cast<const vec_t*>(sp + (int64_t)(z + 2) * stride);
      const vec_t a3 = *reinterpret_cast<const vec_t*>(sp + (int64_t)(z + 3) * stride);
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] = (((acc[v] + (double)a0[v]) + (double)a1[v]) + (double)a2[v]) + (double)a3[v];
    }
    for (; z < nz; ++z) {
      const vec_t a0 = *reinterpret_cast<const vec_t*>(sp + (int64_t)z * stride);
#pragma unroll
      for (int v = 0; v < V; ++v) acc[v] += (double)a0[v];
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) {
    if (!packed) dst[off + v] = acc[v];
    else if (upper) dst[packed_upper_off(r, c + v)] = acc[v];
  }
}

template <typename TS>
void launch_sum_slices(const double* base, const TS* slices, int nslices, int64_t stride, int mp,
                       double* dst, hipStream_t s, int packed, int nslices_diag) {
  constexpr int V = 16 / (int)sizeof(TS);
  hipLaunchKernelGGL(sum_slices_kernel<TS>, dim3((mp / V + 255) / 256, mp), dim3(256), 0, s, base, slices,
                     nslices, nslices_diag > 0 ? nslices_diag : nslices, stride, mp, packed, dst);
  GPR_HIP(hipGetLastError());
}
template void launch_sum_slices<double>(const double*, const double*, int, int64_t, int, double*, hipStream_t, int, int);
template void launch_sum_slices<float>(const double*, const float*, int, int64_t, int, double*, hipStream_t, int, int);

// dst[b][r][c] = sum_z slices[z][b][r][c] over a rows x cols rectangle (leading dimension ld, batch stride bs;
// slices and dst share offsets): combines the split-K partial products of small GEMM launches in a fixed order.
__global__ __launch_bounds__(256) void sum_slices_rect_kernel(const double* __restrict__ slices, int nslices,
                                                              int64_t stride, int rows, int cols, int64_t ld,
                                                              int64_t bs, double* __restrict__ dst) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= cols) return;
  const int64_t off = (int64_t)blockIdx.z * bs + (int64_t)r * ld + c;
  double acc = 0.0;
  for (int z = 0; z < nslices; ++z) acc += slices[(int64_t)z * stride + off];
  dst[off] = acc;
}

void launch_sum_slices_rect(const double* slices, int nslices, int64_t stride, int rows, int cols, int64_t ld,
                            int nbatch, int64_t bs, double* dst, hipStream_t s) {
  hipLaunchKernelGGL(sum_slices_rect_kernel, dim3((cols + 255) / 256, rows, nbatch), dim3(256), 0, s, slices,
                     nslices, stride, rows, cols, ld, bs, dst);
  GPR_HIP(hipGetLastError());
}

__global__ void to_float_kernel(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

void launch_to_float(const double* src, float* dst, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
  GPR_HIP(hipGetLastError());
}

// Whitened W:  W = T - t t^T - U_mat^T diag(v) U_mat  (lib/fitc_gp.ml:1196-1203, T = K_m^-1 - B^-1 :1040-1041)
// equals U^-1 W~ U^-T with  W~ = I - B~^-1 - t~ t~^T - V^T diag(v) V,  B~ = I + V^T diag(is) V, t~ = U t.
// Written as a full symmetric matrix (the inputs' upper triangles are the computed part; the rest is mirrored).
__global__ __launch_bounds__(256) void build_w_kernel(const double* __restrict__ binv,
                                                      const double* __restrict__ t,
                                                      const double* __restrict__ G, int mp,
                                                      double* __restrict__ W) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (c >= mp) return;
  int rr = r, cc = c;
  if (r > c) {  // symmetric inputs are valid in the upper triangle only: mirror
    rr = c;
    cc = r;
  }
  const int64_t off = (int64_t)rr * mp + cc;  // G: packed upper tiles (the exchange-2 buffer)
  W[(int64_t)r * mp + c] = (r == c ? 1.0 : 0.0) - binv[off] - t[rr] * t[cc] - G[packed_upper_off(rr, cc)];
}

void launch_build_w(const double* binv, const double* t, const double* G, int mp, double* W,
                    hipStream_t s) {
  hipLaunchKernelGGL(build_w_kernel, dim3((mp + 255) / 256, mp), dim3(256), 0, s, binv, t, G, mp, W);
  GPR_HIP(hipGetLastError());
}

// Per column c (thread) over a slab of rows r:
//   part[slab][0][c]   = sum_r W_rc K_rc                      -> tr(W K_m)     (`Factor, Mat.symm2_trace)
//   part[slab][1][c]   = sum_r W_rc K_rc |z_r - z_c|^2        -> Log_ell `Dense trace / inv_ell2
//   part[slab][2+k][c] = sum_r W_rc K_rc (z_kr - z_kc)        -> `Sparse_rows trace / (2*scale)
// W and K_m are full symmetric here, so the reference's upper-triangle bookkeeping
// (lib/utils.ml:196-220: 2*sum_{r != c} + diagonal) becomes a plain column sum.
// rows of W .* K_m one block walks: short slabs, so that the m x m pass spreads over the whole chip
// (round 6: 8 rows up to 1024 inducing points -- one batch of loads per thread, four times the workgroups; the partial
// buffer, km_rows x mp doubles per slab, stays below 11 MB)
int km_slab_rows(int m) { return m <= 1024 ? 8 : 32; }

template <int DT>
__global__ __launch_bounds__(256) void km_traces_kernel(const double* __restrict__ W,
                                                        const double* __restrict__ km,
                                                        const double* __restrict__ Z, int m, int mp,
                                                        int d, double* __restrict__ part, int slab) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= mp) return;
  const bool live = c < m;
  double z[DT], g[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    z[k] = (k < d && live) ? Z[(int64_t)c * d + k] : 0.0;
    g[k] = 0.0;
  }
  double s0 = 0.0, s1 = 0.0;
  const int r0 = blockIdx.y * slab, r1 = min(m, r0 + slab);
  if (live) {
    // eight rows' W and K_m entries are loaded before any is used (round 6): one load round trip per eight rows instead of
    // one per row -- with 32 rows per thread the kernel was 32 dependent round trips long (32 us at every m), now four.
    // The sums take the rows in the same order as before.
    for (int rb = r0; rb < r1; rb += 8) {
      double wv[8], kv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(rb + u, r1 - 1);
        wv[u] = W[(int64_t)r * mp + c];  // upper tiles of km are the valid ones; km is written full by cov_upper, W full by build_w
        kv[u] = km[(int64_t)r * mp + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (rb + u < r1) {
          const double wk = wv[u] * kv[u];
          const double* zr = Z + (int64_t)(rb + u) * d;
          double dist = 0.0;
#pragma unroll
          for (int k = 0; k < DT; ++k) {
            if (k < d) {
              const double df = zr[k] - z[k];
              dist += df * df;
              g[k] += wk * df;
            }
          }
          s0 += wk;
          s1 += wk * dist;
        }
      }
    }
  }
  double* p = part + (int64_t)blockIdx.y * (d + 2) * mp;
  p[c] = s0;
  p[(int64_t)mp + c] = s1;
#pragma unroll
  for (int k = 0; k < DT; ++k)
    if (k < d) p[(int64_t)(2 + k) * mp + c] = g[k];
}

// d > 64: the per-dimension accumulators run in passes of 32 dimensions (blockIdx.z); |z_r - z_c|^2 of the Log_ell
// trace is recovered from K_m itself (pass 0).  No multiscales on this path.
__global__ __launch_bounds__(256) void km_traces_wide_kernel(const double* __restrict__ W,
                                                             const double* __restrict__ km,
                                                             const double* __restrict__ Z, int m, int mp, int d,
                                                             double log_sf2, double inv_ell2_05,
                                                             double* __restrict__ part, int slab) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= mp) return;
  const bool live = c < m;
  const int dim0 = blockIdx.z * 32, nd = min(32, d - dim0);
  double z[32], g[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    z[k] = (k < nd && live) ? Z[(int64_t)c * d + dim0 + k] : 0.0;
    g[k] = 0.0;
  }
  double s0 = 0.0, s1 = 0.0;
  const int r0 = blockIdx.y * slab, r1 = min(m, r0 + slab);
  if (live) {
    for (int r = r0; r < r1; ++r) {
      const double kv = km[(int64_t)r * mp + c];
      const double wk = W[(int64_t)r * mp + c] * kv;
      const double* zr = Z + (int64_t)r * d + dim0;
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k < nd) g[k] += wk * (zr[k] - z[k]);
      s0 += wk;
      if (r != c && kv > 0.0) s1 += wk * ((log(kv) - log_sf2) / inv_ell2_05);
    }
  }
  double* p = part + (int64_t)blockIdx.y * (d + 2) * mp;
  if (blockIdx.z == 0) {
    p[c] = s0;
    p[(int64_t)mp + c] = s1;
  }
#pragma unroll
  for (int k = 0; k < 32; ++k)
    if (k < nd) p[(int64_t)(2 + dim0 + k) * mp + c] = g[k];
}

template <int DT>
__global__ __launch_bounds__(256) void km_traces_ms_kernel(const double* __restrict__ W,
                                                           const double* __restrict__ km,
                                                           const double* __restrict__ Z,
                                                           const double* __restrict__ ms, int m, int mp, int d,
                                                           double* __restrict__ part, int slab) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= mp) return;
  const bool live = c < m;
  double z[DT], msc[DT], g[DT], gm[DT];
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    z[k] = (k < d && live) ? Z[(int64_t)c * d + k] : 0.0;
    msc[k] = (k < d && live) ? ms[(int64_t)c * d + k] : 1.0;
    g[k] = 0.0;
    gm[k] = 0.0;
  }
  double s0 = 0.0;
  const int r0 = blockIdx.y * slab, r1 = min(m, r0 + slab);
  if (live) {
    for (int rb = r0; rb < r1; rb += 8) {  // (loads of eight rows ahead of their use, as in km_traces_kernel)
      double wv[8], kv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(rb + u, r1 - 1);
        wv[u] = W[(int64_t)r * mp + c];
        kv[u] = km[(int64_t)r * mp + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u;
        if (r < r1) {
          const double wk = wv[u] * kv[u];
          s0 += wk;
          if (r != c) {
            const double* zr = Z + (int64_t)r * d;
            const double* msr = ms + (int64_t)r * d;
#pragma unroll
            for (int k = 0; k < DT; ++k) {
              if (k < d) {
                const double iscale = 1.0 / ((msr[k] + msc[k]) - 1.0);
                const double sdiff = (zr[k] - z[k]) * iscale;
                g[k] += wk * sdiff;
                gm[k] += wk * (iscale - sdiff * sdiff);
              }
            }
          }
        }
      }
    }
  }
  double* p = part + (int64_t)blockIdx.y * (2 * d + 2) * mp;
  p[c] = s0;
  p[(int64_t)mp + c] = 0.0;
#pragma unroll
  for (int k = 0; k < DT; ++k) {
    if (k < d) {
      p[(int64_t)(2 + k) * mp + c] = g[k];
      p[(int64_t)(2 + d + k) * mp + c] = gm[k];
    }
  }
}

void launch_km_traces_ms(const double* W, const double* km, const double* Z, const double* ms, int m, int mp,
                         int d, double* part, hipStream_t s) {
  if (d > 64) {
    set_error("gprhip: Cov_se_fat multiscales support kernel-space dimension d <= 64");
    throw HipFail{ST_BAD_ARG};
  }
  const int slab = km_slab_rows(m);
  dim3 grid((mp + 255) / 256, (m + slab - 1) / slab);
  auto go = [&](auto dt) {
    hipLaunchKernelGGL((km_traces_ms_kernel<decltype(dt)::value>), grid, dim3(256), 0, s, W, km, Z, ms, m, mp,
                       d, part, slab);
  };
  if (d <= 4) go(std::integral_constant<int, 4>{});
  else if (d <= 8) go(std::integral_constant<int, 8>{});
  else if (d <= 16) go(std::integral_constant<int, 16>{});
  else if (d <= 32) go(std::integral_constant<int, 32>{});
  else go(std::integral_constant<int, 64>{});
  GPR_HIP(hipGetLastError());
}

void launch_km_traces(const double* W, const double* km, const double* Z, int m, int mp, int d,
                      double* part, const CovParams& cp, hipStream_t s) {
  const int slab = km_slab_rows(m);
  dim3 grid((mp + 255) / 256, (m + slab - 1) / slab);
  if (d > 64) {
    grid.z = (d + 31) / 32;
    hipLaunchKernelGGL(km_traces_wide_kernel, grid, dim3(256), 0, s, W, km, Z, m, mp, d, cp.log_sf2, cp.inv_ell2_05,
                       part, slab);
    GPR_HIP(hipGetLastError());
    return;
  }
  auto go = [&](auto dt) {
    hipLaunchKernelGGL((km_traces_kernel<decltype(dt)::value>), grid, dim3(256), 0, s, W, km, Z, m, mp,
                       d, part, slab);
  };
  if (d <= 4) go(std::integral_constant<int, 4>{});
  else if (d <= 8) go(std::integral_constant<int, 8>{});
  else if (d <= 16) go(std::integral_constant<int, 16>{});
  else if (d <= 32) go(std::integral_constant<int, 32>{});
  else go(std::integral_constant<int, 64>{});
  GPR_HIP(hipGetLastError());
}

// The results' way home.  Up to three blocks of doubles (result block, exchange-1 tail, exchange-2 buffer from its column
// block on) written into the pinned host mirror by a kernel instead of by hipMemcpyAsync: behind a kernel a device-to-host
// copy of more than 32 KB (or a second and third one) starts 17-18 us late on this runtime (profiles/r06_timeline_*), a
// kernel starts at once, and a few workgroups storing over the host link move these 40-600 KB as fast as the copy engine.
__global__ __launch_bounds__(256) void ship_kernel(ShipArgs a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const double* __restrict__ src = a.src[b];
    double* __restrict__ dst = a.dst[b];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n[b]; i += stride)
      __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  }
}

void launch_ship(const ShipArgs& a, hipStream_t s) {
  const int64_t total = a.n[0] + a.n[1] + a.n[2];
  if (total <= 0) return;
  const int grid = (int)std::min<int64_t>(32, (std::max(a.n[0], std::max(a.n[1], a.n[2])) + 2047) / 2048);
  hipLaunchKernelGGL(ship_kernel, dim3(std::max(grid, 1)), dim3(256), 0, s, a);
  GPR_HIP(hipGetLastError());
}

}  // namespace gprhip
