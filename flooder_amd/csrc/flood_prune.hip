// flood_prune.hip - which samples can still raise a face maximum? (gfx950)
//
// The filtration value of a face is the MAXIMUM over its samples of the nearest-neighbour distance d, and d
// is 1-Lipschitz: d(p) <= d(q) + |p - q|.  After an exact sweep of a coarse subset C of the sample rows,
//     ub(r)  = min_{q in knn(r)} ( d(q) + |p_r - p_q| )          (q: coarse rows near r in weight space)
//     thr(r) = min_{faces f containing r} max_{q in C on f} d(q)  (what every face of r has reached already)
// and a fine row with ub(r) <= thr(r) cannot change any face maximum: it is never swept.  The surviving
// rows of each simplex are written, in sweep order, to its segment of `row_list` (count in row_cnt).
// One workgroup per simplex.  Margins: ub is inflated by 1e-6 relative + 1e-30 absolute.
#include "flood_common.hpp"

using namespace flooder;

namespace {

constexpr int PRUNE_THREADS = 256;
constexpr int MAX_COARSE = 1536;  // coarse rows per simplex held in LDS
constexpr int MAX_FACES = 32;

template <int DIM>
__global__ __launch_bounds__(PRUNE_THREADS) void prune_kernel(
    const uint32_t* __restrict__ d2, int ld_d2, const float* __restrict__ verts,
    const float* __restrict__ weights, int k1, int R, int Rc, const int32_t* __restrict__ knn, int K,
    const uint32_t* __restrict__ memb, const int32_t* __restrict__ cface_ptr,
    const int32_t* __restrict__ cface_rows, int n_faces, int32_t* __restrict__ row_list,
    int32_t* __restrict__ row_cnt, int list_stride) {
  __shared__ float s_d[MAX_COARSE];         // d at the coarse rows
  __shared__ float s_p[MAX_COARSE][DIM];    // their positions
  __shared__ float s_lb[MAX_FACES];         // per face: max d over its coarse rows
  __shared__ int s_wave_cnt[4];
  __shared__ int s_base;
  const int64_t s = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* vs = verts + s * (int64_t)k1 * DIM;
  for (int q = tid; q < Rc; q += PRUNE_THREADS) {
    s_d[q] = __builtin_sqrtf(__uint_as_float(d2[s * (int64_t)ld_d2 + q]));
    float p[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) p[k] = 0.f;
    for (int j = 0; j < k1; ++j) {
      const float w = weights[(int64_t)q * k1 + j];
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(w, vs[j * DIM + k], p[k]);
    }
#pragma unroll
    for (int k = 0; k < DIM; ++k) s_p[q][k] = p[k];
  }
  if (tid == 0) s_base = 0;
  __syncthreads();
  // per-face maximum over the face's coarse rows (one wave per face, round robin)
  for (int f = wv; f < n_faces; f += 4) {
    float m = 0.f;
    for (int i = cface_ptr[f] + lane; i < cface_ptr[f + 1]; i += 64) m = __builtin_fmaxf(m, s_d[cface_rows[i]]);
    m = wave_max_f32(m);
    if (lane == 0) s_lb[f] = m;
  }
  __syncthreads();
  // fine rows Rc .. R-1, 256 at a time, order-preserving compaction
  for (int r0 = Rc; r0 < R; r0 += PRUNE_THREADS) {
    const int r = r0 + tid;
    bool need = false;
    if (r < R) {
      float p[DIM];
#pragma unroll
      for (int k = 0; k < DIM; ++k) p[k] = 0.f;
      for (int j = 0; j < k1; ++j) {
        const float w = weights[(int64_t)r * k1 + j];
#pragma unroll
        for (int k = 0; k < DIM; ++k) p[k] = __builtin_fmaf(w, vs[j * DIM + k], p[k]);
      }
      float ub = __builtin_inff();
      for (int k = 0; k < K; ++k) {
        const int q = knn[(int64_t)(r - Rc) * K + k];
        float dd = 0.f;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const float t = p[c] - s_p[q][c];
          dd = __builtin_fmaf(t, t, dd);
        }
        ub = __builtin_fminf(ub, s_d[q] + __builtin_sqrtf(dd));
      }
      ub = ub * 1.000001f + 1e-30f;
      float thr = __builtin_inff();
      uint32_t mb = memb[r];
      while (mb) {
        const int f = __builtin_ctz(mb);
        mb &= mb - 1;
        thr = __builtin_fminf(thr, s_lb[f]);
      }
      need = !(ub <= thr);
    }
    const unsigned long long m = __ballot(need);
    if (lane == 0) s_wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wv; ++w) off += s_wave_cnt[w];
    if (need) {
      const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
      row_list[s * (int64_t)list_stride + off + rank] = r;
    }
    __syncthreads();
    if (tid == 0) s_base += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) row_cnt[s] = s_base;
}

template <int DIM>
struct PruneOp {
  static int run(const uint32_t* d2, int ld_d2, const float* verts, const float* weights, int k1, int R, int Rc,
                 const int32_t* knn, int K, const uint32_t* memb, const int32_t* cface_ptr,
                 const int32_t* cface_rows, int n_faces, int32_t* row_list, int32_t* row_cnt, int list_stride,
                 int64_t ns, hipStream_t st) {
    hipLaunchKernelGGL((prune_kernel<DIM>), dim3((unsigned)ns), dim3(PRUNE_THREADS), 0, st, d2, ld_d2, verts,
                       weights, k1, R, Rc, knn, K, memb, cface_ptr, cface_rows, n_faces, row_list, row_cnt,
                       list_stride);
    return check_launch("prune");
  }
};

}  // namespace

extern "C" int flooder_prune_rows_f32(const uint32_t* d2, int ld_d2, int dim, const float* verts,
                                      const float* weights, int k1, int R, int Rc, const int32_t* knn, int K,
                                      const uint32_t* memb, const int32_t* cface_ptr, const int32_t* cface_rows,
                                      int n_faces, int64_t n_simplices, int32_t* row_list, int32_t* row_cnt,
                                      int list_stride, void* stream) {
  if (n_simplices == 0) return FLOODER_OK;
  if (!d2 || !verts || !weights || !knn || !memb || !cface_ptr || !cface_rows || !row_list || !row_cnt ||
      Rc < 1 || Rc > R || Rc > MAX_COARSE || K < 1 || n_faces < 1 || n_faces > MAX_FACES ||
      list_stride < R - Rc || n_simplices > 0x7fffffffLL || k1 < 1 || k1 > FLOODER_MAX_VERTS)
    return fail(FLOODER_E_ARG, "flooder_prune_rows_f32: bad argument");
  return dispatch_dim<PruneOp>(dim, d2, ld_d2, verts, weights, k1, R, Rc, knn, K, memb, cface_ptr, cface_rows,
                               n_faces, row_list, row_cnt, list_stride, n_simplices, (hipStream_t)stream);
}
