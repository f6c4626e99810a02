// flood_params.hip - the parameter-block forms of the long entry points (include/flooder_hip.h, "parameter blocks"):
// one struct, bound by field name, instead of up to 34 positional arguments.  Each forwards to the positional function
// of the same launch; nothing here touches the device.
#include "../../include/flooder_hip.h"
#include "flood_common.hpp"

#include <cstddef>
#include <cstring>

namespace {
using namespace flooder;

// The caller's struct (its `size` bytes) laid over a zeroed struct of OUR size: fields the caller does not have read as
// NULL / 0, a caller that has fields we do not know is refused.
template <typename T>
int take(const T* p, T& out, const char* who) {
  if (!p) return fail(FLOODER_E_ARG, who);
  if (p->abi != FLOODER_PARAMS_ABI || p->size < 2 * sizeof(uint32_t) || p->size > sizeof(T)) return fail(FLOODER_E_ARG, who);
  std::memset(&out, 0, sizeof(T));
  std::memcpy(&out, p, p->size);
  return 0;
}
}  // namespace

extern "C" int flooder_fused_witness(const flooder_fused_sweep_t* p, void* stream) {
  flooder_fused_sweep_t a;
  if (int rc = take(p, a, "flooder_fused_witness: bad parameter block (abi / size)")) return rc;
  return flooder::sweep_witness(a.pts_sorted, a.n_pts, a.dim, a.nodes, a.verts, a.weights, a.k1, a.R, a.n_simplices,
                                a.coarse_rows, a.n_coarse, a.parents, a.wit_queue, a.d2_scratch, a.memb, a.n_faces,
                                a.face_bits, a.face_slot, a.flag_list, a.flag_count, a.flag_key, a.flag_hist, a.top,
                                a.top_list, a.top_count, a.simplex_weight, a.wit_item_list, a.plane_scratch, a.wit_stats,
                                a.density_grid, stream);
}

extern "C" int flooder_fused_cell(const flooder_fused_sweep_t* p, void* stream) {
  flooder_fused_sweep_t a;
  if (int rc = take(p, a, "flooder_fused_cell: bad parameter block (abi / size)")) return rc;
  return flooder_sweep_cell_faces_f32(a.pts_sorted, a.n_pts, a.dim, a.nodes, a.verts, a.weights, a.k1, a.R, a.n_simplices,
                                      a.alpha, a.cell_queue, a.d2_scratch, a.memb, a.n_faces, a.face_bits, a.face_slot,
                                      a.flag_list, a.flag_count, a.flag_key, a.flag_hist, a.top, a.top_list, a.top_count,
                                      a.defer_list, a.defer_c, a.defer_ctl, a.simplex_weight, a.light_list, a.heavy_list,
                                      a.plane_scratch, a.density_grid, a.cloud_box, a.cell_stats, stream);
}

extern "C" int flooder_fused_finish(const flooder_fused_sweep_t* p, void* stream) {
  flooder_fused_sweep_t a;
  if (int rc = take(p, a, "flooder_fused_finish: bad parameter block (abi / size)")) return rc;
  return flooder_finish_faces_f32(a.pts_sorted, a.n_pts, a.dim, a.nodes, a.verts, a.weights, a.k1, a.R, a.n_simplices,
                                  a.flag_list, a.flag_count, a.flag_key, a.flag_hist, a.flag_sorted, a.finish_ctl, a.top,
                                  a.top_list, a.probed, a.d2_scratch, a.memb, a.n_faces, a.face_bits, a.face_slot,
                                  a.hard_scratch, a.hard_cap, a.finish_stats, stream);
}

extern "C" int flooder_sorted_faces(const flooder_sorted_sweep_t* p, void* stream) {
  flooder_sorted_sweep_t a;
  if (int rc = take(p, a, "flooder_sorted_faces: bad parameter block (abi / size)")) return rc;
  return flooder_sweep_bvh_sorted_faces_f32(a.pts_sorted, a.n_pts, a.dim, a.nodes, a.verts, a.weights, a.k1, a.R,
                                            a.n_simplices, a.sample_order, a.queue, a.memb, a.n_faces, a.face_bits,
                                            a.face_slot, a.stats, stream);
}

extern "C" int flooder_sorted_minima(const flooder_sorted_sweep_t* p, void* stream) {
  flooder_sorted_sweep_t a;
  if (int rc = take(p, a, "flooder_sorted_minima: bad parameter block (abi / size)")) return rc;
  if (a.shard_world > 1)
    return flooder_sweep_bvh_sorted_shard_f32(a.pts_sorted, a.n_pts, a.dim, a.nodes, a.verts, a.weights, a.k1, a.R,
                                              a.n_simplices, a.sample_order, a.shard_rank, a.shard_world, a.queue,
                                              a.out_d2, a.stats, stream);
  return flooder_sweep_bvh_sorted_f32(a.pts_sorted, a.n_pts, a.dim, a.nodes, a.verts, a.weights, a.k1, a.R, a.n_simplices,
                                      a.sample_order, a.queue, a.out_d2, a.stats, stream);
}

extern "C" int flooder_fps_batched(const flooder_fps_batched_t* p, void* stream) {
  flooder_fps_batched_t a;
  if (int rc = take(p, a, "flooder_fps_batched: bad parameter block (abi / size)")) return rc;
  return flooder_fps_batched_f32(a.pts, a.n_pts, a.dim, a.ld, a.pts_sorted, a.order, a.n_lms, a.start, a.out_idx, a.minsq,
                                 a.bucket_box, a.bucket_keys, a.bucket_coord, a.work_best, a.work_rec, a.work_ctr,
                                 a.launches_out, stream);
}
