/*
 * flooder_hip.h - C ABI of the MI355X (gfx950) coverage-sweep library, libflooder_hip.so.
 *
 * The reference (plus-rkwitt/flooder) is a pure-Python package; the seam its hot path crosses is the
 * pair of Triton kernel wrappers in flooder/triton_kernels.py and their call sites in
 * flooder/core.py:200-226.  Each entry point below names the reference interface it replaces.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless it says "host"; memory is owned by the caller, the
 *     library never allocates, frees or synchronises; work is enqueued on `stream` (a hipStream_t,
 *     passed as void*; NULL = the default stream) and the pointers must stay valid until that work
 *     has completed;
 *   - return value 0 = success, negative = error (FLOODER_E_*); flooder_last_error() gives the text
 *     for the calling thread; nothing throws;
 *   - points / candidates are float32, row-major, `ld` floats per row (ld >= dim);
 *   - squared distances travel as the uint32 bit pattern of a non-negative float32 ("d2 bits"):
 *     for non-negative floats unsigned integer order equals numeric order, so min/max over them are
 *     exact integer atomics.  +inf (0x7f800000) means "no candidate seen" (triton_kernels.py:70).
 */
#ifndef FLOODER_HIP_H
#define FLOODER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLOODER_ABI_VERSION 1

#define FLOODER_OK 0
#define FLOODER_E_ARG (-1)    /* bad argument (null pointer, unsupported dim, ...) */
#define FLOODER_E_LAUNCH (-2) /* HIP reported an error at launch                    */
#define FLOODER_E_DEVICE (-3) /* no gfx950 device / wrong architecture              */

#define FLOODER_MAX_DIM 8      /* ambient dimensions 1..8 are compiled              */
#define FLOODER_MAX_VERTS 9    /* vertices per simplex: max_dimension + 1 <= 9      */
#define FLOODER_CAND_ALIGN 8   /* each simplex's candidate list is padded to x8     */
#define FLOODER_SWEEP_CHUNK 2048 /* candidates per sweep work item                   */
#define FLOODER_TILE_SAMPLES 512 /* samples per sweep work item (64 lanes x 8)       */
#define FLOODER_BVH_LEAF 16      /* points per leaf of the point hierarchy            */
#define FLOODER_BVH_FANOUT 64    /* children per inner node (one per lane)            */
#define FLOODER_BVH_MAX_LEVELS 6 /* 16 * 64^5 points                                  */
#define FLOODER_QUEUE_SHARDS 16    /* heads of a sharded work queue ...                        */
#define FLOODER_QUEUE_WORDS 512    /* ... and the zeroed int32 words one queue takes (heads 128 B apart) */
#define FLOODER_BBOX_BLOCKS 1024  /* partial results of flooder_bbox_f32               */
#define FLOODER_WIT_MAX_COARSE 256  /* coarse samples per simplex of flooder_sweep_witness_f32 (4 per lane) */
#define FLOODER_WIT_MAX_ROWS 8192   /* samples per simplex it takes at most                                  */

int flooder_abi_version(void);
const char* flooder_last_error(void);

/* Name of the GPU architecture of `device` copied into buf (e.g. "gfx950:sramecc+:xnack-"). */
int flooder_device_arch(int device, char* buf, int buflen);

/* Tuning switches (process-wide); none of them changes a result bit.
 *   "sweep_variant": 0 = packed-fp32 inner loop (default), 1 = plain fp32 inner loop (ball sweep);
 *   "bvh_ks": samples per lane of the tree sweep (0 = auto: 1 for R <= 64, else 2);
 *   "bvh_subs": sub-tiles a tree-sweep item may be split into (default 16);
 *   "bvh_grid": persistent workgroups of the tree sweep (default 1024 = 4 per CU);
 *   "cell_grid": persistent workgroups of the cell sweep (default 768 = the 3 per CU that fit LDS);
 *   "bvh_refine_pct": threshold of the tree sweep's transposed refine in percent of its cost model (100;
 *                     the full sweep uses three times the value), "bvh_leaf_batch": leaves fetched per step
 *                     by the work-list tree sweep (1, or 4 through LDS);
 *   "curve": 1 (default) Hilbert, 0 Morton order of the cloud in flooder_morton_f32; "curve_bits": bits per axis;
 *   "cell_brute_max": kept points up to which a chunk is evaluated straight from the compacted list (160);
 *   "cell_retry_keep" / "cell_retry_pct": a chunk of the per-chunk launch gets its second cell size only if the first
 *                     try kept at most this many points (200) and this share of its open samples has a point within
 *                     twice the cell size (50 %; 0: whatever they are) - else its open tiles go to the finish;
 *   "cell_tries" / "cell_exh_tries": cell sizes tried per chunk (2) / attempts that may fall back to the exhaustive
 *                     evaluation (3); "finish_focus_pct", "finish_items_cap", "finish_budget", "finish_order", "finish_top":
 *                     focus rounds, tile splitting and hard tiles of flooder_finish_faces_f32; "finish_wide_points"
 *                     (4194304; 0 = never): clouds of at least this many points - a box tree of four levels and more -
 *                     run its per-wave passes with eight waves per workgroup sharing one staged tree top (6 waves per
 *                     SIMD at an 80-register cap instead of 4: a search in a deep tree is a longer chain of dependent
 *                     steps; cfg 5 finish 1.68 -> 1.47 ms, no gain at a million points); "fps_switch", "fps_rpl": see flooder_fps_indexed_f32;
 *   "cell_exh_dense": most kept points a dense chunk of the cell sweep evaluates exhaustively before it is
 *                     handed to the tree sweep (default 32768);
 *   "cell_split_launches": 1 (default) the light / heavy simplex lists of a long queue come from ONE launch, 2 from the
 *                     split + reorder pair of rounds 3 - 5 (same lists); "wit_surface_pct", "cell_surface_pct": see
 *                     flooder_cloud_kind. */
int flooder_set_option(const char* name, int value);

/* Row stride (floats) of a padded point / candidate row for ambient dimension `dim`:
 * 1,2 -> 2; 3,4 -> 4; 5..8 -> 8.  Rows in this layout are read with one vector load. */
int flooder_padded_dim(int dim);

/*
 * Ball membership count.  Replaces compute_mask + the per-row count of compute_mask_kernel
 * (flooder/triton_kernels.py:99-158, call site flooder/core.py:210-217): point j is a candidate of
 * simplex i iff sum_k (pts[j,k]-centers[i,k])^2 <= radii[i]^2 (squared, "<=", no sqrt; :137-148).
 * Instead of materialising the (n, m+512) bool mask it only counts, per simplex, over that simplex's
 * own slab [slab_lo[i], slab_hi[i]) of the cloud sorted along its widest axis (core.py:140-144,
 * 201-208; the per-simplex slab is a subset of the reference's per-batch slab and still contains the
 * whole ball, so the candidate sets are identical).
 *   counts[i] (int32) must be zeroed by the caller; it receives the number of candidates.
 */
int flooder_ball_count_f32(const float* pts, int64_t n_pts, int dim, int ld,
                           const float* centers /* n_simplices x dim */,
                           const float* radii /* n_simplices */,
                           const int64_t* slab_lo, const int64_t* slab_hi, int64_t n_simplices,
                           int32_t* counts, void* stream);

/*
 * Candidate compaction.  Replaces torch.nonzero(mask) + the index casts + the gather
 * y[w_idx] inside compute_filtration_kernel (core.py:218, triton_kernels.py:33,39,80-85): writes
 * the coordinates of simplex i's candidates to cand rows [cand_off[i], cand_off[i]+counts[i]) and
 * +inf rows up to cand_off[i+1] (the reference pads with indices that load +inf, :39), in
 * unspecified order.  cand has padded rows (flooder_padded_dim(dim) floats).
 *   cand_off: (n_simplices+1) int64, multiples of FLOODER_CAND_ALIGN;  cursor: n_simplices int32,
 *   zeroed by the caller (scratch).
 */
int flooder_ball_fill_f32(const float* pts, int64_t n_pts, int dim, int ld,
                          const float* centers, const float* radii,
                          const int64_t* slab_lo, const int64_t* slab_hi, int64_t n_simplices,
                          const int32_t* counts, const int64_t* cand_off, int32_t* cursor,
                          float* cand, void* stream);

/*
 * Coverage sweep.  Replaces compute_filtration / compute_filtration_kernel
 * (flooder/triton_kernels.py:12-96, call site core.py:219-226) fused with the sample generation
 * points_on_simplex = weights @ simplex_vertices (core.py:188): for simplex s and sample r
 *     p = sum_j weights[r,j] * verts[s,j,:]
 *     out_d2[s,r] = min(out_d2[s,r], min_w sum_k (p_k - cand[w,k])^2)      (direct differences, :36-44)
 * over the candidates w of s.  out_d2 (n_simplices x R uint32 d2 bits) must be pre-set to +inf bits
 * (flooder_fill_u32); it is combined with atomic min so several launches (point shards) or several
 * work items may target the same cell.  Work is split into items of (simplex, 512-sample tile,
 * 2048-candidate chunk); item_prefix (n_simplices+1 int64) = exclusive prefix sum of
 * tiles * ceil(count/2048) per simplex; queue = one int32 zeroed by the caller (work-queue head).
 */
int flooder_sweep_f32(const float* cand, const int64_t* cand_off, const int32_t* counts, int dim,
                      const float* verts /* n_simplices x k1 x dim */, const float* weights /* R x k1 */,
                      int k1, int R, int64_t n_simplices, const int64_t* item_prefix, int32_t* queue,
                      uint32_t* out_d2, void* stream);

/*
 * Face maxima.  Replaces the extraction at core.py:251-276: out_face[s,f] =
 * sqrt(max_{r in rows of face f} out_d2[s,r]) where face f's rows are
 * face_rows[face_ptr[f] .. face_ptr[f+1]) (CSR over the concatenated faces of every codimension,
 * grid mode) or all R rows (random mode: n_faces = 1, face_ptr = {0,R}, face_rows = 0..R-1).
 * out_dist (n_simplices x R float32, may be NULL) receives sqrt of every cell (the reference's
 * `distances` tensor, triton_kernels.py:44).
 */
int flooder_face_max_f32(const uint32_t* d2, int64_t n_simplices, int R, const int32_t* face_ptr,
                         const int32_t* face_rows, int n_faces, float* out_face, float* out_dist,
                         void* stream);

/*
 * ---- Hierarchically culled sweep (default device path) ---------------------------------------------
 * Replaces the reference's pruning chain as a whole - slab selection by searchsorted (core.py:201-208),
 * compute_mask (triton_kernels.py:161-223), torch.nonzero (core.py:218) and compute_filtration
 * (triton_kernels.py:48-96): instead of a bounding ball per simplex, an implicit bounding-box tree over
 * the Morton-sorted cloud is traversed per tile of samples, and only leaves that can still lower a
 * running minimum are evaluated.  The value computed is the exact minimum over ALL points (what the
 * reference's CPU branch returns, core.py:197-199), with the same direct-difference arithmetic.
 */

/* Bounding box of the cloud, on the device: box[0:dim] = min, box[8:8+dim] = max (box: 16 floats;
 * partial: FLOODER_BBOX_BLOCKS * 16 floats of scratch).  Replaces points.max(dim=0) - points.min(dim=0)
 * of core.py:140-142 without a host round trip. */
int flooder_bbox_f32(const float* pts, int64_t n_pts, int dim, int ld, float* box, float* partial, void* stream);

/* The two stages of flooder_bbox_f32 on their own, for a cloud that arrives in chunks (BASELINE.json configs[4]:
 * points streamed from pinned host memory; the reference's counterpart is its batch-bounded slabs, core.py:193-217):
 * flooder_bbox_chunk_f32 reduces the n_rows rows of one chunk with n_blocks blocks into partial[0 : 16 * n_blocks]
 * as soon as the chunk's copy has landed (enqueue it behind the copy's event), flooder_bbox_reduce_f32 folds the
 * n_partial rows of all chunks into box.  flooder_amd.PointIndex.from_host is the caller. */
int flooder_bbox_chunk_f32(const float* pts, int64_t n_rows, int dim, int ld, float* partial, int n_blocks,
                           void* stream);
int flooder_bbox_reduce_f32(const float* partial, int n_partial, int dim, float* box, void* stream);

/* 64-bit space-filling-curve codes of the points relative to `box` (DEVICE, layout of flooder_bbox_f32);
 * floor(63/dim) (max 21) bits per axis.  The caller sorts the cloud by these codes.  Hilbert codes by
 * default (option "curve" = 1; 0 = Morton / Z-order): consecutive points are neighbours in space, so the
 * 16-point leaves of the box tree are tight - the tree sweep of cfg 2 takes 3.3 ms instead of 6.5 ms. */
int flooder_morton_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* box, int64_t* codes,
                       void* stream);
/* The same, and `zero_words` int32 words at `zero_buf` are set to zero on the way: the density grid that the kernels of
 * flooder_index_rows_f32 - enqueued behind this one - add into, without a fill launch of its own. */
int flooder_morton_zero_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* box, int64_t* codes,
                            int32_t* zero_buf, int64_t zero_words, void* stream);

/* Number of low bits a curve code of flooder_morton_f32 occupies for ambient dimension dim (bits per axis x dim;
 * option "curve_bits": bits per axis, default 8 in 3D, 12 elsewhere, at most floor(63 / dim) and 21).  When
 * this is <= 32 the `codes` buffers of flooder_morton_f32 / flooder_index_sort hold n uint32 words (in the first
 * half of the n x int64 allocation), else n int64 words. */
int flooder_curve_key_bits(int dim);

/* Sort of the curve codes: order[j] = index of the point with the j-th smallest code (stable), codes_sorted = the
 * codes in that order.  rocprim::radix_sort_pairs over the low key_bits bits only (flooder_curve_key_bits).
 * tmp: flooder_index_sort_bytes(n_pts) bytes of device scratch.  Replaces torch.argsort of core.py:143. */
int64_t flooder_index_sort_bytes(int64_t n_pts);
int flooder_index_sort(const int64_t* codes, int64_t n_pts, int key_bits, int64_t* codes_sorted, int32_t* order,
                       void* tmp, int64_t tmp_bytes, void* stream);

/* The same sort (same library kernels, same stable order) without the seven fill launches the library call makes for
 * three 8-bit passes: its state - digit histograms, one look-back array and one block ticket per pass - is
 * `flooder_index_sort_state_words(n_pts, key_bits)` int32 words that the CALLER has zeroed on this stream before the
 * call (the curve-code kernel does it on its way: hand flooder_morton_zero_f32 a zero_buf that ends in them).  Keys
 * of at most 32 bits and fewer than 2^30 rows (state_words returns 0 otherwise: use flooder_index_sort); tmp: 8 * n_pts
 * bytes (flooder_index_sort_bytes is enough).  cfg 2's index build: 7 launches and ~17 us less.  The passes also run
 * in a block shape chosen by the cloud's size instead of the library's 1024 x 16 keys (62 blocks for a million keys on
 * 256 CUs): 512 x 8 below 1.5 M keys, 1024 x 8 above (option "sort_shape" 0; 1 / 2 / 3 force 512 x 8 / 1024 x 16 /
 * 1024 x 8) - index build 163 -> 136 us at 1 M points, 947 -> 905 us at 16 M.  Same order whatever the shape. */
int64_t flooder_index_sort_state_words(int64_t n_pts, int key_bits);
int flooder_index_sort_zeroed(const int64_t* codes, int64_t n_pts, int key_bits, int64_t* codes_sorted, int32_t* order,
                              void* tmp, int64_t tmp_bytes, int32_t* state, void* stream);

/* Order of a balanced k-d tree over the cloud (default of the point index above 3 dimensions; core.KD_ORDER_ABOVE_DIM):
 * order[j] = index of the point at row j, such that every ALIGNED group of 16 * 2^i rows - the leaves and inner nodes
 * of the implicit box tree - is a cell of the tree (each level splits every group along the widest axis of its box at
 * the positional median).  One (segmented box, key, radix sort) round per level; no curve codes.  In 6D the boxes of
 * the 1024-point nodes overlap 3x less than those of a Hilbert order and the sorted sweep tests and evaluates fewer
 * leaves.  Any order is a valid index: results do not depend on it.  tmp: flooder_kd_order_bytes(n_pts) bytes of device
 * scratch (-1: too many points, n_pts <= 2^31 - 1).  Replaces torch.argsort of core.py:143 like flooder_index_sort. */
int64_t flooder_kd_order_bytes(int64_t n_pts);
int flooder_kd_order_f32(const float* pts, int64_t n_pts, int dim, int ld, int32_t* order, void* tmp, int64_t tmp_bytes,
                         void* stream);

/* Sub-cloud of a block of simplices (block-sharded runs): the rows of pts (n_pts x dim floats, row stride ld) that lie
 * inside the box (box: 2 * dim device floats, lo then hi, bounds inclusive) AND - dim 2 / 3, with cell_flags
 * (flooder_select_grid_bytes(dim) zeroed bytes), cloud_box (the 16 floats of flooder_bbox_f32) and n_balls bounding
 * balls (centers n_balls x dim, radii) given - in a cell of a coarse grid over the cloud's box that one of the balls
 * reaches; compacted into out (room for n_pts rows of dim floats; any order), their number in *count (zeroed by the
 * caller).  With landmarks that are cloud points every witness of a simplex lies in its bounding ball
 * (core.py:156-172), so the sweep of the block against this sub-cloud gives the values of the whole cloud. */
int64_t flooder_select_grid_bytes(int dim);
int flooder_box_select_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* box, const float* cloud_box,
                           const float* centers, const float* radii, int64_t n_balls, uint8_t* cell_flags, float* out,
                           int32_t* count, void* stream);

/* out (n_pad x flooder_padded_dim(dim) floats) = rows order[0], order[1], ... of pts, padding columns 0, then
 * +inf rows up to n_pad (a multiple of FLOODER_BVH_LEAF).  Replaces points[indices] of core.py:143. */
int flooder_gather_rows_f32(const float* pts, int64_t n_pts, int dim, int ld, const int32_t* order, float* out,
                            int64_t n_pad, void* stream);

/* Number of nodes (all levels, each padded to a multiple of 64) of the tree over n_pts points. */
int64_t flooder_bvh_node_count(int64_t n_pts);

/* Build the tree.  pts_sorted: padded rows (flooder_padded_dim(dim) floats), Morton order, row count
 * rounded up to a multiple of FLOODER_BVH_LEAF with +inf rows.  nodes: flooder_bvh_node_count(n_pts)
 * x 2 x padded_dim floats (box lo then hi per node). */
int flooder_bvh_build_f32(const float* pts_sorted, int64_t n_pts, int dim, float* nodes, void* stream);

/* flooder_gather_rows_f32 + flooder_bvh_build_f32 in one call: the rows are written in curve order and the leaf boxes
 * are reduced from them while they are in registers (one pass over the cloud instead of two), then the inner levels.
 * density_grid / cloud_box (both NULL, or flooder_density_grid_words(dim) ZEROED int32 and the 16-float box of
 * flooder_bbox_f32; dim 2 and 3): the same pass accumulates the density grid the cell sweep reads its first cell size
 * from (what flooder_density_grid_f32 computes from a finished tree). */
int flooder_index_rows_f32(const float* pts, int64_t n_pts, int dim, int ld, const int32_t* order, float* rows,
                           int64_t n_pad, float* nodes, int32_t* density_grid, const float* cloud_box, void* stream);

/* Sweep: out_d2[s, r] = bits(min over all points of |p(s,r) - x|^2) with p as in flooder_sweep_f32.
 * Plain stores (every cell is written exactly once); queue = FLOODER_QUEUE_WORDS zeroed int32 (sharded heads, see
 * flooder_sweep_cell_f32); stats = NULL or four
 * zeroed uint64 counters {leaves evaluated, leaves tested, inner nodes expanded, most tests by one item}. */
int flooder_sweep_bvh_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                          const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                          int32_t* queue, uint32_t* out_d2, uint64_t* stats, void* stream);

/* Tree sweep over spatially sorted samples (default of flood_complex above 3 dimensions).  Same result in the same
 * (S, R) buffer as flooder_sweep_bvh_f32 - out_d2[s, r] = bits(min over ALL points), bit for bit - but a wave takes 64
 * consecutive samples of a Z-order of ALL (simplex, sample) pairs instead of the samples of one simplex: tight tile
 * boxes whatever the size of the simplices (coarse lattices on large simplices in 6D: 6x fewer box tests).
 *   flooder_sample_key_bits(dim)   bits of a sample key (floor(32 / dim) per axis, at most 10)
 *   flooder_sample_keys_f32        keys[s * R + r] = Morton code of sample (s, r) inside box (16 floats, [0:dim] = min,
 *                                  [8:8+dim] = max: the cloud's box from flooder_bbox_f32); n_simplices * R < 2^32 - 1
 *   (sort)                         flooder_index_sort(keys, n_simplices * R, key bits, ...) -> sample_order
 *   flooder_sweep_bvh_sorted_f32   the sweep; queue = FLOODER_QUEUE_WORDS zeroed int32; stats as flooder_sweep_bvh_f32.
 * Replaces compute_mask + nonzero + compute_filtration (core.py:210-226) like the other sweeps. */
int flooder_sample_key_bits(int dim);
int flooder_sorted_tile_samples(void);   /* samples per tile of the sorted sweep (64 x option "sorted_ks") */
int flooder_sample_keys_f32(const float* verts, const float* weights, int k1, int R, int64_t n_simplices, int dim,
                            const float* box, uint32_t* keys, void* stream);
int flooder_sweep_bvh_sorted_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                 const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                 const int32_t* sample_order, int32_t* queue, uint32_t* out_d2, uint64_t* stats,
                                 void* stream);
/* The same sweep over ONE RANK'S tiles of a multi-GPU run: the contiguous run [T r / W, T (r + 1) / W) of the T tiles
 * of the sorted order - tiles of the unsharded sweep (above 3D a rank that took every W-th SIMPLEX would sweep a W times
 * thinner sample set in W^(1/dim) times wider tiles) in one region of space.  out_d2 is written for the rank's samples
 * only: zero it first, take the face maxima over all of it and combine the ranks' (S, F) values with MAX. */
int flooder_sweep_bvh_sorted_shard_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                       const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                       const int32_t* sample_order, int shard_rank, int shard_world, int32_t* queue,
                                       uint32_t* out_d2, uint64_t* stats, void* stream);

/* The sorted sweep fused with the per-face maxima (core.py:251-276 folded in; the default above 3 dimensions when only
 * the face values are wanted): no (S, R) buffer.  A sample whose running minimum - the distance to a real point - does
 * not exceed the running maximum of any face it lies on (memb[r]: bit f = row r lies on face f, n_faces <= 32;
 * face_bits: n_simplices x n_faces zeroed uint32, or as many words as face_slot addresses) leaves the tile's pruning
 * radius; the samples still in the running at the end of a traversal are exact and raise face_bits (integer atomic
 * max on the d2 bits).  Values equal the exhaustive result bit for bit; flooder_face_values_f32 takes the roots.
 * flooder_sample_keys_late_f32: as flooder_sample_keys_f32, but the rows with late_rows[r] != 0 sort behind all
 * others (top key bit) - the caller marks every row except one PILOT near the centre of each face, so the face
 * maxima are close to final when the bulk of the samples is swept.  Sort over 32 key bits.  Option "sorted_refresh"
 * (4): evaluated leaves between two readings of the face maxima. */
int flooder_sample_keys_late_f32(const float* verts, const float* weights, int k1, int R, int64_t n_simplices, int dim,
                                 const float* box, const uint8_t* late_rows, uint32_t* keys, void* stream);
int flooder_sweep_bvh_sorted_faces_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                       const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                       const int32_t* sample_order, int32_t* queue, const uint32_t* memb, int n_faces,
                                       uint32_t* face_bits, const int32_t* face_slot, uint64_t* stats, void* stream);

/* Same sweep restricted to an explicit work list and SEEDED with the minima already in out_d2 (upper
 * bounds from an earlier pass): item_list[i] = simplex * ceil(R/64) + tile, tiles are 64 consecutive
 * samples; *n_items (device int32) entries.  Finishes what flooder_sweep_cell_f32 could not verify.
 * budget > 0: a wave abandons a tile after that many box tests, stores its current minima and appends the
 * tile to list2 / *count2 (zeroed by the caller) - call again on list2 with budget 0, where each tile is
 * split over up to 64 waves (tiles near the medial axis of the cloud have huge candidate sets). */
int flooder_sweep_bvh_items_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                const float* verts, const float* weights, int k1, int R,
                                int64_t n_simplices, const int32_t* item_list, const int32_t* n_items,
                                int32_t* queue, uint32_t* out_d2, int budget, int32_t* list2,
                                int32_t* count2, uint64_t* stats, void* stream);

/* Density grid of the cloud for the cell sweep (dim 2 and 3): point counts in 64^3 (256^2) cells over the cloud's
 * box, accumulated from the leaves of the box tree (nodes: the array of flooder_index_rows_f32, leaves first).
 * grid: flooder_density_grid_words(dim) int32, ZEROED - the fine grid, then four words that flooder_cloud_kind fills. */
int64_t flooder_density_grid_words(int dim);
int flooder_density_grid_f32(const float* nodes, int64_t n_pts, int dim, const float* cloud_box, int32_t* grid,
                             void* stream);
/* What kind of cloud is it?  Pools the filled density grid into 16^3 (64^2) coarse cells and counts the points in
 * INTERIOR cells - occupied cells whose axis neighbours are all occupied: 91 - 99 % of a cloud that fills a volume
 * (Gaussian, swiss cheese, annulus in the plane), 7 - 21 % of one that lies on a surface (the noisy torus); words [2]
 * and [3] behind the fine grid.  The cell sweep reads them and tries ONE cell size per chunk instead of two on surface
 * clouds (option "cell_surface_pct", 60; 0 = never): cfg 3 3.88 -> 3.75 ms; flooder_fused_witness reads them too and
 * tries no simplex at all on such a cloud (option "wit_surface_pct", 60; 0 = always try): 3.75 -> 3.70 ms.  flooder_index_rows_f32 computes the
 * statistic itself (in spare workgroups of the launch that builds the first inner tree level); this entry point is for grids filled by flooder_density_grid_f32 (one launch; the four words must be zero).
 * Without it the words stay zero and the sweep tries two sizes, as before round 6.  No result depends on it. */
int flooder_cloud_kind(int32_t* density_grid, int dim, void* stream);

/*
 * Cell sweep (dim 2 and 3; the default device path there).  One wave per chunk of 256 consecutive
 * samples of a simplex: the cloud's density inside the chunk's bounding box gives a cell size
 * c = alpha * (volume / points)^(1/dim); the points within c of the chunk are gathered through the box
 * tree and counting-sorted into a cell grid in LDS; every sample visits the 3^dim cells around it.  A
 * minimum <= (0.999 c)^2 is provably the nearest neighbour; otherwise c doubles (up to 3 rounds).  Tiles
 * of 64 samples that stay unverified or do not fit the LDS stage are appended to flag_list
 * (n_simplices * ceil(R/64) int32; *flag_count zeroed by the caller) for flooder_sweep_bvh_items_f32,
 * which finishes them exactly.  alpha > 0 only trades speed (1.35 is a good value), never correctness.
 * queue: 3 x FLOODER_QUEUE_WORDS zeroed int32 (the sharded work-queue heads of up to three launches: one returning
 * atomic on a single head word serves ~88 pops per microsecond chip-wide - half a million chunks would queue for
 * 5.7 ms - so a queue has FLOODER_QUEUE_SHARDS heads 128 B apart).  stats: NULL or nine zeroed uint64 {pairs evaluated, points staged, tiles
 * flagged, re-staging rounds, chunks given up: tree gather overflow at density / at staging, kept list
 * full, cell doublings exhausted; rounds evaluated exhaustively because the LDS stage was full}.
 * plane_scratch: 24 * n_simplices floats of device scratch (the face planes of every simplex, computed once by a
 * small kernel instead of by each of its chunks).
 * density_grid / cloud_box (both NULL, or the grid of flooder_density_grid_f32 and the 16-float box of
 * flooder_bbox_f32): with them a chunk whose box lies over well filled cells (option "cell_density_grid": at least
 * that many points in each, default 16; 0 = never) takes the local density from the grid instead of walking the tree
 * and counting the points under its box.
 */
int flooder_sweep_cell_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                           const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                           float alpha, int32_t* queue, uint32_t* out_d2, int32_t* flag_list, int32_t* flag_count,
                           float* plane_scratch, const int32_t* density_grid, const float* cloud_box, uint64_t* stats,
                           void* stream);

/*
 * Cell sweep fused with the per-face maxima (core.py:251-276 folded into the sweep): as flooder_sweep_cell_f32 over
 * all R rows, but every sample whose nearest neighbour is settled raises face_bits[s * n_faces + f] (integer
 * atomic max on the d2 bits; zeroed by the caller) for each face f whose bit is set in memb[r] (n_faces <= 32;
 * random mode: one face, every memb word = 1).  face_slot (NULL, or n_simplices x n_faces int32): the word of
 * face_bits that face f of simplex s uses, face_slot[s * n_faces + f], so that a triangle / edge / vertex shared by
 * several simplices (its samples are bit-identical from each) has ONE running maximum.  The (S, R) buffer becomes scratch: only the tiles appended to
 * flag_list are written (bit 31 of a word = that sample is already settled), the other cells stay undefined.
 * top / top_list / top_count (all NULL, or n_simplices zeroed uint64 / n_simplices int32 / one zeroed int32): the
 * probe of the finish folded into the sweep - every flagged tile gets one greedy tree descent for its open samples
 * (finite upper bounds) while they are still in registers, and top[s] = (largest such bound << 32 | tile id).
 * defer_list / defer_c / defer_ctl (all NULL, or 5 * n_simplices * ceil(R / 256) int32 / as many float32 / eight
 * zeroed int32): three launches instead of one - first every run of four consecutive chunks (1024 samples) is
 * gathered, filtered and staged ONCE and its chunks are queried against that stage; runs that do not fit the stage and
 * chunks that keep open samples are appended to defer_list and worked off chunk by chunk by the second launch; a chunk
 * whose kept points overflow the stage there (a dense or a surface cloud) hands its open tiles of 64 samples on to
 * the third launch (entries behind the first n_simplices * ceil(R / 256) of both buffers), one sample per lane and
 * a region a quarter the size, instead of evaluating every kept point against all 256 samples - only with option
 * "cell_tiles" 1; off by default: four re-centred gathers and classifications cost more than the exhaustive loop they
 * replace (cfg 3 sweep 3.95 vs 2.55 ms).
 * simplex_weight (NULL or n_simplices floats of flooder_simplex_weight_f32) with light_list / heavy_list
 * (n_simplices int32 scratch each): simplices heavier than option "cell_super_weight" (3000) skip the first launch -
 * in a dense region no run of four chunks fits the stage - and are worked off chunk by chunk by the second.  When
 * fewer than half of the simplices are sparse (weight <= option "cell_super_sparse", 600) the first launch gets no
 * work at all and the second sweeps everything in plain order.
 * Queue order (option "cell_weight_classes", default 1): the heavy list of a long queue - and every simplex of a short
 * one (fewer than "cell_super_min_chunks" chunks: a rank's share, no runs) - is taken in descending weight class
 * (powers of two around "cell_super_weight", the given order kept inside a class), so that the densest simplices -
 * their chunks overflow the stage and keep ONE wave busy for 150 us and more - start first; the chunks the runs
 * deferred come ahead of the heavy simplices ("cell_listed_first", default 1).  A short queue is launched with one
 * workgroup per "cell_chunks_per_block" (12) chunks, at least "cell_min_grid" (384).
 * Option "cell_one_pass" (default 125, 0 = off): a chunk of the per-chunk launch whose kept points outgrow the stage
 * stops recording and evaluates them as they come, so that its candidates are streamed and classified once instead of
 * twice - unless the kept set, extrapolated from the share of the candidates seen so far, exceeds this percentage of
 * the cap ("cell_exh_dense" / "cell_exh_sparse") beyond which the chunk is left to the finish anyway.
 * flag_key / flag_hist (both NULL, or as many uint32 as flag_list holds / 8192 zeroed int32; need top): the probe's
 * bound of every flagged tile, parallel to flag_list, and a histogram of the bounds' top 12 bits - with them the
 * finish works the tiles off longest search first.
 * plane_scratch: as flooder_sweep_cell_f32.
 * Followed by flooder_finish_faces_f32 (probed = 1 when top was passed here) and flooder_face_values_f32.
 */
int flooder_sweep_cell_faces_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                                 const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                                 float alpha, int32_t* queue, uint32_t* d2_scratch, const uint32_t* memb,
                                 int n_faces, uint32_t* face_bits, const int32_t* face_slot, int32_t* flag_list,
                                 int32_t* flag_count, uint32_t* flag_key, int32_t* flag_hist, uint64_t* top,
                                 int32_t* top_list, int32_t* top_count, int32_t* defer_list,
                                 float* defer_c, int32_t* defer_ctl, const float* simplex_weight,
                                 int32_t* light_list, int32_t* heavy_list, float* plane_scratch,
                                 const int32_t* density_grid, const float* cloud_box, uint64_t* stats, void* stream);

/*
 * Witness sweep (dim 2 and 3): the sparse simplices of the fused path, a whole simplex per wave, BEFORE
 * flooder_sweep_cell_faces_f32 (same buffers; replaces compute_filtration_kernel, triton_kernels.py:12-45, and the
 * per-face amax of core.py:251-276 for those simplices).  The points around the simplex are gathered and staged once;
 * a coarse sub-lattice of its samples (coarse_rows: n_coarse <= FLOODER_WIT_MAX_COARSE row numbers padded with -1 to
 * FLOODER_WIT_MAX_COARSE entries) is evaluated first and every coarse sample keeps its witness, the point attaining
 * its minimum; every other sample takes the distance to the witnesses of four nearby coarse samples (parents[r]: four
 * coarse slots, 8 bits each; a coarse row's first parent is itself) as an upper bound, and is dropped when that bound
 * cannot raise the running maximum of any face it lies on.  What is left is evaluated against the stage; samples whose
 * nearest point may lie beyond the staged region are handed to flooder_finish_faces_f32 through flag_list / flag_key /
 * flag_hist / top (as the cell sweep's open tiles; their rows of d2_scratch hold the bound, the other rows of such a
 * tile are marked settled).  Face values are bit-identical to the exhaustive result.
 * simplex_weight (n_simplices floats of flooder_simplex_weight_f32, READ AND WRITTEN): simplices heavier than option
 * "wit_weight" (800), or whose neighbourhood does not fit one LDS stage, are left alone; a simplex handled here gets
 * weight -1, which flooder_sweep_cell_faces_f32 skips.  R <= FLOODER_WIT_MAX_ROWS.  queue: FLOODER_QUEUE_WORDS zeroed
 * int32.  item_list: n_simplices int32 of scratch (the simplices light enough, heaviest class first).  stats: NULL or 24 zeroed uint64 {simplices handled, too heavy, gather overflow, too dense for the stage,
 * points staged, coarse samples certified, samples live after the bound, evaluation rounds, samples handed to the
 * finish, tiles flagged, pairs evaluated, excess bins kept; [12:22] cycles per phase in builds with -DFLOODER_PHASE_TIMERS}.  Options: "wit_cmax_pct" (250: gather radius in percent of
 * the local point spacing), "wit_min_bins" (6), "wit_grid".
 */
int flooder_wit_max_rows(void);
int flooder_wit_max_coarse(void);
int flooder_sweep_witness_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes, const float* verts,
                              const float* weights, int k1, int R, int64_t n_simplices, const int32_t* coarse_rows,
                              int n_coarse, const uint32_t* parents, int32_t* queue, uint32_t* d2_scratch,
                              const uint32_t* memb, int n_faces, uint32_t* face_bits, const int32_t* face_slot,
                              int32_t* flag_list, int32_t* flag_count, uint32_t* flag_key, int32_t* flag_hist,
                              uint64_t* top, int32_t* top_list, int32_t* top_count, float* simplex_weight,
                              int32_t* item_list, float* plane_scratch, uint64_t* stats, void* stream);

/*
 * Exact finish of the flagged tiles when only the face maxima are wanted.  A sample whose upper bound does not
 * exceed the running maximum of every face it lies on cannot change a result and is dropped; the others are
 * traversed exactly (box tree, nearest first) and delivered with integer atomic max.  Passes: a probe (one greedy
 * descent per tile: finite upper bounds, and per simplex the tile with the largest one; skipped when the cell sweep
 * did it), optionally ("finish_top" 1) the top tile of every simplex, then all tiles, largest bound first.  Face
 * values equal the exhaustive result bit for bit.
 *   ctl: 24 + 3 * FLOODER_QUEUE_WORDS zeroed int32 (list lengths and the team passes' queue heads, then the sharded
 *   queue heads of the per-wave passes; ctl[3] = number of entries of top_list, which the cell
 *   sweep's probe may already have filled: then probed = 1 and the probe pass is skipped); top: n_simplices uint64
 *   (zeroed unless probed); top_list: n_simplices int32;
 *   flag_key / flag_hist / flag_sorted: all NULL, or what flooder_sweep_cell_faces_f32 filled plus scratch of the
 *   size of flag_list - a counting sort by descending bound puts the long searches first in the last pass's queue
 *   (option "finish_order" 0 turns it off);
 *   hard_scratch / hard_cap: NULL / 0, or 4 * hard_cap uint64 of scratch - a tile that has evaluated more leaves
 *   than option "finish_budget" (14; a wave raises its issue priority after 32 leaves, so a long search runs ahead of
 *   the three it shares its SIMD with) times the tiles per wave of the whole list is taken off its wave and put on a
 *   list of at most hard_cap entries; the next launch gives every such tile to a workgroup of 16 waves, each
 *   searching an interleaved share of the level-1 nodes of the box tree, the minima combined in LDS round by round
 *   (one more launch; without scratch, or with the budget 0, one wave works every tile off alone);
 *   stats: NULL or 7 zeroed uint64 {leaves evaluated, leaves tested, nodes expanded, -, tiles dropped on arrival in
 *   the last pass, samples live on arrival in the last pass, -}.
 */
int flooder_finish_faces_f32(const float* pts_sorted, int64_t n_pts, int dim, const float* nodes,
                             const float* verts, const float* weights, int k1, int R, int64_t n_simplices,
                             const int32_t* flag_list, const int32_t* flag_count, const uint32_t* flag_key,
                             int32_t* flag_hist, int32_t* flag_sorted, int32_t* ctl,
                             uint64_t* top, int32_t* top_list, int probed, uint32_t* d2_scratch,
                             const uint32_t* memb, int n_faces, uint32_t* face_bits, const int32_t* face_slot,
                             uint64_t* hard_scratch, int hard_cap, uint64_t* stats, void* stream);

/* out_face[i] = sqrt(float(face_bits[i])), i < n: the filtration values (core.py:257, 272: distances, not squares). */
int flooder_face_values_f32(const uint32_t* face_bits, int64_t n, float* out_face, void* stream);

/* weight[s] = rough number of cloud points inside the bounding box of simplex s (walk of the box tree down to its
 * 1024-point nodes, overlapped volume fractions).  The host queues the simplices by descending weight: work per
 * simplex is heavy-tailed and the long ones should start first.  Replaces the sort by ball centre of core.py:174-179
 * as the work order (the order of the simplices never changes a value). */
int flooder_simplex_weight_f32(const float* nodes, int64_t n_pts, int dim, const float* verts, int k1,
                               int64_t n_simplices, float* weight, void* stream);

/* The same launch PREPARING a fused sweep: weights as above, plus (plane_scratch != NULL, dim 2 / 3) the face-plane rows
 * the witness and the cell sweep read - their entry points, next on this stream with the same verts / plane_scratch /
 * n_simplices, then skip their own plane launch - plus a zero fill of zero_words int32 words at zero_buf (the control
 * words, queue heads and face words the sweep's launches start from; NULL / 0: none).  One launch instead of three
 * (fill, weights, planes): ~8 us of cfg 2's step. */
int flooder_simplex_prepare_f32(const float* nodes, int64_t n_pts, int dim, const float* verts, int k1,
                                int64_t n_simplices, float* weight, float* plane_scratch, int32_t* zero_buf,
                                int64_t zero_words, void* stream);
/* The note "the plane rows of (verts, plane_scratch, n_simplices) on this stream are written" lives in the calling
 * thread until the next witness / cell sweep entry point reads it.  A caller that gives up between the two (an error
 * in between) drops the note with this call - a later sweep with recycled buffers at the same addresses must not
 * inherit it. */
void flooder_simplex_planes_forget(void);

/* Device self-test of the 64-lane DPP reductions: out128[0:64] = min(in64), out128[64:128] = max. */
int flooder_selftest(const float* in64, float* out128, void* stream);

/* Set n uint32 words to `value` (used to initialise d2 buffers to +inf bits). */
int flooder_fill_u32(uint32_t* buf, int64_t n, uint32_t value, void* stream);

/*
 * Farthest-point sampling.  Replaces fpsample.bucket_fps_kdline_sampling as called by
 * generate_landmarks (flooder/core.py:337-343): exact FPS order starting at `start`.
 *   out_idx: n_lms int64 (selection order, out_idx[0] = start);
 *   work_min: 4 * n_pts float32 scratch, 16-byte aligned (rows x, y, z, running squared distance to the
 *             selected set for dim <= 3; the first n_pts floats otherwise);
 *   work_best: 64 * n_lms uint64 scratch, zeroed by the caller (64 arg-max slots per iteration).
 */
int flooder_fps_f32(const float* pts, int64_t n_pts, int dim, int ld, int n_lms, int64_t start,
                    int64_t* out_idx, float* work_min, uint64_t* work_best, void* stream);

/*
 * ---- float64 sweep -------------------------------------------------------------------------------------------------
 * The reference's kernels are instantiated with DTYPE = fp64 for float64 inputs (triton_kernels.py:226-229).  Here:
 * the box tree of the float32-rounded cloud (flooder_bvh_build_f32) with the float64 rows gathered in the same order;
 * boxes are widened by one float32 ulp per side inside the kernel and every bound and distance is evaluated in double.
 *   flooder_gather_rows_f64: as flooder_gather_rows_f32 for double rows (padding rows +inf, padding columns 0);
 *   flooder_sweep_bvh_f64:   out_d2[s, r] = bit pattern of the double min over all points of |p(s, r) - x|^2
 *                            (non-negative doubles order like unsigned 64-bit integers); queue: FLOODER_QUEUE_WORDS zeroed int32;
 *   flooder_face_max_f64:    face maxima + sqrt in double (layout as flooder_face_max_f32).
 */
int flooder_gather_rows_f64(const double* pts, int64_t n_pts, int dim, int ld, const int32_t* order, double* out,
                            int64_t n_pad, void* stream);
int flooder_sweep_bvh_f64(const double* pts_sorted, int64_t n_pts, int dim, const float* nodes, const double* verts,
                          const double* weights, int k1, int R, int64_t n_simplices, int32_t* queue,
                          uint64_t* out_d2, void* stream);
int flooder_face_max_f64(const uint64_t* d2, int64_t n_simplices, int R, const int32_t* face_ptr,
                         const int32_t* face_rows, int n_faces, double* out_face, double* out_dist, void* stream);

/*
 * Bucketed exact farthest-point sampling (dim <= 3) over the curve-sorted copy of the cloud that the sweeps use.
 * Replaces fpsample.bucket_fps_kdline_sampling (flooder/core.py:337-343: exact FPS accelerated by kd-tree buckets):
 * buckets of 64 or 256 consecutive sorted rows carry a bounding box and their largest running minimum; a new
 * landmark only updates the buckets whose box is closer to it than that maximum.  Same selection as
 * flooder_fps_f32, bit for bit (same per-point arithmetic; ties: lowest original index).
 *   pts / ld: the cloud in its ORIGINAL order (out_idx refers to it); pts_sorted: padded rows in curve order;
 *   order: int32, order[j] = original index of sorted row j;
 *   rows: 4 * n_pts float32 scratch (16-byte aligned); bucket_box: 8 * flooder_fps_bucket_count(n_pts) float32;
 *   bucket_key: flooder_fps_bucket_count(n_pts) uint64; work_best: 64 * n_lms uint64, zeroed by the caller.
 * Options "fps_switch" (first bucketed iteration; 0 = auto) and "fps_rpl" (rows per lane and bucket: 1 or 4; 0 = auto).
 */
int64_t flooder_fps_bucket_count(int64_t n_pts);
int flooder_fps_indexed_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* pts_sorted,
                            const int32_t* order, int n_lms, int64_t start, int64_t* out_idx, float* rows,
                            float* bucket_box, uint64_t* bucket_key, uint64_t* work_best, void* stream);

/* The same selection with SEVERAL landmarks per launch (default of generate_landmarks on ROCm tensors, dim <= 8).
 * FPS is sequential, but the runners-up of the arg-max stay the next landmarks as long as each is at least its own
 * running minimum away from the ones before it (flood_fps2.hip): a launch selects such a prefix (<= 8) and applies
 * all its updates at once - same indices as flooder_fps_indexed_f32 / flooder_fps_f32 in a fraction of the launches.
 * pts_sorted / order: rows of the cloud in curve order (flooder_padded_dim(dim) floats each) and the sorted row ->
 * original index map (flooder_index_rows_f32 / flooder_index_sort).  n_pts <= flooder_fps_batched_max_points().
 * Workspaces (device): minsq n_pts floats; bucket_box 2 * padded_dim * flooder_fps_bucket_count(n_pts) floats;
 * bucket_keys 3 * bucket count uint64; bucket_coord padded_dim * bucket count floats; work_best 64 * n_lms uint64,
 * ZEROED; work_rec flooder_fps_batched_rec_words(n_pts, dim, n_lms) uint32; work_ctr n_lms + 4 int32, ZEROED.
 * launches_out (host pointer, may be NULL): kernel launches used.
 * The number of launches depends on the data.  UNLIKE every other entry point this one therefore WAITS for the
 * device: every launch stores (launches done, landmarks selected) into a pinned host word (one per device, allocated
 * at first use), the host keeps 24 launches in flight beyond the last one it has seen complete and returns when the
 * word says "all selected" (the stream itself is not drained: a few no-op launches may still be queued).  Option
 * "fps_rounds" = 1 (and a host where pinned memory cannot be had): doubling rounds of launches with a read-back of
 * the landmark counter (4 bytes, hipStreamSynchronize) between rounds.  "fps_lane_best" = 1: test hook, ranks at most
 * one candidate per lane (the path more than 64 candidates take). */
int64_t flooder_fps_batched_max_points(void);
int64_t flooder_fps_batched_rec_words(int64_t n_pts, int dim, int n_lms);
int flooder_fps_batched_f32(const float* pts, int64_t n_pts, int dim, int ld, const float* pts_sorted,
                            const int32_t* order, int n_lms, int64_t start, int64_t* out_idx, float* minsq,
                            float* bucket_box, uint64_t* bucket_keys, float* bucket_coord, uint64_t* work_best,
                            uint32_t* work_rec, int32_t* work_ctr, int32_t* launches_out, void* stream);

/*
 * ---- parameter blocks ----------------------------------------------------------------------------------------------
 * The entry points of the DEFAULT path that take more than a dozen arguments also exist in a form that takes ONE
 * struct: the three launches of the fused 2-D / 3-D sweep share almost all of their buffers (flooder_fused_sweep_t,
 * filled once, passed three times), the sorted-sample sweep above 3-D and the batched landmark selection have a block
 * each.  A maintainer binds a struct field by NAME (cgo / ctypes.Structure / JNA) - a transposed pointer in a run of
 * thirty positional void* is a silently wrong answer, a misspelt field is a compile error.  Every block opens with
 * `size` (sizeof of the struct as the caller compiled it) and `abi` (FLOODER_PARAMS_ABI): a library that knows a longer
 * struct reads the fields the caller has and takes the documented default (NULL / 0) for the rest; a caller newer than
 * the library is refused (FLOODER_E_ARG).  Pointers are device pointers unless said otherwise; what every field
 * means, and which may be NULL, is documented at the positional function it is forwarded to - those stay exported as
 * they were (same symbols, same behaviour; flooder_amd's own default path no longer calls them).
 */
#define FLOODER_PARAMS_ABI 1

typedef struct flooder_fused_sweep_s {
  uint32_t size, abi;
  /* the cloud and its index (flooder_index_rows_f32 / flooder_bvh_build_f32 / flooder_density_grid_f32) */
  const float* pts_sorted;
  int64_t n_pts;
  int32_t dim;
  int32_t k1;                 /* vertices per simplex */
  const float* nodes;
  const int32_t* density_grid;
  const float* cloud_box;
  /* simplices and sample lattice */
  const float* verts;
  const float* weights;
  int32_t R;
  int32_t n_faces;
  int64_t n_simplices;
  const uint32_t* memb;
  float alpha;
  int32_t n_coarse;           /* witness plan: flooder_sweep_witness_f32 */
  const int32_t* coarse_rows;
  const uint32_t* parents;
  /* result */
  uint32_t* face_bits;
  const int32_t* face_slot;
  /* scratch shared by the three launches */
  uint32_t* d2_scratch;
  int32_t* flag_list;
  int32_t* flag_count;
  uint32_t* flag_key;
  int32_t* flag_hist;
  int32_t* flag_sorted;
  uint64_t* top;
  int32_t* top_list;
  int32_t* top_count;
  float* simplex_weight;
  float* plane_scratch;
  /* witness sweep */
  int32_t* wit_queue;
  int32_t* wit_item_list;
  uint64_t* wit_stats;
  /* cell sweep */
  int32_t* cell_queue;
  int32_t* defer_list;
  float* defer_c;
  int32_t* defer_ctl;
  int32_t* light_list;
  int32_t* heavy_list;
  uint64_t* cell_stats;
  /* finish */
  int32_t* finish_ctl;
  uint64_t* hard_scratch;
  int32_t hard_cap;
  int32_t probed;
  uint64_t* finish_stats;
} flooder_fused_sweep_t;

/* flooder_sweep_witness_f32, flooder_sweep_cell_faces_f32, flooder_finish_faces_f32 on the fields of *p (host memory;
 * read during the call only).  flooder_fused_witness also hands `density_grid` (may be NULL) to the witness sweep, which
 * then stands back on a cloud that lies on a surface (option "wit_surface_pct"); the positional function has no such
 * argument and always tries. */
int flooder_fused_witness(const flooder_fused_sweep_t* p, void* stream);
int flooder_fused_cell(const flooder_fused_sweep_t* p, void* stream);
int flooder_fused_finish(const flooder_fused_sweep_t* p, void* stream);

typedef struct flooder_sorted_sweep_s {   /* flooder_sweep_bvh_sorted_faces_f32 / _sorted_f32 / _sorted_shard_f32 */
  uint32_t size, abi;
  const float* pts_sorted;
  int64_t n_pts;
  int32_t dim;
  int32_t k1;
  const float* nodes;
  const float* verts;
  const float* weights;
  int32_t R;
  int32_t n_faces;
  int64_t n_simplices;
  const int32_t* sample_order;
  int32_t* queue;
  const uint32_t* memb;
  uint32_t* face_bits;
  const int32_t* face_slot;
  uint64_t* stats;
  uint32_t* out_d2;           /* flooder_sorted_minima only: the (S, R) minima */
  int32_t shard_rank;         /* flooder_sorted_minima: this rank's run of the tiles when shard_world > 1 */
  int32_t shard_world;        /* 0 or 1: all tiles */
} flooder_sorted_sweep_t;
int flooder_sorted_faces(const flooder_sorted_sweep_t* p, void* stream);    /* fused with the face maxima */
int flooder_sorted_minima(const flooder_sorted_sweep_t* p, void* stream);   /* minima into out_d2 (all tiles / a shard) */

typedef struct flooder_fps_batched_s {    /* flooder_fps_batched_f32 */
  uint32_t size, abi;
  const float* pts;
  int64_t n_pts;
  int32_t dim;
  int32_t ld;
  const float* pts_sorted;
  const int32_t* order;
  int64_t start;
  int32_t n_lms;
  int32_t reserved;
  int64_t* out_idx;
  float* minsq;
  float* bucket_box;
  uint64_t* bucket_keys;
  float* bucket_coord;
  uint64_t* work_best;
  uint32_t* work_rec;
  int32_t* work_ctr;
  int32_t* launches_out;      /* host */
} flooder_fps_batched_t;
int flooder_fps_batched(const flooder_fps_batched_t* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOODER_HIP_H */
