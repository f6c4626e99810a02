"""ctypes binding of ``libflooder_hip.so`` (C ABI declared in ``include/flooder_hip.h``).

There is no CPU fallback: if the shared library is missing or cannot be loaded, every GPU
entry point raises ``ImportError`` telling the user to build it (``python -m flooder_amd.build``).
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# FLOODER_HIP_LIB: another build of the library (the diagnostic variants of tools/gpu_session.sh are built beside
# the product library and selected here - the product file is never overwritten)
LIB_PATH = os.environ.get("FLOODER_HIP_LIB") or os.path.join(_PKG_DIR, "libflooder_hip.so")



class _Block(ctypes.Structure):
    """A parameter block of include/flooder_hip.h: ``size`` and ``abi`` are filled in, fields are set by NAME (a
    keyword that is not a field of the struct is a TypeError), tensors are passed as their ``data_ptr()``."""

    ABI = 1

    def __init__(self, **fields):
        super().__init__()
        self.size = ctypes.sizeof(self)
        self.abi = self.ABI
        known = {name for name, _ in self._fields_}
        for key, value in fields.items():
            if key not in known:
                raise TypeError(f"{type(self).__name__} has no field {key!r}")
            if value is not None and not isinstance(value, (int, float)):
                value = ptr(value)
            setattr(self, key, value)


def _fields(spec: str):
    """``"p pts_sorted; i64 n_pts; i32 dim"`` -> ctypes fields (p: pointer, i64 / i32 / u32 / f32)."""
    kinds = {"p": c_void_p, "i64": c_int64, "i32": c_int32, "u32": c_uint32, "f32": c_float}
    out = []
    for item in spec.replace("\n", " ").split(";"):
        item = item.strip()
        if item:
            kind, name = item.split()
            out.append((name, kinds[kind]))
    return out


class FusedSweep(_Block):
    """``flooder_fused_sweep_t``: the buffers the three launches of the fused 2-D / 3-D sweep share."""

    _fields_ = _fields("""u32 size; u32 abi; p pts_sorted; i64 n_pts; i32 dim; i32 k1; p nodes; p density_grid; p cloud_box;
        p verts; p weights; i32 R; i32 n_faces; i64 n_simplices; p memb; f32 alpha; i32 n_coarse; p coarse_rows; p parents;
        p face_bits; p face_slot; p d2_scratch; p flag_list; p flag_count; p flag_key; p flag_hist; p flag_sorted; p top;
        p top_list; p top_count; p simplex_weight; p plane_scratch; p wit_queue; p wit_item_list; p wit_stats;
        p cell_queue; p defer_list; p defer_c; p defer_ctl; p light_list; p heavy_list; p cell_stats; p finish_ctl;
        p hard_scratch; i32 hard_cap; i32 probed; p finish_stats""")


class SortedSweep(_Block):
    """``flooder_sorted_sweep_t``: the sorted-sample sweep above three dimensions."""

    _fields_ = _fields("""u32 size; u32 abi; p pts_sorted; i64 n_pts; i32 dim; i32 k1; p nodes; p verts; p weights; i32 R;
        i32 n_faces; i64 n_simplices; p sample_order; p queue; p memb; p face_bits; p face_slot; p stats; p out_d2;
        i32 shard_rank; i32 shard_world""")


class FpsBatched(_Block):
    """``flooder_fps_batched_t``: the batched landmark selection."""

    _fields_ = _fields("""u32 size; u32 abi; p pts; i64 n_pts; i32 dim; i32 ld; p pts_sorted; p order; i64 start; i32 n_lms;
        i32 reserved; p out_idx; p minsq; p bucket_box; p bucket_keys; p bucket_coord; p work_best; p work_rec; p work_ctr;
        p launches_out""")


_lib = None
_load_error: Exception | None = None
_load_missing = False  # the last failure was "file not found" (worth another look after a build)

# name -> (restype, argtypes); the single source the symbol test checks against the header
SIGNATURES = {
    "flooder_abi_version": (c_int, []),
    "flooder_last_error": (c_char_p, []),
    "flooder_device_arch": (c_int, [c_int, c_char_p, c_int]),
    "flooder_set_option": (c_int, [c_char_p, c_int]),
    "flooder_padded_dim": (c_int, [c_int]),
    "flooder_ball_count_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_void_p, c_void_p]),
    "flooder_ball_fill_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_sweep_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                  c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_face_max_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p]),
    "flooder_bbox_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "flooder_bbox_chunk_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
    "flooder_bbox_reduce_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "flooder_morton_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "flooder_morton_zero_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "flooder_curve_key_bits": (c_int, [c_int]),
    "flooder_index_sort_bytes": (c_int64, [c_int64]),
    "flooder_index_sort": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "flooder_index_sort_state_words": (c_int64, [c_int64, c_int]),
    "flooder_index_sort_zeroed": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                          c_void_p]),
    "flooder_kd_order_bytes": (c_int64, [c_int64]),
    "flooder_kd_order_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "flooder_select_grid_bytes": (c_int64, [c_int]),
    "flooder_box_select_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_gather_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "flooder_bvh_node_count": (c_int64, [c_int64]),
    "flooder_bvh_build_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "flooder_index_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "flooder_sweep_bvh_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_sample_key_bits": (c_int, [c_int]),
    "flooder_sorted_tile_samples": (c_int, []),
    "flooder_sample_keys_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "flooder_sample_keys_late_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p]),
    "flooder_sweep_bvh_items_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                            c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                            c_void_p, c_void_p, c_void_p]),
    "flooder_sweep_cell_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                       c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "flooder_density_grid_words": (c_int64, [c_int]),
    "flooder_density_grid_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "flooder_cloud_kind": (c_int, [c_void_p, c_int, c_void_p]),
    "flooder_wit_max_rows": (c_int, []),
    "flooder_wit_max_coarse": (c_int, []),
    "flooder_face_values_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "flooder_simplex_weight_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "flooder_simplex_planes_forget": (None, []),
    "flooder_simplex_prepare_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p,
                                            c_int64, c_void_p]),
    "flooder_selftest": (c_int, [c_void_p, c_void_p, c_void_p]),
    "flooder_fill_u32": (c_int, [c_void_p, c_int64, c_uint32, c_void_p]),
    "flooder_gather_rows_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "flooder_sweep_bvh_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                      c_void_p, c_void_p, c_void_p]),
    "flooder_face_max_f64": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "flooder_fps_bucket_count": (c_int64, [c_int64]),
    "flooder_fps_indexed_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_fps_batched_max_points": (c_int64, []),
    "flooder_fps_batched_rec_words": (c_int64, [c_int64, c_int, c_int]),
    "flooder_fps_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    # the parameter-block forms of the long entry points (what the default path calls)
    "flooder_fused_witness": (c_int, [ctypes.POINTER(FusedSweep), c_void_p]),
    "flooder_fused_cell": (c_int, [ctypes.POINTER(FusedSweep), c_void_p]),
    "flooder_fused_finish": (c_int, [ctypes.POINTER(FusedSweep), c_void_p]),
    "flooder_sorted_faces": (c_int, [ctypes.POINTER(SortedSweep), c_void_p]),
    "flooder_sorted_minima": (c_int, [ctypes.POINTER(SortedSweep), c_void_p]),
    "flooder_fps_batched": (c_int, [ctypes.POINTER(FpsBatched), c_void_p]),
}

# The positional forms of the five entry points above: still exported by the library (same symbols as before round 6),
# typed here so that the symbol test covers them and tests can call them, but nothing in flooder_amd does.
POSITIONAL_SIGNATURES = {
    "flooder_sweep_bvh_sorted_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                             c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_sweep_bvh_sorted_shard_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                   c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                   c_void_p]),
    "flooder_sweep_cell_faces_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                             c_int64, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_sweep_witness_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                          c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "flooder_finish_faces_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                         c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "flooder_sweep_bvh_sorted_faces_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                   c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                   c_void_p, c_void_p]),
    "flooder_fps_batched_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_int64, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
}


def load():
    """Load the library once (after torch, so both share one HIP runtime) and type its symbols."""
    global _lib, _load_error, _load_missing
    if _lib is not None:
        return _lib
    if _load_error is not None:
        # (a library that was missing may have been built since - __graft_entry__.build() imports the package
        # before it compiles: look again; any other failure stays cached)
        if not (_load_missing and os.path.exists(LIB_PATH)):
            raise ImportError(str(_load_error)) from _load_error
        _load_error = None
    try:
        import torch  # noqa: F401  (loads libamdhip64.so.7 first; our DT_NEEDED resolves to it)

        _load_missing = not os.path.exists(LIB_PATH)
        if _load_missing:
            raise OSError(f"{LIB_PATH} not found")
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in {**SIGNATURES, **POSITIONAL_SIGNATURES}.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        from .build import ROOT  # noqa: F401

        if lib.flooder_abi_version() != 1:
            raise OSError(f"ABI version mismatch: library reports {lib.flooder_abi_version()}, binding expects 1")
        _lib = lib
        return lib
    except Exception as exc:  # noqa: BLE001
        _load_error = ImportError(
            f"flooder_amd: the HIP kernel library could not be loaded ({exc}). "
            "Build it with `python -m flooder_amd.build` (needs hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback for CUDA/ROCm tensors."
        )
        raise _load_error from exc


def available() -> bool:
    try:
        load()
        return True
    except ImportError:
        return False


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.flooder_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Raw device (or host) address of a contiguous torch tensor; None -> NULL."""
    if t is None:
        return 0
    if not t.is_contiguous():
        raise ValueError("tensor passed to the native library must be contiguous")
    return t.data_ptr()


def current_stream_ptr(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream
