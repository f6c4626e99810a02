"""Flood complex on MI355X: host side of the coverage-radius sweep.

Mirrors the public interface of the reference's ``flooder/core.py`` - ``flood_complex``
(``core.py:32-43``) and ``generate_landmarks`` (``core.py:291-296``), same argument names,
defaults, return types and error behaviour - and replaces the CUDA/Triton branch of its hot loop
(``core.py:200-226`` + ``flooder/triton_kernels.py``) by hand-written HIP kernels for gfx950
reached through the C ABI in ``include/flooder_hip.h`` (``libflooder_hip.so``, ctypes).

PyTorch is used for tensor plumbing only (allocation, sort, searchsorted, cumsum, streams).
For ROCm tensors the native library is mandatory: there is no eager/PyTorch fallback and a missing
library raises ``ImportError``.  CPU tensors take the reference's CPU branch (scipy kd-tree,
``core.py:127-128, 197-199``).

Data layout in HBM (per call, ambient dimension ``dim``, padded row ``DP = 2 | 4 | 8`` floats):

* ``pts``      (N, DP) f32   cloud in Hilbert-curve order (``PointIndex``; ``method="ball"``: sorted along
                               the widest axis as the reference does), rows padded (one 16 B load in 3D)
* ``nodes``    (n_nodes, 2*DP) f32  implicit box tree over ``pts`` (16-point leaves, fan-out 64)
* ``verts``    (S, d+1, dim) f32, ``centers`` (S, dim), ``radii`` (S,)  per dimension-d pass
* ``weights``  (R, d+1) f32  barycentric sample weights (grid or Dirichlet)
* ``cand``     (P_pad, DP) f32  per-simplex candidate lists (points inside the bounding ball),
                               each padded to a multiple of 8 rows with +inf rows
* ``d2``       (S, R) u32    bit patterns of the running minimum squared distance (+inf = 0x7f800000)
* ``face``     (S, F) f32    per-face maxima after sqrt; the only result copied to the host

The (S, R, dim) tensor of sample points the reference materialises (``core.py:188``, 358 MB at
1 M points / 1 k landmarks) never exists: samples are rebuilt in registers inside the sweep.
"""

from __future__ import annotations

import ctypes
import itertools
import os
import weakref
import warnings
from numbers import Integral
from typing import Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import _native
from .simplex_tree import HAS_GUDHI, SimplexTree, delaunay_cells, delaunay_simplices

__all__ = ["flood_complex", "generate_landmarks", "generate_grid", "generate_uniform_weights",
           "SUPPORTED_DTYPES", "HAS_HIP_KERNELS", "PointIndex", "index_from_host"]

# dtype whitelist of the reference (``tl_dtypes_dict``, triton_kernels.py:226-229)
SUPPORTED_DTYPES = (torch.float32, torch.float64)
INF_BITS = 0x7F800000
CAND_ALIGN = 8
SWEEP_CHUNK = 2048
TILE_SAMPLES = 512
BVH_LEAF = 16
QUEUE_WORDS = 512   # FLOODER_QUEUE_WORDS: zeroed int32 words of one sharded work queue
# device sweep: "cell" = per-simplex LDS cell grid + exact tree finish (default in 2D/3D); "bvh" = box-tree
# culled exact nearest neighbour (default in other dimensions); "ball" = the reference's formulation
# (bounding-ball candidate lists + exhaustive sweep of each list)
SWEEP_METHOD = "auto"
# candidate workspace budget per group of simplices (bytes); bounds HBM use like the reference's
# ``batch_size`` bounds its mask tensor
CAND_WORKSPACE_BYTES = 16 << 30


def _has_hip_kernels() -> bool:
    return _native.available()


HAS_HIP_KERNELS = _has_hip_kernels()


# ------------------------------------------------------------------------------ sampling
def generate_grid(n: int, dim: int, device, dtype) -> Tuple[torch.Tensor, List[torch.Tensor], List[torch.Tensor]]:
    """Barycentric grid weights and face bookkeeping (same contract as ``core.py:346-402``).

    Returns ``(weights (C, dim+1), vertex_idxs, face_idxs)``: rows are the compositions of ``n-1``
    into ``dim+1`` parts in lexicographic-combination order divided by ``n-1``;
    ``face_idxs[k]`` (C(dim+1,k), rows) are the grid rows on each face of codimension ``k`` and
    ``vertex_idxs[k]`` (C(dim+1,k), dim+1-k) that face's vertex columns.
    """
    combs = np.array(list(itertools.combinations(range(n + dim - 1), dim)), dtype=np.int64).reshape(-1, dim)
    lo = np.full((combs.shape[0], 1), -1, dtype=np.int64)
    hi = np.full((combs.shape[0], 1), n + dim - 1, dtype=np.int64)
    grid = np.diff(np.concatenate([lo, combs, hi], axis=1), axis=1) - 1  # (C, dim+1)
    face_idxs, vertex_idxs = [], []
    axes = np.arange(dim + 1)
    for k in range(dim + 1):
        rows_k, verts_k = [], []
        for zero_axes in itertools.combinations(range(dim + 1), k):
            on_face = np.ones(grid.shape[0], dtype=bool)
            for a in zero_axes:
                on_face &= grid[:, a] == 0
            rows_k.append(np.nonzero(on_face)[0])
            verts_k.append(np.setdiff1d(axes, np.array(zero_axes, dtype=np.int64)))
        face_idxs.append(torch.as_tensor(np.stack(rows_k), device=device))
        vertex_idxs.append(torch.as_tensor(np.stack(verts_k), device=device))
    grid_t = torch.as_tensor(grid)
    weights = torch.empty(grid_t.shape, dtype=dtype)
    torch.divide(grid_t, n - 1, out=weights)
    return weights.to(device), vertex_idxs, face_idxs


def generate_uniform_weights(num_rand: int, dim: int, device, dtype) -> torch.Tensor:
    """``num_rand`` uniform samples on the unit ``dim``-simplex (``core.py:405-427``): drawn from the
    global *CPU* torch generator so that CPU and GPU runs see the same samples."""
    if dim == 0:
        return torch.ones((num_rand, 1), device=device, dtype=dtype)
    w = -torch.log(1 - torch.rand(num_rand, dim + 1)).to(device, dtype=dtype)
    return w / w.sum(dim=1, keepdim=True)


# ------------------------------------------------------------------------------ landmarks
def generate_landmarks(points: torch.Tensor, n_lms: int, fps_h: Union[None, int] = None,
                       start_idx: Union[int, None] = None, *, index: Optional["PointIndex"] = None,
                       return_index: bool = False):
    """Farthest-point-sampling landmarks (interface of ``core.py:291-343``).

    The reference delegates to ``fpsample.bucket_fps_kdline_sampling`` on the CPU (a kd-tree
    accelerated *exact* FPS); ``fps_h`` is that library's tree height and is accepted for
    compatibility.  Here the selection runs on the GPU for ROCm tensors - bucketed over the curve-sorted
    cloud like the reference's library (``flooder_fps_indexed_f32``; ``index``, keyword-only, passes a
    ``PointIndex`` of ``points`` to reuse) or, for small clouds and dim > 3, one distance-update + arg-max
    sweep of the cloud per landmark (``flooder_fps_f32``) - and in numpy for CPU tensors.
    Returns ``points[index_set]`` in selection order, same device and dtype as ``points``.
    ``return_index=True`` (keyword-only extension): returns ``(landmarks, index)`` where ``index`` is the
    ``PointIndex`` the selection used (built here if none was passed; ``None`` for CPU tensors and the brute-force
    selection) - the handle to give to ``flood_complex(points, landmarks, index=index)`` so that the curve-sorted
    copy of the cloud is built once for both steps.
    """
    if n_lms <= 0:
        raise RuntimeError(f"Number of landmarks ({n_lms}) must be positive")
    n_pts = len(points)
    n_lms = min(int(n_lms), n_pts)
    if start_idx is None:
        start_idx = int(torch.randint(n_pts, (1,)).item())
    if not (0 <= start_idx < n_pts):
        raise RuntimeError(f"start_idx ({start_idx}) out of range for {n_pts} points")
    used: List = []
    index_set = fps_indices(points, n_lms, start_idx, index=index, _used_index=used)
    if return_index:
        return points[index_set], (used[0] if used else None)
    return points[index_set]


FPS_METHOD = "auto"   # "bucket" (dim <= 3: bucketed over the curve-sorted cloud), "brute" (one full sweep per landmark)
FPS_BUCKET_MIN_POINTS = 200_000   # below this a brute-force step costs no more than its launch
FPS_BUCKET_MAX_DIM = 8            # ambient dimensions the bucketed selection supports
FPS_BATCHED = True                # several landmarks per launch (flooder_fps_batched_f32); False: one per launch, dim <= 3
LAST_FPS_LAUNCHES = 0             # kernel launches of the last batched selection (diagnostic)
FPS_KEEP_DIAG = False             # keep the launch counters and block records of the last batched selection
LAST_FPS_DIAG: Dict[str, object] = {}


def fps_indices(points: torch.Tensor, n_lms: int, start_idx: int = 0, method: Optional[str] = None,
                index: Optional["PointIndex"] = None, _used_index: Optional[List] = None) -> torch.Tensor:
    """Indices of the exact FPS order starting at ``start_idx`` (int64, on ``points.device``).

    ROCm tensors: ``method="bucket"`` (default for large clouds) runs ``flooder_fps_batched_f32`` - exact FPS over
    buckets of the curve-sorted copy of the cloud, several landmarks per launch (``FPS_BATCHED = False``: one per
    launch, ``flooder_fps_indexed_f32``, dim <= 3) - (``index``: a ``PointIndex`` of ``points`` to reuse, else built
    here); ``method="brute"`` runs ``flooder_fps_f32``.  All give the same indices."""
    if points.is_cuda:
        lib = _native.load()
        dim = points.shape[1]
        if dim > 8:
            raise RuntimeError("flooder_amd: ambient dimension > 8 is not supported by the HIP kernels")
        pts = points.detach().to(torch.float32).contiguous()
        n = pts.shape[0]
        method = FPS_METHOD if method is None else method
        if method == "auto":
            method = ("bucket" if dim <= FPS_BUCKET_MAX_DIM and (n >= FPS_BUCKET_MIN_POINTS or index is not None)
                      and n_lms > 64 else "brute")
        if method not in ("bucket", "brute"):
            raise ValueError("method must be 'bucket' or 'brute'")
        if method == "bucket" and dim > FPS_BUCKET_MAX_DIM:
            raise ValueError(f"bucketed FPS supports ambient dimension <= {FPS_BUCKET_MAX_DIM}")
        out_idx = torch.empty(n_lms, dtype=torch.int64, device=pts.device)
        if method == "bucket" and (FPS_BATCHED or dim > 3) and n > int(lib.flooder_fps_batched_max_points()):
            method = "bucket" if dim <= 3 else "brute"     # (one landmark per launch / brute force beyond 16 M points)
            if dim <= 3:
                return _fps_bucket_one_per_launch(lib, pts, n, dim, n_lms, start_idx, index, out_idx, points)
        if method == "bucket" and (FPS_BATCHED or dim > 3):
            global LAST_FPS_LAUNCHES
            if index is None:
                index = _recall_index(points)
                if index is None:
                    index = PointIndex(pts)
                    index.source = (points.data_ptr(), _tensor_version(points))
                    _remember_index(points, index)
            if _used_index is not None:
                _used_index.append(index)
            nb = int(lib.flooder_fps_bucket_count(n))
            dp = index.dp
            minsq = torch.empty(n, dtype=torch.float32, device=pts.device)
            box = torch.empty(2 * dp * nb, dtype=torch.float32, device=pts.device)
            keys = torch.empty(3 * nb, dtype=torch.int64, device=pts.device)
            bcoord = torch.empty(dp * nb, dtype=torch.float32, device=pts.device)
            rec = torch.empty(int(lib.flooder_fps_batched_rec_words(n, dim, n_lms)), dtype=torch.int32, device=pts.device)
            zeroed = torch.zeros(64 * n_lms + (n_lms + 4 + 1) // 2, dtype=torch.int64, device=pts.device)
            work_best = zeroed[:64 * n_lms]
            work_ctr = zeroed[64 * n_lms:].view(torch.int32)
            launches = ctypes.c_int32(0)
            with torch.cuda.device(pts.device):
                st = _native.current_stream_ptr(pts.device)
                blk = _native.FpsBatched(pts=pts, n_pts=n, dim=dim, ld=dim, pts_sorted=index.pts, order=index.order32,
                                         start=int(start_idx), n_lms=int(n_lms), out_idx=out_idx, minsq=minsq,
                                         bucket_box=box, bucket_keys=keys, bucket_coord=bcoord, work_best=work_best,
                                         work_rec=rec, work_ctr=work_ctr, launches_out=ctypes.addressof(launches))
                _native.check(lib.flooder_fps_batched(ctypes.byref(blk), st), "flooder_fps_batched")
            LAST_FPS_LAUNCHES = int(launches.value)
            if FPS_KEEP_DIAG:
                LAST_FPS_DIAG.update(ctr=work_ctr, rec=rec, blocks=rec.numel() // ((n_lms + 4) * (12 + max(dp, 4))),
                                     words=12 + max(dp, 4))
            return out_idx
        if method == "bucket":
            return _fps_bucket_one_per_launch(lib, pts, n, dim, n_lms, start_idx, index, out_idx, points, _used_index)
        work_min = torch.empty(4 * n, dtype=torch.float32, device=pts.device)
        work_best = torch.zeros(64 * n_lms, dtype=torch.int64, device=pts.device)
        with torch.cuda.device(pts.device):
            st = _native.current_stream_ptr(pts.device)
            _native.check(lib.flooder_fps_f32(_native.ptr(pts), n, dim, dim, n_lms, int(start_idx),
                                              _native.ptr(out_idx), _native.ptr(work_min),
                                              _native.ptr(work_best), st), "flooder_fps_f32")
        return out_idx
    pts = points.detach().cpu().numpy().astype(np.float32, copy=False)
    n = pts.shape[0]
    idx = np.empty(n_lms, dtype=np.int64)
    idx[0] = start_idx
    d2 = ((pts - pts[start_idx]) ** 2).sum(axis=1, dtype=np.float32)
    for i in range(1, n_lms):
        j = int(np.argmax(d2))
        idx[i] = j
        np.minimum(d2, ((pts - pts[j]) ** 2).sum(axis=1, dtype=np.float32), out=d2)
    return torch.as_tensor(idx, device=points.device)


def _fps_bucket_one_per_launch(lib, pts, n, dim, n_lms, start_idx, index, out_idx, source=None, used=None):
    """``flooder_fps_indexed_f32``: bucketed exact FPS, one launch per landmark (dim <= 3)."""
    if index is None:
        index = _recall_index(source) if source is not None else None
        if index is None:
            index = PointIndex(pts)
            if source is not None:
                index.source = (source.data_ptr(), _tensor_version(source))
                _remember_index(source, index)
    if used is not None:
        used.append(index)
    nb = int(lib.flooder_fps_bucket_count(n))
    rows = torch.empty(4 * index.pts.shape[0], dtype=torch.float32, device=pts.device)
    box = torch.empty(8 * nb, dtype=torch.float32, device=pts.device)
    key = torch.empty(nb, dtype=torch.int64, device=pts.device)
    work_best = torch.zeros(64 * n_lms, dtype=torch.int64, device=pts.device)
    with torch.cuda.device(pts.device):
        st = _native.current_stream_ptr(pts.device)
        _native.check(lib.flooder_fps_indexed_f32(
            _native.ptr(pts), n, dim, dim, _native.ptr(index.pts), _native.ptr(index.order32), n_lms,
            int(start_idx), _native.ptr(out_idx), _native.ptr(rows), _native.ptr(box), _native.ptr(key),
            _native.ptr(work_best), st), "flooder_fps_indexed_f32")
    return out_idx


# ------------------------------------------------------------------------------ complex
def _build_complex(landmarks: torch.Tensor, max_dimension: int):
    """Delaunay triangulation of the landmarks -> (simplex tree, simplices bucketed by dimension).

    ``core.py:130-138``.  With gudhi installed the tree is a ``gudhi.SimplexTree`` built by
    ``gudhi.DelaunayComplex`` exactly as in the reference; otherwise Qhull and the array-backed
    ``flooder_amd.SimplexTree``.
    """
    lm = landmarks.detach().cpu().numpy()
    if HAS_GUDHI:  # (gudhi is absent from the build image: exercised with a stub module, tests/test_host.py)
        import gudhi

        stree = gudhi.DelaunayComplex(lm).create_simplex_tree()
        buckets: List[List[Tuple[int, ...]]] = [[] for _ in range(max_dimension + 1)]
        for simplex, _ in stree.get_simplices():
            if len(simplex) <= max_dimension + 1:
                buckets[len(simplex) - 1].append(tuple(simplex))
        simplices = [np.array(b, dtype=np.int64).reshape(-1, d + 1) for d, b in enumerate(buckets)]
        return stree, simplices
    # Qhull cells; the face tables up to max_dimension now, the higher ones when somebody asks for them
    stree = SimplexTree.from_cells(delaunay_cells(lm), lm.shape[0], eager=max_dimension, trusted=True)
    simplices = [stree.simplices_of_dimension(d) for d in range(max_dimension + 1)]
    return stree, simplices


def _ball_prep(simplex_vertices: torch.Tensor, d: int):
    """Bounding balls (``core.py:156-172``): centre = midpoint of the longest edge; radius =
    max vertex distance x (1.42 if d > 1 else 1.01) + 1e-3."""
    n = simplex_vertices.shape[0]
    flat = torch.argmax(torch.cdist(simplex_vertices, simplex_vertices).flatten(1), dim=1)
    i0 = torch.div(flat, d + 1, rounding_mode="floor")
    i1 = flat - i0 * (d + 1)
    ar = torch.arange(n, device=simplex_vertices.device)
    centers = (simplex_vertices[ar, i0] + simplex_vertices[ar, i1]) / 2.0
    radii = torch.amax((simplex_vertices - centers.unsqueeze(1)).norm(dim=2), dim=1) \
        * (1.42 if d > 1 else 1.01) + 1e-3
    return centers, radii


_GRID_CACHE: Dict[tuple, tuple] = {}


def _grid_tables(points_per_edge: int, dim: int, device, dtype):
    """``generate_grid`` + face table + sample plan of one (points_per_edge, dimension): identical from call to
    call, 2 - 3 ms of host work and a dozen small uploads each time - kept per device."""
    key = (int(points_per_edge), int(dim), str(device), dtype, SAMPLE_UNITS)
    hit = _GRID_CACHE.get(key)
    if hit is None:
        weights, vertex_idxs, face_idxs = generate_grid(points_per_edge, dim, device, dtype)
        faces = _FaceTable(face_idxs, weights.shape[0], device)
        plan = SamplePlan(weights, faces) if torch.device(device).type == "cuda" else None
        v_np = [v.cpu().numpy() for v in vertex_idxs]
        if len(_GRID_CACHE) >= 16:
            _GRID_CACHE.clear()
        hit = _GRID_CACHE[key] = (weights, vertex_idxs, face_idxs, faces, plan, v_np)
    return hit


class _FaceTable:
    """CSR of grid rows per face, concatenated over codimensions, for ``flooder_face_max_f32``."""

    def __init__(self, face_idxs: Optional[List[torch.Tensor]], R: int, device):
        if face_idxs is None:  # random mode: one "face" = all rows (core.py:270)
            ptr = np.array([0, R], dtype=np.int32)
            rows = np.arange(R, dtype=np.int32)
            self.n_per_codim = [1]
        else:
            lens, rows_l = [], []
            self.n_per_codim = []
            for f in face_idxs:
                f = f.cpu().numpy()
                self.n_per_codim.append(f.shape[0])
                for row in f:
                    lens.append(len(row))
                    rows_l.append(row.astype(np.int32))
            ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            rows = np.concatenate(rows_l).astype(np.int32)
        self.n_faces = len(ptr) - 1
        self.ptr = torch.as_tensor(ptr, device=device)
        self.rows = torch.as_tensor(rows, device=device)


def _pad_rows(x: torch.Tensor, dp: int) -> torch.Tensor:
    if x.shape[1] == dp:
        return x.contiguous()
    out = torch.zeros((x.shape[0], dp), dtype=x.dtype, device=x.device)
    out[:, : x.shape[1]] = x
    return out


class SweepStats:
    """Work counters of the last ``flood_complex`` call on this process (for the benchmark)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.n_points = 0
        self.top_simplices = 0
        self.candidate_pairs = 0     # P  = sum_s |X n ball_s|
        self.pair_evals = 0          # P*R
        self.samples_per_simplex = 0
        self.slab_points = 0         # ball tests
        self.groups = 0
        self.deferred_chunks = 0     # chunks the run-of-four launch handed to the per-chunk launch
        self.dense_tiles = 0         # tiles of 64 samples the per-chunk launch handed to the tile launch
        self.hard_entries = (0, 0)   # tiles the finish gave to a whole workgroup (top pass, rest pass)


LAST_STATS = SweepStats()


class _KernelTimer:
    """Optional HIP-event timing of the individual kernels (events on the launch stream).

    ``chain=True``: a span starts at the LAST event this timer saw - the end of the span before it, or an event the
    caller recorded and handed over with ``note()`` (a step boundary) - instead of recording one of its own: spans that
    follow each other share their boundary.  An event record is a barrier packet of ~5 us on the stream; a step of
    four spans carries 5 of them instead of 9 (bench.py: 1.11 -> 1.09 ms at cfg 2).  What runs BETWEEN two spans is
    then counted with the later one (bench.py launches nothing there)."""

    def __init__(self, chain: bool = False):
        self.spans: Dict[str, List[Tuple[torch.cuda.Event, torch.cuda.Event]]] = {}
        self.chain = chain
        self.last: Optional[torch.cuda.Event] = None

    def note(self, event: Optional[torch.cuda.Event]):
        """``event`` (recorded by the caller on the launch stream; None: forget) is where the next span starts."""
        self.last = event

    def span(self, name: str):
        timer = self

        class _Span:
            def __enter__(self_inner):
                if timer.chain and timer.last is not None:
                    self_inner.a = timer.last
                else:
                    self_inner.a = torch.cuda.Event(enable_timing=True)
                    self_inner.a.record()
                self_inner.b = torch.cuda.Event(enable_timing=True)

            def __exit__(self_inner, *exc):
                self_inner.b.record()
                timer.last = self_inner.b
                timer.spans.setdefault(name, []).append((self_inner.a, self_inner.b))

        return _Span()

    def totals_ms(self) -> Dict[str, float]:
        """Sum of elapsed ms per kernel name (call after a device synchronize)."""
        return {k: float(sum(a.elapsed_time(b) for a, b in v)) for k, v in self.spans.items()}

    def counts(self) -> Dict[str, int]:
        return {k: len(v) for k, v in self.spans.items()}


class _NullSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _span(timer: Optional[_KernelTimer], name: str):
    return timer.span(name) if timer is not None else _NullSpan()


def _sweep_dimension_hip(pts_pad: torch.Tensor, search: torch.Tensor, axis: int, dim: int,
                         verts: torch.Tensor, centers: torch.Tensor, radii: torch.Tensor,
                         weights: torch.Tensor, faces: _FaceTable,
                         reduce_hook: Optional[Callable[[torch.Tensor], None]],
                         want_dist: bool = False, timer: Optional[_KernelTimer] = None):
    """All simplices of one dimension against one (sorted, padded) point set -> (S, F) face maxima.

    Steps: per-simplex slab by searchsorted (core.py:201-208) -> ball count -> offsets (cumsum) ->
    candidate fill -> sweep (atomic-min into d2 bits) -> [reduce_hook: cross-shard MIN] -> face max.
    One device->host copy (three totals) sizes the candidate workspace.
    """
    lib = _native.load()
    dev = pts_pad.device
    st = _native.current_stream_ptr(dev)
    S, k1, _ = verts.shape
    R = weights.shape[0]
    dp = pts_pad.shape[1]
    n = pts_pad.shape[0]
    if S == 0:  # (a simplex shard may be empty)
        return (torch.empty((0, faces.n_faces), dtype=torch.float32, device=dev),
                torch.empty((0, R), dtype=torch.float32, device=dev) if want_dist else None)

    verts = verts.to(torch.float32).contiguous()
    centers = centers.to(torch.float32).contiguous()
    radii = radii.to(torch.float32).contiguous()
    weights = weights.to(torch.float32).contiguous()

    lo = torch.searchsorted(search, (centers[:, axis] - radii).contiguous(), right=False)
    hi = torch.searchsorted(search, (centers[:, axis] + radii).contiguous(), right=True)
    counts = torch.zeros(S, dtype=torch.int32, device=dev)
    with _span(timer, "ball_count"):
        _native.check(lib.flooder_ball_count_f32(_native.ptr(pts_pad), n, dim, dp, _native.ptr(centers),
                                                 _native.ptr(radii), _native.ptr(lo), _native.ptr(hi), S,
                                                 _native.ptr(counts), st), "flooder_ball_count_f32")
    tiles = (R + TILE_SAMPLES - 1) // TILE_SAMPLES
    c64 = counts.to(torch.int64)
    padded = (c64 + (CAND_ALIGN - 1)) // CAND_ALIGN * CAND_ALIGN
    items = ((c64 + (SWEEP_CHUNK - 1)) // SWEEP_CHUNK) * tiles

    d2 = torch.empty((S, R), dtype=torch.int32, device=dev)
    _native.check(lib.flooder_fill_u32(_native.ptr(d2), S * R, INF_BITS, st), "flooder_fill_u32")

    csum = torch.cumsum(padded, 0)
    totals = torch.stack([csum[-1], c64.sum(), (hi - lo).sum()]).cpu().tolist()  # the one sync
    total_rows = int(totals[0])
    LAST_STATS.candidate_pairs += int(totals[1])
    LAST_STATS.slab_points += int(totals[2])

    # group simplices so that each group's candidate workspace fits the budget
    rows_budget = max(CAND_WORKSPACE_BYTES // (4 * dp), 1 << 20)
    bounds = [0]
    group_rows: List[int] = []
    if total_rows > rows_budget:
        csum_h = csum.cpu().numpy()
        start_rows = 0
        while bounds[-1] < S:
            e = int(np.searchsorted(csum_h, start_rows + rows_budget, side="right"))
            e = min(max(e, bounds[-1] + 1), S)
            bounds.append(e)
            group_rows.append(int(csum_h[e - 1]) - start_rows)
            start_rows = int(csum_h[e - 1])
    else:
        bounds.append(S)
        group_rows.append(total_rows)

    for b, e, n_rows in zip(bounds[:-1], bounds[1:], group_rows):
        ns = e - b
        cand_off = torch.zeros(ns + 1, dtype=torch.int64, device=dev)
        torch.cumsum(padded[b:e], 0, out=cand_off[1:])
        item_prefix = torch.zeros(ns + 1, dtype=torch.int64, device=dev)
        torch.cumsum(items[b:e], 0, out=item_prefix[1:])
        cand = torch.empty((max(n_rows, 1), dp), dtype=torch.float32, device=dev)
        cursor = torch.zeros(ns + 1, dtype=torch.int32, device=dev)  # [0:ns] fill cursors, [ns] sweep queue
        g_counts = counts[b:e]
        with _span(timer, "ball_fill"):
            _native.check(lib.flooder_ball_fill_f32(
                _native.ptr(pts_pad), n, dim, dp, _native.ptr(centers[b:e]), _native.ptr(radii[b:e]),
                _native.ptr(lo[b:e]), _native.ptr(hi[b:e]), ns, _native.ptr(g_counts),
                _native.ptr(cand_off), _native.ptr(cursor), _native.ptr(cand), st), "flooder_ball_fill_f32")
        with _span(timer, "sweep"):
            _native.check(lib.flooder_sweep_f32(
                _native.ptr(cand), _native.ptr(cand_off), _native.ptr(g_counts), dim,
                _native.ptr(verts[b:e]), _native.ptr(weights), k1, R, ns, _native.ptr(item_prefix),
                cursor[ns:].data_ptr(), d2[b:e].data_ptr(), st), "flooder_sweep_f32")
        LAST_STATS.groups += 1
        del cand

    if reduce_hook is not None:
        with _span(timer, "reduce"):
            reduce_hook(d2)

    out_face = torch.empty((S, faces.n_faces), dtype=torch.float32, device=dev)
    out_dist = torch.empty((S, R), dtype=torch.float32, device=dev) if want_dist else None
    with _span(timer, "face_max"):
        _native.check(lib.flooder_face_max_f32(_native.ptr(d2), S, R, _native.ptr(faces.ptr),
                                               _native.ptr(faces.rows), faces.n_faces,
                                               _native.ptr(out_face), _native.ptr(out_dist), st),
                      "flooder_face_max_f32")
    return out_face, out_dist


def cloud_box(points: torch.Tensor) -> torch.Tensor:
    """Bounding box of a ROCm point tensor computed on the device: 16 floats, [0:dim] = min,
    [8:8+dim] = max (one pass over the cloud; torch's column min/max takes ~1 ms per million points)."""
    lib = _native.load()
    pts32 = points.detach().to(torch.float32).contiguous()
    n, dim = pts32.shape
    box = torch.empty(16, dtype=torch.float32, device=pts32.device)
    partial = torch.empty(1024 * 16, dtype=torch.float32, device=pts32.device)
    _native.check(lib.flooder_bbox_f32(_native.ptr(pts32), n, dim, dim, _native.ptr(box), _native.ptr(partial),
                                       _native.current_stream_ptr(pts32.device)), "flooder_bbox_f32")
    return box


# point index above this dimension: rows in k-d tree order instead of curve order (8 = never)
KD_ORDER_ABOVE_DIM = 3


def _tensor_version(t: torch.Tensor):
    """In-place modification counter of a tensor, or None where torch keeps none (tensors created under
    ``torch.inference_mode()`` raise on ``_version``): the staleness check of ``index=`` is skipped for those."""
    try:
        return None if t.is_inference() else t._version
    except RuntimeError:
        return None


# Cross-call reuse of a PointIndex.  OFF by default (the reference keeps no state between calls): the explicit ways
# are ``lms, idx = generate_landmarks(points, n, return_index=True); flood_complex(points, lms, index=idx)`` and
# ``flood_complex(points, <int>)``, where the library builds the index once for both steps itself.  With
# ``INDEX_CACHE = True`` (or FLOODER_INDEX_CACHE=1 in the environment) the last index built from a caller's tensor is
# remembered by the IDENTITY of that tensor and its version counter, so that plain
#     lms = generate_landmarks(points, 1000); flood_complex(points, lms)
# builds it once.  CAVEAT of that mode, why it is opt-in: torch's ``_version`` does not move when the buffer is
# written through ``.data``, by a raw-pointer kernel or through a DLPack / cupy alias - such a write is not seen and
# the stale sorted copy would be swept.  The entry holds the index strongly but dies WITH the tensor (weakref
# callback): a freed cloud's 16 B / point copy does not stay in HBM.
INDEX_CACHE = os.environ.get("FLOODER_INDEX_CACHE", "0") not in ("", "0", "false", "False")
_LAST_INDEX: List = [None, None, None]   # weakref to the source tensor, its version, the PointIndex


def _index_owner_died(ref) -> None:
    if _LAST_INDEX[0] is ref:
        _LAST_INDEX[:] = [None, None, None]


def _remember_index(points: torch.Tensor, index: "PointIndex") -> None:
    ver = _tensor_version(points)
    if INDEX_CACHE and ver is not None:
        try:
            _LAST_INDEX[:] = [weakref.ref(points, _index_owner_died), ver, index]
        except TypeError:
            _LAST_INDEX[:] = [None, None, None]


def _recall_index(points: torch.Tensor) -> Optional["PointIndex"]:
    ref, ver, index = _LAST_INDEX
    if (INDEX_CACHE and ref is not None and ref() is points and ver == _tensor_version(points)
            and (index.n, index.dim) == tuple(points.shape) and index.pts.device == points.device
            and index.kd == (index.dim > KD_ORDER_ABOVE_DIM and index.n > BVH_LEAF)):   # (the row order asked for now)
        return index
    return None


CHECK_RANK_CONSISTENCY = True   # sharded runs: one 6-word MIN before the first face collective of a dimension pass


def _assert_ranks_agree(min_hook: Callable[[torch.Tensor], None], device, *counts: int) -> None:
    """Every rank passes the same ``counts``?  ``min_hook`` is the caller's in-place elementwise MIN over the ranks:
    MIN of (c, -c) gives (min c, -max c), equal in magnitude only if all ranks hold the same c."""
    c = torch.tensor([x for v in counts for x in (int(v), -int(v))], dtype=torch.int64, device=device)
    min_hook(c)
    got = c.cpu().tolist()
    if any(got[i] != -got[i + 1] for i in range(0, len(got), 2)):
        raise RuntimeError("flood_complex: the ranks of this sharded run do not hold the same complex "
                           f"(simplices / faces per simplex / face slots: min {got[0::2]}, max {[-x for x in got[1::2]]}) - "
                           "different landmarks, or a different Delaunay code path on some rank (native library "
                           "missing on one of them?)")


def forget_index() -> None:
    """Drop the remembered PointIndex (``INDEX_CACHE`` mode; it keeps a padded, sorted copy of the last cloud alive
    for as long as that tensor lives: 16 B per 3-D point)."""
    _LAST_INDEX[:] = [None, None, None]


SWEEP_PREPARE_FUSED = True    # fused 2-D / 3-D sweep: zero fill + simplex weights + plane rows in one launch
SORT_STATE_BY_CALLER = True   # the index's radix sort on state zeroed by the curve-code kernel (no fill launches)


class PointIndex:
    """Copy of a point set sorted along a space-filling curve (Hilbert by default) plus its implicit box tree
    (HBM resident).

    ``pts``   (n_pad, DP) f32: rows in curve order, padded with +inf rows to a multiple of 16
    ``nodes`` (n_nodes, 2*DP) f32: box (lo, hi) of every node, leaves first (16 points each), then
              one level per factor 64, every level padded to a multiple of 64 nodes
    """

    def __init__(self, points: torch.Tensor, timer: Optional[_KernelTimer] = None, box: Optional[torch.Tensor] = None):
        lib = _native.load()
        dev = points.device
        st = _native.current_stream_ptr(dev)
        pts32 = points.detach().to(torch.float32).contiguous()
        n, dim = pts32.shape
        self.n, self.dim = n, dim
        # where the rows came from: flood_complex(index=...) refuses an index whose source tensor has been written to
        # in place since (same storage, other version counter); a different tensor of the same shape cannot be told
        # from a copy of the same cloud and is taken on the caller's word
        self.source = (points.data_ptr(), _tensor_version(points))
        self.dp = lib.flooder_padded_dim(dim)
        self.box = box if box is not None else cloud_box(pts32)  # device: [0:dim] min, [8:8+dim] max
        # sorted row -> original index (int32)
        self.order32 = torch.empty(n, dtype=torch.int32, device=dev)
        dens = None
        self.kd = dim > KD_ORDER_ABOVE_DIM and n > BVH_LEAF
        if self.kd:
            # above 3D: the order of a balanced k-d tree aligned with the box tree's groups (flood_index.hip) - a curve
            # cuts space on a fixed grid and the 1024-point nodes of a 6-D Hilbert order overlap each other
            tmp_bytes = int(lib.flooder_kd_order_bytes(n))
            if tmp_bytes < 0:
                raise RuntimeError("flooder_kd_order_bytes failed")
            tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
            with _span(timer, "sort"):
                _native.check(lib.flooder_kd_order_f32(_native.ptr(pts32), n, dim, dim, _native.ptr(self.order32),
                                                       _native.ptr(tmp), tmp_bytes, st), "flooder_kd_order_f32")
        else:
            codes = torch.empty(n, dtype=torch.int64, device=dev)
            key_bits = int(lib.flooder_curve_key_bits(dim))
            # one buffer the curve-code kernel zeroes on its way (no fill launch): the density grid of the cell sweep
            # and the state of the radix sort (its histograms, look-back arrays, block tickets: flooder_index_sort_zeroed)
            n_dens = int(lib.flooder_density_grid_words(dim)) if dim in (2, 3) and CELL_DENSITY_GRID else 0
            n_state = int(lib.flooder_index_sort_state_words(n, key_bits)) if SORT_STATE_BY_CALLER else 0
            zeroed = torch.empty(n_dens + n_state, dtype=torch.int32, device=dev) if n_dens + n_state else None
            dens = zeroed[:n_dens] if n_dens else None
            sort_state = zeroed[n_dens:] if n_state else None
            with _span(timer, "morton"):
                _native.check(lib.flooder_morton_zero_f32(_native.ptr(pts32), n, dim, dim, _native.ptr(self.box),
                                                          _native.ptr(codes), _native.ptr(zeroed),
                                                          0 if zeroed is None else zeroed.numel(), st), "flooder_morton_zero_f32")
            # radix sort over the bits the codes use
            codes_sorted = torch.empty(n, dtype=torch.int64, device=dev)
            tmp_bytes = int(lib.flooder_index_sort_bytes(n))
            if tmp_bytes < 0:
                raise RuntimeError("flooder_index_sort_bytes failed")
            tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
            with _span(timer, "sort"):
                if sort_state is not None:
                    _native.check(lib.flooder_index_sort_zeroed(_native.ptr(codes), n, key_bits, _native.ptr(codes_sorted),
                                                                _native.ptr(self.order32), _native.ptr(tmp), tmp_bytes,
                                                                _native.ptr(sort_state), st), "flooder_index_sort_zeroed")
                else:
                    _native.check(lib.flooder_index_sort(_native.ptr(codes), n, key_bits, _native.ptr(codes_sorted),
                                                         _native.ptr(self.order32), _native.ptr(tmp), tmp_bytes, st),
                                  "flooder_index_sort")
        n_pad = (n + BVH_LEAF - 1) // BVH_LEAF * BVH_LEAF
        self.pts = torch.empty((n_pad, self.dp), dtype=torch.float32, device=dev)
        n_nodes = int(lib.flooder_bvh_node_count(n))
        self.nodes = torch.empty((n_nodes, 2 * self.dp), dtype=torch.float32, device=dev)
        # density grid for the cell sweep (2D / 3D): point counts per cell, accumulated from the leaf boxes in the same pass
        self.dens = dens
        if self.dens is None and dim in (2, 3) and CELL_DENSITY_GRID:
            self.dens = torch.zeros(int(lib.flooder_density_grid_words(dim)), dtype=torch.int32, device=dev)
        with _span(timer, "bvh_build"):  # rows in curve order + leaf boxes in one pass, then the inner levels
            _native.check(lib.flooder_index_rows_f32(_native.ptr(pts32), n, dim, dim, _native.ptr(self.order32),
                                                     _native.ptr(self.pts), n_pad, _native.ptr(self.nodes),
                                                     _native.ptr(self.dens), _native.ptr(self.box), st),
                          "flooder_index_rows_f32")


def index_from_host(points_cpu: torch.Tensor, device, chunk_rows: int = 1 << 21):
    """``PointIndex`` of a cloud that lives in HOST memory, streamed to the GPU in chunks (BASELINE.json configs[4]:
    "chunked point streaming from host pinned memory"; the reference bounds its device working set with
    ``batch_size`` slabs instead, ``core.py:193-217``).  The cloud is copied chunk by chunk from pinned memory on a
    copy stream (``points_cpu`` is pinned first if it is not: one extra host copy) while the bounding-box reduction of
    the chunks already landed runs on the compute stream; the curve codes need the box of the WHOLE cloud, so the
    sort starts when the last chunk is in.  Returns ``(index, points_dev)``: pass both on -
    ``flood_complex(points_dev, landmarks, index=index)``.  ``h2d_ms_of(index)`` = time from the first copy to
    the box being known (events on the compute stream)."""
    lib = _native.load()
    dev = torch.device(device)
    src = points_cpu.detach().to(torch.float32).contiguous()
    if src.device.type != "cpu":
        raise ValueError("index_from_host: points_cpu must be a CPU tensor")
    if not src.is_pinned():
        src = src.pin_memory()
    n, dim = src.shape
    if dim > 8 or n < 1:
        raise RuntimeError("flooder_amd: index_from_host needs 1 <= dim <= 8 and at least one point")
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        copy = torch.cuda.Stream(dev)
        raw = torch.empty((n, dim), dtype=torch.float32, device=dev)
        n_chunks = (n + chunk_rows - 1) // chunk_rows
        blocks = 64
        partial = torch.empty(n_chunks * blocks * 16, dtype=torch.float32, device=dev)
        box = torch.empty(16, dtype=torch.float32, device=dev)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(main)
        copy.wait_stream(main)
        for c in range(n_chunks):
            a, b = c * chunk_rows, min(n, (c + 1) * chunk_rows)
            with torch.cuda.stream(copy):
                raw[a:b].copy_(src[a:b], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy)
            main.wait_event(ev)
            _native.check(lib.flooder_bbox_chunk_f32(raw[a:b].data_ptr(), b - a, dim, dim,
                                                     partial[c * blocks * 16:].data_ptr(), blocks, main.cuda_stream),
                          "flooder_bbox_chunk_f32")
        _native.check(lib.flooder_bbox_reduce_f32(_native.ptr(partial), n_chunks * blocks, dim, _native.ptr(box),
                                                  main.cuda_stream), "flooder_bbox_reduce_f32")
        t1.record(main)
        index = PointIndex(raw, box=box)
        raw.record_stream(copy)
        index._h2d_events = (t0, t1)
    return index, raw


def h2d_ms_of(index: "PointIndex") -> Optional[float]:
    """Milliseconds from the first chunk's copy to the bounding box of a cloud streamed by ``index_from_host``
    (synchronises the device)."""
    ev = getattr(index, "_h2d_events", None)
    if ev is None:
        return None
    torch.cuda.synchronize()
    return float(ev[0].elapsed_time(ev[1]))


def _row_hash(x: torch.Tensor) -> torch.Tensor:
    """64-bit mix of the float32 bit patterns of every row (-0.0 folded onto 0.0): equal rows, equal hashes."""
    b = (x.to(torch.float32) + 0.0).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    h = torch.full((x.shape[0],), 0x2545F4914F6CDD1D, dtype=torch.int64, device=x.device)
    for k in range(x.shape[1]):
        h = (h ^ (b[:, k] + 0x632BE59BD9B4E019 + (h << 6) + (h >> 2))) * 0x100000001B3
    return h


def _rows_are_subset(rows: torch.Tensor, points32: torch.Tensor) -> bool:
    """Is every row of ``rows`` bit-equal to some row of ``points32``?  (Row hashes through ``torch.isin``; a false
    'yes' needs a 64-bit collision.)  One host synchronisation."""
    if rows.shape[0] == 0:
        return True
    return bool(torch.isin(_row_hash(rows), _row_hash(points32)).all().item())


def block_subcloud(points32: torch.Tensor, verts: torch.Tensor, d: int, box: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The part of the cloud a block of ``d``-simplices (``verts``: (S, d+1, dim)) can see when the landmarks are
    points of the cloud: every nearest neighbour of a sample lies in its simplex's bounding ball (the reference's own
    pruning bound, ``core.py:156-172``: centre = midpoint of the longest edge, radius = 1.42 x the largest vertex
    distance + 1e-3).  The rows inside the union of the balls - taken on a coarse grid over the cloud's box ``box``
    (2D / 3D; other dimensions: the balls' common bounding box) - are compacted by ``flooder_box_select_f32`` (any
    order).  One host synchronisation (the row count).  A rank of a block-sharded run builds its ``PointIndex`` over
    this sub-cloud instead of the whole cloud."""
    lib = _native.load()
    dev = points32.device
    centers, radii = _ball_prep(verts.to(torch.float32), d)
    centers, radii = centers.contiguous(), radii.contiguous()
    lo = (centers - radii[:, None]).min(dim=0).values
    hi = (centers + radii[:, None]).max(dim=0).values
    bbox = torch.cat([lo, hi]).to(torch.float32).contiguous()
    pts = points32.contiguous()
    n, dim = pts.shape
    out = torch.empty_like(pts)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    grid_bytes = int(lib.flooder_select_grid_bytes(dim))
    flags = torch.zeros(grid_bytes, dtype=torch.uint8, device=dev) if grid_bytes > 0 else None
    cbox = (box if box is not None else cloud_box(pts)) if flags is not None else None
    _native.check(lib.flooder_box_select_f32(_native.ptr(pts), n, dim, dim, _native.ptr(bbox), _native.ptr(cbox),
                                             _native.ptr(centers), _native.ptr(radii), centers.shape[0],
                                             _native.ptr(flags), _native.ptr(out), _native.ptr(count),
                                             _native.current_stream_ptr(dev)), "flooder_box_select_f32")
    return out[:int(count.item())]


CPU_WORKERS = -1      # threads of the CPU branch's kd-tree query (scipy: -1 = all cores; the reference's call uses 1)
SHARD_DEAL = "stride"   # how mode="simplices" cuts the queue: "stride" (every world-th row), "size" (largest simplices first, dealt round-robin: measured, round 6 - an eighth of cfg 3 1.48 -> 1.42 ms, of cfg 5 2.97 -> 2.93, of cfg 2 0.59 -> 0.62: the slow rank is the one that holds THE hardest tile, whichever deal hands it over)


def simplex_share(verts, rank: int, world: int, deal: Optional[str] = None) -> np.ndarray:
    """Rows of the simplex queue (``verts``: (S, k1, dim) vertex coordinates in queue order, array or tensor) that rank
    ``rank`` of ``world`` sweeps, ascending.  ``"stride"`` (default): every ``world``-th row.  ``"size"``: the
    simplices sorted by the squared diagonal of their bounding box, largest first, and dealt round-robin - the work of
    a simplex is heavy-tailed and its tail are the LARGE simplices (they span the voids of the cloud: their samples lie
    far from every point, their tiles are the finish's long searches), so each rank gets its share of them instead of
    what the stride happens to hit.  Computed from the landmark coordinates alone, in float64 with a stable sort:
    the same on every rank; any partition gives the same values."""
    deal = SHARD_DEAL if deal is None else deal
    n = int(verts.shape[0])
    if deal == "stride" or world <= 1 or n == 0:
        return np.arange(rank, n, max(world, 1), dtype=np.int64)
    if deal != "size":
        raise ValueError("deal must be 'size' or 'stride'")
    v = verts.detach().cpu().numpy() if isinstance(verts, torch.Tensor) else np.asarray(verts)
    v = v.astype(np.float64, copy=False)
    key = ((v.max(axis=1) - v.min(axis=1)) ** 2).sum(axis=1)
    order = np.argsort(-key, kind="stable")
    return np.sort(order[rank::world]).astype(np.int64)


def shared_face_slots(stree, d: int, order_np: np.ndarray, v_idx_np: List[np.ndarray], device):
    """Slots of the fused face maxima with ONE word per distinct face of the complex: for the top cells of a
    ``SimplexTree.from_cells`` complex (in the order ``order_np`` of the sweep) and the face numbering of
    ``generate_grid`` (``v_idx_np[k]``: vertex positions of the codimension-k faces).  Returns
    ``(slot (S, F) int32 tensor, n_slots, offsets)`` with ``offsets[k]`` = first slot of the faces of codimension k
    (slot - offsets[k] = row of the dimension d-k table), or None when the tree has no cell -> face index."""
    if not isinstance(stree, SimplexTree) or stree._cells is None or stree._cells.shape[1] != d + 1:
        return None
    cols, offsets, total = [], [], 0
    for v_idx in v_idx_np:
        nf, k = v_idx.shape
        index = stree.cell_face_index(k - 1)
        if index is None:
            return None
        combos = list(itertools.combinations(range(d + 1), k))
        pick = [combos.index(tuple(int(x) for x in row)) for row in v_idx]
        cols.append(index[order_np][:, pick] + total)
        offsets.append(total)
        total += stree.simplices_of_dimension(k - 1).shape[0]
    slot = np.ascontiguousarray(np.concatenate(cols, axis=1).astype(np.int32))
    return torch.as_tensor(slot, device=device), total, offsets


def shard_slot_fill(slot: torch.Tensor, n_slots: int) -> torch.Tensor:
    """For a rank that sweeps only the simplices of ``slot`` (its rows of the ``shared_face_slots`` table): the
    (n_slots,) vector with 0 in the words its simplices touch and +inf in the others - ``torch.maximum(values, fill)``
    is what the rank contributes to ``all_reduce(MIN)`` (values are non-negative)."""
    fill = torch.full((n_slots,), float("inf"), dtype=torch.float32, device=slot.device)
    if slot.numel():
        fill[slot.reshape(-1).long()] = 0.0
    return fill


def simplex_order(index: "PointIndex", verts: torch.Tensor) -> torch.Tensor:
    """Work order of the simplices for the culled sweeps: descending estimated number of cloud points inside the
    simplex's bounding box (``flooder_simplex_weight_f32``).  Work per simplex is heavy-tailed - a tetrahedron in
    the dense core of a Gaussian cloud takes 100x the median - and the persistent kernels pop simplices in queue
    order, so the long ones go first and the short ones fill the tail.  Device-side, no synchronisation."""
    lib = _native.load()
    verts = verts.to(torch.float32).contiguous()
    S, k1, _ = verts.shape
    wgt = torch.empty(S, dtype=torch.float32, device=verts.device)
    _native.check(lib.flooder_simplex_weight_f32(_native.ptr(index.nodes), index.n, index.dim, _native.ptr(verts), k1, S,
                                                 _native.ptr(wgt), _native.current_stream_ptr(verts.device)),
                  "flooder_simplex_weight_f32")
    return torch.argsort(wgt, descending=True)


def sample_order(weights: torch.Tensor) -> np.ndarray:
    """Permutation that groups the barycentric samples into compact patches: recursive bisection along the
    widest axis (coordinates of the regular simplex), cutting at multiples of 256, then 64, then 16 samples.
    Every aligned run of 256 / 64 consecutive samples - a chunk of the cell sweep, a tile of the tree sweep -
    then has a tight bounding box in any affine image of the simplex (chunk boxes about 2.5x smaller in total
    and 4x smaller at worst than along a Morton curve of the weights, which jumps across the simplex)."""
    w = weights.detach().cpu().numpy().astype(np.float64)
    R, k1 = w.shape
    if k1 <= 1 or R <= 64:
        return np.arange(R, dtype=np.int64)
    key = (R, k1, hash(w.tobytes()), SAMPLE_UNITS)       # grid tables repeat from call to call: 5 ms saved
    hit = _SAMPLE_ORDER_CACHE.get(key)
    if hit is not None:
        return hit.copy()
    order = _sample_order_bisect(w)
    if len(_SAMPLE_ORDER_CACHE) >= 32:
        _SAMPLE_ORDER_CACHE.clear()
    _SAMPLE_ORDER_CACHE[key] = order
    return order.copy()


_SAMPLE_ORDER_CACHE: Dict[tuple, np.ndarray] = {}
# cut sizes of the recursive bisection, coarse to fine: every aligned run of that many samples is a compact patch
# (1024 = a run of four chunks of the cell sweep, 256 = a chunk, 64 = a tile of the tree sweeps)
SAMPLE_UNITS = (1024, 256, 64, 16)


def _sample_order_bisect(w: np.ndarray) -> np.ndarray:
    R, k1 = w.shape
    nd = k1 - 1
    corners = np.eye(k1) - 1.0 / k1                      # regular simplex, centred
    basis = np.linalg.qr(corners.T)[0][:, :nd]           # orthonormal basis of its hyperplane
    X = w @ (corners @ basis)
    units = SAMPLE_UNITS
    out: List[np.ndarray] = []
    stack = [np.arange(R, dtype=np.int64)]
    while stack:
        idx = stack.pop()
        n = idx.shape[0]
        if n <= units[-1]:
            out.append(idx)
            continue
        P = X[idx]
        ax = int(np.argmax(P.max(axis=0) - P.min(axis=0)))
        idx = idx[np.argsort(P[:, ax], kind="stable")]
        unit = next((u for u in units if n > u), units[-1])
        n_left = (-(-n // unit) // 2) * unit
        if n_left <= 0 or n_left >= n:
            n_left = n // 2
        stack.append(idx[n_left:])   # (popped after the left part: the output keeps left-to-right order)
        stack.append(idx[:n_left])
    return np.concatenate(out)


WIT_MAX_COARSE = 256     # FLOODER_WIT_MAX_COARSE
WIT_MAX_ROWS = 8192      # FLOODER_WIT_MAX_ROWS
WIT_MIN_ROWS = 512       # below this a simplex is too small for a coarse level to pay
_WIT_PLAN_CACHE: Dict[tuple, Optional[tuple]] = {}


def witness_plan(weights: torch.Tensor, perm: np.ndarray) -> Optional[Tuple[np.ndarray, np.ndarray]]:
    """Coarse level of the witness sweep (csrc/flood_wit.hip) for one weight table: ``(coarse_rows, parents)`` in
    SWEEP order (``perm``: sweep position -> original row).  ``coarse_rows``: the positions of at most
    ``WIT_MAX_COARSE`` coarse samples, padded with -1; ``parents``: per position four coarse slots (8 bits each) - the
    coarse samples nearest to it in the regular-simplex embedding of the weights; a coarse sample comes first among
    its own parents.  Lattice weights (``generate_grid``): on every face of the simplex the lattice points whose
    barycentric numerators are congruent to 1 modulo M in all but the last coordinate of the face, M the smallest
    stride that fits; other weight tables (``generate_uniform_weights``): farthest-point samples of the rows.
    Any choice is valid - the bound of a sample is the distance to a real point of the cloud whoever found it -; it
    only decides how many samples survive the bound.  None: table too small or too large for the kernel."""
    w = weights.detach().cpu().numpy().astype(np.float64)
    R, k1 = w.shape
    if R < WIT_MIN_ROWS or R > WIT_MAX_ROWS or k1 < 2:
        return None
    key = (R, k1, hash(w.tobytes()), hash(np.asarray(perm).tobytes()))
    if key in _WIT_PLAN_CACHE:
        return _WIT_PLAN_CACHE[key]
    from scipy.spatial import cKDTree

    corners = np.eye(k1) - 1.0 / k1
    basis = np.linalg.qr(corners.T)[0][:, :k1 - 1]
    X = w @ (corners @ basis)
    pos = w[w > 1e-9]
    m = int(round(1.0 / pos.min())) if pos.size else 0
    lat = np.rint(w * m).astype(np.int64) if m > 0 else None
    coarse = None
    if lat is not None and m <= 4096 and np.abs(w * m - lat).max() < 1e-3 and (lat.sum(axis=1) == m).all():
        nz = lat > 0
        last = (k1 - 1) - np.argmax(nz[:, ::-1], axis=1)          # last non-zero coordinate of every row
        free = nz & (np.arange(k1)[None, :] != last[:, None])    # ... the others decide
        for M in range(2, 65):
            ok = np.where(free, lat % M == 1 % M, True).all(axis=1)
            if ok.sum() <= WIT_MAX_COARSE:
                coarse = np.nonzero(ok)[0]
                break
    if coarse is None or coarse.size < 4:   # no lattice: farthest-point samples
        n_c = int(min(WIT_MAX_COARSE, max(16, R // 16)))
        dmin = np.full(R, np.inf)
        cur, picked = 0, []
        for _ in range(n_c):
            picked.append(cur)
            dmin = np.minimum(dmin, ((X - X[cur]) ** 2).sum(axis=1))
            cur = int(np.argmax(dmin))
        coarse = np.array(sorted(set(picked)), dtype=np.int64)
    inv = np.empty(R, dtype=np.int64)
    inv[np.asarray(perm)] = np.arange(R)
    coarse = coarse[np.argsort(inv[coarse], kind="stable")]        # slots in sweep order
    kq = min(4, coarse.size)
    _, nn = cKDTree(X[coarse]).query(X, k=kq)
    nn = nn.reshape(R, kq)
    if kq < 4:
        nn = np.concatenate([nn] + [nn[:, :1]] * (4 - kq), axis=1)
    slot_of = np.full(R, -1, dtype=np.int64)
    slot_of[coarse] = np.arange(coarse.size)
    own = slot_of >= 0                                             # (ties aside, a coarse row is its own nearest)
    nn[own, 0] = slot_of[own]
    par = (nn[:, 0] | (nn[:, 1] << 8) | (nn[:, 2] << 16) | (nn[:, 3] << 24)).astype(np.uint32)
    rows = np.full(WIT_MAX_COARSE, -1, dtype=np.int32)
    rows[:coarse.size] = inv[coarse]
    out = (rows, np.ascontiguousarray(par[np.asarray(perm)]), int(coarse.size))
    if len(_WIT_PLAN_CACHE) >= 32:
        _WIT_PLAN_CACHE.clear()
    _WIT_PLAN_CACHE[key] = out
    return out


class SamplePlan:
    """Device-resident sample weights in sweep order (``sample_order``) plus the face table remapped to that order
    and, per row, the bit mask of the faces it lies on (the fused face maxima); built once per dimension pass,
    independent of the simplices."""

    def __init__(self, weights: torch.Tensor, faces: _FaceTable):
        dev = weights.device
        self.R, self.k1 = weights.shape
        R = self.R
        perm = sample_order(weights)                      # positions -> original rows, spatially coherent
        f_ptr = faces.ptr.cpu().numpy().astype(np.int64)
        f_rows = faces.rows.cpu().numpy().astype(np.int64)
        n_faces = faces.n_faces
        inv = np.empty(R, dtype=np.int64)
        inv[perm] = np.arange(R)
        self.inv = torch.as_tensor(inv, device=dev)
        self.w_perm = weights.to(torch.float32)[torch.as_tensor(perm, device=dev)].contiguous()
        self.rows_perm = self.inv[faces.rows.long()].to(torch.int32).contiguous()
        self.faces = faces
        self.memb_all = None
        if n_faces <= 32:
            memb_all = np.zeros(R, dtype=np.uint32)
            for f in range(n_faces):
                memb_all[inv[f_rows[f_ptr[f]:f_ptr[f + 1]]]] |= np.uint32(1 << f)
            self.memb_all = torch.as_tensor(memb_all.view(np.int32), device=dev)
        self._perm, self._weights, self._face_rows = perm, weights, (f_ptr, f_rows, inv)
        self._late_rows = self._wit = None
        self._late_built = self._wit_built = False

    # The two tables below cost host work (a device sync for the weights, a cKDTree / farthest-point pass, one loop
    # per face) and only one sweep each reads them: they are built when that sweep first asks - a ``num_rand`` call,
    # whose weight table is new every time, never pays for a witness plan it cannot use.
    @property
    def late_rows(self):
        """fused sorted sweep (above 3D, ``SORTED_FUSED_FACES``): every row but one PILOT per face - the row nearest
        to the centre of the face's rows - sorts behind the pilots, whose values bring the face maxima close to final
        first.  None: more than 32 faces."""
        if not self._late_built:
            self._late_built = True
            if self.memb_all is not None:
                f_ptr, f_rows, inv = self._face_rows
                late = np.ones(self.R, dtype=np.uint8)
                wn = self._weights.detach().cpu().numpy().astype(np.float64)[self._perm]
                for f in range(self.faces.n_faces):
                    rows_f = inv[f_rows[f_ptr[f]:f_ptr[f + 1]]]
                    if rows_f.size > 2:
                        c = wn[rows_f].mean(axis=0)
                        late[rows_f[np.argmin(((wn[rows_f] - c) ** 2).sum(axis=1))]] = 0
                self._late_rows = torch.as_tensor(late, device=self._weights.device)
        return self._late_rows

    @property
    def wit(self):
        """coarse level of the witness sweep (``witness_plan``): (coarse rows, parents, number of coarse samples) on
        the device, or None where the table is not one the kernel takes."""
        if not self._wit_built:
            self._wit_built = True
            wp = witness_plan(self._weights, self._perm) if self.memb_all is not None else None
            if wp is not None:
                dev = self._weights.device
                self._wit = (torch.as_tensor(wp[0], device=dev), torch.as_tensor(wp[1].view(np.int32), device=dev), wp[2])
        return self._wit


def _sweep_dimension_f64(index: PointIndex, pts64_sorted: torch.Tensor, verts: torch.Tensor, weights: torch.Tensor,
                         faces: _FaceTable, reduce_hook: Optional[Callable[[torch.Tensor], None]],
                         want_dist: bool = False, timer: Optional[_KernelTimer] = None):
    """float64 inputs (the reference instantiates its kernels with DTYPE = fp64, triton_kernels.py:226-229): tree
    sweep in double over the float32 box tree of the cloud -> [reduce_hook on the (S, R) int64 bit patterns] ->
    face maxima + sqrt in double.  Returns (S, F) float64."""
    lib = _native.load()
    dev = index.pts.device
    st = _native.current_stream_ptr(dev)
    S, k1, _ = verts.shape
    R = weights.shape[0]
    verts = verts.to(torch.float64).contiguous()
    order = torch.as_tensor(sample_order(weights), device=dev)
    w_perm = weights.to(torch.float64)[order].contiguous()
    inv = torch.empty_like(order)
    inv[order] = torch.arange(R, device=dev)
    rows_perm = inv[faces.rows.long()].to(torch.int32).contiguous()
    d2 = torch.empty((S, R), dtype=torch.int64, device=dev)
    queue = torch.zeros(QUEUE_WORDS, dtype=torch.int32, device=dev)   # sharded work-queue heads
    with _span(timer, "sweep"):
        _native.check(lib.flooder_sweep_bvh_f64(_native.ptr(pts64_sorted), index.n, index.dim, _native.ptr(index.nodes),
                                                _native.ptr(verts), _native.ptr(w_perm), k1, R, S, _native.ptr(queue),
                                                _native.ptr(d2), st), "flooder_sweep_bvh_f64")
    if reduce_hook is not None:
        with _span(timer, "reduce"):
            reduce_hook(d2)
    out_face = torch.empty((S, faces.n_faces), dtype=torch.float64, device=dev)
    out_dist = torch.empty((S, R), dtype=torch.float64, device=dev) if want_dist else None
    with _span(timer, "face_max"):
        _native.check(lib.flooder_face_max_f64(_native.ptr(d2), S, R, _native.ptr(faces.ptr), _native.ptr(rows_perm),
                                               faces.n_faces, _native.ptr(out_face), _native.ptr(out_dist), st),
                      "flooder_face_max_f64")
    if out_dist is not None:
        out_dist = out_dist[:, inv]
    return out_face, out_dist


def bvh_sorts_samples(dim: int, S: int, R: int) -> bool:
    """Does the tree sweep of S simplices x R samples in ``dim`` dimensions run over spatially sorted samples?"""
    want = BVH_SORTED_SAMPLES if BVH_SORTED_SAMPLES is not None else dim > 3
    # (keys, sorted keys, order and the radix sort's scratch: ~20 B per sample on top of the (S, R) buffer - beyond
    # the workspace bound the per-simplex sweep, which needs none of it, takes over)
    return bool(want) and BVH_SORTED_MIN_SAMPLES <= S * R < (1 << 32) - 2 and S * R * 20 <= SORTED_WORKSPACE_BYTES


# A simplex-sharded run whose dimension pass goes through the sorted-sample sweep shards the TILES of the sorted order
# instead of the simplices (every rank sorts all samples; a rank's tiles are tiles of the unsharded sweep).  Every
# W-th simplex thins the samples W times and widens the tiles by W^(1/dim): an eighth of cfg 4's triangles took 14.0 ms
# of the 34.2 ms of all of them; an eighth of the tiles takes an eighth of the time.
SHARD_SORTED_TILES = True


def shards_sorted_tiles(dim: int, S: int, R: int, method: str) -> bool:
    return SHARD_SORTED_TILES and method == "bvh" and bvh_sorts_samples(dim, S, R)


def _sweep_dimension_bvh(index: PointIndex, verts: torch.Tensor, weights: torch.Tensor, faces: _FaceTable,
                         reduce_hook: Optional[Callable[[torch.Tensor], None]],
                         want_dist: bool = False, timer: Optional[_KernelTimer] = None,
                         stats: Optional[torch.Tensor] = None, plan: Optional["SamplePlan"] = None,
                         tile_shard: Optional[Tuple[int, int]] = None):
    """All simplices of one dimension against an indexed point set -> (S, F) face maxima.

    ``tile_shard=(rank, world)`` (sorted-sample sweep only, see ``shards_sorted_tiles``): this rank sweeps a contiguous
    world-th of the TILES of the sorted sample order; the (S, F) result holds the maxima over ITS samples (0 where it has
    none) and the ranks' results combine with MAX.

    sweep_bvh (plain stores into d2 bits, samples in the order given by sample_order) ->
    [reduce_hook: cross-shard MIN] -> face max (face rows remapped to the permuted sample order).
    No host synchronisation.
    """
    lib = _native.load()
    dev = index.pts.device
    st = _native.current_stream_ptr(dev)
    S, k1, _ = verts.shape
    R = weights.shape[0]
    verts = verts.to(torch.float32).contiguous()
    plan = plan if plan is not None else SamplePlan(weights, faces)
    w_perm, rows_perm = plan.w_perm, plan.rows_perm

    sorted_samples = bvh_sorts_samples(index.dim, S, R)
    # sharded work-queue heads + (sorted samples) the state of the radix sort of the sample keys, one zero fill for both:
    # flooder_index_sort_zeroed then makes no memset launch of its own (nine for four passes)
    n_sort_state = (int(lib.flooder_index_sort_state_words(S * R, 32)) if sorted_samples and SORT_STATE_BY_CALLER else 0)
    queue = torch.zeros(QUEUE_WORDS + n_sort_state, dtype=torch.int32, device=dev)
    sort_state = queue[QUEUE_WORDS:] if n_sort_state else None
    queue = queue[:QUEUE_WORDS]

    def sort_sample_keys(keys, n_s, bits, keys_sorted, order, tmp, tmp_bytes):
        if sort_state is not None:
            _native.check(lib.flooder_index_sort_zeroed(_native.ptr(keys), n_s, bits, _native.ptr(keys_sorted), _native.ptr(order),
                                                        _native.ptr(tmp), tmp_bytes, _native.ptr(sort_state), st),
                          "flooder_index_sort_zeroed (samples)")
        else:
            _native.check(lib.flooder_index_sort(_native.ptr(keys), n_s, bits, _native.ptr(keys_sorted), _native.ptr(order),
                                                 _native.ptr(tmp), tmp_bytes, st), "flooder_index_sort (samples)")
    if (sorted_samples and SORTED_FUSED_FACES and not want_dist and reduce_hook is None and plan.memb_all is not None
            and plan.late_rows is not None and tile_shard is None):
        # ---- only the face maxima are wanted: the sorted sweep delivers them itself (no (S, R) buffer, no face-max
        # pass) and drops every sample that cannot raise one - csrc/flood_sorted.hip, FUSED
        n_s = S * R
        F = faces.n_faces
        keys = torch.empty(n_s, dtype=torch.int32, device=dev)
        keys_sorted = torch.empty(n_s, dtype=torch.int32, device=dev)
        order = torch.empty(n_s, dtype=torch.int32, device=dev)
        tmp_bytes = int(lib.flooder_index_sort_bytes(n_s))
        tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
        face_bits = torch.zeros(S * F, dtype=torch.int32, device=dev)
        with _span(timer, "sweep"):
            _native.check(lib.flooder_sample_keys_late_f32(_native.ptr(verts), _native.ptr(w_perm), k1, R, S, index.dim,
                                                           _native.ptr(index.box), _native.ptr(plan.late_rows),
                                                           _native.ptr(keys), st), "flooder_sample_keys_late_f32")
            sort_sample_keys(keys, n_s, 32, keys_sorted, order, tmp, tmp_bytes)
            blk = _native.SortedSweep(pts_sorted=index.pts, n_pts=index.n, dim=index.dim, k1=k1, nodes=index.nodes, verts=verts,
                                      weights=w_perm, R=R, n_faces=F, n_simplices=S, sample_order=order, queue=queue,
                                      memb=plan.memb_all, face_bits=face_bits, stats=stats)
            _native.check(lib.flooder_sorted_faces(ctypes.byref(blk), st), "flooder_sorted_faces")
        del keys, keys_sorted, tmp
        out_face = torch.empty((S, F), dtype=torch.float32, device=dev)
        with _span(timer, "face_max"):
            _native.check(lib.flooder_face_values_f32(_native.ptr(face_bits), S * F, _native.ptr(out_face), st),
                          "flooder_face_values_f32")
        return out_face, None
    if tile_shard is not None and not sorted_samples:
        raise ValueError("tile_shard needs the sorted-sample sweep (shards_sorted_tiles)")
    # (a tile shard writes its own samples only: the others stay 0, the neutral element of the face maximum)
    d2 = (torch.zeros if tile_shard is not None else torch.empty)((S, R), dtype=torch.int32, device=dev)
    with _span(timer, "sweep"):
        if sorted_samples:
            # tiles of 64 spatially consecutive samples of ALL simplices (Z-order keys, one radix sort) instead of
            # the samples of one simplex each: see csrc/flood_sorted.hip
            n_s = S * R
            keys = torch.empty(n_s, dtype=torch.int32, device=dev)
            keys_sorted = torch.empty(n_s, dtype=torch.int32, device=dev)
            order = torch.empty(n_s, dtype=torch.int32, device=dev)
            tmp_bytes = int(lib.flooder_index_sort_bytes(n_s))
            tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
            _native.check(lib.flooder_sample_keys_f32(_native.ptr(verts), _native.ptr(w_perm), k1, R, S, index.dim,
                                                      _native.ptr(index.box), _native.ptr(keys), st),
                          "flooder_sample_keys_f32")
            sort_sample_keys(keys, n_s, int(lib.flooder_sample_key_bits(index.dim)), keys_sorted, order, tmp, tmp_bytes)
            # (a tile shard: every rank sorts ALL samples - the same order everywhere - and takes a contiguous world-th of
            # the tiles)
            blk = _native.SortedSweep(pts_sorted=index.pts, n_pts=index.n, dim=index.dim, k1=k1, nodes=index.nodes, verts=verts,
                                      weights=w_perm, R=R, n_simplices=S, sample_order=order, queue=queue, out_d2=d2,
                                      stats=stats, shard_rank=0 if tile_shard is None else int(tile_shard[0]),
                                      shard_world=0 if tile_shard is None else int(tile_shard[1]))
            _native.check(lib.flooder_sorted_minima(ctypes.byref(blk), st), "flooder_sorted_minima")
            del keys, keys_sorted, tmp
        else:
            _native.check(lib.flooder_sweep_bvh_f32(
                _native.ptr(index.pts), index.n, index.dim, _native.ptr(index.nodes), _native.ptr(verts),
                _native.ptr(w_perm), k1, R, S, _native.ptr(queue), _native.ptr(d2), _native.ptr(stats), st),
                "flooder_sweep_bvh_f32")
    if reduce_hook is not None:
        with _span(timer, "reduce"):
            reduce_hook(d2)
    out_face = torch.empty((S, faces.n_faces), dtype=torch.float32, device=dev)
    out_dist = torch.empty((S, R), dtype=torch.float32, device=dev) if want_dist else None
    with _span(timer, "face_max"):
        _native.check(lib.flooder_face_max_f32(_native.ptr(d2), S, R, _native.ptr(faces.ptr),
                                               _native.ptr(rows_perm), faces.n_faces,
                                               _native.ptr(out_face), _native.ptr(out_dist), st),
                      "flooder_face_max_f32")
    if out_dist is not None:
        out_dist = out_dist[:, plan.inv]  # back to the caller's sample order
    return out_face, out_dist


# tree sweep over spatially sorted samples (csrc/flood_sorted.hip): None = above 3 dimensions, True / False = always / never
BVH_SORTED_SAMPLES: Optional[bool] = None
SORTED_WORKSPACE_BYTES = 16 << 30   # scratch the sorted-sample sweep may take (288 GB of HBM per GPU)
SORTED_FUSED_FACES = False  # the sorted sweep delivers the face maxima itself and drops what cannot raise one (measured SLOWER at cfg 4: 116 vs 87 ms - in 6-D the distances of a triangle's samples concentrate, 83 % of the leaves are still evaluated; kept as an option)
BVH_SORTED_MIN_SAMPLES = 64 * 1024   # below this the sort costs more than it saves
EMPTY_CACHE_ABOVE_BYTES = 1 << 30   # flood_complex releases the allocator's cache when more than this is cached unused
CELL_ALPHA = 1.35   # cell size of the LDS grid in units of the local point spacing
# Per-face maxima folded into the cell sweep and its exact finish (only when neither the per-sample distances nor
# a cross-shard reduction of them is wanted): no (S, R) store, no face-max pass, and the finish skips every sample
# that cannot raise a face maximum.  False: sweep -> finish -> flooder_face_max_f32 over the full (S, R) buffer.
FUSED_FACES = True
CELL_PROBE = True    # the finish's probe (one greedy tree descent per flagged tile) runs inside the cell sweep
SHARED_FACE_SLOTS = True   # one running maximum per distinct face of the complex (top-dimensional grid sweeps)
FINISH_HARD_CAP = 32768  # entries per hard list of the finish (a tile that does not fit is finished by one wave)
CELL_DENSITY_GRID = True   # PointIndex carries a density grid; the cell sweep reads its first cell size from it
CELL_SUPER = True    # runs of four chunks share one gather / classification / stage (two launches: runs, deferred chunks)
WIT_MIN_SIMPLICES = 1536   # fewer simplices than this in a sweep: no witness sweep (1024 persistent workgroups, one simplex each)
# ... and none on a cloud with more than this many points per simplex: the witness sweep takes simplices with at most
# "wit_weight" (800) points in their bounding box - about six times the simplex's own share - and at 634 points per
# simplex (cfg 5) it finds 2000 candidates among 25 000, abandons every one after the gather (too dense for one stage)
# and costs 148 us + 43 us for its item list: 2.2 % of the step for nothing (cfg 2: 165 points per simplex, 4327 of
# 6052 handled; cfg 3: 181).  Results do not depend on it (witness on / off: bit-identical, tested).
WIT_MAX_POINTS_PER_SIMPLEX = 400
CELL_WITNESS = True  # sparse simplices go to the witness sweep first (whole simplex per wave, coarse samples + bounds)


def _sweep_dimension_cell(index: PointIndex, verts: torch.Tensor, weights: torch.Tensor, faces: _FaceTable,
                          reduce_hook: Optional[Callable[[torch.Tensor], None]],
                          want_dist: bool = False, timer: Optional[_KernelTimer] = None,
                          stats: Optional[torch.Tensor] = None, plan: Optional[SamplePlan] = None,
                          face_slots: Optional[Tuple[torch.Tensor, int]] = None):
    """Cell sweep (dim 2 / 3): wave-local LDS cell-grid sweep -> exact tree finish of the unverified tiles
    -> [reduce_hook] -> face maxima.  No host synchronisation.

    Default (only the face maxima are wanted): the fused path - ``flooder_sweep_cell_faces_f32``,
    ``flooder_finish_faces_f32``, ``flooder_face_values_f32``.  With ``want_dist`` or a ``reduce_hook`` (the
    per-sample minima are needed): ``flooder_sweep_cell_f32`` -> ``flooder_sweep_bvh_items_f32`` over the full
    (S, R) buffer -> ``flooder_face_max_f32``.

    ``stats`` (optional, 16 zeroed int64): [0:9] cell sweep {pairs, points staged, tiles flagged,
    re-staging rounds, 4 give-up reasons, exhaustive rounds}, [9:13] finish {leaves evaluated, leaves tested,
    nodes expanded, -}, fused path: [13] tiles the last finish pass dropped on arrival, [14] samples still live
    on arrival in that pass, [15] focus rounds.

    ``face_slots`` = (slot (S, F) int32, n_slots), fused path only: face f of simplex s accumulates into word
    ``slot[s, f]`` and the result is the (n_slots,) vector of per-slot values instead of the (S, F) matrix -
    one slot per DISTINCT face of the complex lets the simplices sharing a triangle / edge / vertex share its
    running maximum (``shared_face_slots``).
    """
    lib = _native.load()
    dev = index.pts.device
    st = _native.current_stream_ptr(dev)
    S, k1, _ = verts.shape
    R = weights.shape[0]
    verts = verts.to(torch.float32).contiguous()
    plan = plan if plan is not None else SamplePlan(weights, faces)
    w_perm, rows_perm = plan.w_perm, plan.rows_perm

    def sub(a, b):
        return None if stats is None else stats[a:b]

    if S == 0:
        return (torch.empty((0, faces.n_faces), dtype=torch.float32, device=dev),
                torch.empty((0, R), dtype=torch.float32, device=dev) if want_dist else None)
    planes = torch.empty(24 * S, dtype=torch.float32, device=dev)   # face planes per simplex (filled by the sweep's entry)
    if FUSED_FACES and not want_dist and reduce_hook is None and plan.memb_all is not None:
        # ---- only the per-face maxima are wanted: fused path.  The cell sweep delivers every settled sample to
        # face_bits (integer atomic max), the finish drops what cannot raise a face maximum, the (S, R) buffer is
        # scratch that only the flagged tiles touch.
        F = faces.n_faces
        tiles = (R + 63) // 64
        slot_t, n_slots = face_slots if face_slots is not None else (None, S * F)
        # [0] sweep queue, [1] flag count, [12] deferred chunks, [13] their queue, [14:17] light / heavy simplices,
        # lists on; [24:48] finish (queue heads, [27] top count, [29], [31] hard tiles of the top / rest pass);
        # [48:] histogram of the flagged tiles' bounds and the cursors of the finish's counting sort
        # (one zero fill for everything that starts at zero: top | ctl | face_bits)
        # (the sharded work-queue heads of the sweep's launches sit in front: QUEUE_WORDS each)
        # (a short queue - a rank's share of a multi-GPU run - leaves the witness sweep's workgroups one simplex each: the
        # launch then lasts as long as its longest item, 250 - 300 us, where the cell sweep balances chunk by chunk;
        # measured on an eighth of cfg 2: 0.65 ms per rank with it, 0.55 without)
        use_wit = (CELL_WITNESS and CELL_SUPER and CELL_PROBE and index.dim in (2, 3) and S >= WIT_MIN_SIMPLICES
                   and index.n <= WIT_MAX_POINTS_PER_SIMPLEX * S and plan.wit is not None)
        # (with the simplex weights wanted, the launch that computes them clears these words and writes the plane rows on
        # its way - flooder_simplex_prepare_f32 below -: one launch instead of three)
        n_zeroed = (1 if use_wit else 0) * QUEUE_WORDS + 6 * QUEUE_WORDS + 24 + 2 * S + 48 + 8192 + n_slots
        zeroed = (torch.empty if CELL_SUPER and SWEEP_PREPARE_FUSED else torch.zeros)(n_zeroed, dtype=torch.int32, device=dev)
        zeroed_all = zeroed
        if use_wit:
            qwit = zeroed[:QUEUE_WORDS]
            zeroed = zeroed[QUEUE_WORDS:]
        qbuf = zeroed[:3 * QUEUE_WORDS]
        fctl = zeroed[3 * QUEUE_WORDS:6 * QUEUE_WORDS + 24]   # finish: 24 control words, then its sharded queue heads
        zeroed = zeroed[6 * QUEUE_WORDS + 24:]
        top = zeroed[:2 * S].view(torch.int64)
        ctl = zeroed[2 * S:2 * S + 48 + 8192]
        face_bits = zeroed[2 * S + 48 + 8192:]
        hard = torch.empty(4 * FINISH_HARD_CAP, dtype=torch.int64, device=dev)
        top_list = torch.empty(S, dtype=torch.int32, device=dev)
        d2 = torch.empty((S, R), dtype=torch.int32, device=dev)
        flags = torch.empty((3, S * tiles), dtype=torch.int32, device=dev)  # flagged tiles, their bounds, ordered
        chunks = (R + 255) // 256
        # (chunk entries, then up to four tile entries per chunk: dense chunks hand their tiles to a third launch)
        defer_list = torch.empty(5 * S * chunks, dtype=torch.int32, device=dev) if CELL_SUPER else None
        defer_c = torch.empty(5 * S * chunks, dtype=torch.float32, device=dev) if CELL_SUPER else None
        wgt = None
        split = torch.empty((2, S), dtype=torch.int32, device=dev) if CELL_SUPER else None  # light / heavy simplices
        if CELL_SUPER:  # rough point count per simplex box: dense simplices skip the run-of-four launch
            wgt = torch.empty(S, dtype=torch.float32, device=dev)
            if SWEEP_PREPARE_FUSED:
                _native.check(lib.flooder_simplex_prepare_f32(_native.ptr(index.nodes), index.n, index.dim, _native.ptr(verts),
                                                              k1, S, _native.ptr(wgt), _native.ptr(planes),
                                                              _native.ptr(zeroed_all), n_zeroed, st),
                              "flooder_simplex_prepare_f32")
            else:
                _native.check(lib.flooder_simplex_weight_f32(_native.ptr(index.nodes), index.n, index.dim, _native.ptr(verts),
                                                             k1, S, _native.ptr(wgt), st), "flooder_simplex_weight_f32")
        # ONE parameter block for the three launches (include/flooder_hip.h: flooder_fused_sweep_t): they share the cloud,
        # the lattice, the result words and most of the scratch
        wst = stats[16:40] if use_wit and stats is not None and stats.numel() >= 40 else None
        blk = _native.FusedSweep(
            pts_sorted=index.pts, n_pts=index.n, dim=index.dim, k1=k1, nodes=index.nodes, density_grid=index.dens,
            cloud_box=index.box, verts=verts, weights=w_perm, R=R, n_faces=F, n_simplices=S, memb=plan.memb_all,
            alpha=float(CELL_ALPHA), face_bits=face_bits, face_slot=slot_t, d2_scratch=d2, flag_list=flags[0],
            flag_count=ctl[1:].data_ptr(), simplex_weight=wgt, plane_scratch=planes,
            cell_queue=qbuf.data_ptr(), defer_list=defer_list, defer_c=defer_c, cell_stats=sub(0, 9),
            finish_ctl=fctl.data_ptr(), hard_scratch=hard, hard_cap=FINISH_HARD_CAP, probed=1 if CELL_PROBE else 0,
            finish_stats=sub(9, 16))
        if CELL_PROBE:
            blk.flag_key, blk.flag_hist, blk.flag_sorted = flags[1].data_ptr(), ctl[48:].data_ptr(), flags[2].data_ptr()
            blk.top, blk.top_list, blk.top_count = top.data_ptr(), top_list.data_ptr(), fctl[3:].data_ptr()
        if CELL_SUPER:
            blk.defer_ctl, blk.light_list, blk.heavy_list = ctl[12:].data_ptr(), split[0].data_ptr(), split[1].data_ptr()
        if use_wit:   # (split[0]: the witness sweep's item list - scratch until the cell sweep's entry fills it)
            blk.n_coarse, blk.coarse_rows, blk.parents = plan.wit[2], plan.wit[0].data_ptr(), plan.wit[1].data_ptr()
            blk.wit_queue, blk.wit_item_list, blk.wit_stats = qwit.data_ptr(), split[0].data_ptr(), _native.ptr(wst) or None
        with _span(timer, "sweep"):
            try:
                if use_wit:
                    _native.check(lib.flooder_fused_witness(ctypes.byref(blk), st), "flooder_fused_witness")
                _native.check(lib.flooder_fused_cell(ctypes.byref(blk), st), "flooder_fused_cell")
            except Exception:
                lib.flooder_simplex_planes_forget()   # (the note flooder_simplex_prepare_f32 left for these two launches)
                raise
        with _span(timer, "fallback"):
            _native.check(lib.flooder_fused_finish(ctypes.byref(blk), st), "flooder_fused_finish")
        if stats is not None:  # (diagnostic runs only: a host synchronisation)
            LAST_STATS.deferred_chunks = int(ctl[12].item())
            LAST_STATS.light_heavy = (int(ctl[14].item()), int(ctl[15].item()))
            LAST_STATS.dense_tiles = int(ctl[18].item())
            c_h = fctl[:24].tolist()
            LAST_STATS.hard_entries = (int(c_h[5]), int(c_h[7]))
        out_face = torch.empty(n_slots if face_slots is not None else (S, F), dtype=torch.float32, device=dev)
        with _span(timer, "face_max"):
            _native.check(lib.flooder_face_values_f32(_native.ptr(face_bits), n_slots, _native.ptr(out_face), st),
                          "flooder_face_values_f32")
        return out_face, None

    ctl = torch.zeros(8 + 4 * QUEUE_WORDS, dtype=torch.int32, device=dev)  # flag counter, list length | queue heads
    qbuf = ctl[8:8 + 3 * QUEUE_WORDS]
    q_items = ctl[8 + 3 * QUEUE_WORDS:]
    d2 = torch.empty((S, R), dtype=torch.int32, device=dev)
    flags = torch.empty(S * ((R + 63) // 64), dtype=torch.int32, device=dev)
    with _span(timer, "sweep"):
        _native.check(lib.flooder_sweep_cell_f32(
            _native.ptr(index.pts), index.n, index.dim, _native.ptr(index.nodes), _native.ptr(verts),
            _native.ptr(w_perm), k1, R, S, float(CELL_ALPHA), qbuf.data_ptr(), _native.ptr(d2), _native.ptr(flags),
            ctl[1:].data_ptr(), _native.ptr(planes), _native.ptr(index.dens), _native.ptr(index.box),
            _native.ptr(sub(0, 9)), st), "flooder_sweep_cell_f32")
    with _span(timer, "fallback"):
        _native.check(lib.flooder_sweep_bvh_items_f32(
            _native.ptr(index.pts), index.n, index.dim, _native.ptr(index.nodes), _native.ptr(verts),
            _native.ptr(w_perm), k1, R, S, _native.ptr(flags), ctl[1:].data_ptr(), q_items.data_ptr(),
            _native.ptr(d2), 0, None, None, _native.ptr(sub(9, 13)), st), "flooder_sweep_bvh_items_f32")

    if reduce_hook is not None:
        with _span(timer, "reduce"):
            reduce_hook(d2)
    out_face = torch.empty((S, faces.n_faces), dtype=torch.float32, device=dev)
    out_dist = torch.empty((S, R), dtype=torch.float32, device=dev) if want_dist else None
    with _span(timer, "face_max"):
        _native.check(lib.flooder_face_max_f32(_native.ptr(d2), S, R, _native.ptr(faces.ptr),
                                               _native.ptr(rows_perm), faces.n_faces,
                                               _native.ptr(out_face), _native.ptr(out_dist), st),
                      "flooder_face_max_f32")
    if out_dist is not None:
        out_dist = out_dist[:, plan.inv]
    return out_face, out_dist


def _face_max_cpu(dist: torch.Tensor, faces: _FaceTable) -> torch.Tensor:
    ptr = faces.ptr.cpu().numpy()
    rows = faces.rows.cpu().numpy()
    d = dist.numpy() if isinstance(dist, torch.Tensor) else dist
    out = np.empty((d.shape[0], faces.n_faces), dtype=d.dtype)
    for f in range(faces.n_faces):
        out[:, f] = d[:, rows[ptr[f]:ptr[f + 1]]].max(axis=1)
    return torch.as_tensor(out)


def flood_complex(
    points: torch.Tensor,
    landmarks: Union[int, torch.Tensor],
    max_dimension: Union[None, int] = None,
    points_per_edge: Union[None, int] = 30,
    num_rand: int = None,
    batch_size: Union[None, int] = 64,
    use_triton: Optional[bool] = None,
    return_simplex_tree: bool = False,
    fps_h: Union[None, int] = None,
    start_idx: Union[int, None] = 0,
    *,
    reduce_hook: Optional[Callable[[torch.Tensor], None]] = None,
    sort_axis: Optional[int] = None,
    method: Optional[str] = None,
    simplex_shard: Optional[Tuple[int, int]] = None,
    face_reduce_hook: Optional[Callable[[torch.Tensor], None]] = None,
    index: Optional["PointIndex"] = None,
    shard_blocks: bool = False,
    landmarks_in_cloud: Optional[bool] = None,
):
    """Flood complex of ``points`` over the Delaunay triangulation of ``landmarks``.

    Drop-in for the reference's ``flood_complex`` (``flooder/core.py:32-288``): same positional and
    keyword arguments, returns ``{simplex tuple: filtration}`` or a simplex tree
    (``gudhi.SimplexTree`` when gudhi is installed, else ``flooder_amd.SimplexTree``).

    Differences that do not change results: ``use_triton`` selects nothing - ROCm tensors always run
    the HIP kernels (``use_triton=True`` still raises ``ImportError`` if they are unavailable, as the
    reference does when Triton is); ``batch_size`` is accepted, the HBM workspace is bounded by
    grouping simplices against ``CAND_WORKSPACE_BYTES`` instead.  ``reduce_hook`` (keyword-only
    extension) is called with the (S, R) tensor of minimum squared-distance bit patterns of every
    dimension pass before the per-face maxima are taken - int32 words of float32 values for float32 inputs,
    int64 words of float64 values for float64 inputs (integer order == numeric order either way; CPU tensors:
    the float distance matrix); ``flooder_amd.distributed`` uses it for the cross-GPU ``all_reduce(MIN)``.  ``sort_axis`` (keyword-only) fixes the coordinate axis used for the
    cloud sort and the simplex order instead of deriving it from ``points`` (ranks holding different
    shards must agree on the simplex order of the reduced buffer).  ``method`` (keyword-only):
    ``"cell"`` (default in 2D/3D: per-simplex cell grid in LDS, exact tree finish) and ``"bvh"`` (default
    otherwise: box-tree culling) evaluate the exact nearest neighbour; ``"ball"`` runs the reference's
    formulation literally (bounding-ball candidates, exhaustive sweep).  All give the same values
    whenever the landmarks are points of the cloud (the reference's precondition for its own GPU path,
    SURVEY.md section 8 a-2); for other landmarks ``"cell"``/``"bvh"`` return the exact value of the
    reference's CPU path.  ``index`` (keyword-only): a ``PointIndex`` built from these very ``points`` (ROCm
    tensors, methods ``"cell"``/``"bvh"``) - the curve-sorted copy and box tree are reused instead of rebuilt
    (callers that sweep one cloud several times, and every rank of a multi-GPU run; the caller vouches that
    ``points`` has not changed since: the shape is checked, and an in-place write that moved torch's version counter is
    refused).  The handle comes from ``generate_landmarks(points, n, return_index=True)``; with integer ``landmarks``
    the function builds one index for the selection and the sweep itself.  No index is remembered between calls
    unless ``core.INDEX_CACHE`` (FLOODER_INDEX_CACHE=1) is switched on - and then by tensor identity and version
    counter only, which does NOT see writes through ``.data``, raw-pointer kernels or DLPack / cupy aliases: with the
    cache on, such a write before the next call sweeps a stale copy of the cloud.  ``simplex_shard=(rank, world)`` sweeps this rank's
    share of the simplices only (the other rows of the (S, F) values are +inf until ``face_reduce_hook`` combines
    them; on the default cell-sweep path the ranks share ONE word per distinct face of the complex as a single GPU
    does, and what ``face_reduce_hook`` receives is that (n_slots,) vector, +inf for the faces none of this rank's
    simplices has): every ``world``-th simplex of the queue, or - ``shard_blocks=True``, float32 ROCm tensors, methods
    ``"cell"``/``"bvh"``, landmarks that are POINTS OF THE CLOUD (the caller vouches; true for every
    ``generate_landmarks`` result) - a contiguous block of the queue, swept against an index of the sub-cloud inside
    the block's bounding balls only (``block_subcloud``): the index build shrinks with the share, too
    (``landmarks_in_cloud=True``: the caller vouches that the landmarks are rows of ``points`` and the O(n log n)
    membership check with its host synchronisation is skipped; ``None``: checked; ``False``: refused).  A dimension
    pass that runs the sorted-sample sweep (above 3D; ``shards_sorted_tiles``) shards the TILES of the sorted sample
    order instead: this rank's (S, F) values are the maxima over ITS samples (0 where it has none), and
    ``face_reduce_hook`` - an in-place elementwise MIN over the ranks, whatever it is given - receives the NEGATED
    matrix, so that its MIN is the MAX of the values.
    """
    if use_triton is None:
        use_triton = HAS_HIP_KERNELS
    if use_triton and not _has_hip_kernels():
        raise ImportError(
            "use_triton=True requested, but the HIP kernels are not available in this environment "
            "(build them with `python -m flooder_amd.build`)."
        )
    if points.dim() != 2 or points.shape[0] == 0:
        raise RuntimeError(f"points must be a non-empty (N, d) tensor, got shape {tuple(points.shape)}")
    method = SWEEP_METHOD if method is None else method
    if method == "auto":
        # (the cell sweep addresses the cloud with 32-bit byte offsets: 16 B rows, below 2^28 points)
        method = "cell" if points.shape[1] in (2, 3) and points.shape[0] < (1 << 28) - 64 else "bvh"
    if method not in ("cell", "bvh", "ball"):
        raise ValueError(f"method must be 'cell', 'bvh' or 'ball', got {method!r}")
    if method == "cell" and points.shape[1] not in (2, 3):
        raise ValueError("method 'cell' supports ambient dimension 2 and 3 only")
    if max_dimension is None:
        max_dimension = points.shape[1]
    shared_index = None  # one curve-sorted copy of the cloud serves the landmark selection and the sweep
    if index is not None:
        if not points.is_cuda or method == "ball":
            raise ValueError("index= applies to ROCm tensors with method 'cell' or 'bvh'")
        if not isinstance(index, PointIndex) or (index.n, index.dim) != tuple(points.shape) or index.pts.device != points.device:
            raise ValueError("index= is not a PointIndex of these points (shape or device differ)")
        src = getattr(index, "source", None)
        ver = _tensor_version(points)
        if src is not None and src[0] == points.data_ptr() and src[1] is not None and ver is not None and src[1] != ver:
            raise ValueError("index= was built from an earlier state of `points` (the tensor has been modified in "
                             "place since): rebuild the PointIndex")
        shared_index = index
    if isinstance(landmarks, Integral):
        if (shared_index is None and points.is_cuda and method != "ball" and points.shape[1] <= FPS_BUCKET_MAX_DIM and points.dtype in SUPPORTED_DTYPES
                and _has_hip_kernels() and points.shape[0] >= FPS_BUCKET_MIN_POINTS and landmarks > 64):
            shared_index = _recall_index(points)
            if shared_index is None:
                shared_index = PointIndex(points.to(torch.float32))
                _remember_index(points, shared_index)
        landmarks = generate_landmarks(points, min(landmarks, points.shape[0]), fps_h, start_idx=start_idx,
                                       index=shared_index)
    if landmarks.device != points.device:
        raise RuntimeError(f"landmarks.device ({landmarks.device}) != points.device ({points.device})")
    if landmarks.dtype != points.dtype:
        raise RuntimeError(f"landmarks.dtype ({landmarks.dtype}) != points.dtype ({points.dtype})")
    device = points.device
    dtype = points.dtype
    if dtype not in SUPPORTED_DTYPES:
        raise TypeError(f"dtype ({dtype}) not supported")
    if dtype is torch.float64:
        # (the reference warns about float64 as well, core.py:118-123: its kernels then run in double, as here)
        warnings.warn("float64 inputs: the device sweep runs its float64 kernel (tree sweep in double, several "
                      "times slower than the float32 path); pass float32 tensors for speed", RuntimeWarning,
                      stacklevel=2)
    if device.type not in ("cuda", "cpu"):
        raise RuntimeError("Device not supported.")
    dim = points.shape[1]
    on_gpu = device.type == "cuda"
    if on_gpu:
        if not _has_hip_kernels():
            _native.load()  # raises ImportError with the build hint: no silent fallback
        if dim > 8:
            raise RuntimeError("flooder_amd: ambient dimension > 8 is not supported by the HIP kernels")
        torch.cuda.set_device(device)
    else:
        from scipy.spatial import KDTree

        kdtree = KDTree(np.asarray(points))

    index = None
    use_f64 = on_gpu and dtype is torch.float64 and method != "ball"
    pts64_sorted = None
    blocks = bool(shard_blocks) and simplex_shard is not None and on_gpu and method != "ball" and not use_f64
    block_box = None
    if blocks:
        pts32 = points.to(torch.float32).contiguous()
        block_box = shared_index.box if shared_index is not None else cloud_box(pts32)
        # (the membership test hashes and sorts the whole cloud and synchronises with the host: a caller whose
        # landmarks come from generate_landmarks vouches with landmarks_in_cloud=True and skips it)
        if landmarks_in_cloud is False or (landmarks_in_cloud is None
                                           and not _rows_are_subset(landmarks.to(torch.float32), pts32)):
            raise ValueError("shard_blocks / mode='blocks' needs landmarks that are rows of `points` (every "
                             "generate_landmarks result is): a block's sub-cloud holds only the rows inside its "
                             "simplices' bounding balls, which bound the nearest neighbours of the samples only then "
                             "- use mode='simplices'")
    elif on_gpu and method != "ball":
        # the curve sort + box tree run on the GPU while the host triangulates the landmarks
        pts32 = points.to(torch.float32)
        if shared_index is None:
            shared_index = _recall_index(points)     # (the index generate_landmarks built from this very tensor)
        if shared_index is not None:
            index = shared_index
        else:
            index = PointIndex(pts32)
            _remember_index(points, index)
        if use_f64:  # the float64 rows in the order of the (float32) index
            lib_ = _native.load()
            pts64_sorted = torch.empty((index.pts.shape[0], index.dp), dtype=torch.float64, device=device)
            _native.check(lib_.flooder_gather_rows_f64(_native.ptr(points.contiguous()), index.n, dim, dim,
                                                       _native.ptr(index.order32), _native.ptr(pts64_sorted),
                                                       index.pts.shape[0], _native.current_stream_ptr(device)),
                          "flooder_gather_rows_f64")
    stree, simplices = _build_complex(landmarks, max_dimension)
    LAST_STATS.reset()
    LAST_STATS.n_points = points.shape[0]

    # widest axis of the cloud (core.py:140-144): sort key of the reference, work order of the simplices here
    if sort_axis is not None:
        axis = int(sort_axis)
    elif on_gpu:
        box = (block_box if block_box is not None else (index.box if index is not None else cloud_box(points))).cpu()
        axis = int(torch.argmax(box[8:8 + dim] - box[:dim]).item())
    else:
        axis = int(torch.argmax(points.max(dim=0).values - points.min(dim=0).values).item())
    if on_gpu:
        pts32 = points.to(torch.float32)
        dp = _native.load().flooder_padded_dim(dim)
        lm32 = landmarks.to(torch.float32)
        if method == "ball":  # the reference's formulation: cloud sorted along the widest axis (core.py:140-144)
            pts_pad = _pad_rows(pts32[torch.argsort(pts32[:, axis])], dp)
            search = pts_pad[:, axis].contiguous()
    lm_np = landmarks.detach().to(torch.float64 if use_f64 else (torch.float32 if on_gpu else dtype)).cpu().numpy()

    results: List[tuple] = []  # (simplices (n,k), values (n,)) or indexed cell-face assignments, in update order
    peak_ws = 0
    for d in range(max_dimension + 1):
        if num_rand is None and d < max_dimension:
            continue
        num_simplices = simplices[d].shape[0]
        if num_simplices == 0:
            continue
        plan = None
        centers = radii = None
        if on_gpu and method != "ball":
            # the culled sweeps need no bounding balls; the simplices are queued along the widest axis (any order
            # gives the same values), prepared on the host so that nothing here waits for the device
            v_np = lm_np[simplices[d]]
            # (by the vertex sum along that axis alone: the mean over all axes first was 0.4 ms of a 6.8 ms call at cfg 2)
            order_np = np.argsort(v_np[:, :, axis].sum(axis=1), kind="stable")
            simp_h = simplices[d][order_np]
            simplex_vertices = torch.as_tensor(np.ascontiguousarray(v_np[order_np]), device=device)
        else:
            d_simplices = torch.as_tensor(simplices[d], device=device)
            lm = lm32 if on_gpu else landmarks
            simplex_vertices = lm[d_simplices]
            centers, radii = _ball_prep(simplex_vertices, d)
            splx_idx = torch.argsort(centers[:, axis])
            simplex_vertices = simplex_vertices[splx_idx]
            centers = centers[splx_idx]
            radii = radii[splx_idx]
            order_np = splx_idx.cpu().numpy()
            simp_h = simplices[d][order_np]

        w_dtype = torch.float64 if use_f64 else (torch.float32 if on_gpu else dtype)
        if num_rand is None:
            weights, vertex_idxs, face_idxs, faces, plan, v_idx_np = _grid_tables(
                points_per_edge, max_dimension, device, w_dtype)
            if use_f64:
                plan = None
        else:
            weights = generate_uniform_weights(num_rand, d, device, w_dtype)
            vertex_idxs = face_idxs = v_idx_np = None
            faces = _FaceTable(None, weights.shape[0], device)
        LAST_STATS.top_simplices = num_simplices
        LAST_STATS.samples_per_simplex = weights.shape[0]
        # (rough device workspace of this pass: 4 B per sample, 24 with the sorted-sample sweep's keys and orders)
        peak_ws = max(peak_ws, num_simplices * weights.shape[0] *
                      (24 if (method == "bvh" and bvh_sorts_samples(dim, num_simplices, weights.shape[0])) else 8))

        tile_shard = None
        if simplex_shard is not None:
            sh_rank, sh_world = simplex_shard
            if blocks:   # a contiguous block of the queue (the simplices are ordered along the widest axis)
                mine = torch.arange(num_simplices * sh_rank // sh_world, num_simplices * (sh_rank + 1) // sh_world,
                                    device=device)
            elif (on_gpu and not use_f64 and reduce_hook is None
                  and shards_sorted_tiles(dim, num_simplices, weights.shape[0], method)):
                mine = None          # all simplices, a contiguous run of the TILES of the sorted sample order
                tile_shard = (sh_rank, sh_world)
            else:
                mine = torch.as_tensor(simplex_share(simplex_vertices, sh_rank, sh_world), device=device)
        else:
            mine = None
        sv = simplex_vertices if mine is None else simplex_vertices[mine]
        if blocks and sv.shape[0] > 0:
            index = PointIndex(block_subcloud(pts32, sv, d, box=block_box))
        # one running maximum per DISTINCT face of the complex (the simplices that share a triangle / an edge / a vertex
        # share its word): also on a shard of the simplices - a rank then holds the (n_slots,) vector, +inf for the
        # faces none of its simplices has, and the ranks are combined with MIN as before (round 5: a rank's finish
        # used to search every shared face once per simplex)
        slots = None
        if (on_gpu and method == "cell" and not use_f64 and SHARED_FACE_SLOTS and FUSED_FACES and num_rand is None
                and reduce_hook is None and faces.n_faces <= 32):
            slots = shared_face_slots(stree, d, order_np if mine is None else order_np[mine.cpu().numpy()], v_idx_np,
                                      device)
        if on_gpu and sv.shape[0] == 0 and (blocks or slots is not None):   # (more ranks than simplices)
            face_dev = (torch.full((slots[1],), float("inf"), dtype=torch.float32, device=device) if slots is not None
                        else torch.empty((0, faces.n_faces), dtype=torch.float32, device=device))
        elif on_gpu:
            if method == "ball":
                face_dev, _ = _sweep_dimension_hip(pts_pad, search, axis, dim, sv,
                                                   centers if mine is None else centers[mine],
                                                   radii if mine is None else radii[mine], weights, faces,
                                                   reduce_hook)
            elif use_f64:
                face_dev, _ = _sweep_dimension_f64(index, pts64_sorted, sv, weights, faces, reduce_hook)
            elif method == "cell":
                face_dev, _ = _sweep_dimension_cell(index, sv, weights, faces, reduce_hook, plan=plan,
                                                    face_slots=None if slots is None else slots[:2])
            else:
                face_dev, _ = _sweep_dimension_bvh(index, sv, weights, faces, reduce_hook, plan=plan, tile_shard=tile_shard)
        else:
            samples = weights.unsqueeze(0) @ sv
            # (the reference queries with scipy's default of ONE worker, core.py:198; the distances do not depend on
            # the number of workers)
            dist, _ = kdtree.query(np.asarray(samples), workers=CPU_WORKERS)
            dist = torch.as_tensor(dist)
            if reduce_hook is not None:
                reduce_hook(dist)
            face_dev = _face_max_cpu(dist, faces)
        if (simplex_shard is not None and face_reduce_hook is not None and simplex_shard[1] > 1 and CHECK_RANK_CONSISTENCY
                and getattr(face_reduce_hook, "checks_ranks", False)):   # (hooks of distributed.min_reduce_hook: real collectives)
            # the collective's buffer must have the same shape on every rank: the complexes agree (same Delaunay code
            # path, same landmarks) or we stop here instead of hanging in a mismatched all-reduce
            _assert_ranks_agree(face_reduce_hook, device, num_simplices, faces.n_faces,
                                -1 if slots is None else int(slots[1]))
        if mine is not None and slots is not None:
            # every distinct face is on some rank, and whoever has it holds its exact value; the faces none of this
            # rank's simplices has are +inf until the hook's MIN brings them in
            face_dev = torch.maximum(face_dev, shard_slot_fill(slots[0], slots[1]))
            if face_reduce_hook is not None:
                face_reduce_hook(face_dev)
        elif mine is not None:
            full = torch.full((num_simplices, faces.n_faces), float("inf"), dtype=face_dev.dtype, device=device)
            full[mine] = face_dev
            if face_reduce_hook is not None:
                face_reduce_hook(full)
            face_dev = full
        elif tile_shard is not None and face_reduce_hook is not None:
            # every rank holds the maxima over its own samples: MAX over the ranks, through the hook's MIN on the
            # negated matrix (the values are non-negative floats: exact either way)
            neg = -face_dev
            face_reduce_hook(neg)
            face_dev = -neg
        face_vals = face_dev.cpu().numpy().astype(np.float64)

        if on_gpu and method == "cell" and slots is not None:
            # one value per distinct face: straight into the tables
            for v_idx, off in zip(v_idx_np, slots[2]):
                k = v_idx.shape[1]
                n_rows = stree.simplices_of_dimension(k - 1).shape[0]
                results.append(("table", k - 1, face_vals[off:off + n_rows]))
        elif num_rand is None:
            # faces of the swept simplices.  When those are the top cells of the complex, the rows of the face tables
            # are known from the enumeration of the faces (no search); else located by key.
            top_cells = isinstance(stree, SimplexTree) and stree._cells is not None and stree._cells.shape[1] == d + 1
            col = 0
            for v_idx in v_idx_np:
                nf, k = v_idx.shape
                vals_k = face_vals[:, col:col + nf]
                col += nf
                if top_cells and stree.cell_face_index(k - 1) is not None:
                    combos = list(itertools.combinations(range(d + 1), k))
                    results.append(("cells", k - 1, order_np, [combos.index(tuple(int(x) for x in row)) for row in v_idx],
                                    vals_k))
                elif isinstance(stree, SimplexTree) and k == d + 1 and nf == 1:
                    # the swept simplices themselves: simp_h is the dimension-d table in queue order
                    results.append(("rows", d, order_np, vals_k.reshape(-1)))
                elif isinstance(stree, SimplexTree) and k == 1 and stree._cells is not None:
                    # vertices of a complex built from cells: row = vertex id (later rows win, as a dict update does)
                    results.append(("rows", 0, simp_h[:, v_idx].reshape(-1), vals_k.reshape(-1)))
                else:
                    results.append((simp_h[:, v_idx].reshape(-1, k), vals_k.reshape(-1)))
        else:
            results.append((simp_h, face_vals[:, 0]))

    # hand-off (core.py:278-288)
    if isinstance(stree, SimplexTree):
        for item in results:
            if isinstance(item[0], str) and item[0] == "table":
                stree._vals[item[1]][:] = item[2]
                stree._persistence = None
            elif isinstance(item[0], str) and item[0] == "rows":
                stree.filtrations_of_dimension(item[1])[item[2]] = item[3]
                stree._persistence = None
            elif isinstance(item[0], str):
                stree.assign_cell_faces(item[1], item[2], item[3], item[4])
            else:
                stree.assign_filtration_bulk(item[0], item[1])
    else:  # a gudhi tree (core.py:278-280 of the reference)
        for simp, vals in results:
            for s, v in zip(simp.tolist(), vals.tolist()):
                stree.assign_filtration(s, v)
    stree.make_filtration_non_decreasing()
    # (asking the allocator costs 0.2 ms of a 10 ms call: only after a call that can have left that much behind)
    if (on_gpu and peak_ws > EMPTY_CACHE_ABOVE_BYTES // 4
            and torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device) > EMPTY_CACHE_ABOVE_BYTES):
        # (core.py:281 of the reference empties the allocator's cache after every call - it has just dropped several GB
        # of masks and index pairs; here a call leaves tens of MB behind and the release costs a millisecond of an 11 ms
        # call, so it is done only when there is something worth releasing)
        torch.cuda.empty_cache()
    if return_simplex_tree:
        return stree
    if isinstance(stree, SimplexTree):
        return stree.to_dict()
    return dict((tuple(simplex), filtr) for (simplex, filtr) in stree.get_simplices())
