"""Tensor-level wrappers over the C ABI: take torch tensors (device memory + stream plumbing only),
pass raw pointers to libgaot3d_hip.so.  Every function requires CUDA(HIP) tensors and raises otherwise."""
from __future__ import annotations

import ctypes as C
import functools
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import GaotError, MlpGradT, MlpT, check

Tensor = torch.Tensor


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> C.c_void_p:
    """the current stream's hipStream_t (the raw getter is ~20x cheaper than building a torch.cuda.Stream per launch)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req(t: Tensor, dtype, name: str) -> Tensor:
    if not t.is_cuda:
        raise GaotError(f"{name}: expected a tensor on the GPU (the HIP path has no CPU fallback), got {t.device}")
    if t.dtype != dtype:
        raise GaotError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# Optional per-call device timing (bench.py): HIP events recorded on the stream the kernels are launched on.
TIMING = {"enabled": False, "events": {}}


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMING["enabled"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if TIMING["enabled"]:
            self.e1.record(torch.cuda.current_stream())
            TIMING["events"].setdefault(self.name, []).append((self.e0, self.e1))
        return False


def timing_reset(enabled: bool):
    TIMING["enabled"] = enabled
    TIMING["events"] = {}


def timing_summary():
    """-> {name: (calls, total_ms)} ; call after torch.cuda.synchronize()"""
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in TIMING["events"].items()}


# ------------------------------------------------------------------------------------------------
# CSR
# ------------------------------------------------------------------------------------------------
@dataclass
class SortedEdges:
    """Edge list sorted (stably) by one endpoint.  ``key`` = the sort endpoint, ``other`` = the other."""
    rowptr: Tensor   # int32 [num_rows+1]
    perm: Tensor     # int32 [E]
    key: Tensor      # int32 [E]
    other: Tensor    # int32 [E]
    num_rows: int

    @property
    def num_edges(self) -> int:
        return int(self.key.shape[0])


def csr_build(edge_index: Tensor, sort_row: int, num_rows: int) -> SortedEdges:
    lib = _lib.load()
    if edge_index.dtype not in (torch.int32, torch.int64):
        raise GaotError(f"edge_index must be int32 or int64, got {edge_index.dtype}")
    ei = _req(edge_index, edge_index.dtype, "edge_index")
    if ei.dim() != 2 or ei.shape[0] != 2:
        raise GaotError(f"edge_index must be [2,E], got {tuple(ei.shape)}")
    e = int(ei.shape[1])
    dev = ei.device
    rowptr = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    key = torch.empty(e, dtype=torch.int32, device=dev)
    other = torch.empty(e, dtype=torch.int32, device=dev)
    nb = lib.gaot_csr_workspace_bytes(e, num_rows)
    ws = _ws(nb, dev)
    check(lib.gaot_csr_build(_ptr(ei), int(ei.dtype == torch.int64), e, sort_row, num_rows, _ptr(rowptr), _ptr(perm),
                             _ptr(key), _ptr(other), _ptr(ws), ws.numel(), _stream()), "gaot_csr_build")
    return SortedEdges(rowptr, perm, key, other, num_rows)


@dataclass
class BipartiteGraph:
    """Both orderings of one reference ``edge_index`` (row 0 = source, row 1 = query)."""
    by_dst: SortedEdges   # key = query,  other = source
    by_src: SortedEdges   # key = source, other = query
    num_src: int
    num_dst: int


def build_graph(edge_index: Tensor, num_src: int, num_dst: int) -> BipartiteGraph:
    return BipartiteGraph(csr_build(edge_index, 1, num_dst), csr_build(edge_index, 0, num_src), num_src, num_dst)


# ------------------------------------------------------------------------------------------------
# GNO integral transform
# ------------------------------------------------------------------------------------------------
def _mlp_struct(weights: Sequence[Tensor], biases: Sequence[Tensor]) -> Tuple[MlpT, List[Tensor]]:
    n = len(weights)
    if n < 2 or n > _lib.MAX_MLP_LAYERS:
        raise GaotError(f"kernel MLP must have 2..{_lib.MAX_MLP_LAYERS} linear layers, got {n}")
    keep = []
    m = MlpT()
    m.n_hidden = n - 1
    m.hidden = int(weights[0].shape[0])
    m.channels = int(weights[-1].shape[0])
    for l, (w, b) in enumerate(zip(weights, biases)):
        if w.dim() == 3:  # Conv1d(k=1) storage of mlp_type='channel'
            w = w[:, :, 0]
        w = _req(w, torch.float32, f"mlp.weight[{l}]")
        b = _req(b, torch.float32, f"mlp.bias[{l}]")
        exp_in = 6 if l == 0 else m.hidden
        exp_out = m.channels if l == n - 1 else m.hidden
        if tuple(w.shape) != (exp_out, exp_in):
            raise GaotError(f"kernel MLP layer {l}: weight shape {tuple(w.shape)} unsupported by the fused HIP path "
                            f"(needs [{exp_out},{exp_in}]: coord_dim 3, equal hidden widths)")
        keep += [w, b]
        m.weight[l] = w.data_ptr()
        m.bias[l] = b.data_ptr()
    return m, keep


def gno_forward(weights, biases, y_pos: Tensor, x_pos: Tensor, f_y: Tensor, g: BipartiteGraph,
                precision: Optional[int] = None) -> Tensor:
    lib = _lib.load()
    prec = _PRECISION["mode"] if precision is None else precision
    m, keep = _mlp_struct(weights, biases)
    y_pos = _req(y_pos, torch.float32, "y_pos")
    x_pos = _req(x_pos, torch.float32, "x_pos")
    f_y = _req(f_y, torch.float32, "f_y")
    if f_y.shape[1] != m.channels:
        raise GaotError(f"f_y has {f_y.shape[1]} channels, kernel MLP outputs {m.channels}")
    q = g.num_dst
    e = g.by_dst.num_edges
    out = torch.empty(q, m.channels, dtype=torch.float32, device=x_pos.device)
    ws = _ws(lib.gaot_gno_fwd_workspace_bytes(e, m.channels), x_pos.device)
    with _timed(f"gno_fwd_nh{m.n_hidden}"):
        check(lib.gaot_gno_fwd(C.byref(m), _ptr(y_pos), _ptr(x_pos), _ptr(f_y), _ptr(g.by_dst.other), _ptr(g.by_dst.key),
                               _ptr(g.by_dst.rowptr), e, q, _ptr(out), prec, _ptr(ws), ws.numel(), _stream()), "gaot_gno_fwd")
    return out


def gno_backward(weights, biases, y_pos: Tensor, x_pos: Tensor, f_y: Tensor, grad_out: Tensor, g: BipartiteGraph,
                 precision: Optional[int] = None):
    """-> (grad_f_y, [grad_w...], [grad_b...])"""
    lib = _lib.load()
    prec = _PRECISION["mode"] if precision is None else precision
    m, keep = _mlp_struct(weights, biases)
    y_pos = _req(y_pos, torch.float32, "y_pos")
    x_pos = _req(x_pos, torch.float32, "x_pos")
    f_y = _req(f_y, torch.float32, "f_y")
    grad_out = _req(grad_out, torch.float32, "grad_out")
    e = g.by_src.num_edges
    dev = x_pos.device
    grad_f = torch.empty(g.num_src, m.channels, dtype=torch.float32, device=dev)
    gw = [torch.empty(tuple(w.shape), dtype=torch.float32, device=dev) for w in weights]
    gb = [torch.empty(tuple(b.shape), dtype=torch.float32, device=dev) for b in biases]
    gs = MlpGradT()
    for l in range(len(weights)):
        gs.weight[l] = gw[l].data_ptr()
        gs.bias[l] = gb[l].data_ptr()
    ws = _ws(lib.gaot_gno_bwd_workspace_bytes(C.byref(m), e, g.num_dst), dev)
    with _timed(f"gno_bwd_nh{m.n_hidden}"):
        check(lib.gaot_gno_bwd(C.byref(m), _ptr(y_pos), _ptr(x_pos), _ptr(f_y), _ptr(grad_out), _ptr(g.by_dst.rowptr),
                               _ptr(g.by_src.key), _ptr(g.by_src.other), _ptr(g.by_src.rowptr), e, g.num_src, g.num_dst,
                               _ptr(grad_f), C.byref(gs), prec, _ptr(ws), ws.numel(), _stream()), "gaot_gno_bwd")
    return grad_f, gw, gb


# ------------------------------------------------------------------------------------------------
# precision switch: "fp32" = exact-fp32 MFMA everywhere (parity mode); "bf16" = bf16 operands with
# fp32 accumulation for the dense GEMMs / attention (BASELINE config 1)
# ------------------------------------------------------------------------------------------------
_PRECISION = {"mode": 0}
_ATTN_F32_FUSED = os.environ.get("GAOT_ATTN_F32_FUSED", "1") != "0"    # fp32 mode: the one-pass attention backward (A/B switch)
_ATTN_F32_FUSED_CAP = int(os.environ.get("GAOT_ATTN_F32_FUSED_MAX_MB", "4096")) << 20     # ... while its slab partials fit in this


def set_precision(mode: str):
    if mode not in ("fp32", "bf16"):
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {mode}")
    _PRECISION["mode"] = 0 if mode == "fp32" else 1


def get_precision() -> str:
    return "bf16" if _PRECISION["mode"] else "fp32"


# activation ids of the GEMM epilogues / gaot_act_bwd (csrc/common.h GAOT_ACT_*)
ACT = {None: 0, "none": 0, "gelu": 1, "relu": 2, "silu": 3, "tanh": 4, "leaky_relu": 5, "elu": 6, "sigmoid": 7, "softplus": 8,
       "selu": 9, "relu6": 10, "hardswish": 11, "mish": 12, "gelu_tanh": 13}


def gemm(a: Tensor, b: Tensor, m: int, n: int, k: int, lda: int, ldb: int, a_trans: bool, b_trans: bool,
         bias: Optional[Tensor] = None, act: int = 0, residual: Optional[Tensor] = None, ldr: int = 0,
         want_preact: bool = False, out: Optional[Tensor] = None, ldc: Optional[int] = None,
         precision: Optional[int] = None, out_dtype: torch.dtype = torch.float32):
    """Raw GEMM on 2-D row-major buffers (see include/gaot3d_hip.h: gaot_gemm / gaot_gemm_ex).  Operands are fp32;
    a bf16 tensor (an FFN intermediate of the bf16 path) is passed through as bf16-in-memory."""
    lib = _lib.load()
    dev = a.device
    if out is None:
        out = torch.empty(m, n, dtype=out_dtype, device=dev)
        ldc = n
    pre = torch.empty(m, n, dtype=torch.float32, device=dev) if want_preact else None
    if want_preact and ldc != n:
        raise GaotError("preact output requires a dense output")
    nb = lib.gaot_gemm_workspace_bytes(m, n, k)
    ws = _ws(nb, dev) if nb else None
    prec = _PRECISION["mode"] if precision is None else precision
    a16, b16, c16 = a.dtype == torch.bfloat16, b.dtype == torch.bfloat16, out.dtype == torch.bfloat16
    if a16 or b16 or c16:
        check(lib.gaot_gemm_ex(_ptr(a), _ptr(b), _ptr(out), m, n, k, lda, ldb, ldc, int(a_trans), int(b_trans), int(a16),
                               int(b16), int(c16), _ptr(bias), act, _ptr(residual), ldr, _ptr(pre), prec, _ptr(ws),
                               ws.numel() if ws is not None else 0, _stream()), "gaot_gemm_ex")
    else:
        check(lib.gaot_gemm(_ptr(a), _ptr(b), _ptr(out), m, n, k, lda, ldb, ldc, int(a_trans), int(b_trans), _ptr(bias),
                            act, _ptr(residual), ldr, _ptr(pre), prec, _ptr(ws), ws.numel() if ws is not None else 0,
                            _stream()), "gaot_gemm")
    return (out, pre) if want_preact else out


# ---- deferred completion of fixed-order reductions (include/gaot3d_hip.h, ABI 10) -------------------------------------------------
# Weight gradients are read by nobody before the optimizer step.  Their producers (split-K dy^T x products, RMSNorm weight
# gradients, bias column sums) leave their partials in a workspace; ONE gaot_reduce_multi launch at the end of the backward pass
# (an autograd-engine final callback: it runs before loss.backward() / torch.autograd.grad() return) sums them in the very order
# the in-call passes use -- the values are bit-identical, ~90 launches of ~5 us per configs[1] step become two.
class _ReduceDesc(C.Structure):   # gaot_reduce_desc_t
    _fields_ = [("part", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("parts", C.c_int32), ("lanes", C.c_int32)]


# The deferral is OPT-IN (round 6): a step that knows every parameter has ONE consumer that goes through these operators
# (bench.py, sharding.ShardedStep at world size 1, a trainer after reading INTEGRATION.md §5) calls ``defer_reductions(True)`` or sets
# GAOT_DEFER_REDUCE=1.  Off, every reduction completes inside the call that produced it.
_DEFER = {"enabled": os.environ.get("GAOT_DEFER_REDUCE", "0") == "1",
          "stack": [],            # [(graph-task id, {id(param)})]: the running backward pass and the passes it is nested in
          "pending": [], "pending_bytes": 0,
          "cap_bytes": int(os.environ.get("GAOT_DEFER_CAP_MB", "1024")) << 20}


def defer_reductions(enabled: bool) -> bool:
    """switch the deferral on / off (default OFF; GAOT_DEFER_REDUCE=1 turns it on); returns the previous setting.  Whatever is
    pending is completed first and the per-pass bookkeeping is dropped."""
    prev, _DEFER["enabled"] = _DEFER["enabled"], bool(enabled)
    _flush_pending()
    _DEFER["stack"].clear()
    return prev


def _task_done(task: int) -> None:
    """final callback of graph task ``task``: complete what is pending, forget the pass (and any nested pass above it that died
    without its callback).  A callback that runs outside every autograd node belongs to a TOP-LEVEL pass: nothing below it on the
    stack can be alive (a pass that raised half-way never runs its callbacks) -- drop it all."""
    _flush_pending()
    st = _DEFER["stack"]
    node = getattr(torch._C, "_current_autograd_node", None)
    if node is None or node() is None:
        st.clear()
        return
    for i, (t, _s) in enumerate(st):
        if t == task:
            del st[i:]
            break


def defer_ok(params) -> bool:
    """May the gradients of these parameters be completed at the end of the running backward pass?  Only when the caller opted in
    (``defer_reductions(True)``), inside a backward pass, and only for plain contiguous leaf parameters that nothing can observe
    earlier: no gradient to accumulate into yet (AccumulateGrad then takes the returned tensor itself, no arithmetic on it), no
    tensor / post-accumulate hooks (gradient buckets of a sharded step, user hooks), NO initialised process group whatever its
    size (DistributedDataParallel's reducer hooks the gradient accumulators in C++ at world size 1 too), no create_graph /
    anomaly mode.  A parameter that turns up a second time -- in this pass or in a pass this one is nested in -- completes
    everything pending first and takes the in-call pass.  A NESTED pass (reentrant ``torch.utils.checkpoint``, a custom Function
    calling ``torch.autograd.grad`` in its backward) first completes what the enclosing pass left pending, then defers on its own
    list, completed by its own final callback.  ``params=None`` (callers that cannot name their parameters): no.

    What this cannot see: a gradient that reaches one of these parameters through a plain torch op in the same
    pass (a weight regulariser, weight tying through torch ops) -- the engine would add it to a buffer whose values do not exist
    yet.  Hence opt-in."""
    if params is None or not _DEFER["enabled"]:
        return False
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized():
        return False
    get_task = getattr(torch._C, "_current_graph_task_id", None)      # (a private hook of the autograd engine: without it, no deferral)
    task = get_task() if get_task is not None else -1
    if task < 0 or torch.is_grad_enabled() or torch.is_anomaly_enabled():   # not in a backward pass / create_graph / NaN checks
        return False
    st = _DEFER["stack"]
    if not st or st[-1][0] != task:
        at = next((i for i, (t, _s) in enumerate(st) if t == task), -1)
        if at >= 0:
            del st[at + 1:]                 # nested passes above this one ended without their callback (they raised)
        else:
            # a pass we have not seen: top-level, or nested in the one on top of the stack.  What that one left pending is
            # completed NOW (its partials are all written, same stream; the storages are kept alive) -- never dropped
            _flush_pending()
            st.append((task, set()))
            torch.autograd.Variable._execution_engine.queue_callback(functools.partial(_task_done, task))
    ok = True
    for p in params:
        if p is None:
            continue
        if (not p.is_leaf) or p.grad_fn is not None or p.grad is not None or p._backward_hooks \
                or getattr(p, "_post_accumulate_grad_hooks", None) or not p.is_contiguous():
            ok = False
        if any(id(p) in seen for _t, seen in st):
            # the parameter is used twice: the engine is about to ADD this gradient to the one deferred earlier -- complete
            # everything pending now (same stream, ahead of that add) and take the in-call pass
            _flush_pending()
            ok = False
    if ok:
        st[-1][1].update(id(p) for p in params if p is not None)
    return ok


def _defer(part: Tensor, out: Tensor, n: int, parts: int, lanes: int) -> None:
    # `out` is about to be handed to autograd: keep its STORAGE alive, not the tensor -- a second reference to the returned tensor
    # (a view of it holds one too: its base) would make AccumulateGrad clone it, a copy of values that do not exist yet
    _DEFER["pending"].append((part, out.untyped_storage(), out.data_ptr(), int(n), int(parts), int(lanes)))
    # the partial tables stay allocated until the flush (configs[1]: ~70 MB per layer, 0.7 GB per pass): past the cap, complete now
    _DEFER["pending_bytes"] += part.numel() * part.element_size()
    if _DEFER["pending_bytes"] > _DEFER["cap_bytes"]:
        _flush_pending()


def flush_deferred() -> None:
    """sum every pending partial table into its output (one launch per 64 tables) -- safe at any time: the partials of a pending
    entry are complete in stream order the moment it is queued"""
    _flush_pending()


def _flush_pending() -> None:
    pend, _DEFER["pending"], _DEFER["pending_bytes"] = _DEFER["pending"], [], 0
    if not pend:
        return
    arr = (_ReduceDesc * len(pend))()
    for i, (part, _keep, out_ptr, n, parts, lanes) in enumerate(pend):
        arr[i].part, arr[i].out, arr[i].n, arr[i].parts, arr[i].lanes = part.data_ptr(), out_ptr, n, parts, lanes
    check(_lib.load().gaot_reduce_multi(arr, len(pend), _stream()), "gaot_reduce_multi")


def deferred_pending() -> int:
    return len(_DEFER["pending"])


def gemm_dw(a: Tensor, b: Tensor, m: int, n: int, k: int, lda: int, ldb: int, precision: Optional[int] = None,
            defer: bool = False) -> Tensor:
    """the weight-gradient product dW[m, n] = a^T b (a [k, lda], b [k, ldb], reduction over the k rows); ``defer``: its split-K
    completion waits for flush_deferred (see defer_ok)"""
    if not defer:
        return gemm(a, b, m, n, k, lda, ldb, True, False, precision=precision)
    lib = _lib.load()
    dev = a.device
    out = torch.empty(m, n, dtype=torch.float32, device=dev)
    nb = lib.gaot_gemm_workspace_bytes(m, n, k)
    ws = _ws(nb, dev) if nb else None
    prec = _PRECISION["mode"] if precision is None else precision
    splits, lanes = C.c_int(0), C.c_int(0)
    check(lib.gaot_gemm_ex_partials(_ptr(a), _ptr(b), _ptr(out), m, n, k, lda, ldb, n, 1, 0, int(a.dtype == torch.bfloat16),
                                    int(b.dtype == torch.bfloat16), prec, _ptr(ws), ws.numel() if ws is not None else 0,
                                    C.byref(splits), C.byref(lanes), _stream()), "gaot_gemm_ex_partials")
    if splits.value > 1:
        _defer(ws, out, m * n, splits.value, lanes.value)
    return out


def colsum(x: Tensor, m: int, n: int, ld: int, defer: bool = False) -> Tensor:
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    ws = _ws(lib.gaot_colsum_workspace_bytes(m, n), x.device)
    if defer and m > 0 and n > 0:
        check(lib.gaot_colsum(_ptr(x), m, n, ld, None, _ptr(ws), ws.numel(), _stream()), "gaot_colsum")
        _defer(ws, out, n, lib.gaot_colsum_parts(m), 32)
        return out
    check(lib.gaot_colsum(_ptr(x), m, n, ld, _ptr(out), _ptr(ws), ws.numel(), _stream()), "gaot_colsum")
    return out


def act_fwd(z: Tensor, act: int) -> Tensor:
    """h = act(z) for the activation ids the GEMM epilogue does not carry (ACT >= 4)"""
    lib = _lib.load()
    h = torch.empty_like(z)
    check(lib.gaot_act_fwd(_ptr(z), _ptr(h), z.numel(), act, _stream()), "gaot_act_fwd")
    return h


def act_bwd(z: Tensor, dh: Tensor, act: int) -> Tensor:
    lib = _lib.load()
    dz = torch.empty_like(z)
    check(lib.gaot_act_bwd(_ptr(z), _ptr(dh), _ptr(dz), z.numel(), act, _stream()), "gaot_act_bwd")
    return dz


def axpy(a: Tensor, b: Tensor, alpha: float = 1.0, period: Optional[int] = None) -> Tensor:
    lib = _lib.load()
    out = torch.empty_like(a)
    check(lib.gaot_axpy(_ptr(a), _ptr(b), float(alpha), _ptr(out), a.numel(), period or a.numel(), _stream()),
          "gaot_axpy")
    return out


def stream_copy_gbps(nbytes: int = 1 << 30, reps: int = 10, variant: Optional[int] = None) -> float:
    """measured float4 streaming-copy rate of this device in GB/s (read + written bytes over the HIP-event time of ``reps``
    copies of ``nbytes``): the practical HBM ceiling quoted beside the 8 TB/s spec figure.  ``variant``: a form of
    gaot_stream_copy_ex (default: the library's fastest)"""
    lib = _lib.load()
    src = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)

    def go():
        if variant is None:
            check(lib.gaot_stream_copy(_ptr(src), _ptr(dst), nbytes, _stream()), "gaot_stream_copy")
        else:
            check(lib.gaot_stream_copy_ex(_ptr(src), _ptr(dst), nbytes, int(variant), _stream()), "gaot_stream_copy_ex")
    for _ in range(2):
        go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        go()
    b.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9


def rmsnorm_fwd(x: Tensor, w: Tensor, eps: float, want_bf16: bool = False):
    """-> (y, rstd, y_bf16 | None); the bf16 copy (same rounding a GEMM would apply to y as its A operand) is written by
    the same pass"""
    lib = _lib.load()
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    yb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if (want_bf16 and d % 8 == 0) else None
    check(lib.gaot_rmsnorm_fwd(_ptr(x), _ptr(w), _ptr(y), _ptr(rstd), _ptr(yb), rows, d, float(eps), _stream()),
          "gaot_rmsnorm_fwd")
    return y, rstd, yb


def rmsnorm_bwd(x: Tensor, w: Tensor, dy: Tensor, rstd: Tensor, dx_add: Optional[Tensor] = None, defer: bool = False,
                dx_add2: Optional[Tensor] = None):
    """``defer``: the weight gradient's partial rows are summed by flush_deferred (see defer_ok); ``dx_add`` / ``dx_add2``: gradients
    that reach x through other consumers (the block's residual; the U-ViT skip tap), added by the same pass"""
    lib = _lib.load()
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    ws = _ws(lib.gaot_rmsnorm_bwd_workspace_bytes(rows, d), x.device)
    if dx_add is not None:
        dx_add = _req(dx_add, torch.float32, "dx_add")
    if dx_add2 is not None:
        dx_add2 = _req(dx_add2, torch.float32, "dx_add2")
        if dx_add is None:
            dx_add, dx_add2 = dx_add2, None
    defer = defer and rows > 0
    if dx_add2 is not None:
        check(lib.gaot_rmsnorm_bwd2(_ptr(x), _ptr(w), _ptr(dy), _ptr(rstd), _ptr(dx_add), _ptr(dx_add2), _ptr(dx), None if defer else _ptr(dw),
                                    rows, d, _ptr(ws), ws.numel(), _stream()), "gaot_rmsnorm_bwd2")
    else:
        check(lib.gaot_rmsnorm_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(rstd), _ptr(dx_add), _ptr(dx), None if defer else _ptr(dw), rows, d,
                                   _ptr(ws), ws.numel(), _stream()), "gaot_rmsnorm_bwd")
    if defer:
        _defer(ws, dw, d, lib.gaot_rmsnorm_bwd_parts(rows), 32)
    return dx, dw


def rope_(buf: Tensor, rows: int, ld: int, col0: int, nheads: int, seq_len: int, freqs: Tensor, inverse: bool,
          head_dim: int = 32):
    lib = _lib.load()
    check(lib.gaot_rope(_ptr(buf), rows, ld, col0, nheads, head_dim, seq_len, _ptr(freqs), int(inverse), _stream()),
          "gaot_rope")


def row_softmax(scores: Tensor, grad: Optional[Tensor] = None, w: Optional[Tensor] = None) -> Tensor:
    """softmax over the last axis of a dense [rows, n] fp32 matrix (grad / w given: its backward ds = w (dw - sum(w dw))):
    the per-row segment kernels of csrc/edgeops.hip on the row pointer 0, n, 2n, ..."""
    lib = _lib.load()
    rows, n = scores.shape
    if rows * n >= 2 ** 31:
        raise GaotError("row_softmax: more than 2^31 scores")
    rowptr = torch.arange(0, rows * n + 1, n, dtype=torch.int32, device=scores.device)
    out = torch.empty_like(scores)
    if grad is None:
        check(lib.gaot_segment_softmax_fwd(_ptr(_req(scores, torch.float32, "scores")), _ptr(rowptr), rows, _ptr(out), _stream()),
              "gaot_segment_softmax_fwd")
    else:
        check(lib.gaot_segment_softmax_bwd(_ptr(w), _ptr(_req(grad, torch.float32, "dw")), _ptr(rowptr), rows, _ptr(out), _stream()),
              "gaot_segment_softmax_bwd")
    return out


def _drop_args(p: float, seed: Optional[Tensor]):
    if p <= 0.0:
        return 0.0, None
    if seed is None or seed.dtype != torch.int64 or not seed.is_cuda or seed.numel() != 1:
        raise ValueError("attention dropout needs a one-element int64 device tensor as seed")
    return float(p), _ptr(seed)


def dropout_seed_next(state: Tensor, stride: int) -> Tensor:
    """-> a fresh one-element int64 tensor holding the current seed word; ``state`` advances by ``stride`` (one launch)"""
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.int64, device=state.device)
    check(lib.gaot_dropout_seed_next(_ptr(state), stride & 0xFFFFFFFFFFFFFFFF, _ptr(out), _stream()), "gaot_dropout_seed_next")
    return out


def dropout_seed_block(state: Tensor, stride: int, n: int) -> Tensor:
    """-> int64 [n]: the seed words of the next n dropout calls; ``state`` advances by n * stride (one launch)"""
    lib = _lib.load()
    out = torch.empty(n, dtype=torch.int64, device=state.device)
    check(lib.gaot_dropout_seed_block(_ptr(state), stride & 0xFFFFFFFFFFFFFFFF, int(n), _ptr(out), _stream()), "gaot_dropout_seed_block")
    return out


def dropout(x: Tensor, seed: Tensor, p: float) -> Tensor:
    """x .* keep / (1 - p) with keep(i) = hash(seed, i) >= round(p 2^32)  (include/gaot3d_hip.h: gaot_dropout)"""
    lib = _lib.load()
    x = _req(x, torch.float32, "x")
    out = torch.empty_like(x)
    check(lib.gaot_dropout(_ptr(x), _ptr(seed), float(p), x.numel(), _ptr(out), _stream()), "gaot_dropout")
    return out


def attn_dropout_mask(seed: Tensor, p: float, b: int, h: int, s: int) -> Tensor:
    """keep[b, h, q, k] (uint8) of the attention dropout mask the kernels regenerate from ``seed`` (checks only)"""
    lib = _lib.load()
    keep = torch.empty(b, h, s, s, dtype=torch.uint8, device=seed.device)
    check(lib.gaot_attn_dropout_mask(_ptr(seed), float(p), b, h, s, _ptr(keep), _stream()), "gaot_attn_dropout_mask")
    return keep


def attn_fwd(qkv: Tensor, b: int, s: int, h: int, hkv: int, scale: float, dropout_p: float = 0.0,
             seed: Optional[Tensor] = None, head0: int = 0, heads_total: int = 0):
    """qkv: [B*S, (h + 2*hkv)*32] fused projection output (q | k | v column blocks).  ``head0`` / ``heads_total``: the h heads
    are heads head0 .. of heads_total (a rank's slice): the dropout mask is keyed by the global head index"""
    lib = _lib.load()
    ld = qkv.shape[1]
    dev = qkv.device
    o = torch.empty(b * s, h * 32, dtype=torch.float32, device=dev)
    lse = torch.empty(b, h, s, dtype=torch.float32, device=dev)
    base = qkv.data_ptr()
    q, k, v = C.c_void_p(base), C.c_void_p(base + 4 * h * 32), C.c_void_p(base + 4 * (h + hkv) * 32)
    with _timed("attn_fwd"):
        dp, sp = _drop_args(dropout_p, seed)
        check(lib.gaot_attn_fwd(q, k, v, _ptr(o), _ptr(lse), ld, ld, ld, h * 32, b, s, h, hkv, 32, float(scale), dp, sp,
                                int(head0), int(heads_total), _PRECISION["mode"], _stream()), "gaot_attn_fwd")
    return o, lse


def attn_bwd(qkv: Tensor, o: Tensor, d_o: Tensor, lse: Tensor, b: int, s: int, h: int, hkv: int, scale: float,
             dropout_p: float = 0.0, seed: Optional[Tensor] = None, head0: int = 0, heads_total: int = 0) -> Tensor:
    lib = _lib.load()
    dp, sp = _drop_args(dropout_p, seed)
    ld = qkv.shape[1]
    dev = qkv.device
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(b, h, s, dtype=torch.float32, device=dev)
    base, gbase = qkv.data_ptr(), dqkv.data_ptr()
    offk, offv = 4 * h * 32, 4 * (h + hkv) * 32
    # dK, dV and dQ from one pass (5 S^2 d products instead of 7); its fp32 dQ slab partials live in a buffer of this call -- b * h *
    # ceil(s / 256) * s * 128 bytes (1.07 GB at S = 16 384, 8 heads; quadratic in S): past the cap the two-pass kernels run
    nb = int(lib.gaot_attn_bwd_fused_f32_scratch_bytes(b, s, h)) if _ATTN_F32_FUSED else 0
    # (a long sequence with few kv heads per launch -- one rank's share of a sharded step -- leaves the one-pass kernel fewer than one
    # workgroup per CU, 256 keys each; the two-pass kernels split twice as fine)
    starved = s >= 4096 and b * hkv * ((s + 255) // 256) < 256
    if _ATTN_F32_FUSED and nb <= _ATTN_F32_FUSED_CAP and not starved:
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        with _timed("attn_bwd_delta"):
            check(lib.gaot_attn_bwd(C.c_void_p(base), C.c_void_p(base + offk), C.c_void_p(base + offv), _ptr(o),
                                    _ptr(d_o), _ptr(lse), _ptr(delta), C.c_void_p(gbase), C.c_void_p(gbase + offk),
                                    C.c_void_p(gbase + offv), ld, ld, ld, h * 32, h * 32, ld, ld, ld, b, s, h, hkv, 32,
                                    float(scale), dp, sp, int(head0), int(heads_total), _PRECISION["mode"], 1, _stream()),
                  "gaot_attn_bwd")
        with _timed("attn_bwd_fused_f32"):
            check(lib.gaot_attn_bwd_fused_f32(C.c_void_p(base), C.c_void_p(base + offk), C.c_void_p(base + offv), _ptr(o),
                                              _ptr(d_o), _ptr(lse), _ptr(delta), C.c_void_p(gbase), C.c_void_p(gbase + offk),
                                              C.c_void_p(gbase + offv), ld, ld, ld, h * 32, h * 32, ld, ld, ld, b, s, h, hkv, 32,
                                              float(scale), dp, sp, int(head0), int(heads_total), 0, _ptr(scratch), nb, _stream()),
                  "gaot_attn_bwd_fused_f32")
        return dqkv
    for name, mask in (("attn_bwd_delta", 1), ("attn_bwd_dkv", 2), ("attn_bwd_dq", 4)):
        with _timed(name):
            check(lib.gaot_attn_bwd(C.c_void_p(base), C.c_void_p(base + offk), C.c_void_p(base + offv), _ptr(o),
                                    _ptr(d_o), _ptr(lse), _ptr(delta), C.c_void_p(gbase), C.c_void_p(gbase + offk),
                                    C.c_void_p(gbase + offv), ld, ld, ld, h * 32, h * 32, ld, ld, ld, b, s, h, hkv, 32,
                                    float(scale), dp, sp, int(head0), int(heads_total), _PRECISION["mode"], mask, _stream()),
                  "gaot_attn_bwd")
    return dqkv


def attn_fwd_bf16(qkv: Optional[Tensor], freqs: Optional[Tensor], b: int, s: int, h: int, hkv: int, scale: float,
                  dropout_p: float = 0.0, seed: Optional[Tensor] = None, image: Optional[Tensor] = None, head0: int = 0,
                  heads_total: int = 0):
    """bf16 matrix-core attention on the fused fp32 projection; returns (o, lse, bf16 image kept for backward).
    ``image``: the projection already written as the kernels' image (qkv_image): qkv is then not read"""
    lib = _lib.load()
    dev = image.device if image is not None else qkv.device
    o = torch.empty(b * s, h * 32, dtype=torch.float32, device=dev)
    lse = torch.empty(b, h, s, dtype=torch.float32, device=dev)
    img = image if image is not None else _ws(lib.gaot_attn_bf16_image_bytes(b, s, h, hkv), dev)
    with _timed("attn_fwd"):
        dp, sp = _drop_args(dropout_p, seed)
        check(lib.gaot_attn_fwd_bf16(_ptr(None if image is not None else qkv), _ptr(freqs), _ptr(img), _ptr(o), _ptr(lse), b, s,
                                     h, hkv, 32, float(scale), dp, sp, int(head0), int(heads_total), _stream()), "gaot_attn_fwd_bf16")
    return o, lse, img


_ROPE_TABLES: dict = {}


def rope_table(freqs: Tensor, s: int) -> Tensor:
    """[s, 16, 2] (cos, sin) of position * frequency for a 32-wide head, cached per (frequencies, s): the frequencies are a
    frozen parameter, so the table is built once"""
    key = (id(freqs), int(s))
    ent = _ROPE_TABLES.get(key)
    # the entry holds the frequencies tensor itself: validated by identity + in-place version, never by address (an
    # address can be re-used by another model's frequencies)
    if ent is not None and ent[0] is freqs and ent[1] == freqs._version:
        _ROPE_TABLES[key] = _ROPE_TABLES.pop(key)     # most recently used last
        return ent[2]
    if freqs.numel() != 16 or freqs.dtype != torch.float32:
        raise GaotError("rope_table: 16 fp32 frequencies (head_dim 32) expected")
    t = torch.empty(s, 16, 2, dtype=torch.float32, device=freqs.device)
    check(_lib.load().gaot_rope_table(_ptr(freqs), int(s), 16, _ptr(t), _stream()), "gaot_rope_table")
    _ROPE_TABLES.pop(key, None)
    _ROPE_TABLES[key] = (freqs, freqs._version, t)
    while len(_ROPE_TABLES) > 64:                     # evict the least recently used entry only (a captured graph may
        _ROPE_TABLES.pop(next(iter(_ROPE_TABLES)))    # still reference the tables of the live models)
    return t


def qkv_image(xb: Tensor, wcat: Tensor, rows: int, b: int, s: int, h: int, hkv: int, freqs: Optional[Tensor],
              scale: float) -> Tensor:
    """x [rows, 256] bf16 times the co-located q|k|v weights [(h + 2 hkv) * 32, 256] bf16, written straight as the attention
    kernels' bf16 image (RoPE on q and k, q pre-scaled): the fp32 projection never exists (csrc/gemm_k256.hip)"""
    lib = _lib.load()
    if xb.dtype != torch.bfloat16 or wcat.dtype != torch.bfloat16 or not (xb.is_contiguous() and wcat.is_contiguous()):
        raise GaotError("qkv_image: contiguous bf16 operands expected")
    img = _ws(lib.gaot_attn_bf16_image_bytes(b, s, h, hkv), xb.device)
    table = rope_table(freqs, s) if freqs is not None else None
    with _timed("qkv_image"):
        check(lib.gaot_qkv_image(_ptr(xb), _ptr(wcat), _ptr(img), rows, xb.shape[1], wcat.shape[1], s, h, hkv, _ptr(table),
                                 _qscale(scale), _stream()), "gaot_qkv_image")
    return img


def attn_bwd_scratch(b: int, s: int, h: int, hkv: int, device) -> Tensor:
    """the backward's scratch buffer; its first b*s*h*32 bf16 elements are the dO image"""
    return _ws(_lib.load().gaot_attn_bwd_bf16_scratch_bytes(b, s, h, hkv), device)


def attn_bwd_bf16(img: Tensor, o: Tensor, d_o: Optional[Tensor], lse: Tensor, b: int, s: int, h: int, hkv: int, scale: float,
                  dropout_p: float = 0.0, seed: Optional[Tensor] = None, freqs: Optional[Tensor] = None,
                  do_image: Optional[Tensor] = None, fused: Optional[bool] = None, head0: int = 0, heads_total: int = 0,
                  delta: Optional[Tensor] = None) -> Tensor:
    """``freqs``: the forward's RoPE frequencies -> the returned dq / dk are w.r.t. the UNrotated projection.
    ``do_image`` (instead of d_o): an attn_bwd_scratch buffer whose head already holds the bf16 dO (sequence-parallel
    exchange): only delta is computed from it.
    ``fused``: dK / dV / dQ from one pass over the score tiles (csrc/attn_bf16.hip: k_attn_bwd_fused) instead of the dK/dV
    pass + the dQ pass; default: whenever its grid fills the chip (gaot_attn_bwd_bf16_fused_eligible)"""
    lib = _lib.load()
    dp, sp = _drop_args(dropout_p, seed)
    dev = o.device
    dqkv = torch.empty(b * s, (h + 2 * hkv) * 32, dtype=torch.float32, device=dev)
    have_delta = delta is not None      # ``delta`` (with do_image): the row constants are already there too (oproj_bwd_image): no first phase
    if have_delta and (do_image is None or tuple(delta.shape) != (b, h, s) or delta.dtype != torch.float32):
        raise GaotError("attn_bwd_bf16: delta needs do_image and the shape [b, h, s]")
    if not have_delta:
        delta = torch.empty(b, h, s, dtype=torch.float32, device=dev)
    doimg = do_image if do_image is not None else attn_bwd_scratch(b, s, h, hkv, dev)
    can_fuse = bool(lib.gaot_attn_bwd_bf16_fused_eligible(b, s, h, hkv))
    if fused and not can_fuse:
        raise GaotError("attn_bwd_bf16: the fused backward needs >= 128 workgroups (ceil(S/512) * hkv * b, times up to 8 query "
                        "parts of >= 1024 rows) and <= 1 GiB of dQ slab partials")
    phases = ((("attn_bwd", 16), ("attn_bwd_dq_reduce", 32)) if (can_fuse if fused is None else fused)
              else (("attn_bwd_dkv", 2), ("attn_bwd_dq", 4)))
    first = () if have_delta else (("attn_bwd_delta", 8 if do_image is not None else 1),)
    for name, mask in first + phases:
        with _timed(name):
            check(lib.gaot_attn_bwd_bf16(_ptr(img), _ptr(o), _ptr(d_o), _ptr(lse), _ptr(doimg), _ptr(delta), _ptr(dqkv),
                                         _ptr(freqs), b, s, h, hkv, 32, float(scale), dp, sp, int(head0), int(heads_total), mask,
                                         _stream()), "gaot_attn_bwd_bf16")
    return dqkv


def _qscale(scale: float) -> float:
    """fp32(scale) * fp32(log2 e), rounded once in fp32 (host arithmetic: numpy, not a torch op on the step's path)"""
    import numpy as np
    return float(np.float32(scale) * np.float32(1.4426950408889634))


def qkv_image_packed(xb: Tensor, wcat: Tensor, rows: int, pos0: int, s_total: int, h: int, hkv: int, freqs: Optional[Tensor],
                     scale: float, world: int) -> Tensor:
    """the q|k|v projection of this rank's token rows written as the send buffer of the sequence-parallel all-to-all:
    bf16 [world, rows, (h + 2 hkv) / world * 32] (include/gaot3d_hip.h: gaot_qkv_image_packed)"""
    lib = _lib.load()
    if xb.dtype != torch.bfloat16 or wcat.dtype != torch.bfloat16 or not (xb.is_contiguous() and wcat.is_contiguous()):
        raise GaotError("qkv_image_packed: contiguous bf16 operands expected")
    lw = (h + 2 * hkv) // world * 32
    out = torch.empty(world, rows, lw, dtype=torch.bfloat16, device=xb.device)
    table = rope_table(freqs, s_total) if freqs is not None else None
    with _timed("qkv_image"):
        check(lib.gaot_qkv_image_packed(_ptr(xb), _ptr(wcat), _ptr(out), rows, xb.shape[1], wcat.shape[1], int(pos0), h, hkv,
                                        _ptr(table), _qscale(scale), world, _stream()), "gaot_qkv_image_packed")
    return out


_DT = {torch.float32: 0, torch.bfloat16: 1}


def pack_heads(rows_buf: Tensor, blocks: Tensor, world: int, segs, to_blocks: bool) -> None:
    """rows layout [rows, ld] <-> blocks layout [world, rows, sum(width)]; ``segs`` = [(col0, width), ...] (<= 3 column
    segments, each `world` groups of `width` columns); fp32 or bf16 on either side (include/gaot3d_hip.h: gaot_pack_heads)"""
    lib = _lib.load()
    if not (rows_buf.is_contiguous() and blocks.is_contiguous() and rows_buf.is_cuda and blocks.is_cuda):
        raise GaotError("pack_heads: contiguous device tensors expected")
    rows, ld = rows_buf.shape
    col0 = (C.c_int * len(segs))(*[int(c) for c, _ in segs])
    wid = (C.c_int * len(segs))(*[int(w) for _, w in segs])
    if tuple(blocks.shape) != (world, rows, sum(int(w) for _, w in segs)):
        raise GaotError(f"pack_heads: blocks buffer has shape {tuple(blocks.shape)}")
    check(lib.gaot_pack_heads(_ptr(rows_buf), _ptr(blocks), rows, ld, world, len(segs), col0, wid, _DT[rows_buf.dtype],
                              _DT[blocks.dtype], int(to_blocks), _stream()), "gaot_pack_heads")


def swiglu_fwd(ag: Tensor, f: int) -> Tensor:
    lib = _lib.load()
    rows = ag.shape[0]
    u = torch.empty(rows, f, dtype=torch.float32, device=ag.device)
    check(lib.gaot_swiglu_fwd(_ptr(ag), _ptr(u), rows, f, _stream()), "gaot_swiglu_fwd")
    return u


def swiglu_bwd(ag: Tensor, du: Tensor, f: int) -> Tensor:
    lib = _lib.load()
    dag = torch.empty_like(ag)
    check(lib.gaot_swiglu_bwd(_ptr(ag), _ptr(du), _ptr(dag), ag.shape[0], f, _stream()), "gaot_swiglu_bwd")
    return dag


def cast_bf16(x: Tensor) -> Tensor:
    """bf16 copy (round to nearest even) of a contiguous fp32 tensor"""
    lib = _lib.load()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.gaot_cast_bf16(_ptr(x), _ptr(out), x.numel(), _stream()), "gaot_cast_bf16")
    return out


class _CastEntry(C.Structure):   # gaot_cast_tensor_t
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("numel", C.c_int64)]


def cast_bf16_multi(xs: Sequence[Tensor]) -> List[Tensor]:
    """bf16 copies of many contiguous fp32 tensors in ONE launch; the copies are slices of one buffer (16-byte aligned)"""
    lib = _lib.load()
    if not xs:
        return []
    offs, total = [], 0
    for x in xs:
        offs.append(total)
        total += (x.numel() + 7) // 8 * 8
    buf = torch.empty(total, dtype=torch.bfloat16, device=xs[0].device)
    outs = [buf[o:o + x.numel()].view(x.shape) for o, x in zip(offs, xs)]
    entries = (_CastEntry * len(xs))()
    for i, (x, o) in enumerate(zip(xs, outs)):
        x = _req(x, torch.float32, "x")
        entries[i] = _CastEntry(x.data_ptr(), o.data_ptr(), x.numel())
    check(lib.gaot_cast_bf16_multi(entries, len(xs), _stream()), "gaot_cast_bf16_multi")
    return outs


def cast_bf16_transpose_multi(xs: Sequence[Tensor]) -> List[Tensor]:
    """bf16 TRANSPOSED copies ([cols, rows]) of many contiguous fp32 matrices in ONE launch"""
    lib = _lib.load()
    if not xs:
        return []
    offs, total = [], 0
    for x in xs:
        offs.append(total)
        total += (x.numel() + 7) // 8 * 8
    buf = torch.empty(total, dtype=torch.bfloat16, device=xs[0].device)
    outs = [buf[o:o + x.numel()].view(x.shape[1], x.shape[0]) for o, x in zip(offs, xs)]
    entries = (_CastEntry * len(xs))()
    rows, cols = (C.c_int * len(xs))(), (C.c_int * len(xs))()
    for i, (x, o) in enumerate(zip(xs, outs)):
        x = _req(x, torch.float32, "x")
        if x.dim() != 2 or not x.is_contiguous():
            raise GaotError("cast_bf16_transpose_multi: contiguous 2-D tensors expected")
        entries[i] = _CastEntry(x.data_ptr(), o.data_ptr(), x.numel())
        rows[i], cols[i] = x.shape[0], x.shape[1]
    check(lib.gaot_cast_bf16_transpose_multi(entries, rows, cols, len(xs), _stream()), "gaot_cast_bf16_transpose_multi")
    return outs


def swiglu_fwd_bf16(ag: Tensor, f: int) -> Tensor:
    """bf16 [rows, 2F] -> bf16 [rows, F]"""
    lib = _lib.load()
    u = torch.empty(ag.shape[0], f, dtype=torch.bfloat16, device=ag.device)
    check(lib.gaot_swiglu_fwd_bf16(_ptr(ag), _ptr(u), ag.shape[0], f, _stream()), "gaot_swiglu_fwd_bf16")
    return u


def ffn_w13_swiglu(xb: Tensor, w13b: Tensor, f: int):
    """x [rows, 256] bf16 times the co-located [w1; w3] bf16 weights -> (a | g bf16 [rows, 2F], silu(a) g bf16 [rows, F]) in one
    launch (include/gaot3d_hip.h: gaot_ffn_w13_swiglu)"""
    lib = _lib.load()
    if xb.dtype != torch.bfloat16 or w13b.dtype != torch.bfloat16 or not (xb.is_contiguous() and w13b.is_contiguous()):
        raise GaotError("ffn_w13_swiglu: contiguous bf16 operands expected")
    rows = xb.shape[0]
    ag = torch.empty(rows, 2 * f, dtype=torch.bfloat16, device=xb.device)
    u = torch.empty(rows, f, dtype=torch.bfloat16, device=xb.device)
    with _timed("ffn_w13_swiglu"):
        check(lib.gaot_ffn_w13_swiglu(_ptr(xb), _ptr(w13b), _ptr(ag), _ptr(u), rows, xb.shape[1], w13b.shape[1], int(f), _stream()),
              "gaot_ffn_w13_swiglu")
    return ag, u


def ffn_pack(w13: Tensor, w2: Tensor, f: int, with_backward: bool) -> Tensor:
    """the fp32 co-located [w1; w3] ([2F, 256]) and w2 ([256, F]) as the fragment-ordered bf16 images the fused FFN kernels stream
    (include/gaot3d_hip.h: gaot_ffn_pack); one uint8 buffer"""
    lib = _lib.load()
    if w13.dtype != torch.float32 or w2.dtype != torch.float32 or not (w13.is_contiguous() and w2.is_contiguous()):
        raise GaotError("ffn_pack: contiguous fp32 weights expected")
    if tuple(w13.shape) != (2 * f, 256) or tuple(w2.shape) != (256, f):
        raise GaotError(f"ffn_pack: expected [2F, 256] and [256, F] with F = {f}, got {tuple(w13.shape)} and {tuple(w2.shape)}")
    packed = torch.empty(lib.gaot_ffn_packed_bytes(int(f), int(with_backward)), dtype=torch.uint8, device=w13.device)
    check(lib.gaot_ffn_pack(_ptr(w13), _ptr(w2), int(f), _ptr(packed), int(with_backward), _stream()), "gaot_ffn_pack")
    return packed


class _FfnPackItem(C.Structure):   # gaot_ffn_pack_t
    _fields_ = [("w13", C.c_void_p), ("w2", C.c_void_p), ("packed", C.c_void_p)]


def ffn_pack_multi(pairs, f: int, with_backward: bool) -> List[Tensor]:
    """ffn_pack for every (w13, w2) pair of a Transformer in ONE launch; the images are slices of one buffer"""
    lib = _lib.load()
    pairs = list(pairs)
    if not pairs:
        return []
    nb = int(lib.gaot_ffn_packed_bytes(int(f), int(with_backward)))
    buf = torch.empty(len(pairs) * nb, dtype=torch.uint8, device=pairs[0][0].device)
    outs = [buf[i * nb:(i + 1) * nb] for i in range(len(pairs))]
    items = (_FfnPackItem * len(pairs))()
    for i, ((w13, w2), o) in enumerate(zip(pairs, outs)):
        if w13.dtype != torch.float32 or w2.dtype != torch.float32 or not (w13.is_contiguous() and w2.is_contiguous()) \
                or tuple(w13.shape) != (2 * f, 256) or tuple(w2.shape) != (256, f):
            raise GaotError("ffn_pack_multi: contiguous fp32 [2F, 256] / [256, F] weights expected")
        items[i] = _FfnPackItem(w13.data_ptr(), w2.data_ptr(), o.data_ptr())
    check(lib.gaot_ffn_pack_multi(items, len(pairs), int(f), int(with_backward), _stream()), "gaot_ffn_pack_multi")
    return outs


def ffn_fwd(xb: Tensor, packed: Tensor, f: int, residual: Optional[Tensor] = None, save: bool = True):
    """y = w2(silu(w1 x) * w3 x) + residual in one launch (include/gaot3d_hip.h: gaot_ffn_fwd): x [rows, 256] bf16, ``packed`` from
    ffn_pack -> (y fp32 [rows, 256], a | g bf16 [rows, 2F] or None, u bf16 [rows, F] or None)"""
    lib = _lib.load()
    if xb.dtype != torch.bfloat16 or not xb.is_contiguous() or xb.shape[1] != 256:
        raise GaotError("ffn_fwd: contiguous bf16 [rows, 256] input expected")
    rows = xb.shape[0]
    if residual is not None and (residual.dtype != torch.float32 or residual.stride(-1) != 1 or tuple(residual.shape) != (rows, 256)):
        raise GaotError("ffn_fwd: fp32 [rows, 256] residual expected")
    y = torch.empty(rows, 256, dtype=torch.float32, device=xb.device)
    ag = torch.empty(rows, 2 * f, dtype=torch.bfloat16, device=xb.device) if save else None
    u = torch.empty(rows, f, dtype=torch.bfloat16, device=xb.device) if save else None
    with _timed("ffn_fwd"):
        check(lib.gaot_ffn_fwd(_ptr(xb), _ptr(packed), _ptr(residual), residual.stride(0) if residual is not None else 0, _ptr(y),
                               _ptr(ag), _ptr(u), rows, int(f), _stream()), "gaot_ffn_fwd")
    return y, ag, u


class _BlockPackItem(C.Structure):   # gaot_block_pack_t
    _fields_ = [("w13", C.c_void_p), ("w2", C.c_void_p), ("wo", C.c_void_p), ("packed", C.c_void_p)]


def block_pack_multi(triples, f: int) -> List[Tensor]:
    """ffn_pack_multi with the backward images plus the fragment image of o_proj.weight, for every (w13, w2, wo) of a Transformer in ONE
    launch (include/gaot3d_hip.h: gaot_block_pack_multi); the images are slices of one buffer"""
    lib = _lib.load()
    triples = list(triples)
    if not triples:
        return []
    nb = int(lib.gaot_block_packed_bytes(int(f)))
    buf = torch.empty(len(triples) * nb, dtype=torch.uint8, device=triples[0][0].device)
    outs = [buf[i * nb:(i + 1) * nb] for i in range(len(triples))]
    items = (_BlockPackItem * len(triples))()
    for i, ((w13, w2, wo), o) in enumerate(zip(triples, outs)):
        for t, shp in ((w13, (2 * f, 256)), (w2, (256, f)), (wo, (256, 256))):
            if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shp:
                raise GaotError("block_pack_multi: contiguous fp32 [2F, 256] / [256, F] / [256, 256] weights expected")
        items[i] = _BlockPackItem(w13.data_ptr(), w2.data_ptr(), wo.data_ptr(), o.data_ptr())
    check(lib.gaot_block_pack_multi(items, len(triples), int(f), _stream()), "gaot_block_pack_multi")
    return outs


def block_tail_fwd(attn_out: Tensor, x: Tensor, norm_weight: Tensor, eps: float, packed: Tensor, f: int):
    """h = x + o_proj(attn_out); n = RMSNorm(h); y = n + ffn(n) in one launch (include/gaot3d_hip.h: gaot_block_tail_fwd) ->
    (y fp32 [rows, 256], h fp32 [rows, 256], yb = bf16(n), rstd [rows]); ``packed`` from block_pack_multi"""
    lib = _lib.load()
    rows = attn_out.shape[0]
    for t, nm in ((attn_out, "attn_out"), (x, "x")):
        if t.dtype != torch.float32 or t.dim() != 2 or tuple(t.shape) != (rows, 256) or t.stride(1) != 1:
            raise GaotError(f"block_tail_fwd: fp32 [rows, 256] {nm} expected")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    dev = attn_out.device
    y = torch.empty(rows, 256, dtype=torch.float32, device=dev)
    h = torch.empty(rows, 256, dtype=torch.float32, device=dev)
    yb = torch.empty(rows, 256, dtype=torch.bfloat16, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    with _timed("block_tail_fwd"):
        check(lib.gaot_block_tail_fwd(_ptr(attn_out), attn_out.stride(0), _ptr(x), x.stride(0), _ptr(nw), float(eps), _ptr(packed), _ptr(h),
                                      _ptr(y), _ptr(yb), _ptr(rstd), rows, int(f), _stream()), "gaot_block_tail_fwd")
    return y, h, yb, rstd


class _QkvPackItem(C.Structure):   # gaot_qkv_pack_t
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p)]


def qkv_pack_multi(ws, with_backward: bool) -> List[Tensor]:
    """the co-located fp32 q | k | v weights ([N, 256]) of every block as fragment images in ONE launch (gaot_qkv_pack_multi)"""
    lib = _lib.load()
    ws = list(ws)
    if not ws:
        return []
    n = ws[0].shape[0]
    nb = int(lib.gaot_qkv_packed_bytes(n, int(with_backward)))
    buf = torch.empty(len(ws) * nb, dtype=torch.uint8, device=ws[0].device)
    outs = [buf[i * nb:(i + 1) * nb] for i in range(len(ws))]
    items = (_QkvPackItem * len(ws))()
    for i, (w, o) in enumerate(zip(ws, outs)):
        if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (n, 256):
            raise GaotError("qkv_pack_multi: contiguous fp32 [N, 256] weights of one N expected")
        items[i] = _QkvPackItem(w.data_ptr(), o.data_ptr())
    check(lib.gaot_qkv_pack_multi(items, len(ws), n, int(with_backward), _stream()), "gaot_qkv_pack_multi")
    return outs


def skip_pack_multi(ws) -> List[Tensor]:
    """skip_proj weights ([256, 512] fp32) of the decoder blocks as fragment images in ONE launch (gaot_skip_pack_multi)"""
    lib = _lib.load()
    ws = list(ws)
    if not ws:
        return []
    nb = int(lib.gaot_skip_packed_bytes())
    buf = torch.empty(len(ws) * nb, dtype=torch.uint8, device=ws[0].device)
    outs = [buf[i * nb:(i + 1) * nb] for i in range(len(ws))]
    items = (_QkvPackItem * len(ws))()
    for i, (w, o) in enumerate(zip(ws, outs)):
        if w.dtype != torch.float32 or not w.is_contiguous() or tuple(w.shape) != (256, 512):
            raise GaotError("skip_pack_multi: contiguous fp32 [256, 512] weights expected")
        items[i] = _QkvPackItem(w.data_ptr(), o.data_ptr())
    check(lib.gaot_skip_pack_multi(items, len(ws), _stream()), "gaot_skip_pack_multi")
    return outs


def cat_norm_qkv_image(xa: Tensor, xb: Tensor, skip_packed: Tensor, skip_bias: Optional[Tensor], norm_weight: Tensor, eps: float, packed: Tensor,
                       b: int, s: int, h: int, hkv: int, freqs: Optional[Tensor], scale: float):
    """skip_proj(cat([xa, xb])) + attn_norm + q | k | v image in one launch (gaot_cat_norm_qkv_image) ->
    (image, x_out fp32 [rows, 256], yb, rstd)"""
    lib = _lib.load()
    rows = xa.shape[0]
    for t in (xa, xb):
        if t.dtype != torch.float32 or t.dim() != 2 or tuple(t.shape) != (rows, 256) or t.stride(1) != 1 or rows != b * s:
            raise GaotError("cat_norm_qkv_image: fp32 [b * s, 256] inputs expected")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    bs = None if skip_bias is None else _req(skip_bias, torch.float32, "skip_bias")
    img = _ws(lib.gaot_attn_bf16_image_bytes(b, s, h, hkv), xa.device)
    xo = torch.empty(rows, 256, dtype=torch.float32, device=xa.device)
    yb = torch.empty(rows, 256, dtype=torch.bfloat16, device=xa.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=xa.device)
    table = rope_table(freqs, s) if freqs is not None else None
    with _timed("cat_norm_qkv_image"):
        check(lib.gaot_cat_norm_qkv_image(_ptr(xa), xa.stride(0), _ptr(xb), xb.stride(0), _ptr(skip_packed), _ptr(bs), _ptr(xo), _ptr(nw), float(eps),
                                          _ptr(packed), _ptr(img), _ptr(yb), _ptr(rstd), rows, s, h, hkv, _ptr(table), _qscale(scale), _stream()),
              "gaot_cat_norm_qkv_image")
    return img, xo, yb, rstd


def norm_qkv_image(x: Tensor, norm_weight: Tensor, eps: float, packed: Tensor, b: int, s: int, h: int, hkv: int, freqs: Optional[Tensor],
                   scale: float):
    """attn_norm + q | k | v projection written as the attention kernels' bf16 image in one launch (gaot_norm_qkv_image) ->
    (image, yb = bf16(norm(x)) [rows, 256], rstd [rows])"""
    lib = _lib.load()
    rows = x.shape[0]
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != 256 or x.stride(1) != 1 or rows != b * s:
        raise GaotError("norm_qkv_image: fp32 [b * s, 256] input expected")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    img = _ws(lib.gaot_attn_bf16_image_bytes(b, s, h, hkv), x.device)
    yb = torch.empty(rows, 256, dtype=torch.bfloat16, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    table = rope_table(freqs, s) if freqs is not None else None
    with _timed("norm_qkv_image"):
        check(lib.gaot_norm_qkv_image(_ptr(x), x.stride(0), _ptr(nw), float(eps), _ptr(packed), _ptr(img), _ptr(yb), _ptr(rstd), rows, s, h, hkv,
                                      _ptr(table), _qscale(scale), _stream()), "gaot_norm_qkv_image")
    return img, yb, rstd


def _finish_parts(part: Tensor, n: int, parts: int, lanes: int, defer: bool) -> Tensor:
    """out[n] = the fixed-order sum of ``parts`` partial rows: left to flush_deferred (defer) or one gaot_reduce_multi launch now"""
    out = torch.empty(n, dtype=torch.float32, device=part.device)
    if defer:
        _defer(part, out, n, parts, lanes)
        return out
    arr = (_ReduceDesc * 1)()
    arr[0].part, arr[0].out, arr[0].n, arr[0].parts, arr[0].lanes = part.data_ptr(), out.data_ptr(), n, parts, lanes
    check(_lib.load().gaot_reduce_multi(arr, 1, _stream()), "gaot_reduce_multi")
    return out


def ffn_bwd_norm(yb: Tensor, dy: Tensor, packed: Tensor, f: int, h: Tensor, norm_weight: Tensor, rstd: Tensor, defer: bool = False):
    """ffn_bwd with ffn_norm's backward in its epilogue (include/gaot3d_hip.h: gaot_ffn_bwd_norm) -> (dh fp32 [rows, 256], dag, u, dyb,
    d(norm weight) [256])"""
    lib = _lib.load()
    rows = yb.shape[0]
    if yb.dtype != torch.bfloat16 or not yb.is_contiguous() or yb.shape[1] != 256:
        raise GaotError("ffn_bwd_norm: contiguous bf16 [rows, 256] input expected")
    if dy.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape) != (rows, 256):
        raise GaotError("ffn_bwd_norm: contiguous fp32 [rows, 256] gradient expected")
    if h.dtype != torch.float32 or tuple(h.shape) != (rows, 256) or h.stride(1) != 1:
        raise GaotError("ffn_bwd_norm: fp32 [rows, 256] norm input expected")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    dev = yb.device
    dag = torch.empty(rows, 2 * f, dtype=torch.bfloat16, device=dev)
    u = torch.empty(rows, f, dtype=torch.bfloat16, device=dev)
    dyb = torch.empty(rows, 256, dtype=torch.bfloat16, device=dev)
    dh = torch.empty(rows, 256, dtype=torch.float32, device=dev)
    parts = int(lib.gaot_norm_bwd_parts(rows))
    part = torch.empty(parts, 256, dtype=torch.float32, device=dev)
    with _timed("ffn_bwd_norm"):
        check(lib.gaot_ffn_bwd_norm(_ptr(yb), _ptr(dy), _ptr(packed), _ptr(h), h.stride(0), _ptr(nw), _ptr(rstd), _ptr(dag), _ptr(u), _ptr(dyb),
                                    _ptr(dh), _ptr(part), rows, int(f), _stream()), "gaot_ffn_bwd_norm")
    return dh, dag, u, dyb, _finish_parts(part, 256, parts, 32, defer)


def qkv_bwd_norm(dqkv: Tensor, packed: Tensor, x: Tensor, norm_weight: Tensor, rstd: Tensor, dres: Optional[Tensor] = None,
                 dtap: Optional[Tensor] = None, defer: bool = False):
    """d(norm x) = dqkv Wqkv with attn_norm's backward in its epilogue (include/gaot3d_hip.h: gaot_qkv_bwd_norm; ``packed`` from
    qkv_pack_multi(..., with_backward=True)) -> (dx fp32 [rows, 256], d(norm weight) [256])"""
    lib = _lib.load()
    rows, n = dqkv.shape
    if dqkv.dtype != torch.float32 or not dqkv.is_contiguous() or n % 256:
        raise GaotError("qkv_bwd_norm: contiguous fp32 [rows, N] gradient with N a multiple of 256 expected")
    if x.dtype != torch.float32 or tuple(x.shape) != (rows, 256) or x.stride(1) != 1:
        raise GaotError("qkv_bwd_norm: fp32 [rows, 256] norm input expected")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    dres = None if dres is None else _req(dres, torch.float32, "dres")
    dtap = None if dtap is None else _req(dtap, torch.float32, "dtap")
    dev = dqkv.device
    dx = torch.empty(rows, 256, dtype=torch.float32, device=dev)
    parts = int(lib.gaot_norm_bwd_parts(rows))
    part = torch.empty(parts, 256, dtype=torch.float32, device=dev)
    with _timed("qkv_bwd_norm"):
        check(lib.gaot_qkv_bwd_norm(_ptr(dqkv), n, _ptr(packed), _ptr(x), x.stride(0), _ptr(nw), _ptr(rstd), _ptr(dres), _ptr(dtap), _ptr(dx),
                                    _ptr(part), rows, _stream()), "gaot_qkv_bwd_norm")
    return dx, _finish_parts(part, 256, parts, 32, defer)


def qkv_bwd_norm_cat(dqkv: Tensor, packed: Tensor, x: Tensor, norm_weight: Tensor, rstd: Tensor, dres: Optional[Tensor], skip_packed: Tensor,
                     same: bool, defer: bool = False):
    """qkv_bwd_norm for a decoder block (include/gaot3d_hip.h: gaot_qkv_bwd_norm_cat): also the skip projection's two input gradients
    from the dx rows on chip -> (dx, dxa, dxb or None when ``same`` (dxa then holds the sum), d(norm weight))"""
    lib = _lib.load()
    rows, n = dqkv.shape
    if dqkv.dtype != torch.float32 or not dqkv.is_contiguous() or n % 256:
        raise GaotError("qkv_bwd_norm_cat: contiguous fp32 [rows, N] gradient with N a multiple of 256 expected")
    if x.dtype != torch.float32 or tuple(x.shape) != (rows, 256) or x.stride(1) != 1:
        raise GaotError("qkv_bwd_norm_cat: fp32 [rows, 256] norm input expected")
    if skip_packed.numel() != int(lib.gaot_skip_packed_bytes()):
        raise GaotError("qkv_bwd_norm_cat: skip_packed is not a skip_pack_multi image")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    dres = None if dres is None else _req(dres, torch.float32, "dres")
    dev = dqkv.device
    dx = torch.empty(rows, 256, dtype=torch.float32, device=dev)
    dxa = torch.empty(rows, 256, dtype=torch.float32, device=dev)
    dxb = None if same else torch.empty(rows, 256, dtype=torch.float32, device=dev)
    parts = int(lib.gaot_norm_bwd_parts(rows))
    part = torch.empty(parts, 256, dtype=torch.float32, device=dev)
    with _timed("qkv_bwd_norm_cat"):
        check(lib.gaot_qkv_bwd_norm_cat(_ptr(dqkv), n, _ptr(packed), _ptr(x), x.stride(0), _ptr(nw), _ptr(rstd), _ptr(dres), _ptr(skip_packed),
                                        _ptr(dx), _ptr(dxa), _ptr(dxb), int(bool(same)), _ptr(part), rows, _stream()), "gaot_qkv_bwd_norm_cat")
    return dx, dxa, dxb, _finish_parts(part, 256, parts, 32, defer)


def oproj_bwd_image(dh: Tensor, attn_out: Tensor, packed: Tensor, f: int, b: int, s: int, h: int, hkv: int):
    """d_o = dh Wo as the flash backward's operands (include/gaot3d_hip.h: gaot_oproj_bwd_image): -> (an attn_bwd_scratch buffer whose head
    holds the bf16 dO image, delta fp32 [b, h, s]) for attn_bwd_bf16(do_image=..., delta=...)"""
    lib = _lib.load()
    rows = dh.shape[0]
    if h != 8 or rows != b * s or any(t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (rows, 256) for t in (dh, attn_out)):
        raise GaotError("oproj_bwd_image: contiguous fp32 [b * s, 256] tensors and 8 heads of 32 expected")
    scratch = attn_bwd_scratch(b, s, h, hkv, dh.device)
    delta = torch.empty(b, h, s, dtype=torch.float32, device=dh.device)
    with _timed("oproj_bwd_image"):
        check(lib.gaot_oproj_bwd_image(_ptr(dh), _ptr(attn_out), _ptr(packed), int(f), _ptr(scratch), _ptr(delta), rows, int(s), _stream()),
              "gaot_oproj_bwd_image")
    return scratch, delta


def norm_ffn_fwd(h: Tensor, norm_weight: Tensor, eps: float, packed: Tensor, f: int):
    """RMSNorm + FFN + residual in one launch (include/gaot3d_hip.h: gaot_norm_ffn_fwd): h fp32 [rows, 256] ->
    (y fp32 [rows, 256] = n + ffn(n), yb = bf16(n) [rows, 256], rstd [rows]) with n = RMSNorm(h)"""
    lib = _lib.load()
    if h.dtype != torch.float32 or h.dim() != 2 or h.shape[1] != 256 or h.stride(1) != 1:
        raise GaotError("norm_ffn_fwd: fp32 [rows, 256] input expected")
    nw = _req(norm_weight, torch.float32, "norm_weight")
    rows = h.shape[0]
    y = torch.empty(rows, 256, dtype=torch.float32, device=h.device)
    yb = torch.empty(rows, 256, dtype=torch.bfloat16, device=h.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=h.device)
    with _timed("norm_ffn_fwd"):
        check(lib.gaot_norm_ffn_fwd(_ptr(h), h.stride(0), _ptr(nw), float(eps), _ptr(packed), _ptr(y), _ptr(yb), _ptr(rstd), rows, int(f),
                                    _stream()), "gaot_norm_ffn_fwd")
    return y, yb, rstd


def ffn_bwd_dag(xb: Tensor, dy: Tensor, packed: Tensor, f: int, want_dyb: bool = True):
    """the first half of the FFN backward for a forward that saved nothing (include/gaot3d_hip.h: gaot_ffn_bwd_dag): x [rows, 256] bf16,
    dy fp32 [rows, 256], ``packed`` from ffn_pack(..., with_backward=True) -> (dag bf16 [rows, 2F], u bf16 [rows, F], dyb bf16 or None)"""
    lib = _lib.load()
    if xb.dtype != torch.bfloat16 or not xb.is_contiguous() or xb.shape[1] != 256:
        raise GaotError("ffn_bwd_dag: contiguous bf16 [rows, 256] input expected")
    rows = xb.shape[0]
    if dy.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape) != (rows, 256):
        raise GaotError("ffn_bwd_dag: contiguous fp32 [rows, 256] gradient expected")
    dag = torch.empty(rows, 2 * f, dtype=torch.bfloat16, device=xb.device)
    u = torch.empty(rows, f, dtype=torch.bfloat16, device=xb.device)
    dyb = torch.empty(rows, 256, dtype=torch.bfloat16, device=xb.device) if want_dyb else None
    with _timed("ffn_bwd_dag"):
        check(lib.gaot_ffn_bwd_dag(_ptr(xb), _ptr(dy), _ptr(packed), _ptr(dag), _ptr(u), _ptr(dyb), rows, int(f), _stream()), "gaot_ffn_bwd_dag")
    return dag, u, dyb


def ffn_bwd(xb: Tensor, dy: Tensor, packed: Tensor, f: int, add_dy: bool, want_dyb: bool = True):
    """ffn_bwd_dag with the input gradient in the same launch (include/gaot3d_hip.h: gaot_ffn_bwd) ->
    (dx fp32 [rows, 256] = dag W13 (+ dy), dag, u, dyb)"""
    lib = _lib.load()
    if xb.dtype != torch.bfloat16 or not xb.is_contiguous() or xb.shape[1] != 256:
        raise GaotError("ffn_bwd: contiguous bf16 [rows, 256] input expected")
    rows = xb.shape[0]
    if dy.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape) != (rows, 256):
        raise GaotError("ffn_bwd: contiguous fp32 [rows, 256] gradient expected")
    dag = torch.empty(rows, 2 * f, dtype=torch.bfloat16, device=xb.device)
    u = torch.empty(rows, f, dtype=torch.bfloat16, device=xb.device)
    dyb = torch.empty(rows, 256, dtype=torch.bfloat16, device=xb.device) if want_dyb else None
    dx = torch.empty(rows, 256, dtype=torch.float32, device=xb.device)
    with _timed("ffn_bwd"):
        check(lib.gaot_ffn_bwd(_ptr(xb), _ptr(dy), _ptr(packed), _ptr(dag), _ptr(u), _ptr(dyb), _ptr(dx), int(add_dy), rows, int(f), _stream()),
              "gaot_ffn_bwd")
    return dx, dag, u, dyb


def ffn_w2_bwd_swiglu(dyb: Tensor, w2t: Tensor, ag: Tensor, f: int) -> Tensor:
    """dy [rows, 256] bf16, W2^T [F, 256] bf16, a | g bf16 [rows, 2F] -> d(a) | d(g) bf16 [rows, 2F]: the du = dy W2 product with the
    SwiGLU backward in its epilogue (include/gaot3d_hip.h: gaot_ffn_w2_bwd_swiglu)"""
    lib = _lib.load()
    if any(t.dtype != torch.bfloat16 or not t.is_contiguous() for t in (dyb, w2t, ag)):
        raise GaotError("ffn_w2_bwd_swiglu: contiguous bf16 operands expected")
    dag = torch.empty_like(ag)
    with _timed("ffn_w2_bwd_swiglu"):
        check(lib.gaot_ffn_w2_bwd_swiglu(_ptr(dyb), _ptr(w2t), _ptr(ag), _ptr(dag), dyb.shape[0], dyb.shape[1], w2t.shape[1], int(f),
                                         _stream()), "gaot_ffn_w2_bwd_swiglu")
    return dag


def swiglu_bwd_bf16(ag: Tensor, du: Tensor, f: int) -> Tensor:
    lib = _lib.load()
    dag = torch.empty_like(ag)
    check(lib.gaot_swiglu_bwd_bf16(_ptr(ag), _ptr(du), _ptr(dag), ag.shape[0], f, _stream()), "gaot_swiglu_bwd_bf16")
    return dag


def patchify(src: Tensor, b: int, d: int, h: int, w: int, p: int, c: int, to_tokens: bool) -> Tensor:
    lib = _lib.load()
    dst = torch.empty_like(src)
    check(lib.gaot_patchify(_ptr(src), _ptr(dst), b, d, h, w, p, c, int(to_tokens), _stream()), "gaot_patchify")
    return dst


def mse_fwd(pred: Tensor, target: Tensor) -> Tensor:
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    ws = _ws(lib.gaot_mse_workspace_bytes(), pred.device)
    check(lib.gaot_mse_fwd(_ptr(pred), _ptr(target), pred.numel(), _ptr(loss), _ptr(ws), ws.numel(), _stream()),
          "gaot_mse_fwd")
    return loss


def mse_bwd(pred: Tensor, target: Tensor, grad_loss: Tensor) -> Tensor:
    lib = _lib.load()
    dp = torch.empty_like(pred)
    check(lib.gaot_mse_bwd(_ptr(pred), _ptr(target), pred.numel(), _ptr(grad_loss), _ptr(dp), _stream()), "gaot_mse_bwd")
    return dp


def geoembed_stats_sharded_queries(source_pos: Tensor, query_pos: Tensor, g: BipartiteGraph, group, num_queries_total: int) -> Tensor:
    """statistical features when the QUERY rows are spread over the ranks of ``group`` (each rank holds all edges of its
    queries): local raw features (two sweeps per row: centroid, then centred second moments), one fp64 SUM all-reduce of
    the 18 column sums, z-score over all rows.  ``group=None``: no exchange (all rows are local)"""
    import torch.distributed as dist
    lib = _lib.load()
    source_pos = _req(source_pos, torch.float32, "source_pos")
    query_pos = _req(query_pos, torch.float32, "query_pos")
    if source_pos.shape[1] != 3:
        raise GaotError("geoembed statistical features: coord_dim must be 3 on the HIP path")
    q = g.num_dst
    feat = torch.empty(q, 9, dtype=torch.float32, device=query_pos.device)
    sums = torch.empty(18, dtype=torch.float64, device=query_pos.device)
    ws = _ws(lib.gaot_geoembed_stats_workspace_bytes(), query_pos.device)
    check(lib.gaot_geoembed_raw(_ptr(source_pos), _ptr(query_pos), _ptr(g.by_dst.rowptr), _ptr(g.by_dst.other), q, _ptr(feat),
                                _ptr(sums), _ptr(ws), ws.numel(), _stream()), "gaot_geoembed_raw")
    if group is not None:
        from . import comm
        comm.run(lambda: dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group), (sums,), "all_reduce")
    check(lib.gaot_geoembed_finalize(_ptr(feat), q, _ptr(sums), int(num_queries_total), _ptr(ws), ws.numel(), _stream()),
          "gaot_geoembed_finalize")
    return feat


def geoembed_moments(source_pos: Tensor, query_pos: Tensor, g: BipartiteGraph) -> Tensor:
    """fp64 [Q, 12] additive moments of every query row's neighbourhood (include/gaot3d_hip.h: gaot_geoembed_moments)"""
    lib = _lib.load()
    source_pos = _req(source_pos, torch.float32, "source_pos")
    query_pos = _req(query_pos, torch.float32, "query_pos")
    if source_pos.shape[1] != 3:
        raise GaotError("geoembed statistical features: coord_dim must be 3 on the HIP path")
    mom = torch.empty(g.num_dst, 12, dtype=torch.float64, device=query_pos.device)
    check(lib.gaot_geoembed_moments(_ptr(source_pos), _ptr(query_pos), _ptr(g.by_dst.rowptr), _ptr(g.by_dst.other), g.num_dst,
                                    _ptr(mom), _stream()), "gaot_geoembed_moments")
    return mom


def geoembed_from_moments(mom: Tensor) -> Tensor:
    lib = _lib.load()
    q = mom.shape[0]
    feat = torch.empty(q, 9, dtype=torch.float32, device=mom.device)
    ws = _ws(lib.gaot_geoembed_stats_workspace_bytes(), mom.device)
    check(lib.gaot_geoembed_from_moments(_ptr(mom), q, _ptr(feat), _ptr(ws), ws.numel(), _stream()), "gaot_geoembed_from_moments")
    return feat


def _ptr_array(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def scale_mix_fwd(xs, logits: Tensor):
    lib = _lib.load()
    n, c = xs[0].shape
    out = torch.empty_like(xs[0])
    w = torch.empty(n, len(xs), dtype=torch.float32, device=out.device)
    check(lib.gaot_scale_mix_fwd(_ptr_array(xs), len(xs), _ptr(logits), _ptr(out), _ptr(w), n, c, _stream()),
          "gaot_scale_mix_fwd")
    return out, w


def scale_mix_bwd(xs, w: Tensor, dout: Tensor):
    lib = _lib.load()
    n, c = xs[0].shape
    dxs = [torch.empty_like(x) for x in xs]
    dlog = torch.empty_like(w)
    check(lib.gaot_scale_mix_bwd(_ptr_array(xs), len(xs), _ptr(w), _ptr(dout), _ptr_array(dxs), _ptr(dlog), n, c,
                                 _stream()), "gaot_scale_mix_bwd")
    return dxs, dlog


def mlp2_forward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Optional[Tensor]) -> Tensor:
    """out = w2 gelu(w1 x + b1) + b2, x [rows, 32] (include/gaot3d_hip.h: gaot_mlp2_fwd)"""
    lib = _lib.load()
    rows, hid, oc = x.shape[0], w1.shape[0], w2.shape[0]
    out = torch.empty(rows, oc, dtype=torch.float32, device=x.device)
    check(lib.gaot_mlp2_fwd(_ptr(x), rows, x.shape[1], hid, oc, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(out), _stream()),
          "gaot_mlp2_fwd")
    return out


def mlp2_backward(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, dout: Tensor):
    lib = _lib.load()
    rows, hid, oc = x.shape[0], w1.shape[0], w2.shape[0]
    dx = torch.empty_like(x)
    dw1, db1, dw2 = torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2)
    ws = _ws(lib.gaot_mlp2_bwd_workspace_bytes(hid, oc), x.device)
    check(lib.gaot_mlp2_bwd(_ptr(x), rows, x.shape[1], hid, oc, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(dout), _ptr(dx), _ptr(dw1),
                            _ptr(db1), _ptr(dw2), _ptr(ws), ws.numel(), _stream()), "gaot_mlp2_bwd")
    return dx, dw1, db1, dw2


def launch_count_reset() -> None:
    _lib.load().gaot_launch_count(1)


def launch_count() -> int:
    """kernel launches issued by the library since the last reset (ATen launches -- fills, cats, index plumbing -- excluded)"""
    return int(_lib.load().gaot_launch_count(0))
