"""round 6 lab (measured, NOT shipped): the Transformer's weight-gradient products from fragment-ordered operands (tools/lab/dw_frag.hip)
against the generic split-K GEMM, six rotating operand sets (one set sits in the Infinity Cache and flatters every variant).  The kernel
lives in its own shared object (tools/lab/bin/libdw_frag_lab.so, built here on first use against the product library for gaot_set_error);
`python tools/lab/dw_frag_lab.py test` runs its parity checks (tools/lab/dw_frag_test.py).  Results: profiles/r6_zg ... r6_zj."""
import ctypes as C
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import gaot_3d_amd
from gaot_3d_amd import ops
from gaot_3d_amd.ops import GaotError, Tensor, _finish_parts, _ptr, _stream, check


def _lab_lib():
    so = os.path.join(ROOT, "tools", "lab", "bin", "libdw_frag_lab.so")
    src = os.path.join(ROOT, "tools", "lab", "dw_frag.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        libdir = os.path.join(ROOT, "gaot_3d_amd", "lib")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-shared", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast",
                               "-I", os.path.join(ROOT, "gaot_3d_amd", "csrc"), "-I", os.path.join(ROOT, "include"), src, "-o", so,
                               "-L", libdir, "-lgaot3d_hip", "-Wl,-rpath," + libdir])
    ops._lib.load()
    lib = C.CDLL(so, mode=C.RTLD_GLOBAL)
    i64, p, i = C.c_int64, C.c_void_p, C.c_int
    lib.gaot_timg_bytes.restype, lib.gaot_timg_bytes.argtypes = i64, [i64, i64]
    lib.gaot_timg_pack.restype, lib.gaot_timg_pack.argtypes = i, [p, i, i64, i64, i64, p, p]
    lib.gaot_dw_frag_splits.restype, lib.gaot_dw_frag_splits.argtypes = i, [i64, i64, i64]
    lib.gaot_dw_frag.restype, lib.gaot_dw_frag.argtypes = i, [p, p, i64, i64, i64, p, p]
    return lib


LAB = _lab_lib()


def timg_pack(x: Tensor) -> Tensor:
    """row-major fp32 / bf16 [rows, cols] (cols % 128 == 0) -> its T-image (include/gaot3d_hip.h: gaot_timg_pack), the operand form of dw_frag"""
    lib = LAB
    rows, cols = x.shape
    if x.dtype not in (torch.float32, torch.bfloat16) or x.stride(1) != 1 or cols % 128:
        raise GaotError("timg_pack: fp32 / bf16 [rows, cols] with unit column stride and cols a multiple of 128 expected")
    img = torch.empty(int(lib.gaot_timg_bytes(rows, cols)), dtype=torch.uint8, device=x.device)
    check(lib.gaot_timg_pack(_ptr(x), int(x.dtype == torch.bfloat16), x.stride(0), rows, cols, _ptr(img), _stream()), "gaot_timg_pack")
    return img


def dw_frag(a_image: Tensor, b_image: Tensor, rows: int, n1: int, n2: int, defer: bool = False) -> Tensor:
    """dW [n1, n2] = A^T B from the T-images of A [rows, n1] and B [rows, n2] (include/gaot3d_hip.h: gaot_dw_frag); ``defer``: the sum of
    the split partials waits for flush_deferred (see defer_ok)"""
    lib = LAB
    if a_image.numel() != int(lib.gaot_timg_bytes(rows, n1)) or b_image.numel() != int(lib.gaot_timg_bytes(rows, n2)):
        raise GaotError("dw_frag: operand images do not match (rows, n1, n2)")
    splits = int(lib.gaot_dw_frag_splits(rows, n1, n2))
    if splits <= 0:
        raise GaotError("dw_frag: n1 and n2 must be multiples of 128")
    part = torch.empty(splits, n1 * n2, dtype=torch.float32, device=a_image.device)
    check(lib.gaot_dw_frag(_ptr(a_image), _ptr(b_image), rows, n1, n2, _ptr(part), _stream()), "gaot_dw_frag")
    return _finish_parts(part, n1 * n2, splits, 4, defer).view(n1, n2)


gaot_3d_amd.set_precision("bf16")
dev = "cuda:0"
if len(sys.argv) > 1 and sys.argv[1] == "test":
    import pytest
    sys.exit(pytest.main([os.path.join(ROOT, "tools", "lab", "dw_frag_test.py"), "-q", "-s", "-p", "no:cacheprovider"]))
rows = 16384
NSET = int(os.environ.get('GAOT_LAB_NSET', '6'))


def timeit(fn, name, reps=60):
    """device time per call: the calls are captured into one hipGraph and replayed (a Python loop over ctypes calls issues one launch per
    ~10 us and would time the host for the short kernels)"""
    for i in range(6):
        fn(i % NSET)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(reps):
                fn(i % NSET)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / (3 * reps) * 1e3:.1f} us")


for n1, n2, name in ((2048, 256, "dW13"), (256, 1024, "dW2"), (768, 256, "dWqkv"), (256, 256, "dWo")):
    torch.manual_seed(0)
    A = [torch.randn(rows, n1, device=dev).bfloat16() for _ in range(NSET)]
    B = [torch.randn(rows, n2, device=dev).bfloat16() for _ in range(NSET)]
    Ai = [timg_pack(x) for x in A]
    Bi = [timg_pack(x) for x in B]
    timeit(lambda i: ops.gemm_dw(A[i], B[i], n1, n2, rows, n1, n2, 1), f"{name} [{n1} x {n2}] generic (bf16 operands, in-call reduction)")
    timeit(lambda i: dw_frag(Ai[i], Bi[i], rows, n1, n2), f"{name} [{n1} x {n2}] dw_frag (in-call reduction)")
    lib = LAB
    splits = int(lib.gaot_dw_frag_splits(rows, n1, n2))
    part = torch.empty(splits, n1 * n2, device=dev)
    timeit(lambda i: ops.check(lib.gaot_dw_frag(ops._ptr(Ai[i]), ops._ptr(Bi[i]), rows, n1, n2, ops._ptr(part), ops._stream()), "x"),
           f"{name} [{n1} x {n2}] dw_frag kernel alone ({splits} splits)")
