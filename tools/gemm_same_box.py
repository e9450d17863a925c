"""One-box cross-check of the GEMM timing paths (boxes of the pool differ): ops.gemm (tail split), direct rv_gemm_bf16 of the
product library, the stamp build with stamps off / on, and the vendor library, all on the qkv_fwd shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib as L, ops
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
m, n, k = 22528, 12288, 4096
a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
b = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
z = torch.zeros(64, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
prod = L.load()
S = ctypes.CDLL(os.path.join(root, "radvlm_amd", "lib_S.so"))
S.rv_gemm_bf16.argtypes = L._SIGS["rv_gemm_bf16"]; S.rv_gemm_bf16.restype = ctypes.c_int
S.rv_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
S.rv_gemm_select_kernel(2)
buf = torch.zeros(4224 * 4, dtype=torch.int64, device="cuda")


def direct(l):
    assert l.rv_gemm_bf16(a.data_ptr(), k, b.data_ptr(), k, c.data_ptr(), n, None, None, 0, m, n, k, 0, 0, 1.0, 0, 0, 0, z.data_ptr(), st) == 0


def t(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cases = [("ops.gemm (product, tail split)", lambda: ops.gemm(a, b, out=c)), ("rv_gemm_bf16 direct (product)", lambda: direct(prod)),
         ("stamp build, stamps off", lambda: direct(S)), ("vendor library", lambda: torch.matmul(a, b.t(), out=c))]
for rnd in range(3):
    for name, fn in cases:
        if name.startswith("rv_gemm"):
            prod.rv_gemm_select_kernel(2)
        ms = t(fn)
        print(f"round {rnd}: {name:34s} {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)
    S.rv_debug_set_stamp_buffer(buf.data_ptr())
    ms = t(lambda: direct(S))
    S.rv_debug_set_stamp_buffer(None)
    print(f"round {rnd}: {'stamp build, stamps on':34s} {ms * 1e3:8.1f} us  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)
prod.rv_gemm_select_kernel(0)
