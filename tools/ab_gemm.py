"""Same-box A/B of two builds of the GEMM (radvlm_amd/lib_A.so vs lib_B.so), interleaved rounds, random data."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib as L
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {}
for k in ("A", "B"):
    l = ctypes.CDLL(os.path.join(root, "radvlm_amd", f"lib_{k}.so"))
    l.rv_gemm_bf16.argtypes = L._SIGS["rv_gemm_bf16"]; l.rv_gemm_bf16.restype = ctypes.c_int
    l.rv_gemm_select_kernel(2)
    libs[k] = l
z = torch.zeros(64, dtype=torch.uint8, device="cuda")
T = 22528
CASES = [("sq4096", 4096, 4096, 4096, 0, 0), ("sq8192", 8192, 8192, 8192, 0, 0), ("qkv_fwd", T, 12288, 4096, 0, 0), ("o_fwd", T, 4096, 4096, 0, 0), ("gu_fwd", T, 22016, 4096, 0, 0), ("down_fwd", T, 4096, 11008, 0, 0),
         ("head_fwd", T, 32000, 4096, 0, 0), ("dh2 NN", T, 4096, 22016, 0, 1), ("dact NN", T, 11008, 4096, 0, 1), ("gu_wgrad TT", 22016, 4096, T, 1, 1),
         ("down_wgrad TT", 4096, 11008, T, 1, 1), ("o_wgrad TT", 4096, 4096, T, 1, 1), ("qkv_wgrad TT", 12288, 4096, T, 1, 1), ("vit_fc1", 18464, 4096, 1024, 0, 0)]
tot = {"A": 0.0, "B": 0.0}
st = torch.cuda.current_stream().cuda_stream
for name, m, n, k, ta, tb in CASES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    def run(l):
        rc = l.rv_gemm_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), n, None, None, 0, m, n, k, ta, tb, 1.0, 0, 0, 0, z.data_ptr(), st)
        assert rc == 0
    # the two builds must agree bit for bit (same accumulation order), also under repetition (race screen)
    run(libs["A"]); ref = c.clone()
    for _ in range(6):
        c.zero_(); run(libs["B"])
        assert os.environ.get("RV_AB_NOCHECK") or torch.equal(c, ref), f"{name}: B differs from A"
    best = {"A": 1e9, "B": 1e9}
    for rnd in range(4):
        for key in ("A", "B"):
            run(libs[key]); 
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run(libs[key])
            e1.record(); torch.cuda.synchronize()
            best[key] = min(best[key], e0.elapsed_time(e1) / 5)
    fl = 2.0 * m * n * k
    tot["A"] += best["A"]; tot["B"] += best["B"]
    print(f"{name:14s} A {fl/best['A']/1e9:7.1f} TF/s   B {fl/best['B']/1e9:7.1f} TF/s   B/A {best['A']/best['B']:.3f}", flush=True)
print(f"sum of times: A {tot['A']:.2f} ms  B {tot['B']:.2f} ms  B/A speed {tot['A']/tot['B']:.3f}")
