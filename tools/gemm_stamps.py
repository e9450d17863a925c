"""Diagnostic (tools/build_variant.sh lib_S.so -DRV_STAMPS): where a 256x256 GEMM tile spends its time.
Per block: entry -> first barrier (prologue: 2 K-tiles issued, tile 0 landed), K loop, epilogue; plus the idle gap of a CU
between consecutive blocks, by matching every block to its predecessor on the same CU slot (greedy on start times)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib as L
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
l = ctypes.CDLL(os.path.join(root, "radvlm_amd", "lib_S.so"))
l.rv_gemm_bf16.argtypes = L._SIGS["rv_gemm_bf16"]; l.rv_gemm_bf16.restype = ctypes.c_int
l.rv_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
l.rv_gemm_select_kernel(2)
z = torch.zeros(64, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
# box calibration: device copy bandwidth (GB/s moved = 2 x bytes) -- boxes of the pool differ
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); y = torch.empty_like(x)
y.copy_(x); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y.copy_(x)
e1.record(); torch.cuda.synchronize()
print(f"box: d2d copy {2 * 10 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9:.0f} GB/s", flush=True)
del x, y
T = 22528
for name, m, n, k, ta, tb in [("qkv_fwd NT", T, 12288, 4096, 0, 0), ("o_fwd NT", T, 4096, 4096, 0, 0), ("down_fwd NT", T, 4096, 11008, 0, 0),
                              ("gu_wgrad TT", 22016, 4096, T, 1, 1)]:
    a = torch.randn((k, m) if ta else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    nblk = ((m + 255) // 256) * ((n + 255) // 256)
    buf = torch.zeros(nblk * 4 + nblk * 8 * 3, dtype=torch.int64, device="cuda")
    def run():
        assert l.rv_gemm_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), n, None, None, 0, m, n, k, ta, tb, 1.0, 0, 0, 0, z.data_ptr(), st) == 0
    l.rv_debug_set_stamp_buffer(None)
    for _ in range(3):
        run()
    l.rv_debug_set_stamp_buffer(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    l.rv_debug_set_stamp_buffer(None)
    s = buf[:nblk * 4].view(nblk, 4).cpu().numpy().astype(np.float64) * 0.01          # us
    park = buf[nblk * 4:].view(nblk, 8, 3).cpu().numpy().astype(np.float64)   # shader cycles per wave over the K loop
    t0 = s[:, 0].min()
    pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    first = np.argsort(s[:, 0])[:256]
    print(f"{name} {(m, n, k)}: {nblk} tiles, kernel {s[:, 3].max() - t0:.1f} us; per tile: prologue {pro.mean():.2f} (first round {pro[first].mean():.2f}, "
          f"later {np.delete(pro, first).mean():.2f}) loop {loop.mean():.2f} epilogue {epi.mean():.2f} us; total {(s[:, 3] - s[:, 0]).mean():.2f}", flush=True)
    nt = (k + 63) // 64
    pk = park.mean(axis=(0, 1)) / nt
    print(f"    parked cycles per K-tile per wave: mid barrier(+own LDS reads) {pk[0]:.0f}, end-of-tile vmcnt {pk[1]:.0f}, end barrier {pk[2]:.0f}; "
          f"loop = {loop.mean() / nt * 1e3:.0f} ns per K-tile", flush=True)
    # CU-slot idle gaps: sort block ends and starts; the i-th start after the first 256 follows the i-th end
    ends = np.sort(s[:, 3]); starts = np.sort(s[:, 0])[256:]
    if len(starts):
        gap = starts - ends[:len(starts)]
        print(f"    gap between a block's end and the next block's start on the freed slot: mean {gap.mean():.2f} us, p90 {np.percentile(gap, 90):.2f}", flush=True)
