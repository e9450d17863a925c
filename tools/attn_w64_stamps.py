"""Diagnostic (tools/build_variant.sh lib_S.so -DRV_W64_STAMPS): where the one-wave-per-SIMD forward kernel's waves spend their cycles, per
wave-tile: parked at the tile's counted wait + barrier, issuing the K staging pieces, in each of the four MFMA slots, idle behind their
diagonal (s_memtime stamps; each stamp drains the fragment reads in flight: read shares, not lengths)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib as L
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
l = ctypes.CDLL(os.path.join(root, "radvlm_amd", "lib_S.so"))
l.rv_attn_fwd_nat.argtypes = L._SIGS["rv_attn_fwd_nat"]; l.rv_attn_fwd_nat.restype = ctypes.c_int
l.rv_debug_set_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
z = torch.zeros(64, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
hd = 128
for B, H, Hkv, S in ((32, 32, 32, 704), (8, 32, 32, 3056), (2, 28, 4, 7499)):
    d, kvd = H * hd, Hkv * hd
    s_pad = (S + 63) // 64 * 64
    qkv = torch.randn(B * S, d + 2 * kvd, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(B * S, d, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, H, s_pad, dtype=torch.float32, device="cuda")
    nblk = ((S + 255) // 256) * H * B
    buf = torch.zeros(nblk * 4 * 8, dtype=torch.int64, device="cuda")
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]

    def run():
        assert l.rv_attn_fwd_nat(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0), lse.data_ptr(),
                                 None, None, B, H, Hkv, S, s_pad, hd, 1, hd ** -0.5, z.data_ptr(), st) == 0
    l.rv_debug_set_attn_stamp_buffer(None)
    for _ in range(3):
        run()
    l.rv_debug_set_attn_stamp_buffer(buf.data_ptr())
    run()
    torch.cuda.synchronize()
    l.rv_debug_set_attn_stamp_buffer(None)
    a = buf.cpu().numpy().reshape(nblk, 4, 8).astype(np.float64)
    tot, tiles = a[:, :, 6].sum(), a[:, :, 7].sum()
    names = ["wait + barrier (all iterations)", "slot 1: S(0-31) | exp of the previous half", "slot 2: P V (previous half) | row maxima", "slot 3: S(32-63) | exp",
             "slot 4: P V | row maxima", "K staging pieces (4 per tile)"]
    print(f"B={B} H={H}:{Hkv} S={S}: {tiles / nblk / 4:.1f} computed tiles per wave, {tot / tiles:.0f} cycles of wave life per computed tile")
    for i, n in enumerate(names):
        print(f"    {n:46s} {a[:, :, i].sum() / tiles:7.0f} cycles per computed tile  ({100 * a[:, :, i].sum() / tot:5.1f} %)")
    rest = tot - a[:, :, :6].sum()
    print(f"    {'prologue, epilogue, iterations past the diagonal':46s} {rest / tiles:7.0f} cycles per computed tile  ({100 * rest / tot:5.1f} %)")
