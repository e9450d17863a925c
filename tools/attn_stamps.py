"""Diagnostic (tools/build_variant.sh lib_S.so -DRV_ATTN_STAMPS): where the forward attention kernel's waves spend their cycles --
parked at the tile's wait + barrier, in the K reads + score MFMAs, in the softmax, in the P V MFMAs -- per wave, summed over its tiles
(s_memtime stamps; the instrumentation itself costs ~10 % of the wave's cycles)."""
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
    nblk = ((S + 127) // 128) * H * B
    buf = torch.zeros(nblk * 4 * 6, dtype=torch.int64, device="cuda")
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]

    def run():
        assert l.rv_attn_fwd_nat(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0), lse.data_ptr(),
                                 None, None, B, H, Hkv, S, s_pad, hd, 1, hd ** -0.5, z.data_ptr(), st) == 0
    l.rv_debug_set_attn_stamp_buffer(None)
    for _ in range(3):
        run()
    l.rv_debug_set_attn_stamp_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 1e3
    l.rv_debug_set_attn_stamp_buffer(None)
    raw = buf.cpu().numpy().reshape(nblk, 4, 6)
    dma = (raw[:, :, 4] >> 20).astype(np.float64)
    raw[:, :, 4] &= (1 << 20) - 1
    a = raw.astype(np.float64)
    dma = dma[a[:, :, 4].sum(1) > 0]
    a = a[a[:, :, 4].sum(1) > 0]                       # blocks that ran tiles
    tot = a[:, :, 5].sum()
    tiles = a[:, :, 4].sum()
    names = ["wait + barrier", "K reads + score MFMAs", "softmax", "P V MFMAs"]
    print(f"B={B} H={H}:{Hkv} S={S}: {tiles / a.shape[0] / 4:.1f} tiles per wave, {tot / tiles:.0f} cycles per wave-tile (incl. prologue / epilogue "
          f"{100 * (1 - a[:, :, :4].sum() / tot):.1f} % of the wave's life)")
    life = a[:, :, 5].max(1)                          # a block lives as long as its slowest wave
    slots = 2 * 256
    print(f"    blocks that ran {a.shape[0]}, mean block life {life.mean():.0f} cycles; sum of block lives / {slots} resident slots = {life.sum() / slots:.0f} cycles "
          f"= {life.sum() / slots / 2.1e3:.0f} us at 2.1 GHz; the launch took {wall_us:.0f} us (stamped build)")
    for i, n in enumerate(names):
        print(f"    {n:24s} {a[:, :, i].sum() / tiles:7.0f} cycles per wave-tile  ({100 * a[:, :, i].sum() / tot:5.1f} %)")
    print(f"    of the softmax phase: issuing the next tile's 8 LDS-DMA pieces {dma.sum() / tiles:7.0f} cycles per wave-tile ({100 * dma.sum() / tot:5.1f} %)")
