"""Diagnostic (ONLY=attention tools/build_variant.sh lib_S.so -DRV_ATTN_STAMPS): when and where every block of the forward attention kernel ran
(s_memtime at start / end, HW_ID, XCC_ID) -> how many blocks each CU held over the launch, and the gaps between a block's end and its successor's
start on the same CU."""
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
    nq = (S + 127) // 128
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]

    def run():
        assert l.rv_attn_fwd_nat(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), out.data_ptr(), out.stride(0), lse.data_ptr(),
                                 None, None, B, H, Hkv, S, s_pad, hd, 1, hd ** -0.5, z.data_ptr(), st) == 0
    for nblk in (nq * H * B, ((nq + 1) // 2) * H * B):        # the launcher picks single query blocks or pairs: try both record layouts
        buf = torch.zeros(nblk * (4 * 6 + 3), dtype=torch.int64, device="cuda")
        l.rv_debug_set_attn_stamp_buffer(None)
        run(); run()
        l.rv_debug_set_attn_stamp_buffer(buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        l.rv_debug_set_attn_stamp_buffer(None)
        w = buf.cpu().numpy()[nblk * 24:].reshape(nblk, 3)
        if (w[:, 0] > 0).sum() > 0.5 * nblk:
            break
    wall_us = e0.elapsed_time(e1) * 1e3
    w = w[w[:, 0] > 0]
    hw, xcc = w[:, 2] & 0xFFFFFFFF, (w[:, 2] >> 32) & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    cus = np.unique(key)
    print(f"B={B} H={H}:{Hkv} S={S}: {len(w)} blocks on {len(cus)} distinct (xcc, se, sh, cu); launch {wall_us:.0f} us; XCCs {sorted(np.unique(xcc).tolist())}")
    occ, gaps, per_cu, spans, lives = [], [], [], [], []
    for kk in cus:
        b_ = w[key == kk]
        b_ = b_[np.argsort(b_[:, 0])]
        t0, t1 = b_[:, 0].min(), b_[:, 1].max()
        occ.append((b_[:, 1] - b_[:, 0]).sum() / max(1, (t1 - t0)))
        per_cu.append(len(b_))
        spans.append(t1 - t0)
        lives.append((b_[:, 1] - b_[:, 0]).mean())
        # gap between a block's end and the start of the next block that starts after it on this CU
        starts = np.sort(b_[:, 0])
        for e in b_[:, 1]:
            i = np.searchsorted(starts, e, side="left")
            if i < len(starts):
                gaps.append(starts[i] - e)
    spans, gaps = np.array(spans, dtype=np.float64), np.array(gaps, dtype=np.float64)
    print(f"    blocks per CU: min {min(per_cu)} mean {np.mean(per_cu):.1f} max {max(per_cu)}; mean block life {np.mean(lives):.0f} ticks; resident blocks per CU over its own busy span: "
          f"mean {np.mean(occ):.2f} min {min(occ):.2f}")
    print(f"    per-CU busy span (first start -> last end) in ticks: min {spans.min():.0f} median {np.median(spans):.0f} max {spans.max():.0f}; "
          f"max span / launch time = {spans.max() / wall_us / 1e3:.2f} ticks per ns")
    print(f"    end -> next start on the same CU: median {np.median(gaps):.0f} p90 {np.percentile(gaps, 90):.0f} max {gaps.max():.0f} ticks")
