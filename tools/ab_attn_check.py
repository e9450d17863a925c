"""Bit-identity check of two library builds on the forward attention (lib_A.so vs lib_B.so), causal + ragged."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib as L
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = torch.zeros(64, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
outs = {}
for key in ("A", "B"):
    l = ctypes.CDLL(os.path.join(root, "radvlm_amd", f"lib_{key}.so"))
    l.rv_attn_fwd_nat.argtypes = L._SIGS["rv_attn_fwd_nat"]; l.rv_attn_fwd_nat.restype = ctypes.c_int
    res = []
    for B, H, Hkv, S, lens in ((3, 4, 4, 704, [704, 500, 77]), (2, 8, 2, 1000, [1000, 999]), (1, 2, 2, 3056, None)):
        g = torch.Generator(device="cuda").manual_seed(S)
        hd, d, kvd = 128, H * 128, Hkv * 128
        s_pad = (S + 63) // 64 * 64
        qkv = torch.randn(B * S, d + 2 * kvd, device="cuda", dtype=torch.bfloat16, generator=g)
        out = torch.zeros(B * S, d, device="cuda", dtype=torch.bfloat16)
        lse = torch.zeros(B, H, s_pad, dtype=torch.float32, device="cuda")
        lt = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens else None
        for causal in (1, 0):
            rc = l.rv_attn_fwd_nat(qkv.data_ptr(), qkv.stride(0), qkv[:, d:].data_ptr(), qkv.stride(0), qkv[:, d + kvd:].data_ptr(), qkv.stride(0), out.data_ptr(), d,
                                   lse.data_ptr(), lt.data_ptr() if lt is not None else None, None, B, H, Hkv, S, s_pad, hd, causal, hd ** -0.5, z.data_ptr(), st)
            assert rc == 0
            torch.cuda.synchronize()
            res.append((out.clone(), lse.clone()))
    outs[key] = res
for (oa, la), (ob, lb) in zip(outs["A"], outs["B"]):
    assert torch.equal(oa, ob) and torch.equal(la, lb)
print("A and B agree bit for bit on", len(outs["A"]), "cases")
