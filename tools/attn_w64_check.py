"""One-wave-per-SIMD attention kernels (attention_w64.hip) against round 3's kernels and an fp32 reference, same process:
correctness on small shapes (all mask forms), then interleaved timing on the three bench shapes.
    python tools/attn_w64_check.py [--no-time]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radvlm_amd import lib, ops

hd = 128
L = lib.load()


def ref_attn(q, k, v, B, S, H, Hkv, causal, lens):
    qf = q.float().view(B, S, H, hd).transpose(1, 2)
    kf = k.float().view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, 1)
    vf = v.float().view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, 1)
    s = qf @ kf.transpose(2, 3) / math.sqrt(hd)
    mask = torch.zeros(B, 1, S, S, dtype=torch.bool, device=q.device)
    if causal:
        mask |= torch.triu(torch.ones(S, S, dtype=torch.bool, device=q.device), 1)
    if lens is not None:
        mask = mask | (torch.arange(S, device=q.device)[None, None, None, :] >= lens[:, None, None, None])
    s = s.masked_fill(mask, -math.inf)
    p = torch.softmax(s, -1)
    o = (p @ vf).transpose(1, 2).reshape(B * S, H * hd)
    lse = torch.logsumexp(s, -1)
    return o, lse


def check(B, S, H, Hkv, causal, lens, spike=False):
    torch.manual_seed(S * 7 + H)
    d, kvd = H * hd, Hkv * hd
    s_pad = (S + 63) // 64 * 64
    qkv = torch.randn(B * S, d + 2 * kvd, device="cuda", dtype=torch.bfloat16)
    q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]
    if spike:      # force late rescales: one key row far larger than the rest, late in the sequence
        k[S - 3] *= 12
        k[S // 2 + 5] *= 8
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda") if lens is not None else None
    res = {}
    for fam in (2, 1):
        L.rv_attn_select_kernel(fam)
        o, lse = ops.attn_fwd(q, k, None, B, S, H, hd, s_pad, causal, lens=lens_t, kv_heads=Hkv, v=v)
        torch.cuda.synchronize()
        res[fam] = (o.float().clone(), lse.clone())
    L.rv_attn_select_kernel(0)
    ro, rl = ref_attn(q, k, v, B, S, H, Hkv, causal, lens_t)
    valid = torch.ones(B, S, dtype=torch.bool, device="cuda")
    if lens is not None:
        valid = torch.arange(S, device="cuda")[None, :] < lens_t[:, None]
    vm = valid.reshape(-1)
    out = []
    for fam in (2, 1):
        o, lse = res[fam]
        eo = ((o - ro)[vm].abs().max() / ro[vm].abs().max()).item()
        el = (lse[:, :, :S] - rl)[valid[:, None, :].expand(B, H, S)].abs().max().item()
        out.append((eo, el))
    d01 = (res[2][0] - res[1][0])[vm].abs().max().item()
    nan = torch.isnan(res[2][0][vm]).any().item() or torch.isnan(res[1][0][vm]).any().item()
    ok = max(out[0][0], out[1][0]) < 2 ** -7 and max(out[0][1], out[1][1]) < 1e-3 and not nan
    print(f"B={B} S={S} H={H}:{Hkv} causal={causal} lens={lens} spike={spike}: w64 err {out[0][0]:.2e} lse {out[0][1]:.2e} | old err {out[1][0]:.2e} lse {out[1][1]:.2e} | "
          f"w64-old max {d01:.2e} {'OK' if ok else 'FAIL'}", flush=True)
    return ok


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    allok = True
    for (B, S, H, Hkv, causal, lens, spike) in [] if "--time-only" in sys.argv else [
        (1, 64, 1, 1, True, None, False), (1, 256, 2, 2, True, None, False), (2, 300, 2, 2, True, None, False), (2, 704, 4, 4, True, None, False),
        (2, 704, 4, 4, True, [704, 391], False), (1, 1000, 4, 2, True, None, False), (2, 577, 2, 2, False, None, False), (2, 577, 2, 2, False, [577, 130], False),
        (1, 1537, 4, 1, True, None, True), (1, 70, 2, 2, True, None, False), (3, 129, 2, 2, True, [129, 1, 64], False)]:
        allok &= check(B, S, H, Hkv, causal, lens, spike)
    print("ALL OK" if allok else "SOME FAILED", flush=True)
    if "--no-time" in sys.argv:
        return 0 if allok else 1
    for B, H, Hkv, S in ((32, 32, 32, 704), (8, 32, 32, 3056), (2, 28, 4, 7499)):
        d, kvd = H * hd, Hkv * hd
        s_pad = (S + 63) // 64 * 64
        qkv = torch.randn(B * S, d + 2 * kvd, device="cuda", dtype=torch.bfloat16)
        q, k, v = qkv[:, :d], qkv[:, d:d + kvd], qkv[:, d + kvd:]
        out, lse = ops.attn_fwd(q, k, None, B, S, H, hd, s_pad, True, kv_heads=Hkv, v=v)
        fl = 4.0 * B * H * S * S * hd / 2
        ts = {2: [], 1: []}
        for r in range(3):
            for fam in (2, 1):
                L.rv_attn_select_kernel(fam)
                ts[fam].append(timeit(lambda: ops.attn_fwd(q, k, None, B, S, H, hd, s_pad, True, kv_heads=Hkv, out=out, lse=lse, v=v)))
        L.rv_attn_select_kernel(0)
        for fam, name in ((2, "w64"), (1, "two-wave")):
            t = min(ts[fam])
            print(f"fwd B={B} H={H}:{Hkv} S={S} {name}: {t*1e3:.0f} us ({fl/t/1e9:.0f} TF/s = {fl/t/1e9/25:.1f} % of peak)  runs {[round(x*1e3) for x in ts[fam]]}", flush=True)
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
