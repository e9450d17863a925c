import sys, os
sys.path.insert(0, "/root/repo")
import torch
from radvlm_amd import ops
torch.manual_seed(0)
M, V, d, nlab = 857, 152064, 3584, 100
dev = "cuda"
lab = torch.sort(torch.randperm(M)[:nlab]).values.to(dev)
hN = torch.randn(M, d, device=dev).bfloat16()
dl = torch.zeros(M, V, device=dev, dtype=torch.bfloat16)
dl[lab] = (torch.randn(nlab, V, device=dev) * 1e-3).bfloat16()
W = (torch.randn(V, d, device=dev) * 0.02).bfloat16()
nsel = (nlab + 63) // 64 * 64
hN_s = torch.zeros(nsel, d, device=dev, dtype=torch.bfloat16); hN_s[:nlab] = hN[lab]
dl_s = torch.zeros(nsel, V, device=dev, dtype=torch.bfloat16); dl_s[:nlab] = dl[lab]
gw_all = ops.gemm(dl, hN, ta=True, tb=True)
gw_sel = ops.gemm(dl_s, hN_s, ta=True, tb=True)
ref = dl[lab].float().T @ hN[lab].float()
def rel(a, b): return float((a.float() - b).norm() / b.norm())
print("wgrad: ||all|| %.6f ||sel|| %.6f  ||ref|| %.6f  rel(all,ref) %.3e rel(sel,ref) %.3e  differing elements %.3e" % (float(gw_all.float().norm()), float(gw_sel.float().norm()), float(ref.norm()), rel(gw_all, ref), rel(gw_sel, ref), float((gw_all != gw_sel).float().mean())))
dx_all = ops.gemm(dl, W, tb=True)
dx_sel = ops.gemm(dl_s, W, tb=True)
refx = dl[lab].float() @ W.float()
print("dgrad: rel(all,ref) %.3e rel(sel,ref) %.3e  rows equal: %s, differing elements %.3e" % (rel(dx_all[lab], refx), rel(dx_sel[:nlab], refx), torch.equal(dx_all[lab], dx_sel[:nlab]), float((dx_all[lab] != dx_sel[:nlab]).float().mean())))
lg_all = ops.gemm_nt(hN, W)
lg_sel = ops.gemm_nt(hN_s, W)
print("logits rows equal:", torch.equal(lg_all[lab], lg_sel[:nlab]), float((lg_all[lab] != lg_sel[:nlab]).float().mean()))
