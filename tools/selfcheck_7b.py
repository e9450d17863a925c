"""Self-consistency at the full 7B geometry (no CPU oracle fits): one forward + backward + optimizer step with the fused GEMM epilogues
(RV_FUSED=1) against the unfused kernel sequences (RV_FUSED=0): loss, gradient norm and a checksum of the updated parameters must be
bit-identical (the fused epilogues keep the unfused rounding points).  python tools/selfcheck_7b.py  (spawns one child per setting)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, bench
    from radvlm_amd import lib
    from radvlm_amd.config import GEOMETRIES
    from radvlm_amd.engine import LlavaEngine
    lib.load()
    geo = GEOMETRIES["llava15_7b"]
    eng = LlavaEngine(geo, device="cuda:0", init="fast", seed=0, packed="auto")
    eng.init_optimizer()
    out = []
    for i in range(2):
        loss = eng.forward(*bench.synthetic_batch(geo, 4, seed=77 + i)); eng.backward()
        eng.optimizer_step(lr=1e-4, weight_decay=0.0, max_grad_norm=1.0)
        flat = eng.lm.flat
        out.append({"loss": float(loss), "grad_norm": float(eng.last_grad_norm), "param_sum": float(flat.double().sum()),
                    "param_abs": float(flat.double().abs().sum()), "tail_abs": float(flat[-(1 << 20):].double().abs().sum())})
    print("RESULT " + json.dumps(out))
    sys.exit(0)
res = {}
for fused in ("1", "0"):
    env = dict(os.environ, RV_FUSED=fused)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, p.stdout[-2000:] + p.stderr[-2000:]
    res[fused] = json.loads(line[0][7:])
    print("RV_FUSED=" + fused, res[fused], flush=True)
assert res["1"] == res["0"], "fused and unfused paths differ"
print("bit-identical: loss, grad norm, parameter checksums after two optimizer steps")
