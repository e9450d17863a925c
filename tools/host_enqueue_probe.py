"""How far ahead of the GPU does the host run?  Times the host-side enqueue of one training step (no synchronisation inside) against
the step's GPU duration, on the headline workload (python tools/host_enqueue_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from radvlm_amd import lib
from radvlm_amd.config import GEOMETRIES
from radvlm_amd.engine import LlavaEngine

lib.load()
geo = GEOMETRIES["llava15_7b"]
eng = LlavaEngine(geo, device="cuda:0", init="fast", seed=0, packed="auto")
eng.init_optimizer()
batch = bench.synthetic_batch(geo, 32, seed=1234)
def step():
    t0 = time.perf_counter(); eng.forward(*batch); t1 = time.perf_counter(); eng.backward(); t2 = time.perf_counter()
    eng.optimizer_step(lr=1e-5, weight_decay=0.0, max_grad_norm=1.0); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2
for _ in range(3): step()
torch.cuda.synchronize()
import gc
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if mode == "nogc":
    gc.collect(); gc.disable()
print("mode", mode, "gc thresholds", gc.get_threshold(), flush=True)
for i in range(24):
    torch.cuda.synchronize(); t = time.perf_counter()
    f, b, o = step()
    th = time.perf_counter() - t
    torch.cuda.synchronize(); tg = time.perf_counter() - t
    st = torch.cuda.memory_stats()
    print(f"[{i:2d}] gc {gc.get_count()} allocs {st['num_device_alloc']} frees {st['num_device_free']} retries {st['num_alloc_retries']}  host enqueue {th*1e3:7.1f} ms (fwd {f*1e3:.1f} bwd {b*1e3:.1f} opt {o*1e3:.1f})   step incl. GPU {tg*1e3:7.1f} ms   cpus {os.cpu_count()}", flush=True)
