"""CPU tests: the C-ABI library loads and exports every symbol include/radvlm_hip.h declares (no compute without a
GPU), the product path fails loudly without it, and the data-parallel gradient sync works at world_size 2 (gloo)."""
import os
import re
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from radvlm_amd import lib
    so = os.path.join(ROOT, "radvlm_amd", "libradvlm_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["bash", os.path.join(ROOT, "radvlm_amd", "csrc", "build.sh")])
    l = lib.load()
    hdr = open(os.path.join(ROOT, "include", "radvlm_hip.h")).read()
    declared = set(re.findall(r"\b(rv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTED_SYMBOLS), declared ^ set(lib.EXPORTED_SYMBOLS)
    for s in declared:
        assert hasattr(l, s), s
    assert b"gfx950" in l.rv_version()


def test_no_cpu_fallback():
    """Ops refuse CPU tensors; a missing library raises instead of falling back."""
    from radvlm_amd import lib, ops
    with pytest.raises(AssertionError):
        ops.gemm_nt(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
    code = textwrap.dedent(f"""
        import sys; sys.path.insert(0, {ROOT!r})
        from radvlm_amd import lib
        lib._LIB_PATH = '/nonexistent/libradvlm_hip.so'
        try:
            lib.load()
        except lib.RadvlmHipError as e:
            print('RAISED'); sys.exit(0)
        sys.exit(1)
    """)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "RAISED" in out.stdout, out.stderr
    assert "oracle" not in open(os.path.join(ROOT, "radvlm_amd", "engine.py")).read()


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from radvlm_amd.ddp import FlatGradSync
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 1000
flat = (torch.arange(n, dtype=torch.float32) + 1) * (rank + 1)
sync = FlatGradSync(flat, None, min_bucket_elems=300)
# backward order: back-to-front contiguous buckets, then a disjoint one
for s, e in [(900, 1000), (700, 900), (650, 700), (100, 650), (0, 50), (50, 100)]:
    sync.bucket_done(s, e)
launched = sync.finish()
want = (torch.arange(n, dtype=torch.float32) + 1) * sum(r + 1 for r in range(world))
assert torch.equal(flat, want), (flat[:5], want[:5])
cov = sorted(launched)
assert cov[0][0] == 0 and cov[-1][1] == n and all(a[1] == b[0] for a, b in zip(cov, cov[1:])), cov
assert len(launched) < 6  # adjacent ready ranges were coalesced into larger collectives
# mean-of-means semantics: per-rank grads pre-scaled by 1/world sum to the average
g = torch.full((8,), float(rank + 1)) / world
s2 = FlatGradSync(g, None)
s2.bucket_done(0, 8); s2.finish()
assert torch.allclose(g, torch.full((8,), sum(r + 1 for r in range(world)) / world))
# sampler sharding is disjoint across ranks
from radvlm_amd.llava.train.llava_trainer import get_length_grouped_indices, shard_for_rank
order = get_length_grouped_indices(list(range(1, 65)), 4, world, generator=torch.Generator().manual_seed(0))
mine = torch.tensor(shard_for_rank(order, 4, world, rank))
allr = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(allr, mine)
flat_all = torch.cat(allr).tolist()
assert len(set(flat_all)) == len(flat_all) == 64
# replica consistency (what DDP's wrap-time broadcast + nothing else gives the reference): identical buffers pass, one differing element
# on one rank is caught on EVERY rank, a broadcast from rank 0 repairs it
from radvlm_amd.ddp import assert_replicas_equal, broadcast_from_rank0
w = torch.arange(5000, dtype=torch.float32).to(torch.bfloat16)
st = torch.linspace(0, 1, 777)
assert assert_replicas_equal([("parameters", w), ("absent", None), ("exp_avg", st)], None, "unit") == ["parameters", "exp_avg"]
st2 = st.clone()
if rank == 1:
    st2[500] += 1e-3
try:
    assert_replicas_equal([("parameters", w), ("exp_avg", st2)], None, "after a faulty step")
    raise SystemExit("divergence not detected")
except RuntimeError as e:
    assert "exp_avg" in str(e) and "parameters" not in str(e).split("differ")[0].split(":")[-1], str(e)
swapped = w.clone()
if rank == 1:
    swapped[[10, 11]] = swapped[[11, 10]]         # same multiset of values: only the index-weighted sum sees it
try:
    assert_replicas_equal([("parameters", swapped)], None, "swap")
    raise SystemExit("swap not detected")
except RuntimeError:
    pass
broadcast_from_rank0([("exp_avg", st2), ("parameters", swapped)], None)
assert_replicas_equal([("parameters", swapped), ("exp_avg", st2)], None, "after broadcast")
dist.destroy_process_group()
print("OK", rank)
"""


def test_flat_grad_sync_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    import socket
    with socket.socket() as sock:        # a free rendezvous port (a fixed one collides with a concurrent or lingering run)
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=300)
        assert p.returncode == 0 and "OK" in out, err[-2000:]
