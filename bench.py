"""bench.py -- train image-instruction pairs/s, LLaVA-1.5-7B (CLIP-ViT-L/14-336 + Vicuna-7B geometry), bf16, MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = forward + backward + gradient all-reduce (N > 1, RCCL, overlapped with backward) + AdamW update of all
6.76 B trainable parameters, on a synthetic batch of SURVEY.md section 8d's config-2 sample (one 336x336 image,
129 ids with the image placeholder at position 35 -> S = 704), random-init weights.  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TF_PER_PAIR = {"llava15_7b": 28.75, "config1": 0.089, "toy": None}  # BASELINE.md section 3 (algorithmic, ViT frozen)
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def synthetic_batch_anyres(geo, b, seed):
    """BASELINE config 4: 672x672 image, pinpoints [[336,672],[672,336],[672,672],[1008,336],[336,1008]], spatial_unpad ->
    5 tiles, 2928 image tokens + 128 text tokens = S 3056."""
    ids, mask, labels, _ = synthetic_batch(geo, b, seed)
    g = torch.Generator().manual_seed(seed)
    img = geo["vision"]["image"]
    images = [torch.randn(5, 3, img, img, generator=g).to(torch.bfloat16) for _ in range(b)]
    return ids, mask, labels, images, [(672, 672)] * b


def synthetic_batch_radvlm(geo, b, seed):
    """SURVEY 8f.1 / finetune_radio_7b.sh recipe shape: one 1024x1024 radiograph per sample -> anyres best resolution
    1152x1152 = 3x3 tiles + the base tile (10 tiles of 384 px), spatial_unpad + anyres_max_9 (81x81 grid = exactly the 9-tile
    limit, so no down-sampling), 129 ids -> S = 128 + 729 + 81*82 = 7499."""
    ids, mask, labels, _ = synthetic_batch(geo, b, seed)
    g = torch.Generator().manual_seed(seed)
    img = geo["vision"]["image"]
    images = [torch.randn(10, 3, img, img, generator=g).to(torch.bfloat16) for _ in range(b)]
    return ids, mask, labels, images, [(1024, 1024)] * b


def synthetic_batch(geo, b, seed, text_lens=None):
    """SURVEY.md section 8d inputs: ids uniform in [3, V), IMAGE_TOKEN_INDEX at 35, first 64 text positions ignored.
    text_lens=(lo, hi): per-sample id counts uniform in [lo, hi] (right padded) instead of the fixed 129 -- the
    variable-length case that packed batches are for (not the BASELINE workload)."""
    V = geo["lm"]["vocab"]
    rng = np.random.default_rng(seed)
    n_ids = 129 if text_lens is None else text_lens[1]
    ids = rng.integers(3, V, size=(b, n_ids), dtype=np.int64)
    labels = ids.copy()
    labels[:, :64] = -100
    ids[:, 35] = -200
    labels[:, 35] = -100
    mask = np.ones_like(ids, dtype=bool)
    if text_lens is not None:
        for i, n in enumerate(rng.integers(text_lens[0], text_lens[1] + 1, size=b)):
            mask[i, n:] = False
            ids[i, n:] = 0
            labels[i, n:] = -100
    g = torch.Generator().manual_seed(seed)
    img = geo["vision"]["image"]
    images = [torch.randn(3, img, img, generator=g).to(torch.bfloat16) for _ in range(b)]
    return ids, mask, labels, images


def cpu_baseline(geo, seconds_budget=25.0):
    """Reference CPU path (oracle, torch fp32) timed on this host's cores on a bounded sample: one decoder layer
    fwd+bwd and one ViT layer fwd at b=1 (S=704 / 577 tokens), scaled by layer counts, + lm_head/CE fwd+bwd."""
    from oracle import llava_oracle as O
    import torch.nn.functional as F
    # the GPU box grants this job a 16-core share of the host (more threads only oversubscribe: 256 threads ran 4x slower)
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    l, v = geo["lm"], geo["vision"]
    d, S = l["d"], 704
    g = torch.Generator().manual_seed(0)
    P = {}
    pre = "model.layers.0."
    kvh = l.get("kv_heads", l["heads"])
    kvd = d // l["heads"] * kvh             # grouped-query attention (Qwen2): k / v projections are kv_heads * head_dim wide
    for n, shp in (("self_attn.q_proj.weight", (d, d)), ("self_attn.k_proj.weight", (kvd, d)), ("self_attn.v_proj.weight", (kvd, d)),
                   ("self_attn.o_proj.weight", (d, d)), ("mlp.gate_proj.weight", (l["ffn"], d)), ("mlp.up_proj.weight", (l["ffn"], d)),
                   ("mlp.down_proj.weight", (d, l["ffn"])), ("input_layernorm.weight", (d,)), ("post_attention_layernorm.weight", (d,))):
        P[pre + n] = (torch.randn(*shp, generator=g) * 0.02).requires_grad_(True)
    x = torch.randn(1, S, d, generator=g).requires_grad_(True)
    cos, sin = O.rope_cos_sin(S, d // l["heads"], l.get("rope_theta", 1e4))

    def dec():
        y = O.decoder_layer(x, P, pre, l["heads"], [S], cos, sin, eps=l.get("rms_eps", 1e-5), kv_heads=kvh)
        y.sum().backward()

    def timeit(fn, reps):
        fn()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t) / reps

    t_dec = timeit(dec, 8)
    # ViT layer forward (frozen tower)
    dv, N = v["d"], (v["image"] // v["patch"]) ** 2 + 1
    vp = "model.vision_tower.vision_tower.vision_model."
    geo1 = {"vision": dict(v, layers=2), "lm": l}
    PV = {k: torch.randn(*shp, generator=g) * 0.02 for k, shp in O.param_shapes(geo1).items() if k.startswith(vp)}
    pix = torch.randn(1, 3, v["image"], v["image"], generator=g)
    with torch.no_grad():
        if v.get("kind") == "siglip":
            t_vit = timeit(lambda: O.siglip_vision_hidden(PV, geo1, pix), 40)    # embeddings + 1 layer (the loaded tower has layers - 1)
        else:
            t_vit = timeit(lambda: O.clip_vision_hidden(PV, geo1, pix, -2), 40)  # embeddings + 1 layer
    # head
    wh = (torch.randn(l["vocab"], d, generator=g) * 0.02).requires_grad_(True)
    hN = torch.randn(1, S, d, generator=g).requires_grad_(True)
    lab = torch.randint(0, l["vocab"], (1, S), generator=g)

    def head():
        O.causal_lm_loss(F.linear(hN, wh), lab).backward()

    t_head = timeit(head, 4)
    step = l["layers"] * t_dec + (v["layers"] - 1) * t_vit + t_head
    return {"value": 1.0 / step, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32, b=1: 1 decoder layer fwd+bwd ({t_dec:.2f}s) x{l['layers']} + 1 ViT layer fwd ({t_vit:.2f}s) "
                      f"x{v['layers'] - 1} + lm_head/CE fwd+bwd ({t_head:.2f}s); optimizer not included"}


class GemmTimer:
    """Wraps the GEMM entry points with HIP events (on the launch stream) to get the dominant kernel's flops and time."""

    def __init__(self, ops):
        self.ops = ops
        self.orig_nt, self.orig = ops.gemm_nt, ops.gemm
        self.orig_fused = (ops.gemm_rope, ops.gemm_swiglu_fwd, ops.gemm_swiglu_bwd)
        self.records = []
        self.depth = 0

    def _timed(self, fn, flops, abytes, *args, **kw):
        if self.depth:                      # gemm_nt forwards to gemm: count a launch once
            return fn(*args, **kw)
        self.depth += 1
        try:
            return self._timed1(fn, flops, abytes, *args, **kw)
        finally:
            self.depth -= 1

    def _timed1(self, fn, flops, abytes, *args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kw)
        e1.record()
        # algorithmic bytes of the launch: every operand read once, the output written once
        abytes += sum(o.numel() * o.element_size() for o in (out if isinstance(out, tuple) else (out,)))
        for k in ("residual", "bias", "a2", "b2"):
            t = kw.get(k)
            if t is not None:
                abytes += t.numel() * t.element_size()
        self.records.append((flops, e0, e1, abytes))
        return out

    def __enter__(self):
        nbytes = lambda t: t.numel() * t.element_size()

        def nt(a, b, *args, **kw):
            return self._timed(self.orig_nt, 2.0 * a.shape[0] * b.shape[0] * a.shape[1], nbytes(a) + nbytes(b), a, b, *args, **kw)

        def gen(a, b, ta=False, tb=False, **kw):
            k, m = a.shape if ta else a.shape[::-1]
            n = b.shape[1] if tb else b.shape[0]
            return self._timed(self.orig, 2.0 * m * n * k, nbytes(a) + nbytes(b), a, b, ta=ta, tb=tb, **kw)
        def rope(a, b, *args, **kw):
            return self._timed(self.orig_fused[0], 2.0 * a.shape[0] * b.shape[0] * a.shape[1], nbytes(a) + nbytes(b), a, b, *args, **kw)

        def sw_fwd(a, wgu, F, *args, **kw):      # algorithmic bytes: reads x and [gate; up], writes gate|up and act
            return self._timed(self.orig_fused[1], 2.0 * a.shape[0] * 2 * F * a.shape[1], nbytes(a) + nbytes(wgu) + a.shape[0] * F * 2, a, wgu, F, *args, **kw)

        def sw_bwd(dy, wd, gu, F, *args, **kw):  # reads dy, W_down and gate|up, writes d(gate|up)
            return self._timed(self.orig_fused[2], 2.0 * dy.shape[0] * F * dy.shape[1], nbytes(dy) + nbytes(wd) + nbytes(gu), dy, wd, gu, F, *args, **kw)
        self.ops.gemm_nt, self.ops.gemm = nt, gen
        self.ops.gemm_rope, self.ops.gemm_swiglu_fwd, self.ops.gemm_swiglu_bwd = rope, sw_fwd, sw_bwd
        return self

    def __exit__(self, *exc):
        self.ops.gemm_nt, self.ops.gemm = self.orig_nt, self.orig
        self.ops.gemm_rope, self.ops.gemm_swiglu_fwd, self.ops.gemm_swiglu_bwd = self.orig_fused

    def summary(self):
        torch.cuda.synchronize()
        flops = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return flops, ms, len(self.records), sum(r[3] for r in self.records)


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator is created; the contract is ONE JSON line on stdout, so file
    descriptor 1 points at stderr while the process group comes up (fd level: the banner is written by the C library)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="pairs per GPU per step (global batch 256 at 8 GPUs)")
    ap.add_argument("--geometry", default="llava15_7b")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="cxr", choices=["cxr", "anyres", "lora", "radvlm"],
                    help="cxr = BASELINE configs 2/3 (headline); anyres = config 4 (S=3056); lora = config 5 (r=64 adapters); "
                         "radvlm = SURVEY 8f.1 (Qwen2-7B + SigLIP, anyres_max_9, all parts tunable; use --geometry llava_ov_qwen2_7b --batch 2)")
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--text-lens", default=None, help="lo,hi: variable per-sample text lengths (cxr workload only; not the BASELINE workload)")
    ap.add_argument("--packed", default="auto", choices=["auto", "0", "1"], help="packed (varlen) decoder batches")
    ap.add_argument("--reserved-cus", type=int, default=None,
                    help="compute units the GEMM round planning leaves to the overlapped RCCL all-reduce (default 0: the collectives are active for a small part of backward only)")
    ap.add_argument("--head-rows", default="labeled", choices=["labeled", "head", "all"],
                    help="labeled (default): the last decoder layer's o_proj / norm / MLP, the final norm, lm_head and the cross entropy run on the "
                         "rows that carry a label (same loss and gradients); head: only the final norm / lm_head / cross entropy do; all: every row (A/B)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="N = 1 only: create the nccl (= RCCL) process group of one rank and issue every gradient bucket's all-reduce on the "
                         "side stream anyway (single-GPU rehearsal of the N > 1 path: stream / event ordering, one-tile GEMM blocks beside the "
                         "collectives).  Not the headline configuration.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU over RCCL), BEFORE anything touches the
        # GPU -- no HIP library is loaded and no torch.cuda call has run in this process -- as a child process whose exit code
        # becomes ours (never exec: a process that has initialised the GPU must not be replaced)
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from radvlm_amd import lib, ops
    from radvlm_amd.config import GEOMETRIES
    from radvlm_amd.engine import LlavaEngine
    lib.load()
    if os.environ.get("RV_TAIL_SPLIT") == "0":   # A/B switch for the GEMM tail-round K-split (default on)
        lib.load().rv_gemm_select_kernel(20)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or torchrun --nproc-per-node N")
    # rehearsal hook (tests only): RV_BENCH_REHEARSAL=1 runs all ranks on cuda:0 over gloo, because RCCL refuses two
    # ranks on one device and the multi-GPU node is the driver's; the measured path always uses nccl (= RCCL)
    rehearsal = os.environ.get("RV_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    pg = None
    force_pg = args.force_process_group and world == 1
    quiet = _StdoutToStderr()
    if world > 1 or force_pg:
        quiet.__enter__()
    if force_pg:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        pg = torch.distributed.group.WORLD
    if world > 1:
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        pg = torch.distributed.group.WORLD
    ranks_seen, backend = 1, None
    if world > 1 or force_pg:
        seen = torch.zeros(world, dtype=torch.int32, device=f"cuda:{local}")
        seen[rank] = 1
        torch.distributed.all_reduce(seen)
        ranks_seen, backend = int(seen.sum()), torch.distributed.get_backend()
        assert ranks_seen == world, f"{ranks_seen} of {world} ranks answered"
        torch.cuda.synchronize()
        quiet.__exit__()          # the communicator exists now (first collective done): stdout is ours again
    # GEMM round planning budget: the device's CU count minus what --reserved-cus leaves to the RCCL kernels that run beside backward
    reserved = args.reserved_cus if args.reserved_cus is not None else int(os.environ.get("RV_GEMM_RESERVED_CUS", "0"))
    cu_budget = lib.load().rv_gemm_set_cu_budget(0, reserved)
    geo = GEOMETRIES[args.geometry]
    kw = {}
    if args.workload == "anyres":
        kw = dict(merge_type="spatial_unpad", image_aspect_ratio="anyres",
                  image_grid_pinpoints=[[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]])
    if args.workload == "lora":
        kw = dict(lora=dict(r=64, alpha=16, dropout=0.05))
    if args.workload == "radvlm":
        kw = dict(merge_type="spatial_unpad", image_aspect_ratio="anyres_max_9", image_grid_pinpoints="(1x1),...,(6x6)",
                  train_vision_tower=True)
    kw["packed"] = {"auto": "auto", "0": False, "1": True}[args.packed]
    eng = LlavaEngine(geo, device=f"cuda:{local}", init="fast", seed=0, process_group=pg, force_grad_sync=force_pg, **kw)
    eng.head_rows = "all" if args.head_rows == "all" else "labeled"
    eng.last_layer_rows = "labeled" if args.head_rows == "labeled" else "all"
    eng.init_optimizer()
    make = {"anyres": synthetic_batch_anyres, "radvlm": synthetic_batch_radvlm}.get(args.workload, synthetic_batch)
    if args.text_lens and args.workload == "cxr":
        batch = synthetic_batch(geo, args.batch, seed=1234 + rank, text_lens=tuple(int(x) for x in args.text_lens.split(",")))
    else:
        batch = make(geo, args.batch, seed=1234 + rank)
    # four batches of the same shape, cycled: with one batch re-used every step the 7B model memorises it within the bench's 15 steps
    # (loss 11 -> 0.01), and the backward GEMMs of a solved batch multiply near-zero gradients
    batches = [batch]
    if not (args.text_lens and args.workload == "cxr"):
        batches += [make(geo, args.batch, seed=1234 + rank + 1000 * k) for k in range(1, 4)]
    counter = [0]

    def step():
        batch = batches[counter[0] % len(batches)]
        counter[0] += 1
        loss = eng.forward(*batch)
        eng.backward()
        eng.optimizer_step(lr=args.lr, weight_decay=0.0, max_grad_norm=1.0)
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's own K steps, before it waits for the others (straggler spread, N > 1)
    fence()
    dt = time.perf_counter() - t0
    dt_min = dt_max = dt_own
    if world > 1:
        t = torch.tensor([dt, dt_own, -dt_own], device=f"cuda:{local}")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, dt_max, dt_min = float(t[0]), float(t[1]), -float(t[2])
    final_loss = float(loss)
    ms_per_step = dt / args.steps * 1e3
    pairs = args.batch * world * args.steps
    value = pairs / dt

    # the same step with NOTHING restricted to the labelled rows (every row through the last layer's MLP, the final norm, lm_head and the cross
    # entropy: the reference's own evaluation order), timed beside the headline number on the same box: `config.every_row_variant`
    every_row = None
    if args.head_rows == "labeled":
        eng.head_rows = eng.last_layer_rows = "all"
        for _ in range(max(2, args.warmup)):
            step()
        fence()
        t1 = time.perf_counter()
        n_alt = args.steps
        for _ in range(n_alt):
            step()
        fence()
        dt_alt = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_alt], device=f"cuda:{local}")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_alt = float(t[0])
        every_row = {"pairs_per_s": args.batch * world * n_alt / dt_alt, "ms_per_step": dt_alt / n_alt * 1e3, "steps": n_alt}
        eng.head_rows = eng.last_layer_rows = "labeled"

    # exposed communication: time the compute stream spends waiting for the last gradient buckets (events around finish_grad_sync)
    comm_wait_ms = None
    if eng.sync is not None:
        eng.sync.timing = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        comm_wait_ms = sum(a.elapsed_time(b) for a, b in eng.sync.timing) / 2
        eng.sync.timing = None

    # dominant kernel (bf16 MFMA GEMM): algorithmic flops / measured launch time, one extra instrumented step
    with GemmTimer(ops) as gt:
        step()
        gflops, gms, nlaunch, gbytes = gt.summary()
    from radvlm_amd.build_id import kernel_source_sha256
    src_hash = kernel_source_sha256()
    roofline = {"bound": "mfma", "kernel": "gemm_kernel_256 (all operand forms; 128x128 kernel for small shapes)", "achieved": gflops / (gms * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "traffic": None, "launches_per_step": nlaunch, "gemm_ms_per_step": gms,
                "algorithmic_flops_per_launch": gflops / max(1, nlaunch), "algorithmic_bytes_per_launch": gbytes / max(1, nlaunch),
                "avg_launch_us": gms * 1e3 / max(1, nlaunch)}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    # counter-derived fields come from separate `rocprofv3 --pmc` passes of this very command (profiles/README.md); they are quoted
    # only when they were taken on the kernel sources this run uses (radvlm_amd.build_id) and for the same workload -- else null
    headline = args.workload == "cxr" and args.batch == 32 and args.geometry == "llava15_7b" and not args.text_lens

    def pmc(suffix):
        """The newest profiles/rNN_<suffix> whose kernel-source hash is the one this run executes (else None)."""
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True):
            try:
                with open(path) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                continue
            if headline and d.get("kernel_source_sha256") == src_hash:
                return d, "profiles/" + os.path.basename(path)
        return None, None
    tr, tr_path = pmc("pmc_gemm_hbm_traffic.json")
    if tr:
        roofline["traffic"] = tr["gemm_kernel_256"]["hbm_bytes_per_launch"]
        roofline["traffic_over_algorithmic"] = roofline["traffic"] / roofline["algorithmic_bytes_per_launch"]
        roofline["traffic_source"] = f"{tr_path} (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command, kernel sources {src_hash})"
    else:
        roofline["traffic_source"] = f"null: no PMC pass on record for kernel sources {src_hash} / this workload"
    mf, _ = pmc("pmc_mfma_util.json")
    if mf:
        roofline["pmc_mfma"] = mf["gemm_kernel_256_all_forms"]
    tf_pair = TF_PER_PAIR.get(args.geometry) if args.workload == "cxr" else None
    # the whole step against the MFMA peak, from the flops the GEMM launches of one step EXECUTE (measured above), not from a nominal count:
    # the final norm / lm_head / cross entropy run on the rows that carry a label (engine.head_rows; 64 of 704 rows per pair here), so the
    # step executes less than SURVEY 8d's 28.75 TF/pair, which counts the head on every row -- that nominal figure is reported beside it
    roofline["step_executed_gemm_tflops"] = gflops / 1e12 / (ms_per_step * 1e-3)
    roofline["step_frac_of_mfma_peak"] = roofline["step_executed_gemm_tflops"] / PEAK_BF16_TFLOPS
    if tf_pair:
        roofline["step_nominal_tflops_survey_8d"] = tf_pair * args.batch / (ms_per_step * 1e-3)
    # sequence length of the spliced decoder input for the fixed-shape cxr workload: image tokens + ids - the placeholder
    v = geo["vision"]
    s_cxr = (v["image"] // v["patch"]) ** 2 + 129 - 1
    if rank == 0:
        out = {
            "metric": ("train image-instruction pairs/sec, LLaVA-OV Qwen2-7B + SigLIP-so400m 384px anyres_max_9 (SURVEY 8f.1; not the "
                       "BASELINE metric)" if args.workload == "radvlm" else
                       ("train image-instruction pairs/sec, LLaVA-1.5-7B 336px" if args.geometry == "llava15_7b" else
                        f"train image-instruction pairs/sec, geometry {args.geometry} (not the BASELINE metric)")),
            "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.geometry} {'LoRA r=64' if args.workload == 'lora' else 'full fine-tune'} step "
                                   f"({'tower tunable' if args.workload == 'radvlm' else 'ViT frozen'}): fwd+bwd+AdamW, "
                                   f"{ {'anyres': 'anyres 5 tiles, S=3056', 'radvlm': 'anyres_max_9 10 tiles, S=7499'}.get(args.workload, 'S=%d' % s_cxr)}, "
                                   f"{args.batch} pairs/GPU/step", "global_batch": args.batch * world,
                       "seq_len": {"anyres": 3056, "radvlm": 7499}.get(args.workload, s_cxr),
                       "parallelism": f"dp{world}", "final_loss": final_loss, "rows_with_a_label_only": {"all": "nothing", "head": "final norm, lm_head, cross entropy", "labeled": "last layer's o_proj / norm / MLP, final norm, lm_head, cross entropy"}[args.head_rows],
                       **({"every_row_variant": every_row} if every_row else {}),
                       **({"text_lens": args.text_lens, "packed": args.packed} if args.text_lens else {})},
            "roofline": roofline,
            "distributed": {"world": world, "ranks_seen": ranks_seen, "backend": backend if not rehearsal else f"{backend} (rehearsal: all ranks on one GPU)",
                            "grad_sync": None if eng.sync is None else "bucketed sum-all-reduce of the flat bf16 gradient buffer, per-layer buckets on a side stream",
                            "exposed_comm_ms_per_step": comm_wait_ms, "gemm_cu_budget": cu_budget, "reserved_cus": reserved,
                            # each rank's own time for the K steps before the closing barrier: max - min = straggler spread over ranks
                            "ms_per_step_min": dt_min / args.steps * 1e3, "ms_per_step_max": dt_max / args.steps * 1e3,
                            "bucket_count": None if eng.sync is None else len(eng.sync.last_buckets),
                            "bucket_mbytes": None if eng.sync is None else [round((e - s) * 2 / 2 ** 20, 1) for s, e in eng.sync.last_buckets][:4] + ["..."] * (len(eng.sync.last_buckets) > 4)},
            "build": {"kernel_source_sha256": src_hash},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(geo)
        print(json.dumps(out))
    if world > 1 or force_pg:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
