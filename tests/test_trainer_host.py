"""Host-side training logic that needs no GPU: LR schedule and warm-up of HF Trainer, the per-epoch sample stream, and loading
pretrained checkpoints from local directories (what the reference gets from from_pretrained, train/train.py:1358-1427)."""
import json
import math
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from radvlm_amd.config import GEOMETRIES
from radvlm_amd.llava.train.llava_trainer import (LengthGroupedSampler, cosine_lr, epoch_index_batches, lr_at, shard_for_rank,
                                                  warmup_steps_for)


def test_lr_schedule_matches_hf_lambda_lr():
    """HF steps the scheduler after the optimizer: update k runs at lambda(k - 1).  Known values of
    get_cosine_schedule_with_warmup(num_warmup_steps=3, num_training_steps=10) x base 2e-5."""
    total, warm, base = 10, 3, 2e-5
    lam = lambda s: s / warm if s < warm else max(0.0, 0.5 * (1 + math.cos(math.pi * (s - warm) / (total - warm))))
    got = [lr_at(k, total, base, warm) for k in range(1, total + 1)]
    assert got == pytest.approx([base * lam(k - 1) for k in range(1, total + 1)], rel=1e-12)
    assert got[0] == 0.0 and got[-1] > 0.0 and got[3] == base          # first update lr 0, peak right after warm-up, last one > 0
    assert warmup_steps_for(SimpleNamespace(warmup_steps=0, warmup_ratio=0.03), 100) == 3
    assert warmup_steps_for(SimpleNamespace(warmup_steps=0, warmup_ratio=0.03), 101) == 4      # ceil, not int()
    assert warmup_steps_for(SimpleNamespace(warmup_steps=7, warmup_ratio=0.5), 100) == 7
    assert lr_at(1, 10, 1.0, 0, "linear") == 1.0 and lr_at(10, 10, 1.0, 0, "linear") == pytest.approx(0.1)
    assert lr_at(5, 10, 1.0, 2, "constant_with_warmup") == 1.0
    with pytest.raises(ValueError):
        lr_at(1, 10, 1.0, 0, "polynomial")


def test_epoch_stream_reshuffles_and_covers_whole_epochs():
    n, bs, world, accum = 23, 2, 2, 2
    lengths = [5 + (i * 7) % 11 for i in range(n)]
    seen_orders = []
    for rank in range(world):
        sampler = LengthGroupedSampler(bs, world * accum, lengths=lengths, generator=torch.Generator().manual_seed(42))
        it, spe = epoch_index_batches(n, sampler, 42, bs, world, rank, accum)
        assert spe == (n // (bs * world)) // accum == 2
        epochs = [[next(it) for _ in range(spe * accum)] for _ in range(3)]
        seen_orders.append(epochs)
        for e in epochs:
            flat = [i for b in e for i in b]
            assert len(flat) == len(set(flat)) == spe * accum * bs          # no sample twice inside an epoch
    for e in range(3):      # ranks are disjoint within an epoch, and epochs differ from each other (fresh order per epoch)
        a = {i for b in seen_orders[0][e] for i in b}
        b = {i for b in seen_orders[1][e] for i in b}
        assert not (a & b)
    assert seen_orders[0][0] != seen_orders[0][1] != seen_orders[0][2]
    # without a sampler: seeded permutations, new one per epoch, same stream for the same seed
    it1, _ = epoch_index_batches(10, None, 7, 2, 1, 0, 1)
    it2, _ = epoch_index_batches(10, None, 7, 2, 1, 0, 1)
    s1 = [next(it1) for _ in range(15)]
    assert s1 == [next(it2) for _ in range(15)] and s1[:5] != s1[5:10]
    with pytest.raises(ValueError):
        epoch_index_batches(3, None, 0, 2, 2, 0, 1)


def _save_toy_checkpoint(tmp_path, geo, drop=None, split_tower=False):
    """A local HF-format LLaVA checkpoint (config.json + model.safetensors) of the toy geometry from the oracle's portable params."""
    from oracle import llava_oracle as O
    from safetensors.torch import save_file
    P = {k: v.to(torch.bfloat16) for k, v in O.make_params(geo, seed=0).items()}
    l, v = geo["lm"], geo["vision"]
    lm_dir = tmp_path / "lm"
    lm_dir.mkdir()
    cfg = {"model_type": "llava_llama", "hidden_size": l["d"], "intermediate_size": l["ffn"], "num_hidden_layers": l["layers"],
           "num_attention_heads": l["heads"], "num_key_value_heads": l["heads"], "vocab_size": l["vocab"], "rms_norm_eps": 1e-5,
           "rope_theta": 10000.0}
    tower_dir = None
    if split_tower:     # a plain LM checkpoint + a separate CLIP checkpoint in transformers' own key layout
        tower_dir = tmp_path / "clip"
        tower_dir.mkdir()
        pre = "model.vision_tower.vision_tower."
        save_file({k[len(pre):]: t.contiguous() for k, t in P.items() if k.startswith(pre)}, str(tower_dir / "model.safetensors"))
        (tower_dir / "config.json").write_text(json.dumps({"model_type": "clip_vision_model", "hidden_size": v["d"], "intermediate_size": v["ffn"],
                                                           "num_hidden_layers": v["layers"], "num_attention_heads": v["heads"],
                                                           "image_size": v["image"], "patch_size": v["patch"]}))
        P = {k: t for k, t in P.items() if not k.startswith(pre) and "mm_projector" not in k}
    else:
        cfg["mm_vision_geometry"] = dict(v)
    if drop:
        P = {k: t for k, t in P.items() if drop not in k}
    # two shards + an index, like the 7B checkpoints
    keys = sorted(P)
    half = len(keys) // 2
    shards = {"model-00001-of-00002.safetensors": keys[:half], "model-00002-of-00002.safetensors": keys[half:]}
    for name, ks in shards.items():
        save_file({k: P[k].contiguous() for k in ks}, str(lm_dir / name))
    (lm_dir / "model.safetensors.index.json").write_text(json.dumps({"weight_map": {k: n for n, ks in shards.items() for k in ks}}))
    (lm_dir / "config.json").write_text(json.dumps(cfg))
    return str(lm_dir), (str(tower_dir) if tower_dir else None)


@pytest.mark.parametrize("split_tower", [False, True])
def test_pretrained_checkpoint_loads_into_the_flat_stores(tmp_path, split_tower):
    from oracle import llava_oracle as O
    from radvlm_amd.checkpoint_io import load_pretrained
    from radvlm_amd.engine import LlavaEngine
    from radvlm_amd.llava.train.train import ModelArguments, resolve_model_sources
    geo = GEOMETRIES["toy"]
    lm_dir, tower_dir = _save_toy_checkpoint(tmp_path, geo, split_tower=split_tower)
    geometry, ld, td, is_qwen, true_vocab = resolve_model_sources(ModelArguments(model_name_or_path=lm_dir, vision_tower=tower_dir))
    assert not is_qwen and true_vocab == 1000 and ld == lm_dir
    assert geometry["lm"] == dict(geo["lm"], rope_theta=10000.0, rms_eps=1e-5) and geometry["vision"] == geo["vision"]
    eng = LlavaEngine(geometry, device="cpu", init="fast", seed=123)          # random first: loading must overwrite every LM / tower tensor
    before = eng.lm.view("model.mm_projector.0.weight").clone()
    load_pretrained(eng, lm_path=ld, tower_path=td)
    P = O.make_params(geo, seed=0)
    for name in ("model.embed_tokens.weight", "model.layers.1.mlp.down_proj.weight", "model.norm.weight", "lm_head.weight",
                 "model.vision_tower.vision_tower.vision_model.encoder.layers.2.mlp.fc1.bias",
                 "model.vision_tower.vision_tower.vision_model.embeddings.class_embedding"):
        store = eng.lm if name in eng.lm.offsets else eng.vis
        assert torch.equal(store.view(name), P[name].to(torch.bfloat16)), name
    proj = eng.lm.view("model.mm_projector.0.weight")
    assert torch.equal(proj, before) if split_tower else torch.equal(proj, P["model.mm_projector.0.weight"].to(torch.bfloat16))


def test_missing_weights_and_remote_names_raise(tmp_path):
    from radvlm_amd.checkpoint_io import load_pretrained
    from radvlm_amd.engine import LlavaEngine
    from radvlm_amd.llava.train.train import ModelArguments, resolve_model_sources
    geo = GEOMETRIES["toy"]
    with pytest.raises(FileNotFoundError):           # a hub name is not fetched and does not fall back to random weights
        resolve_model_sources(ModelArguments(model_name_or_path="lmsys/vicuna-7b-v1.5", vision_tower="openai/clip-vit-large-patch14-336"))
    lm_dir, _ = _save_toy_checkpoint(tmp_path, geo, drop="layers.1.self_attn.o_proj")
    eng = LlavaEngine(geo, device="cpu", init=None)
    with pytest.raises(KeyError, match="o_proj"):
        load_pretrained(eng, lm_path=lm_dir)
    g, *_ = resolve_model_sources(ModelArguments(model_name_or_path=None, geometry="toy"))      # the explicit random-init switch
    assert g == geo


def test_drop_last_false_completes_the_last_world_batch_from_the_epoch_head():
    """--dataloader_drop_last (llava_trainer.py:348; finetune_radio_7b.sh:86 passes True): True drops the incomplete last world batch,
    False completes it with the first samples of the same epoch's order (accelerate's even_batches sharding)."""
    n, bs, world = 22, 4, 2
    per_rank = {}
    for drop in (True, False):
        for rank in range(world):
            it, steps = epoch_index_batches(n, None, 7, bs, world, rank, 1, drop_last=drop)
            per_rank[(drop, rank)] = ([next(it) for _ in range(steps)], steps)
    assert per_rank[(True, 0)][1] == 2 and per_rank[(False, 0)][1] == 3
    order = torch.randperm(n, generator=torch.Generator().manual_seed(7)).tolist()
    seen = [i for r in range(world) for b in per_rank[(False, r)][0] for i in b]
    assert sorted(seen) == sorted(order + order[:2])                      # 22 samples + the 2 of the epoch's head that fill batch 3
    assert per_rank[(False, 0)][0][:2] == per_rank[(True, 0)][0]          # the complete batches are the same either way


def test_save_total_limit_keeps_the_newest_checkpoints(tmp_path):
    """--save_total_limit N (HF Trainer._rotate_checkpoints; finetune_radio_7b.sh:72 passes 1)."""
    from radvlm_amd.llava.train.llava_trainer import LLaVATrainer
    for step in (5, 10, 100, 20):
        os.makedirs(tmp_path / f"checkpoint-{step}")
        (tmp_path / f"checkpoint-{step}" / "trainer_state.json").write_text("{}")
    os.makedirs(tmp_path / "checkpoint-final")            # not a step directory: never touched
    tr = LLaVATrainer(args=SimpleNamespace(save_total_limit=2))
    assert tr._rotate_checkpoints(str(tmp_path), rank=1) == []              # only rank 0 deletes
    assert sorted(tr._rotate_checkpoints(str(tmp_path), rank=0)) == ["checkpoint-10", "checkpoint-5"]
    assert sorted(os.listdir(tmp_path)) == ["checkpoint-100", "checkpoint-20", "checkpoint-final"]
    tr.args.save_total_limit = None
    assert tr._rotate_checkpoints(str(tmp_path), rank=0) == []


@pytest.mark.parametrize("geometry", ["toy", "toy_qwen"])
def test_bench_cpu_baseline_runs_on_both_model_families(geometry):
    """bench.py's `cpu_baseline` leg (the oracle timed on the host cores) on a CLIP + Llama geometry and on a SigLIP + Qwen2 one (grouped-query
    k / v widths, rope theta, the SigLIP tower): the non-headline workloads' bench lines carry it too."""
    import copy
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rv_bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from radvlm_amd.config import GEOMETRIES
    out = bench.cpu_baseline(copy.deepcopy(GEOMETRIES[geometry]))
    assert out["kind"] == "port" and out["unit"] == "pairs/s" and out["value"] > 0 and out["cores"] >= 1
