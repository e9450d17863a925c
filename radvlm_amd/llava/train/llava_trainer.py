"""Training-loop host logic: length/modality-grouped sampling, optimizer grouping and a minimal data-parallel
trainer on LlavaEngine.

Restates reference finetuning/llava/train/llava_trainer.py: split_to_even_chunks :51-71,
get_variable_length_grouped_indices :73-86, get_modality_length_grouped_indices :89-126,
get_length_grouped_indices :129-149, get_length_grouped_indices_auto_single :152-164, LengthGroupedSampler :196-237,
_get_train_sampler :273-317 (note: world_size * gradient_accumulation_steps, :283), create_optimizer :356-433.
The loop body the reference inherits from HF Trainer + accelerate + DeepSpeed is replaced by the engine's own step.
"""
import math
import time

import torch
from torch.utils.data import Sampler


def split_to_even_chunks(indices, lengths, num_chunks):
    """Deal `indices` (already sorted by length) into `num_chunks` chunks of equal count and balanced total length."""
    if len(indices) % num_chunks:
        return [indices[i::num_chunks] for i in range(num_chunks)]
    cap = len(indices) // num_chunks
    chunks = [[] for _ in range(num_chunks)]
    load = [0] * num_chunks
    for idx in indices:
        k = load.index(min(load))
        chunks[k].append(idx)
        load[k] += lengths[idx]
        if len(chunks[k]) == cap:
            load[k] = float("inf")
    return chunks


def _megabatches(seq, size):
    return [seq[i:i + size] for i in range(0, len(seq), size)]


def get_length_grouped_indices(lengths, batch_size, world_size, generator=None, merge=True):
    perm = torch.randperm(len(lengths), generator=generator)
    mb = [m.tolist() for m in _megabatches(perm, world_size * batch_size)]
    mb = [sorted(m, key=lambda i: lengths[i], reverse=True) for m in mb]
    mb = [split_to_even_chunks(m, lengths, world_size) for m in mb]
    return [i for m in mb for chunk in m for i in chunk]


def _hf_length_grouped(lengths, batch_size, mega_batch_mult=None, generator=None):
    """transformers.trainer_pt_utils.get_length_grouped_indices (third-party; restated from its published algorithm)."""
    if mega_batch_mult is None:
        mega_batch_mult = min(len(lengths) // (batch_size * 4), 50)
        if mega_batch_mult == 0:
            mega_batch_mult = 1
    perm = torch.randperm(len(lengths), generator=generator)
    size = mega_batch_mult * batch_size
    mbs = [perm[i:i + size].tolist() for i in range(0, len(lengths), size)]
    mbs = [sorted(m, key=lambda i: lengths[i], reverse=True) for m in mbs]
    maxes = [lengths[m[0]] for m in mbs]
    k = torch.argmax(torch.tensor(maxes)).item()
    mbs[0][0], mbs[k][0] = mbs[k][0], mbs[0][0]
    return [i for m in mbs for i in m]


def get_length_grouped_indices_auto_single(lengths, batch_size, world_size, generator=None):
    idx = _hf_length_grouped(lengths, batch_size * world_size, generator=generator)
    mb = _megabatches(idx, world_size * batch_size)
    mb = [sorted(m, key=lambda i: lengths[i], reverse=True) for m in mb]
    mb = [split_to_even_chunks(m, lengths, world_size) for m in mb]
    order = torch.randperm(len(mb), generator=generator)
    return [i for k in order for chunk in mb[k] for i in chunk]


def get_variable_length_grouped_indices(lengths, batch_size, world_size, megabatch_mult=8, generator=None):
    key = torch.randperm(len(lengths), generator=generator)
    by_len = sorted(range(len(lengths)), key=lambda i: lengths[i], reverse=True)
    mb = _megabatches(by_len, world_size * batch_size * megabatch_mult)
    flat = [i for m in mb for i in sorted(m, key=lambda i: key[i], reverse=True)]
    batches = _megabatches(flat, world_size * batch_size)
    order = torch.randperm(len(batches), generator=generator)
    return [i for k in order for i in batches[k]]


def _by_modality(lengths, batch_size, world_size, generator, single, keep_tail):
    assert all(l != 0 for l in lengths), "Should not have zero length."
    if all(l > 0 for l in lengths) or all(l < 0 for l in lengths):
        return single(lengths, batch_size, world_size, generator=generator)
    mm = [(i, l) for i, l in enumerate(lengths) if l > 0]
    lang = [(i, -l) for i, l in enumerate(lengths) if l < 0]
    mm_order = [mm[i][0] for i in single([l for _, l in mm], batch_size, world_size, generator=None)]
    lang_order = [lang[i][0] for i in single([l for _, l in lang], batch_size, world_size, generator=None)]
    size = world_size * batch_size
    mm_mb, lang_mb = _megabatches(mm_order, size), _megabatches(lang_order, size)
    tail = mm_mb[-1] + lang_mb[-1]
    mbs = mm_mb[:-1] + lang_mb[:-1]
    order = torch.randperm(len(mbs), generator=generator)
    mbs = [mbs[k] for k in order]
    if keep_tail and tail:
        mbs.append(sorted(tail))
    return [i for m in mbs for i in m]


def get_modality_length_grouped_indices(lengths, batch_size, world_size, generator=None):
    """lengths > 0: multimodal samples, < 0: text-only; megabatches never mix modalities except the last."""
    return _by_modality(lengths, batch_size, world_size, generator, get_length_grouped_indices, True)


def get_modality_length_grouped_indices_auto(lengths, batch_size, world_size, generator=None):
    return _by_modality(lengths, batch_size, world_size, generator, get_length_grouped_indices_auto_single, False)


class LengthGroupedSampler(Sampler):
    def __init__(self, batch_size, world_size, lengths=None, generator=None, variable_length=False, group_by_modality=False,
                 group_by_modality_auto=False):
        if lengths is None:
            raise ValueError("Lengths must be provided.")
        self.batch_size, self.world_size, self.lengths, self.generator = batch_size, world_size, lengths, generator
        self.variable_length, self.group_by_modality, self.group_by_modality_auto = variable_length, group_by_modality, group_by_modality_auto

    def __len__(self):
        return len(self.lengths)

    def __iter__(self):
        a = (self.lengths, self.batch_size, self.world_size)
        if self.variable_length:
            assert not self.group_by_modality, "Variable length grouping is not supported with modality grouping."
            return iter(get_variable_length_grouped_indices(*a, generator=self.generator))
        if self.group_by_modality:
            return iter(get_modality_length_grouped_indices(*a, generator=self.generator))
        if self.group_by_modality_auto:
            return iter(get_modality_length_grouped_indices_auto(*a, generator=self.generator))
        return iter(get_length_grouped_indices_auto_single(*a, generator=self.generator))


def shard_for_rank(indices, per_device_batch, world_size, rank):
    """Rank r takes the r-th `per_device_batch` slice of every consecutive world batch (what DistributedSampler-free
    batching under the grouped order amounts to): disjoint samples per rank, length-balanced by construction."""
    wb = per_device_batch * world_size
    out = []
    for s in range(0, len(indices) - wb + 1, wb):
        out.extend(indices[s + rank * per_device_batch: s + (rank + 1) * per_device_batch])
    return out


def cosine_lr(step, total_steps, base_lr, warmup_steps):
    """HF 'cosine' schedule with linear warmup (lr_scheduler_type cosine, warmup_ratio 0.03 in the reference script)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    p = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * p)))


class LLaVATrainer:
    """Minimal trainer over LlavaEngine with the reference trainer's constructor shape
    (model, tokenizer, args, train_dataset, data_collator) and .train()."""

    def __init__(self, model=None, tokenizer=None, args=None, train_dataset=None, data_collator=None, **_):
        self.model, self.tokenizer, self.args = model, tokenizer, args
        self.train_dataset, self.data_collator = train_dataset, data_collator
        self.state = {"global_step": 0, "log_history": []}

    def _get_train_sampler(self):
        a = self.args
        world = getattr(a, "world_size", 1) * getattr(a, "gradient_accumulation_steps", 1)
        g = torch.Generator().manual_seed(getattr(a, "seed", 42))
        kw = dict(batch_size=a.per_device_train_batch_size, world_size=world, generator=g)
        if getattr(a, "group_by_length", False):
            return LengthGroupedSampler(lengths=self.train_dataset.lengths, **kw)
        if getattr(a, "group_by_modality_length", False):
            return LengthGroupedSampler(lengths=self.train_dataset.modality_lengths, group_by_modality=True, **kw)
        if getattr(a, "group_by_modality_length_auto", False):
            return LengthGroupedSampler(lengths=self.train_dataset.modality_lengths, group_by_modality_auto=True, **kw)
        if getattr(a, "group_by_varlen", False):
            return LengthGroupedSampler(batch_size=a.per_device_train_batch_size * a.gradient_accumulation_steps,
                                        world_size=getattr(a, "world_size", 1), lengths=self.train_dataset.lengths,
                                        generator=g, variable_length=True)
        return None

    def train(self, resume_from_checkpoint=None):
        a = self.args
        eng = self.model.engine
        world, rank = getattr(a, "world_size", 1), getattr(a, "process_index", 0)
        sampler = self._get_train_sampler()
        n = len(self.train_dataset)
        order = list(iter(sampler)) if sampler is not None else torch.randperm(n, generator=torch.Generator().manual_seed(getattr(a, "seed", 42))).tolist()
        mine = shard_for_rank(order, a.per_device_train_batch_size, world, rank) if world > 1 else order
        bs, accum = a.per_device_train_batch_size, getattr(a, "gradient_accumulation_steps", 1)
        steps_per_epoch = len(mine) // (bs * accum)
        total = int(getattr(a, "max_steps", -1)) if getattr(a, "max_steps", -1) > 0 else int(steps_per_epoch * a.num_train_epochs)
        warm = int(getattr(a, "warmup_steps", 0) or total * getattr(a, "warmup_ratio", 0.0))
        pos = 0
        for step in range(1, total + 1):
            t0 = time.perf_counter()
            losses = []
            for micro in range(accum):
                idx = mine[pos:pos + bs]
                pos = (pos + bs) % max(1, len(mine) - bs + 1)
                batch = self.data_collator([self.train_dataset[i] for i in idx])
                eng.sync_this_backward = micro == accum - 1
                out = self.model(**batch)
                out.loss.backward()
                losses.append(out.loss)
            lr = cosine_lr(step, total, a.learning_rate, warm)
            eng.optimizer_step(lr=lr, weight_decay=a.weight_decay, betas=(getattr(a, "adam_beta1", 0.9), getattr(a, "adam_beta2", 0.999)),
                               eps=getattr(a, "adam_epsilon", 1e-8), max_grad_norm=getattr(a, "max_grad_norm", 1.0),
                               mm_projector_lr=getattr(a, "mm_projector_lr", None),
                               mm_vision_tower_lr=getattr(a, "mm_vision_tower_lr", None))
            self.state["global_step"] = step
            if step % max(1, getattr(a, "logging_steps", 1)) == 0:
                rec = {"step": step, "loss": float(sum(float(l) for l in losses) / len(losses)), "learning_rate": lr,
                       "step_time_s": time.perf_counter() - t0}
                self.state["log_history"].append(rec)
                if rank == 0:
                    print(rec, flush=True)
        return self.state
