"""Training-loop host logic: length/modality-grouped sampling, optimizer grouping and a minimal data-parallel
trainer on LlavaEngine.

Restates reference finetuning/llava/train/llava_trainer.py: split_to_even_chunks :51-71,
get_variable_length_grouped_indices :73-86, get_modality_length_grouped_indices :89-126,
get_length_grouped_indices :129-149, get_length_grouped_indices_auto_single :152-164, LengthGroupedSampler :196-237,
_get_train_sampler :273-317 (note: world_size * gradient_accumulation_steps, :283), create_optimizer :356-433.
The loop body the reference inherits from HF Trainer + accelerate + DeepSpeed is replaced by the engine's own step.
"""
import math
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor

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
    """HF 'cosine' schedule with linear warmup (get_cosine_schedule_with_warmup, num_cycles 0.5): the multiplier at scheduler
    step `step` (0-based: the k-th optimizer update of a run uses step = k - 1, see lr_at)."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    p = (step - warmup_steps) / max(1, total_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * p)))


def warmup_steps_for(args, total_steps):
    """TrainingArguments.get_warmup_steps: warmup_steps if > 0 else ceil(total * warmup_ratio)."""
    ws = int(getattr(args, "warmup_steps", 0) or 0)
    return ws if ws > 0 else math.ceil(total_steps * float(getattr(args, "warmup_ratio", 0.0) or 0.0))


def lr_at(update, total_steps, base_lr, warmup_steps, kind="cosine"):
    """Learning rate of the `update`-th optimizer step (1-based).  HF Trainer steps the scheduler AFTER the optimizer, so update k
    runs at lambda(k - 1): the first update of a warmed-up run has lr 0 and the last one lr > 0 (finetune_radio_7b.sh:
    --lr_scheduler_type cosine --warmup_ratio 0.03)."""
    s = update - 1
    if kind == "cosine":
        return cosine_lr(s, total_steps, base_lr, warmup_steps)
    if kind == "constant":       # HF get_constant_schedule: no warm-up (only constant_with_warmup has one)
        return base_lr
    if s < warmup_steps:
        return base_lr * s / max(1, warmup_steps)
    if kind == "linear":
        return base_lr * max(0.0, (total_steps - s) / max(1, total_steps - warmup_steps))
    if kind == "constant_with_warmup":
        return base_lr
    raise ValueError(f"lr_scheduler_type {kind!r}: cosine, linear, constant and constant_with_warmup are built")


def epoch_index_batches(n, sampler, seed, per_device_batch, world, rank, accum, drop_last=True):
    """The stream of per-rank micro-batches (lists of dataset indices), epoch after epoch, without end.

    Every epoch draws a fresh order -- `iter(sampler)` on a sampler whose generator was seeded once (what re-iterating the
    DataLoader does in HF Trainer), or a seeded permutation without a sampler -- and is consumed whole: the r-th
    `per_device_batch` slice of every world batch belongs to rank r (accelerate's batch sharding of the grouped order,
    llava_trainer.py:129-149), an incomplete last world batch and the micro-batches that do not fill a last optimizer step
    are dropped.  drop_last=False (--dataloader_drop_last False, the HF default; llava_trainer.py:348): the incomplete last world
    batch is completed with samples from the head of the same epoch's order, as accelerate's even_batches sharding does, instead of
    being dropped.  Returns (generator, optimizer steps per epoch)."""
    wb = per_device_batch * world
    world_batches = n // wb if drop_last else -(-n // wb)
    steps_per_epoch = world_batches // accum
    if steps_per_epoch < 1:
        raise ValueError(f"{n} samples do not fill one optimizer step of {wb} x {accum}")
    g = torch.Generator().manual_seed(seed)

    def gen():
        while True:
            order = list(iter(sampler)) if sampler is not None else torch.randperm(n, generator=g).tolist()
            if not drop_last and len(order) % wb:
                order = order + order[:wb - len(order) % wb]
            mine = shard_for_rank(order, per_device_batch, world, rank)
            for j in range(steps_per_epoch * accum):
                yield mine[j * per_device_batch:(j + 1) * per_device_batch]
    return gen(), steps_per_epoch


class BatchPrefetcher:
    """Host input pipeline that keeps ahead of the GPU step (the reference's torch DataLoader with
    ``dataloader_num_workers`` workers, scripts: 4).  ``num_workers`` threads run ``dataset[i]`` (PIL decode / resize /
    anyres tiling release the GIL) for up to ``depth`` batches ahead of the consumer; a feeder thread collates each batch
    in order and pins its image tensors so the engine's host-to-device copies are asynchronous.  With num_workers = 0 it
    degenerates to the synchronous loop.  Sample order and contents are exactly those of the synchronous loop."""

    def __init__(self, dataset, collate, index_batches, num_workers=4, depth=3, pin=None):
        self.dataset, self.collate = dataset, collate
        self.batches = iter(index_batches)
        self.pin = torch.cuda.is_available() if pin is None else pin
        self.sync = num_workers <= 0
        if not self.sync:
            self.pool = ThreadPoolExecutor(max_workers=num_workers, thread_name_prefix="rv-data")
            self.q = queue.Queue(maxsize=max(1, depth))
            self.stop = threading.Event()
            self.feeder = threading.Thread(target=self._feed, name="rv-data-feeder", daemon=True)
            self.feeder.start()

    def _finish(self, samples):
        batch = self.collate(samples)
        if self.pin and "images" in batch:
            batch["images"] = [im.pin_memory() if torch.is_tensor(im) and not im.is_pinned() else im for im in batch["images"]]
        return batch

    def _feed(self):
        try:
            pending = []
            for idx in self.batches:
                if self.stop.is_set():
                    return
                pending.append([self.pool.submit(self.dataset.__getitem__, i) for i in idx])
                while len(pending) > self.q.maxsize:       # keep `depth` batches of samples in flight beyond the queue
                    self._put(self._finish([f.result() for f in pending.pop(0)]))
            for futs in pending:
                self._put(self._finish([f.result() for f in futs]))
            self._put(None)
        except BaseException as e:  # noqa: BLE001 -- surfaced to the consumer
            self._put(e)

    def _put(self, item):
        while not self.stop.is_set():
            try:
                self.q.put(item, timeout=0.2)
                return
            except queue.Full:
                continue

    def __iter__(self):
        return self

    def __next__(self):
        if self.sync:
            return self._finish([self.dataset[i] for i in next(self.batches)])
        item = self.q.get()
        if item is None:
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self):
        """Stop early or clean up at the end: queued samples are cancelled, running ones finish, both threads are joined
        (a worker still alive at interpreter exit aborts the process)."""
        if self.sync:
            return
        self.stop.set()
        self.pool.shutdown(wait=True, cancel_futures=True)
        self.feeder.join(timeout=10)


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
        bs, accum = a.per_device_train_batch_size, max(1, int(getattr(a, "gradient_accumulation_steps", 1) or 1))
        it, steps_per_epoch = epoch_index_batches(len(self.train_dataset), self._get_train_sampler(), getattr(a, "seed", 42), bs, world, rank, accum,
                                                  drop_last=bool(getattr(a, "dataloader_drop_last", False)))
        max_steps = int(getattr(a, "max_steps", -1) or -1)
        total = max_steps if max_steps > 0 else math.ceil(steps_per_epoch * float(a.num_train_epochs))
        warm = warmup_steps_for(a, total)
        kind = getattr(a, "lr_scheduler_type", "cosine") or "cosine"
        kind = getattr(kind, "value", kind)

        start = self._maybe_resume(resume_from_checkpoint)
        eng.check_replicas("start of training")       # after init / load / resume: every rank holds the same parameters and optimizer state
        for _ in range(start * accum):       # a resumed run continues the same sample stream (epoch orders are regenerated from the seed)
            next(it)
        loader = BatchPrefetcher(self.train_dataset, self.data_collator, (next(it) for _ in range((total - start) * accum)),
                                 num_workers=int(getattr(a, "dataloader_num_workers", 0) or 0))
        save_steps = int(getattr(a, "save_steps", 0) or 0)
        # HF Trainer back-propagates loss / gradient_accumulation_steps for every micro-batch (training_step), so the accumulated
        # gradient is the MEAN over the micro-batches: same pre-clip norm, same effect of max_grad_norm as one large batch
        eng.loss_scale = 1.0 / accum
        try:
            for step in range(start + 1, total + 1):
                t0 = time.perf_counter()
                losses = []
                t_data = 0.0
                for micro in range(accum):
                    td = time.perf_counter()
                    batch = next(loader)
                    t_data += time.perf_counter() - td
                    eng.sync_this_backward = micro == accum - 1
                    try:
                        out = self.model(**batch)
                        out.loss.backward()
                    except torch.cuda.OutOfMemoryError:
                        # the "auto" recompute policy sized this batch from an estimate of free memory (fragmented cache blocks count as
                        # free): retry once with every layer recomputed.  Only the first micro-batch of a step can be redone -- later ones
                        # have already accumulated part of their gradients.
                        if micro > 0 or eng.recompute != "auto" or eng._recompute_forced or eng.sync is not None:
                            raise
                        eng._recompute_forced = True
                        eng.ctx = None
                        eng.grad_accum_started = False
                        torch.cuda.empty_cache()
                        if rank == 0:
                            print("[train] out of device memory under recompute='auto': retrying the step with every decoder layer recomputed", flush=True)
                        out = self.model(**batch)
                        out.loss.backward()
                    losses.append(out.loss)
                lr = lr_at(step, total, a.learning_rate, warm, kind)
                scale = lr / a.learning_rate if a.learning_rate else 0.0      # the per-module rates follow the same schedule
                plr, vlr = getattr(a, "mm_projector_lr", None), getattr(a, "mm_vision_tower_lr", None)
                eng.optimizer_step(lr=lr, weight_decay=a.weight_decay, betas=(getattr(a, "adam_beta1", 0.9), getattr(a, "adam_beta2", 0.999)),
                                   eps=getattr(a, "adam_epsilon", 1e-8), max_grad_norm=getattr(a, "max_grad_norm", 1.0),
                                   mm_projector_lr=None if plr is None else plr * scale,
                                   mm_vision_tower_lr=None if vlr is None else vlr * scale)
                self.state["global_step"] = step
                if step % max(1, getattr(a, "logging_steps", 1)) == 0:
                    rec = {"step": step, "loss": float(sum(float(l.detach()) for l in losses) / len(losses)), "learning_rate": lr,
                           "grad_norm": float(eng.last_grad_norm) if eng.last_grad_norm is not None else None,
                           "epoch": step / steps_per_epoch, "step_time_s": time.perf_counter() - t0, "data_wait_s": t_data}
                    self.state["log_history"].append(rec)
                    if rank == 0:
                        print(rec, flush=True)
                if save_steps and step % save_steps == 0 and getattr(a, "output_dir", None):
                    eng.check_replicas(f"step {step}")
                    self.save_checkpoint(os.path.join(a.output_dir, f"checkpoint-{step}"), rank)
                    self._rotate_checkpoints(a.output_dir, rank)
                    # the other ranks wait here while rank 0 writes (a 7B checkpoint with optimizer state is ~95 GB): no rank runs ahead
                    # into the next step's collectives, a resumed run never sees a half-written directory as the newest checkpoint
                    eng.barrier()
        finally:
            eng.loss_scale = 1.0
            loader.close()
        return self.state

    # ------------------------------------------------------------------ checkpoints
    def _adapter_only(self):
        """tune_mm_mlp_adapter, or mm_tunable_parts naming only the projector (llava_trainer.py:436-438)."""
        a = self.args
        parts = (getattr(a, "mm_tunable_parts", None) or "").split(",")
        return bool(getattr(a, "tune_mm_mlp_adapter", False)) or (len(parts) == 1 and parts[0].strip() in ("mm_mlp_adapter", "mm_vision_resampler"))

    def save_checkpoint(self, path, rank=0):
        """checkpoint-N directory with what a resumed run needs (train.py:1699-1702 auto-resume looks for checkpoint-*):
          * full fine-tune: model.safetensors under the reference's state-dict names;
          * projector-only runs (llava_trainer.py:435-457): mm_projector.bin = torch.save of the tensors whose names match
            'mm_projector' (+ 'embed_tokens' with use_im_start_end), and NO full model;
          * LoRA runs: adapter_model.bin + adapter_config.json + non_lora_trainables.bin (the layout of train.py:1708-1717);
          * always: fp32 master + AdamW moments of the trainable buffer and the trainer state.
        Written by rank 0 only (replicas are identical under data parallelism)."""
        if rank != 0:
            return
        import json
        from safetensors.torch import save_file
        eng = self.model.engine
        os.makedirs(path, exist_ok=True)
        if self._adapter_only() and not eng.lora:
            keys = ["mm_projector", "vision_resampler"] + (["embed_tokens", "embed_in"] if getattr(self.args, "use_im_start_end", False) else [])
            sd = {k: v.detach().clone().cpu() for k, v in eng.state_dict().items() if any(m in k for m in keys)}
            torch.save(sd, os.path.join(path, "mm_projector.bin"))
        else:
            self.model.save_pretrained(path)
            if not eng.lora:
                proj = {k: v.detach().clone().cpu() for k, v in eng.state_dict().items() if "mm_projector" in k}
                torch.save(proj, os.path.join(path, "mm_projector.bin"))
        if eng.master is not None and not getattr(self.args, "save_only_model", False):
            save_file({"master": eng.master.detach().cpu(), "exp_avg": eng.m.detach().cpu(), "exp_avg_sq": eng.vv.detach().cpu()},
                      os.path.join(path, "optimizer.safetensors"))
        with open(os.path.join(path, "trainer_state.json"), "w") as f:
            json.dump({"global_step": self.state["global_step"], "opt_step": eng.opt_step, "lora_step": eng.lora_step,
                       "log_history": self.state["log_history"]}, f)

    def _rotate_checkpoints(self, output_dir, rank=0):
        """--save_total_limit N (HF Trainer._rotate_checkpoints; finetune_radio_7b.sh:72 passes 1): keep the N newest checkpoint-* directories
        of output_dir, delete the older ones (13.5 + 81 GB each for the 7B model: an ignored limit fills the disk)."""
        limit = getattr(self.args, "save_total_limit", None)
        if rank != 0 or not limit or limit <= 0:
            return []
        import shutil
        cands = sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint-") and d[11:].isdigit()), key=lambda d: int(d[11:]))
        doomed = cands[:-int(limit)]
        for d in doomed:
            shutil.rmtree(os.path.join(output_dir, d), ignore_errors=True)
        return doomed

    def _maybe_resume(self, resume_from_checkpoint):
        """True -> newest checkpoint-* under output_dir (the reference's auto-resume, train.py:1699-1702); str -> that directory."""
        a = self.args
        path = resume_from_checkpoint
        if path is True:
            root = getattr(a, "output_dir", None)
            cands = [d for d in (os.listdir(root) if root and os.path.isdir(root) else []) if d.startswith("checkpoint-") and d[11:].isdigit()]
            path = os.path.join(root, max(cands, key=lambda d: int(d[11:]))) if cands else None
        if not path:
            return 0
        import json
        from safetensors.torch import load_file
        eng = self.model.engine
        if eng.lora:
            self.model.load_adapter(path)
        elif os.path.exists(os.path.join(path, "model.safetensors")):
            eng.load_state_dict(load_file(os.path.join(path, "model.safetensors")))
        else:       # projector-only checkpoint: the frozen parts are those the run started from
            eng.load_state_dict(torch.load(os.path.join(path, "mm_projector.bin"), map_location="cpu", weights_only=True))
        opt = os.path.join(path, "optimizer.safetensors")
        if os.path.exists(opt):
            eng.init_optimizer()
            st = load_file(opt)
            eng.master.copy_(st["master"]), eng.m.copy_(st["exp_avg"]), eng.vv.copy_(st["exp_avg_sq"])
        with open(os.path.join(path, "trainer_state.json")) as f:
            ts = json.load(f)
        eng.opt_step = ts["opt_step"] if os.path.exists(opt) else 0      # save_only_model checkpoints: a fresh optimizer, as HF restarts it
        eng.lora_step = ts.get("lora_step", eng.lora_step)      # the dropout masks of a resumed LoRA run continue their counter
        self.state["global_step"] = ts["global_step"]
        self.state["log_history"] = ts["log_history"]
        return ts["global_step"]
