"""Host input pipeline: the threaded prefetcher yields exactly the synchronous loop's batches, in order."""
import time

import pytest
import torch

from radvlm_amd.llava.train.llava_trainer import BatchPrefetcher


class _DS:
    def __init__(self, n, fail_at=None):
        self.n, self.fail_at = n, fail_at

    def __getitem__(self, i):
        if i == self.fail_at:
            raise ValueError(f"bad sample {i}")
        time.sleep(0.002 * (i % 3))                      # uneven per-sample latency: completion order != submission order
        return {"id": i, "image": torch.full((2, 2), float(i))}


def _collate(samples):
    return {"ids": [s["id"] for s in samples], "images": [s["image"] for s in samples]}


def _batches(n, bs):
    return [list(range(i, i + bs)) for i in range(0, n, bs)]


@pytest.mark.parametrize("workers", [0, 1, 4])
def test_prefetcher_matches_synchronous_order(workers):
    ds = _DS(40)
    want = [_collate([ds[i] for i in idx]) for idx in _batches(40, 4)]
    p = BatchPrefetcher(ds, _collate, _batches(40, 4), num_workers=workers, depth=2, pin=False)
    got = list(p)
    p.close()
    assert [g["ids"] for g in got] == [w["ids"] for w in want]
    assert all(torch.equal(a, b) for g, w in zip(got, want) for a, b in zip(g["images"], w["images"]))


def test_prefetcher_surfaces_worker_errors_and_can_stop_early():
    p = BatchPrefetcher(_DS(40, fail_at=9), _collate, _batches(40, 4), num_workers=3, depth=2, pin=False)
    assert next(p)["ids"] == [0, 1, 2, 3] and next(p)["ids"] == [4, 5, 6, 7]
    with pytest.raises(ValueError, match="bad sample 9"):
        next(p)
    p.close()
    q = BatchPrefetcher(_DS(4000), _collate, _batches(4000, 4), num_workers=2, depth=2, pin=False)
    assert next(q)["ids"] == [0, 1, 2, 3]
    q.close()                                            # abandoning a long stream must not hang
