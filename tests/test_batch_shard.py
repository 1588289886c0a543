"""Utterance-batch data parallelism (puresound_amd/batch_shard.py) on CPU: world_size-2 gloo processes.
The compute inside each rank is the CPU oracle here (tests may use it); on GPUs it is the HIP path and the
collective is RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from detweights import det_state_dict, det_wave
from oracle import separator_oracle as O
from puresound_amd.batch_shard import OverlappedGather, gather_utterances, shard_bounds, sharded_inference
import puresound_amd.nnet as PA


def test_shard_bounds_cover_the_batch_exactly():
    for batch in (1, 2, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, results, name="tiny_free", length=900):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = det_state_dict(cases.build(PA.NS, name))
        cfg = cases.oracle_cfg(name)
        noisy = det_wave(21, batch, length)

        def infer(x):
            return O.inference(x, sd, cfg)

        out = sharded_inference(infer, noisy)
        lo, hi = shard_bounds(batch, world, rank)
        # ragged explicit gather as well
        again = gather_utterances(infer(noisy[lo:hi]), batch)
        full = infer(noisy)
        results[rank] = (bool(torch.allclose(out, full, atol=1e-6)), bool(torch.equal(out, again)), tuple(out.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch", [4, 5])
def test_sharded_inference_two_ranks_gloo(batch):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, batch, results), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        same_as_unsharded, deterministic, shape = results[r]
        assert same_as_unsharded and deterministic and shape[0] == batch


@pytest.mark.parametrize("batch", [4, 3])
def test_config4_dprnn_sharded_two_ranks_gloo(batch):
    """BASELINE config 4 (the DPRNN separator, data-parallel): the split tools/bench_recurrent.py --gpus N times, with the
    oracle as each rank's compute -- equal and ragged shards reassemble to the unsharded result."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, port, batch, results, "cfg4_short", 1300), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        same_as_unsharded, deterministic, shape = results[r]
        assert same_as_unsharded and deterministic and shape[0] == batch


def _overlap_worker(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, width, steps = 6, 50, 5
        og = OverlappedGather(batch)
        lo, hi = shard_bounds(batch, world, rank)
        fulls = [torch.arange(batch * width, dtype=torch.float32).reshape(batch, width) * (s + 1) + s for s in range(steps)]
        ok = True
        for s in range(steps):
            prev = og.submit(fulls[s][lo:hi].clone())
            ok = ok and ((prev is None) if s == 0 else bool(torch.equal(prev, fulls[s - 1])))
        last = og.flush()
        ok = ok and bool(torch.equal(last, fulls[-1])) and og.flush() is last
        try:
            OverlappedGather(batch + 1)
            ragged_refused = False
        except ValueError:
            ragged_refused = True
        results[rank] = (ok, ragged_refused)
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_two_ranks_gloo():
    """The asynchronous, double-buffered all-gather bench.py --gpus N uses: step i's result is handed out at step i+1,
    buffers alternate, flush() completes everything; ragged batches are refused (they take gather_utterances)."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, port, results), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        assert results[r] == (True, True)
