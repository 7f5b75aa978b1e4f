"""The N>1 path on CPU: two gloo ranks shard a ragged utterance batch with no data-path collective;
the only collectives are the benchmark contract's barrier / MAX-over-ranks (and a SUM used to report)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

import cases


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)
    import gnuspeech_amd as g
    from gnuspeech_amd import shard
    ip = g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0))
    order = shard.sort_by_length(nframes)
    lo, hi = shard.shard_range(len(nframes), rank, world)
    mine = order[lo:hi]
    local = sum(shard.samples_for_frames(ip, int(nframes[v])) for v in mine)
    td.barrier()
    total = shard.sum_over_ranks(local)
    tmax = shard.max_over_ranks(1.0 + rank)          # rank 1 is "slower": MAX must report 2.0
    q.put((rank, lo, hi, [int(v) for v in mine], local, total, tmax))
    td.destroy_process_group()


def test_two_ranks_shard_without_collective():
    rng = np.random.default_rng(5)
    nframes = rng.integers(0, 400, size=257)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nframes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, v0, s0, t0, m0), (r1, lo1, hi1, v1, s1, t1, m1) = res
    assert (lo0, hi1) == (0, 257) and hi0 == lo1 and abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert sorted(v0 + v1) == list(range(257))                        # disjoint, complete
    import gnuspeech_amd as g
    from gnuspeech_amd import shard
    ip = g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0))
    whole = sum(shard.samples_for_frames(ip, int(n)) for n in nframes)
    assert t0 == t1 == s0 + s1 == whole                               # SUM of shard sizes = single-process total
    assert m0 == m1 == 2.0                                            # MAX over ranks
    # length-sorted order: a rank's voices are a contiguous slice of the descending-length order
    assert all(nframes[v0[i]] >= nframes[v0[i + 1]] for i in range(len(v0) - 1))
    assert nframes[v0[-1]] >= nframes[v1[0]]


def test_shard_range_properties():
    from gnuspeech_amd import shard
    for V in (0, 1, 7, 64, 65536, 65537):
        for W in (1, 2, 3, 8):
            rs = [shard.shard_range(V, r, W) for r in range(W)]
            assert rs[0][0] == 0 and rs[-1][1] == V
            assert all(rs[i][1] == rs[i + 1][0] for i in range(W - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_sample_count_without_device():
    import gnuspeech_amd as g
    from gnuspeech_amd import shard
    ip = g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0))
    assert shard.samples_for_frames(ip, 251) == 44159
    assert shard.samples_for_frames(ip, 0) == 0 and shard.samples_for_frames(ip, 1) == -((-26 * 65536) // 29350)
    d = shard.derive(ip)
    assert (d["controlPeriod"], d["sampleRate"], d["padSize"], d["timeRegisterIncrement"], d["firTaps"]) == (79, 19750, 13, 29350, 49)
    ip2 = g.TRMInputParameters.from_dict(cases.monet_default_params(22050.0))
    ip2.length = 10.0                                                  # down-sampling rates (golden short_tube_downsample)
    d2 = shard.derive(ip2)
    assert (d2["controlPeriod"], d2["sampleRate"], d2["padSize"], d2["timeRegisterIncrement"]) == (139, 34750, 21, 103282)
    assert shard.samples_for_frames(ip2, 50) == 4349
