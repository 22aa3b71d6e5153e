"""N>1 path on CPU: world_size-2 gloo run of the only exchange the path has (start-up weight
broadcast) plus stream sharding and the benchmark's max-over-ranks timer."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from infercam_onnx_amd import parallel, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.synthetic_weights() if rank == 0 else None
    got = parallel.broadcast_weights(w, dist)
    t = parallel.max_over_ranks(1.0 + rank, dist)
    streams = parallel.stream_for_rank(rank, world, 5)
    # every rank encodes its own stream's frames: distinct seeds, no exchange
    jpeg = synth.synth_jpeg_pool(rank, 1, 64, 48)[0]
    q.put((rank, float(got.sum()), got.size, t, streams, len(jpeg), hash(jpeg)))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    import torch.multiprocessing as mp
    from infercam_onnx_amd import parallel, synth

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = synth.synthetic_weights()
    for rank, s, n, t, streams, _, _ in res:
        assert n == ref.size and s == float(ref.sum())
        assert t == 2.0  # slowest rank
    assert res[0][4] == [0, 2, 4] and res[1][4] == [1, 3]
    assert res[0][6] != res[1][6]  # distinct streams
    assert parallel.aggregate_fps(320, 2, 2.0) == 320.0


def test_single_process_passthrough():
    from infercam_onnx_amd import parallel, synth

    w = synth.synthetic_weights()
    assert np.array_equal(parallel.broadcast_weights(w, None), w)
    assert parallel.max_over_ranks(1.5, None) == 1.5
    assert parallel.stream_for_rank(0, 1) == [0]
