"""The north-star world size on a one-GPU box: EIGHT ranks with the real engine, each holding a
12 500-row shard of range_db_large on cuda:0, through the body of the RCCL first-contact test
(tests/test_gpu_rccl.py::_body: C4 shape with ragged shares, chunked == unchunked pass 1, top-k with one
all-gather, the beta sweep, every ``row_shards`` divisor of 8 - 1x8, 2x4, 4x2 beside 8x1 -, the drop-in
call, the sharded ``save_embeddings``).  The pool's boxes admit 6 processes on a card, so the ranks are
threads of this process (tools/thread_ranks.py) and the collectives plain device copies; the shapes, the
chunk schedule, the buffers and the order of the collectives are those of an 8-GPU run."""
import pytest
import torch

from tools.thread_ranks import run_rank_threads, threaded_backend_available
from test_gpu_rccl import H, L, SEED, _bank_arrays, _body

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not threaded_backend_available(), reason="torch's threaded process group is not in this build")
def test_world8_rank_threads_at_c4_c5_shapes(tmp_path):
    from range_amd.bank import prepare_bank
    from range_amd.bankfile import write_bankfile
    from tools import synth
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=H, seed=SEED)
    rbank = write_bankfile(str(tmp_path / "large.rbank"), prepare_bank(*_bank_arrays()))
    dev = torch.device("cuda", 0)

    def rank_fn(rank, world):
        torch.cuda.set_device(dev)
        _body(rank, world, dev, ck, rbank, str(tmp_path), "threaded", total_queries=100_000)

    res = run_rank_threads(8, rank_fn)
    assert res == {r: "ok" for r in range(8)}, res
