"""The N > 1 path on CPU: gloo ranks, triples sharded by hash(subject), the constant-subject bindings
all-gathered, the rest of Q5 local (rdf-fusion_amd/sharding.py).  The executor here is the CPU oracle
(the HIP library cannot run without a GPU); what is under test is the sharding + exchange logic: the
union of the ranks' bindings must equal the unsharded answer."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRODUCTS = (3, 77, 401, 599)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from rdf_fusion_amd import bsbm, sharding
    from oracle import oracle as orc
    import kat_util as ku
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = bsbm.generate(600)
        g, s, p, o = sharding.shard_dataset(ds, rank, world)
        st = orc.OracleStore()
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values, ds.decimals)
        products = [ds.product(i) for i in PRODUCTS]
        rows = []

        def run_const(desc):
            cols, n, _ = st.execute(desc)
            return cols[0]

        def run_local(desc, tables):
            if any(len(t) == 0 for t in tables):
                return 0
            cols, n, _ = st.execute(desc, tables=[[t] for t in tables])
            rows.append(ku.multiset(cols, n))
            return n

        def all_gather(recs):
            mine = torch.from_numpy(recs)
            out = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(out, mine)
            return torch.stack(out).numpy()

        n = sharding.run_q5_batch_sharded(ds, products, run_const, run_local, all_gather)
        mine = np.concatenate(rows) if rows else np.zeros((0, 2), np.uint32)
        q.put((rank, n, mine))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_q5_equals_unsharded(world):
    import torch.multiprocessing as mp
    from rdf_fusion_amd import bsbm, sharding
    from oracle import oracle as orc
    import kat_util as ku
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ds = bsbm.generate(600)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values, ds.decimals)
    expected = []
    for i in PRODUCTS:
        cols, n, _ = st.execute(bsbm.q5_plan(ds, ds.product(i)))
        expected.append(ku.multiset(cols, n))
    expected = np.concatenate(expected)
    expected = ku.multiset(list(expected.T))
    got = np.concatenate([r[2] for r in results])
    got = ku.multiset(list(got.T))
    assert sum(r[1] for r in results) == len(expected) > 0
    np.testing.assert_array_equal(got, expected)
    # shards are disjoint and cover the dataset
    sizes = [len(sharding.shard_dataset(ds, r, world)[0]) for r in range(world)]
    assert sum(sizes) == ds.n_triples and min(sizes) > 0.5 * ds.n_triples / world


def test_exchange_record_roundtrip():
    from rdf_fusion_amd import sharding
    rec0 = sharding.pack_record([], [], [])
    rec1 = sharding.pack_record([5, 9, 4000000000], [77], [88])
    out = sharding.unpack_records(np.stack([np.stack([rec0, rec1]), np.stack([rec1, rec0])]))
    assert out[0][0].tolist() == [5, 9, 4000000000] and out[0][1].tolist() == [77] and out[0][2].tolist() == [88]
    assert out[1][0].tolist() == [5, 9, 4000000000]


# --------------------------------------------------------------------------- the batched flow bench.py --gpus N runs
BATCH = 96          # instances per batch: products drawn WITH repetition, so instances sharing a product are covered


def _batch_products(ds):
    rng = np.random.default_rng(4)
    return np.array([ds.product(int(i)) for i in rng.integers(0, ds.n_products, BATCH)], dtype=np.uint32)


def _batch_worker(rank, world, port, q):
    """bench.py's N > 1 step with the oracle as executor: phase A (three constant-subject plans over the local shard,
    whole batch, one plan) -> BatchExchange.pack -> ONE all_gather_into_tensor -> unpack -> phase B over the local shard."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from rdf_fusion_amd import bsbm, sharding
    from oracle import oracle as orc
    import kat_util as ku
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = bsbm.generate(600)
        g, s, p, o = sharding.shard_dataset(ds, rank, world)
        st = orc.OracleStore()
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values, ds.decimals)
        batch = _batch_products(ds)
        params = [np.arange(1, BATCH + 1, dtype=np.uint32), batch]
        ex = sharding.BatchExchange(BATCH, world)
        mine = torch.zeros(ex.buf_len, dtype=torch.int32)
        cols, n, _ = st.execute(bsbm.q5_batch_const_plan(ds), tables=[params])
        ex.pack(mine, [torch.from_numpy(np.ascontiguousarray(c[:n]).view(np.int32)) for c in cols], n)
        out = torch.empty(world * ex.buf_len, dtype=torch.int32)
        dist.all_gather_into_tensor(out, mine)
        tab = [c.numpy().view(np.uint32) for c in ex.unpack(out)]
        assert len(tab) == 5 and all(len(c) == world * ex.cap for c in tab)
        cols, n, _ = st.execute(bsbm.q5_batch_plan(ds, tables=True), tables=[tab])
        q.put((rank, n, ku.multiset(cols, n), int((tab[0] != 0).sum())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_q5_batch_exchange_equals_unsharded(world):
    import torch.multiprocessing as mp
    from rdf_fusion_amd import bsbm
    from oracle import oracle as orc
    import kat_util as ku
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ds = bsbm.generate(600)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values, ds.decimals)
    batch = _batch_products(ds)
    cols, n, _ = st.execute(bsbm.q5_batch_plan(ds), tables=[[np.arange(1, BATCH + 1, dtype=np.uint32), batch]])
    expected = ku.multiset(cols, n)
    got = np.concatenate([r[2] for r in results])
    got = ku.multiset(list(got.T))
    assert sum(r[1] for r in results) == n > 0
    np.testing.assert_array_equal(got, expected)
    # every rank saw the same gathered tables; the padding rows (inst = 0) joined with nothing
    assert all(r[3] == results[0][3] for r in results) and results[0][3] > 0


def test_batch_exchange_layout_and_overflow():
    import torch
    from rdf_fusion_amd import sharding
    ex = sharding.BatchExchange(1000, 4)
    assert ex.inst_cap == 531 and ex.cap == 531 * 28 and ex.buf_len == 5 * ex.cap
    big = sharding.BatchExchange(262144, 8)
    assert big.inst_cap == 36300 and big.cap == 36300 * 21                    # large batches: mean fan-out + margin, not the worst case
    assert sharding.BatchExchange(10, 4).inst_cap == 10                       # never more than the batch itself
    bufs = []
    for r in range(4):
        b = torch.zeros(ex.buf_len, dtype=torch.int32)
        rows = 3 * r                                                           # ragged, including an empty table
        ex.pack(b, [torch.full((rows,), 100 * r + k + 1, dtype=torch.int32) for k in range(5)], rows)
        bufs.append(b)
    t = ex.unpack(torch.cat(bufs))
    assert t.shape == (5, 4 * ex.cap) and t.is_contiguous()
    for r in range(4):
        seg = t[:, r * ex.cap:(r + 1) * ex.cap]
        for k in range(5):
            assert seg[k, :3 * r].tolist() == [100 * r + k + 1] * (3 * r) and int(seg[k, 3 * r:].abs().sum()) == 0
    with pytest.raises(RuntimeError, match="exchange buffer too small"):
        ex.pack(torch.zeros(ex.buf_len, dtype=torch.int32), [torch.zeros(ex.cap + 1, dtype=torch.int32)] * 5, ex.cap + 1)
