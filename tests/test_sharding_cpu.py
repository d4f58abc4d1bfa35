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


def _oracle_executor(st, ku):
    def execute(desc, tables):
        cols, n, _ = st.execute(desc, tables=tables)
        return [np.ascontiguousarray(c[:n]) for c in cols]
    return execute


def _batch_worker(rank, world, port, q):
    """bench.py's N > 1 step with the oracle as executor and gloo as the wire: phase A over the local shard -> all-gatherv of
    C (sized from the row counts) -> phase B over the local shard (sharding.run_q5_batch_sharded_tables)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
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
        ex = sharding.NumpyExchange(dist, world, rank)
        seen = {}

        def allgatherv(cols):
            out = ex.allgatherv(cols)
            seen["local"], seen["all"] = len(cols[0]), len(out[0])
            return out
        cols = sharding.run_q5_batch_sharded_tables(ds, params, _oracle_executor(st, ku), allgatherv)
        q.put((rank, len(cols[0]), ku.multiset(cols), seen["local"], seen["all"]))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, timeout=300):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results, key=lambda r: r[0])


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_q5_batch_exchange_equals_unsharded(world):
    from rdf_fusion_amd import bsbm
    from oracle import oracle as orc
    import kat_util as ku
    results = _spawn(_batch_worker, world)
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
    # the all-gatherv delivered exactly the sum of the ranks' rows to every rank: nothing padded, nothing clipped
    total = sum(r[3] for r in results)
    assert total > 0 and all(r[4] == total for r in results)


def _hybrid_worker(rank, world, port, q):
    """bench.py's N > 1 step over shard_dataset_hybrid's layout: phase A on the subject shard (default graph), C re-sharded by
    prodFeature (hash repartition), phase B on the object-sharded candidate triples + the replicated star predicates."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from rdf_fusion_amd import bsbm, sharding
    from oracle import oracle as orc
    import kat_util as ku
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = bsbm.generate(600)
        g, s, p, o = sharding.shard_dataset_hybrid(ds, rank, world)
        st = orc.OracleStore()
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values, ds.decimals)
        batch = _batch_products(ds)
        params = [np.arange(1, BATCH + 1, dtype=np.uint32), batch]
        ex = sharding.NumpyExchange(dist, world, rank)
        seen = {}

        def repartition(cols, key_col):
            out = ex.repartition(cols, key_col)
            seen["sent"], seen["received"] = len(cols[0]), len(out[0])
            assert np.all(sharding.shard_of(out[key_col], world) == rank)       # a rank only receives the features it holds
            return out
        cols = sharding.run_q5_batch_hybrid(ds, params, _oracle_executor(st, ku), repartition)
        q.put((rank, len(cols[0]), ku.multiset(cols), seen["sent"], seen["received"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_q5_batch_hybrid_layout_equals_unsharded(world):
    """Both sides of the candidate join sharded (C re-partitioned by feature; the productFeature triples by object): the
    union of the ranks' bindings is the unsharded answer, and no row of C is lost or duplicated by the repartition."""
    from rdf_fusion_amd import bsbm
    from oracle import oracle as orc
    import kat_util as ku
    results = _spawn(_hybrid_worker, world)
    ds = bsbm.generate(600)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values, ds.decimals)
    batch = _batch_products(ds)
    params = [np.arange(1, BATCH + 1, dtype=np.uint32), batch]
    cols, n, _ = st.execute(bsbm.q5_batch_plan(ds), tables=[params])
    got = ku.multiset(list(np.concatenate([r[2] for r in results]).T))
    assert sum(r[1] for r in results) == n > 0
    np.testing.assert_array_equal(got, ku.multiset(cols, n))
    _, n_c, _ = st.execute(bsbm.q5_batch_const_plan(ds), tables=[params])
    assert sum(r[3] for r in results) == sum(r[4] for r in results) == n_c


# --------------------------------------------------------------------------- LUBM Q9: joins whose key is not the shard key
def _q9_worker(rank, world, port, q):
    """lubm.q9_sharded_stages with the oracle as executor: three hash repartitions (by advisor, by course, by student)
    between four local plans; the triangle cannot be answered shard-locally."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from rdf_fusion_amd import lubm, sharding
    from oracle import oracle as orc
    import kat_util as ku
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ds = lubm.generate(2)
        g, s, p, o = sharding.shard_dataset(ds, rank, world)
        st = orc.OracleStore()
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values)
        st.set_strings(ds.str_offsets, ds.str_heap)
        ex = sharding.NumpyExchange(dist, world, rank)
        moved = []

        def repartition(cols, key_col):
            out = ex.repartition(cols, key_col)
            assert (sharding.shard_of(out[key_col], world) == rank).all()       # every row is where its key lives
            moved.append((len(cols[0]), len(out[0])))
            return out
        cols = sharding.run_stages(lubm.q9_sharded_stages(ds, "^GraduateStudent1", ""), _oracle_executor(st, ku), repartition)
        q.put((rank, len(cols[0]), ku.multiset(cols), moved))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lubm_q9_with_repartitions_equals_unsharded(world):
    from rdf_fusion_amd import lubm
    from oracle import oracle as orc
    import kat_util as ku
    results = _spawn(_q9_worker, world)
    ds = lubm.generate(2)
    st = orc.OracleStore()
    st.extend(ds.g, ds.s, ds.p, ds.o)
    st.set_typed_values(ds.typed_values)
    st.set_strings(ds.str_offsets, ds.str_heap)
    cols, n, _ = st.execute(lubm.q9_optional_regex_plan(ds, "^GraduateStudent1", ""))
    expected = ku.multiset(cols, n)
    got = ku.multiset(list(np.concatenate([r[2] for r in results]).T))
    assert sum(r[1] for r in results) == n > 0
    np.testing.assert_array_equal(got, expected)
    # a repartition moves rows, it neither drops nor invents any: per step, rows sent over all ranks == rows received
    for step in range(3):
        assert sum(r[3][step][0] for r in results) == sum(r[3][step][1] for r in results) > 0
