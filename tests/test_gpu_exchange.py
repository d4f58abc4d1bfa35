"""The multi-GPU exchange steps behind the C ABI (include/rdfgpu.h section 6) on ONE GPU: RCCL with a one-rank communicator,
and — several ranks on the same GPU, where RCCL refuses to run — the host-staged transport with an in-process wire, so
that the device-side partitioning / packing / unpacking and the staged sharded plans run with world > 1.  The xGMI wire
itself needs the 8-GPU node (bench.py --gpus N)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import rdf_fusion_amd as rf
from rdf_fusion_amd import bsbm, lubm, sharding
from oracle import oracle as orc
import kat_util as ku


def on_device(torch, cols):
    ts = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.uint32).view(np.int32)).cuda() for c in cols]
    return ts, [t.data_ptr() for t in ts]


def from_device(torch, ptrs, rows):
    out = []
    for p in ptrs:
        class Col:
            __cuda_array_interface__ = {"shape": (rows,), "typestr": "<i4", "data": (int(p), False), "version": 2}
        out.append(torch.as_tensor(Col(), device="cuda").cpu().numpy().view(np.uint32).copy() if rows else np.zeros(0, np.uint32))
    return out


class Wire:
    """An all-to-all between `world` threads of this process: what gloo / MPI / RCCL does between processes."""

    def __init__(self, world):
        self.world = world
        self.box = [[None] * world for _ in range(world)]
        self.barrier = threading.Barrier(world)

    def fn(self, rank):
        def alltoallv(blocks):
            for d in range(self.world):
                self.box[rank][d] = blocks[d]
            self.barrier.wait()
            got = [self.box[s][rank] for s in range(self.world)]
            self.barrier.wait()
            return got
        return alltoallv


def run_ranks(world, body):
    """body(rank, comm) on `world` threads, each with its own host-transport communicator on GPU 0."""
    wire = Wire(world)
    results, errors = [None] * world, []

    def main(rank):
        try:
            comm = rf.Comm(rank, world, device=0, host_alltoallv=wire.fn(rank))
            try:
                results[rank] = body(rank, comm)
            finally:
                comm.close()
        except BaseException as e:      # noqa: BLE001 — a dead rank must not leave the others at the barrier
            errors.append(e)
            wire.barrier.abort()
    ts = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    if errors:
        raise errors[0]
    return results


def test_rccl_single_rank_communicator(torch_cuda):
    """ncclCommInitRank with one rank: the RCCL transport's calls (count all-gather, grouped send / recv to self) run."""
    rng = np.random.default_rng(0)
    cols = [rng.integers(0, 1 << 32, 5000, dtype=np.uint64).astype(np.uint32) for _ in range(3)]
    keep, ptrs = on_device(torch_cuda, cols)
    comm = rf.Comm(0, 1, device=0, unique_id=rf.Comm.unique_id())
    out, rows = comm.allgatherv(ptrs, 5000)
    assert rows == 5000
    for a, b in zip(from_device(torch_cuda, out, rows), cols):
        np.testing.assert_array_equal(a, b)
    out, rows = comm.repartition(ptrs, 5000, 1)
    np.testing.assert_array_equal(ku.multiset(from_device(torch_cuda, out, rows)), ku.multiset(cols))
    out, rows = comm.allgatherv(ptrs, 0)           # an empty contribution
    assert rows == 0
    comm.close()
    del keep


@pytest.mark.parametrize("world", [2, 3, 5])
def test_exchange_steps_between_ranks_on_one_gpu(torch_cuda, world):
    rng = np.random.default_rng(world)
    tables = [[rng.integers(1, 50_000, n).astype(np.uint32) for _ in range(3)] for n in [rng.integers(0, 40_000) if r else 0 for r in range(world)]]   # rank 0 contributes nothing

    def body(rank, comm):
        keep, ptrs = on_device(torch_cuda, tables[rank])
        n = len(tables[rank][0])
        out, rows = comm.allgatherv(ptrs, n)
        gathered = from_device(torch_cuda, out, rows)
        out, rows = comm.repartition(ptrs, n, 2)
        mine = from_device(torch_cuda, out, rows)
        del keep
        return gathered, mine
    res = run_ranks(world, body)
    everything = [np.concatenate([t[k] for t in tables]) for k in range(3)]
    for rank, (gathered, mine) in enumerate(res):
        for a, b in zip(gathered, everything):
            np.testing.assert_array_equal(a, b)                                  # rank order, nothing padded
        assert (sharding.shard_of(mine[2], world) == rank).all()                # every row where its key lives
    moved = [np.concatenate([r[1][k] for r in res]) for k in range(3)]
    np.testing.assert_array_equal(ku.multiset(moved), ku.multiset(everything))  # moved, not lost, not invented
    # the repartition is stable: a rank receives, source rank after source rank, that rank's rows for it in their order
    for rank, (_, mine) in enumerate(res):
        want = [np.concatenate([t[k][sharding.shard_of(t[2], world) == rank] for t in tables]) for k in range(3)]
        for a, b in zip(mine, want):
            np.testing.assert_array_equal(a, b)


def test_repartition_is_stable_at_scale(torch_cuda):
    """3 M rows sorted by the key column, sent by one of four ranks: rows of every destination keep their order, so a sorted
    table leaves (and arrives) as sorted blocks — what the band join's partition pass counts on after a repartition."""
    rng = np.random.default_rng(4)
    n = 3_000_017
    key = np.sort(rng.integers(1, 60_000, n).astype(np.uint32))
    cols = [np.arange(n, dtype=np.uint32), rng.integers(0, 1 << 31, n).astype(np.uint32), key]

    def body(rank, comm):
        keep, ptrs = on_device(torch_cuda, cols if rank == 0 else [c[:0] for c in cols])
        out, rows = comm.repartition(ptrs, n if rank == 0 else 0, 2)
        got = from_device(torch_cuda, out, rows)
        del keep
        return got
    world = 4
    res = run_ranks(world, body)
    for rank, got in enumerate(res):
        sel = sharding.shard_of(key, world) == rank
        for a, b in zip(got, cols):
            np.testing.assert_array_equal(a, b[sel])
        assert (np.diff(got[2].astype(np.int64)) >= 0).all()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lubm_q9_on_device_equals_oracle(torch_cuda, world):
    """LUBM Q9 + REGEX + OPTIONAL over `world` graph shards on one GPU: four local plans chained through three
    rdfgpu_exchange_repartition steps, result columns handed from plan to exchange to plan in HBM."""
    ds = lubm.generate(3)
    full = orc.OracleStore()
    full.extend(ds.g, ds.s, ds.p, ds.o)
    full.set_typed_values(ds.typed_values); full.set_strings(ds.str_offsets, ds.str_heap)
    exp, n_exp, _ = full.execute(lubm.q9_optional_regex_plan(ds, "^GraduateStudent1", ""))

    def body(rank, comm):
        g, s, p, o = sharding.shard_dataset(ds, rank, world)
        st = rf.GpuQuadStore(device=0)
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values); st.set_strings(ds.str_offsets, ds.str_heap)
        plans = []

        def execute(desc, tabs):
            plan = st.plan(desc)
            for slot, (ptrs, rows) in enumerate(tabs):
                plan.bind_table(slot, ptrs, rows)
            plan.execute()
            plans.append(plan)                                                   # its result columns stay alive
            return plan.result_device()

        def repartition(tab, key_col):
            return comm.repartition(tab[0], tab[1], key_col)
        ptrs, rows = sharding.run_stages(lubm.q9_sharded_stages(ds, "^GraduateStudent1", ""), execute, repartition)
        out = from_device(torch_cuda, ptrs, rows)
        for pl in plans:
            pl.close()
        st.close()
        return out
    res = run_ranks(world, body)
    got = [np.concatenate([r[k] for r in res]) for k in range(5)]
    assert len(got[0]) == n_exp > 0
    np.testing.assert_array_equal(ku.multiset(got), ku.multiset(exp, n_exp))


def test_sharded_q5_batch_on_device_equals_oracle(torch_cuda):
    """bench.py --gpus N's step on 3 shards of one GPU: phase A, rdfgpu_exchange_allgatherv of C, phase B."""
    world = 3
    ds = bsbm.generate(2500)
    full = orc.OracleStore()
    full.extend(ds.g, ds.s, ds.p, ds.o)
    full.set_typed_values(ds.typed_values, ds.decimals)
    rng = np.random.default_rng(8)
    batch = 400
    prods = np.array([ds.product(int(i)) for i in rng.choice(ds.n_products, batch, replace=False)], dtype=np.uint32)
    params = [np.arange(1, batch + 1, dtype=np.uint32), prods]
    exp, n_exp, _ = full.execute(bsbm.q5_batch_plan(ds), [params])

    def body(rank, comm):
        g, s, p, o = sharding.shard_dataset(ds, rank, world)
        st = rf.GpuQuadStore(device=0)
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values, ds.decimals)
        keep, pp = on_device(torch_cuda, params)
        plans = []

        def execute(desc, tabs):
            plan = st.plan(desc)
            for slot, (ptrs, rows) in enumerate(tabs):
                plan.bind_table(slot, ptrs, rows)
            plan.execute()
            plans.append(plan)
            return plan.result_device()
        out = None
        for rep in range(3):                       # re-executions: speculation, cached tables, the fused chain / band join
            ptrs, rows = sharding.run_q5_batch_sharded_tables(ds, (pp, batch), execute, lambda t: comm.allgatherv(t[0], t[1]))
            out = from_device(torch_cuda, ptrs, rows)
        del keep
        return out
    res = run_ranks(world, body)
    got = [np.concatenate([r[k] for r in res]) for k in range(3)]
    assert len(got[0]) == n_exp > 0
    np.testing.assert_array_equal(ku.multiset(got), ku.multiset(exp, n_exp))


def test_sharded_q5_batch_hybrid_layout_on_device_equals_oracle(torch_cuda):
    """bench.py --gpus N's step as it runs now, on 3 shards of one GPU: phase A on the subject shard, C re-sharded by
    prodFeature (rdfgpu_exchange_repartition), phase B on the object-sharded candidate triples in their named graph."""
    world = 3
    ds = bsbm.generate(2500)
    full = orc.OracleStore()
    full.extend(ds.g, ds.s, ds.p, ds.o)
    full.set_typed_values(ds.typed_values, ds.decimals)
    rng = np.random.default_rng(9)
    batch = 500
    prods = np.array([ds.product(int(i)) for i in rng.integers(0, ds.n_products, batch)], dtype=np.uint32)   # with repeats
    params = [np.arange(1, batch + 1, dtype=np.uint32), prods]
    exp, n_exp, _ = full.execute(bsbm.q5_batch_plan(ds), [params])

    def body(rank, comm):
        g, s, p, o = sharding.shard_dataset_hybrid(ds, rank, world)
        st = rf.GpuQuadStore(device=0)
        st.extend(g, s, p, o)
        st.set_typed_values(ds.typed_values, ds.decimals)
        keep, pp = on_device(torch_cuda, params)
        plans, calls = {}, [0]

        def execute(desc, tabs):
            stage = calls[0] % 2                                  # phase A, phase B, phase A, ...
            calls[0] += 1
            plan = plans.get(stage)
            if plan is None:
                plan = plans[stage] = st.plan(desc)               # compiled once per stage: re-executions speculate and fuse
            for slot, (ptrs, rows) in enumerate(tabs):
                plan.bind_table(slot, ptrs, rows)
            plan.execute()
            return plan.result_device()
        out = None
        for rep in range(3):
            ptrs, rows = sharding.run_q5_batch_hybrid(ds, (pp, batch), execute, lambda t, key_col: comm.repartition(t[0], t[1], key_col))
            out = from_device(torch_cuda, ptrs, rows)
        del keep
        return out
    res = run_ranks(world, body)
    got = [np.concatenate([r[k] for r in res]) for k in range(3)]
    assert len(got[0]) == n_exp > 0
    np.testing.assert_array_equal(ku.multiset(got), ku.multiset(exp, n_exp))
