"""ctypes binding of ``lib/librdfgpu.so`` — the product path.

There is no fallback of any kind in here: if the HIP library is missing or no gfx950 device is
usable, every call that touches data raises ``RdfGpuError``.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))


class RdfGpuError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rdfgpu status {status}: {message}")
        self.status = status


def library_path():
    return os.path.join(_HERE, "lib", "librdfgpu.so")


def kernel_source_sha16():
    """First 16 hex digits of the SHA-256 over the device sources (csrc/*.hip, *.hpp, in name order): what a committed rocprofv3
    counter summary is stamped with (profiles/summarize.py) and what bench.py compares before it quotes that summary as the HBM
    traffic of the kernels it has just timed — counters measured on other kernel code are not quoted."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.hpp"))):
        h.update(os.path.basename(f).encode()); h.update(b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_library():
    """Loads the in-tree HIP library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RdfGpuError(abi.ERR_NO_DEVICE,
                          f"{path} is missing: build it with `python __graft_entry__.py build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as
    # /opt/rocm's).  If ours were mapped first, torch would later map a second runtime and report no
    # device.  Importing torch first makes the dynamic loader resolve librdfgpu.so's libamdhip64
    # dependency to the copy already in the process.  torch is plumbing here (streams / RCCL), not a
    # dependency of the library: without torch installed, /opt/rocm's runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    u32p, u64p, vp = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.c_void_p
    lib.rdfgpu_last_error.restype = C.c_char_p
    lib.rdfgpu_abi_version.restype = C.c_uint32
    lib.rdfgpu_store_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
    lib.rdfgpu_store_destroy.argtypes = [vp]
    lib.rdfgpu_store_destroy.restype = None
    lib.rdfgpu_store_extend.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, u64p]
    lib.rdfgpu_store_extend_device.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, u64p]
    lib.rdfgpu_store_remove.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, u64p]
    lib.rdfgpu_store_clear.argtypes = [vp]
    lib.rdfgpu_store_drop_tables.argtypes = [vp]
    lib.rdfgpu_store_remove_graph.argtypes = [vp, C.c_uint32, u64p]
    lib.rdfgpu_store_len.argtypes = [vp, u64p]
    lib.rdfgpu_store_set_typed_values.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
    lib.rdfgpu_store_set_strings.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
    lib.rdfgpu_store_read_index.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.c_uint64, u64p]
    lib.rdfgpu_plan_compile.argtypes = [vp, C.POINTER(abi.PlanDesc), C.POINTER(vp)]
    lib.rdfgpu_plan_destroy.argtypes = [vp]
    lib.rdfgpu_plan_destroy.restype = None
    lib.rdfgpu_plan_bind_table.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.c_uint32, C.c_uint64]
    lib.rdfgpu_plan_execute.argtypes = [vp]
    lib.rdfgpu_plan_result_info.argtypes = [vp, u64p, u32p]
    lib.rdfgpu_plan_result_device.argtypes = [vp, C.POINTER(vp), C.c_uint32]
    lib.rdfgpu_plan_fetch.argtypes = [vp, C.POINTER(vp), C.c_uint32]
    lib.rdfgpu_plan_next.argtypes = [vp, C.POINTER(abi.ArrowArray), C.POINTER(abi.ArrowSchema)]
    lib.rdfgpu_plan_rewind.argtypes = [vp]
    lib.rdfgpu_ntriples_parse.argtypes = [C.c_int32, C.c_char_p, C.c_uint64, C.c_uint32, C.POINTER(vp)]
    lib.rdfgpu_ntriples_info.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.rdfgpu_ntriples_terms.argtypes = [vp, C.c_void_p, C.c_void_p]
    lib.rdfgpu_ntriples_decoded_info.argtypes = [vp, u64p, u64p]
    lib.rdfgpu_ntriples_decoded.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.rdfgpu_ntriples_columns.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.rdfgpu_ntriples_destroy.argtypes = [vp]
    lib.rdfgpu_ntriples_destroy.restype = None
    lib.rdfgpu_plan_decode_terms.argtypes = [vp, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(abi.ArrowArray), C.POINTER(abi.ArrowSchema)]
    lib.rdfgpu_plan_metrics.argtypes = [vp, C.POINTER(abi.Metrics)]
    lib.rdfgpu_plan_selected_index.argtypes = [vp, C.c_uint32, u32p]
    lib.rdfgpu_plan_stream.argtypes = [vp, C.POINTER(vp)]
    lib.rdfgpu_plan_enable_kernel_timing.argtypes = [vp, C.c_int]
    lib.rdfgpu_plan_kernel_stats.argtypes = [vp, C.POINTER(abi.KernelStat), C.c_uint32, u32p]
    lib.rdfgpu_plan_pushdown_filters.argtypes = [vp, C.c_uint32, C.POINTER(abi.PushdownFilter), C.c_uint32, C.POINTER(C.c_uint8)]
    lib.rdfgpu_plan_set_dynamic_filters.argtypes = [vp, C.c_uint32, C.POINTER(abi.PushdownFilter), C.c_uint32]
    lib.rdfgpu_plan_source_predicate.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(abi.Predicate)]
    lib.rdfgpu_store_set_option.argtypes = [vp, C.c_uint32, C.c_uint64]
    lib.rdfgpu_store_get_option.argtypes = [vp, C.c_uint32, u64p]
    lib.rdfgpu_plan_set_option.argtypes = [vp, C.c_uint32, C.c_uint64]
    lib.rdfgpu_option_name.argtypes = [C.c_uint32]
    lib.rdfgpu_option_name.restype = C.c_char_p
    lib.rdfgpu_comm_unique_id.argtypes = [C.c_void_p]
    lib.rdfgpu_comm_create.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(vp)]
    lib.rdfgpu_comm_create_host.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, abi.HOST_ALLTOALLV_FN, C.c_void_p, C.POINTER(vp)]
    lib.rdfgpu_comm_destroy.argtypes = [vp]
    lib.rdfgpu_comm_destroy.restype = None
    lib.rdfgpu_exchange_allgatherv.argtypes = [vp, C.POINTER(vp), C.c_uint32, C.c_uint64, C.POINTER(vp), u64p]
    lib.rdfgpu_exchange_repartition.argtypes = [vp, C.POINTER(vp), C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(vp), u64p]
    lib.rdfgpu_shard_of.argtypes = [C.c_uint32, C.c_uint32]
    lib.rdfgpu_shard_of.restype = C.c_uint32
    lib.rdfgpu_scan_score.argtypes = [C.POINTER(abi.ScanInstruction)]
    lib.rdfgpu_scan_score.restype = C.c_uint64
    lib.rdfgpu_choose_index.argtypes = [C.POINTER(abi.ScanInstruction), C.c_uint32]
    lib.rdfgpu_choose_index.restype = C.c_uint32
    lib.rdfgpu_predicate_and.argtypes = [C.POINTER(abi.Predicate), C.POINTER(abi.Predicate),
                                         C.POINTER(abi.Predicate), u32p]
    lib.rdfgpu_pushdown_to_scan_predicate.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(abi.Predicate)]
    lib.rdfgpu_regex_check.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]
    if lib.rdfgpu_abi_version() != abi.ABI_VERSION:
        raise RdfGpuError(abi.ERR_INVALID, "ABI version mismatch between abi.py and librdfgpu.so")
    _LIB = lib
    return lib


def _check(status):
    if status < 0:
        raise RdfGpuError(status, load_library().rdfgpu_last_error().decode("utf-8", "replace"))
    return status


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------------
# host logic (no device access): choose_index / score / predicate algebra / push-down
# ------------------------------------------------------------------------------------------------
def _instr_array(builder_instrs):
    arr = (abi.ScanInstruction * 4)(*builder_instrs)
    return arr


def scan_score(instrs):
    """MemQuadIndex::compute_scan_score (quad_index.rs:100-130); `instrs` = 4 abi.ScanInstruction
    in the order of the index being scored."""
    return int(load_library().rdfgpu_scan_score(_instr_array(instrs)))


def choose_index(gspo_instrs, available=0b111):
    """IndexPermutations::choose_index (permutations.rs:81-96)."""
    return int(load_library().rdfgpu_choose_index(_instr_array(gspo_instrs), available))


def _pred_struct(p, keep):
    s = abi.Predicate()
    s.pred = p.kind
    if p.kind == abi.PRED_IN:
        ids = (C.c_uint32 * len(p.ids))(*p.ids)
        keep.append(ids)
        s.ids, s.n_ids = ids, len(p.ids)
    elif p.kind == abi.PRED_BETWEEN:
        s.from_, s.to = p.lo, p.hi
    elif p.kind == abi.PRED_EQUAL_TO:
        s.equal_to = 0
    return s


def _pred_from_struct(s, ids_buf):
    from .plan import MemIndexScanPredicate as P
    if s.pred == abi.PRED_FALSE:
        return P.false()
    if s.pred == abi.PRED_IN:
        if s.n_ids == 1 and not ids_buf:
            return P.in_([s.from_])
        return P.in_([ids_buf[i] for i in range(s.n_ids)])
    if s.pred == abi.PRED_BETWEEN:
        return P.between(s.from_, s.to)
    return None


def predicate_and(lhs, rhs, lib_fn=None):
    """MemIndexScanPredicate::try_and_with (scan_instructions.rs:170-210); None = not combinable."""
    keep = []
    a, b, out = _pred_struct(lhs, keep), _pred_struct(rhs, keep), abi.Predicate()
    n = max(1, len(lhs.ids or []), len(rhs.ids or []))
    ids = (C.c_uint32 * n)()
    fn = lib_fn or load_library().rdfgpu_predicate_and
    ok = fn(C.byref(a), C.byref(b), C.byref(out), ids)
    if ok < 0:
        _check(ok)
    if ok == 0:
        return None
    return _pred_from_struct(out, ids)


def pushdown_to_scan_predicate(op, value, lib_fn=None):
    """MemStoragePredicateExpr::to_scan_predicate (predicate_pushdown.rs:120-157)."""
    out = abi.Predicate()
    fn = lib_fn or load_library().rdfgpu_pushdown_to_scan_predicate
    ok = fn(op, value, C.byref(out))
    if ok < 0:
        _check(ok)
    return _pred_from_struct(out, None)


def regex_check(pattern, flags=""):
    """compile_pattern for the device (regex.rs:107-141): number of automaton positions; raises RdfGpuError
    (UNSUPPORTED) for syntax outside the supported subset."""
    p, f = (x.encode("utf-8") if isinstance(x, str) else bytes(x) for x in (pattern, flags))
    n = C.c_uint32(0)
    _check(load_library().rdfgpu_regex_check(p, len(p), f, len(f), C.byref(n)))
    return n.value


# ------------------------------------------------------------------------------------------------
# store and plans
# ------------------------------------------------------------------------------------------------
class GpuQuadStore:
    """The QuadStorage of this path (lib/extensions/src/storage/quad_storage.rs:14-78): three sorted
    u32 permutations + the typed-value table, resident in HBM."""

    def __init__(self, device=-1, batch_size=8192):
        self._lib = load_library()
        self._h = C.c_void_p()
        cfg = abi.Config(device, batch_size, 0, 0)
        _check(self._lib.rdfgpu_store_create(C.byref(cfg), C.byref(self._h)))
        self.batch_size = batch_size

    def close(self):
        if self._h:
            self._lib.rdfgpu_store_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def extend(self, g, s, p, o):
        (g, gp), (s, sp), (p, pp), (o, op) = _u32(g), _u32(s), _u32(p), _u32(o)
        n = C.c_uint64()
        _check(self._lib.rdfgpu_store_extend(self._h, gp, sp, pp, op, len(g), C.byref(n)))
        return n.value

    def extend_device(self, g_ptr, s_ptr, p_ptr, o_ptr, n_rows):
        n = C.c_uint64()
        _check(self._lib.rdfgpu_store_extend_device(self._h, g_ptr, s_ptr, p_ptr, o_ptr, n_rows, C.byref(n)))
        return n.value

    def remove(self, g, s, p, o):
        (g, gp), (s, sp), (p, pp), (o, op) = _u32(g), _u32(s), _u32(p), _u32(o)
        n = C.c_uint64()
        _check(self._lib.rdfgpu_store_remove(self._h, gp, sp, pp, op, len(g), C.byref(n)))
        return n.value

    def clear(self):
        _check(self._lib.rdfgpu_store_clear(self._h))

    def drop_tables(self):
        """rdfgpu_store_drop_tables: forget every cached join table and bump the store version (what any mutation does to the
        caches); the next execution of a plan builds what it needs again."""
        _check(self._lib.rdfgpu_store_drop_tables(self._h))

    def remove_graph(self, graph_id):
        """QuadStorage::clear_graph / the quads of drop_named_graph: every quad of one graph (0 = default graph)."""
        n = C.c_uint64()
        _check(self._lib.rdfgpu_store_remove_graph(self._h, int(graph_id), C.byref(n)))
        return n.value

    def __len__(self):
        n = C.c_uint64()
        _check(self._lib.rdfgpu_store_len(self._h, C.byref(n)))
        return n.value

    def set_strings(self, offsets, heap):
        """Lexical forms of the string ids: id i = heap[offsets[i]:offsets[i+1]] (UTF-8); len(offsets) = n_ids + 1."""
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        heap = np.frombuffer(bytes(heap), dtype=np.uint8) if not isinstance(heap, np.ndarray) else np.ascontiguousarray(heap, dtype=np.uint8)
        _check(self._lib.rdfgpu_store_set_strings(self._h, offsets.ctypes.data_as(C.c_void_p), len(offsets) - 1,
                                                   heap.ctypes.data_as(C.c_void_p), len(heap)))

    def set_typed_values(self, values, decimals=None):
        """values: numpy structured array / bytes of rdfgpu_typed_value, index = object id."""
        values = np.ascontiguousarray(values)
        assert values.dtype.itemsize == 16, "rdfgpu_typed_value is 16 bytes"
        dec = np.ascontiguousarray(decimals if decimals is not None else np.zeros((0, 2), np.int64), dtype=np.int64).reshape(-1, 2)   # (lo, hi) per i128
        _check(self._lib.rdfgpu_store_set_typed_values(
            self._h, values.ctypes.data_as(C.c_void_p), len(values), dec.ctypes.data_as(C.c_void_p), len(dec)))

    def read_index(self, components):
        n = C.c_uint64()
        _check(self._lib.rdfgpu_store_read_index(self._h, components, None, None, None, None, 0, C.byref(n)))
        cols = [np.empty(n.value, np.uint32) for _ in range(4)]
        ptrs = [c.ctypes.data_as(C.c_void_p) for c in cols]
        _check(self._lib.rdfgpu_store_read_index(self._h, components, *ptrs, n.value, C.byref(n)))
        return cols

    def set_option(self, name, value=1):
        """Engine option of this store (plans compiled afterwards copy it); `name` as in abi.OPTION_NAMES, with or
        without the RDFGPU_ prefix of its environment variable."""
        _check(self._lib.rdfgpu_store_set_option(self._h, abi.OPTIONS[name.replace("RDFGPU_", "", 1)], int(value)))
        return self

    def get_option(self, name):
        v = C.c_uint64()
        _check(self._lib.rdfgpu_store_get_option(self._h, abi.OPTIONS[name.replace("RDFGPU_", "", 1)], C.byref(v)))
        return v.value

    def plan(self, description):
        return GpuPlan(self, description)


class Comm:
    """A communicator of the multi-GPU exchange steps (include/rdfgpu.h section 6): one per process and GPU.
    `unique_id` (bytes from Comm.unique_id() on one rank, shipped by the launcher) selects RCCL over xGMI;
    `host_alltoallv(blocks) -> blocks` (a list of `world` uint8 arrays to send -> the list received) selects the
    host-staged transport (several ranks on one GPU, tests)."""

    def __init__(self, rank, world, device=-1, unique_id=None, host_alltoallv=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        if (unique_id is None) == (host_alltoallv is None):
            raise ValueError("Comm: give either the RCCL unique id or a host all-to-all function")
        if unique_id is not None:
            buf = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
            _check(self._lib.rdfgpu_comm_create(buf, rank, world, device, C.byref(self._h)))
        else:
            def wire(_ctx, send, send_bytes, recv, recv_bytes):
                try:
                    sb = [send_bytes[p] for p in range(world)]
                    rb = [recv_bytes[p] for p in range(world)]
                    blocks, at = [], 0
                    for p in range(world):
                        blocks.append(np.ctypeslib.as_array(C.cast(send + at, C.POINTER(C.c_uint8)), (sb[p],)).copy() if sb[p] else np.zeros(0, np.uint8))
                        at += sb[p]
                    got = host_alltoallv(blocks)
                    at = 0
                    for p in range(world):
                        b = np.ascontiguousarray(got[p], dtype=np.uint8)
                        if len(b) != rb[p]:
                            return 1
                        if rb[p]:
                            C.memmove(recv + at, b.ctypes.data, rb[p])
                        at += rb[p]
                    return 0
                except Exception:      # no exception may cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 2
            self._wire = abi.HOST_ALLTOALLV_FN(wire)      # keep the trampoline alive
            _check(self._lib.rdfgpu_comm_create_host(rank, world, device, self._wire, None, C.byref(self._h)))

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
        _check(load_library().rdfgpu_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if self._h:
            self._lib.rdfgpu_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, fn, device_ptrs, n_rows, *extra):
        n_cols = len(device_ptrs)
        cols = (C.c_void_p * n_cols)(*device_ptrs)
        out = (C.c_void_p * n_cols)()
        rows = C.c_uint64()
        _check(fn(self._h, cols, n_cols, n_rows, *extra, out, C.byref(rows)))
        return [out[i] or 0 for i in range(n_cols)], rows.value

    def allgatherv(self, device_ptrs, n_rows):
        """Every rank's rows to every rank (rank order): ([device pointers of the received columns], rows); the columns belong
        to the communicator until its next exchange."""
        return self._call(self._lib.rdfgpu_exchange_allgatherv, device_ptrs, n_rows)

    def repartition(self, device_ptrs, n_rows, key_col):
        """Row i goes to rank shard_of(cols[key_col][i]): the table re-sharded by the key of the next join."""
        return self._call(self._lib.rdfgpu_exchange_repartition, device_ptrs, n_rows, C.c_uint32(key_col))


class NTriples:
    """rdfgpu_ntriples_*: N-Triples text -> object ids on the device (the per-triple half of the reference's bulk load,
    store.rs:477-493 + object_id_mapping.rs:106-116).  `terms()` = the distinct terms as written, term t has id first_id + t;
    `columns()` = device pointers of the s / p / o id columns (file order)."""

    def __init__(self, text, first_id=1, device=-1):
        self._lib = load_library()
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _check(self._lib.rdfgpu_ntriples_parse(device, data, len(data), first_id, C.byref(h)))
        self._h = h
        self.first_id = first_id
        nt, nm, nb = C.c_uint64(), C.c_uint32(), C.c_uint64()
        _check(self._lib.rdfgpu_ntriples_info(self._h, C.byref(nt), C.byref(nm), C.byref(nb)))
        self.n_triples, self.n_terms, self.term_bytes = nt.value, nm.value, nb.value

    def terms(self):
        off = np.zeros(self.n_terms + 1, dtype=np.uint64)
        buf = np.zeros(max(1, self.term_bytes), dtype=np.uint8)
        _check(self._lib.rdfgpu_ntriples_terms(self._h, off.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p)))
        raw = buf.tobytes()
        return [raw[int(off[t]):int(off[t + 1])] for t in range(self.n_terms)]

    def decoded(self):
        """rdfgpu_ntriples_decoded: per distinct term (id first_id + t) a tuple (kind, lexical form, suffix) — escapes decoded, the
        language tag in lower case / the datatype IRI as suffix — and the typed-value rows the device derived (TV_DTYPE array; flags &
        abi.TVF_NEEDS_HOST: the host parses that literal) with the decimals' high words."""
        nl, ns = C.c_uint64(), C.c_uint64()
        _check(self._lib.rdfgpu_ntriples_decoded_info(self._h, C.byref(nl), C.byref(ns)))
        n = self.n_terms
        kind = np.zeros(max(1, n), np.uint8)
        lo, so = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64)
        lex, sfx = np.zeros(max(1, nl.value), np.uint8), np.zeros(max(1, ns.value), np.uint8)
        typed, hi = np.zeros(max(1, n), TV_DTYPE), np.zeros(max(1, n), np.int64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _check(self._lib.rdfgpu_ntriples_decoded(self._h, p(kind), p(lo), p(lex), p(so), p(sfx), p(typed), p(hi)))
        lb, sb = lex.tobytes(), sfx.tobytes()
        terms = [(int(kind[t]), lb[int(lo[t]):int(lo[t + 1])], sb[int(so[t]):int(so[t + 1])]) for t in range(n)]
        return terms, typed[:n], hi[:n]

    def columns(self):
        s, p, o = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self._lib.rdfgpu_ntriples_columns(self._h, C.byref(s), C.byref(p), C.byref(o)))
        return s.value or 0, p.value or 0, o.value or 0

    def close(self):
        if self._h:
            self._lib.rdfgpu_ntriples_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_of(object_id, world):
    """rdfgpu_shard_of: the shard of an object id among `world` ranks."""
    return int(load_library().rdfgpu_shard_of(int(object_id), int(world)))


TV_DTYPE = np.dtype([("lo", "<i8"), ("aux", "<u4"), ("tag", "u1"), ("flags", "u1"), ("reserved", "<u2")])


class GpuPlan:
    """A compiled operator tree (DataSourceExec / FilterExec / HashJoinExec / CrossJoinExec ...)."""

    def __init__(self, store, description):
        self._lib = store._lib
        self._store = store
        self._desc = description
        self._h = C.c_void_p()
        self._keep = []
        _check(self._lib.rdfgpu_plan_compile(store._h, C.byref(description.desc), C.byref(self._h)))

    def close(self):
        if self._h:
            self._lib.rdfgpu_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind_table(self, slot, device_ptrs, n_rows):
        arr = (C.c_void_p * len(device_ptrs))(*device_ptrs)
        _check(self._lib.rdfgpu_plan_bind_table(self._h, slot, arr, len(device_ptrs), n_rows))

    def execute(self):
        _check(self._lib.rdfgpu_plan_execute(self._h))
        return self

    def result_info(self):
        n, c = C.c_uint64(), C.c_uint32()
        _check(self._lib.rdfgpu_plan_result_info(self._h, C.byref(n), C.byref(c)))
        return n.value, c.value

    def result_device(self):
        n, c = self.result_info()
        ptrs = (C.c_void_p * max(1, c))()
        _check(self._lib.rdfgpu_plan_result_device(self._h, ptrs, c))
        return [ptrs[i] or 0 for i in range(c)], n

    def fetch(self):
        """Whole result as host numpy columns."""
        n, c = self.result_info()
        cols = [np.empty(n, np.uint32) for _ in range(c)]
        ptrs = (C.c_void_p * max(1, c))(*[x.ctypes.data_as(C.c_void_p) for x in cols])
        _check(self._lib.rdfgpu_plan_fetch(self._h, ptrs, c))
        return cols

    def batches(self):
        """Drains the Arrow batch stream; yields pyarrow StructArrays of UInt32 children."""
        import pyarrow as pa
        _check(self._lib.rdfgpu_plan_rewind(self._h))
        while True:
            arr, sch = abi.ArrowArray(), abi.ArrowSchema()
            st = _check(self._lib.rdfgpu_plan_next(self._h, C.byref(arr), C.byref(sch)))
            if st == abi.END:
                return
            yield pa.Array._import_from_c(C.addressof(arr), C.addressof(sch))

    def decode_terms(self, col, first_row=0, n_rows=None):
        """ENC_PT of rows [first_row, first_row + n_rows) of result column `col` (object_id_mapping.rs:331-374), decoded on the
        device: a pyarrow StructArray<term_type: uint8, value: utf8, tag: uint8, aux: uint32>, null where the id is null."""
        import pyarrow as pa
        if n_rows is None:
            n_rows = self.result_info()[0] - first_row
        arr, sch = abi.ArrowArray(), abi.ArrowSchema()
        _check(self._lib.rdfgpu_plan_decode_terms(self._h, col, first_row, n_rows, C.byref(arr), C.byref(sch)))
        return pa.Array._import_from_c(C.addressof(arr), C.addressof(sch))

    def metrics(self):
        m = abi.Metrics()
        _check(self._lib.rdfgpu_plan_metrics(self._h, C.byref(m)))
        return m

    def set_option(self, name, value=1):
        """Engine option of this plan only (its later executions)."""
        _check(self._lib.rdfgpu_plan_set_option(self._h, abi.OPTIONS[name.replace("RDFGPU_", "", 1)], int(value)))
        return self

    @staticmethod
    def _filters(filters):
        """[("true",) | ("unsupported",) | ("binary", var_slot, abi.OP_*, object_id) | ("between", var_slot, from, to)] -> ctypes array"""
        arr = (abi.PushdownFilter * max(1, len(filters)))()
        for i, f in enumerate(filters):
            if f[0] == "binary":
                arr[i] = abi.PushdownFilter(abi.PUSH_BINARY, f[1], f[2], f[3], 0, 0)
            elif f[0] == "between":
                arr[i] = abi.PushdownFilter(abi.PUSH_BETWEEN, f[1], 0, 0, f[2], f[3])
            elif f[0] == "true":
                arr[i] = abi.PushdownFilter(abi.PUSH_TRUE, 0, 0, 0, 0, 0)
            else:
                arr[i] = abi.PushdownFilter(abi.PUSH_UNSUPPORTED, 0, 0, 0, 0, 0)
        return arr

    def pushdown_filters(self, node, filters):
        """MemQuadPatternDataSource::try_pushdown_filters (pattern_data_source.rs:107-151) on a DataSourceExec node:
        returns PushedDown::Yes / No per filter; the supported ones are folded into the scan, the index chosen again."""
        arr = self._filters(filters)
        pushed = (C.c_uint8 * max(1, len(filters)))()
        _check(self._lib.rdfgpu_plan_pushdown_filters(self._h, node, arr, len(filters), pushed))
        return [bool(pushed[i]) for i in range(len(filters))]

    def set_dynamic_filters(self, node, filters):
        """The current predicates of the node's dynamic filters (scan.rs:241-261), applied by every execute until replaced."""
        _check(self._lib.rdfgpu_plan_set_dynamic_filters(self._h, node, self._filters(filters), len(filters)))
        return self

    def source_predicate(self, node, level):
        """The predicate of G,S,P,O level `level` of a DataSourceExec node as it will be scanned, as a
        plan.MemIndexScanPredicate (None = no predicate); repr() gives the reference's display (`in (2..9)`, `== 1`)."""
        from .plan import MemIndexScanPredicate as P
        s = abi.Predicate()
        _check(self._lib.rdfgpu_plan_source_predicate(self._h, node, level, C.byref(s)))
        if s.pred == abi.PRED_NONE:
            return None
        if s.pred == abi.PRED_FALSE:
            return P.false()
        if s.pred == abi.PRED_BETWEEN:
            return P.between(s.from_, s.to)
        if s.pred == abi.PRED_IN:
            return P.in_([s.from_]) if s.n_ids == 1 else P(abi.PRED_IN, ids=list(range(s.n_ids)))
        return P.equal_to(s.equal_to)

    def enable_kernel_timing(self, on=True):
        _check(self._lib.rdfgpu_plan_enable_kernel_timing(self._h, int(on)))
        return self

    def kernel_stats(self):
        """[(kernel name, launches, total ms, algorithmic bytes, rows in)] of the last execute."""
        arr = (abi.KernelStat * 32)()
        n = C.c_uint32()
        _check(self._lib.rdfgpu_plan_kernel_stats(self._h, arr, 32, C.byref(n)))
        return [(arr[i].kernel.decode(), arr[i].launches, arr[i].total_ms, arr[i].algorithmic_bytes, arr[i].rows_in)
                for i in range(n.value)]

    def selected_index(self, node):
        out = C.c_uint32()
        _check(self._lib.rdfgpu_plan_selected_index(self._h, node, C.byref(out)))
        return out.value

    def stream(self):
        out = C.c_void_p()
        _check(self._lib.rdfgpu_plan_stream(self._h, C.byref(out)))
        return out.value
