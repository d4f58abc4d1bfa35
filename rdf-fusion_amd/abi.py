"""ctypes mirror of ``include/rdfgpu.h`` (the C ABI of the hot path).

Only plain data definitions live here, so both the product binding (``engine.py``) and the
test-only CPU checker binding can describe a plan with the same structs.
"""
import ctypes as C

ABI_VERSION = 3

# status codes
OK, END = 0, 1
ERR_INVALID, ERR_DEVICE, ERR_UNSUPPORTED, ERR_OOM, ERR_NO_DEVICE = -1, -2, -3, -4, -5

# index permutations (mem_storage.rs:41-45)
GSPO, GPOS, GOSP = 0, 1, 2
INDEX_NAMES = {GSPO: "GSPO", GPOS: "GPOS", GOSP: "GOSP"}

# scan instruction kinds / predicates (scan_instructions.rs:157-166, 247-252)
TRAVERSE, SCAN = 0, 1
PRED_NONE, PRED_FALSE, PRED_IN, PRED_BETWEEN, PRED_EQUAL_TO = 0, 1, 2, 3, 4

# typed value tags (typed_value/encoding.rs:248-268)
(TV_NULL, TV_NAMED_NODE, TV_BLANK_NODE, TV_STRING, TV_BOOLEAN, TV_FLOAT, TV_DOUBLE, TV_DECIMAL,
 TV_INT, TV_INTEGER, TV_DATE_TIME, TV_TIME, TV_DATE, TV_DURATION, TV_OTHER) = range(15)
TVF_EMPTY_STRING = 1
TVF_NEEDS_HOST = 0x80
NT_IRI, NT_BNODE, NT_SIMPLE, NT_LANG, NT_TYPED = 1, 2, 3, 4, 5   # rdfgpu_ntriples_decoded: kind of a distinct term

# expression ops
(EX_COLUMN, EX_LIT_ID, EX_LIT_TV, EX_ENC_TV, EX_GT, EX_LT, EX_GEQ, EX_LEQ, EX_EQ, EX_ADD, EX_SUB,
 EX_EBV, EX_ID_EQ, EX_ID_NEQ, EX_AND, EX_OR, EX_NOT, EX_IS_COMPATIBLE, EX_BOUND, EX_BOOL_AS_TV,
 EX_LIT_BOOL, EX_NEQ, EX_REGEX, EX_CONTAINS, EX_STRSTARTS, EX_STRENDS, EX_LANG_IN, EX_REGEX_VAR,
 EX_STR, EX_LIT_STR, EX_STRLEN, EX_SUBSTR, EX_UCASE, EX_LCASE, EX_STRBEFORE, EX_STRAFTER) = range(1, 37)

# plan nodes
(NODE_DATA_SOURCE, NODE_FILTER, NODE_HASH_JOIN, NODE_CROSS_JOIN, NODE_NESTED_LOOP_JOIN,
 NODE_PROJECTION, NODE_TABLE, NODE_TOPK, NODE_UNION, NODE_CLOSURE) = range(1, 11)
SORT_BY_ID, SORT_BY_TERM, SORT_BY_DOUBLE = 0, 1, 2
JOIN_INNER, JOIN_LEFT = 0, 1
MAX_KEYS = 4
MAX_COLUMNS = 16
NO_PROJECTION = 0xFFFFFFFF

OP_EQ, OP_GT, OP_GTEQ, OP_LT, OP_LTEQ = range(5)

# engine options (include/rdfgpu.h section 4b); the environment variable of option X is RDFGPU_X, read once per process
OPTION_NAMES = ["FORCE_GENERIC_VM", "NO_JOIN_REORDER", "NO_SPECULATION", "NO_FIRST_RUN_SPECULATION", "NO_STRING_VERDICTS",
                "NO_TABLE_CACHE", "NO_INDEX_JOIN", "NO_CHAIN_FUSION", "NO_VALUE_TABLES", "NO_RANGE_INDEX", "NO_FILTER_FUSION",
                "NO_LDS_JOIN", "NO_GLOBAL_TABLE_JOIN", "NO_DIRECT_TABLE", "NO_BAND_JOIN", "NO_PARTITIONED_JOIN", "NO_VALUE_VERDICTS", "NO_PRIMING", "NO_ORDERED_JOIN", "NO_BAND_PACK16", "NO_RUN_COPY", "NO_RANGE_PARTITION",
                "LDS_MAX_BUILD", "CSR_ROW_LANES_LOG2", "JOIN_WAVE_Q", "PARTITION_MIN_BUILD", "PARTITION_TWO_PASS_ROWS", "NO_OWN_PARTITION_PASS", "NO_BAND_COMPACT", "NO_PROBE_OUTER_JOIN", "NO_STREAM_JOIN", "PARTITION_ROWS", "PARTITION_SLOTS"]
OPTIONS = {name: i for i, name in enumerate(OPTION_NAMES)}


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("batch_size", C.c_uint32), ("flags", C.c_uint32),
                ("reserved", C.c_uint32)]


class TypedValue(C.Structure):
    _fields_ = [("lo", C.c_int64), ("aux", C.c_uint32), ("tag", C.c_uint8), ("flags", C.c_uint8),
                ("reserved", C.c_uint16)]


class ScanInstruction(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("pred", C.c_uint8), ("reserved", C.c_uint16),
                ("var", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32)]


class ExprNode(C.Structure):
    _fields_ = [("op", C.c_uint8), ("tag", C.c_uint8), ("flags", C.c_uint8), ("reserved", C.c_uint8),
                ("u", C.c_uint32), ("lo", C.c_int64), ("hi", C.c_int64)]


class PlanNode(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("left", C.c_int32), ("right", C.c_int32),
                ("join_type", C.c_uint32), ("scan", ScanInstruction * 4), ("n_keys", C.c_uint32),
                ("left_keys", C.c_uint32 * MAX_KEYS), ("right_keys", C.c_uint32 * MAX_KEYS),
                ("expr_off", C.c_uint32), ("expr_len", C.c_uint32), ("proj_off", C.c_uint32),
                ("n_proj", C.c_uint32), ("table_slot", C.c_uint32), ("table_cols", C.c_uint32)]


class Regex(C.Structure):
    _fields_ = [("pattern", C.c_char_p), ("flags", C.c_char_p), ("pattern_len", C.c_uint32), ("flags_len", C.c_uint32),
                ("pattern_id", C.c_uint32), ("reserved", C.c_uint32)]


class PlanDesc(C.Structure):
    _fields_ = [("nodes", C.POINTER(PlanNode)), ("n_nodes", C.c_uint32), ("root", C.c_uint32),
                ("exprs", C.POINTER(ExprNode)), ("n_exprs", C.c_uint32),
                ("pool", C.POINTER(C.c_uint32)), ("n_pool", C.c_uint32), ("flags", C.c_uint32),
                ("regexes", C.POINTER(Regex)), ("n_regexes", C.c_uint32), ("reserved", C.c_uint32)]


class Metrics(C.Structure):
    _fields_ = [("output_rows", C.c_uint64), ("input_rows", C.c_uint64),
                ("intermediate_rows", C.c_uint64), ("device_bytes", C.c_uint64),
                ("elapsed_compute_ms", C.c_double), ("kernels_launched", C.c_uint32),
                ("host_syncs", C.c_uint32), ("exact_reruns", C.c_uint32), ("device_mallocs", C.c_uint32),
                ("device_malloc_ms", C.c_double), ("tables_built", C.c_uint32), ("reserved", C.c_uint32)]


class KernelStat(C.Structure):
    _fields_ = [("kernel", C.c_char_p), ("launches", C.c_uint32), ("reserved", C.c_uint32),
                ("total_ms", C.c_double), ("algorithmic_bytes", C.c_uint64), ("rows_in", C.c_uint64)]


class Predicate(C.Structure):
    _fields_ = [("pred", C.c_uint32), ("from_", C.c_uint32), ("to", C.c_uint32),
                ("ids", C.POINTER(C.c_uint32)), ("n_ids", C.c_uint32), ("equal_to", C.c_uint32)]


PUSH_UNSUPPORTED, PUSH_TRUE, PUSH_BINARY, PUSH_BETWEEN = range(4)


class PushdownFilter(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("var", C.c_uint32), ("op", C.c_uint32), ("value", C.c_uint32),
                ("from_", C.c_uint32), ("to", C.c_uint32)]


class ArrowSchema(C.Structure):
    pass


ArrowSchema._fields_ = [
    ("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
    ("n_children", C.c_int64), ("children", C.POINTER(C.POINTER(ArrowSchema))),
    ("dictionary", C.POINTER(ArrowSchema)), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowArray(C.Structure):
    pass


ArrowArray._fields_ = [
    ("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
    ("n_children", C.c_int64), ("buffers", C.POINTER(C.c_void_p)),
    ("children", C.POINTER(C.POINTER(ArrowArray))), ("dictionary", C.POINTER(ArrowArray)),
    ("release", C.c_void_p), ("private_data", C.c_void_p)]

assert C.sizeof(TypedValue) == 16
assert C.sizeof(ScanInstruction) == 16
assert C.sizeof(ExprNode) == 24

# every symbol include/rdfgpu.h declares (the CPU test-suite checks the .so exports them all)
EXPORTED_SYMBOLS = [
    "rdfgpu_last_error", "rdfgpu_abi_version",
    "rdfgpu_store_create", "rdfgpu_store_destroy", "rdfgpu_store_extend", "rdfgpu_store_extend_device",
    "rdfgpu_store_remove", "rdfgpu_store_clear", "rdfgpu_store_drop_tables", "rdfgpu_store_remove_graph", "rdfgpu_store_len", "rdfgpu_store_set_typed_values", "rdfgpu_store_set_strings",
    "rdfgpu_store_read_index",
    "rdfgpu_plan_compile", "rdfgpu_plan_destroy", "rdfgpu_plan_bind_table", "rdfgpu_plan_execute",
    "rdfgpu_plan_result_info", "rdfgpu_plan_result_device", "rdfgpu_plan_fetch", "rdfgpu_plan_next",
    "rdfgpu_plan_rewind", "rdfgpu_plan_decode_terms", "rdfgpu_ntriples_parse", "rdfgpu_ntriples_info", "rdfgpu_ntriples_terms", "rdfgpu_ntriples_decoded_info", "rdfgpu_ntriples_decoded", "rdfgpu_ntriples_columns", "rdfgpu_ntriples_destroy", "rdfgpu_plan_metrics", "rdfgpu_plan_selected_index", "rdfgpu_plan_stream",
    "rdfgpu_plan_enable_kernel_timing", "rdfgpu_plan_kernel_stats",
    "rdfgpu_plan_pushdown_filters", "rdfgpu_plan_set_dynamic_filters", "rdfgpu_plan_source_predicate",
    "rdfgpu_store_set_option", "rdfgpu_store_get_option", "rdfgpu_plan_set_option", "rdfgpu_option_name",
    "rdfgpu_scan_score", "rdfgpu_choose_index", "rdfgpu_predicate_and",
    "rdfgpu_pushdown_to_scan_predicate", "rdfgpu_regex_check",
    "rdfgpu_comm_unique_id", "rdfgpu_comm_create", "rdfgpu_comm_create_host", "rdfgpu_comm_destroy",
    "rdfgpu_exchange_allgatherv", "rdfgpu_exchange_repartition", "rdfgpu_shard_of",
]
COMM_ID_BYTES = 128
HOST_ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
