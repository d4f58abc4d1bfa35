"""LUBM-shaped synthetic data and the Q9 + OPTIONAL + REGEX plan (BASELINE.json configs[4]).

The reference holds no LUBM generator or query ("config 5 has no counterpart in the reference"); shape and
fan-outs follow the public LUBM / UBA description: universities of 15-25 departments; per department 30-42 faculty,
8-14 undergraduates and 3-4 graduate students per faculty member, 1-2 courses and 1-2 graduate courses per faculty
member, ~10 publications per faculty member; undergraduates take 2-4 courses, graduate students 1-3, every graduate
student and a fifth of the undergraduates have an advisor; superclass types (Student, Faculty, Course) are
materialised, as LUBM Q9 needs them.  ~134 k triples per university (LUBM-8000 ~ 1.07 G).  Everything is vectorised
numpy; ids are dense from 1; default graph only.

Strings: names come from a small shared pool ("GraduateStudent17" is the same literal in every department, as in
LUBM); e-mail addresses are one literal per person, fixed-width ("GS00017@D03.U00042.edu") so that the heap is built by
digit arithmetic instead of 10^8 Python strings.  The typed-value `lo` of an e-mail literal follows id order, not
lexical order: this generator is for joins / REGEX, not for ordering e-mail addresses.
"""
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .engine import TV_DTYPE
from .plan import PlanBuilder, quad_pattern, col, ENC_TV, EBV, REGEX

PREDICATES = ["rdf:type", "ub:name", "ub:emailAddress", "ub:memberOf", "ub:worksFor", "ub:subOrganizationOf", "ub:teacherOf",
              "ub:takesCourse", "ub:advisor", "ub:undergraduateDegreeFrom", "ub:degreeFrom", "ub:publicationAuthor"]
CLASSES = ["ub:University", "ub:Department", "ub:Faculty", "ub:FullProfessor", "ub:AssociateProfessor", "ub:AssistantProfessor",
           "ub:Lecturer", "ub:Student", "ub:UndergraduateStudent", "ub:GraduateStudent", "ub:Course", "ub:GraduateCourse",
           "ub:Publication"]


@dataclass
class LubmDataset:
    n_universities: int
    g: np.ndarray
    s: np.ndarray
    p: np.ndarray
    o: np.ndarray
    typed_values: np.ndarray
    pred: dict
    cls: dict
    n_ids: int
    str_offsets: np.ndarray = None      # u64[n_ids + 1], set when strings were requested
    str_heap: bytes = None
    counts: dict = field(default_factory=dict)

    @property
    def n_triples(self):
        return len(self.s)


def _segments(counts):
    """start offset of every segment and, per element, its segment index"""
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
    return starts, np.repeat(np.arange(len(counts), dtype=np.int64), counts)


def _pick(rng, seg_start, seg_len, owner):
    """one uniformly random member of segment owner[i], for every i"""
    return seg_start[owner] + (rng.random(len(owner)) * seg_len[owner]).astype(np.int64)


def _digits(values, width):
    """(n, width) uint8 of zero-padded decimal digits"""
    v = np.asarray(values, dtype=np.int64)
    out = np.empty((len(v), width), dtype=np.uint8)
    for k in range(width):
        out[:, width - 1 - k] = 48 + (v // 10 ** k) % 10
    return out


def generate(n_universities, seed=None, strings=True):
    U = int(n_universities)
    rng = np.random.default_rng(U if seed is None else seed)
    next_id = [1]

    def take(n):
        b = next_id[0]
        next_id[0] += int(n)
        return b
    pred = {n: take(1) for n in PREDICATES}
    cls = {n: take(1) for n in CLASSES}
    n_dept_u = rng.integers(15, 26, U)
    D = int(n_dept_u.sum())
    dept_start_u, univ_of_dept = _segments(n_dept_u)
    n_fac_d = rng.integers(30, 43, D)
    n_ug_d = n_fac_d * rng.integers(8, 15, D)
    n_gs_d = (n_fac_d * rng.uniform(3.0, 4.0, D)).astype(np.int64)
    F, NUG, NGS = int(n_fac_d.sum()), int(n_ug_d.sum()), int(n_gs_d.sum())
    fac_start_d, dept_of_fac = _segments(n_fac_d)
    ug_start_d, dept_of_ug = _segments(n_ug_d)
    gs_start_d, dept_of_gs = _segments(n_gs_d)
    n_c_f, n_gc_f = rng.integers(1, 3, F), rng.integers(1, 3, F)
    C, GC = int(n_c_f.sum()), int(n_gc_f.sum())
    _, fac_of_c = _segments(n_c_f)
    _, fac_of_gc = _segments(n_gc_f)
    n_c_d = np.bincount(dept_of_fac[fac_of_c], minlength=D)
    n_gc_d = np.bincount(dept_of_fac[fac_of_gc], minlength=D)
    c_start_d = np.concatenate([[0], np.cumsum(n_c_d)[:-1]])
    gc_start_d = np.concatenate([[0], np.cumsum(n_gc_d)[:-1]])
    n_pub_f = rng.integers(5, 16, F)
    NP = int(n_pub_f.sum())
    _, fac_of_pub = _segments(n_pub_f)

    univ_base, dept_base, fac_base = take(U), take(D), take(F)
    ug_base, gs_base, c_base, gc_base, pub_base = take(NUG), take(NGS), take(C), take(GC), take(NP)
    # literal pools: names by (kind, index within the department), e-mails one per person
    idx_fac = np.arange(F) - fac_start_d[dept_of_fac]
    idx_ug = np.arange(NUG) - ug_start_d[dept_of_ug]
    idx_gs = np.arange(NGS) - gs_start_d[dept_of_gs]
    idx_c = np.arange(C) - c_start_d[dept_of_fac[fac_of_c]]
    idx_gc = np.arange(GC) - gc_start_d[dept_of_fac[fac_of_gc]]
    rank_f = rng.integers(0, 4, F)                     # full / associate / assistant professor, lecturer
    fac_kinds = ["FullProfessor", "AssociateProfessor", "AssistantProfessor", "Lecturer"]
    kinds = fac_kinds + ["UndergraduateStudent", "GraduateStudent", "Course", "GraduateCourse", "Department"]
    width = {k: 0 for k in kinds}
    for k, m in zip(fac_kinds, [int(idx_fac[rank_f == r].max(initial=0)) + 1 for r in range(4)]):
        width[k] = m
    width.update({"UndergraduateStudent": int(idx_ug.max(initial=0)) + 1, "GraduateStudent": int(idx_gs.max(initial=0)) + 1,
                  "Course": int(idx_c.max(initial=0)) + 1, "GraduateCourse": int(idx_gc.max(initial=0)) + 1, "Department": 26})
    name_base, pool = {}, []
    for k in kinds:
        name_base[k] = take(width[k])
        pool += [f"{k}{i}" for i in range(width[k])]
    pool_base = name_base[kinds[0]]
    n_people = F + NUG + NGS
    email_base = take(n_people)
    n_ids = next_id[0]

    S, P, O = [], [], []

    def emit(s, p, o):
        s = np.asarray(s, dtype=np.uint32)
        S.append(s)
        P.append(np.full(len(s), p, dtype=np.uint32))
        O.append(np.broadcast_to(np.asarray(o, dtype=np.uint32), s.shape) if np.ndim(o) == 0 else np.asarray(o, dtype=np.uint32))

    univ = univ_base + np.arange(U)
    dept = dept_base + np.arange(D)
    fac = fac_base + np.arange(F)
    ug = ug_base + np.arange(NUG)
    gs = gs_base + np.arange(NGS)
    crs = c_base + np.arange(C)
    gcrs = gc_base + np.arange(GC)
    pub = pub_base + np.arange(NP)
    T = pred["rdf:type"]
    emit(univ, T, cls["ub:University"])
    emit(dept, T, cls["ub:Department"])
    emit(dept, pred["ub:subOrganizationOf"], univ_base + univ_of_dept)
    emit(dept, pred["ub:name"], name_base["Department"] + (np.arange(D) - dept_start_u[univ_of_dept]))
    # faculty
    emit(fac, T, np.array([cls["ub:FullProfessor"], cls["ub:AssociateProfessor"], cls["ub:AssistantProfessor"], cls["ub:Lecturer"]])[rank_f])
    emit(fac, T, cls["ub:Faculty"])
    emit(fac, pred["ub:worksFor"], dept_base + dept_of_fac)
    emit(fac, pred["ub:name"], np.array([name_base[k] for k in fac_kinds])[rank_f] + idx_fac)
    for _ in range(3):
        emit(fac, pred["ub:degreeFrom"], univ_base + rng.integers(0, U, F))
    emit(fac_base + fac_of_c, pred["ub:teacherOf"], crs)
    emit(fac_base + fac_of_gc, pred["ub:teacherOf"], gcrs)
    # courses
    emit(crs, T, cls["ub:Course"])
    emit(crs, pred["ub:name"], name_base["Course"] + idx_c)
    emit(gcrs, T, cls["ub:GraduateCourse"])
    emit(gcrs, T, cls["ub:Course"])
    emit(gcrs, pred["ub:name"], name_base["GraduateCourse"] + idx_gc)
    # undergraduates
    emit(ug, T, cls["ub:UndergraduateStudent"])
    emit(ug, T, cls["ub:Student"])
    emit(ug, pred["ub:memberOf"], dept_base + dept_of_ug)
    emit(ug, pred["ub:name"], name_base["UndergraduateStudent"] + idx_ug)
    for k in range(4):
        sel = np.arange(NUG) if k < 2 else np.nonzero(rng.random(NUG) < 0.5)[0]
        emit(ug[sel], pred["ub:takesCourse"], c_base + _pick(rng, c_start_d, n_c_d, dept_of_ug[sel]))
    adv = np.nonzero(rng.random(NUG) < 0.2)[0]
    emit(ug[adv], pred["ub:advisor"], fac_base + _pick(rng, fac_start_d, n_fac_d, dept_of_ug[adv]))
    # graduate students
    emit(gs, T, cls["ub:GraduateStudent"])
    emit(gs, T, cls["ub:Student"])
    emit(gs, pred["ub:memberOf"], dept_base + dept_of_gs)
    emit(gs, pred["ub:name"], name_base["GraduateStudent"] + idx_gs)
    emit(gs, pred["ub:undergraduateDegreeFrom"], univ_base + rng.integers(0, U, NGS))
    for k in range(3):
        sel = np.arange(NGS) if k < 1 else np.nonzero(rng.random(NGS) < 0.5)[0]
        emit(gs[sel], pred["ub:takesCourse"], gc_base + _pick(rng, gc_start_d, n_gc_d, dept_of_gs[sel]))
    emit(gs, pred["ub:advisor"], fac_base + _pick(rng, fac_start_d, n_fac_d, dept_of_gs))
    # publications
    emit(pub, T, cls["ub:Publication"])
    emit(pub, pred["ub:publicationAuthor"], fac_base + fac_of_pub)
    co = np.nonzero(rng.random(NP) < 0.6)[0]
    emit(pub[co], pred["ub:publicationAuthor"], gs_base + _pick(rng, gs_start_d, n_gs_d, dept_of_fac[fac_of_pub[co]]))
    # e-mail addresses: 60 % of the students, every faculty member
    people = np.concatenate([fac, ug, gs])
    has_mail = np.concatenate([np.ones(F, bool), rng.random(NUG + NGS) < 0.6])
    emit(people[has_mail], pred["ub:emailAddress"], (email_base + np.arange(n_people))[has_mail])

    s = np.concatenate(S)
    p = np.concatenate(P)
    o = np.concatenate(O)
    del S, P, O
    g = np.zeros(len(s), dtype=np.uint32)

    tv = np.zeros(n_ids, dtype=TV_DTYPE)
    tv["tag"][1:] = abi.TV_NAMED_NODE
    tv["lo"][1:] = np.arange(1, n_ids)
    order = {t: r for r, t in enumerate(sorted(pool))}
    tv["tag"][pool_base:pool_base + len(pool)] = abi.TV_STRING
    tv["lo"][pool_base:pool_base + len(pool)] = [order[t] for t in pool]
    tv["tag"][email_base:email_base + n_people] = abi.TV_STRING
    tv["lo"][email_base:email_base + n_people] = len(pool) + np.arange(n_people)

    offsets = heap = None
    if strings:
        pool_bytes = [t.encode() for t in pool]
        kind_code = np.concatenate([np.array([b"FP", b"AP", b"AS", b"LE"], dtype="S2")[rank_f].view(np.uint8).reshape(-1, 2),
                                    np.tile(np.frombuffer(b"UG", np.uint8), (NUG, 1)), np.tile(np.frombuffer(b"GS", np.uint8), (NGS, 1))])
        person_dept = np.concatenate([dept_of_fac, dept_of_ug, dept_of_gs])
        person_idx = np.concatenate([idx_fac, idx_ug, idx_gs])
        mail = np.empty((n_people, 22), dtype=np.uint8)          # KK00000@D00.U00000.edu
        mail[:, 0:2] = kind_code
        mail[:, 2:7] = _digits(person_idx, 5)
        mail[:, 7:9] = np.frombuffer(b"@D", np.uint8)
        mail[:, 9:11] = _digits(person_dept - dept_start_u[univ_of_dept[person_dept]], 2)
        mail[:, 11:13] = np.frombuffer(b".U", np.uint8)
        mail[:, 13:18] = _digits(univ_of_dept[person_dept], 5)
        mail[:, 18:22] = np.frombuffer(b".edu", np.uint8)
        offsets = np.zeros(n_ids + 1, dtype=np.uint64)
        lens = np.zeros(n_ids, dtype=np.uint64)
        lens[pool_base:pool_base + len(pool)] = [len(b) for b in pool_bytes]
        lens[email_base:email_base + n_people] = 22
        offsets[1:] = np.cumsum(lens)
        heap = b"".join(pool_bytes) + mail.tobytes()             # the pool's ids precede the e-mail ids
    counts = {"universities": U, "departments": D, "faculty": F, "undergraduates": NUG, "graduate_students": NGS,
              "courses": C + GC, "publications": NP, "strings": len(pool) + n_people}
    return LubmDataset(U, g, s, p, o, tv, pred, cls, n_ids, offsets, heap, counts)


def q9_optional_regex_plan(ds, pattern="^GraduateStudent1", flags=""):
    """LUBM Q9 (the student - advisor - course triangle over the materialised superclasses) + the student's name with a
    REGEX FILTER + OPTIONAL e-mail address, as the chain of HashJoinExecs DataFusion plans for it (inner joins on the
    shared variables, the two-key join closing the triangle, FilterExec, then the left-outer join of the OPTIONAL).
    Output: (x, y, z, name, email?)."""
    pr, cl = ds.pred, ds.cls
    pb = PlanBuilder()
    node = pb.hash_join(pb.data_source(quad_pattern("x", pr["ub:advisor"], "y")),
                        pb.data_source(quad_pattern("x", pr["rdf:type"], cl["ub:Student"])), on=[(0, 0)], projection=[0, 1])
    node = pb.hash_join(node, pb.data_source(quad_pattern("y", pr["rdf:type"], cl["ub:Faculty"])), on=[(1, 0)], projection=[0, 1])
    node = pb.hash_join(node, pb.data_source(quad_pattern("y", pr["ub:teacherOf"], "z")), on=[(1, 0)], projection=[0, 1, 3])
    node = pb.hash_join(node, pb.data_source(quad_pattern("x", pr["ub:takesCourse"], "z")), on=[(0, 0), (2, 1)], projection=[0, 1, 2])
    node = pb.hash_join(node, pb.data_source(quad_pattern("z", pr["rdf:type"], cl["ub:Course"])), on=[(2, 0)], projection=[0, 1, 2])
    node = pb.hash_join(node, pb.data_source(quad_pattern("x", pr["ub:name"], "n")), on=[(0, 0)], projection=[0, 1, 2, 4])
    node = pb.filter(node, EBV(REGEX(ENC_TV(col(3)), pattern, flags)))
    node = pb.hash_join(node, pb.data_source(quad_pattern("x", pr["ub:emailAddress"], "e")), on=[(0, 0)], join_type=abi.JOIN_LEFT,
                        projection=[0, 1, 2, 3, 5])
    return pb.build(node)


def q9_sharded_stages(ds, pattern="^GraduateStudent1", flags=""):
    """LUBM Q9 + REGEX + OPTIONAL over triples sharded by hash(subject) (rdfgpu_shard_of): the triangle student x -> advisor y ->
    course z -> x joins on three different subjects, so the intermediate bindings are re-sharded by the key of the next
    join (rdfgpu_exchange_repartition) three times.  Returns [(plan description, repartition column or None), ...]: stage k
    reads the re-sharded output of stage k - 1 as bound table 0 (stage 0 reads no table); inner joins commute, so the
    union of the ranks' last outputs is q9_optional_regex_plan's answer as a multiset.  Output: (x, y, z, name, email?)."""
    pr, cl = ds.pred, ds.cls
    stages = []
    pb = PlanBuilder()          # local by x: (x advisor y) JOIN (x a Student)
    node = pb.hash_join(pb.data_source(quad_pattern("x", pr["ub:advisor"], "y")),
                        pb.data_source(quad_pattern("x", pr["rdf:type"], cl["ub:Student"])), on=[(0, 0)], projection=[0, 1])
    stages.append((pb.build(node), 1))                                   # -> re-shard by y
    pb = PlanBuilder()          # local by y: JOIN (y a Faculty) JOIN (y teacherOf z)
    node = pb.hash_join(pb.table(0, 2, ["x", "y"]), pb.data_source(quad_pattern("y", pr["rdf:type"], cl["ub:Faculty"])), on=[(1, 0)], projection=[0, 1])
    node = pb.hash_join(node, pb.data_source(quad_pattern("y", pr["ub:teacherOf"], "z")), on=[(1, 0)], projection=[0, 1, 3])
    stages.append((pb.build(node), 2))                                   # -> re-shard by z
    pb = PlanBuilder()          # local by z: JOIN (z a Course)
    node = pb.hash_join(pb.table(0, 3, ["x", "y", "z"]), pb.data_source(quad_pattern("z", pr["rdf:type"], cl["ub:Course"])), on=[(2, 0)], projection=[0, 1, 2])
    stages.append((pb.build(node), 0))                                   # -> re-shard by x
    pb = PlanBuilder()          # local by x: closes the triangle with the two-key join, then name + REGEX, then the OPTIONAL
    node = pb.hash_join(pb.table(0, 3, ["x", "y", "z"]), pb.data_source(quad_pattern("x", pr["ub:takesCourse"], "z")), on=[(0, 0), (2, 1)], projection=[0, 1, 2])
    node = pb.hash_join(node, pb.data_source(quad_pattern("x", pr["ub:name"], "n")), on=[(0, 0)], projection=[0, 1, 2, 4])
    node = pb.filter(node, EBV(REGEX(ENC_TV(col(3)), pattern, flags)))
    node = pb.hash_join(node, pb.data_source(quad_pattern("x", pr["ub:emailAddress"], "e")), on=[(0, 0)], join_type=abi.JOIN_LEFT,
                        projection=[0, 1, 2, 3, 5])
    stages.append((pb.build(node), None))
    return stages
